#!/usr/bin/env python3
"""Launch-by-launch timeline of the LAST forward in a rocprofv3 --kernel-trace CSV (start offset, duration, grid, kernel):
    python tools/trace_timeline.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys


def main(path):
    f = glob.glob(os.path.join(path, '**', '*_kernel_trace.csv'), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'cond_fc' in r['Kernel_Name'] or 'cond_affine_eval' in r['Kernel_Name']]
    s, e = idx[-2], idx[-1]
    t0 = int(rows[s]['Start_Timestamp'])
    busy = 0
    for r in rows[s:e]:
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:64]
        a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        busy += b - a
        print(f'{(a - t0) / 1e3:8.1f} us +{(b - a) / 1e3:7.1f} us grid={r["Grid_Size_X"]:>8s} wg={r["Workgroup_Size_X"]:>4s} {n}')
    print(f'span {(int(rows[e - 1]["End_Timestamp"]) - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us, {e - s} launches')


if __name__ == '__main__':
    main(sys.argv[1])
