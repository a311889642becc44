#!/usr/bin/env python3
"""Per-kernel totals of the LAST step (cond_fc to cond_fc) in a rocprofv3 --kernel-trace CSV:
    python tools/trace_step_summary.py <dir with *_kernel_trace.csv> [top N] [which]   (which = -1: last step, -6: the sixth from the end)"""
import csv
import glob
import os
import sys


def main(path, top=45, which=-1):
    f = glob.glob(os.path.join(path, '**', '*_kernel_trace.csv'), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'cond_fc' in r['Kernel_Name']]
    s, e = idx[which - 1], idx[which]
    agg = {}
    for r in rows[s:e]:
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        a = agg.setdefault(n, [0, 0])
        a[0] += 1; a[1] += d
    tot = sum(v[1] for v in agg.values())
    span = int(rows[e - 1]['End_Timestamp']) - int(rows[s]['Start_Timestamp'])
    print(f'step: {e - s} launches, span {span / 1e6:.3f} ms, sum of kernel durations {tot / 1e6:.3f} ms')
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f'{n:72s} x{c:4d} {d / 1e6:8.3f} ms  avg {d / c / 1e3:8.1f} us')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 45, int(sys.argv[3]) if len(sys.argv) > 3 else -1)
