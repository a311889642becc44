#!/usr/bin/env python3
"""Timeline of one forward out of a rocprofv3 --kernel-trace CSV: start offset, duration, queue and the idle time of the queue in front
of every kernel (what the sum of the kernel durations does not show).  usage: trace_gaps.py DIR [which=-2] [marker=cond_fc]"""
import csv
import glob
import os
import sys


def main(path, which=-2, marker='cond_fc'):
    f = glob.glob(os.path.join(path, '**', '*_kernel_trace.csv'), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
    s, e = idx[which - 1], idx[which]
    t0 = int(rows[s]['Start_Timestamp'])
    last_end = {}
    busy = {}
    for r in rows[max(0, s - 40):s]:
        last_end[r['Queue_Id']] = int(r['End_Timestamp'])
    for r in rows[s:e]:
        st, en, q = int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id']
        n = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:56]
        gap = (st - last_end[q]) / 1e3 if q in last_end else float('nan')
        last_end[q] = en
        busy[q] = busy.get(q, 0.0) + (en - st) / 1e3
        print(f'{(st - t0) / 1e3:9.1f} us  {(en - st) / 1e3:8.1f} us  q={q}  idle before {gap:7.1f} us  {n}')
    print('period us', (int(rows[e]['Start_Timestamp']) - t0) / 1e3, 'busy per queue', {k: round(v, 1) for k, v in busy.items()})


if __name__ == '__main__':
    a = sys.argv[1:]
    main(a[0], int(a[1]) if len(a) > 1 else -2, a[2] if len(a) > 2 else 'cond_fc')
