#!/usr/bin/env python3
"""One training step (opt.zero_grad .. opt.step) of a rocprofv3 --kernel-trace CSV, by HIP stream (queue): span, busy time per queue, time no
queue is busy, and per queue the kernels in start order with the gaps in front of them.
    python tools/trace_step_streams.py <dir with *_kernel_trace.csv> [which step from the end, default -2] [min gap us to print, default 30]"""
import csv
import glob
import os
import sys


def main(path, which=-2, mingap=30.0):
    f = glob.glob(os.path.join(path, '**', '*_kernel_trace.csv'), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    name = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
    # a training forward starts with cond_fc (side stream) / a fold kernel; steps are delimited by tanh_bwd (first kernel of a backward)
    marks = [i for i, r in enumerate(rows) if 'tanh_bwd' in r['Kernel_Name']]
    if len(marks) < 3:
        raise SystemExit('fewer than 3 training steps in the trace')
    s, e = marks[which - 1], marks[which]
    step = rows[s:e]
    t0, t1 = int(step[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in step)
    print(f'step (backward k .. forward k+1): {len(step)} launches, span {(t1 - t0) / 1e6:.3f} ms')
    queues = {}
    for r in step:
        queues.setdefault(r['Queue_Id'], []).append(r)
    ev = []
    for q, rs in queues.items():
        busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
        print(f'  queue {q}: {len(rs):4d} launches, busy {busy / 1e6:7.3f} ms')
        for r in rs:
            ev.append((int(r['Start_Timestamp']), 1)); ev.append((int(r['End_Timestamp']), -1))
    ev.sort()
    depth, last, idle, both = 0, t0, 0, 0
    for t, d in ev:
        if depth == 0:
            idle += t - last
        if depth >= 2:
            both += t - last
        depth += d
        last = t
    print(f'  no queue busy {idle / 1e6:.3f} ms, two or more busy {both / 1e6:.3f} ms')
    for q, rs in sorted(queues.items(), key=lambda kv: -len(kv[1])):
        agg = {}
        for r in rs:
            a = agg.setdefault(name(r), [0, 0])
            a[0] += 1; a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        print(f'--- queue {q}: kernels by total time')
        for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            print(f'  {n:62s} x{c:4d} {d / 1e6:8.3f} ms  avg {d / c / 1e3:8.1f} us')
    for q, rs in sorted(queues.items(), key=lambda kv: -len(kv[1])):
        print(f'--- queue {q}: gaps >= {mingap} us (gap, then the kernel that follows; the kernel before it)')
        prev = None
        for r in rs:
            if prev is not None:
                gap = (int(r['Start_Timestamp']) - int(prev['End_Timestamp'])) / 1e3
                if gap >= mingap:
                    print(f'  {(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} us: gap {gap:7.1f} us before {name(r)}  (after {name(prev)})')
            prev = r


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else -2, float(sys.argv[3]) if len(sys.argv) > 3 else 30.0)
