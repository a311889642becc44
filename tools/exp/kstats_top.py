"""Top kernels of a rocprofv3 --kernel-trace --stats directory.  argv: DIR [N]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print(f'total kernel time {tot / 1e6:.2f} ms in {calls} launches')
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:88]:88s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e6:8.2f} ms {float(r['AverageNs']) / 1e3:8.1f} us")
