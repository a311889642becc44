#!/bin/bash
# usage: pmc_bench.sh   (SQ stall counters per kernel over a short bench run)
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_bench -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt > $R/gpurun_out/pmc_bench.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_bench/**/p_counter_collection.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:48]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, c in agg.items():
    if 'conv_tile' not in k and 'stage' not in k: continue
    wc = c.get('SQ_WAVE_CYCLES', 1)
    print(f'{k:50s} ' + ' '.join(f'{n[3:]}={v / wc:5.2f}' for n, v in sorted(c.items()) if n != 'SQ_WAVE_CYCLES'))
PY
