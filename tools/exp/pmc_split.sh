#!/bin/bash
# usage: pmc_split.sh <case idx>   (run from repo root on the GPU box)
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_split -o p -- python3 $R/tools/split_microbench.py $1 > $R/gpurun_out/pmc_split.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_split/**/p_counter_collection.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, c in agg.items():
    if 'conv_' not in k: continue
    print(k)
    wc = c.get('SQ_WAVE_CYCLES', 1)
    for name, v in sorted(c.items()):
        print(f'   {name:28s} {v:14.4g}  ({v / wc:6.3f} of WAVE_CYCLES)')
PY
