#!/bin/bash
# usage: tools/exp/pmc_kernel.sh <kernel-name substring> <python script + args ...>: SQ counters of the matching kernel (mean per launch)
R=$PWD; PAT=$1; shift
O=$R/gpurun_out/pmck; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/a -o a -- python3 $R/"$@" > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/b -o b -- python3 $R/"$@" > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/c -o c -- python3 $R/"$@" > $O/c.log 2>&1
cd $R
python3 - "$O" "$PAT" <<'PYEOF'
import csv, glob, sys, collections
O, pat = sys.argv[1], sys.argv[2]
for sub in 'abc':
    acc = collections.defaultdict(list)
    for f in glob.glob(f'{O}/{sub}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()):
        print(f'{k:34s} {sum(v) / len(v):16.0f}  (n={len(v)})')
PYEOF
