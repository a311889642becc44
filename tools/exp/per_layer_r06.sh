#!/bin/bash
# the two per-layer views of the bf16 pipeline (tools/trace_layers.py) from fresh kernel traces -> gpurun_out/profiles_new/
R=$PWD; O=$R/gpurun_out/profiles_new; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_cfg3 -o ks -- $B --precision bf16 --batch 64 --frames 512 > $O/kt_cfg3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_cfg2b -o ks -- $B --precision bf16 > $O/kt_cfg2b.log 2>&1
cd $R
python3 tools/trace_layers.py $O/kt_cfg3 64 512 -2 2 > $O/cfg3_bf16_per_layer.txt 2>&1
python3 tools/trace_layers.py $O/kt_cfg2b 32 256 -2 2 > $O/cfg2_bf16_per_layer.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
