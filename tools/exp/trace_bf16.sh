#!/bin/bash
# usage: tools/exp/trace_bf16.sh B T TAG   (kernel trace of a short bf16 bench run; timeline of one forward -> gpurun_out/TAG_timeline.txt)
R=$PWD; B=${1:-32}; T=${2:-256}; TAG=${3:-cfg2bf}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_$TAG -o t -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-alt --precision bf16 --batch $B --frames $T > $R/gpurun_out/tr_$TAG.log 2>&1
cd $R
python3 tools/trace_gaps.py gpurun_out/tr_$TAG -2 cond_fc > gpurun_out/${TAG}_timeline.txt 2>&1
find gpurun_out/tr_$TAG -name "*.csv" -delete
