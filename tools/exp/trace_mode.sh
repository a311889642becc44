#!/bin/bash
# usage: tools/exp/trace_mode.sh B T mode marker  (kernel trace of plain bf16 forwards in one mode; timeline of one -> gpurun_out/MODE_timeline.txt)
R=$PWD; B=${1:-32}; T=${2:-256}; M=${3:-train}; K=${4:-split_rowscale}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_$M -o t -- python3 $R/tools/exp/fwd_loop.py $B $T $M 12 > $R/gpurun_out/tr_$M.log 2>&1
cd $R
python3 tools/trace_gaps.py gpurun_out/tr_$M -2 $K > gpurun_out/${M}_timeline.txt 2>&1
find gpurun_out/tr_$M -name "*.csv" -delete
