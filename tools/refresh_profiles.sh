#!/bin/bash
# Regenerates the round's profiles on a GPU box (run from the repo root): bench line, rocprofv3 kernel stats, per-layer view,
# PMC HBM traffic, SQ counters, kernel stats of the precision modes / training step.  Outputs land in gpurun_out/profiles_new/;
# usage: V2W_COMMIT=<hash> V2W_DATE=<date> tools/refresh_profiles.sh rNN
R=$PWD; TAG=${1:-r02}; O=$R/gpurun_out/profiles_new; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt"
B3="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- $B > $O/ks.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pf -- $B3 > $O/pf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pw -- $B3 > $O/pw.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/sq_a -o sa -- $B3 > $O/sa.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq_b -o sb -- $B3 > $O/sb.log 2>&1
for prec in f16x3 bf16; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$prec -o ks -- $B --precision $prec > $O/ks_$prec.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_train -o ks -- python3 $R/tools/train_step_bench.py 32 256 5 > $O/ks_train.log 2>&1
# BASELINE configs[2] in its own arithmetic (bf16 compute / fp32 accumulate, bf16 activation storage) and the cfg1 inference latency
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_cfg3 -o ks -- $B --precision bf16 --batch 64 --frames 512 > $O/ks_cfg3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/ks_lat -o ks -- python3 $R/tools/latency_bench.py 1 50 > $O/ks_lat.log 2>&1
# one train.py GAN iteration (generator, mel, D step, G step) at the cfg2 shape: kernel stats of 1 + 2 warm-up iterations
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_gan -o ks -- python3 $R/tools/gan_step_bench.py 32 256 1 > $O/ks_gan.log 2>&1
# the north-star shape B = 32 x T = 256 in the bf16 arithmetic: fabric traffic per kernel (bench.py reads cfg2_bf16's `traffic` from it)
B3B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --precision bf16"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/c2b_fetch -o pf -- $B3B > $O/c2bpf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c2b_write -o pw -- $B3B > $O/c2bpw.log 2>&1
B3C="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --precision bf16 --batch 64 --frames 512"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/c3_fetch -o pf -- $B3C > $O/c3pf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c3_write -o pw -- $B3C > $O/c3pw.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/c3_sq_a -o sa -- $B3C > $O/c3sa.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/c3_sq_b -o sb -- $B3C > $O/c3sb.log 2>&1
cd $R
python3 tools/trace_layers.py $O/ks > $O/per_layer.txt 2>&1
python3 tools/trace_layers.py $O/ks_cfg3 64 512 -2 2 > $O/cfg3_bf16_per_layer.txt 2>&1
python3 tools/trace_timeline.py $O/ks_lat > $O/cfg1_latency_timeline.txt 2>&1
python3 tools/latency_bench.py > $O/latency.txt 2>&1
python3 tools/train_step_bench.py 32 256 5 > $O/train_step.txt 2>&1
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/hbm_traffic.json 2> $O/pmc.err
python3 tools/pmc_sq.py $O/sq_a $O/sq_b > $O/sq_counters.json 2> $O/sq.err
python3 tools/pmc_traffic.py $O/c3_fetch $O/c3_write > $O/cfg3_bf16_hbm_traffic.json 2> $O/c3pmc.err
python3 tools/pmc_traffic.py $O/c2b_fetch $O/c2b_write > $O/cfg2_bf16_hbm_traffic.json 2> $O/c2bpmc.err
python3 tools/trace_layers.py $O/ks_bf16 32 256 -2 2 > $O/cfg2_bf16_per_layer.txt 2>&1
python3 tools/pmc_sq.py $O/c3_sq_a $O/c3_sq_b > $O/cfg3_bf16_sq_counters.json 2> $O/c3sq.err
cp $O/hbm_traffic.json profiles/${TAG}_cfg2_hbm_traffic.json    # bench.py reads the traffic of its dominant kernel from here
cp $O/cfg3_bf16_hbm_traffic.json profiles/${TAG}_cfg3_bf16_hbm_traffic.json
cp $O/cfg2_bf16_hbm_traffic.json profiles/${TAG}_cfg2_bf16_hbm_traffic.json
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err            # the driver's line (compact: ends in `summary`)
cp gpurun_out/bench_detail.json $O/bench_detail.json                   # ... and the full-precision line with every per-kernel table
# the RCCL branch on a one-rank communicator: the per-step cost of the five statistics all-reduces (bench.py `stat_sync`)
python3 - "$O" > $O/sync_overhead.txt <<'PYEOF'
import json, sys
d = json.load(open(sys.argv[1] + '/bench.json'))
s = d.get('stat_sync') or {}
print('bench.py stat_sync block (one-rank RCCL group on one MI355X, BASELINE configs[1] forward, fp32):')
for k, v in s.items():
    print(f'  {k}: {v}')
PYEOF
python3 tools/config_bench.py > $O/all_configs.txt 2>&1
# one full GAN iteration (vec2wav/train.py:160-215) at the cfg2 shape and at the reference's own batch_size = 2
python3 tools/gan_step_bench.py 32 256 3 > $O/gan_iteration.txt 2>&1
python3 tools/gan_step_bench.py 2 256 3 >> $O/gan_iteration.txt 2>&1
python3 tools/gan_step_bench.py 32 256 3 frozen >> $O/gan_iteration.txt 2>&1
# the same iteration with the split-f16 convs (discriminators: the dense five-tap 1024 -> 1024 layers; generator: precision = 'f16x3'); weight gradients exact
python3 tools/gan_step_bench.py 32 256 3 hip f16x3 >> $O/gan_iteration.txt 2>&1
python3 tools/gan_step_bench.py 32 256 3 hip f16x3 frozen >> $O/gan_iteration.txt 2>&1
tail -c 600 $O/bench.json; grep ms/step $O/ks_train.log
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
