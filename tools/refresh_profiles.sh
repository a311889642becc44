#!/bin/bash
# Regenerates profiles/ on a GPU box (run from the repo root): bench line, rocprofv3 kernel stats, per-layer view, PMC HBM traffic,
# the kernel stats of the f16x3 / bf16 precision modes and of the generator training step, the discriminator forwards and the full
# GAN iteration.  Outputs land in gpurun_out/profiles_new/.
R=$PWD; O=$R/gpurun_out/profiles_new; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- $B > $O/ks.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pf -- $B > $O/pf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pw -- $B > $O/pw.log 2>&1
for prec in f16x3 bf16; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$prec -o ks -- $B --precision $prec > $O/ks_$prec.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_train -o ks -- python3 $R/tools/train_step_bench.py 32 256 5 > $O/ks_train.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_disc -o ks -- python3 $R/tools/disc_bench.py 32 81920 3 > $O/ks_disc.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_gan -o ks -- python3 $R/tools/gan_step_bench.py 32 256 1 > $O/ks_gan.log 2>&1
cd $R
timeout 300 python3 tools/gan_step_bench.py 32 256 3 > $O/gan_iteration.txt 2>&1
python3 tools/trace_layers.py $O/ks > $O/per_layer.txt 2>&1
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/hbm_traffic.json 2> $O/pmc.err
cp $O/hbm_traffic.json profiles/r01_cfg2_hbm_traffic.json    # bench.py reads the traffic of its dominant kernel from here
timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json; grep ms/step $O/ks_train.log
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
