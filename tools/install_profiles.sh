#!/bin/bash
# Copies the summaries tools/refresh_profiles.sh wrote (gpurun_out/profiles_new/, merged back by gpurun) into profiles/ under the round tag.
# usage: tools/install_profiles.sh rNN
T=${1:?round tag}; O=gpurun_out/profiles_new; P=profiles
cp $O/all_configs.txt $P/${T}_all_configs.txt
cp $O/cfg1_latency_timeline.txt $P/${T}_cfg1_latency_timeline.txt
cp $O/bench.json $P/${T}_cfg2_bench.json
cp $O/bench_detail.json $P/${T}_cfg2_bench_detail.json
cp $O/cfg2_bf16_hbm_traffic.json $P/${T}_cfg2_bf16_hbm_traffic.json
cp $O/cfg2_bf16_per_layer.txt $P/${T}_cfg2_bf16_per_layer.txt
cp $O/ks/ks_kernel_stats.csv $P/${T}_cfg2_kernel_stats.csv
cp $O/ks_bf16/ks_kernel_stats.csv $P/${T}_cfg2_bf16_kernel_stats.csv
cp $O/ks_f16x3/ks_kernel_stats.csv $P/${T}_cfg2_f16x3_kernel_stats.csv
cp $O/ks_train/ks_kernel_stats.csv $P/${T}_cfg2_train_step_kernel_stats.csv
cp $O/ks_cfg3/ks_kernel_stats.csv $P/${T}_cfg3_bf16_kernel_stats.csv
cp $O/hbm_traffic.json $P/${T}_cfg2_hbm_traffic.json
cp $O/cfg3_bf16_hbm_traffic.json $P/${T}_cfg3_bf16_hbm_traffic.json
cp $O/sq_counters.json $P/${T}_sq_counters.json
cp $O/cfg3_bf16_sq_counters.json $P/${T}_cfg3_bf16_sq_counters.json
cp $O/per_layer.txt $P/${T}_cfg2_per_layer.txt
cp $O/cfg3_bf16_per_layer.txt $P/${T}_cfg3_bf16_per_layer.txt
cp $O/latency.txt $P/${T}_latency.txt
cp $O/train_step.txt $P/${T}_train_step.txt
cp $O/sync_overhead.txt $P/${T}_sync_overhead.txt
cp $O/gan_iteration.txt $P/${T}_gan_iteration.txt
cp $O/ks_gan/ks_kernel_stats.csv $P/${T}_gan_iteration_kernel_stats.csv
ls -la $P/${T}_*
