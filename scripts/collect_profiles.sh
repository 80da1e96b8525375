#!/bin/bash
# Copies the summaries of the last scripts/gpu_session.sh evidence session (PARTS="host smoke pytest infer train ab fixed decoder profinfer proftrain pmc sq") from gpurun_out/ (scratch) into profiles/ (tracked).
set -u
cd "$(dirname "$0")/.."
R=${ROUND:-r06}
O=gpurun_out
P=profiles
cp $O/bench.json $P/${R}_bench_n1.json
cp $O/bench_layers.json $P/${R}_bench_layers.json
cp $O/bench_train.json $P/${R}_bench_train_n1.json
cp $O/prof_infer.json $P/${R}_bench_n1_under_rocprof.json
cp $O/prof_train.json $P/${R}_bench_train_under_rocprof.json
cp $O/prof_infer_kernel_stats.csv $P/${R}_infer_kernel_stats.csv
cp $O/prof_train_kernel_stats.csv $P/${R}_train_kernel_stats.csv
cp $O/pmc_fetch_pmc.csv $P/${R}_infer_pmc_fetch_size.csv
cp $O/pmc_write_pmc.csv $P/${R}_infer_pmc_write_size.csv
for f in sq1 sq2 sq3 tcc; do cp $O/pmc_$f.csv $P/${R}_infer_pmc_$f.csv; done
# optional parts: only what THIS session wrote (gpurun_out/ keeps the files of earlier sessions and rounds; lib_sha256.txt is written when a session starts)
fresh() { [ -f "$1" ] && [ "$1" -nt $O/lib_sha256.txt ]; }
for f in up2x_wino_ab dgrad_up2x_ab wgrad_up_sweep wgrad_wino_ab; do fresh $O/$f.json && cp $O/$f.json $P/${R}_$f.json; done
fresh $O/e2e_real_network_report.json && cp $O/e2e_real_network_report.json $P/${R}_e2e_real_network_report.json
# (the library that ran on the GPU box is the one in the tree: the session ships the tree; its sha256 keys the counters to the build)
python scripts/conv_traffic.py $O/pmc_fetch_pmc.csv $O/pmc_write_pmc.csv 3 $P/conv_traffic.json ${COMMIT:-$(git rev-parse --short HEAD)} $(cat $O/lib_sha256.txt 2>/dev/null || sha256sum tracknetv3_amd/libtnv3_hip.so | cut -d" " -f1) $(cat $O/src_sha256.txt 2>/dev/null || python -c "from tracknetv3_amd import _build; print(_build.source_sha256())" | tail -1)
[ -f $O/lib_sha256.txt ] && cp $O/lib_sha256.txt $P/${R}_lib_sha256.txt
[ -f $O/src_sha256.txt ] && cp $O/src_sha256.txt $P/${R}_src_sha256.txt
for f in wino43_variant_ab up2x_wino43_ab wino43s_timeline wgrad_wino43_ab; do fresh $O/$f.json && cp $O/$f.json $P/${R}_$f.json; done
for f in fullsize_train_parity_n2_f43fwd fullsize_train_parity_n2_f22fwd fullsize_train_parity_n10_default; do fresh $O/$f.json && cp $O/$f.json $P/${R}_$f.json; done
[ -f $O/infer_sq_summary.json ] && cp $O/infer_sq_summary.json $P/${R}_infer_sq_summary.json
# round 5: the training step's counters (one stream), its stand-alone kernel times, the precision sweep and channel-plan tables, the new A/Bs
fresh $O/train_sq_summary.json && cp $O/train_sq_summary.json $P/${R}_train_sq_summary.json
for f in sq1 sq2 sq3 tcc fetch write; do fresh $O/pmc_train_$f.csv && cp $O/pmc_train_$f.csv $P/${R}_train_pmc_$f.csv; done
fresh $O/prof_train1_kernel_stats.csv && cp $O/prof_train1_kernel_stats.csv $P/${R}_train_one_stream_kernel_stats.csv
for f in precision_sweep_eval precision_sweep_train channel_plan_9to3 channel_plan_24to8 channel_plan_8to8 channel_plan_32to8 wgrad_up2x_wino43_ab dgrad_up2x_wino43_ab; do
  fresh $O/$f.json && cp $O/$f.json $P/${R}_$f.json; done
cp $O/session.log $P/${R}_session_final.log
echo "profiles/ refreshed from $O"
