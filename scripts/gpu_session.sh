#!/bin/bash
# One GPU-box session (the only session script: the frozen per-session copies of rounds 1-2 were folded into the PARTS below).
# PARTS selects what runs, in this order (default: smoke pytest infer train profinfer proftrain pmc):
#   host      cv2 probe, CPU model, GPU name
#   smoke     __graft_entry__.smoke()
#   pytest    pytest -m gpu            (PYTEST_ARGS: extra arguments, e.g. "-x -k wino tests/test_gpu_tracknet.py")
#   infer     default bench.py line    (BENCH_ARGS; -> bench.json, bench_layers.json)
#   train     bench.py --mode train
#   micro     scripts/microbench.py
#   ab        scripts/wino_ab.py $AB   (Winograd kernel variants / timing twins, e.g. AB="3 5 6")
#   fixed     scripts/wino_fixed_cost.py $FC
#   decoder   the decoder-entry and weight-gradient A/B sweeps
#   custom    $CUSTOM_CMD (a shell command line; output -> custom.log)
#   profinfer rocprofv3 --kernel-trace --stats of the one-stream inference bench
#   proftrain rocprofv3 --kernel-trace --stats of the training bench
#   pmc       FETCH_SIZE / WRITE_SIZE passes (separate runs, --kernel-trace only)
#   sq        SQ / TCC counter passes (scripts/gpu_pmc_sq.sh)
#   sqtrain   the same of the TRAINING step on one stream, plus FETCH_SIZE / WRITE_SIZE (-> train_sq_summary.json)
#   proftrain1 rocprofv3 --kernel-trace --stats of the training bench on ONE stream (TNV3_WGRAD_OVERLAP=0: stand-alone kernel times)
# Everything that must come back is written under gpurun_out/; scripts/collect_profiles.sh copies the summaries to profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
export TNV3_REPORT_DIR="$PWD/gpurun_out"      # the parity tests write their measurement reports there
PARTS="${PARTS:-smoke pytest infer train profinfer proftrain pmc}"
REPO=$PWD
OUT=$PWD/gpurun_out
mkdir -p $OUT
LOG=$OUT/session.log
: > $LOG
has() { [[ " $PARTS " == *" $1 "* ]]; }
sha256sum tracknetv3_amd/libtnv3_hip.so | cut -d" " -f1 > $OUT/lib_sha256.txt      # which build this session measured
python -c "from tracknetv3_amd import _build; print(_build.source_sha256())" 2>/dev/null | tail -1 > $OUT/src_sha256.txt      # ... and its path-independent key
if has host; then
  python -c "import cv2; print('cv2', cv2.__version__)" > $OUT/cv2_probe.txt 2>&1
  nproc > $OUT/host.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/host.txt; rocm-smi --showproductname 2>/dev/null | head -20 >> $OUT/host.txt
  python -c "import torch; print('gpus', torch.cuda.device_count())" >> $OUT/host.txt 2>&1; cat $OUT/host.txt | tee -a $LOG; fi
if has smoke; then echo "== smoke" | tee -a $LOG
  timeout 600 python __graft_entry__.py smoke >> $LOG 2>&1; echo "smoke rc=$?" | tee -a $LOG; fi
if has pytest; then echo "== pytest -m gpu" | tee -a $LOG
  timeout 1800 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=12 ${PYTEST_ARGS:-} > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $LOG
  tail -40 $OUT/pytest_gpu.log | cut -c1-300 | tee -a $LOG; fi
if has infer; then echo "== bench infer" | tee -a $LOG
  timeout 900 python bench.py ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $LOG
  cat $OUT/bench.json | tee -a $LOG; grep -E "^\[layer\]|^\[tune\]" $OUT/bench.err | tee -a $LOG; tail -3 $OUT/bench.err | tee -a $LOG; fi
if has train; then echo "== bench train" | tee -a $LOG
  timeout 600 python bench.py --mode train --steps 10 --warmup 2 > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "bench train rc=$?" | tee -a $LOG
  cat $OUT/bench_train.json | tee -a $LOG; tail -5 $OUT/bench_train.err | tee -a $LOG; fi
if has micro; then echo "== microbench" | tee -a $LOG
  timeout 600 python scripts/microbench.py > $OUT/microbench.json 2> $OUT/microbench.err; echo "microbench rc=$?" | tee -a $LOG
  cat $OUT/microbench.json | tee -a $LOG; tail -5 $OUT/microbench.err | tee -a $LOG; fi
if has ab; then echo "== winograd A/B: ${AB:-3 5}" | tee -a $LOG
  timeout 400 python scripts/wino_ab.py ${AB:-3 5} > $OUT/wino_ab.log 2>&1; echo "ab rc=$?" | tee -a $LOG
  grep -v amdgpu.ids $OUT/wino_ab.log | python scripts/ab_fmt.py | tee -a $LOG; fi
if has fixed; then echo "== winograd per-tile fixed cost: ${FC:-86 85}" | tee -a $LOG
  timeout 300 python scripts/wino_fixed_cost.py ${FC:-86 85} > $OUT/wino_fixed_cost.log 2>&1; echo "fixed rc=$?" | tee -a $LOG
  grep -v amdgpu.ids $OUT/wino_fixed_cost.log | cut -c1-1200 | tee -a $LOG; fi
if has decoder; then echo "== decoder-entry / weight-gradient A/B sweeps" | tee -a $LOG
  for s in up2x_wino_ab dgrad_up2x_ab wgrad_up_sweep; do timeout 200 python scripts/$s.py > $OUT/$s.log 2>&1; echo "$s rc=$?" | tee -a $LOG; grep -v amdgpu $OUT/$s.log | cut -c1-260 | tee -a $LOG; done
  timeout 200 python scripts/wgrad_wino_ab.py ${WGRAD_AB:-0 1} > $OUT/wgrad_wino_ab.log 2>&1; echo "wgrad wino ab rc=$?" | tee -a $LOG; grep -v amdgpu $OUT/wgrad_wino_ab.log | cut -c1-260 | tee -a $LOG; fi
if has custom; then echo "== custom: ${CUSTOM_CMD:-true}" | tee -a $LOG
  timeout ${CUSTOM_TIMEOUT:-600} bash -c "${CUSTOM_CMD:-true}" > $OUT/custom.log 2>&1; echo "custom rc=$?" | tee -a $LOG; grep -v amdgpu.ids $OUT/custom.log | tail -${CUSTOM_TAIL:-60} | cut -c1-400 | tee -a $LOG; fi
cd /tmp
if has profinfer; then echo "== rocprofv3 kernel trace (infer)" | tee -a $LOG
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_infer -o trace -- python $REPO/bench.py --steps 5 --warmup 2 --blocks 1 --no-cpu-baseline --overlap-streams 0 --infer-split 0 --train-steps 0 --extras 0 --layers-out $OUT/prof_infer_layers.json > $OUT/prof_infer.json 2> $OUT/prof_infer.err; echo "rocprof infer rc=$?" | tee -a $LOG; fi
if has proftrain; then echo "== rocprofv3 kernel trace (train)" | tee -a $LOG
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_train -o trace -- python $REPO/bench.py --mode train --steps 3 --warmup 1 --strong-steps 0 --no-cpu-baseline > $OUT/prof_train.json 2> $OUT/prof_train.err; echo "rocprof train rc=$?" | tee -a $LOG; fi
if has proftrain1; then echo "== rocprofv3 kernel trace (train, one stream)" | tee -a $LOG
  TNV3_WGRAD_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_train1 -o trace -- python $REPO/bench.py --mode train --steps 3 --warmup 1 --strong-steps 0 --no-cpu-baseline > $OUT/prof_train1.json 2> $OUT/prof_train1.err; echo "rocprof train (one stream) rc=$?" | tee -a $LOG; fi
if has pmc; then echo "== rocprofv3 PMC passes (separate runs; counters only with --kernel-trace)" | tee -a $LOG
  for c in FETCH_SIZE WRITE_SIZE; do n=$(echo $c | cut -d_ -f1 | tr A-Z a-z)
    timeout 600 rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$n -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --blocks 1 --no-cpu-baseline --overlap-streams 0 --infer-split 0 --train-steps 0 --extras 0 --layers-out /tmp/pmc_layers.json > /dev/null 2> $OUT/pmc_$n.err; echo "pmc $n rc=$?" | tee -a $LOG; done; fi
cd $REPO
for d in prof_infer prof_train prof_train1 pmc_fetch pmc_write; do
  for f in $(find $OUT/$d -name "*.db" 2>/dev/null); do python scripts/rocpd_summary.py $f $OUT/${d}_kernel_stats.csv >> $LOG 2>&1; python scripts/rocpd_pmc.py $f $OUT/${d}_pmc.csv >> $LOG 2>&1; done
done
if has sq; then echo "== SQ / TCC counter passes" | tee -a $LOG; bash scripts/gpu_pmc_sq.sh 2>&1 | tail -20 | tee -a $LOG; fi
if has sqtrain; then echo "== SQ / TCC / FETCH / WRITE counter passes of the training step (one stream)" | tee -a $LOG; SQ_MODE=train bash scripts/gpu_pmc_sq.sh 2>&1 | tail -24 | tee -a $LOG; fi
find $OUT -name "*.db" -size +15M -delete
echo "== done" | tee -a $LOG
