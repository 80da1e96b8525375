#!/bin/bash
# One GPU-box session.  PARTS selects what runs (default: everything):
#   smoke pytest infer train micro profinfer proftrain pmc
# Everything that must come back is written under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
PARTS="${PARTS:-smoke pytest infer train micro profinfer proftrain pmc}"
REPO=$PWD
OUT=$PWD/gpurun_out
mkdir -p $OUT
LOG=$OUT/session.log
: > $LOG
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has smoke; then echo "== smoke" | tee -a $LOG
  timeout 600 python __graft_entry__.py smoke >> $LOG 2>&1; echo "smoke rc=$?" | tee -a $LOG; fi
if has pytest; then echo "== pytest -m gpu" | tee -a $LOG
  timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider ${PYTEST_ARGS:-} > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $LOG
  tail -40 $OUT/pytest_gpu.log | cut -c1-300 | tee -a $LOG; fi
if has infer; then echo "== bench infer" | tee -a $LOG
  timeout 600 python bench.py ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $LOG
  cat $OUT/bench.json | tee -a $LOG; grep -E "^\[layer\]|^\[tune\]" $OUT/bench.err | tee -a $LOG; fi
if has train; then echo "== bench train" | tee -a $LOG
  timeout 600 python bench.py --mode train --steps 10 --warmup 2 > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "bench train rc=$?" | tee -a $LOG
  cat $OUT/bench_train.json | tee -a $LOG; tail -5 $OUT/bench_train.err | tee -a $LOG; fi
if has micro; then echo "== microbench" | tee -a $LOG
  timeout 600 python scripts/microbench.py > $OUT/microbench.json 2> $OUT/microbench.err; echo "microbench rc=$?" | tee -a $LOG
  cat $OUT/microbench.json | tee -a $LOG; tail -5 $OUT/microbench.err | tee -a $LOG; fi
cd /tmp
if has profinfer; then echo "== rocprofv3 kernel trace (infer)" | tee -a $LOG
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_infer -o trace -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --overlap-streams 0 --infer-split 0 --train-steps 0 --layers-out $OUT/prof_infer_layers.json > $OUT/prof_infer.json 2> $OUT/prof_infer.err; echo "rocprof infer rc=$?" | tee -a $LOG; fi
if has proftrain; then echo "== rocprofv3 kernel trace (train)" | tee -a $LOG
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_train -o trace -- python $REPO/bench.py --mode train --steps 3 --warmup 1 > $OUT/prof_train.json 2> $OUT/prof_train.err; echo "rocprof train rc=$?" | tee -a $LOG; fi
if has pmc; then echo "== rocprofv3 PMC passes (separate runs; counters only with --kernel-trace)" | tee -a $LOG
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --overlap-streams 0 --infer-split 0 --train-steps 0 --layers-out /tmp/pmc_layers.json > /dev/null 2> $OUT/pmc_fetch.err; echo "pmc fetch rc=$?" | tee -a $LOG
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --overlap-streams 0 --infer-split 0 --train-steps 0 --layers-out /tmp/pmc_layers.json > /dev/null 2> $OUT/pmc_write.err; echo "pmc write rc=$?" | tee -a $LOG; fi
cd $REPO
for d in prof_infer prof_train pmc_fetch pmc_write; do
  for f in $(find $OUT/$d -name "*.db" 2>/dev/null); do python scripts/rocpd_summary.py $f $OUT/${d}_kernel_stats.csv >> $LOG 2>&1; python scripts/rocpd_pmc.py $f $OUT/${d}_pmc.csv >> $LOG 2>&1; done
done
find $OUT -name "*.db" -size +15M -delete
echo "== done" | tee -a $LOG
