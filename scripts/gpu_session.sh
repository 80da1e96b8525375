#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, tile-config tuning + bench, rocprofv3 kernel trace.
# Everything that must come back is written under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
echo "== device" | tee $OUT/session.log
(rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8; nproc; lscpu | grep -E "Model name|^CPU\(s\)|Socket|Core") >> $OUT/session.log 2>&1
echo "== smoke" | tee -a $OUT/session.log
timeout 600 python __graft_entry__.py smoke >> $OUT/session.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/session.log
echo "== pytest -m gpu" | tee -a $OUT/session.log
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/session.log
tail -15 $OUT/pytest_gpu.log | tee -a $OUT/session.log
echo "== bench (tune)" | tee -a $OUT/session.log
timeout 900 python bench.py --tune --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/session.log
cat $OUT/bench.json | tee -a $OUT/session.log
grep -E "^\[layer\]|^\[tune\]" $OUT/bench.err | tee -a $OUT/session.log
echo "== rocprofv3 kernel trace" | tee -a $OUT/session.log
cp $OUT/conv_tuning.json tracknetv3_amd/conv_tuning.json.session 2>/dev/null
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o trace -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OLDPWD/$OUT/prof_bench.json 2> $OLDPWD/$OUT/prof_bench.err); echo "rocprof rc=$?" | tee -a $OUT/session.log
find $OUT/prof -name "*stats*" | head | tee -a $OUT/session.log
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -25 $f | cut -c1-220 | tee -a $OUT/session.log; done
# keep only the small summaries (the raw trace can be large)
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
echo "== done" | tee -a $OUT/session.log
