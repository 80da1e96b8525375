#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_training.py -m gpu -q --no-header -p no:cacheprovider -k "fused or trainer_with or second_backward or eval_after" > $OUT/pytest_s2.log 2>&1; echo "pytest rc=$?"
tail -25 $OUT/pytest_s2.log | cut -c1-250
timeout 300 python scripts/wino_timeline.py 27 > $OUT/wino_timeline.log 2>&1; echo "timeline rc=$?"; cat $OUT/wino_timeline.log | cut -c1-1200
cd /tmp
for mode in fused torch; do for k in 3 8; do
  extra=""; [ $mode = torch ] && extra="torch"
  timeout 300 rocprofv3 --memory-copy-trace --kernel-trace --stats --output-format csv -d $OUT/copy_${mode}_$k -o t -- python $REPO/scripts/train_copy_probe.py $k $extra > $OUT/copy_${mode}_$k.log 2>&1
  echo "copyprobe $mode $k rc=$?"
  f=$(find $OUT/copy_${mode}_$k -name "*memory_copy_trace.csv" | head -1)
  if [ -n "$f" ]; then echo "$mode K=$k copies: $(($(wc -l < $f) - 1))"; cut -d, -f1-4 $f | sort | uniq -c | sort -rn | head -5; else echo "no memcpy csv"; ls $OUT/copy_${mode}_$k; find $OUT/copy_${mode}_$k | head; fi
done; done
cd $REPO
find $OUT -name "*.db" -delete
