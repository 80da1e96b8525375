#!/bin/bash
# persistent Winograd kernel (variant 5): A/B against variant 3, per-tile fixed-cost timeline, bench with either default
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 300 python scripts/wino_ab.py 3 5 > $OUT/wino_ab5.log 2>&1; echo "ab rc=$?"; cat $OUT/wino_ab5.log | cut -c1-400; cp $OUT/wino_ab.json $OUT/wino_ab5.json
timeout 300 python scripts/wino_fixed_cost.py 83 85 > $OUT/wino_fixed_cost.log 2>&1; echo "fixed rc=$?"; cat $OUT/wino_fixed_cost.log | cut -c1-1200
for V in 3 5; do
  TNV3_WINO_VARIANT=$V timeout 300 python bench.py --no-cpu-baseline --train-steps 3 > $OUT/bench_v$V.json 2> $OUT/bench_v$V.err; echo "bench v$V rc=$?"
  python -c "
import json; b=json.load(open('$OUT/bench_v$V.json')); print('v$V', b['value'], b['ms_per_step'], b['roofline']['frac'], b['train']['value'], b['train']['ms_per_step'])"
done
