#!/bin/bash
# streaming Winograd kernel (variant 5) as the default: full GPU suite, A/B 3 / 56 / 5, per-tile fixed cost 83 / 86 / 85, bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 300 python scripts/wino_ab.py 3 56 5 > $OUT/wino_ab5.log 2>&1; echo "ab rc=$?"; cat $OUT/wino_ab5.log | grep -v amdgpu.ids | cut -c1-600; cp $OUT/wino_ab.json $OUT/wino_ab5.json
timeout 300 python scripts/wino_fixed_cost.py 83 86 85 > $OUT/wino_fixed_cost.log 2>&1; echo "fixed rc=$?"; grep -v amdgpu.ids $OUT/wino_fixed_cost.log | cut -c1-1200
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -6 $OUT/pytest_gpu.log | cut -c1-250
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "
import json; b=json.load(open('$OUT/bench.json')); print(b['value'], b['ms_per_step'], b['roofline']['frac'], b['overlap'] and b['overlap']['value'], b['train']['value'], b['train']['ms_per_step'])"
grep "^\[layer\]" $OUT/bench.err
