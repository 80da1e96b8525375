#!/bin/bash
# evidence run for the Winograd study + full GPU suite + bench with variant 3 as default
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -6 $OUT/pytest_gpu.log | cut -c1-250
timeout 300 python scripts/wino_ab.py 2 3 4 33 41 51 61 > $OUT/wino_ab.log 2>&1; echo "ab rc=$?"; cp $OUT/wino_ab.json $OUT/r02_wino_variants_ab.json
timeout 300 python scripts/wino_timeline.py 27 37 47 > $OUT/wino_timeline.log 2>&1; echo "timeline rc=$?"; cp $OUT/wino_timeline.json $OUT/r02_wino_timeline.json
timeout 100 python scripts/coissue_probe.py > $OUT/coissue.log 2>&1; cp $OUT/coissue_probe.json $OUT/r02_mfma_f32_coissue.json
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "
import json; b=json.load(open('$OUT/bench.json')); print(b['value'], b['ms_per_step'], b['roofline']['frac'], b['overlap']['value'], b['train']['value'], b['train']['ms_per_step'])"
grep "^\[layer\]" $OUT/bench.err
