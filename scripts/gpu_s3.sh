#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_tracknet.py -m gpu -q --no-header -p no:cacheprovider -k "wino" > $OUT/pytest_s3.log 2>&1; echo "pytest rc=$?"
tail -8 $OUT/pytest_s3.log | cut -c1-250
timeout 300 python scripts/wino_ab.py ${AB_VARIANTS:-2 3} > $OUT/wino_ab.log 2>&1; echo "ab rc=$?"; cat $OUT/wino_ab.log | cut -c1-400
timeout 300 python scripts/wino_timeline.py ${TL_VARIANTS:-27 37} > $OUT/wino_timeline.log 2>&1; echo "timeline rc=$?"; grep -v amdgpu.ids $OUT/wino_timeline.log | cut -c1-900
