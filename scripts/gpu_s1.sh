#!/bin/bash
# round-2 session 1: GPU test-suite (incl. the new full-size parity tests) + default bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -c "import cv2; print('cv2', cv2.__version__)" > $OUT/cv2_probe.txt 2>&1
nproc > $OUT/host.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/host.txt; rocm-smi --showproductname 2>/dev/null | head -20 >> $OUT/host.txt
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -30 $OUT/pytest_gpu.log | cut -c1-250
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json | cut -c1-3000
grep "^\[layer\]" $OUT/bench.err
tail -3 $OUT/bench.err
