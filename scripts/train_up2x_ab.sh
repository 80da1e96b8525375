#!/bin/bash
# A/B of the training step (bench.py --mode train, batch 10) over the forms of the decoder entries' upsampled halves, in ONE session (box-to-box
# variance is +-3 %): data gradient 0 (one-GEMM F(2x2)) / 2 (25-of-36 F(4x4)), weight gradient 1 (9-GEMM F(2x2)) / -1 (25-of-36), forward 0 / 2.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
  for cfg in "0 1 0" "2 1 0" "0 -1 0" "2 -1 0" "2 -1 2"; do
    set -- $cfg
    ms=$(TNV3_DGRAD_UP2X_WINO_VARIANT=$1 TNV3_WGRAD_UP2X_VARIANT=$2 TNV3_UP2X_WINO_VARIANT_TRAIN=$3 python bench.py --mode train --steps 10 --warmup 3 --strong-steps 0 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "rep $rep  dgrad_up2x=$1 wgrad_up2x=$2 fwd_up2x_train=$3  ms_per_step=$ms"
  done
done
