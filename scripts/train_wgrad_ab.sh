# Step-level A/B of the DISPATCHABLE Winograd weight-gradient generations (1, 5: F(2x2); 8: F(4x4), the default; the twins 0 / 2-4 / 6 / 7 live in libtnv3_diag.so) and of the stem layer's route (one GPU box):
#   PARTS=custom CUSTOM_CMD="bash scripts/train_wgrad_ab.sh" bash scripts/gpu_session.sh
run() { python bench.py --mode train --steps 20 --warmup 3 --strong-steps 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms_per_step', b['ms_per_step'])"; }
for rep in 1 2; do for v in ${VARIANTS:-1 5 8}; do TNV3_WGRAD_WINO_VARIANT=$v TNV3_WINO_WGRAD_MIN_CIN=65 run "variant $v (stem direct)"; done; done
TNV3_WINO_WGRAD_MIN_CIN=65 run "default, stem direct"
run "default (stem in Winograd form)"
