# Step-level sweep of tuning.WGRAD_WINO_TAIL (how many of the first Conv2DBlocks take the no-role weight-gradient kernel):
#   PARTS=custom CUSTOM_CMD="bash scripts/train_tail_ab.sh" CUSTOM_TIMEOUT=1200 bash scripts/gpu_session.sh
run() { python bench.py --mode train --steps 20 --warmup 3 --strong-steps 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms_per_step', b['ms_per_step'])"; }
for rep in 1 2; do for k in ${TAILS:-0 2 4 7 10}; do TNV3_WGRAD_WINO_TAIL=$k run "tail $k"; done; done
