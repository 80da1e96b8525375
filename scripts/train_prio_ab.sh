run() { python bench.py --mode train --steps 20 --warmup 3 --strong-steps 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms_per_step', b['ms_per_step'])"; }
for rep in 1 2; do for p in 0 -1; do TNV3_WGRAD_STREAM_PRIORITY=$p run "side-stream priority $p"; done; done
