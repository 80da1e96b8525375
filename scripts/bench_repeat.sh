# Run-to-run spread of the headline lines on ONE box: the default inference line (without the CPU baseline) and the training line, three times each.
#   PARTS=custom CUSTOM_CMD="bash scripts/bench_repeat.sh" CUSTOM_TIMEOUT=1200 bash scripts/gpu_session.sh
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --train-steps 0 --extras 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer run $i', b['value'], 'frames/s', b['ms_per_step'], 'ms', b['blocks']['ms_per_step'], 'frac', b['roofline']['frac'], 'conv_ms', b['roofline']['conv_ms_per_step'])"
done
for i in 1 2 3; do
  python bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train run $i', b['value'], 'frames/s', b['ms_per_step'], 'ms', 'strong(batch 80)', b['strong']['ms_per_step'], 'ms')"
done
