# Sweep of the eval forward's intra-batch split (shares of the batch, one HIP stream each) on the headline workload:
#   PARTS=custom CUSTOM_CMD="bash scripts/infer_split_sweep.sh" bash scripts/gpu_session.sh
for parts in ${SPLITS:-6,4 5,5 7,3 4,3,3 4,4,2 5,3,2 3,3,2,2}; do
  TNV3_INFER_SPLIT_PARTS=$parts python bench.py --no-cpu-baseline --train-steps 0 --extras 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('parts $parts', 'ms_per_step', b['ms_per_step'], b['blocks']['ms_per_step'])"
done
