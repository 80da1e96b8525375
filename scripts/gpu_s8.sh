#!/bin/bash
# Round-2 evidence session: smoke, full GPU suite, default bench (with CPU baselines), training bench, Winograd A/Bs + fixed-cost
# timeline + timing twins, rocprofv3 kernel traces (inference one-stream, training), PMC traffic + SQ counter passes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
LOG=$OUT/session.log; : > $LOG
echo "== smoke" | tee -a $LOG
timeout 600 python __graft_entry__.py smoke >> $LOG 2>&1; echo "smoke rc=$?" | tee -a $LOG
echo "== pytest -m gpu" | tee -a $LOG
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $LOG
tail -5 $OUT/pytest_gpu.log | cut -c1-200 | tee -a $LOG
echo "== bench (default)" | tee -a $LOG
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $LOG
cat $OUT/bench.json >> $LOG; grep "^\[layer\]" $OUT/bench.err >> $LOG
echo "== bench train" | tee -a $LOG
timeout 600 python bench.py --mode train --steps 10 --warmup 2 > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "bench train rc=$?" | tee -a $LOG
cat $OUT/bench_train.json >> $LOG
echo "== winograd A/B, fixed cost, twins" | tee -a $LOG
timeout 300 python scripts/wino_ab.py 2 3 56 5 > $OUT/ab1.log 2>&1; cp $OUT/wino_ab.json $OUT/r02_wino_variants_ab2.json; python scripts/ab_fmt.py < $OUT/ab1.log | tee -a $LOG
timeout 300 python scripts/wino_ab.py 5 95 96 97 98 99 > $OUT/ab2.log 2>&1; cp $OUT/wino_ab.json $OUT/r02_wino_stream_twins.json; python scripts/ab_fmt.py < $OUT/ab2.log | tee -a $LOG
timeout 300 python scripts/wino_fixed_cost.py 83 86 85 > $OUT/wino_fixed_cost.log 2>&1; cp $OUT/wino_fixed_cost.json $OUT/r02_wino_fixed_cost.json; echo "fixed rc=$?" | tee -a $LOG
echo "== decoder-entry kernels: forward / data gradient / weight gradient A/Bs" | tee -a $LOG
timeout 200 python scripts/up2x_wino_ab.py > $OUT/up2x_wino_ab.log 2>&1; grep -v amdgpu $OUT/up2x_wino_ab.log | cut -c1-260 | tee -a $LOG
timeout 200 python scripts/dgrad_up2x_ab.py > $OUT/dgrad_up2x_ab.log 2>&1; grep -v amdgpu $OUT/dgrad_up2x_ab.log | cut -c1-260 | tee -a $LOG
timeout 200 python scripts/wgrad_up_sweep.py > $OUT/wgrad_up_sweep.log 2>&1; echo "wgrad up sweep rc=$?" | tee -a $LOG
timeout 200 python scripts/wgrad_wino_ab.py 0 1 > $OUT/wgrad_wino_ab.log 2>&1; echo "wgrad wino ab rc=$?" | tee -a $LOG
cd /tmp
echo "== rocprofv3 kernel traces" | tee -a $LOG
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_infer -o trace -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --overlap-streams 0 --infer-split 0 --train-steps 0 --layers-out $OUT/prof_infer_layers.json > $OUT/prof_infer.json 2> $OUT/prof_infer.err; echo "rocprof infer rc=$?" | tee -a $LOG
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_train -o trace -- python $REPO/bench.py --mode train --steps 3 --warmup 1 > $OUT/prof_train.json 2> $OUT/prof_train.err; echo "rocprof train rc=$?" | tee -a $LOG
echo "== PMC passes" | tee -a $LOG
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --overlap-streams 0 --infer-split 0 --train-steps 0 --layers-out /tmp/pmc_layers.json > /dev/null 2> $OUT/pmc_fetch.err; echo "pmc fetch rc=$?" | tee -a $LOG
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --overlap-streams 0 --infer-split 0 --train-steps 0 --layers-out /tmp/pmc_layers.json > /dev/null 2> $OUT/pmc_write.err; echo "pmc write rc=$?" | tee -a $LOG
cd $REPO
for d in prof_infer prof_train pmc_fetch pmc_write; do
  for f in $(find $OUT/$d -name "*.db" 2>/dev/null); do python scripts/rocpd_summary.py $f $OUT/${d}_kernel_stats.csv >> $LOG 2>&1; python scripts/rocpd_pmc.py $f $OUT/${d}_pmc.csv >> $LOG 2>&1; done
done
bash scripts/gpu_pmc_sq.sh >> $LOG 2>&1
find $OUT -name "*.db" -delete
echo "== done" | tee -a $LOG
python -c "
import json; b=json.load(open('$OUT/bench.json')); r=b['roofline']; print('INFER', b['value'], b['ms_per_step'], r['frac'], r['achieved'], r['single_stream_ms_per_step'], 'overlap', b['overlap']['value'], 'train', b['train']['value'], b['train']['ms_per_step'], 'cpu', b['cpu_baseline']['value'], b['cpu_baseline'].get('one_thread'))
t=json.load(open('$OUT/bench_train.json')); print('TRAIN', t['value'], t['ms_per_step'], t['roofline']['frac'], 'cpu', t['cpu_baseline'] and t['cpu_baseline']['value'])"
