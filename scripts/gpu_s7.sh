#!/bin/bash
# quick A/B of the Winograd variants + per-tile fixed cost + bench (no test suite)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 300 python scripts/wino_ab.py ${AB:-3 56 5} > $OUT/wino_ab5.log 2>&1; echo "ab rc=$?"; cat $OUT/wino_ab5.log | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    k, _, j = l.partition(' ')
    try: d = json.loads(j)
    except Exception: print(l.strip()); continue
    print(k, ' '.join(f\"{v}:{d[v]['ms']}/{d[v]['executed_tflops']}{'' if d[v].get('bit_equal_to_first', True) else ' DIFF'}\" for v in d))
"
timeout 300 python scripts/wino_fixed_cost.py ${FC:-86 85} > $OUT/wino_fixed_cost.log 2>&1; echo "fixed rc=$?"; grep -v amdgpu.ids $OUT/wino_fixed_cost.log | python -c "
import sys, json
for l in sys.stdin:
    parts = l.split(' ', 2)
    try: d = json.loads(parts[2])
    except Exception: print(l.strip()); continue
    g0, g1 = d['grp0'], d['grp1']
    print(parts[0], parts[1], d['ms_per_launch'], 'cyc/tile', d['workgroup_cycles_per_tile'], 'loop/chunk', round(g0['chunk_loop'] / d['chunks'], 1), 'fixed', d['fixed_cycles_per_tile'], [round(g0[k]) for k in g0], [round(g1[k]) for k in g1])
"
timeout 600 python bench.py --no-cpu-baseline --train-steps 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "
import json; b=json.load(open('$OUT/bench.json')); print(b['value'], b['ms_per_step'], b['roofline']['frac'], b['train']['value'], b['train']['ms_per_step'])"
