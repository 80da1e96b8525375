#!/bin/bash
# SQ / GRBM counters of the inference step (separate PMC passes, --kernel-trace only).  SQ_MODE=train: of the TRAINING step instead, on ONE
# stream (TNV3_WGRAD_OVERLAP=0: a kernel's GRBM_GUI_ACTIVE is then its own), plus its FETCH_SIZE / WRITE_SIZE passes -> train_sq_summary.json
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -oE "\b(SQ_[A-Z0-9_]+|GRBM_[A-Z0-9_]+|TCC_[A-Z0-9_]+|TCP_[A-Z0-9_]+)\b" $OUT/counters_list.txt | sort -u > $OUT/counter_names.txt
wc -l $OUT/counter_names.txt
cd /tmp
MODE="${SQ_MODE:-infer}"
if [ "$MODE" = train ]; then
  export TNV3_WGRAD_OVERLAP=0
  BENCH="--mode train --steps 2 --warmup 1 --strong-steps 0 --no-cpu-baseline"; PFX=train_; STEPS=3
else
  BENCH="--steps 2 --warmup 1 --blocks 1 --no-cpu-baseline --overlap-streams 0 --infer-split 0 --train-steps 0 --extras 0 --layers-out /tmp/sq_layers.json"; PFX=; STEPS=3
fi
run() { name=$PFX$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/pmc_$name -o pmc -- python $REPO/bench.py $BENCH > /dev/null 2> $OUT/pmc_$name.err; echo "pmc $name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum
if [ "$MODE" = train ] && [ "${SQ_TRAIN_TRAFFIC:-1}" = 1 ]; then run fetch FETCH_SIZE; run write WRITE_SIZE; fi
cd $REPO
LIST="sq1 sq2 sq3 tcc"; [ "$MODE" = train ] && LIST="$LIST fetch write"
for d in $LIST; do for f in $(find $OUT/pmc_$PFX$d -name "*.db" 2>/dev/null); do python scripts/rocpd_pmc.py $f $OUT/pmc_$PFX${d}.csv; python scripts/rocpd_summary.py $f $OUT/pmc_$PFX${d}_kernel_stats.csv; done; [ -f $OUT/pmc_$PFX$d.err ] && tail -3 $OUT/pmc_$PFX$d.err; done
if [ "$MODE" = train ]; then
  mkdir -p $OUT/train_sq && for d in sq1 sq2 sq3 tcc; do cp $OUT/pmc_train_$d.csv $OUT/train_sq/pmc_$d.csv 2>/dev/null; done; cp $OUT/lib_sha256.txt $OUT/train_sq/ 2>/dev/null
  python scripts/sq_summary.py $OUT/train_sq $STEPS > $OUT/train_sq_summary.json 2>> $OUT/pmc_train_tcc.err
else
  python scripts/sq_summary.py $OUT 3 > $OUT/infer_sq_summary.json 2>> $OUT/pmc_tcc.err
fi
find $OUT -name "*.db" -delete
