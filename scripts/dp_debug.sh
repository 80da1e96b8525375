# (debug aid) the 2-rank gloo DP test under different kernel choices
for env in "X=1" "TNV3_WINO43_TRAIN=0" "TNV3_WINO43_TRAIN=0 TNV3_WINO43_DGRAD=0" "TNV3_WINO_REPACK_MULTI=0" "TNV3_WINO43=0"; do
  echo "== $env"; env $env python -m pytest tests/test_gpu_dp.py -q -x -m gpu -k "gloo" 2>&1 | grep -E "passed|failed|assert \(" | head -3
done
