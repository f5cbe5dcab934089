#!/bin/bash
# Round 3, GPU session 20: which tests does the lane-run CSR role fail, and what do the x gathers cost at 8 rows?
O=gpurun_out/r03_s20; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_parity.py -q -m gpu 2>&1 | grep -v "^E  \|^    \|^$" > $O/pytest_gpu.txt; grep -c FAILED $O/pytest_gpu.txt; grep FAILED $O/pytest_gpu.txt | head -40; tail -2 $O/pytest_gpu.txt
AB=squeezellm_amd/libsqllm_hip_ablation.so
for shp in 13824x5120; do
 for B in 2 8; do
  for mode in "0 0 0" "0.0045 0 0" "0.0045 0 4" "0.0045 0 8"; do
    set -- $mode
    SQLLM_OPTIONS="cols_min_batch=1000" SQLLM_LIB=$AB timeout 200 python tools/sweep.py --shapes $shp --batch $B --bits 4 --sparse $1 --topx $2 --ablate-csr $3 --reps 3 2>>$O/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], 'rows', d['batch'], 'sparse $1 topx $2 ablate_csr $3', 'grid', d['grid'], 'wall', d['wall_us'], 'ev', d['us_mean'])" | tee -a $O/sparse_cost_batch.txt
  done
 done
done
