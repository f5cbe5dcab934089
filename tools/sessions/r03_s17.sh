#!/bin/bash
# Round 3, GPU session 17: what do the sparse roles cost at 2 / 4 / 8 rows?  (batch tiles forced; CSR ablation bits)
O=gpurun_out/r03_s17; mkdir -p $O
AB=squeezellm_amd/libsqllm_hip_ablation.so
for shp in 13824x5120 5120x13824; do
 for B in 2 4 8; do
  for mode in "0 0 0" "0.0045 0 0" "0 10 0" "0.0045 10 0" "0.0045 10 1" "0.0045 10 2" "0.0045 10 4"; do
    set -- $mode
    SQLLM_OPTIONS="cols_min_batch=1000" SQLLM_LIB=$AB timeout 200 python tools/sweep.py --shapes $shp --batch $B --bits 4 --sparse $1 --topx $2 --ablate-csr $3 --reps 3 2>>$O/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], 'rows', d['batch'], 'sparse $1 topx $2 ablate_csr $3', 'grid', d['grid'], 'wall', d['wall_us'], 'ev', d['us_mean'])" | tee -a $O/sparse_cost_batch.txt
  done
 done
done
