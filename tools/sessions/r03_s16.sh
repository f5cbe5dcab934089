#!/bin/bash
# Round 3, GPU session 16: what do the sparse roles cost the 7B s45 launches?  (CSR ablation bits; CSR only vs hybrid)
O=gpurun_out/r03_s16; mkdir -p $O
AB=squeezellm_amd/libsqllm_hip_ablation.so
for spec in "4096x4096 1" "4096x4096 3" "4096x11008 2" "11008x4096 1"; do
  set -- $spec
  for mode in "0 0 0" "0.0045 0 0" "0.0045 10 0" "0.0045 10 1" "0.0045 10 2" "0.0045 10 4"; do
    set -- $spec $mode
    SQLLM_LIB=$AB timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --sparse $3 --topx $4 --ablate-csr $5 --reps 3 2>>$O/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], 'x', d['group'], 'sparse $3 topx $4 ablate_csr $5', 'grid', d['grid'], 'wall', d['wall_us'], 'ev', d['us_mean'])" | tee -a $O/sparse_cost.txt
  done
done
