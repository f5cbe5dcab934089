#!/bin/bash
# Round 3, GPU session 9: the fused kernel's TRUE no-decode floor (the half-stage step ignored ablation bit 2)
O=gpurun_out/r03_s9; mkdir -p $O
AB=squeezellm_amd/libsqllm_hip_ablation.so
for spec in "4096x4096 1" "4096x4096 3" "4096x11008 2" "11008x4096 1"; do
  set -- $spec
  SQLLM_LIB=$AB timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --ablate 0,2,1,8,14 --reps 3 2>>$O/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], 'x', d['group'], 'abl', d['ablate'], 'wall', d['wall_us'], 'ev', d['us_mean'], 'min', d['us_min'])" | tee -a $O/sweep_ablate_fused.txt
done
