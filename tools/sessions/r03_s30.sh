#!/bin/bash
# Round 3, GPU session 30: is the batch-1 kernel limited by workgroups per CU?  (unused dynamic LDS caps them at 3 / 2)
O=gpurun_out/r03_s30; mkdir -p $O
AB=squeezellm_amd/libsqllm_hip_ablation.so
for bits in 4 3; do
for spec in "4096x4096 1" "4096x4096 3" "4096x11008 2" "11008x4096 1"; do
  set -- $spec
  for pad in 0 30720 46080; do
    for tw in 0 512 768; do
    SQLLM_OPTIONS="lds_pad=$pad" SQLLM_LIB=$AB timeout 300 python tools/sweep.py --shapes $1 --group $2 --bits $bits --target-wgs $tw --reps 3 2>>$O/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('w$bits', d['shape'], 'x', d['group'], 'lds_pad $pad', 'target_wgs', d['target_wgs'], 'k_slices', d['k_slices'], 'wall', d['wall_us'], 'ev', d['us_mean'])" | tee -a $O/occupancy.txt
    done
  done
done
done
