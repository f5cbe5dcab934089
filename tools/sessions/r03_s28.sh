#!/bin/bash
# Round 3, GPU session 28: does WHEN the chunk workgroups run matter?  (role held back by s_sleep, measurement build)
O=gpurun_out/r03_s28; mkdir -p $O
AB=squeezellm_amd/libsqllm_hip_ablation.so
for spec in "4096x4096 1" "4096x4096 3" "4096x11008 2" "11008x4096 1"; do
  for delay in 0 2 4 8 16; do
    for last in 0 1; do
    set -- $spec
    SQLLM_OPTIONS="sparse_last=$last" SQLLM_LIB=$AB timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --sparse 0.0045 --topx 10 --ablate-csr $((delay * 16)) --reps 3 2>>$O/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], 'x', d['group'], 'sparse_last $last hold-back ~%.1f us' % ($delay * 0.21), 'wall', d['wall_us'], 'ev', d['us_mean'])" | tee -a $O/csr_delay.txt
    done
  done
done
