#!/bin/bash
# Round 3, GPU session 45: 3-bit four-row batch tile at 80 VGPRs + 2 spilled registers (three workgroups per CU) against 88 (two)
O=gpurun_out/r03_s45; mkdir -p $O
for lib in squeezellm_amd/ab/libBase.so squeezellm_amd/ab/libW3BT4x6.so; do
for B in 3 4; do
for spec in "4096x4096 1" "4096x4096 3" "4096x11008 2" "11008x4096 1" "5120x5120 3" "5120x13824 2" "13824x5120 1"; do set -- $spec
  SQLLM_LIB=$lib SQLLM_OPTIONS="cols_min_batch=1000" timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 3 --batch $B --sparse 0.0045 --topx 10 --reps 3 --total-mb 400 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', d['shape'], 'x', d['group'], 'rows', d['batch'], 'ev', d['us_mean'], 'wall', d['wall_us'])" | tee -a $O/w3bt4.txt
done; done; done
