#!/bin/bash
# Round 3, GPU session 46: 8-row batch tiles with 4-wave workgroups (five per CU at 96 VGPRs) against 8-wave ones (two per CU)
O=gpurun_out/r03_s46; mkdir -p $O
for lib in squeezellm_amd/ab/libBase.so squeezellm_amd/ab/libW4.so; do
for B in 8 4; do
for spec in "5120x5120 1" "5120x5120 3" "5120x13824 2" "13824x5120 1" "4096x4096 3" "11008x4096 1"; do set -- $spec
  SQLLM_LIB=$lib SQLLM_OPTIONS="cols_min_batch=1000" timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --batch $B --sparse 0.0045 --topx 10 --reps 3 --total-mb 400 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', d['shape'], 'x', d['group'], 'rows', d['batch'], 'ev', d['us_mean'], 'wall', d['wall_us'])" | tee -a $O/waves4.txt
done; done; done
tail -3 $O/err.txt
