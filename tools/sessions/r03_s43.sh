#!/bin/bash
# Round 3, GPU session 43: 4-bit four-row batch tiles with two steps in flight at 74 VGPRs (three workgroups per CU)
# against four steps at 94 (two per CU); then tiles against the column-lane kernel at 3 / 4 rows
O=gpurun_out/r03_s43; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_batched.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
for B in 3 4; do
for spec in "4096x4096 1" "4096x4096 3" "4096x11008 2" "11008x4096 1" "5120x5120 1" "5120x5120 3" "5120x13824 2" "13824x5120 1" "8192x8192 3" "22016x8192 1"; do set -- $spec
  SQLLM_LIB=squeezellm_amd/ab/prev.so SQLLM_OPTIONS="cols_min_batch=1000" timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --batch $B --sparse 0.0045 --topx 10 --reps 3 --total-mb 400 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('tile-before', d['shape'], 'x', d['group'], 'rows', d['batch'], 'ev', d['us_mean'], 'wall', d['wall_us'])" | tee -a $O/bt4.txt
  for opt in "cols_min_batch=1000" "cols_min_batch=1,cols_max_batch=1000"; do
  SQLLM_OPTIONS="$opt" timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --batch $B --sparse 0.0045 --topx 10 --reps 3 --total-mb 400 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('tile' if '1000' == '$opt'.split('=')[1] else 'cols', d['shape'], 'x', d['group'], 'rows', d['batch'], 'ev', d['us_mean'], 'wall', d['wall_us'])" | tee -a $O/bt4.txt
  done
done
done
