#!/bin/bash
# Round 3, GPU session 42: two rows, 4-bit: batch tiles (now three workgroups per CU) against the column-lane kernel, ops and groups
O=gpurun_out/r03_s42; mkdir -p $O
for spec in "4096x4096 1" "4096x4096 3" "4096x11008 1" "4096x11008 2" "11008x4096 1" "5120x5120 1" "5120x5120 3" "5120x13824 1" "5120x13824 2" "13824x5120 1" "8192x8192 1" "8192x8192 3" "8192x22016 2" "22016x8192 1"; do set -- $spec
  for opt in "cols_min_batch=1000" "cols_min_batch=1,cols_max_batch=1000"; do
  SQLLM_OPTIONS="$opt" timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --batch 2 --sparse 0.0045 --topx 10 --reps 3 --total-mb 400 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('tile' if '1000' == '$opt'.split('=')[1] else 'cols', d['shape'], 'x', d['group'], 'rows', d['batch'], 'ev', d['us_mean'], 'wall', d['wall_us'])" | tee -a $O/tile_vs_cols_2rows.txt
  done
done
