#!/bin/bash
# Round 3, GPU session 52: 4-bit three-op groups at 5-8 rows: 8-row batch tiles against the column-lane kernel
O=gpurun_out/r03_s52; mkdir -p $O
for B in 5 6 8; do
for spec in "4096x4096 3" "5120x5120 3" "8192x8192 3"; do set -- $spec
  for opt in "cols_min_batch=1000" "cols_min_batch=1,cols_max_batch=1000"; do
  SQLLM_OPTIONS="$opt" timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --batch $B --sparse 0.0045 --topx 10 --reps 3 --total-mb 400 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('tile' if '1000' == '$opt'.split('=')[1] else 'cols', d['shape'], 'x', d['group'], 'rows', d['batch'], 'ev', d['us_mean'], 'wall', d['wall_us'])" | tee -a $O/groups_8rows.txt
  done
done; done
