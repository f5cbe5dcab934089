#!/bin/bash
# Round 3, GPU session 48: dense workgroup count for the 2- / 4-row batch tiles now that three fit a CU
O=gpurun_out/r03_s48; mkdir -p $O
for B in 2 4; do
for spec in "5120x5120 1" "13824x5120 1" "5120x13824 2" "11008x4096 1" "4096x11008 2" "5120x5120 3"; do set -- $spec
  SQLLM_OPTIONS="cols_min_batch=1000" timeout 300 python tools/sweep.py --shapes $1 --group $2 --bits 4 --batch $B --sparse 0.0045 --topx 10 --target-wgs 0,256,384,512,768,1024,1536 --reps 3 --total-mb 400 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], 'x', d['group'], 'rows', d['batch'], 'target_wgs', d['target_wgs'], 'k_slices', d['k_slices'], 'ev', d['us_mean'], 'wall', d['wall_us'])" | tee -a $O/target_wgs_tiles.txt
done; done
