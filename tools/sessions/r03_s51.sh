#!/bin/bash
# Round 3, GPU session 51: dense workgroup count of SINGLE-op batch-1 launches (13B / 65B o_proj and down), 4- and 3-bit
O=gpurun_out/r03_s51; mkdir -p $O
for bits in 4 3; do
for spec in "5120x5120 1" "13824x5120 1" "8192x8192 1" "22016x8192 1"; do set -- $spec
  timeout 300 python tools/sweep.py --shapes $1 --group $2 --bits $bits --sparse 0.0045 --topx 10 --target-wgs 0,256,384,512,768,1024,1536 --reps 4 --total-mb 500 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('w$bits', d['shape'], 'x', d['group'], 'target_wgs', d['target_wgs'], 'k_slices', d['k_slices'], 'ev', d['us_mean'], 'min', d['us_min'], 'wall', d['wall_us'])" | tee -a $O/target_wgs_singles_b1.txt
done; done
