#!/bin/bash
# Round 3, GPU session 29: does the dense workgroup count want re-tuning when sparse workgroups share the launch?
O=gpurun_out/r03_s29; mkdir -p $O
for bits in 4 3; do
for spec in "4096x4096 1" "4096x4096 3" "4096x11008 2" "11008x4096 1"; do
  set -- $spec
  timeout 300 python tools/sweep.py --shapes $1 --group $2 --bits $bits --sparse 0.0045 --topx 10 --target-wgs 0,256,384,512,640,768,1024,1280,1536 --reps 3 2>>$O/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('w$bits', d['shape'], 'x', d['group'], 'target_wgs', d['target_wgs'], 'k_slices', d['k_slices'], 'grid/op', d['grid'], 'wall', d['wall_us'], 'ev', d['us_mean'])" | tee -a $O/target_wgs_s45.txt
done
done
