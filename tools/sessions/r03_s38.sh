#!/bin/bash
# Round 3, GPU session 38: batch tiles vs column-lane kernel by shape at 2-16 rows (router threshold), hybrid ops
O=gpurun_out/r03_s38; mkdir -p $O
for bits in 4 3; do
 for shp in 4096x4096 5120x5120 8192x8192 11008x4096 4096x11008 13824x5120 5120x13824; do
  b="2,3,4"; [ $bits = 3 ] && b="2,4,8,16"
  timeout 300 python tools/batch_sweep.py --shape $shp --bits $bits --paths tile,cols --batches $b --reps 3 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('w$bits', d['shape'], 'rows', d['batch'], d['path'], 'ev', d.get('us_mean'), 'wall', d.get('wall_us'))" | tee -a $O/tile_vs_cols.txt
 done
done
