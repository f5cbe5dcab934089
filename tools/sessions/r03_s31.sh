#!/bin/bash
# Round 3, GPU session 31: wide batches -- the sparse launch with the transposed vec (lane = batch row) against the
# gather path (lane runs + DPP scan, rows in groups of 32), 13B gate/up shape, hybrid
O=gpurun_out/r03_s31; mkdir -p $O
for st in 1 0; do
  timeout 600 python tools/batch_sweep.py --paths mfma --batches 9,16,32,64,128,256 --sparse-transpose $st --reps 3 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('sparse_transpose $st', d['shape'], 'rows', d['batch'], d['path'], 'wall', d.get('wall_us'), 'ev', d.get('us_mean'))" | tee -a $O/wide_sparse.txt
done
timeout 600 python tools/batch_sweep.py --paths mfma --batches 9,16,32,64,128,256 --sparse 0 --topx 0 --reps 3 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('dense only', d['shape'], 'rows', d['batch'], d['path'], 'wall', d.get('wall_us'), 'ev', d.get('us_mean'))" | tee -a $O/wide_sparse.txt
