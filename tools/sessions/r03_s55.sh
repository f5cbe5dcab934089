#!/bin/bash
# Round 3, GPU session 55: matrix-core kernel with tile-aligned ranges where they fit (prev.so = contiguous ranges only)
O=gpurun_out/r03_s55; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_property.py -x -q -m gpu 2>&1 | tail -2
for lib in squeezellm_amd/ab/prev.so squeezellm_amd/libsqllm_hip.so; do
 for shp in 13824x5120 11008x4096 5120x13824 22016x8192; do
  SQLLM_LIB=$lib timeout 300 python tools/batch_sweep.py --shape $shp --paths mfma --batches 16,64,2048 --sparse 0 --topx 0 --reps 3 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib', d['shape'], 'rows', d['batch'], d['path'], 'ev', d.get('us_mean'), 'TFLOPs', d.get('TFLOPs'))" | tee -a $O/mfma_aligned.txt
 done
done
