#!/bin/bash
# Round 3, GPU session 53: column-lane kernel with tile-aligned ranges where they fit (prev.so = contiguous ranges only)
O=gpurun_out/r03_s53; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py -x -q -m gpu 2>&1 | tail -2
for lib in squeezellm_amd/ab/prev.so squeezellm_amd/libsqllm_hip.so; do
 for bits in 4 3; do
  for B in 2 4; do
   for spec in "5120x5120 3" "5120x5120 1" "13824x5120 1" "4096x4096 3" "11008x4096 1"; do set -- $spec
   SQLLM_LIB=$lib SQLLM_OPTIONS="cols_min_batch=1,cols_max_batch=16" timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits $bits --batch $B --sparse 0.0045 --topx 10 --reps 3 --total-mb 400 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', 'w$bits', d['shape'], 'x', d['group'], 'rows', d['batch'], 'ev', d['us_mean'], 'wall', d['wall_us'])" | tee -a $O/cols_aligned.txt
   done
  done
 done
done
