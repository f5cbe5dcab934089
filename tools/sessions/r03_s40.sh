#!/bin/bash
O=gpurun_out/r03_s40; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py -x -q -m gpu 2>&1 | tail -3
for opt in "cols_groups=0" "cols_groups=1"; do
SQLLM_OPTIONS=$opt timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
s = d['sub_records']['13b-w4-s45-batched']
print('$opt', d['value'], {b: (s[b]['ms_per_decoder_layer'], {k: v['us_mean'] for k, v in s[b]['per_layer_us'].items()}) for b in ('batch2', 'batch4', 'batch8')})" | tee -a $O/cols_groups.txt
done
for bits in 3; do for B in 2 4 8 16; do for opt in "cols_groups=0" "cols_groups=1"; do
  for spec in "4096x4096 3" "4096x11008 2" "5120x5120 3" "5120x13824 2"; do set -- $spec
  SQLLM_OPTIONS=$opt timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits $bits --batch $B --sparse 0.0045 --topx 10 --reps 3 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('w$bits', '$opt', d['shape'], 'x', d['group'], 'rows', d['batch'], 'ev', d['us_mean'], 'wall', d['wall_us'])" | tee -a $O/cols_groups.txt
  done; done; done; done
