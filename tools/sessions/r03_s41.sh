#!/bin/bash
# Round 3, GPU session 41: 4-bit two-row batch tiles at 80 VGPRs (three workgroups per CU) against 82 (two)
O=gpurun_out/r03_s41; mkdir -p $O
for lib in squeezellm_amd/ab/prev.so squeezellm_amd/libsqllm_hip.so; do
  for spec in "4096x4096 1" "4096x4096 3" "4096x11008 2" "11008x4096 1" "5120x5120 1" "5120x5120 3"; do set -- $spec
  SQLLM_OPTIONS="cols_min_batch=1000" SQLLM_LIB=$lib timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --batch 2 --sparse 0.0045 --topx 10 --reps 3 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', d['shape'], 'x', d['group'], 'rows', d['batch'], 'ev', d['us_mean'], 'wall', d['wall_us'])" | tee -a $O/bt2_occupancy.txt
  done
done
for lib in squeezellm_amd/ab/prev.so squeezellm_amd/libsqllm_hip.so; do
SQLLM_LIB=$lib timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
s = d['sub_records']['13b-w4-s45-batched']
print('$lib', d['value'], {b: (s[b]['ms_per_decoder_layer'], {k: v['us_mean'] for k, v in s[b]['per_layer_us'].items()}) for b in ('batch2',)})" | tee -a $O/bt2_occupancy.txt
done
