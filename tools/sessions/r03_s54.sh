#!/bin/bash
O=gpurun_out/r03_s54; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for lib in squeezellm_amd/ab/prev.so squeezellm_amd/libsqllm_hip.so; do
SQLLM_LIB=$lib timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
s = d['sub_records']['13b-w4-s45-batched']
print('$lib', d['value'], {b: (s[b]['ms_per_decoder_layer'], {k: v['us_mean'] for k, v in s[b]['per_layer_us'].items()}) for b in ('batch2', 'batch4', 'batch8')})" | tee -a $O/aligned_router.txt
done
