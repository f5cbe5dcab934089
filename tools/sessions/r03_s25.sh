#!/bin/bash
# Round 3, GPU session 25: lockstep row searches in the CSR role (product) against the same library without them
# (ab/libT256.so), after the parity suite
O=gpurun_out/r03_s25; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
for rep in 1 2; do
 for v in squeezellm_amd/ab/libT256.so squeezellm_amd/libsqllm_hip.so; do
  for c in 7b-w4-s45 7b-w3-s45; do
    SQLLM_LIB=$v timeout 200 python bench.py --config $c --no-cpu-baseline --no-sub-records 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v', d['config']['config_name'], d['value'], d['repeats']['value_median'], {k: v['us_mean'] for k, v in d['per_layer_us'].items()})" | tee -a $O/ab_search.txt
  done
 done
done
for v in squeezellm_amd/ab/libT256.so squeezellm_amd/libsqllm_hip.so; do
SQLLM_LIB=$v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v', d['value'], {b: v['ms_per_decoder_layer'] for b, v in d['sub_records']['13b-w4-s45-batched'].items() if b.startswith('batch')})" | tee -a $O/ab_search.txt
done
