#!/bin/bash
# Round 3, GPU session 4: first run of the streaming batch-1 kernel (4-bit): parity, then A/B against the fused kernel
O=gpurun_out/r03_s4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_decoder_layer.py tests/test_gpu_module.py -x -q > $O/pytest_a.txt 2>&1; tail -15 $O/pytest_a.txt
for opt in "stream=0" "stream=1" "stream=0" "stream=1"; do
  SQLLM_OPTIONS=$opt timeout 200 python bench.py --no-cpu-baseline --no-sub-records 2>>$O/bench.err | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$opt', d['value'], d['roofline']['frac'], {k: v['us_mean'] for k, v in d['per_layer_us'].items()})" | tee -a $O/ab.txt
done
for opt in "stream=0" "stream=1"; do
  SQLLM_OPTIONS=$opt timeout 200 python bench.py --config 7b-w4-s45 --no-cpu-baseline --no-sub-records 2>>$O/bench.err | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$opt s45', d['value'], d['roofline']['frac'], {k: v['us_mean'] for k, v in d['per_layer_us'].items()})" | tee -a $O/ab.txt
done
