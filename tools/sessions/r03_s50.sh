#!/bin/bash
O=gpurun_out/r03_s50; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decoder_layer.py tests/test_gpu_module.py tests/test_gpu_linear.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do for lib in squeezellm_amd/ab/prev.so squeezellm_amd/libsqllm_hip.so; do
  SQLLM_LIB=$lib timeout 200 python bench.py --config 13b-w4-s45 --no-cpu-baseline --no-sub-records 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', d['config']['config_name'], d['value'], d['repeats']['value_median'], {k: v['us_mean'] for k, v in d['per_layer_us'].items()})" | tee -a $O/plan13b.txt
done; done
