#!/bin/bash
# Round 3, GPU session 24: CSR chunk of 2048 / 4096 non-zeros (4 / 8 per lane) and top-X slabs of 256 / 512 rows,
# parity of the variants, then same-box A/B on the s45 configs
O=gpurun_out/r03_s24; mkdir -p $O
for v in C2048 C4096 T512; do
  SQLLM_LIB=squeezellm_amd/ab/lib$v.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batched.py tests/test_gpu_linear.py -q -m gpu -x 2>&1 | tail -2 | sed "s/^/$v: /" | tee -a $O/variants.txt
done
for rep in 1 2; do
 for v in T128 T256 T512 C2048 C2048T512 C4096; do
  for c in 7b-w4-s45 7b-w3-s45; do
    SQLLM_LIB=squeezellm_amd/ab/lib$v.so timeout 200 python bench.py --config $c --no-cpu-baseline --no-sub-records 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v', d['config']['config_name'], d['value'], d['repeats']['value_median'], {k: v['us_mean'] for k, v in d['per_layer_us'].items()})" | tee -a $O/variants.txt
  done
 done
done
v=C2048; SQLLM_LIB=squeezellm_amd/ab/lib$v.so timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v', d['value'], {b: v['ms_per_decoder_layer'] for b, v in d['sub_records']['13b-w4-s45-batched'].items() if b.startswith('batch')})" | tee -a $O/variants.txt
v=T128; SQLLM_LIB=squeezellm_amd/ab/lib$v.so timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v', d['value'], {b: v['ms_per_decoder_layer'] for b, v in d['sub_records']['13b-w4-s45-batched'].items() if b.startswith('batch')})" | tee -a $O/variants.txt
