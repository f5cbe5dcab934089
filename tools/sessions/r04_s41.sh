#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batched.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r04_s41.txt
timeout 600 python bench.py 2>/dev/null | grep '^{' > gpurun_out/r04_s41_bench.json
python - >> gpurun_out/r04_s41.txt <<'PY'
import json
r=json.loads(open('gpurun_out/r04_s41_bench.json').read())
print({k:r[k] for k in ('value','ms_per_step')}, r['roofline']['frac'])
b=r['sub_records']['13b-w4-s45-batched']
for k,v in b.items():
    if isinstance(v,dict): print(k, v.get('ms_per_decoder_layer'), v.get('dense_TFLOPs_wall'), {kk:vv['us_mean'] for kk,vv in v.get('per_layer_us',{}).items()})
PY
cat gpurun_out/r04_s41.txt
