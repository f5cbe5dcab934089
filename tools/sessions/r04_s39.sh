#!/bin/bash
R=$PWD
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r04_s39_gpu_tests.log
cat gpurun_out/r04_s39_gpu_tests.log
timeout 600 python bench.py > gpurun_out/r04_s39_bench.json 2> gpurun_out/r04_s39_bench.err
python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r04_s39_bench.json') if l.startswith('{')][-1])
print({k:r[k] for k in ('value','ms_per_step')}, r['roofline']['frac'], r['roofline']['avg_kernel_us'])
b=r.get('sub_records',{}).get('13b-w4-s45-batched',{})
for k,v in b.items():
    if isinstance(v,dict): print(k, v.get('ms_per_decoder_layer'), v.get('dense_TFLOPs_wall'), {kk:vv['us_mean'] for kk,vv in v.get('per_layer_us',{}).items()})
print({k:(v if not isinstance(v,dict) else '...') for k,v in r.get('drop_in',{}).items()})
PY
tail -3 gpurun_out/r04_s39_bench.err
cd /tmp && export TMPDIR=/tmp
