#!/bin/bash
R=$PWD
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r04_s33_gpu_tests.log
cat gpurun_out/r04_s33_gpu_tests.log
timeout 600 python bench.py > gpurun_out/r04_s33_bench.json 2> gpurun_out/r04_s33_bench.err
python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r04_s33_bench.json') if l.startswith('{')][-1])
print({k:r[k] for k in ('value','ms_per_step')}, r['roofline']['frac'], r['roofline']['avg_kernel_us'])
b=r.get('sub_records',{}).get('13b-w4-s45-batched',{})
for k,v in b.items():
    if isinstance(v,dict): print(k, v.get('ms_per_decoder_layer'), v.get('dense_TFLOPs_wall'), {kk:vv['us_mean'] for kk,vv in v.get('per_layer_us',{}).items()})
print({k:(v if not isinstance(v,dict) else '...') for k,v in r.get('drop_in',{}).items()})
PY
tail -3 gpurun_out/r04_s33_bench.err
timeout 300 python tools/experiments/split_planes_check.py --timing-only > gpurun_out/r04_s33_wide_final.txt 2>&1
tail -30 gpurun_out/r04_s33_wide_final.txt | cut -c1-230
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o x -- python $R/tools/batch_sweep.py --paths mfma --batches 16,128,512,2048 --reps 2 > /tmp/kt.log 2>&1
python $R/tools/rocprof_summary.py "$(find /tmp/prof_kt -name '*.db' | head -1)" --by-grid --match sqllm --top 16 > $R/gpurun_out/r04_s33_kt_batched.summary.txt
grep '^{' /tmp/kt.log >> $R/gpurun_out/r04_s33_kt_batched.summary.txt
cat $R/gpurun_out/r04_s33_kt_batched.summary.txt
