#!/bin/bash
# Round 3, GPU session 3: effect of reading the kernel-argument block in one round (all kernels); full GPU test suite
O=gpurun_out/r03_s3; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 300 python bench.py --no-cpu-baseline 2>$O/bench.err | grep '^{' > $O/bench_7b-w4-s0.json
AB=squeezellm_amd/libsqllm_hip_ablation.so
for spec in "4096x4096 1" "4096x4096 3" "4096x11008 2" "11008x4096 1"; do
  set -- $spec
  SQLLM_LIB=$AB timeout 120 python tools/timeline.py --shape $1 --bits 4 --group $2 >> $O/timeline_w4.txt 2>&1
  SQLLM_LIB=$AB timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --ablate 0,2,8 --reps 3 >> $O/sweep_ablate_w4.jsonl 2>>$O/sweep.err
done
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_s3/bench_7b-w4-s0.json').read())
print(d['value'], d['roofline']['frac'], {k:v['us_mean'] for k,v in d['per_layer_us'].items()})
for k,v in d.get('sub_records',{}).items(): print(k, v.get('value'), v.get('roofline',{}).get('frac'), v.get('us_per_layer_by_batch', ''))
PY
