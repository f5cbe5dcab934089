#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pass.py -q 2>&1 | grep -E "^E  |FAILED|passed|failed|Error" | head -30 > gpurun_out/r04_s7_tests.log
SQLLM_LIB=squeezellm_amd/libsqllm_hip_ablation.so timeout 300 python tools/pass_timeline.py --layers 4 --groups 12 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_s7_timeline.log
SQLLM_LIB=squeezellm_amd/libsqllm_hip_ablation.so SQLLM_OPTIONS=groups_per_wave=64,pass_wgs_per_cu=2 timeout 300 python tools/pass_timeline.py --layers 4 --groups 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_s7_timeline_gpw64_2cu.log
timeout 900 python tools/pass_bench.py --config 7b-w4-s0 --sweep 2>/dev/null > gpurun_out/r04_s7_w4s0.jsonl
cat gpurun_out/r04_s7_tests.log gpurun_out/r04_s7_timeline.log gpurun_out/r04_s7_timeline_gpw64_2cu.log; python - <<'PY'
import json
for l in open('gpurun_out/r04_s7_w4s0.jsonl'):
    d=json.loads(l); print({k:d[k] for k in d if k in ('path','tag','items','grid','ms_per_token','tokens_per_s','status','pass_poll_sleep','pass_wgs_per_cu','target_wgs','groups_per_wave')})
PY
