#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/experiments/pass_debug.py > gpurun_out/r04_s2_debug.log 2>&1
timeout 600 python -m pytest tests/test_gpu_pass.py -q 2>&1 | tail -15 > gpurun_out/r04_s2_tests.log
timeout 900 python tools/pass_bench.py --config 7b-w4-s0 --sweep > gpurun_out/r04_s2_w4s0.jsonl 2> gpurun_out/r04_s2_w4s0.err
cat gpurun_out/r04_s2_debug.log | head -50; tail -8 gpurun_out/r04_s2_tests.log; python - <<'PY'
import json
for l in open('gpurun_out/r04_s2_w4s0.jsonl'):
    d=json.loads(l); print({k:d[k] for k in d if k in ('path','tag','items','grid','ms_per_token','tokens_per_s','status','pass_poll_sleep','pass_wgs_per_cu','target_wgs','groups_per_wave')})
PY
