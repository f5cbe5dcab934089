#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/experiments/pass_debug.py > gpurun_out/r04_s3_debug.log 2>&1
SQLLM_LIB=squeezellm_amd/libsqllm_hip_ablation.so timeout 300 python tools/pass_timeline.py --layers 4 --groups 16 > gpurun_out/r04_s3_timeline.log 2>&1
timeout 600 python -m pytest tests/test_gpu_pass.py -q 2>&1 | grep -E "^E  |FAILED|passed|failed|Error" | head -40 > gpurun_out/r04_s3_tests.log
cat gpurun_out/r04_s3_debug.log | head -60; cat gpurun_out/r04_s3_timeline.log; cat gpurun_out/r04_s3_tests.log
