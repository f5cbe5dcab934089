#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decoder_layer.py tests/test_gpu_batched.py -q -x 2>&1 | tail -6 > gpurun_out/r04_s14_tests.log
timeout 600 python tools/experiments/small_batch_breakdown.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_s14_breakdown.jsonl
timeout 600 python tools/experiments/small_batch_layer.py 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_s14_small_batch_w4.jsonl
cat gpurun_out/r04_s14_tests.log gpurun_out/r04_s14_breakdown.jsonl gpurun_out/r04_s14_small_batch_w4.jsonl
