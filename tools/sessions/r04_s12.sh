#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decoder_layer.py tests/test_gpu_batched.py -q -x 2>&1 | tail -6 > gpurun_out/r04_s12_tests.log
timeout 600 python tools/experiments/small_batch_layer.py 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_s12_small_batch_w4.jsonl
timeout 600 python tools/experiments/small_batch_layer.py 3 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_s12_small_batch_w3.jsonl
cat gpurun_out/r04_s12_tests.log gpurun_out/r04_s12_small_batch_w4.jsonl gpurun_out/r04_s12_small_batch_w3.jsonl
