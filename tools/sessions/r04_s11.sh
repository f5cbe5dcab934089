#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batched.py -q -x 2>&1 | tail -6 > gpurun_out/r04_s11_tests.log
timeout 900 python tools/experiments/mfma_split_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_s11_mfma_split.jsonl
cat gpurun_out/r04_s11_tests.log gpurun_out/r04_s11_mfma_split.jsonl
