#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_batched.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r04_s26_batched_tests.log
cat gpurun_out/r04_s26_batched_tests.log
