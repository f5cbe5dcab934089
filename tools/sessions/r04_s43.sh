#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_property.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r04_s43_mid_rows_fused.txt
timeout 600 python tools/experiments/mid_rows_fused_sparse.py >> gpurun_out/r04_s43_mid_rows_fused.txt 2>&1
cat gpurun_out/r04_s43_mid_rows_fused.txt
