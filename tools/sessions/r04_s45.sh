#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batched.py -q -m gpu -x -k "many_row_blocks" 2>&1 | tail -6 > gpurun_out/r04_s45.txt
cat gpurun_out/r04_s45.txt
