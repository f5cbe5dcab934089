#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_batched.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r04_s31_slabs.txt
timeout 600 python tools/experiments/split_planes_check.py --timing-only >> gpurun_out/r04_s31_slabs.txt 2>&1
cat gpurun_out/r04_s31_slabs.txt
