#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_property.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r04_s44.txt
timeout 300 python tools/batch_sweep.py --paths mfma --batches 32,64,128,256,512,2048 --reps 2 >> gpurun_out/r04_s44.txt 2>&1
cat gpurun_out/r04_s44.txt
