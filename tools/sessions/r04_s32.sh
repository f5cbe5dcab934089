#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py tests/test_gpu_property.py tests/test_gpu_sharding.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r04_s32.txt
timeout 300 python tools/batch_sweep.py --paths mfma --batches 64,128,256,512,2048 --reps 2 >> gpurun_out/r04_s32.txt 2>&1
timeout 300 python tools/batch_sweep.py --paths mfma --batches 64,128,256,512,2048 --reps 2 --sparse 0 --topx 0 >> gpurun_out/r04_s32.txt 2>&1
cat gpurun_out/r04_s32.txt
