#!/bin/bash
# round 5, session 7: the walk with staged row ids (no data-dependent loop per step)
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py tests/test_gpu_parity.py tests/test_gpu_module.py -m gpu -q --maxfail=30 2>&1 | tail -15) > gpurun_out/r05_s7_tests.log
E=tools/experiments/small_batch_r05.py
(timeout 300 python $E --rows 5,8,12,16 --sets "default;sparse_transpose=2" 2>&1 | grep '^{') > gpurun_out/r05_s7.txt
(timeout 300 python $E --rows 8,16 --no-ws 2>&1 | grep '^{') >> gpurun_out/r05_s7.txt
(timeout 300 python $E --bits 3 --rows 5,8,16 --sets "default;mfma_min_batch=5" 2>&1 | grep '^{') >> gpurun_out/r05_s7.txt
tail -3 gpurun_out/r05_s7_tests.log; cat gpurun_out/r05_s7.txt
