#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py -m gpu -q --maxfail=30 2>&1 | tail -4) > gpurun_out/r05_s15_tests.log
E=tools/experiments/small_batch_r05.py
(timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') > gpurun_out/r05_s15.txt
(SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s15.txt
tail -2 gpurun_out/r05_s15_tests.log; cat gpurun_out/r05_s15.txt
