#!/bin/bash
# round 5, session 5: folded CSR walk reading a TRANSPOSED copy of vec from the caller's workspace
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py tests/test_gpu_parity.py tests/test_gpu_module.py -m gpu -q --maxfail=30 2>&1 | tail -15) > gpurun_out/r05_s5_tests.log
E=tools/experiments/small_batch_r05.py
(timeout 300 python $E --rows 2,4,5,8,12,16 2>&1 | grep '^{') > gpurun_out/r05_s5.txt
(timeout 300 python $E --rows 5,8,16 --no-ws 2>&1 | grep '^{') >> gpurun_out/r05_s5.txt
(SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --rows 2,4,5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s5.txt
(timeout 300 python $E --rows 3,4 --sets "mfma_min_batch=3" 2>&1 | grep '^{') >> gpurun_out/r05_s5.txt
(timeout 300 python $E --bits 3 --rows 5,8,16 --sets "default;mfma_min_batch=5" 2>&1 | grep '^{') >> gpurun_out/r05_s5.txt
tail -3 gpurun_out/r05_s5_tests.log; cat gpurun_out/r05_s5.txt
