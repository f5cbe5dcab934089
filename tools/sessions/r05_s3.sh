#!/bin/bash
# round 5, session 3: the walker wave (one wave of every dense workgroup walks the tile's CSR share, groups by LDS ticket)
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py tests/test_gpu_parity.py -m gpu -q --maxfail=30 2>&1 | tail -15) > gpurun_out/r05_s3_tests.log
E=tools/experiments/small_batch_r05.py
(timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') > gpurun_out/r05_s3.txt
(SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s3.txt
(timeout 300 python $E --dense-only --rows 8,16 2>&1 | grep '^{') >> gpurun_out/r05_s3.txt
(timeout 300 python $E --bits 3 --rows 8,16 --sets "default;mfma_min_batch=5" 2>&1 | grep '^{') >> gpurun_out/r05_s3.txt
(SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --bits 3 --rows 8,16 --sets "default;mfma_min_batch=5" 2>&1 | grep '^{') >> gpurun_out/r05_s3.txt
tail -3 gpurun_out/r05_s3_tests.log; cat gpurun_out/r05_s3.txt
