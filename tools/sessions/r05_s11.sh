#!/bin/bash
# round 5, session 11: top-X slabs out of the transposed vec, 8-16 workgroups per op, dense ranges planned beside them
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py tests/test_gpu_parity.py -m gpu -q --maxfail=30 2>&1 | tail -5) > gpurun_out/r05_s11_tests.log
E=tools/experiments/small_batch_r05.py
(timeout 300 python $E --rows 5,8,12,16 --sets "default;sparse_transpose=2" 2>&1 | grep '^{') > gpurun_out/r05_s11.txt
(timeout 300 python $E --rows 8,16 --no-ws 2>&1 | grep '^{') >> gpurun_out/r05_s11.txt
(timeout 300 python $E --bits 3 --rows 8,16 --sets "default;mfma_min_batch=5" 2>&1 | grep '^{') >> gpurun_out/r05_s11.txt
(SQLLM_LIB=$PWD/squeezellm_amd/libsqllm_hip_ablation.so timeout 200 python tools/experiments/small_split_timeline.py --rows 16 2>&1 | grep -v amdgpu.ids) > gpurun_out/r05_s11_timeline.txt
tail -2 gpurun_out/r05_s11_tests.log; cat gpurun_out/r05_s11.txt gpurun_out/r05_s11_timeline.txt
