#!/bin/bash
mkdir -p gpurun_out
E=tools/experiments/small_batch_r05.py
(timeout 300 python $E --rows 5,8,16 --sets "default;sparse_transpose=2" 2>&1 | grep '^{') > gpurun_out/r05_s12.txt
(SQLLM_LIB=$PWD/squeezellm_amd/libsqllm_hip_ablation.so timeout 200 python tools/experiments/small_split_timeline.py --rows 16 2>&1 | grep -v amdgpu.ids) > gpurun_out/r05_s12_timeline.txt
cat gpurun_out/r05_s12.txt gpurun_out/r05_s12_timeline.txt
