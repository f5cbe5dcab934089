#!/bin/bash
mkdir -p gpurun_out
(SQLLM_LIB=$PWD/squeezellm_amd/libsqllm_hip_ablation.so timeout 200 python tools/experiments/small_split_timeline.py --rows 16 2>&1 | grep -v amdgpu.ids; SQLLM_LIB=$PWD/squeezellm_amd/libsqllm_hip_ablation.so timeout 200 python tools/experiments/small_split_timeline.py --rows 16 --dense-only 2>&1 | grep -v amdgpu.ids) > gpurun_out/r05_s13_timeline.txt
cat gpurun_out/r05_s13_timeline.txt
