#!/bin/bash
# round 5, session 8: timeline of the fused small launch (measurement library)
mkdir -p gpurun_out
export SQLLM_LIB=$PWD/squeezellm_amd/libsqllm_hip_ablation.so
T=tools/experiments/small_split_timeline.py
(timeout 200 python $T --rows 16; timeout 200 python $T --rows 8; timeout 200 python $T --rows 16 --no-ws; timeout 200 python $T --rows 16 --dense-only) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_s8_timeline.txt
cat gpurun_out/r05_s8_timeline.txt
