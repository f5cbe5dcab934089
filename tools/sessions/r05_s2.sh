#!/bin/bash
# round 5, session 2: dense-only A/B of the small split launch: round-4 build vs 80-register build vs the same code at 128 registers
mkdir -p gpurun_out
E=tools/experiments/small_batch_r05.py
(SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --dense-only --rows 8,16 2>&1 | grep '^{') > gpurun_out/r05_s2_dense.txt
(timeout 300 python $E --dense-only --rows 8,16 --sets "small_wgs_per_cu=2;small_wgs_per_cu=3" 2>&1 | grep '^{') >> gpurun_out/r05_s2_dense.txt
(SQLLM_LIB=$PWD/squeezellm_amd/ab/libcap4.so timeout 300 python $E --dense-only --rows 8,16 --sets "small_wgs_per_cu=2;small_wgs_per_cu=3" 2>&1 | grep '^{') >> gpurun_out/r05_s2_dense.txt
(SQLLM_LIB=$PWD/squeezellm_amd/ab/libcap4.so timeout 300 python $E --rows 8,16 --sets "small_wgs_per_cu=2" 2>&1 | grep '^{') >> gpurun_out/r05_s2_dense.txt
cat gpurun_out/r05_s2_dense.txt
