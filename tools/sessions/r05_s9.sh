#!/bin/bash
# round 5, session 9: one round of workgroups including the top-X slabs
mkdir -p gpurun_out
E=tools/experiments/small_batch_r05.py
(timeout 300 python $E --rows 5,8,12,16 --sets "default;sparse_transpose=2" 2>&1 | grep '^{') > gpurun_out/r05_s9.txt
(SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s9.txt
(timeout 300 python $E --rows 8,16 --dense-only 2>&1 | grep '^{') >> gpurun_out/r05_s9.txt
(timeout 300 python $E --bits 3 --rows 8,16 --sets "default;mfma_min_batch=5" 2>&1 | grep '^{') >> gpurun_out/r05_s9.txt
cat gpurun_out/r05_s9.txt
