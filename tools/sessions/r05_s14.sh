#!/bin/bash
# round 5, session 14: full GPU suite + A/B of the 13B s45 layer (walk in one round of 40 gathers; workspace-less names on stream-ordered scratch)
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -8) > gpurun_out/r05_s14_tests.log
E=tools/experiments/small_batch_r05.py
(timeout 300 python $E --rows 1,2,4,5,6,8,12,16 2>&1 | grep '^{') > gpurun_out/r05_s14.txt
(SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --rows 1,2,4,5,6,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s14.txt
(timeout 300 python $E --rows 8,16 --no-ws 2>&1 | grep '^{') >> gpurun_out/r05_s14.txt
(timeout 300 python $E --bits 3 --rows 8,9,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s14.txt
(SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --bits 3 --rows 8,9,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s14.txt
tail -3 gpurun_out/r05_s14_tests.log; cat gpurun_out/r05_s14.txt
