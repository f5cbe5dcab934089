#!/bin/bash
# round 5, session 1: parity of the folded CSR walk + same-box A/B of the 13B s45 layer by rows (HEAD vs the round-4 build)
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -40) > gpurun_out/r05_s1_tests.log
E=tools/experiments/small_batch_r05.py
(timeout 300 python $E --rows 1,2,3,4,5,6,8,12,16 2>&1 | grep '^{') > gpurun_out/r05_s1_new.txt
(SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --rows 1,2,3,4,5,6,8,12,16 2>&1 | grep '^{') > gpurun_out/r05_s1_base.txt
(timeout 300 python $E --rows 2,4 --sets "csr_fold=0;mfma_min_batch=2;cols_min_batch=1073741824;cols_max_batch=4" 2>&1 | grep '^{') > gpurun_out/r05_s1_sweep_small.txt
(timeout 300 python $E --rows 5,8,16 --sets "small_wgs_per_cu=2;small_wgs_per_cu=4;small_wgs_per_cu=6;mfma_min_batch=1048576" 2>&1 | grep '^{') > gpurun_out/r05_s1_sweep_split.txt
(timeout 300 python $E --dense-only --rows 1,2,4,8,16 2>&1 | grep '^{') > gpurun_out/r05_s1_dense_only.txt
(timeout 300 python $E --bits 3 --rows 2,4,8,16 2>&1 | grep '^{') > gpurun_out/r05_s1_new_w3.txt
(SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --bits 3 --rows 2,4,8,16 2>&1 | grep '^{') > gpurun_out/r05_s1_base_w3.txt
tail -5 gpurun_out/r05_s1_tests.log; cat gpurun_out/r05_s1_new.txt gpurun_out/r05_s1_base.txt
