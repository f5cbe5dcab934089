#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/experiments/wide_ablate/run.py run > gpurun_out/r04_s27_wide_lds.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_batched.py -q -m gpu -x -k "wide_form or ppl_eval" 2>&1 | tail -8 >> gpurun_out/r04_s27_wide_lds.txt
cat gpurun_out/r04_s27_wide_lds.txt
