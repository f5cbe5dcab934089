#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/experiments/wide_ablate/run.py run > gpurun_out/r04_s24_wide_ablate.txt 2>&1
cat gpurun_out/r04_s24_wide_ablate.txt
