#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/experiments/pass_debug2.py 18 > gpurun_out/r04_s4_debug2.log 2>&1
cat gpurun_out/r04_s4_debug2.log | head -60
