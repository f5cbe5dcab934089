#!/bin/bash
mkdir -p gpurun_out
bash tools/ab_libs.sh "A B" "7b-w4-s0" 3 > gpurun_out/r04_s40_ab_session_libs.txt 2>&1
cat gpurun_out/r04_s40_ab_session_libs.txt
