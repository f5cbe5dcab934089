#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/experiments/split_planes_check.py > gpurun_out/r04_s23_wide.txt 2>&1
tail -48 gpurun_out/r04_s23_wide.txt
