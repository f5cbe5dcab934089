#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/experiments/split_planes_check.py > gpurun_out/r04_s28_wide_product.txt 2>&1
tail -20 gpurun_out/r04_s28_wide_product.txt
