#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/experiments/small_batch_layer.py 3 > gpurun_out/r04_s37_small_batch_layer_w3.txt 2>&1
cat gpurun_out/r04_s37_small_batch_layer_w3.txt
