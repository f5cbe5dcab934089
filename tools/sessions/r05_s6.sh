#!/bin/bash
# round 5, session 6: what the transposition launch costs (timing-only knob: the walk reads a stale transposed copy)
mkdir -p gpurun_out
E=tools/experiments/small_batch_r05.py
(timeout 300 python $E --rows 5,8,16 --sets "default;sparse_transpose=2" 2>&1 | grep '^{') > gpurun_out/r05_s6.txt
(timeout 300 python $E --rows 8,16 --dense-only 2>&1 | grep '^{') >> gpurun_out/r05_s6.txt
cat gpurun_out/r05_s6.txt
