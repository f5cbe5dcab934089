#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_checkpoint.py tests/test_gpu_linear.py tests/test_gpu_parity.py -q -x 2>&1 | tail -12 > gpurun_out/r04_s10_tests.log
for c in 7b-w4-s45 7b-w4-s0 7b-w3-s45; do timeout 600 python tests/ref_kernel_bench.py --config $c 2>/dev/null | tail -1 > gpurun_out/r04_ref_vs_ours_$c.json; done
python tools/host_cost.py 2>/dev/null | tail -1 > gpurun_out/r04_s10_host_cost.json
cat gpurun_out/r04_s10_tests.log gpurun_out/r04_ref_vs_ours_*.json gpurun_out/r04_s10_host_cost.json
