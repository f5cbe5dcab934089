#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -8 > gpurun_out/r04_s17_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_s17_smoke.log 2>&1
cat gpurun_out/r04_s17_gpu_tests.log; tail -3 gpurun_out/r04_s17_smoke.log
