#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r04_s35_gpu_tests.log
cat gpurun_out/r04_s35_gpu_tests.log
timeout 120 python -c "
import __graft_entry__ as g
g.smoke(); print('smoke ok')" 2>&1 | tail -2
