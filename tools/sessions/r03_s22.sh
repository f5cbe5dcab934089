#!/bin/bash
O=gpurun_out/r03_s22; mkdir -p $O
python tools/debug/dbg_csr.py 2>&1 | grep -v amdgpu.ids
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -v "^E  \|^    \|^$" > $O/pytest_gpu.txt; grep -c FAILED $O/pytest_gpu.txt; grep FAILED $O/pytest_gpu.txt | head -20; tail -2 $O/pytest_gpu.txt
