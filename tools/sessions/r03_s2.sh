#!/bin/bash
# Round 3, GPU session 2: kernarg-chain latency; the new full-size grouped parity tests on the current kernels
O=gpurun_out/r03_s2; mkdir -p $O
RAMP_CHAIN_ONLY=1 timeout 60 build/exp/dispatch_ramp > $O/kernarg_chain.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_decoder_layer.py -x -q > $O/pytest_decoder_layer.txt 2>&1
tail -5 $O/pytest_decoder_layer.txt; cat $O/kernarg_chain.txt
