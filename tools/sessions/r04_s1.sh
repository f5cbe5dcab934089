#!/bin/bash
# round 4, session 1: first run of the dependency-gated pass -- parity, then same-box A/B against grouped launches
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pass.py -x -q 2>&1 | tail -40 > gpurun_out/r04_s1_tests.log
timeout 900 python tools/pass_bench.py --config 7b-w4-s0 --sweep > gpurun_out/r04_s1_w4s0.jsonl 2> gpurun_out/r04_s1_w4s0.err
timeout 400 python tools/pass_bench.py --config 7b-w4-s45 > gpurun_out/r04_s1_w4s45.jsonl 2> gpurun_out/r04_s1_w4s45.err
timeout 400 python tools/pass_bench.py --config 7b-w3-s45 > gpurun_out/r04_s1_w3s45.jsonl 2> gpurun_out/r04_s1_w3s45.err
tail -5 gpurun_out/r04_s1_tests.log; cat gpurun_out/r04_s1_w4s0.jsonl | head -40; tail -3 gpurun_out/r04_s1_w4s0.err
