#!/bin/bash
# round 5, session 16: full GPU suite incl. the non-finite cases; shim through the workspace
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -q --maxfail=40 2>&1 | tail -60) > gpurun_out/r05_s16_tests.log
tail -40 gpurun_out/r05_s16_tests.log
