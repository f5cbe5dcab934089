#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_decoder_layer.py tests/test_gpu_batched.py tests/test_gpu_property.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r04_s38.txt
timeout 300 python tools/experiments/small_batch_layer.py 4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['batch'], r['default'])" >> gpurun_out/r04_s38.txt
cat gpurun_out/r04_s38.txt
