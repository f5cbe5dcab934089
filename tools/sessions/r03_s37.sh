#!/bin/bash
# Round 3, GPU session 37: fused linear with the top-X rows folded into the CSR -- tests, then overhead with / without
O=gpurun_out/r03_s37; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_linear.py tests/test_gpu_module.py tests/test_gpu_decoder_layer.py tests/test_gpu_property.py -x -q -m gpu 2>&1 | tail -3
for f in 0 1; do echo "== FOLD_TOPX=$f"; FOLD_TOPX=$f python tools/linear_overhead.py 2>/dev/null | grep '"topX": 10'; done | tee $O/linear_fold.txt
