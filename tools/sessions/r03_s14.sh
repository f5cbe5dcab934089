#!/bin/bash
# Round 3, GPU session 14: batch tiles with x through LDS instead of DPP: parity, then same-box A/B on the 13B shapes
O=gpurun_out/r03_s14; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_parity.py tests/test_gpu_linear.py tests/test_gpu_decoder_layer.py tests/test_gpu_property.py -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for lib in squeezellm_amd/ab/libxlds0.so squeezellm_amd/libsqllm_hip_ablation.so; do
  for shp in 5120x13824 13824x5120; do
    for sp in "0 0" "0.0045 10"; do
      set -- $sp
      echo "== $lib $shp sparse $1" | tee -a $O/batch_sweep.txt
      SQLLM_LIB=$lib timeout 300 python tools/batch_sweep.py --shape $shp --sparse $1 --topx $2 --batches 2,4,8 --paths tile --reps 3 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['batch'], d['path'], d['us_mean'])" | tee -a $O/batch_sweep.txt
    done
  done
done
