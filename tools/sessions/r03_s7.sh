#!/bin/bash
# Round 3, GPU session 7: streaming kernel variants (ring slots / workgroups per CU / slots before the barrier), same box
O=gpurun_out/r03_s7; mkdir -p $O
SQLLM_OPTIONS=stream=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_decoder_layer.py tests/test_gpu_module.py -x -q > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
one() {  # tag, lib, options
  SQLLM_LIB=$2 SQLLM_OPTIONS=$3 timeout 200 python bench.py --no-cpu-baseline --no-sub-records 2>>$O/bench.err | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', d['value'], d['roofline']['frac'], {k: v['us_mean'] for k, v in d['per_layer_us'].items()})" | tee -a $O/ab.txt
}
for rep in 1 2; do
  one fused squeezellm_amd/libsqllm_hip.so stream=0
  for v in r2w4p1 r2w4p2 r3w3p1 r4w2p1 r4w2p2; do one $v squeezellm_amd/ab/lib$v.so stream=1; done
done
AB=squeezellm_amd/ab/libr2w4p1.so
for spec in "4096x4096 3" "4096x11008 2"; do
  set -- $spec
  SQLLM_OPTIONS=stream=1 SQLLM_LIB=$AB timeout 120 python tools/timeline.py --shape $1 --bits 4 --group $2 2>&1 | grep -v amdgpu.ids | tee -a $O/timeline_w4.txt
  SQLLM_OPTIONS=stream=1 SQLLM_LIB=$AB timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --ablate 0,2,4,8 --reps 3 2>>$O/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], 'x', d['group'], 'abl', d['ablate'], 'wall', d['wall_us'], 'ev', d['us_mean'])" | tee -a $O/sweep_ablate.txt
done
