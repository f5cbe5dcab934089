#!/bin/bash
# Round 3, GPU session 11: column-pair-table kernel (4-bit, 16-wave workgroups): parity forced on every 4-bit batch-1 launch, then A/B
O=gpurun_out/r03_s11; mkdir -p $O
SQLLM_OPTIONS=pair4=1,pair4_min_mb=0 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_decoder_layer.py tests/test_gpu_module.py tests/test_gpu_property.py -x -q > $O/pytest_pair.txt 2>&1; tail -6 $O/pytest_pair.txt
one() {  # tag, options, extra bench args
  SQLLM_OPTIONS=$2 timeout 200 python bench.py --no-cpu-baseline --no-sub-records $3 2>>$O/bench.err | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', d['value'], d['roofline']['frac'], {k: v['us_mean'] for k, v in d['per_layer_us'].items()})" | tee -a $O/ab.txt
}
for rep in 1 2; do
  one fused pair4=0
  one pair4_ge12MB pair4=1
  one pair4_all pair4=1,pair4_min_mb=0
done
one fused_s45 pair4=0 "--config 7b-w4-s45"
one pair4_s45 pair4=1 "--config 7b-w4-s45"
one fused_13b pair4=0 "--config 13b-w4-s45"
one pair4_13b pair4=1 "--config 13b-w4-s45"
AB=squeezellm_amd/libsqllm_hip_ablation.so
for spec in "4096x4096 3" "4096x11008 2" "11008x4096 1"; do
  set -- $spec
  SQLLM_OPTIONS=pair4=1 SQLLM_LIB=$AB timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --ablate 0,2,8,14 --reps 3 2>>$O/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], 'x', d['group'], 'abl', d['ablate'], 'grid', d['grid'], 'wall', d['wall_us'], 'ev', d['us_mean'])" | tee -a $O/sweep_ablate_pair.txt
done
