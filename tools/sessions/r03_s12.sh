#!/bin/bash
# Round 3, GPU session 12: pair-table kernel with conflict-free lookups: parity, A/B, ablations
O=gpurun_out/r03_s12; mkdir -p $O
SQLLM_OPTIONS=pair4=1,pair4_min_mb=0 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_decoder_layer.py -x -q > $O/pytest_pair.txt 2>&1; tail -3 $O/pytest_pair.txt
one() {
  SQLLM_OPTIONS=$2 timeout 200 python bench.py --no-cpu-baseline --no-sub-records $3 2>>$O/bench.err | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', d['value'], d['roofline']['frac'], {k: v['us_mean'] for k, v in d['per_layer_us'].items()})" | tee -a $O/ab.txt
}
for rep in 1 2; do
  one fused pair4=0
  one pair4_ge12MB pair4=1
done
AB=squeezellm_amd/libsqllm_hip_ablation.so
for spec in "4096x4096 3" "4096x11008 2" "11008x4096 1"; do
  set -- $spec
  for tw in 0 256 384; do
  SQLLM_OPTIONS=pair4=1 SQLLM_LIB=$AB timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --ablate 0,2 --target-wgs $tw --reps 3 2>>$O/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], 'x', d['group'], 'abl', d['ablate'], 'twg', d['target_wgs'], 'grid', d['grid'], 'wall', d['wall_us'], 'ev', d['us_mean'])" | tee -a $O/sweep_ablate_pair.txt
  done
done
