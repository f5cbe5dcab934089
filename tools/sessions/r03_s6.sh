#!/bin/bash
# Round 3, GPU session 6: streaming kernel v2 (two pieces, buffer loads, ring of 4 slots, 2 WG/CU): parity, timeline, A/B
O=gpurun_out/r03_s6; mkdir -p $O
SQLLM_OPTIONS=stream=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_decoder_layer.py tests/test_gpu_module.py -x -q > $O/pytest_a.txt 2>&1; tail -5 $O/pytest_a.txt
AB=squeezellm_amd/libsqllm_hip_ablation.so
for spec in "4096x4096 1" "4096x4096 3" "4096x11008 2" "11008x4096 1"; do
  set -- $spec
  SQLLM_OPTIONS=stream=1 SQLLM_LIB=$AB timeout 120 python tools/timeline.py --shape $1 --bits 4 --group $2 2>&1 | grep -v amdgpu.ids >> $O/timeline_w4.txt
  SQLLM_OPTIONS=stream=1 SQLLM_LIB=$AB timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --ablate 0,2,4,8 --reps 3 >> $O/sweep_ablate_stream.jsonl 2>>$O/sweep.err
  SQLLM_OPTIONS=stream=1 timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --target-wgs 0,256,384,512 --reps 3 >> $O/sweep_wgs_stream.jsonl 2>>$O/sweep.err
done
for opt in "stream=0" "stream=1" "stream=0" "stream=1"; do
  SQLLM_OPTIONS=$opt timeout 200 python bench.py --no-cpu-baseline --no-sub-records 2>>$O/bench.err | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$opt', d['value'], d['roofline']['frac'], {k: v['us_mean'] for k, v in d['per_layer_us'].items()})" | tee -a $O/ab.txt
done
cat $O/timeline_w4.txt
python - <<'PY'
import json
for f in ("sweep_ablate_stream.jsonl","sweep_wgs_stream.jsonl"):
    for l in open("gpurun_out/r03_s6/"+f):
        d=json.loads(l); print(d['shape'],'x',d['group'],'abl',d['ablate'],'twg',d['target_wgs'],'wall',d['wall_us'],'ev',d['us_mean'])
PY
