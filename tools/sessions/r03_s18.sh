#!/bin/bash
# Round 3, GPU session 18: CSR role with lane runs + DPP scans -- parity, then the cost sweep of session 17 again
O=gpurun_out/r03_s18; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
AB=squeezellm_amd/libsqllm_hip_ablation.so
for shp in 13824x5120 5120x13824; do
 for B in 1 2 4 8; do
  for mode in "0 0 0" "0.0045 10 0" "0.0045 10 4"; do
    set -- $mode
    SQLLM_OPTIONS="cols_min_batch=1000" SQLLM_LIB=$AB timeout 200 python tools/sweep.py --shapes $shp --batch $B --bits 4 --sparse $1 --topx $2 --ablate-csr $3 --reps 3 2>>$O/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], 'rows', d['batch'], 'sparse $1 topx $2 ablate_csr $3', 'grid', d['grid'], 'wall', d['wall_us'], 'ev', d['us_mean'])" | tee -a $O/sparse_cost_batch.txt
  done
 done
done
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_s18/bench.json') if l.startswith('{')][-1])
print(d['value'], d['roofline']['frac'])
for k,v in d['sub_records'].items():
    print(k, v.get('value'), (v.get('roofline') or {}).get('frac'))
    if 'batch1' in v:
        print({b: v[b]['ms_per_decoder_layer'] for b in v if b.startswith('batch')})
PY
