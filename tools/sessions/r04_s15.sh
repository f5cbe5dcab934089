#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sharding.py -q -x 2>&1 | tail -12 > gpurun_out/r04_s15_tests.log
timeout 600 python bench.py --parallel columns --no-cpu-baseline --no-roofline --no-sub-records --steps 20 2>/dev/null | tail -1 > gpurun_out/r04_s15_columns_one_rank.json
timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-sub-records --steps 20 2>/dev/null | tail -1 > gpurun_out/r04_s15_replica.json
cat gpurun_out/r04_s15_tests.log; python - <<'PY'
import json
for f in ('gpurun_out/r04_s15_columns_one_rank.json','gpurun_out/r04_s15_replica.json'):
    d=json.loads(open(f).read()); print(f, d['value'], d['ms_per_step'], d.get('column_parallel'))
PY
