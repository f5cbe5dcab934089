#!/bin/bash
# Round 3, GPU session 10: full GPU suite with the round's additions; the full default bench line; column-parallel bench on one rank
O=gpurun_out/r03_s10; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
SECONDS=0; timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench.py default run: $SECONDS s"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_s10/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['roofline']['frac'], d.get('parity_spot'))
print(json.dumps(d.get('drop_in'))[:900])
print(json.dumps(d.get('cpu_baseline'))[:1500])
for k,v in d.get('sub_records',{}).items(): print(k, v.get('value'), (v.get('roofline') or {}).get('frac'))
PY
timeout 300 python bench.py --parallel columns --no-cpu-baseline --no-sub-records 2>$O/cols.err | grep '^{' > $O/bench_columns_1rank.json; python -c "
import json; d=json.loads(open('$O/bench_columns_1rank.json').read()); print('columns 1 rank', d['value'], d['config']['parallelism'], d.get('column_parallel'))"
SQLLM_BENCH_FORCE_DIST=1 timeout 300 python bench.py --parallel columns --no-cpu-baseline --no-sub-records 2>>$O/cols.err | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('columns 1 rank + RCCL init', d['value'], d['config']['rccl_ranks'], d.get('column_parallel'))"
tail -3 $O/cols.err
