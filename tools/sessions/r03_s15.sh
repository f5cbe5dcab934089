#!/bin/bash
# Round 3, GPU session 15: whole suite; pipeline bench with the captured tick (1 rank, with and without RCCL init)
O=gpurun_out/r03_s15; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
timeout 300 python bench.py --parallel pipeline --no-cpu-baseline --no-sub-records 2>$O/pipe.err | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('pipeline 1 rank', d['value'], d.get('pipeline_tick_us'), d.get('pipeline_tick_captured'))"
SQLLM_BENCH_FORCE_DIST=1 timeout 300 python bench.py --parallel pipeline --no-cpu-baseline --no-sub-records 2>>$O/pipe.err | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('pipeline 1 rank + RCCL', d['value'], d.get('pipeline_tick_us'), d.get('pipeline_tick_captured'), d['config']['rccl_ranks'])"
SECONDS=0; timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench.py default run: $SECONDS s"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_s15/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['roofline']['frac'], d.get('parity_spot',{}).get('max_rel_err'))
print({k:(v.get('tokens_per_s'), v.get('ms_per_token')) for k,v in d['drop_in'].items() if isinstance(v,dict)})
cb=d['cpu_baseline']; print(cb['value'], cb['kind'], cb['cores'], {k:v.get('value') for k,v in cb['paths'].items()}, cb['config1_opt1.3b_seq128'])
PY
tail -2 $O/pipe.err
