#!/bin/bash
# Round 3, final validation on a fresh box: build entry point, smoke, the whole GPU suite, the default bench line
O=gpurun_out/r03_final; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build_smoke.txt 2>&1; tail -4 $O/build_smoke.txt
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
SECONDS=0; timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench.py default run: $SECONDS s"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_final/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['repeats']['value_median'], d['roofline']['frac'], d['roofline']['traffic'], d['parity_spot']['max_rel_err'], d['parity_spot']['ok'])
cb=d['cpu_baseline']; print(cb['value'], cb['kind'], cb['cores'], {k:(v.get('value'), v.get('threads')) for k,v in cb['paths'].items()})
print(cb['config1_opt1.3b_seq128'])
for k,v in d['sub_records'].items(): print(k, v.get('value'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('traffic'))
PY
