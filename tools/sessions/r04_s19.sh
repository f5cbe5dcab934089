#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_batched.py tests/test_gpu_property.py -q -x 2>&1 | tail -5 > gpurun_out/r04_s19_tests.log
timeout 900 python bench.py > gpurun_out/r04_s19_bench.json 2>/dev/null
cat gpurun_out/r04_s19_tests.log; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_s19_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], {k:v.get("tokens_per_s") for k,v in d["drop_in"].items() if isinstance(v,dict) and "tokens_per_s" in v})
print({k:v["ms_per_decoder_layer"] for k,v in d["sub_records"]["13b-w4-s45-batched"].items() if isinstance(v,dict)})
print(d["cpu_baseline"].get("cgroup_cpu_quota_cores"), d["cpu_baseline"].get("affinity_cores"), d["cpu_baseline"]["value"])
PY
