#!/bin/bash
mkdir -p gpurun_out
v=$(timeout 300 python bench.py --no-cpu-baseline --no-sub-records 2>/dev/null | grep '^{' | python -c "import sys,json; print(int(json.loads(sys.stdin.read())['value']))")
echo "box speed probe: $v tokens/s" | tee gpurun_out/r05_box_probe.txt
if [ "${v:-0}" -lt 868 ]; then echo "slow box: not collecting"; exit 0; fi
bash tools/ab_libs.sh "r04 head" "7b-w4-s0 7b-w4-s45 7b-w3-s45" 2 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_batch1_ab_vs_r04.txt
bash tools/collect_profiles.sh r05 > gpurun_out/r05_collect.log 2>&1
tail -2 gpurun_out/r05_collect.log; cat gpurun_out/r05_batch1_ab_vs_r04.txt
