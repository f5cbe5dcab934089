#!/bin/bash
# Boxes differ by +-7 % on the default line; the committed profile set should come from a box of the class the previous
# rounds' sets were taken on.  Probe the box first (15 s), collect (tools/collect_profiles.sh) only if it reads >= $1 tokens/s:
#   gpurun --timeout 2700 -- 'bash tools/collect_if_fast.sh 868 r05'
min=${1:-868}; tag=${2:-r05}
mkdir -p gpurun_out
v=$(timeout 300 python bench.py --no-cpu-baseline --no-sub-records 2>/dev/null | grep '^{' | python -c "import sys,json; print(int(json.loads(sys.stdin.read())['value']))")
echo "box speed probe: $v tokens/s" | tee gpurun_out/${tag}_box_probe.txt
if [ "${v:-0}" -lt "$min" ]; then echo "slow box: not collecting"; exit 0; fi
bash tools/collect_profiles.sh $tag > gpurun_out/${tag}_collect.log 2>&1
tail -2 gpurun_out/${tag}_collect.log
