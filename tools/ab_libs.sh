#!/bin/bash
# Same-box A/B of prebuilt library variants: squeezellm_amd/ab/lib<V>.so for every V given.
#   bash tools/ab_libs.sh "A B" "7b-w4-s0 7b-w3-s45" [reps]
variants=${1:-"A B"}; configs=${2:-"7b-w4-s0"}; reps=${3:-2}
cp squeezellm_amd/libsqllm_hip.so /tmp/lib_orig.so
for rep in $(seq $reps); do for v in $variants; do cp squeezellm_amd/ab/lib$v.so squeezellm_amd/libsqllm_hip.so; for c in $configs; do timeout 200 python bench.py --config $c --no-cpu-baseline --no-sub-records 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v', d['config']['config_name'], d['value'], {k: v['us_mean'] for k, v in d['per_layer_us'].items()})
"; done; done; done
cp /tmp/lib_orig.so squeezellm_amd/libsqllm_hip.so
