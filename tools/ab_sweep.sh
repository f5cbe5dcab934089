#!/bin/bash
# Same-box sweep of launch geometry for prebuilt library variants (squeezellm_amd/ab/lib<V>.so).
#   bash tools/ab_sweep.sh "W8 W4"
cp squeezellm_amd/libsqllm_hip.so /tmp/lib_orig.so
for v in ${1:-"A B"}; do cp squeezellm_amd/ab/lib$v.so squeezellm_amd/libsqllm_hip.so; echo "== $v";
(timeout 300 python tools/sweep.py --shapes 4096x4096 --bits 4 --group 1 --total-mb 500 --reps 4 --target-wgs 256,512,1024 2>&1 | grep "^{"; timeout 300 python tools/sweep.py --shapes 4096x4096 --bits 4 --group 3 --total-mb 500 --reps 4 --target-wgs 256,512 2>&1 | grep "^{"; timeout 300 python tools/sweep.py --shapes 4096x11008 --bits 4 --group 2 --total-mb 700 --reps 4 --target-wgs 768,1536 2>&1 | grep "^{"; timeout 300 python tools/sweep.py --shapes 11008x4096 --bits 4 --group 1 --total-mb 500 --reps 4 --target-wgs 768,1536 2>&1 | grep "^{"; timeout 300 python tools/sweep.py --shapes 4096x11008 --bits 3 --sparse 0.0045 --topx 10 --group 2 --total-mb 700 --reps 4 --target-wgs 768,1536 2>&1 | grep "^{") | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], 'w', d['bits'], 'group', d['group'], 'target', d['target_wgs'], 'grid', d['grid'], 'k_slices', d['k_slices'], 'us', d['us_mean'], 'wall', d['wall_us'])
"; done
cp /tmp/lib_orig.so squeezellm_amd/libsqllm_hip.so
