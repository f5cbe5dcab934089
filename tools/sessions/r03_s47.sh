#!/bin/bash
# Round 3, GPU session 47: the whole round on one box -- library of the last round-2 commit against HEAD, default bench line
O=gpurun_out/r03_s47; mkdir -p $O
for rep in 1 2; do
for lib in squeezellm_amd/ab/r2.so squeezellm_amd/libsqllm_hip.so; do
SQLLM_LIB=$lib timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
s = d['sub_records']
print('$lib', '7b-w4-s0', d['value'], d['roofline']['frac'], '| 7b-w4-s45', s['7b-w4-s45']['value'], s['7b-w4-s45']['roofline']['frac'], '| 7b-w3-s45', s['7b-w3-s45']['value'], s['7b-w3-s45']['roofline']['frac'], '| 13B s45 layer us at 1/2/4/8 rows', [round(1e3 * s['13b-w4-s45-batched'][b]['ms_per_decoder_layer'], 1) for b in ('batch1', 'batch2', 'batch4', 'batch8')], '| drop_in', {k: v.get('tokens_per_s') for k, v in d['drop_in'].items() if isinstance(v, dict)})" | tee -a $O/r2_vs_r3.txt
done; done
