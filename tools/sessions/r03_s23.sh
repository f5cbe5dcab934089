#!/bin/bash
# Round 3, GPU session 23: CSR role after the lane-run rewrite, A/B on one box against the previous commit's library
# (squeezellm_amd/ab/prev.so), and sparse workgroups first / last in the grid at 2-8 rows
O=gpurun_out/r03_s23; mkdir -p $O
for lib in squeezellm_amd/ab/prev.so squeezellm_amd/libsqllm_hip.so; do
 for opts in "cols_min_batch=1000" "cols_min_batch=1000,sparse_last=1"; do
  for shp in 13824x5120 5120x13824; do
   for B in 1 2 4 8; do
    SQLLM_OPTIONS="$opts" SQLLM_LIB=$lib timeout 200 python tools/sweep.py --shapes $shp --batch $B --bits 4 --sparse 0.0045 --topx 10 --reps 3 2>>$O/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', '$opts', d['shape'], 'rows', d['batch'], 'grid', d['grid'], 'wall', d['wall_us'], 'ev', d['us_mean'])" | tee -a $O/ab_csr_role.txt
   done
  done
 done
done
for lib in squeezellm_amd/ab/prev.so squeezellm_amd/libsqllm_hip.so; do
  SQLLM_LIB=$lib timeout 300 python bench.py --no-cpu-baseline 2> $O/bench.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$lib', d['value'], d['roofline']['frac'], {k: v.get('value') for k, v in d['sub_records'].items()}, {b: v['ms_per_decoder_layer'] for b, v in d['sub_records']['13b-w4-s45-batched'].items() if b.startswith('batch')})" | tee -a $O/ab_csr_role.txt
done
