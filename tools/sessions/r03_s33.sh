#!/bin/bash
# Round 3, GPU session 33: timeline of the wide-batch sparse launch's chunk workgroups (13B gate/up shape, 9-64 rows)
O=gpurun_out/r03_s33; mkdir -p $O
AB=squeezellm_amd/libsqllm_hip_ablation.so
for B in 9 16 32 64; do
  SQLLM_LIB=$AB timeout 200 python tools/timeline.py --shape 5120x13824 --bits 4 --batch $B --sparse 0.0045 --topx 10 --copies 8 2>>$O/err.txt | tee -a $O/timeline_wide.txt
done
tail -5 $O/err.txt
