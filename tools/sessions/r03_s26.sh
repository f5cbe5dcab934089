#!/bin/bash
# Round 3, GPU session 26: timeline of the CSR chunk workgroups inside the 7B s45 launches
O=gpurun_out/r03_s26; mkdir -p $O
AB=squeezellm_amd/libsqllm_hip_ablation.so
for spec in "4096x4096 1" "4096x4096 3" "4096x11008 2" "11008x4096 1"; do
  set -- $spec
  SQLLM_LIB=$AB timeout 200 python tools/timeline.py --shape $1 --bits 4 --group $2 --sparse 0.0045 --topx 10 2>>$O/err.txt | tee -a $O/timeline_csr.txt
done
tail -3 $O/err.txt
