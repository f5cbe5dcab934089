#!/bin/bash
# Round 3, GPU session 5: timeline + ablations of the streaming kernel vs the fused kernel, same box
O=gpurun_out/r03_s5; mkdir -p $O
AB=squeezellm_amd/libsqllm_hip_ablation.so
for spec in "4096x4096 1" "4096x4096 3" "4096x11008 2" "11008x4096 1"; do
  set -- $spec
  for opt in "stream=1" "stream=0"; do
    echo "== $opt" >> $O/timeline_w4.txt
    SQLLM_OPTIONS=$opt SQLLM_LIB=$AB timeout 120 python tools/timeline.py --shape $1 --bits 4 --group $2 2>&1 | grep -v amdgpu.ids >> $O/timeline_w4.txt
  done
  SQLLM_OPTIONS=stream=1 SQLLM_LIB=$AB timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --ablate 0,2,4,8 --reps 3 >> $O/sweep_ablate_stream.jsonl 2>>$O/sweep.err
  SQLLM_OPTIONS=stream=1 timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --target-wgs 0,256,512,768 --reps 3 >> $O/sweep_wgs_stream.jsonl 2>>$O/sweep.err
done
cat $O/timeline_w4.txt
