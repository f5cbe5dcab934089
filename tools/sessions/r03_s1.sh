#!/bin/bash
# Round 3, GPU session 1: where does the launch time go?  (dispatch ramp, clocks, phases of the product kernel)
#   gpurun --timeout 900 -- 'bash tools/sessions/r03_s1.sh'
O=gpurun_out/r03_s1; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/device.txt 2>&1
timeout 120 build/exp/dispatch_ramp > $O/dispatch_ramp.txt 2>&1
timeout 200 python bench.py --no-cpu-baseline --no-sub-records 2>$O/bench.err | grep '^{' > $O/bench_7b-w4-s0.json
AB=squeezellm_amd/libsqllm_hip_ablation.so
for spec in "4096x4096 1" "4096x4096 3" "4096x11008 2" "11008x4096 1"; do
  set -- $spec
  SQLLM_LIB=$AB timeout 120 python tools/timeline.py --shape $1 --bits 4 --group $2 >> $O/timeline_w4.txt 2>&1
  SQLLM_LIB=$AB timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --ablate 0,2,4,8 --reps 3 >> $O/sweep_ablate_w4.jsonl 2>>$O/sweep.err
  timeout 200 python tools/sweep.py --shapes $1 --group $2 --bits 4 --target-wgs 0,256,512,768,1024 --reps 3 >> $O/sweep_wgs_w4.jsonl 2>>$O/sweep.err
done
timeout 200 bash tools/clock_sample.sh > $O/clocks.txt 2>&1
ls -la $O
