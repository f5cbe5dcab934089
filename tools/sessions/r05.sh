#!/bin/bash
# Round 5: every gpurun session of the round, replayable -- gpurun --timeout 1500 -- 'bash tools/sessions/r05.sh s15'
# (run from the repo root; results land in gpurun_out/, the ones quoted in DESIGN.md / LABNOTES.md were copied to profiles/r05_*).
# Variant libraries some sessions compare (squeezellm_amd/ab/lib*.so) are built by the recipes in LABNOTES.md, round 5.
# (Round 6: the timing-only mode those sessions switched on as `sparse_transpose=2` left the product library; it is now
# `skip_prepare_small=1` of the measurement library -- run those lines with SQLLM_LIB=squeezellm_amd/libsqllm_hip_ablation.so.)
case "$1" in
s1)
  # round 5, session 1: parity of the folded CSR walk + same-box A/B of the 13B s45 layer by rows (HEAD vs the round-4 build)
  mkdir -p gpurun_out
  (timeout 900 python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -40) > gpurun_out/r05_s1_tests.log
  E=tools/experiments/small_batch_r05.py
  (timeout 300 python $E --rows 1,2,3,4,5,6,8,12,16 2>&1 | grep '^{') > gpurun_out/r05_s1_new.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --rows 1,2,3,4,5,6,8,12,16 2>&1 | grep '^{') > gpurun_out/r05_s1_base.txt
  (timeout 300 python $E --rows 2,4 --sets "csr_fold=0;mfma_min_batch=2;cols_min_batch=1073741824;cols_max_batch=4" 2>&1 | grep '^{') > gpurun_out/r05_s1_sweep_small.txt
  (timeout 300 python $E --rows 5,8,16 --sets "small_wgs_per_cu=2;small_wgs_per_cu=4;small_wgs_per_cu=6;mfma_min_batch=1048576" 2>&1 | grep '^{') > gpurun_out/r05_s1_sweep_split.txt
  (timeout 300 python $E --dense-only --rows 1,2,4,8,16 2>&1 | grep '^{') > gpurun_out/r05_s1_dense_only.txt
  (timeout 300 python $E --bits 3 --rows 2,4,8,16 2>&1 | grep '^{') > gpurun_out/r05_s1_new_w3.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --bits 3 --rows 2,4,8,16 2>&1 | grep '^{') > gpurun_out/r05_s1_base_w3.txt
  tail -5 gpurun_out/r05_s1_tests.log; cat gpurun_out/r05_s1_new.txt gpurun_out/r05_s1_base.txt
  ;;
s2)
  # round 5, session 2: dense-only A/B of the small split launch: round-4 build vs 80-register build vs the same code at 128 registers
  mkdir -p gpurun_out
  E=tools/experiments/small_batch_r05.py
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --dense-only --rows 8,16 2>&1 | grep '^{') > gpurun_out/r05_s2_dense.txt
  (timeout 300 python $E --dense-only --rows 8,16 --sets "small_wgs_per_cu=2;small_wgs_per_cu=3" 2>&1 | grep '^{') >> gpurun_out/r05_s2_dense.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libcap4.so timeout 300 python $E --dense-only --rows 8,16 --sets "small_wgs_per_cu=2;small_wgs_per_cu=3" 2>&1 | grep '^{') >> gpurun_out/r05_s2_dense.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libcap4.so timeout 300 python $E --rows 8,16 --sets "small_wgs_per_cu=2" 2>&1 | grep '^{') >> gpurun_out/r05_s2_dense.txt
  cat gpurun_out/r05_s2_dense.txt
  ;;
s3)
  # round 5, session 3: the walker wave (one wave of every dense workgroup walks the tile's CSR share, groups by LDS ticket)
  mkdir -p gpurun_out
  (timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py tests/test_gpu_parity.py -m gpu -q --maxfail=30 2>&1 | tail -15) > gpurun_out/r05_s3_tests.log
  E=tools/experiments/small_batch_r05.py
  (timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') > gpurun_out/r05_s3.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s3.txt
  (timeout 300 python $E --dense-only --rows 8,16 2>&1 | grep '^{') >> gpurun_out/r05_s3.txt
  (timeout 300 python $E --bits 3 --rows 8,16 --sets "default;mfma_min_batch=5" 2>&1 | grep '^{') >> gpurun_out/r05_s3.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --bits 3 --rows 8,16 --sets "default;mfma_min_batch=5" 2>&1 | grep '^{') >> gpurun_out/r05_s3.txt
  tail -3 gpurun_out/r05_s3_tests.log; cat gpurun_out/r05_s3.txt
  ;;
s4)
  # round 5, session 4: the staged all-waves walk after the dense loop (columns and values preloaded before it)
  mkdir -p gpurun_out
  (timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py tests/test_gpu_parity.py -m gpu -q --maxfail=30 2>&1 | tail -15) > gpurun_out/r05_s4_tests.log
  E=tools/experiments/small_batch_r05.py
  (timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') > gpurun_out/r05_s4.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s4.txt
  (timeout 300 python $E --dense-only --rows 8,16 2>&1 | grep '^{') >> gpurun_out/r05_s4.txt
  (timeout 300 python $E --bits 3 --rows 8,16 --sets "default;mfma_min_batch=5" 2>&1 | grep '^{') >> gpurun_out/r05_s4.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --bits 3 --rows 8,16 --sets "default;mfma_min_batch=5" 2>&1 | grep '^{') >> gpurun_out/r05_s4.txt
  tail -3 gpurun_out/r05_s4_tests.log; cat gpurun_out/r05_s4.txt
  ;;
s5)
  # round 5, session 5: folded CSR walk reading a TRANSPOSED copy of vec from the caller's workspace
  mkdir -p gpurun_out
  (timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py tests/test_gpu_parity.py tests/test_gpu_module.py -m gpu -q --maxfail=30 2>&1 | tail -15) > gpurun_out/r05_s5_tests.log
  E=tools/experiments/small_batch_r05.py
  (timeout 300 python $E --rows 2,4,5,8,12,16 2>&1 | grep '^{') > gpurun_out/r05_s5.txt
  (timeout 300 python $E --rows 5,8,16 --no-ws 2>&1 | grep '^{') >> gpurun_out/r05_s5.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --rows 2,4,5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s5.txt
  (timeout 300 python $E --rows 3,4 --sets "mfma_min_batch=3" 2>&1 | grep '^{') >> gpurun_out/r05_s5.txt
  (timeout 300 python $E --bits 3 --rows 5,8,16 --sets "default;mfma_min_batch=5" 2>&1 | grep '^{') >> gpurun_out/r05_s5.txt
  tail -3 gpurun_out/r05_s5_tests.log; cat gpurun_out/r05_s5.txt
  ;;
s6)
  # round 5, session 6: what the transposition launch costs (timing-only knob: the walk reads a stale transposed copy)
  mkdir -p gpurun_out
  E=tools/experiments/small_batch_r05.py
  (timeout 300 python $E --rows 5,8,16 --sets "default;skip_prepare_small=1" 2>&1 | grep '^{') > gpurun_out/r05_s6.txt
  (timeout 300 python $E --rows 8,16 --dense-only 2>&1 | grep '^{') >> gpurun_out/r05_s6.txt
  cat gpurun_out/r05_s6.txt
  ;;
s7)
  # round 5, session 7: the walk with staged row ids (no data-dependent loop per step)
  mkdir -p gpurun_out
  (timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py tests/test_gpu_parity.py tests/test_gpu_module.py -m gpu -q --maxfail=30 2>&1 | tail -15) > gpurun_out/r05_s7_tests.log
  E=tools/experiments/small_batch_r05.py
  (timeout 300 python $E --rows 5,8,12,16 --sets "default;skip_prepare_small=1" 2>&1 | grep '^{') > gpurun_out/r05_s7.txt
  (timeout 300 python $E --rows 8,16 --no-ws 2>&1 | grep '^{') >> gpurun_out/r05_s7.txt
  (timeout 300 python $E --bits 3 --rows 5,8,16 --sets "default;mfma_min_batch=5" 2>&1 | grep '^{') >> gpurun_out/r05_s7.txt
  tail -3 gpurun_out/r05_s7_tests.log; cat gpurun_out/r05_s7.txt
  ;;
s8)
  # round 5, session 8: timeline of the fused small launch (measurement library)
  mkdir -p gpurun_out
  export SQLLM_LIB=$PWD/squeezellm_amd/libsqllm_hip_ablation.so
  T=tools/experiments/small_split_timeline.py
  (timeout 200 python $T --rows 16; timeout 200 python $T --rows 8; timeout 200 python $T --rows 16 --no-ws; timeout 200 python $T --rows 16 --dense-only) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_s8_timeline.txt
  cat gpurun_out/r05_s8_timeline.txt
  ;;
s9)
  # round 5, session 9: one round of workgroups including the top-X slabs
  mkdir -p gpurun_out
  E=tools/experiments/small_batch_r05.py
  (timeout 300 python $E --rows 5,8,12,16 --sets "default;skip_prepare_small=1" 2>&1 | grep '^{') > gpurun_out/r05_s9.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s9.txt
  (timeout 300 python $E --rows 8,16 --dense-only 2>&1 | grep '^{') >> gpurun_out/r05_s9.txt
  (timeout 300 python $E --bits 3 --rows 8,16 --sets "default;mfma_min_batch=5" 2>&1 | grep '^{') >> gpurun_out/r05_s9.txt
  cat gpurun_out/r05_s9.txt
  ;;
s10)
  # round 5, session 10: top-X slabs in passes of 8 batch rows
  mkdir -p gpurun_out
  (timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py tests/test_gpu_parity.py -m gpu -q --maxfail=30 2>&1 | tail -5) > gpurun_out/r05_s10_tests.log
  E=tools/experiments/small_batch_r05.py
  (timeout 300 python $E --rows 5,8,12,16 --sets "default;skip_prepare_small=1" 2>&1 | grep '^{') > gpurun_out/r05_s10.txt
  (timeout 300 python $E --rows 8,16 --no-ws 2>&1 | grep '^{') >> gpurun_out/r05_s10.txt
  (timeout 300 python $E --bits 3 --rows 8,16 --sets "default;mfma_min_batch=5" 2>&1 | grep '^{') >> gpurun_out/r05_s10.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/libsqllm_hip_ablation.so timeout 200 python tools/experiments/small_split_timeline.py --rows 16 2>&1 | grep -v amdgpu.ids) > gpurun_out/r05_s10_timeline.txt
  tail -2 gpurun_out/r05_s10_tests.log; cat gpurun_out/r05_s10.txt gpurun_out/r05_s10_timeline.txt
  ;;
s11)
  # round 5, session 11: top-X slabs out of the transposed vec, 8-16 workgroups per op, dense ranges planned beside them
  mkdir -p gpurun_out
  (timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py tests/test_gpu_parity.py -m gpu -q --maxfail=30 2>&1 | tail -5) > gpurun_out/r05_s11_tests.log
  E=tools/experiments/small_batch_r05.py
  (timeout 300 python $E --rows 5,8,12,16 --sets "default;skip_prepare_small=1" 2>&1 | grep '^{') > gpurun_out/r05_s11.txt
  (timeout 300 python $E --rows 8,16 --no-ws 2>&1 | grep '^{') >> gpurun_out/r05_s11.txt
  (timeout 300 python $E --bits 3 --rows 8,16 --sets "default;mfma_min_batch=5" 2>&1 | grep '^{') >> gpurun_out/r05_s11.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/libsqllm_hip_ablation.so timeout 200 python tools/experiments/small_split_timeline.py --rows 16 2>&1 | grep -v amdgpu.ids) > gpurun_out/r05_s11_timeline.txt
  tail -2 gpurun_out/r05_s11_tests.log; cat gpurun_out/r05_s11.txt gpurun_out/r05_s11_timeline.txt
  ;;
s12)
  mkdir -p gpurun_out
  E=tools/experiments/small_batch_r05.py
  (timeout 300 python $E --rows 5,8,16 --sets "default;skip_prepare_small=1" 2>&1 | grep '^{') > gpurun_out/r05_s12.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/libsqllm_hip_ablation.so timeout 200 python tools/experiments/small_split_timeline.py --rows 16 2>&1 | grep -v amdgpu.ids) > gpurun_out/r05_s12_timeline.txt
  cat gpurun_out/r05_s12.txt gpurun_out/r05_s12_timeline.txt
  ;;
s13)
  mkdir -p gpurun_out
  (SQLLM_LIB=$PWD/squeezellm_amd/libsqllm_hip_ablation.so timeout 200 python tools/experiments/small_split_timeline.py --rows 16 2>&1 | grep -v amdgpu.ids; SQLLM_LIB=$PWD/squeezellm_amd/libsqllm_hip_ablation.so timeout 200 python tools/experiments/small_split_timeline.py --rows 16 --dense-only 2>&1 | grep -v amdgpu.ids) > gpurun_out/r05_s13_timeline.txt
  cat gpurun_out/r05_s13_timeline.txt
  ;;
s14)
  # round 5, session 14: full GPU suite + A/B of the 13B s45 layer (walk in one round of 40 gathers; workspace-less names on stream-ordered scratch)
  mkdir -p gpurun_out
  (timeout 900 python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -8) > gpurun_out/r05_s14_tests.log
  E=tools/experiments/small_batch_r05.py
  (timeout 300 python $E --rows 1,2,4,5,6,8,12,16 2>&1 | grep '^{') > gpurun_out/r05_s14.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --rows 1,2,4,5,6,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s14.txt
  (timeout 300 python $E --rows 8,16 --no-ws 2>&1 | grep '^{') >> gpurun_out/r05_s14.txt
  (timeout 300 python $E --bits 3 --rows 8,9,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s14.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --bits 3 --rows 8,9,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s14.txt
  tail -3 gpurun_out/r05_s14_tests.log; cat gpurun_out/r05_s14.txt
  ;;
s15)
  mkdir -p gpurun_out
  (timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py -m gpu -q --maxfail=30 2>&1 | tail -4) > gpurun_out/r05_s15_tests.log
  E=tools/experiments/small_batch_r05.py
  (timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') > gpurun_out/r05_s15.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s15.txt
  tail -2 gpurun_out/r05_s15_tests.log; cat gpurun_out/r05_s15.txt
  ;;
s16)
  # round 5, session 16: full GPU suite incl. the non-finite cases; shim through the workspace
  mkdir -p gpurun_out
  (timeout 1200 python -m pytest tests -m gpu -q --maxfail=40 2>&1 | tail -60) > gpurun_out/r05_s16_tests.log
  tail -40 gpurun_out/r05_s16_tests.log
  ;;
s17)
  # round 5, session 17: full GPU suite; batch-1 A/B of kernel-argument preload; loads-only calibration on this box; default bench line
  mkdir -p gpurun_out
  (timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | tail -12) > gpurun_out/r05_s17_tests.log
  (bash tools/ab_libs.sh "head preload" "7b-w4-s0 7b-w4-s45" 2 2>&1 | grep -v amdgpu.ids) > gpurun_out/r05_s17_preload_ab.txt
  (/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/stream_patterns.hip -o /tmp/sp 2>&1 | tail -3; timeout 300 /tmp/sp) > gpurun_out/r05_s17_stream_patterns.txt 2>&1
  (timeout 600 python bench.py 2>/dev/null | grep '^{') > gpurun_out/r05_s17_bench.json
  tail -4 gpurun_out/r05_s17_tests.log; cat gpurun_out/r05_s17_preload_ab.txt; grep -c . gpurun_out/r05_s17_stream_patterns.txt; python -c "
  import json; d=json.load(open('gpurun_out/r05_s17_bench.json')); print(d['value'], d['roofline']['frac'], d['roofline'].get('frac_rocprof'), [ (k, v) for k,v in d.get('sub_records',{}).items()][:3])" 2>&1 | cut -c1-1500
  ;;
s18)
  # round 5, session 18: groups of the fused small launch by LDS ticket (variant library) against round-robin
  mkdir -p gpurun_out
  E=tools/experiments/small_batch_r05.py
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libtickets.so timeout 300 python -m pytest tests/test_gpu_decoder_layer.py tests/test_gpu_batched.py -m gpu -q 2>&1 | tail -3) > gpurun_out/r05_s18.txt
  for rep in 1 2; do for v in head tickets; do
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/lib$v.so timeout 300 python $E --rows 8,16 2>&1 | grep '^{') >> gpurun_out/r05_s18.txt
  done; done
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libtickets.so timeout 300 python $E --rows 8,16 --dense-only 2>&1 | grep '^{') >> gpurun_out/r05_s18.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libhead.so timeout 300 python $E --rows 8,16 --dense-only 2>&1 | grep '^{') >> gpurun_out/r05_s18.txt
  cat gpurun_out/r05_s18.txt
  ;;
s19)
  # round 5, session 19: the 16-byte transposition kernel; batch-1 A/B against the round-4 build (SQLLM_LIB)
  mkdir -p gpurun_out
  (timeout 600 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py tests/test_gpu_nonfinite.py tests/test_gpu_workspace.py -m gpu -q 2>&1 | tail -3) > gpurun_out/r05_s19.txt
  E=tools/experiments/small_batch_r05.py
  (timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s19.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s19.txt
  for rep in 1 2; do for v in r04 head; do for c in 7b-w4-s0 7b-w4-s45 7b-w3-s45; do
  L=$PWD/squeezellm_amd/ab/lib$v.so; [ $v = head ] && L=$PWD/squeezellm_amd/libsqllm_hip.so
  (echo -n "$v "; SQLLM_LIB=$L timeout 200 python bench.py --config $c --no-cpu-baseline --no-sub-records 2>/dev/null | grep "^{") >> gpurun_out/r05_s19_batch1_ab.txt; done; done; done
  cat gpurun_out/r05_s19.txt; wc -l gpurun_out/r05_s19_batch1_ab.txt
  ;;
s20)
  # round 5, session 20: every other CU-load of workgroups walks its CSR share BEFORE its dense loop; batch-1 A/B against the round-4 build
  mkdir -p gpurun_out
  (timeout 600 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py tests/test_gpu_nonfinite.py tests/test_gpu_workspace.py -m gpu -q 2>&1 | tail -3) > gpurun_out/r05_s20.txt
  E=tools/experiments/small_batch_r05.py
  (timeout 300 python $E --rows 5,8,12,16 --sets "default;small_walk_first=0" 2>&1 | grep '^{') >> gpurun_out/r05_s20.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s20.txt
  (timeout 300 python $E --bits 3 --rows 9,16 --sets "default;small_walk_first=0" 2>&1 | grep '^{') >> gpurun_out/r05_s20.txt
  for rep in 1 2; do for v in r04 head; do for c in 7b-w4-s0 7b-w4-s45 7b-w3-s45; do
  L=$PWD/squeezellm_amd/ab/lib$v.so; [ $v = head ] && L=$PWD/squeezellm_amd/libsqllm_hip.so
  (echo -n "$v "; SQLLM_LIB=$L timeout 200 python bench.py --config $c --no-cpu-baseline --no-sub-records 2>/dev/null | grep "^{") >> gpurun_out/r05_s20_batch1_ab.txt; done; done; done
  cat gpurun_out/r05_s20.txt; wc -l gpurun_out/r05_s20_batch1_ab.txt
  ;;
s23)
  # round 5, session 23: vec already split into bf16 planes for the dense term of the fused small launch (written by the kernel in front, with xT)
  mkdir -p gpurun_out
  (timeout 600 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py tests/test_gpu_nonfinite.py tests/test_gpu_workspace.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -3) > gpurun_out/r05_s23.txt
  E=tools/experiments/small_batch_r05.py
  (timeout 300 python $E --rows 5,8,12,16 --sets "default;small_planes=0" 2>&1 | grep '^{') >> gpurun_out/r05_s23.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s23.txt
  (timeout 300 python $E --bits 3 --rows 9,16 --sets "default;small_planes=0" 2>&1 | grep '^{') >> gpurun_out/r05_s23.txt
  cat gpurun_out/r05_s23.txt
  ;;
s24)
  # round 5, session 24: five partial products for fp16-born vec on the fused small launch (flag from the kernel in front)
  mkdir -p gpurun_out
  (timeout 600 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py tests/test_gpu_nonfinite.py tests/test_gpu_workspace.py -m gpu -q 2>&1 | tail -3) > gpurun_out/r05_s24.txt
  E=tools/experiments/small_batch_r05.py
  (timeout 300 python $E --rows 5,8,12,16 --sets "default;small_planes=0" 2>&1 | grep '^{') >> gpurun_out/r05_s24.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libplanes6.so timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s24.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr04.so timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s24.txt
  (timeout 300 python $E --bits 3 --rows 9,16 2>&1 | grep '^{') >> gpurun_out/r05_s24.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libplanes6.so timeout 300 python $E --bits 3 --rows 9,16 2>&1 | grep '^{') >> gpurun_out/r05_s24.txt
  cat gpurun_out/r05_s24.txt
  ;;
s25)
  # round 5, session 25: the top-X role in passes of several batch rows in every batched kernel (libplanes6.so = the tree before)
  mkdir -p gpurun_out
  (timeout 600 python -m pytest tests/test_gpu_batched.py tests/test_gpu_decoder_layer.py tests/test_gpu_parity.py tests/test_gpu_property.py -m gpu -q 2>&1 | tail -3) > gpurun_out/r05_s25.txt
  E=tools/experiments/small_batch_r05.py
  for rep in 1 2; do
  (timeout 300 python $E --rows 2,3,4,17,32,64 2>&1 | grep '^{') >> gpurun_out/r05_s25.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libplanes6.so timeout 300 python $E --rows 2,3,4,17,32,64 2>&1 | grep '^{') >> gpurun_out/r05_s25.txt
  done
  cat gpurun_out/r05_s25.txt
  ;;
s26)
  # round 5, session 26: packed words of TWO groups ahead in the fused small launch's dense loop (libpf2.so: a variant build, described in LABNOTES.md round 5 item 4c; not kept in the tree)
  mkdir -p gpurun_out
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libpf2.so timeout 600 python -m pytest tests/test_gpu_batched.py tests/test_gpu_nonfinite.py -m gpu -q 2>&1 | tail -3) > gpurun_out/r05_s26.txt
  E=tools/experiments/small_batch_r05.py
  for rep in 1 2; do
  (timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s26.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libpf2.so timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s26.txt
  done
  (timeout 300 python $E --dense-only --rows 8,16 2>&1 | grep '^{') >> gpurun_out/r05_s26.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libpf2.so timeout 300 python $E --dense-only --rows 8,16 2>&1 | grep '^{') >> gpurun_out/r05_s26.txt
  cat gpurun_out/r05_s26.txt
  ;;
s27)
  # round 5, session 27: where the waves of the 16-row launches wait (s_waitcnt / barrier vs issue stalls), s45 and dense-only
  R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
  for v in s45 s0; do
    extra=""; [ $v = s0 ] && extra="--dense-only"
    rm -rf /tmp/prof_w_$v
    timeout 400 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/prof_w_$v -o x -- python $R/tools/experiments/small_batch_r05.py --rows 16 --no-breakdown $extra > /tmp/prof_w_$v.log 2>&1
    python $R/tools/rocprof_summary.py "$(find /tmp/prof_w_$v -name '*.db' | head -1)" --by-grid --match sqllm --top 12 > $R/gpurun_out/r05_s27_waits_$v.txt
  done
  cd $R; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  tail -40 gpurun_out/r05_s27_waits_s45.txt
  ;;
s28)
  # round 5, session 28: the next column's lookups issued between the six matrix instructions of a column (libpipe.so = -DSQLLM_SMALL_PIPE)
  mkdir -p gpurun_out
  L=${2:-pipe}
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/lib$L.so timeout 600 python -m pytest tests/test_gpu_batched.py tests/test_gpu_nonfinite.py -m gpu -q 2>&1 | tail -3) > gpurun_out/r05_s28_$L.txt
  E=tools/experiments/small_batch_r05.py
  for rep in 1 2; do
  (timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s28_$L.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/lib$L.so timeout 300 python $E --rows 5,8,12,16 2>&1 | grep '^{') >> gpurun_out/r05_s28_$L.txt
  done
  (timeout 300 python $E --bits 3 --rows 9,16 2>&1 | grep '^{') >> gpurun_out/r05_s28_$L.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/lib$L.so timeout 300 python $E --bits 3 --rows 9,16 2>&1 | grep '^{') >> gpurun_out/r05_s28_$L.txt
  cat gpurun_out/r05_s28_$L.txt
  ;;
s29)
  # round 5, session 29: what the 16-row launch costs without its matrix instructions (abl1), without its LDS lookups (abl2), with the
  # weight loads (abl3) / the vec plane loads (abl4) folded onto eight groups per wave -- variant libraries with WRONG results, timing only
  mkdir -p gpurun_out
  E=tools/experiments/small_batch_r05.py
  for L in ${2:-HEAD abl1 abl2 abl3 abl4 HEAD}; do
    X=""; [ $L != HEAD ] && X=$PWD/squeezellm_amd/ab/lib$L.so
    (SQLLM_LIB=$X timeout 300 python $E --rows 16 2>&1 | grep '^{') >> gpurun_out/r05_s29.txt
    (SQLLM_LIB=$X timeout 300 python $E --rows 16 --dense-only 2>&1 | grep '^{') >> gpurun_out/r05_s29.txt
  done
  cat gpurun_out/r05_s29.txt
  ;;
*) echo "usage: $0 s1..s29"; exit 2;;
esac
