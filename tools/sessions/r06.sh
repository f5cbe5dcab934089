#!/bin/bash
# Round 6: every gpurun session of the round, replayable -- gpurun --timeout 1500 -- 'bash tools/sessions/r06.sh s1'
# (run from the repo root; results land in gpurun_out/, the ones quoted in DESIGN.md / LABNOTES.md were copied to profiles/r06_*).
mkdir -p gpurun_out
case "$1" in
s1)
  # the full GPU suite with every workspace poisoned (tests/conftest.py) and the three entries of tests/helpers.py:call_op
  (timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider 2>&1 | tail -80) > gpurun_out/r06_s1_tests.log
  tail -30 gpurun_out/r06_s1_tests.log
  (timeout 600 python bench.py 2>&1 | tail -3) > gpurun_out/r06_s1_bench.log
  tail -c 3000 gpurun_out/r06_s1_bench.log
  ;;
s2)
  # configs[3], per-op times by route: default routing, everything on the batch tiles / column-lane
  # kernel (mfma_min_batch huge), and the workspace-less entry (no kernel in front, the sparse terms gather)
  E=tools/experiments/small_batch_r05.py
  (timeout 600 python $E --rows 1,2,3,4,5,6,8,12,16 --sets "default;mfma_min_batch=1048576" 2>&1 | grep '^{') > gpurun_out/r06_s2_routes.txt
  (timeout 600 python $E --rows 5,6,8,12,16 --no-ws 2>&1 | grep '^{') > gpurun_out/r06_s2_routes_no_ws.txt
  (timeout 600 python $E --rows 2,3,4 --sets "cols_min_batch=1073741824;cols_min_batch=1,cols_max_batch=4;cols_groups=0" 2>&1 | grep '^{') > gpurun_out/r06_s2_rows_2_4.txt
  cat gpurun_out/r06_s2_routes.txt gpurun_out/r06_s2_routes_no_ws.txt gpurun_out/r06_s2_rows_2_4.txt
  ;;
first_use)
  # the parametrisations of test_wide_batch_routes_behind_options that failed once on a fresh box in round 5,
  # (A) the failing run's own prefix -- the file's first two tests in one fresh process -- $2 times; (B) each of the 24 cases at
  # 16 / 40 / 130 rows as the FIRST GPU work of a fresh process, $3 times plain and $3 times under AMD_SERIALIZE_KERNEL=3
  # HIP_LAUNCH_BLOCKING=1.  Workspaces poisoned throughout (tests/conftest.py).
  A=${2:-40}; B=${3:-4}
  out=gpurun_out/r06_first_use.txt; : > $out
  for i in $(seq 1 $A); do
    r=$(timeout 600 python -m pytest tests/test_gpu_batched.py -q -p no:cacheprovider -k "(test_wide_batches_vs_oracle and module) or test_wide_batch_routes_behind_options" 2>&1 | tail -1)
    echo "A $i $r" >> $out
  done
  for env in plain serialize; do
    for rep in $(seq 1 $B); do
      for batch in 16 40 130; do for shape in 4-1024-132 3-1024-776; do for opt in fp32-instruction split-unfused fp32-unfused sparse-launch-of-its-own; do
        id="tests/test_gpu_batched.py::test_wide_batch_routes_behind_options[$batch-$shape-$opt]"
        if [ $env = serialize ]; then
          r=$(AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 timeout 300 python -m pytest "$id" -q -p no:cacheprovider 2>&1 | tail -1)
        else
          r=$(timeout 300 python -m pytest "$id" -q -p no:cacheprovider 2>&1 | tail -1)
        fi
        echo "B $env $rep $batch-$shape-$opt $r" >> $out
      done; done; done
    done
  done
  echo "A runs: $(grep -c '^A ' $out), with failures: $(grep '^A ' $out | grep -c failed)"
  echo "B runs: $(grep -c '^B ' $out), with failures: $(grep '^B ' $out | grep -c failed)"
  grep failed $out | head -20
  ;;
*)
  echo "unknown session $1"; exit 2 ;;
esac
