#!/bin/bash
# Round 6: every gpurun session of the round, replayable -- gpurun --timeout 1500 -- 'bash tools/sessions/r06.sh s1'
# (run from the repo root; results land in gpurun_out/, the ones quoted in DESIGN.md / LABNOTES.md were copied to profiles/r06_*).
# Variant libraries the A/B sessions load (squeezellm_amd/ab/lib<name>.so, git-ignored, not kept): libr05.so = the round-5 tree (git worktree at 87e8bf2,
# python -m squeezellm_amd.build); libhead.so / libfinal.so / libprev.so = the tree of the moment before the change under test; libv<N>.so = a copy of csrc/
# with the one-line patch the session's comment and LABNOTES.md (round 6, sections 4-5) describe, compiled with the flags of squeezellm_amd/build.py.
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
s3)
  # the 3- / 5- / 6-row batch tiles (round 6): parity, then same-box A/B of the 13B s45 layer by rows against the round-5 build
  # (squeezellm_amd/ab/libr05.so = the tree at 87e8bf2) with the split kernel taking over from 5 (default until now), 7 and 9 rows
  E=tools/experiments/small_batch_r05.py
  (timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "test_batch_tiles_every_row_count or test_three_batched or test_small_shapes or test_llama13b" 2>&1 | tail -5) > gpurun_out/r06_s3_tests.log
  tail -3 gpurun_out/r06_s3_tests.log
  for rep in 1 2; do
  (timeout 600 python $E --rows 2,3,4,5,6,7,8 --sets "default;mfma_min_batch=7;mfma_min_batch=9" 2>&1 | grep '^{') >> gpurun_out/r06_s3_w4.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr05.so timeout 600 python $E --rows 2,3,4,5,6,7,8 2>&1 | grep '^{') >> gpurun_out/r06_s3_w4.txt
  done
  (timeout 600 python $E --bits 3 --rows 2,3,4,5,6,8,9 2>&1 | grep '^{') > gpurun_out/r06_s3_w3.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr05.so timeout 600 python $E --bits 3 --rows 2,3,4,5,6,8,9 2>&1 | grep '^{') >> gpurun_out/r06_s3_w3.txt
  python - <<'PY'
import json
for f in ("gpurun_out/r06_s3_w4.txt", "gpurun_out/r06_s3_w3.txt"):
    print(f)
    for l in open(f):
        d = json.loads(l)
        print(d["lib"], d["rows"], d["set"], d["layer_us"], d.get("qkv"), d.get("o"), d.get("gate_up"), d.get("down"))
PY
  ;;
s4)
  # new default routing (4-bit: tiles up to 6 rows, small single ops up to 8): full suite with workspaces AND LDS poisoned, same-box
  # A/B against the round-5 build and against the explicit thresholds, 3-bit tiles against the column-lane kernel, bench line
  E=tools/experiments/small_batch_r05.py
  (timeout 1800 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/r06_s4_tests.log
  tail -4 gpurun_out/r06_s4_tests.log
  for rep in 1 2; do
  (timeout 600 python $E --rows 1,2,3,4,5,6,7,8,12,16 --sets "default;mfma_min_batch=7;mfma_min_batch=5" 2>&1 | grep '^{') >> gpurun_out/r06_s4_w4.txt
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libr05.so timeout 600 python $E --rows 1,2,3,4,5,6,7,8,12,16 2>&1 | grep '^{') >> gpurun_out/r06_s4_w4.txt
  done
  (timeout 600 python $E --bits 3 --rows 2,3,4,5,6,8 --sets "default;cols_min_batch=1073741824" 2>&1 | grep '^{') > gpurun_out/r06_s4_w3.txt
  python - <<'PY'
import json
for f in ("gpurun_out/r06_s4_w4.txt", "gpurun_out/r06_s4_w3.txt"):
    print(f)
    for l in open(f):
        d = json.loads(l)
        print(d["lib"], d["rows"], d["set"], d["layer_us"], d.get("qkv"), d.get("o"), d.get("gate_up"), d.get("down"))
PY
  (timeout 900 python bench.py 2>/dev/null | grep '^{') > gpurun_out/r06_s4_bench.json
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_s4_bench.json"))
print(d["value"], d["roofline"]["frac"], {k: v.get("tokens_per_s") for k, v in d["drop_in"].items() if isinstance(v, dict) and "tokens_per_s" in v})
print({k: v["ms_per_decoder_layer"] for k, v in d["sub_records"]["13b-w4-s45-batched"].items() if isinstance(v, dict) and "ms_per_decoder_layer" in v})
PY
  ;;
micro)
  # batch-1 micro-variants of the decode step, same-box A/B through variant builds (squeezellm_amd/ab/lib<v>.so, built from patched copies of
  # csrc/: LABNOTES.md round 6, 4): head = HEAD; v1 = two independent FMA chains per column pair; v2 = s_setprio 1 for the younger half of a
  # workgroup's waves; v4 = s_setprio 1 for every dense wave (above the sparse roles' waves).  Plus the small-op geometry sweep (item 3c).
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libv1.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "llama7b or small_shapes" 2>&1 | tail -2) > gpurun_out/r06_micro_parity.txt
  (bash tools/ab_libs.sh "head v1 v2 v4" "7b-w4-s0 7b-w3-s45 7b-w4-s45" 3 2>&1 | grep -v amdgpu.ids) > gpurun_out/r06_micro_ab.txt
  cat gpurun_out/r06_micro_parity.txt gpurun_out/r06_micro_ab.txt
  (timeout 900 python tools/experiments/small_op_geometry.py --bits 4 2>&1 | grep '^{') > gpurun_out/r06_small_op_geometry.txt
  (timeout 900 python tools/experiments/small_op_geometry.py --bits 3 2>&1 | grep '^{') >> gpurun_out/r06_small_op_geometry.txt
  cat gpurun_out/r06_small_op_geometry.txt
  ;;
prio)
  # the adopted rule (dense waves at s_setprio 1 iff the batch-1 launch has sparse roles and fits the resident slots): libv5.so = this tree, libhead.so = the tree before it
  (bash tools/ab_libs.sh "head v5" "7b-w3-s45 7b-w4-s45 13b-w4-s45 65b-w3-s45 7b-w4-s0" 3 2>&1 | grep -v amdgpu.ids) > gpurun_out/r06_dense_priority_ab.txt
  cat gpurun_out/r06_dense_priority_ab.txt
  ;;
prio2)
  # does the priority help the multi-round launches too once the sparse workgroups come LAST in the grid (option sparse_last)?  libv6.so = priority for every
  # batch-1 launch with sparse roles; libv5.so = the adopted rule (only launches that fit the resident slots)
  cp squeezellm_amd/libsqllm_hip.so /tmp/lib_orig.so
  for rep in 1 2 3; do for v in "v5 sparse_last=0" "v5 sparse_last=1" "v6 sparse_last=0" "v6 sparse_last=1"; do set -- $v
    cp squeezellm_amd/ab/lib$1.so squeezellm_amd/libsqllm_hip.so
    for c in 7b-w3-s45 7b-w4-s45 13b-w4-s45; do SQLLM_OPTIONS=$2 timeout 200 python bench.py --config $c --no-cpu-baseline --no-sub-records 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', '$2', d['config']['config_name'], d['value'], {k: v['us_mean'] for k, v in d['per_layer_us'].items()})
"; done; done; done > gpurun_out/r06_dense_priority_sparse_last.txt
  cp /tmp/lib_orig.so squeezellm_amd/libsqllm_hip.so
  cat gpurun_out/r06_dense_priority_sparse_last.txt
  ;;
prio3)
  # the adopted rule (libv7.so = this tree): sparse workgroups first + dense priority when the batch-1 launch fits the resident slots, sparse workgroups last otherwise;
  # against libv5.so (priority rule only) and libhead.so (neither); then the full GPU suite on the new ordering, and the same switch forced on the batch tiles
  (bash tools/ab_libs.sh "head v5 v7" "7b-w3-s45 7b-w4-s45 13b-w4-s45 65b-w3-s45" 3 2>&1 | grep -v amdgpu.ids) > gpurun_out/r06_sparse_order_ab.txt
  cat gpurun_out/r06_sparse_order_ab.txt
  (timeout 1800 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider 2>&1 | tail -5) > gpurun_out/r06_prio3_tests.log
  tail -3 gpurun_out/r06_prio3_tests.log
  E=tools/experiments/small_batch_r05.py
  (timeout 600 python $E --rows 2,3,4,5,6 --sets "default;sparse_last=1;sparse_last=2;default" 2>&1 | grep '^{') > gpurun_out/r06_sparse_order_tiles.txt
  python - <<'PY'
import json
for l in open("gpurun_out/r06_sparse_order_tiles.txt"):
    d = json.loads(l)
    print(d["rows"], d["set"], d["layer_us"], d.get("qkv"), d.get("o"), d.get("gate_up"), d.get("down"))
PY
  ;;
final)
  # the round's final tree: full GPU suite, smoke, the default bench line, and the adopted 3-bit priority rule once more (libhead.so = before it)
  (timeout 1800 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider 2>&1 | tail -5) > gpurun_out/r06_final_tests.log
  tail -3 gpurun_out/r06_final_tests.log
  (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/r06_final_smoke.log; cat gpurun_out/r06_final_smoke.log
  cp squeezellm_amd/libsqllm_hip.so squeezellm_amd/ab/libfinal.so
  (bash tools/ab_libs.sh "head final" "7b-w3-s45 7b-w4-s45 7b-w3-s0" 3 2>&1 | grep -v amdgpu.ids) > gpurun_out/r06_final_ab.txt; cat gpurun_out/r06_final_ab.txt
  (timeout 900 python bench.py 2>/dev/null | grep '^{') > gpurun_out/r06_final_bench.json
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_final_bench.json"))
print(d["value"], d["roofline"]["frac"], d["roofline"].get("frac_rocprof"), {k: v.get("tokens_per_s") for k, v in d["drop_in"].items() if isinstance(v, dict) and "tokens_per_s" in v})
print({k: v.get("value") for k, v in d["sub_records"].items() if isinstance(v, dict)})
print({k: v["ms_per_decoder_layer"] for k, v in d["sub_records"]["13b-w4-s45-batched"].items() if isinstance(v, dict) and "ms_per_decoder_layer" in v})
PY
  ;;
kt_w3)
  # the 3-bit kernel trace of the standard set again, on the final tree (the set of tools/collect_profiles.sh r06 predates the dense-priority rule)
  R=$PWD; cd /tmp && export TMPDIR=/tmp
  (cd $R && timeout 400 python bench.py --config 7b-w3-s45 --no-cpu-baseline --no-sub-records 2>/dev/null | grep '^{' > gpurun_out/r06_bench_samebox_7b-w3-s45.json)
  rm -rf /tmp/prof_kt_w3; timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_w3 -o x -- python $R/bench.py --steps 20 --no-cpu-baseline --no-sub-records --config 7b-w3-s45 > /tmp/prof_kt_w3.log 2>&1
  grep '^{' /tmp/prof_kt_w3.log > $R/gpurun_out/r06_kt_w3.bench.json
  python $R/tools/rocprof_summary.py "$(find /tmp/prof_kt_w3 -name '*.db' | head -1)" --by-grid --match sqllm --top 12 > $R/gpurun_out/r06_kt_w3.summary.txt
  cat $R/gpurun_out/r06_kt_w3.summary.txt; python -c "import json; d=json.load(open('$R/gpurun_out/r06_bench_samebox_7b-w3-s45.json')); print(d['value'], d['roofline']['frac'])"
  ;;
prio4)
  # the opposite priority for the launches that do NOT fit the resident slots: libv8.so = libfinal.so + the CSR / top-X workgroups' waves at s_setprio 1 there
  # (they hold slots the next dense workgroups wait for: let them finish sooner)
  (bash tools/ab_libs.sh "final v8" "7b-w3-s45 7b-w4-s45 13b-w4-s45 65b-w3-s45" 3 2>&1 | grep -v amdgpu.ids) > gpurun_out/r06_sparse_priority_ab.txt
  cat gpurun_out/r06_sparse_priority_ab.txt
  ;;
prio5)
  # libv9.so = libfinal.so + rule 2 of set_role_priority (sparse waves at priority 1: 4-bit launches, 3-bit multi-round launches with >= 2 x CUs sparse workgroups)
  (bash tools/ab_libs.sh "final v9" "7b-w3-s45 7b-w4-s45 13b-w4-s45 65b-w3-s45" 3 2>&1 | grep -v amdgpu.ids) > gpurun_out/r06_role_priority_ab.txt
  cat gpurun_out/r06_role_priority_ab.txt | cut -c1-60
  ;;
prio6)
  # the role priority on the batch tiles too (libv10.so: set_role_priority up to 6 rows): 13B w4 s45 layer at 2-6 rows, libv9.so = the adopted batch-1-only rule; alternating
  E=tools/experiments/small_batch_r05.py
  for rep in 1 2 3; do for v in v9 v10; do
    (SQLLM_LIB=$PWD/squeezellm_amd/ab/lib$v.so timeout 300 python $E --rows 2,3,4,5,6 2>&1 | grep '^{' | sed "s/^{/{\"variant\": \"$v\", /") >> gpurun_out/r06_role_priority_tiles.txt
  done; done
  python - <<'PY'
import json
for l in open("gpurun_out/r06_role_priority_tiles.txt"):
    d = json.loads(l)
    print(d["variant"], d["rows"], d["layer_us"], d.get("qkv"), d.get("o"), d.get("gate_up"), d.get("down"))
PY
  ;;
final2)
  # the final tree (role-priority rules 1 + 2): full GPU suite, kernel traces of the two s45 configs the rules touch, default bench line
  (timeout 1800 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider 2>&1 | tail -5) > gpurun_out/r06_final2_tests.log
  tail -3 gpurun_out/r06_final2_tests.log
  R=$PWD; cd /tmp && export TMPDIR=/tmp
  for c in "kt_w3 7b-w3-s45" "kt_w4s45 7b-w4-s45"; do set -- $c
    (cd $R && timeout 400 python bench.py --config $2 --no-cpu-baseline --no-sub-records 2>/dev/null | grep '^{' > gpurun_out/r06_bench_samebox_$2.json)
    rm -rf /tmp/prof_$1; timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -o x -- python $R/bench.py --steps 20 --no-cpu-baseline --no-sub-records --config $2 > /tmp/prof_$1.log 2>&1
    grep '^{' /tmp/prof_$1.log > $R/gpurun_out/r06_$1.bench.json
    python $R/tools/rocprof_summary.py "$(find /tmp/prof_$1 -name '*.db' | head -1)" --by-grid --match sqllm --top 12 > $R/gpurun_out/r06_$1.summary.txt
    cat $R/gpurun_out/r06_$1.summary.txt; python -c "import json; d=json.load(open('$R/gpurun_out/r06_bench_samebox_$2.json')); print('$2 un-profiled on this box:', d['value'], d['roofline']['frac'])"
  done
  cd $R
  (timeout 900 python bench.py 2>/dev/null | grep '^{') > gpurun_out/r06_final2_bench.json
  python -c "
import json
d = json.load(open('gpurun_out/r06_final2_bench.json'))
print(d['value'], d['roofline']['frac'], d['roofline'].get('frac_rocprof'), {k: v.get('value') for k, v in d['sub_records'].items() if isinstance(v, dict)})
print({k: v['ms_per_decoder_layer'] for k, v in d['sub_records']['13b-w4-s45-batched'].items() if isinstance(v, dict) and 'ms_per_decoder_layer' in v})
"
  ;;
geometry_groups)
  # the q/k/v and gate/up GROUP launches under explicit per-op workgroup counts (tools/experiments/small_op_geometry.py --groups)
  (timeout 900 python tools/experiments/small_op_geometry.py --groups --bits 4 2>&1 | grep '^{') > gpurun_out/r06_group_geometry.txt
  (timeout 900 python tools/experiments/small_op_geometry.py --groups --bits 3 2>&1 | grep '^{') >> gpurun_out/r06_group_geometry.txt
  cat gpurun_out/r06_group_geometry.txt
  ;;
geom3)
  # libv11.so = libv9.so + two workgroups per CU and op for the 3-bit 7B gate/up pair (make_plan)
  (bash tools/ab_libs.sh "v9 v11" "7b-w3-s0 7b-w3-s45" 3 2>&1 | grep -v amdgpu.ids) > gpurun_out/r06_w3_gateup_geometry_ab.txt
  cat gpurun_out/r06_w3_gateup_geometry_ab.txt
  ;;
final3)
  # last check of the round's tree: the driver's own sequence -- GPU suite with -x, smoke, default bench line
  (timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4) > gpurun_out/r06_final3_tests.log; tail -2 gpurun_out/r06_final3_tests.log
  (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/r06_final3_smoke.log; cat gpurun_out/r06_final3_smoke.log
  (timeout 900 python bench.py 2>/dev/null | grep '^{') > gpurun_out/r06_final3_bench.json
  python -c "
import json
d = json.load(open('gpurun_out/r06_final3_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_rocprof'), d['cpu_baseline']['value'])
print({k: v.get('value') for k, v in d['sub_records'].items() if isinstance(v, dict)})
print({k: v['ms_per_decoder_layer'] for k, v in d['sub_records']['13b-w4-s45-batched'].items() if isinstance(v, dict) and 'ms_per_decoder_layer' in v})
print({k: v.get('tokens_per_s') for k, v in d['drop_in'].items() if isinstance(v, dict) and 'tokens_per_s' in v})
"
  ;;
cols356)
  # passes of exactly 3 / 5 / 6 rows in the COLUMN-LANE kernel too: libv9.so = before (4- / 8-row passes), libv12.so = with them and the 3-bit "3 / 5 rows -> tiles" rule,
  # libv13.so = with them and without that rule; 13B s45 layer, parity of the new instantiations first
  (timeout 900 python -m pytest tests/test_gpu_batched.py -m gpu -q -p no:cacheprovider -k "column_lane or three_batched or small_shapes or test_batch_tiles" 2>&1 | tail -2) > gpurun_out/r06_cols356_tests.log; cat gpurun_out/r06_cols356_tests.log
  E=tools/experiments/small_batch_r05.py
  for rep in 1 2; do for v in v9 v12 v13; do
    (SQLLM_LIB=$PWD/squeezellm_amd/ab/lib$v.so timeout 300 python $E --rows 2,3,4 2>&1 | grep '^{' | sed "s/^{/{\"variant\": \"$v\", /") >> gpurun_out/r06_cols356_w4.txt
    (SQLLM_LIB=$PWD/squeezellm_amd/ab/lib$v.so timeout 300 python $E --bits 3 --rows 3,5,6,7,8 2>&1 | grep '^{' | sed "s/^{/{\"variant\": \"$v\", /") >> gpurun_out/r06_cols356_w3.txt
  done; done
  python - <<'PY'
import json
for f in ("gpurun_out/r06_cols356_w4.txt", "gpurun_out/r06_cols356_w3.txt"):
    print(f)
    for l in open(f):
        d = json.loads(l)
        print(d["variant"], d["rows"], d["layer_us"], d.get("qkv"), d.get("o"), d.get("gate_up"), d.get("down"))
PY
  ;;
cols56)
  # libv13b.so = this tree (column-lane passes of exactly 3 / 5 / 6 / 7 rows; 4-bit column-lane kernel up to 4 rows), libv14.so = the same with the 4-bit column-lane
  # kernel up to 6 rows (three-op groups and single ops >= 20 MB), libv9.so = before the exact passes; 13B s45 layer
  E=tools/experiments/small_batch_r05.py
  (timeout 600 python -m pytest tests/test_gpu_batched.py -m gpu -q -p no:cacheprovider -k "column_lane or three_batched" 2>&1 | tail -2)
  for rep in 1 2; do for v in v9 v13b v14; do
    (SQLLM_LIB=$PWD/squeezellm_amd/ab/lib$v.so timeout 300 python $E --rows 3,5,6 2>&1 | grep '^{' | sed "s/^{/{\"variant\": \"$v\", /") >> gpurun_out/r06_cols56_w4.txt
    (SQLLM_LIB=$PWD/squeezellm_amd/ab/lib$v.so timeout 300 python $E --bits 3 --rows 3,5,6,7 2>&1 | grep '^{' | sed "s/^{/{\"variant\": \"$v\", /") >> gpurun_out/r06_cols56_w3.txt
  done; done
  python - <<'PY'
import json
for f in ("gpurun_out/r06_cols56_w4.txt", "gpurun_out/r06_cols56_w3.txt"):
    print(f)
    for l in open(f):
        d = json.loads(l)
        print(d["variant"], d["rows"], d["layer_us"], d.get("qkv"), d.get("o"), d.get("gate_up"), d.get("down"))
PY
  ;;
cols_single)
  # libv15.so = this tree (4-bit single ops >= 20 MB on the column-lane kernel up to 7 rows) against libv13b.so (up to 4 rows); 13B w4 s45 layer at 5-8 rows
  E=tools/experiments/small_batch_r05.py
  for rep in 1 2 3; do for v in v13b v15; do
    (SQLLM_LIB=$PWD/squeezellm_amd/ab/lib$v.so timeout 300 python $E --rows 5,6,7,8 2>&1 | grep '^{' | sed "s/^{/{\"variant\": \"$v\", /") >> gpurun_out/r06_cols_single_ops_5_7.txt
  done; done
  python - <<'PY'
import json
for l in open("gpurun_out/r06_cols_single_ops_5_7.txt"):
    d = json.loads(l)
    print(d["variant"], d["rows"], d["layer_us"], d.get("qkv"), d.get("o"), d.get("gate_up"), d.get("down"))
PY
  ;;
tile7)
  # the 7-row batch tile (libv16.so) against the tree before it (libprev.so: 7 rows on the 8-row tile): parity, then the 13B s45 layer at 7 rows, 4-bit and 3-bit
  (timeout 600 python -m pytest tests/test_gpu_batched.py -m gpu -q -p no:cacheprovider -k "test_batch_tiles_every_row_count or three_batched" 2>&1 | tail -1)
  E=tools/experiments/small_batch_r05.py
  for rep in 1 2 3; do for v in prev v16; do
    (SQLLM_LIB=$PWD/squeezellm_amd/ab/lib$v.so timeout 300 python $E --rows 7 2>&1 | grep '^{' | sed "s/^{/{\"variant\": \"$v\", /") >> gpurun_out/r06_tile7.txt
    (SQLLM_LIB=$PWD/squeezellm_amd/ab/lib$v.so timeout 300 python $E --bits 3 --rows 7 2>&1 | grep '^{' | sed "s/^{/{\"variant\": \"$v\", /") >> gpurun_out/r06_tile7.txt
  done; done
  python - <<'PY'
import json
for l in open("gpurun_out/r06_tile7.txt"):
    d = json.loads(l)
    print(d["variant"], d["config"], d["rows"], d["layer_us"], d.get("qkv"), d.get("o"), d.get("gate_up"), d.get("down"))
PY
  ;;
loadprio)
  # dense waves at s_setprio 2 while they ISSUE loads (v17: the first chunk only; v18: every chunk), back to the role's priority for the decode; libhead.so = this tree
  (bash tools/ab_libs.sh "head v17 v18" "7b-w4-s0 7b-w3-s45 7b-w4-s45 13b-w4-s45" 3 2>&1 | grep -v amdgpu.ids) > gpurun_out/r06_load_phase_priority_ab.txt
  cat gpurun_out/r06_load_phase_priority_ab.txt
  ;;
chunks)
  # the sparse roles' granularity again, now that their waves have a priority of their own: CSR chunk 1024 (product) / 2048 non-zeros, top-X slab 256 (product) / 512 / 128 k's
  # (builds of the product sources with the measurement switches: -DSQLLM_ABLATION_BUILD -DSQLLM_CSR_CHUNK=... / -DSQLLM_TOPX_ROWS=...)
  (bash tools/ab_libs.sh "c1024 c2048 t512 t128" "7b-w4-s45 7b-w3-s45 13b-w4-s45" 3 2>&1 | grep -v amdgpu.ids) > gpurun_out/r06_sparse_granularity_ab.txt
  cat gpurun_out/r06_sparse_granularity_ab.txt | cut -c1-60
  ;;
wide_chunks)
  # libwide.so = this tree (CSR chunks of 2048 non-zeros in batch-1 launches that exceed the resident slots with >= 1.25 x CUs sparse workgroups) against libprev.so (1024 everywhere);
  # parity of the wide role first (7B / 13B / 65B shapes at batch 1 against the C oracle)
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libwide.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_decoder_layer.py tests/test_gpu_batched.py -m gpu -q -p no:cacheprovider -k "llama7b or llama65b or decoder_layer or sparse_edge" 2>&1 | tail -2)
  (bash tools/ab_libs.sh "prev wide" "7b-w4-s45 7b-w3-s45 13b-w4-s45 65b-w3-s45" 3 2>&1 | grep -v amdgpu.ids) > gpurun_out/r06_wide_chunks_ab.txt
  cat gpurun_out/r06_wide_chunks_ab.txt
  ;;
wide_tiles)
  # wide CSR chunks on the batch tiles too (libv20.so: up to 5 rows, launches of more than 3 x CUs workgroups) against this tree (libhead.so: batch 1 only); 13B w4 s45 layer
  E=tools/experiments/small_batch_r05.py
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libv20.so timeout 600 python -m pytest tests/test_gpu_decoder_layer.py tests/test_gpu_batched.py -m gpu -q -p no:cacheprovider -k "decoder_layer or test_batch_tiles or llama13b" 2>&1 | tail -1)
  for rep in 1 2 3; do for v in head v20; do
    (SQLLM_LIB=$PWD/squeezellm_amd/ab/lib$v.so timeout 300 python $E --rows 2,3,4,5 2>&1 | grep '^{' | sed "s/^{/{\"variant\": \"$v\", /") >> gpurun_out/r06_wide_chunks_tiles.txt
  done; done
  python - <<'PY'
import json
for l in open("gpurun_out/r06_wide_chunks_tiles.txt"):
    d = json.loads(l)
    print(d["variant"], d["rows"], d["layer_us"], d.get("qkv"), d.get("o"), d.get("gate_up"), d.get("down"))
PY
  ;;
envknobs)
  # HIP runtime knobs that could move the launch boundary inside a replayed graph (~1.5 us x 128 launches per 7B token): same box, alternating, two repetitions
  # each; 7b-w4-s0 default line (graph replay), value = tokens/s, ms = wall per token, sum = sum of the four per-shape kernel means x 32 (events)
  for rep in 1 2; do
  for e in "BASE=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "AMD_OPT_FLUSH=0" "ROC_SYSTEM_SCOPE_SIGNAL=0" "GPU_MAX_HW_QUEUES=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=256" "ROC_USE_FGS_KERNARG=0" "DEBUG_HIP_KERNARG_COPY_OPT=0"; do
    (env $e timeout 200 python bench.py --config 7b-w4-s0 --no-cpu-baseline --no-sub-records 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$e', d['value'], d['ms_per_step'], round(32 * sum(v['us_mean'] for v in d['per_layer_us'].values()) / 1e3, 4), {k: v['us_mean'] for k, v in d['per_layer_us'].items()})
") >> gpurun_out/r06_env_knobs.txt
  done; done
  cat gpurun_out/r06_env_knobs.txt
  ;;
fuzz)
  # bug hunt: the randomised GPU parity test with fresh (non-derandomised) draws, three processes of FUZZ_N (default 1500) examples each
  for i in 1 2 3; do
    (SQLLM_PROPERTY_EXAMPLES=${FUZZ_N:-1500} timeout 1500 python -m pytest tests/test_gpu_property.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -25) > gpurun_out/r06_fuzz_$i.log
    tail -3 gpurun_out/r06_fuzz_$i.log
  done
  ;;
fuzz_large)
  # bug hunt at large shapes (multi-round launches, wide CSR chunks, role priorities, groups of 1-4 ops): the suite's 24 fixed draws, then FUZZ_N fresh ones
  (timeout 1200 python -m pytest tests/test_gpu_fuzz_large.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r06_fuzz_large_fixed.log
  tail -3 gpurun_out/r06_fuzz_large_fixed.log
  (SQLLM_FUZZ_LARGE=${FUZZ_N:-200} timeout 2400 python -m pytest tests/test_gpu_fuzz_large.py -m gpu -q -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/r06_fuzz_large_fresh.log
  tail -5 gpurun_out/r06_fuzz_large_fresh.log
  ;;
tile2_half)
  # the 4-bit 2-row batch tile on half stages at 64 VGPRs / NBUF 2 (four workgroups per CU; libv21.so, tools/build_variant.sh) against this tree (libhead.so): parity, then 13B s45 layer at 2 rows
  E=tools/experiments/small_batch_r05.py
  (SQLLM_LIB=$PWD/squeezellm_amd/ab/libv21.so timeout 600 python -m pytest tests/test_gpu_decoder_layer.py tests/test_gpu_batched.py -m gpu -q -p no:cacheprovider -k "decoder_layer or test_batch_tiles or llama13b" 2>&1 | tail -1)
  for rep in 1 2 3; do for v in head v21; do
    (SQLLM_LIB=$PWD/squeezellm_amd/ab/lib$v.so timeout 300 python $E --rows 2 2>&1 | grep '^{' | sed "s/^{/{\"variant\": \"$v\", /") >> gpurun_out/r06_tile2_half.txt
    (SQLLM_LIB=$PWD/squeezellm_amd/ab/lib$v.so timeout 300 python $E --rows 2 --dense-only 2>&1 | grep '^{' | sed "s/^{/{\"variant\": \"$v\", \"dense_only\": 1, /") >> gpurun_out/r06_tile2_half.txt
  done; done
  python - <<'PY'
import json
for l in open("gpurun_out/r06_tile2_half.txt"):
    d = json.loads(l)
    print(d["variant"], d.get("dense_only", 0), d["rows"], d["layer_us"], d.get("qkv"), d.get("o"), d.get("gate_up"), d.get("down"))
PY
  ;;
oproj_one_round)
  # an op of > 12 MB alone in its launch on a tile that holds three workgroups per CU: planned as one resident round (libtree2.so) against 800 workgroups (libtree.so)
  E=tools/experiments/small_batch_r05.py
  for rep in 1 2; do timeout 600 python $E --config 13b-w4-s45 --rows 3,4,5,6,7,8 --sets "default;target_wgs=768;target_wgs=640;target_wgs=512;target_wgs=400" 2>&1 | grep "^{"; done > gpurun_out/r06_oproj_target_sweep.txt
  for rep in 1 2 3; do for v in tree tree2; do
    (SQLLM_LIB=$PWD/squeezellm_amd/ab/lib$v.so timeout 300 python $E --config 13b-w4-s45 --rows 3,4,5,6 2>&1 | grep '^{' | sed "s/^{/{\"variant\": \"$v\", /") >> gpurun_out/r06_oproj_one_round.txt
  done; done
  ;;
geometry)
  # the planner against the residency of the kernel it plans for (second session): tools/experiments/launch_geometry.py sweeps of target_wgs per launch shape.  Outputs concatenated into
  # profiles/r06_oproj_one_round.txt, r06_launch_geometry_cols.txt, r06_launch_geometry_b1.txt with the same-box A/Bs of the variant libraries (libtree..libtree7.so: the tree before each rule)
  G=tools/experiments/launch_geometry.py
  for b in 2 4 5; do timeout 900 python $G --shapes "gateup13:5120x13824x2,qkv13:5120x5120x3" --batch $b --sparse 0.0045 --topx 10 --targets 256,384,512,640,768,1024 2>&1 | grep "^{" >> gpurun_out/r06_launch_geometry_13b.txt; done
  for b in 2 3 4 6; do timeout 900 python $G --shapes "down13:13824x5120x1" --batch $b --sparse 0.0045 --topx 10 --targets 320,400,480,560,640,720,800,880,960 2>&1 | grep "^{" >> gpurun_out/r06_launch_geometry_cols.txt; done
  for b in 3 4; do timeout 900 python $G --shapes "qkv13:5120x5120x3" --batch $b --sparse 0.0045 --topx 10 --targets 320,400,480,560,640,720,800,880,960 2>&1 | grep "^{" >> gpurun_out/r06_launch_geometry_cols.txt; done
  for b in 2 3 4; do timeout 900 python $G --bits 3 --shapes "qkv13:5120x5120x3" --batch $b --sparse 0.0045 --topx 10 --targets 480,720,960 2>&1 | grep "^{" >> gpurun_out/r06_launch_geometry_cols_w3.txt; done
  for b in 3 4; do timeout 900 python $G --shapes "qkv7:4096x4096x3,qkv65:8192x8192x3" --batch $b --sparse 0.0045 --topx 10 --targets 384,480,576,768 2>&1 | grep "^{" >> gpurun_out/r06_launch_geometry_cols_7b65b.txt; done
  for b in 2 3 4 5 6 8; do timeout 900 python $G --bits 3 --shapes "o13:5120x5120x1" --batch $b --sparse 0.0045 --topx 10 --targets 240,320,400,480,560,640 2>&1 | grep "^{" >> gpurun_out/r06_launch_geometry_o13_w3.txt; done
  for b in 7 8; do timeout 900 python $G --shapes "o7:4096x4096x1" --batch $b --sparse 0.0045 --topx 10 --targets 256,320,384,448 2>&1 | grep "^{" >> gpurun_out/r06_launch_geometry_o7.txt; done
  for bits in 4 3; do timeout 900 python $G --bits $bits --shapes "qkv7:4096x4096x3,o7:4096x4096x1,gateup7:4096x11008x2,down7:11008x4096x1" --sparse 0.0045 --topx 10 --targets 128,192,256,320,384,512,640,768,1024 2>&1 | grep "^{" >> gpurun_out/r06_launch_geometry_b1_s45.txt; done
  timeout 900 python $G --shapes "qkv13:5120x5120x3,o13:5120x5120x1,gateup13:5120x13824x2,down13:13824x5120x1" --sparse 0.0045 --topx 10 --targets 256,384,512,640,768,1024,1280 2>&1 | grep "^{" >> gpurun_out/r06_launch_geometry_b1_13b65b.txt
  timeout 900 python $G --bits 3 --shapes "qkv65:8192x8192x3,o65:8192x8192x1,gateup65:8192x22016x2,down65:22016x8192x1" --sparse 0.0045 --topx 10 --targets 256,384,512,768,1024,1536 2>&1 | grep "^{" >> gpurun_out/r06_launch_geometry_b1_13b65b.txt
  bash tools/ab_libs.sh "tree4 tree5" "13b-w4-s45 7b-w4-s0" 3   # one chunk per K slice
  bash tools/ab_libs.sh "tree5 tree6" "7b-w4-s0 7b-w4-s45 13b-w4-s45" 3   # two-step chunks for two-step slices
  bash tools/ab_libs.sh "tree6 tree7" "65b-w3-s45" 2   # 3-bit op alone at batch 1 cut to one round
  ;;
ceiling)
  # product and loads-only kernels on ONE clock (graph wall per launch, same box, same session)
  (/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/stream_patterns.hip -o /tmp/sp 2>&1 | tail -3)
  (timeout 1200 python tools/ceiling_same_clock.py --loads-only /tmp/sp --raw-out gpurun_out/r06_stream_patterns.txt 2>&1 | grep '^{') > gpurun_out/r06_ceiling_same_clock.txt
  cat gpurun_out/r06_ceiling_same_clock.txt
  ;;
timeline)
  # where the sparse workgroups of a batch-1 s45 launch sit, and of the 2- / 4-row launches (item 2d)
  L=$PWD/squeezellm_amd/libsqllm_hip_ablation.so
  for sh in "4096x4096 1" "4096x4096 3" "4096x11008 2" "11008x4096 1"; do set -- $sh
    (SQLLM_LIB=$L timeout 300 python tools/timeline.py --shape $1 --group $2 --bits 4 2>&1 | grep -v amdgpu.ids) >> gpurun_out/r06_timeline_batch1.txt
    (SQLLM_LIB=$L timeout 300 python tools/timeline.py --shape $1 --group $2 --bits 4 --sparse 0.0045 --topx 10 2>&1 | grep -v amdgpu.ids) >> gpurun_out/r06_timeline_batch1.txt
  done
  for b in 2 4; do for sh in "5120x5120 1" "5120x5120 3" "5120x13824 2" "13824x5120 1"; do set -- $sh
    (SQLLM_LIB=$L timeout 300 python tools/timeline.py --shape $1 --group $2 --bits 4 --batch $b 2>&1 | grep -v amdgpu.ids) >> gpurun_out/r06_timeline_rows_2_4.txt
    (SQLLM_LIB=$L timeout 300 python tools/timeline.py --shape $1 --group $2 --bits 4 --sparse 0.0045 --topx 10 --batch $b 2>&1 | grep -v amdgpu.ids) >> gpurun_out/r06_timeline_rows_2_4.txt
  done; done
  cat gpurun_out/r06_timeline_batch1.txt gpurun_out/r06_timeline_rows_2_4.txt
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
