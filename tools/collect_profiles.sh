#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel traces of the default bench (w4 s0) and of w3 s45, FETCH_SIZE / WRITE_SIZE / SQ passes.
# Databases stay in /tmp (tens of MB); only text summaries and the bench JSON go to gpurun_out/.
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01'
tag=${1:-r01}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
sum() { python $R/tools/rocprof_summary.py "$(find $1 -name '*.db' | head -1)" --by-grid --match sqllm --top 12 "${@:2}"; }
run() {  # name, rocprof args..., -- bench args
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 500 rocprofv3 "$@" > /tmp/prof_$name.log 2>&1
}
# the un-profiled default line of THIS box first: the profiles below are quoted beside it
(cd $R && timeout 400 python bench.py --no-cpu-baseline --no-sub-records 2>/dev/null | grep '^{' > gpurun_out/${tag}_bench_samebox_7b-w4-s0.json)
run kt_w4 --kernel-trace --stats -d /tmp/prof_kt_w4 -o x -- python $R/bench.py --steps 20 --no-cpu-baseline --no-sub-records
grep '^{' /tmp/prof_kt_w4.log > $R/gpurun_out/${tag}_kt_w4.bench.json
sum /tmp/prof_kt_w4 > $R/gpurun_out/${tag}_kt_w4.summary.txt
run kt_w3 --kernel-trace --stats -d /tmp/prof_kt_w3 -o x -- python $R/bench.py --steps 20 --no-cpu-baseline --no-sub-records --config 7b-w3-s45
grep '^{' /tmp/prof_kt_w3.log > $R/gpurun_out/${tag}_kt_w3.bench.json
sum /tmp/prof_kt_w3 > $R/gpurun_out/${tag}_kt_w3.summary.txt
run kt_w4_unfused --kernel-trace --stats -d /tmp/prof_kt_w4_unfused -o x -- python $R/bench.py --steps 20 --no-cpu-baseline --no-sub-records --no-fuse
sum /tmp/prof_kt_w4_unfused > $R/gpurun_out/${tag}_kt_w4_unfused.summary.txt
for pmc in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $pmc | tr A-Z a-z | sed 's/_size//')
  run pmc_$n --pmc $pmc --kernel-trace -d /tmp/prof_pmc_$n -o x -- python $R/bench.py --steps 3 --warmup 1 --launch sequence --no-cpu-baseline --no-sub-records --no-roofline --repeats 1
  sum /tmp/prof_pmc_$n > $R/gpurun_out/${tag}_pmc_$n.summary.txt
done
run pmc_fetch_w3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_pmc_fetch_w3 -o x -- python $R/bench.py --steps 3 --warmup 1 --launch sequence --no-cpu-baseline --no-sub-records --no-roofline --repeats 1 --config 7b-w3-s45
sum /tmp/prof_pmc_fetch_w3 > $R/gpurun_out/${tag}_pmc_fetch_w3.summary.txt
run pmc_fetch_w4s45 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_pmc_fetch_w4s45 -o x -- python $R/bench.py --steps 3 --warmup 1 --launch sequence --no-cpu-baseline --no-sub-records --no-roofline --repeats 1 --config 7b-w4-s45
sum /tmp/prof_pmc_fetch_w4s45 > $R/gpurun_out/${tag}_pmc_fetch_w4s45.summary.txt
run kt_w4s45 --kernel-trace --stats -d /tmp/prof_kt_w4s45 -o x -- python $R/bench.py --steps 20 --no-cpu-baseline --no-sub-records --config 7b-w4-s45
grep '^{' /tmp/prof_kt_w4s45.log > $R/gpurun_out/${tag}_kt_w4s45.bench.json
sum /tmp/prof_kt_w4s45 > $R/gpurun_out/${tag}_kt_w4s45.summary.txt
run pmc_sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/prof_pmc_sq -o x -- python $R/bench.py --steps 3 --warmup 1 --launch sequence --no-cpu-baseline --no-sub-records --no-roofline --repeats 1 --layers 8
sum /tmp/prof_pmc_sq > $R/gpurun_out/${tag}_pmc_sq.summary.txt
run pmc_sq_waits --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT --kernel-trace -d /tmp/prof_pmc_sq_waits -o x -- python $R/bench.py --steps 3 --warmup 1 --launch sequence --no-cpu-baseline --no-sub-records --no-roofline --repeats 1 --layers 8
sum /tmp/prof_pmc_sq_waits > $R/gpurun_out/${tag}_pmc_sq_waits.summary.txt
# wide-batch (matrix-core) path: 13B gate/up shape, hybrid, 16 and 2048 rows
run kt_batched --kernel-trace --stats -d /tmp/prof_kt_batched -o x -- python $R/tools/batch_sweep.py --paths mfma --batches 16,2048 --reps 2
sum /tmp/prof_kt_batched > $R/gpurun_out/${tag}_kt_batched.summary.txt
grep '^{' /tmp/prof_kt_batched.log >> $R/gpurun_out/${tag}_kt_batched.summary.txt
run pmc_batched_fetch --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_pmc_batched_fetch -o x -- python $R/tools/batch_sweep.py --paths mfma --batches 16,2048 --reps 1
sum /tmp/prof_pmc_batched_fetch > $R/gpurun_out/${tag}_pmc_batched_fetch.summary.txt
run pmc_batched_mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d /tmp/prof_pmc_batched_mfma -o x -- python $R/tools/batch_sweep.py --paths mfma --batches 16,2048 --reps 1 --sparse 0 --topx 0
sum /tmp/prof_pmc_batched_mfma > $R/gpurun_out/${tag}_pmc_batched_mfma.summary.txt
# small batches: batch tiles vs column-lane kernel, 13B gate/up shape, dense and hybrid
run kt_small_batches --kernel-trace --stats -d /tmp/prof_kt_small -o x -- python $R/tools/batch_sweep.py --paths tile,cols --batches 2,4,8 --reps 2 --sparse 0 --topx 0
sum /tmp/prof_kt_small > $R/gpurun_out/${tag}_kt_small_batches.summary.txt
# configs[3], 13B w4 s45 decoder layer by batch rows: kernel trace at 8 / 16 rows and the same-box A/B against the previous round's build
run kt_13b_rows --kernel-trace --stats -d /tmp/prof_kt_13b_rows -o x -- python $R/tools/experiments/small_batch_r05.py --rows 8,16 --no-breakdown
sum /tmp/prof_kt_13b_rows > $R/gpurun_out/${tag}_kt_13b_rows.summary.txt
(cd $R && timeout 300 python tools/experiments/small_batch_r05.py --rows 1,2,4,5,6,8,12,16 2>/dev/null | grep '^{' > gpurun_out/${tag}_small_batch_ab.txt
 [ -f squeezellm_amd/ab/libr04.so ] && SQLLM_LIB=$R/squeezellm_amd/ab/libr04.so timeout 300 python tools/experiments/small_batch_r05.py --rows 1,2,4,5,6,8,12,16 2>/dev/null | grep '^{' >> gpurun_out/${tag}_small_batch_ab.txt)
# un-profiled bench lines, all configs, one box
cd $R
for c in 7b-w4-s0 7b-w3-s45 7b-w4-s45 7b-w3-s0 13b-w4-s45 65b-w3-s45; do
  extra="--no-sub-records"; [ $c = 7b-w4-s0 ] && extra=""
  timeout 400 python bench.py --config $c $extra 2>/dev/null | grep '^{' > gpurun_out/${tag}_bench_$c.json
done
timeout 300 python bench.py --no-fuse --no-sub-records --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${tag}_bench_7b-w4-s0_unfused.json
ls -la $R/gpurun_out/ | grep ${tag}_
