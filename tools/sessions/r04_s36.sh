#!/bin/bash
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
sum() { python $R/tools/rocprof_summary.py "$(find $1 -name '*.db' | head -1)" --by-grid --match sqllm --top 12 "${@:2}"; }
run() { name=$1; shift; rm -rf /tmp/prof_$name; timeout 300 rocprofv3 "$@" > /tmp/prof_$name.log 2>&1 || tail -5 /tmp/prof_$name.log; }
out=$R/gpurun_out/r04_s36_wide_pmc_final.txt
: > $out
B=$R/tools/experiments/wide_ablate/bin/base
run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d /tmp/prof_sq -o x -- $B 4 2048 0 0
sum /tmp/prof_sq >> $out
run fetch --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d /tmp/prof_fetch -o x -- $B 4 2048 0 0
sum /tmp/prof_fetch | grep -A100 "PMC" >> $out
cat $out
