#!/bin/bash
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
sum() { python $R/tools/rocprof_summary.py "$(find $1 -name '*.db' | head -1)" --by-grid --match sqllm --top 12 "${@:2}"; }
run() { name=$1; shift; rm -rf /tmp/prof_$name; timeout 300 rocprofv3 "$@" > /tmp/prof_$name.log 2>&1 || tail -5 /tmp/prof_$name.log; }
out=$R/gpurun_out/r04_s25_wide_pmc.txt
: > $out
B=$R/tools/experiments/wide_ablate/bin/base
run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d /tmp/prof_sq -o x -- $B 4 2048 0 0
sum /tmp/prof_sq >> $out
run tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace -d /tmp/prof_tcp -o x -- $B 4 2048 0 0
sum /tmp/prof_tcp | grep -A100 "PMC" >> $out
run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d /tmp/prof_tcc -o x -- $B 4 2048 0 0
sum /tmp/prof_tcc | grep -A100 "PMC" >> $out
run ta --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum --kernel-trace -d /tmp/prof_ta -o x -- $B 4 2048 0 0
sum /tmp/prof_ta | grep -A100 "PMC" >> $out
run sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --kernel-trace -d /tmp/prof_sq2 -o x -- $B 4 2048 0 0
sum /tmp/prof_sq2 | grep -A100 "PMC" >> $out
cat $out
