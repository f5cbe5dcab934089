#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf /tmp/prof_kt; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o x -- python $R/tools/batch_sweep.py --paths mfma --batches 32,64,128 --reps 2 > /tmp/kt.log 2>&1
python $R/tools/rocprof_summary.py "$(find /tmp/prof_kt -name '*.db' | head -1)" --by-grid --match sqllm --top 20 > $R/gpurun_out/r04_s42_kt_mid_rows.txt
grep '^{' /tmp/kt.log >> $R/gpurun_out/r04_s42_kt_mid_rows.txt
cat $R/gpurun_out/r04_s42_kt_mid_rows.txt
