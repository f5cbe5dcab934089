#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf /tmp/prof_kt; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o x -- python $R/tools/batch_sweep.py --paths mfma --batches 2048 --reps 2 > /tmp/kt.log 2>&1
python $R/tools/rocprof_summary.py "$(find /tmp/prof_kt -name '*.db' | head -1)" --by-grid --match sqllm --top 16 > $R/gpurun_out/r04_s34_sparse_half.txt
grep '^{' /tmp/kt.log >> $R/gpurun_out/r04_s34_sparse_half.txt
cat $R/gpurun_out/r04_s34_sparse_half.txt
