#!/bin/bash
# Round 3, GPU session 32: wide-batch sparse role with lane groups (16 / 32 / 64 rows per pass): parity, then timing
O=gpurun_out/r03_s32; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_property.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/batch_sweep.py --paths mfma --batches 9,16,17,32,33,64,128,256,2048 --reps 3 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('lane groups', d['shape'], 'rows', d['batch'], d['path'], 'wall', d.get('wall_us'), 'ev', d.get('us_mean'))" | tee -a $O/wide_sparse.txt
SQLLM_LIB=squeezellm_amd/ab/prev.so timeout 600 python tools/batch_sweep.py --paths mfma --batches 9,16,17,32,33,64,128,256,2048 --reps 3 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('before', d['shape'], 'rows', d['batch'], d['path'], 'wall', d.get('wall_us'), 'ev', d.get('us_mean'))" | tee -a $O/wide_sparse.txt
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ws -o x -- python $OLDPWD/tools/batch_sweep.py --paths mfma --batches 16,32 --reps 2 > /tmp/prof_ws.log 2>&1; cd $OLDPWD
python tools/rocprof_summary.py "$(find /tmp/prof_ws -name '*.db' | head -1)" --by-grid --match sqllm --top 12 | tee $O/kt_wide_sparse.txt
