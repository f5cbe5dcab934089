#!/bin/bash
# Round 3, GPU session 27: CSR chunk tables -- tests, timeline with the table, A/B on the s45 configs (option chunk_tables)
O=gpurun_out/r03_s27; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_chunk_table.py -x -q -m gpu 2>&1 | tail -15
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
AB=squeezellm_amd/libsqllm_hip_ablation.so
for opt in "chunk_tables=0" "chunk_tables=1"; do
 for spec in "4096x4096 1" "4096x4096 3" "11008x4096 1"; do
  set -- $spec
  echo "== $opt" | tee -a $O/timeline_csr.txt
  SQLLM_OPTIONS=$opt SQLLM_LIB=$AB timeout 200 python tools/timeline.py --shape $1 --bits 4 --group $2 --sparse 0.0045 --topx 10 2>>$O/err.txt | grep -v "^  [wbef]" | tee -a $O/timeline_csr.txt
 done
done
for rep in 1 2 3; do
 for opt in "chunk_tables=0" "chunk_tables=1"; do
  for c in 7b-w4-s45 7b-w3-s45; do
    SQLLM_OPTIONS=$opt timeout 200 python bench.py --config $c --no-cpu-baseline --no-sub-records 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$opt', d['config']['config_name'], d['value'], d['repeats']['value_median'], {k: v['us_mean'] for k, v in d['per_layer_us'].items()})" | tee -a $O/ab_tables.txt
  done
 done
done
for opt in "chunk_tables=0" "chunk_tables=1"; do
SQLLM_OPTIONS=$opt timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$opt', d['value'], {b: v['ms_per_decoder_layer'] for b, v in d['sub_records']['13b-w4-s45-batched'].items() if b.startswith('batch')})" | tee -a $O/ab_tables.txt
done
