#!/bin/bash
# Round 3, GPU session 13: where does the time of the small batches (2/4/8 rows, 13B shapes) go?  dense vs hybrid, tile vs cols
O=gpurun_out/r03_s13; mkdir -p $O
for shp in 5120x5120 5120x13824 13824x5120; do
  for sp in "0 0" "0.0045 10"; do
    set -- $sp
    echo "== $shp sparse $1 topx $2" | tee -a $O/batch_sweep.txt
    timeout 300 python tools/batch_sweep.py --shape $shp --sparse $1 --topx $2 --batches 1,2,4,8 --paths tile,cols --reps 3 2>>$O/err.txt | tee -a $O/batch_sweep.txt
  done
done
