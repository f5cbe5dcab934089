#!/usr/bin/env python3
"""Where do the sparse terms' microseconds go at 8 rows?  13B s45 decoder layer on the 8-row batch tiles (measurement
library), sum of the per-launch kernel times, with the CSR role's ablation bits: 1 = role skipped, 2 = no flush to global
memory, 4 = no accumulation, 8 = no x gathers.   SQLLM_LIB=squeezellm_amd/libsqllm_hip_ablation.so python tools/experiments/csr_ablate_small_batch.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench
from squeezellm_amd import _lib, decode

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
layers = bench.build_layers(bench.CONFIGS["13b-w4-s45"], dev, 0, 4)
for B in (4, 8):
    xs, ys = bench.decoder_inputs(layers, dev, gen, batch=B)
    _lib.set_option("mfma_min_batch", 1 << 20)
    for bits in (0, 8, 4, 12, 2, 1):
        _lib.set_option("ablate_csr", bits)
        seq = decode.OpSequence(layers, xs, ys, batched=True, fuse_shared_input=True)
        seq.profile(reps=1)
        us = seq.profile(reps=3).reshape(4, 4).mean(axis=0)
        print(json.dumps(dict(batch=B, ablate_csr=bits, qkv=round(us[0], 1), o=round(us[1], 1), gate_up=round(us[2], 1), down=round(us[3], 1),
                              layer=round(float(us.sum()), 1))), flush=True)
    _lib.set_option("ablate_csr", 0)
