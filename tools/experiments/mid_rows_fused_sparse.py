#!/usr/bin/env python3
"""Hybrid ops at 17..127 rows (tile form of the split matrix-core kernel): the sparse terms as a launch of their own
(mfma_fuse_sparse = 0) vs in the dense launch's grid (1).  13B shapes, s45; microseconds per op (events), fp16-born vec."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from squeezellm_amd import _lib, decode, synth

dev = torch.device("cuda:0")
for bits in (4, 3):
    for K, N in ((5120, 13824), (13824, 5120), (5120, 5120)):
        layers = [synth.make_layer(K, N, bits, sparse_frac=0.0045, topX=10, heavy_rows=10, device=dev, seed=i) for i in range(4)]
        for B in (17, 24, 32, 48, 64, 100, 127):
            xs = [torch.randn((B, K), device=dev, dtype=torch.float16).float() for _ in layers]
            ys = [torch.zeros((B, N), device=dev) for _ in xs]
            row = dict(shape=f"{K}x{N}", bits=bits, batch=B)
            for fuse in (0, 1):
                _lib.set_option("mfma_fuse_sparse", fuse)
                _lib.set_option("mfma_wide_min_batch", 1 << 30)  # the tile form at every size here
                seq = decode.OpSequence(layers, xs, ys, batched=True)
                seq.profile(reps=1)
                row["fused_us" if fuse else "separate_us"] = round(float(seq.profile(reps=3).mean()), 2)
            _lib.set_option("mfma_fuse_sparse", 1)
            _lib.set_option("mfma_wide_min_batch", 0)
            seq = decode.OpSequence(layers, xs, ys, batched=True)
            seq.profile(reps=1)
            row["default_us"] = round(float(seq.profile(reps=3).mean()), 2)
            print(json.dumps(row), flush=True)
        del layers
        torch.cuda.empty_cache()
