#!/usr/bin/env python3
"""Accuracy and time of the wide-batch dense term: fp32 matrix instruction vs bf16 matrix instructions on split operands."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from squeezellm_amd import _lib, decode, quant_cuda as qc, synth
from tests import helpers as H

dev = torch.device("cuda:0")
lib = H.c_oracle()
_lib.set_option("mfma_min_batch", 1)
for bits in (4, 3):
    case = H.make_case(bits, 4096, 1024, seed=5 + bits)
    t = H.to_torch(case, dev)
    for B in (16, 48):
        rng = np.random.default_rng(B)
        x = rng.normal(size=(B, 4096)).astype(np.float32)
        mul = np.zeros((B, 1024), np.float32)
        ref = H.c_matvec(lib, case, x, mul, batched=True)
        out = {}
        for split in (0, 1):
            _lib.set_option("mfma_split", split)
            y = torch.zeros(B, 1024, device=dev)
            H.call_op(qc, t, torch.from_numpy(x).to(dev), y, "dense", True)
            torch.cuda.synchronize()
            out["split" if split else "fp32"] = float(f"{H.rel_err(y.cpu().numpy(), ref):.3e}")
        print(json.dumps(dict(bits=bits, batch=B, K=4096, N=1024, max_rel_err_vs_fp64=out)), flush=True)
_lib.set_option("mfma_min_batch", 0)
for shape, bits in (((5120, 13824), 4), ((5120, 5120), 4), ((13824, 5120), 4), ((5120, 13824), 3)):
    K, N = shape
    copies = max(4, int(600e6 / synth.algorithmic_bytes(K, N, bits)))
    layers = [synth.make_layer(K, N, bits, device=dev, seed=i) for i in range(copies)]
    for B in (8, 9, 16, 32, 64, 256, 2048):
        xs = [torch.randn((B, K), device=dev) for _ in layers[:4 if B > 256 else copies]]
        ys = [torch.zeros((B, N), device=dev) for _ in xs]
        row = dict(shape=f"{K}x{N}", bits=bits, batch=B)
        for split in (0, 1):
            _lib.set_option("mfma_split", split)
            _lib.set_option("mfma_min_batch", 1)
            seq = decode.OpSequence(layers[:len(xs)], xs, ys, batched=True)
            seq.profile(reps=1)
            us = seq.profile(reps=3)
            row["split_us" if split else "fp32_us"] = round(float(us.mean()), 2)
        _lib.set_option("mfma_min_batch", 0)
        row["speedup"] = round(row["fp32_us"] / row["split_us"], 2)
        row["TFLOPs_split"] = round(2.0 * B * K * N / row["split_us"] / 1e6, 1)
        print(json.dumps(row), flush=True)
        del xs, ys
    del layers
    torch.cuda.empty_cache()
