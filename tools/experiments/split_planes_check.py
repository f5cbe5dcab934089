#!/usr/bin/env python3
"""Wide-batch dense term on the split matrix-core kernel: vec split in registers per column tile vs split once into
bf16 planes (option split_planes_min_batch) -- accuracy against the fp64 oracle and time per op (events around the op,
the split kernel included)."""
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
for bits in (() if "--timing-only" in sys.argv else (4, 3)):
    for K in (4096, 4160):
        case = H.make_case(bits, K, 640, seed=5 + bits)
        t = H.to_torch(case, dev)
        for B in (17, 100, 150):
            rng = np.random.default_rng(B)
            for kind in ("fp32", "fp16"):
                x = rng.normal(size=(B, K)).astype(np.float32)
                if kind == "fp16":
                    x = x.astype(np.float16).astype(np.float32)
                mul = rng.normal(size=(B, 640)).astype(np.float32)
                ref = H.c_matvec(lib, case, x, mul.copy(), batched=True)
                out = {}
                for wide in (1 << 30, 1):
                    for planes in (1 << 30, 1):
                        _lib.set_option("mfma_wide_min_batch", wide)
                        _lib.set_option("split_planes_min_batch", planes)
                        y = torch.from_numpy(mul.copy()).to(dev)
                        H.call_op(qc, t, torch.from_numpy(x).to(dev), y, "dense", True)
                        torch.cuda.synchronize()
                        out[("wide_" if wide == 1 else "tile_") + ("planes" if planes == 1 else "regs")] = float(f"{H.rel_err(y.cpu().numpy(), ref):.3e}")
                print(json.dumps(dict(bits=bits, K=K, batch=B, vec=kind, max_rel_err_vs_fp64=out)), flush=True)
_lib.set_option("split_planes_min_batch", 0)
_lib.set_option("mfma_wide_min_batch", 0)
if "--accuracy-only" in sys.argv:
    sys.exit(0)
OFF = 1 << 30
VARIANTS = (("tile_regs", OFF, OFF, "fp32"), ("wide_regs", 1, OFF, "fp32"), ("wide_planes_fp32", 1, 1, "fp32"), ("wide_planes_fp16", 1, 1, "fp16"))
for shape, bits in (((5120, 13824), 4), ((13824, 5120), 4), ((5120, 5120), 4), ((5120, 13824), 3)):
    K, N = shape
    layers = [synth.make_layer(K, N, bits, device=dev, seed=i) for i in range(4)]
    for B in (64, 128, 256, 384, 512, 1024, 2048):
        row = dict(shape=f"{K}x{N}", bits=bits, batch=B)
        for name, wide, planes, kind in VARIANTS:
            xs = [torch.randn((B, K), device=dev) for _ in layers]
            if kind == "fp16":
                xs = [x.half().float() for x in xs]
            ys = [torch.zeros((B, N), device=dev) for _ in xs]
            _lib.set_option("mfma_wide_min_batch", wide)
            _lib.set_option("split_planes_min_batch", planes)
            seq = decode.OpSequence(layers, xs, ys, batched=True)
            seq.profile(reps=1)
            us = seq.profile(reps=3)
            row[name + "_us"] = round(float(us.mean()), 2)
            del xs, ys
        _lib.set_option("split_planes_min_batch", 0)
        _lib.set_option("mfma_wide_min_batch", 0)
        row["TFLOPs_best"] = round(2.0 * B * K * N / min(v for k, v in row.items() if k.endswith("_us")) / 1e6, 1)
        print(json.dumps(row), flush=True)
    del layers
    torch.cuda.empty_cache()
