#!/usr/bin/env python3
"""Per-launch microseconds of a 13B decoder layer at a few batch sizes: hybrid (s45) and dense-only, tiles vs fused small split."""
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
for name in ("13b-w4-s45", "13b-w4-s0"):
    cfg = dict(bench.CONFIGS["13b-w4-s45"])
    if name.endswith("s0"):
        cfg.update(sparse=0.0, topX=0)
    layers = bench.build_layers(cfg, dev, 0, 4)
    for B in (8, 16):
        xs, ys = bench.decoder_inputs(layers, dev, gen, batch=B)
        for tag, opts in (("tiles", dict(mfma_min_batch=1 << 20)), ("small_split", dict(mfma_min_batch=5))):
            for k, v in opts.items():
                _lib.set_option(k, v)
            seq = decode.OpSequence(layers, xs, ys, batched=True, fuse_shared_input=True)
            seq.profile(reps=1)
            us = seq.profile(reps=3).reshape(4, 4).mean(axis=0)
            print(json.dumps(dict(config=name, batch=B, path=tag, qkv=round(us[0], 1), o=round(us[1], 1), gate_up=round(us[2], 1), down=round(us[3], 1),
                                  layer=round(float(us.sum()), 1))), flush=True)
            _lib.set_option("mfma_min_batch", 0)
    del layers
    torch.cuda.empty_cache()
