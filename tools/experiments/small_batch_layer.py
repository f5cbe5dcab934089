#!/usr/bin/env python3
"""13B s45 decoder layer (grouped launches, graph replay) by batch rows and routing: microseconds per layer."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench
from squeezellm_amd import _lib, decode

dev = torch.device("cuda:0")
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = dict(bench.CONFIGS["13b-w4-s45"], bits=bits)
layers = bench.build_layers(cfg, dev, 0, 4)
gen = torch.Generator(device=dev).manual_seed(1)


def sync():
    torch.cuda.synchronize()


for B in (1, 2, 4, 5, 6, 8, 9, 12, 16):
    xs, ys = bench.decoder_inputs(layers, dev, gen, batch=0 if B == 1 else B)
    row = dict(bits=bits, batch=B)
    for tag, opts in (("default", {}), ("min5", dict(mfma_min_batch=5)), ("unfused_min5", dict(mfma_min_batch=5, mfma_fuse_small=0)),
                      ("fp32_mfma_min5", dict(mfma_min_batch=5, mfma_split=0)), ("tiles_only", dict(mfma_min_batch=1 << 20))):
        if B < 5 and tag != "default":
            continue
        for k, v in opts.items():
            _lib.set_option(k, v)
        seq = decode.OpSequence(layers, xs, ys, batched=B > 1, fuse_shared_input=True)
        g = seq.graph(warmup=1)
        blocks = bench.time_blocks(g.replay, sync, 20, 3, 3)
        row[tag] = round(statistics.median(blocks) / 20 / 4 * 1e6, 1)
        row[tag + "_launches"] = None
        for k in opts:
            _lib.set_option(k, 1 if k in ("mfma_split", "mfma_fuse_small") else 0)
        del g, seq
    print(json.dumps({k: v for k, v in row.items() if v is not None}), flush=True)
