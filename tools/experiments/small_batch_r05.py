#!/usr/bin/env python3
"""Round 5, configs[3]: 13B s45 decoder layer (grouped launches) by batch rows -- graph-replay microseconds per layer and the
per-launch kernel microseconds (q/k/v, o, gate/up, down), under option sets given on the command line.

    [SQLLM_LIB=squeezellm_amd/ab/libr04.so] python tools/experiments/small_batch_r05.py [--bits 4] [--rows 1,2,4,5,8,16]
        [--sets "default;small_wgs_per_cu=2;mfma_min_batch=2"] [--dense-only]

A set is a comma-separated list of option=value (applied for the measurement, then restored to 0 / the library default);
options a library does not know (the round-4 build under SQLLM_LIB) are reported as "n/a".
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench
from squeezellm_amd import _lib, decode

DEFAULTS = dict(mfma_split=1, mfma_fuse_small=1, mfma_fuse_sparse=1, cols_groups=1, sparse_transpose=1, small_planes=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--rows", default="1,2,4,5,8,16")
    ap.add_argument("--sets", default="default")
    ap.add_argument("--dense-only", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--config", default="13b-w4-s45", help="a bench.py config for the shapes (7b-w4-s45, 65b-w3-s45 ...); --bits overrides its width")
    ap.add_argument("--no-ws", action="store_true", help="the workspace-less entry points (no transposed vec for the folded CSR walk)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = dict(bench.CONFIGS[a.config], bits=a.bits)
    if a.dense_only:
        cfg.update(sparse=0.0, topX=0)
    layers = bench.build_layers(cfg, dev, 0, 4)
    gen = torch.Generator(device=dev).manual_seed(1)
    sets = []
    for s in a.sets.split(";"):
        s = s.strip()
        sets.append((s, {} if s == "default" else {kv.split("=")[0]: int(kv.split("=")[1]) for kv in s.split(",")}))

    def sync():
        torch.cuda.synchronize()

    for B in [int(r) for r in a.rows.split(",")]:
        xs, ys = bench.decoder_inputs(layers, dev, gen, batch=0 if B == 1 else B)
        for tag, opts in sets:
            row = dict(lib=os.path.basename(os.environ.get("SQLLM_LIB", "HEAD")), config="%s-w%d-%s" % (a.config.split("-")[0], a.bits, "s0" if a.dense_only else "s45"), rows=B, set=tag)
            try:
                for k, v in opts.items():
                    _lib.set_option(k, v)
            except Exception as e:  # an option this library does not have
                row["layer_us"] = "n/a (%s)" % type(e).__name__
                print(json.dumps(row), flush=True)
                for k in opts:
                    try:
                        _lib.set_option(k, DEFAULTS.get(k, 0))
                    except Exception:
                        pass
                continue
            seq = decode.OpSequence(layers, xs, ys, batched=B > 1, fuse_shared_input=True, workspace=not a.no_ws)
            row["ws"] = 0 if seq._ws is None else seq._ws.numel()
            g = seq.graph(warmup=1)
            blocks = bench.time_blocks(g.replay, sync, 20, 3, 3)
            row["layer_us"] = round(statistics.median(blocks) / 20 / 4 * 1e6, 1)
            del g
            if not a.no_breakdown:
                seq.profile(reps=1)
                us = seq.profile(reps=3).reshape(4, 4).mean(axis=0)
                row.update(qkv=round(float(us[0]), 1), o=round(float(us[1]), 1), gate_up=round(float(us[2]), 1), down=round(float(us[3]), 1),
                           kernels=round(float(us.sum()), 1))
            del seq
            for k in opts:
                _lib.set_option(k, DEFAULTS.get(k, 0))
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
