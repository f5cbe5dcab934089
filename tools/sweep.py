#!/usr/bin/env python3
"""Kernel sweep: per-kernel device time of one op shape over many distinct weight copies (so the
stream comes from HBM, not L2 / Infinity Cache), for a list of launch-geometry options.

    python tools/sweep.py --shapes 4096x4096,4096x11008 --bits 4 --target-wgs 256,512,1024
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="4096x4096,4096x11008,11008x4096")
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--sparse", type=float, default=0.0)
    ap.add_argument("--topx", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--target-wgs", default="0")
    ap.add_argument("--gpw", default="0")
    ap.add_argument("--variant", default="0", help="comma list: waves*10+prefetch (ablation build)")
    ap.add_argument("--ablate", default="0", help="comma list of ablation masks (needs SQLLM_ABLATION=1 build)")
    ap.add_argument("--total-mb", type=float, default=700.0, help="distinct weight bytes to rotate over")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--sparse-last", type=int, default=0)
    ap.add_argument("--ablate-csr", type=int, default=0, help="CSR-role ablation bits (ablation build): 1 skip role, 2 skip flush, 4 skip accumulation")
    ap.add_argument("--group", type=int, default=1, help="ops per launch (sharing one input vector)")
    args = ap.parse_args()
    import numpy as np
    import torch

    from squeezellm_amd import _lib, decode, synth

    dev = torch.device("cuda:0")
    _lib.set_option("sparse_last", args.sparse_last)
    if args.ablate_csr:
        _lib.set_option("ablate_csr", args.ablate_csr)
    rows = []
    for shp in args.shapes.split(","):
        K, N = map(int, shp.split("x"))
        one = synth.algorithmic_bytes(K, N, args.bits)
        copies = max(4 * args.group, int(args.total_mb * 1e6 / one) // args.group * args.group)
        layers = [synth.make_layer(K, N, args.bits, sparse_frac=args.sparse, topX=args.topx,
                                   heavy_rows=10 if args.sparse > 0 else 0, device=dev, seed=i) for i in range(copies)]
        B = max(args.batch, 1)
        xs = []
        for i in range(len(layers)):
            xs.append(xs[-1] if i % args.group else torch.randn((B, K) if args.batch else (K,), device=dev))
        ys = [torch.zeros((B, N) if args.batch else (N,), device=dev) for _ in layers]
        seq = decode.OpSequence(layers, xs, ys, batched=args.batch > 0, fuse_shared_input=args.group > 1)
        nbytes = synth.layer_bytes(layers[0], B) * args.group
        for tw in map(int, args.target_wgs.split(",")):
         for var in [0]:
          for abl in map(int, args.ablate.split(",")):
            for gpw in map(int, args.gpw.split(",")):
                if abl or args.ablate != "0":
                    _lib.set_option("ablate", abl)
                _lib.set_option("target_wgs", tw)
                _lib.set_option("groups_per_wave", gpw)
                seq2 = decode.OpSequence(layers, xs, ys, batched=args.batch > 0, fuse_shared_input=args.group > 1)
                plan = _lib.plan_query(args.bits, K, N, args.batch, nnz=layers[0]["vals"].numel() if args.sparse else 0, topX=args.topx)
                seq2.profile(reps=1)
                us = seq2.profile(reps=args.reps)
                # wall clock of a graph replay of all copies back to back: (kernel + boundary) per op
                g = seq2.graph(warmup=1)
                g.replay(); torch.cuda.synchronize()
                import time
                t0 = time.perf_counter()
                for _ in range(args.reps * 4):
                    g.replay()
                torch.cuda.synchronize()
                wall_us = (time.perf_counter() - t0) / (args.reps * 4) / seq2.n_groups * 1e6
                r = dict(shape=shp, group=args.group, bits=args.bits, batch=args.batch, variant=var, ablate=abl, target_wgs=tw, gpw=plan["groups_per_wave"], grid=plan["grid_x"],
                         k_slices=plan["k_slices"], wall_us=round(wall_us, 3), wall_GBps=round(nbytes / wall_us / 1e3, 1), us_mean=round(float(us.mean()), 3), us_min=round(float(us.min()), 3),
                         GBps=round(nbytes / us.mean() / 1e3, 1), frac=round(nbytes / us.mean() / 1e3 / 8000, 4), copies=copies)
                rows.append(r)
                print(json.dumps(r), flush=True)
        del layers, xs, ys, seq
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
