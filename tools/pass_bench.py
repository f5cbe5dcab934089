#!/usr/bin/env python3
"""Same-box A/B of the two ways to run a decode pass: one launch per group (sqllm_launch_groups, graph replay) against
the dependency-gated persistent pass (sqllm_pass_*, graph replay), on bench.py's own workload.

    python tools/pass_bench.py [--config 7b-w4-s0] [--layers 32] [--steps 30] [--sweep]

Prints one JSON line per measurement.  --sweep walks the pass's knobs (poll interval, resident workgroups per CU,
planner target) after the default measurement."""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def timed(fn, sync, steps, warmup=3, repeats=3):
    blocks = bench.time_blocks(fn, sync, steps, warmup, repeats)
    return statistics.median(blocks) / steps * 1e3, min(blocks) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="7b-w4-s0", choices=sorted(bench.CONFIGS))
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--chain", action="store_true", help="chain the activations (group g + 1 reads group g's output buffer)")
    args = ap.parse_args()
    import torch

    from squeezellm_amd import _lib, decode, experimental, synth

    dev = torch.device("cuda", 0)
    cfg = bench.CONFIGS[args.config]
    spec = synth.MODEL_SHAPES[cfg["model"]]
    n_layers = spec["layers"] if args.layers is None else args.layers
    layers = bench.build_layers(cfg, dev, 0, n_layers)
    pass_bytes = float(sum(synth.layer_bytes(l, 1) for l in layers))
    gen = torch.Generator(device=dev).manual_seed(1234)
    xs, ys = bench.decoder_inputs(layers, dev, gen)

    def sync():
        torch.cuda.synchronize(dev)

    def emit(**kw):
        ms = kw["ms_per_token"]
        kw["tokens_per_s"] = round(1e3 / ms, 1)
        kw["hbm_frac_wall"] = round(pass_bytes / (ms * 1e-3) / 8e12, 4)
        kw["ms_per_token"] = round(ms, 4)
        print(json.dumps(kw), flush=True)

    seq = decode.OpSequence(layers, xs, ys, fuse_shared_input=True)
    g = seq.graph(warmup=1)
    med, best = timed(g.replay, sync, args.steps)
    emit(path="grouped_launches_graph", config=args.config, layers=n_layers, launches=seq.n_groups, ms_per_token=med, ms_min=round(best, 4))
    us = seq.profile(reps=3)
    emit(path="grouped_launches_sum_of_kernels", config=args.config, ms_per_token=float(us.sum()) * 1e-3)

    def run_pass(tag, **opts):
        for k, v in opts.items():
            experimental.set_option(k, v)
        p = experimental.GatedPass(seq)
        gp = p.graph(warmup=1)
        med, best = timed(gp.replay, sync, args.steps)
        err = p.status()
        kus = p.profile(reps=3)
        emit(path="gated_pass_graph", tag=tag, config=args.config, layers=n_layers, items=p.n_items, grid=p.grid, ms_per_token=med,
             ms_min=round(best, 4), kernel_us=round(kus, 1), status=err, **opts)
        del gp, p

    run_pass("default")
    if args.sweep:
        for s in (1, 2, 8, 16):
            run_pass("poll_sleep", pass_poll_sleep=s)
        experimental.set_option("pass_poll_sleep", 4)
        for w in (3, 2):
            run_pass("wgs_per_cu", pass_wgs_per_cu=w)
        experimental.set_option("pass_wgs_per_cu", 0)
        for t in (256, 512, 768, 1024, 1536):
            run_pass("target_wgs", target_wgs=t)
        experimental.set_option("target_wgs", 0)
        for gpw in (16, 32, 64):
            run_pass("groups_per_wave", groups_per_wave=gpw)
        experimental.set_option("groups_per_wave", 0)


if __name__ == "__main__":
    main()
