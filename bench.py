#!/usr/bin/env python3
"""bench.py -- decode-pass benchmark of the MI355X quant_cuda path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 7b-w4-s0]

A "step" is one batch-1 decode pass over ALL quantised linears of the model named by the config
(LLaMA-7B: 32 layers x 7 = 224 QuantLinearLUT ops, each with its own synthetic weights, 3.3 GB in
total for w4 so nothing is served from the 256 MiB Infinity Cache), enqueued through the C ABI of
libsqllm_hip.so and replayed as a HIP graph.  Inputs are resident in HBM before the timed region.
`value` = tokens/s of the whole job = steps / wall time (barrier + synchronize on both sides, MAX
over ranks), attention / norms / lm_head excluded exactly as in BASELINE.md section 2.

Legs after the timed region (rank 0 of a 1-GPU run only):
  * roofline     every kernel dispatch of one pass bracketed by its own HIP start/stop events
                 (sqllm_profile_sequence): achieved = algorithmic bytes per launch / average kernel
                 duration, against the 8 TB/s HBM peak;
  * cpu_baseline the C port of the same algorithm (oracle/sqllm_oracle.c, OpenMP) timed on the
                 host cores over a bounded sample of the same workload.

N > 1 (launched by torch.distributed.run, one rank per GPU, backend nccl = RCCL), two modes:
  * replicas (default when the model fits one GPU, i.e. every config but 65B): the unit of work is
    an independent token stream; every rank holds the whole model and decodes its own stream, no
    data-path collective -- weak scaling, `value` = N x per-rank tokens/s over the slowest rank;
  * pipeline (--parallel pipeline; default for 65b-*): layers sharded over the ranks, N sequences
    decoded around the ring with one small RCCL all-gather of the hidden state per tick
    (squeezellm_amd/sharding.py); a step = N ticks = one token for each of the N sequences.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

CONFIGS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "7b-w4-s0": dict(model="llama-7b", bits=4, sparse=0.0, topX=0, op="vecquant4matmul_nuq_perchannel"),
    # configs[2]
    "7b-w3-s45": dict(model="llama-7b", bits=3, sparse=0.0045, topX=10, op="vecquant3matmul_spmv_hybrid_nuq_perchannel"),
    "7b-w4-s45": dict(model="llama-7b", bits=4, sparse=0.0045, topX=10, op="vecquant4matmul_spmv_hybrid_nuq_perchannel"),
    "7b-w3-s0": dict(model="llama-7b", bits=3, sparse=0.0, topX=0, op="vecquant3matmul_nuq_perchannel"),
    # configs[3] (batch 1 leg) and configs[4] (single-GPU leg)
    "13b-w4-s45": dict(model="llama-13b", bits=4, sparse=0.0045, topX=10, op="vecquant4matmul_spmv_hybrid_nuq_perchannel"),
    "65b-w3-s45": dict(model="llama-65b", bits=3, sparse=0.0045, topX=10, op="vecquant3matmul_spmv_hybrid_nuq_perchannel"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="7b-w4-s0", choices=sorted(CONFIGS))
    ap.add_argument("--layers", type=int, default=None, help="decoder layers to build (default: the model's)")
    ap.add_argument("--launch", default="graph", choices=["graph", "sequence"],
                    help="timed region: HIP-graph replay of the pass, or one C call enqueuing it eagerly")
    ap.add_argument("--no-fuse", action="store_true",
                    help="one launch per linear (224 per token for 7B) instead of fusing the linears of a decoder "
                         "layer that read the same input (q/k/v, gate/up) into one launch each")
    ap.add_argument("--parallel", default="auto", choices=["auto", "replicas", "pipeline"],
                    help="N > 1: independent replicas (weak scaling) or layer-sharded ring pipeline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--per-shape", action="store_true", help="print the per-shape kernel table to stderr")
    return ap.parse_args()


def cpu_baseline(layers, model_layers: int, budget_s: float = 12.0):
    """Time the C port (oracle/libsqllm_oracle.so, OpenMP over the host cores) on decoder layers of
    the same workload until ~budget_s of CPU time is spent; scale to a whole pass."""
    import numpy as np

    path = os.path.join(ROOT, "oracle", "libsqllm_oracle.so")
    lib = ctypes.CDLL(path)
    lib.sqo_matvec.restype = ctypes.c_int
    lib.sqo_num_threads.restype = ctypes.c_int
    threads = lib.sqo_num_threads()
    per_layer = len(layers) // model_layers
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)

    def P(a, t):
        return None if a is None else a.ctypes.data_as(t)

    spent, done_layers, t_layers = 0.0, 0, []
    rng = np.random.default_rng(0)
    while spent < budget_s and done_layers < model_layers and done_layers < 4:
        ops = []
        for lay in layers[done_layers * per_layer:(done_layers + 1) * per_layer]:
            h = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in lay.items()}
            x = rng.normal(size=h["K"]).astype(np.float32)
            ops.append((h, x, np.zeros(h["N"], np.float32), np.zeros(h["N"], np.float64)))
        t0 = time.perf_counter()
        for h, x, mul, out in ops:
            topX = 0 if h["full_rows"] is None else h["full_rows"].shape[1]
            rc = lib.sqo_matvec(h["bits"], 0, P(x, fp), P(h["qweight"], ip), P(mul, fp), P(h["lookup_table"], fp),
                                h["K"], h["N"], P(h["rows"], ip), P(h["cols"], ip), P(h["vals"], fp),
                                P(h["full_rows"], fp), P(h["full_row_indices"], ip), topX,
                                out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
            assert rc == 0
        dt = time.perf_counter() - t0
        t_layers.append(dt)
        spent += dt
        done_layers += 1
    best = min(t_layers)  # the first layer pays page faults / thread start-up
    return dict(value=1.0 / (best * model_layers), unit="tokens/s", cores=threads, kind="port",
                sample=f"{done_layers} of {model_layers} decoder layers ({per_layer} linears each) of the same workload, "
                       f"C port oracle/sqllm_oracle.c with OpenMP, best layer {best * 1e3:.1f} ms, scaled x{model_layers}")


def pmc_traffic_per_launch(config_name: str, fused: bool):
    """Mean HBM bytes per launch of this config from the committed rocprofv3 PMC summaries
    (profiles/r01_pmc_fetch[_w3].summary.txt + r01_pmc_write.summary.txt, collected by
    tools/collect_profiles.sh with the same launch grouping): FETCH_SIZE [KiB] x 2 (gfx950 tallies the
    128-B requests of wide coalesced reads at 64 B, MI355X_MICROARCH.md section HBM) + WRITE_SIZE [KiB]."""
    import re

    names = {"7b-w4-s0": ("r01_pmc_fetch.summary.txt", "r01_pmc_write.summary.txt"),
             "7b-w3-s45": ("r01_pmc_fetch_w3.summary.txt", None)}
    if config_name not in names or not fused:
        return None
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")

    def mean_kib(fname, counter):
        try:
            txt = open(os.path.join(prof, fname)).read()
        except OSError:
            return None
        rows = re.findall(rf"grid=\d+\s+{counter}\s+n=(\d+)\s+mean=\s*([\d,\.]+)", txt)
        if not rows:
            return None
        n = sum(int(a) for a, _ in rows)
        return sum(int(a) * float(b.replace(",", "")) for a, b in rows) / n

    fetch = mean_kib(names[config_name][0], "FETCH_SIZE")
    if fetch is None:
        return None
    write = mean_kib(names[config_name][1], "WRITE_SIZE") if names[config_name][1] else 0.0
    return int((2.0 * fetch + (write or 0.0)) * 1024)


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from squeezellm_amd import decode, sharding, synth

    cfg = CONFIGS[args.config]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SQLLM_BENCH_FORCE_DIST=1 initialises RCCL even for one rank (smoke test of the N > 1 plumbing)
    use_dist = world > 1 or os.environ.get("SQLLM_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    spec = synth.MODEL_SHAPES[cfg["model"]]
    model_layers = spec["layers"] if args.layers is None else args.layers
    per_layer = len(spec["linears"])
    hidden = spec["linears"][0][1]
    mode = args.parallel
    if mode == "auto":
        mode = "pipeline" if (world > 1 and args.config.startswith("65b")) else "replicas"
    if world == 1 and args.parallel != "pipeline":
        mode = "replicas"  # (an explicit --parallel pipeline on one GPU runs a ring of one stage)
    lo, hi = sharding.partition_layers(model_layers, world)[rank] if mode == "pipeline" else (0, model_layers)

    # ---- build this rank's layers (distinct weights per linear), resident in HBM ----
    layers = []
    for li in range(lo, hi):
        for j, (lname, K, N) in enumerate(spec["linears"]):
            lay = synth.make_layer(K, N, cfg["bits"], sparse_frac=cfg["sparse"], topX=cfg["topX"],
                                   heavy_rows=10 if cfg["sparse"] > 0 else 0, device=dev, seed=li * per_layer + j)
            lay["name"] = f"layers.{li}.{lname}"
            layers.append(lay)
    bytes_per_op = [synth.layer_bytes(l, 1) for l in layers]

    if mode == "replicas":
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        # activations as in the decoder layer: q/k/v read the same hidden state, gate/up the same
        # post-attention state, o_proj and down_proj their own inputs (model_parse.py:53-61)
        shared = {"k_proj": "q_proj", "v_proj": "q_proj", "up_proj": "gate_proj"}
        xs, last = [], {}
        for l in layers:
            lname = l["name"].rsplit(".", 1)[1]
            src = shared.get(lname)
            if src is not None and src in last:
                xs.append(last[src])
            else:
                xs.append(torch.randn(l["K"], device=dev, generator=g, dtype=torch.float16).float())
            last[lname] = xs[-1]
        ys = [torch.zeros(l["N"], device=dev, dtype=torch.float32) for l in layers]
        seq = decode.OpSequence(layers, xs, ys, fuse_shared_input=not args.no_fuse)
        if args.launch == "graph":
            graph = seq.graph(warmup=1)
            step = graph.replay
        else:
            step = seq.launch
        tokens_per_step = 1
    else:
        stage = sharding.DecodeStage(layers, hidden, dev, seed=rank)
        g = torch.Generator(device=dev).manual_seed(99 + rank)
        h0 = torch.randn(hidden, device=dev, generator=g, dtype=torch.float16)
        pipe = sharding.RingPipeline(stage, hidden, rank=rank, world_size=world, device=dev, h0=h0)
        step = lambda: pipe.run(world)  # noqa: E731  W ticks = every sequence advances one token
        tokens_per_step = world

    def sync():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    # replicas: every rank produced `steps` tokens of its own stream; pipeline: the job as a whole
    # produced `world` tokens per step
    value = (world if mode == "replicas" else tokens_per_step) * args.steps / elapsed

    result = {
        "metric": "LLaMA-7B-shaped quantised-linear decode throughput (batch 1, all QuantLinearLUT matvecs of the model per token)"
        if cfg["model"] == "llama-7b" else f"{cfg['model']}-shaped quantised-linear decode throughput (batch 1)",
        "value": round(value, 2),
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong" if mode == "pipeline" else "weak",
        "vs_baseline": None,  # BASELINE.md holds no published number for this metric
        "dtype": "f32",  # fp32 LUT values, fp32 activations, fp32 accumulate (3/4-bit integer indices)
        "data": "synthetic",
        "config": {
            "workload": f"{cfg['model']} w{cfg['bits']} " + (f"s{int(round(cfg['sparse'] * 10000))} (0.45% CSR outliers + top-{cfg['topX']} rows)" if cfg["sparse"] else "s0 (dense-only)")
                        + f", batch=1 decode, {model_layers} layers x {per_layer} linears, op {cfg['op']}",
            "config_name": args.config,
            "launch": (args.launch + (", one launch per linear" if args.no_fuse else
                                      ", linears sharing an input (q/k/v, gate/up) fused into one launch each"))
            if mode == "replicas" else "sequence + ring all-gather",
            "launches_per_token": seq.n_groups if mode == "replicas" else None,
            "parallelism": "single GPU" if world == 1 else (
                f"dp{world}: independent token streams, one full model replica per GPU, no data-path collective"
                if mode == "replicas" else
                f"pp{world}: layer-sharded ring pipeline, RCCL all-gather of the hidden state per tick"),
            "ops_per_token": model_layers * per_layer,
            "algorithmic_bytes_per_token": int(sum(bytes_per_op)) if mode == "replicas" else None,
        },
    }

    if world == 1 and rank == 0 and mode == "replicas":
        pass_bytes = float(sum(bytes_per_op))
        result["hbm_frac_wall"] = round(pass_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if not args.no_roofline:
            import numpy as np

            seq.profile(reps=1)  # warm
            us = seq.profile(reps=5)  # one entry per launch (= per group of fused linears)
            avg_us = float(us.mean())
            n_launch = seq.n_groups
            achieved = pass_bytes / n_launch / (avg_us * 1e-6) / 1e9
            result["roofline"] = {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                # HBM bytes per launch from the PMC counters: they cannot be read live (rocprofv3 --pmc
                # passes, one counter group each), so this is the committed summary of those passes
                # for this config, gfx950 correction applied (FETCH_SIZE x 2); null if none is committed
                "traffic": pmc_traffic_per_launch(args.config, not args.no_fuse),
                "kernel": f"sqllm_fused_matvec<{cfg['bits']},1>",
                "avg_kernel_us": round(avg_us, 3),
                "launches_per_step": n_launch,
                "algorithmic_bytes_per_launch": int(pass_bytes / n_launch),
                "sum_kernel_ms_per_step": round(float(us.sum()) * 1e-3, 4),
            }
            # per-layer matvec microseconds by shape (the other half of BASELINE.json's metric)
            table = {}
            for grp, u in zip(seq.groups, us):
                key = "+".join(f"{layers[i]['K']}x{layers[i]['N']}" for i in grp)
                table.setdefault(key, []).append((u, sum(bytes_per_op[i] for i in grp)))
            per_shape = {}
            for key, lst in table.items():
                u = np.array([a for a, _ in lst])
                b = lst[0][1]
                per_shape[key] = {"us_mean": round(float(u.mean()), 3), "us_min": round(float(u.min()), 3),
                                  "MB": round(b / 1e6, 3), "GBps": round(b / (u.mean() * 1e-6) / 1e9, 1),
                                  "hbm_frac": round(b / (u.mean() * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
            result["per_layer_us"] = per_shape
            if args.per_shape:
                print(json.dumps(per_shape, indent=1), file=sys.stderr)
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(layers, model_layers)

    if rank == 0:
        print(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
