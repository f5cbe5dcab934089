#!/usr/bin/env python3
"""bench.py -- decode-pass benchmark of the MI355X quant_cuda path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 7b-w4-s0]

A "step" is one batch-1 decode pass over ALL quantised linears of the model named by the config
(LLaMA-7B: 32 layers x 7 = 224 QuantLinearLUT ops, each with its own synthetic weights, 3.3 GB in
total for w4 so nothing is served from the 256 MiB Infinity Cache), enqueued through the C ABI of
libsqllm_hip.so and replayed as a HIP graph.  Inputs are resident in HBM before the timed region.
`value` = tokens/s of the whole job = steps / wall time (barrier + synchronize on both sides, MAX
over ranks), attention / norms / lm_head excluded exactly as in BASELINE.md section 2.

The ONE JSON line answers every number BASELINE.json's metric asks for:
  * the headline fields: config 7b-w4-s0 (BASELINE.json configs[1]), timed for exactly --steps
    steps; the same block is repeated (--repeats, default 5) and the median / min are reported too;
  * `roofline`      every kernel dispatch of one pass bracketed by its own HIP start/stop events
                    (sqllm_profile_groups): achieved = algorithmic bytes per launch / average
                    kernel duration, against the 8 TB/s HBM peak; `per_layer_us` = the per-shape
                    kernel microseconds;
  * `sub_records`   (1-GPU default run) the other halves of the metric, measured the same way in
                    the same process: 7b-w4-s45 and 7b-w3-s45 (tokens/s, ms, roofline, per-shape
                    microseconds) and the 13B batch-{1,2,4,8} leg of configs[3] on the 13B shapes;
  * `cpu_baseline`  the reference-style CPU path on the host cores: torch dequant (codebook gather)
                    + torch.matmul, torch.matmul alone on pre-dequantised weights, and the OpenMP C
                    port of the kernels' algorithm, each on a bounded sample of the same workload.

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
import statistics
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
SUB_RECORD_CONFIGS = ("7b-w4-s45", "7b-w3-s45")  # the s45 halves of the metric
SHARED_INPUT = {"k_proj": "q_proj", "v_proj": "q_proj", "up_proj": "gate_proj"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="blocks of --steps steps timed for the median / min (the first is the contract's)")
    ap.add_argument("--config", default="7b-w4-s0", choices=sorted(CONFIGS))
    ap.add_argument("--layers", type=int, default=None, help="decoder layers to build (default: the model's)")
    ap.add_argument("--launch", default="graph", choices=["graph", "sequence"],
                    help="timed region: HIP-graph replay of the pass, or one C call enqueuing it eagerly")
    ap.add_argument("--no-fuse", action="store_true",
                    help="one launch per linear (224 per token for 7B) instead of fusing the linears of a decoder "
                         "layer that read the same input (q/k/v, gate/up) into one launch each")
    ap.add_argument("--parallel", default="auto", choices=["auto", "replicas", "pipeline", "columns"],
                    help="N > 1: independent replicas (weak scaling), layer-sharded ring pipeline, or every linear split by "
                         "output column with one all-gather per launch group (the latency split; opt-in)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-sub-records", action="store_true", help="skip the s45 / 13B-batch sub-records of the default run")
    ap.add_argument("--per-shape", action="store_true", help="print the per-shape kernel table to stderr")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------
# workload construction
# ---------------------------------------------------------------------------------------------------
def build_layers(cfg, dev, lo, hi):
    from squeezellm_amd import synth

    spec = synth.MODEL_SHAPES[cfg["model"]]
    per_layer = len(spec["linears"])
    layers = []
    for li in range(lo, hi):
        for j, (lname, K, N) in enumerate(spec["linears"]):
            lay = synth.make_layer(K, N, cfg["bits"], sparse_frac=cfg["sparse"], topX=cfg["topX"],
                                   heavy_rows=10 if cfg["sparse"] > 0 else 0, device=dev, seed=li * per_layer + j)
            lay["name"] = f"layers.{li}.{lname}"
            layers.append(lay)
    return layers


def decoder_inputs(layers, dev, gen, batch=0):
    """Activations as in the decoder layer: q/k/v read the same hidden state, gate/up the same
    post-attention state, o_proj and down_proj their own inputs (model_parse.py:53-61)."""
    import torch

    xs, last = [], {}
    for l in layers:
        lname = l["name"].rsplit(".", 1)[1]
        src = SHARED_INPUT.get(lname)
        if src is not None and src in last:
            xs.append(last[src])
        else:
            shape = (batch, l["K"]) if batch else (l["K"],)
            xs.append(torch.randn(shape, device=dev, generator=gen, dtype=torch.float16).float())
        last[lname] = xs[-1]
    ys = [torch.zeros((batch, l["N"]) if batch else (l["N"],), device=dev, dtype=torch.float32) for l in layers]
    return xs, ys


def per_shape_table(seq, layers, bytes_per_op, us):
    import numpy as np

    table = {}
    for grp, u in zip(seq.groups, us):
        key = "+".join(f"{layers[i]['K']}x{layers[i]['N']}" for i in grp)
        table.setdefault(key, []).append((u, sum(bytes_per_op[i] for i in grp)))
    out = {}
    for key, lst in table.items():
        u = np.array([a for a, _ in lst])
        b = lst[0][1]
        out[key] = {"us_mean": round(float(u.mean()), 3), "us_min": round(float(u.min()), 3), "MB": round(b / 1e6, 3),
                    "GBps": round(b / (u.mean() * 1e-6) / 1e9, 1), "hbm_frac": round(b / (u.mean() * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
    return out


def roofline_leg(seq, layers, bytes_per_op, cfg, config_name, fused):
    """achieved = algorithmic bytes per launch / mean kernel duration, each dispatch bracketed by its own
    HIP events on the launch stream (sqllm_profile_groups)."""
    pass_bytes = float(sum(bytes_per_op))
    seq.profile(reps=1)  # warm
    us = seq.profile(reps=5)  # one entry per launch (= per group of fused linears)
    avg_us = float(us.mean())
    n_launch = seq.n_groups
    achieved = pass_bytes / n_launch / (avg_us * 1e-6) / 1e9
    traffic, source = pmc_traffic_per_launch(config_name, fused)
    kt_us, kt_source = rocprof_kernel_us_per_launch(config_name, fused)
    roof = {
        "bound": "hbm",
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        # HBM bytes per launch from the PMC counters.  They cannot be read live (rocprofv3 --pmc runs,
        # one counter group per pass), so this is the COMMITTED summary of those passes over this very
        # command (tools/collect_profiles.sh), gfx950 correction applied (FETCH_SIZE x 2); null if none
        "traffic": traffic,
        "traffic_source": source,
        # the same fraction from the COMMITTED rocprofv3 --kernel-trace --stats summary of this command (profiled clocks,
        # the box of that session: it need not be this one): algorithmic bytes per launch / its mean kernel duration
        "frac_rocprof": None if kt_us is None else round(pass_bytes / n_launch / (kt_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
        "avg_kernel_us_rocprof": None if kt_us is None else round(kt_us, 3),
        "frac_rocprof_source": kt_source,
        # the launches of a pass are routed per launch shape (sqllm_capi.hip: cols_pays_batch1): the fused kernel, or the column-lane kernel for the
        # dense-only q/k/v group and down_proj (and the 3-bit gate/up)
        "kernel": f"sqllm_fused_matvec<{cfg['bits']},1> / sqllm_fused_cols<{cfg['bits']},1> by launch shape",
        "avg_kernel_us": round(avg_us, 3),
        "launches_per_step": n_launch,
        "algorithmic_bytes_per_launch": int(pass_bytes / n_launch),
        "sum_kernel_ms_per_step": round(float(us.sum()) * 1e-3, 4),
    }
    return roof, per_shape_table(seq, layers, bytes_per_op, us)


def rocprof_kernel_us_per_launch(config_name: str, fused: bool):
    """Mean duration of a launch of this config's pass in the newest committed kernel-trace summary
    (profiles/<round>_kt_<w4|w3|w4s45>.summary.txt, tools/collect_profiles.sh: total_us / calls over the sqllm kernels)."""
    names = {"7b-w4-s0": "kt_w4.summary.txt", "7b-w3-s45": "kt_w3.summary.txt", "7b-w4-s45": "kt_w4s45.summary.txt"}
    if config_name not in names or not fused:
        return None, None
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_{names[config_name]}")
        try:
            lines = open(path).read().splitlines()
        except OSError:
            continue
        calls, total = 0, 0.0
        for ln in lines:
            f = ln.split()
            if len(f) >= 8 and (f[0].startswith("sqllm::sqllm_fused_matvec") or f[0].startswith("sqllm::sqllm_fused_cols")):  # (the pass's launches: both dense kernels)
                try:
                    c, t = int(f[-6]), float(f[-5])
                except ValueError:
                    continue
                calls, total = calls + c, total + t
        if calls:
            return total / calls, f"committed kernel trace profiles/{rnd}_{names[config_name]} ({calls} dispatches; not a live measurement)"
    return None, None


def pmc_traffic_per_launch(config_name: str, fused: bool):
    """Mean HBM bytes per launch of this config from the committed rocprofv3 PMC summaries
    (profiles/<round>_pmc_fetch[_w3].summary.txt + <round>_pmc_write.summary.txt, collected by
    tools/collect_profiles.sh with the same launch grouping): FETCH_SIZE [KiB] x 2 (gfx950 tallies the
    128-B requests of wide coalesced reads at 64 B, MI355X_MICROARCH.md section HBM) + WRITE_SIZE [KiB]."""
    import re

    names = {"7b-w4-s0": ("pmc_fetch.summary.txt", "pmc_write.summary.txt"), "7b-w3-s45": ("pmc_fetch_w3.summary.txt", None),
             "7b-w4-s45": ("pmc_fetch_w4s45.summary.txt", None)}
    if config_name not in names or not fused:
        return None, None
    prof = os.path.join(ROOT, "profiles")

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

    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):  # newest committed round first
        fetch = mean_kib(f"{rnd}_{names[config_name][0]}", "FETCH_SIZE")
        if fetch is None:
            continue
        wname = names[config_name][1]
        write = mean_kib(f"{rnd}_{wname}", "WRITE_SIZE") if wname else 0.0
        return int((2.0 * fetch + (write or 0.0)) * 1024), f"committed PMC summary profiles/{rnd}_{names[config_name][0]} (not a live measurement)"
    return None, None


def time_blocks(step, sync, steps, warmup, repeats):
    """The contract's timed region (exactly `steps` steps between barrier + synchronize pairs), then
    repeats - 1 more such blocks for the spread.  Returns the list of block times in seconds."""
    for _ in range(warmup):
        step()
    out = []
    for _ in range(max(1, repeats)):
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        out.append(time.perf_counter() - t0)
    return out


# ---------------------------------------------------------------------------------------------------
# CPU baselines (rank 0 of a 1-GPU run only, bounded samples)
# ---------------------------------------------------------------------------------------------------
def cpu_baseline_c_port(layers, model_layers: int, xs=None, budget_s: float = 8.0):
    """The C port (oracle/libsqllm_oracle.so, OpenMP over the host cores) on decoder layers of the same
    workload until ~budget_s of CPU time is spent; scaled to a whole pass.  With `xs` (the GPU pass's own
    input tensors) the fp64 results are returned too: bench.py's parity spot compares the GPU pass with them."""
    import numpy as np

    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libsqllm_oracle.so"))
    lib.sqo_matvec.restype = ctypes.c_int
    lib.sqo_num_threads.restype = ctypes.c_int
    threads = lib.sqo_num_threads()
    per_layer = len(layers) // model_layers
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)

    def P(a, t):
        return None if a is None else a.ctypes.data_as(t)

    spent, done_layers, t_layers, outs = 0.0, 0, [], []
    rng = np.random.default_rng(0)
    while spent < budget_s and done_layers < model_layers and done_layers < 4:
        ops = []
        for i in range(done_layers * per_layer, (done_layers + 1) * per_layer):
            lay = layers[i]
            h = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in lay.items()}
            x = xs[i].cpu().numpy().astype(np.float32) if xs is not None else rng.normal(size=h["K"]).astype(np.float32)
            ops.append((h, np.ascontiguousarray(x), np.zeros(h["N"], np.float32), np.zeros(h["N"], np.float64)))
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
        outs.extend(o[3] for o in ops)
    best = min(t_layers)  # the first layer pays page faults / thread start-up
    rec = dict(value=round(1.0 / (best * model_layers), 3), unit="tokens/s", threads=threads,
               sample=f"{done_layers} of {model_layers} decoder layers, best layer {best * 1e3:.1f} ms, scaled x{model_layers}")
    return rec, outs


def parity_spot(seq, ys, ref_outs):
    """One pass of the very launch sequence that was timed, from zeroed outputs, against the C port's fp64
    results for the first len(ref_outs) ops (same weights, same inputs): max-norm relative error per op."""
    import numpy as np
    import torch

    for y in ys:
        y.zero_()
    seq.launch()
    torch.cuda.synchronize()
    worst = 0.0
    for y, ref in zip(ys, ref_outs):
        got = y.cpu().numpy().astype(np.float64)
        worst = max(worst, float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)))
    return {"ops_checked": len(ref_outs), "max_rel_err": float(f"{worst:.3e}"), "tolerance": 2e-5, "ok": bool(worst <= 2e-5),
            "against": "C port of the reference kernels (oracle/sqllm_oracle.c), fp64 accumulation, same weights and inputs"}


def torch_dequant_T(q, lut, bits):
    """W^T [K, N] = lookup_table[n, idx[k, n]] on the CPU with torch ops only.  4-bit: the int32 words are viewed as
    bytes, split into nibbles and used as the row index of a gather from the transposed codebook [16, N] (the
    general unpacker + an int64 flat index measured 3-4x slower); 3-bit: the product's tensor-level unpacker."""
    import torch

    from squeezellm_amd import pack

    if bits == 4:
        K8, N = q.shape
        b = q.view(torch.uint8).reshape(K8, N, 4)  # little-endian: byte j holds k = 2j (low nibble) and 2j + 1 (high)
        idx = torch.stack((b & 15, b >> 4), dim=-1).reshape(K8, N, 8).permute(0, 2, 1).reshape(K8 * 8, N)
    else:
        idx = pack.unpack_qweight(q, bits)
    return torch.gather(lut.t().contiguous(), 0, idx.to(torch.int64))


def cgroup_cpu_quota():
    """CPU quota of this container in cores (cgroup v2 cpu.max / v1 cfs quota), or None if unlimited / unreadable: the
    torch CPU paths below run on what the cgroup grants, not on the cores the box shows."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 2)
    except (OSError, ValueError):
        return None


def pick_torch_threads(probe):
    """torch's CPU ops on the GPU boxes run many times SLOWER with one thread per visible core (256) than with a
    few dozen (cgroup-limited containers: the visible cores are not all ours).  Times `probe()` at a few thread
    counts and keeps the fastest: the baseline should be the best the host does, not an artefact."""
    import torch

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    best, best_t = None, None
    for n in sorted({avail, 128, 64, 32, 16, 8}, reverse=True):
        if n > avail:
            continue
        torch.set_num_threads(n)
        probe()
        t0 = time.perf_counter()
        probe()
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best, best_t = n, dt
    torch.set_num_threads(best)
    return best, avail


def cpu_baseline_torch(layers, model_layers: int, budget_s: float = 6.0):
    """The reference-style CPU path (BASELINE.json configs[0] / north_star): the codebook gather
    W[n, k] = lookup_table[n, idx[k, n]] (indices unpacked by the product's tensor-level unpacker)
    followed by torch.matmul, and torch.matmul alone on the pre-dequantised W, fp32, all host
    cores.  Dense term only (the s0 headline); one decoder layer at a time until the budget is spent."""
    import torch

    from squeezellm_amd import pack

    cores = os.cpu_count() or 1
    per_layer = len(layers) // model_layers
    g = torch.Generator().manual_seed(0)
    _q, _lut, _x = layers[0]["qweight"].cpu(), layers[0]["lookup_table"].cpu(), torch.randn(layers[0]["K"], generator=g)
    pick_torch_threads(lambda: _x @ torch_dequant_T(_q, _lut, layers[0]["bits"]))
    t_deq, t_mm, done, spent = [], [], 0, 0.0
    while (spent < budget_s or done < 2) and done < min(model_layers, 4):
        ops = []
        for lay in layers[done * per_layer:(done + 1) * per_layer]:
            ops.append((lay["qweight"].cpu(), lay["lookup_table"].cpu(), lay["bits"], torch.randn(lay["K"], generator=g)))
        t0 = time.perf_counter()
        Ws = []
        for q, lut, bits, x in ops:
            WT = torch_dequant_T(q, lut, bits)                   # [K, N]
            Ws.append(WT)
            _ = x @ WT
        t1 = time.perf_counter()
        for _ in range(2):  # second pass: warm
            t2 = time.perf_counter()
            for WT, (_, _, _, x) in zip(Ws, ops):
                _ = x @ WT
            t3 = time.perf_counter()
        t_deq.append(t1 - t0)
        t_mm.append(t3 - t2)
        spent += t3 - t0
        done += 1
        del Ws
    return dict(cores=cores, threads=torch.get_num_threads(),
                dequant_matmul_f32=dict(value=round(1.0 / (min(t_deq) * model_layers), 4), unit="tokens/s",
                                        ms_per_decoder_layer=round(min(t_deq) * 1e3, 1)),
                matmul_only_f32=dict(value=round(1.0 / (min(t_mm) * model_layers), 3), unit="tokens/s",
                                     ms_per_decoder_layer=round(min(t_mm) * 1e3, 2)),
                sample=f"{done} of {model_layers} decoder layers ({per_layer} linears each) of the same workload, dense term, "
                       f"best layer scaled x{model_layers}")


def cpu_leg_config1(budget_s: float = 10.0):
    """BASELINE.json configs[0]: OPT-1.3B w4 dense-only, batch 1 x seq 128, the CPU torch dequant + matmul
    reference path (squeezellm/quant.py:313-383 is the batched branch such an input takes; the reference
    has no CPU kernel, so its "CPU path" is the dequantise-then-matmul restated in oracle/sqllm_oracle.py).
    One decoder layer of the model's shapes (models/opt-1.3b/config.json:13-14,20: 2048 -> 2048 x4,
    2048 -> 8192, 8192 -> 2048, with bias), x fp16 [128, 2048] -> fp32, all host cores; plus the C port."""
    import numpy as np
    import torch

    from squeezellm_amd import synth

    cores = os.cpu_count() or 1
    spec = synth.MODEL_SHAPES["opt-1.3b"]
    B = 128
    layers = [synth.make_layer(K, N, 4, bias=True, device="cpu", seed=900 + j) for j, (_, K, N) in enumerate(spec["linears"])]
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(B, l["K"], generator=g, dtype=torch.float16).float() for l in layers]
    pick_torch_threads(lambda: xs[0] @ torch_dequant_T(layers[0]["qweight"], layers[0]["lookup_table"], 4))
    t_deq, t_mm = [], []
    Ws = None
    t_start = time.perf_counter()
    for _ in range(4):
        t0 = time.perf_counter()
        Ws = []
        for l, x in zip(layers, xs):
            WT = torch_dequant_T(l["qweight"], l["lookup_table"], 4)
            Ws.append(WT)
            _ = (x @ WT).to(torch.float16) + l["bias"]
        t1 = time.perf_counter()
        for WT, l, x in zip(Ws, layers, xs):
            _ = (x @ WT).to(torch.float16) + l["bias"]
        t2 = time.perf_counter()
        t_deq.append(t1 - t0)
        t_mm.append(t2 - t1)
        if time.perf_counter() - t_start > budget_s:
            break
    # the C port over the same layer (batched entry point, OpenMP)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libsqllm_oracle.so"))
    lib.sqo_matvec.restype = ctypes.c_int
    lib.sqo_num_threads.restype = ctypes.c_int
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)
    t_c = []
    for _ in range(2):
        t0 = time.perf_counter()
        for l, x in zip(layers, xs):
            xn = np.ascontiguousarray(x.numpy())
            mul = np.zeros((B, l["N"]), np.float32)
            out = np.zeros((B, l["N"]), np.float64)
            q, lut = l["qweight"].numpy(), l["lookup_table"].numpy()
            rc = lib.sqo_matvec(4, B, xn.ctypes.data_as(fp), q.ctypes.data_as(ip), mul.ctypes.data_as(fp), lut.ctypes.data_as(fp),
                                l["K"], l["N"], None, None, None, None, None, 0, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
            assert rc == 0
        t_c.append(time.perf_counter() - t0)
    n_layers = spec["layers"]
    tok = lambda t: round(B / (t * n_layers), 2)  # noqa: E731  tokens/s of a 128-token pass over the whole model
    return {"workload": "opt-1.3b w4 s0 (dense-only), batch 1 x seq 128 (128 rows), CPU path; 1 of 24 decoder layers (6 linears, bias) timed, scaled x24",
            "cores": cores, "torch_threads": torch.get_num_threads(),
            "torch_dequant_matmul_f32": {"ms_per_decoder_layer": round(min(t_deq) * 1e3, 2), "tokens_per_s": tok(min(t_deq))},
            "torch_matmul_only_f32": {"ms_per_decoder_layer": round(min(t_mm) * 1e3, 2), "tokens_per_s": tok(min(t_mm))},
            "c_port_openmp": {"ms_per_decoder_layer": round(min(t_c) * 1e3, 2), "tokens_per_s": tok(min(t_c)), "threads": lib.sqo_num_threads()}}


def drop_in_legs(layers, xs, dev, args, sync, model_layers, per_layer):
    """What a caller gets WITHOUT the grouped-launch extension (squeezellm/llama.py:226-246
    times the forward): the same 7B pass (a) as one operator launch per linear, graph-replayed, and (b)
    through QuantLinearLUT.forward -- zeros / x.float() / operator / cast, four launches per linear
    (squeezellm/quant.py:214-223,311-312; the mirror in squeezellm_amd/quant.py issues the same four) --
    eager and graph-replayed, and (c) through the one-kernel fused linear, graph-replayed.  (b) and (c)
    on the first 8 decoder layers, scaled to the model."""
    import torch

    from squeezellm_amd import _lib, decode, quant, synth

    rec = {}
    ys = [torch.zeros(l["N"], device=dev) for l in layers]
    seq = decode.OpSequence(layers, xs, ys, fuse_shared_input=False)
    g = seq.graph(warmup=1)
    blocks = time_blocks(g.replay, sync, 20, 3, 3)
    ms = statistics.median(blocks) / 20 * 1e3
    rec["op_per_linear_graph"] = {"launches_per_token": seq.n_groups, "ms_per_token": round(ms, 4), "tokens_per_s": round(1e3 / ms, 1)}
    del seq, g
    # the same 224 operator calls EAGER through the Python names of quant_cuda (what unchanged quant.py issues, minus its
    # three torch launches per linear): host-bound or kernel-bound, whichever is slower
    from squeezellm_amd import quant_cuda as qc

    fn = getattr(qc, CONFIGS["7b-w4-s0"]["op"])
    calls = [(x, l["qweight"], y, l["lookup_table"]) for l, x, y in zip(layers, xs, ys)]

    def eager_ops():
        for c in calls:
            fn(*c)

    blocks = time_blocks(eager_ops, sync, 5, 2, 3)
    ms = statistics.median(blocks) / 5 * 1e3
    rec["op_per_linear_eager"] = {"launches_per_token": len(calls), "ms_per_token": round(ms, 4), "tokens_per_s": round(1e3 / ms, 1)}
    # what one call costs on the host: the queue is never drained inside the loop, so wall / calls = host time per call
    tiny = synth.make_layer(256, 256, 4, device=dev, seed=7)
    tx, ty = torch.randn(256, device=dev), torch.zeros(256, device=dev)

    def host_us(f, n=4000):
        for _ in range(200):
            f()
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        dt = time.perf_counter() - t0
        sync()
        return round(dt / n * 1e6, 2)

    lib = _lib.load()
    op = _lib.SqllmOp(bits=4, batch=0, K=256, N=256)
    op.vec, op.qweight, op.mul, op.lookup_table = tx.data_ptr(), tiny["qweight"].data_ptr(), ty.data_ptr(), tiny["lookup_table"].data_ptr()
    ref, stream = ctypes.byref(op), torch.cuda.current_stream(dev).cuda_stream
    rec["host_cost_us"] = {
        "quant_cuda_call": host_us(lambda: fn(tx, tiny["qweight"], ty, tiny["lookup_table"])),
        "c_abi_launch_premarshalled": host_us(lambda: lib.sqllm_launch(ref, stream)),
        "torch_zero_": host_us(lambda: ty.zero_()),
        "torch_zeros": host_us(lambda: torch.zeros(256, device=dev)),
        "note": "eager host microseconds per call on this box (256 x 256 layer: the kernel is shorter than the call); quant.py's forward adds "
                "three torch launches (zeros / x.float() / y.to) to every operator call",
    }
    del ys, calls
    n_dec = min(8, model_layers)
    sub = layers[:n_dec * per_layer]
    scale = model_layers / n_dec
    x16 = {}
    xs16 = []
    for x in xs[:len(sub)]:
        if id(x) not in x16:
            x16[id(x)] = x.half().reshape(1, 1, -1)
        xs16.append(x16[id(x)])
    mods = [quant.QuantLinearLUT.from_operands(l) for l in sub]

    def run(ms_):
        with torch.no_grad():
            for m, x in zip(ms_, xs16):
                m(x)

    def timed(fn, reps):
        """best of three blocks of `reps` (the eager legs are host-bound: a neighbour on the host's cores can slow one
        block several-fold -- seen: 72 ms against 5 ms for the same loop)"""
        for _ in range(2):
            fn()
        best = None
        for _ in range(3):
            sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            sync()
            dt = (time.perf_counter() - t0) / reps * 1e3 * scale
            best = dt if best is None else min(best, dt)
        return best

    def capture(fn):
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        return gr

    t = timed(lambda: run(mods), 5)
    rec["forward_eager"] = {"launches_per_linear": 4, "ms_per_token": round(t, 4), "tokens_per_s": round(1e3 / t, 1)}
    gr = capture(lambda: run(mods))
    t = timed(gr.replay, 20)
    rec["forward_graph"] = {"launches_per_linear": 4, "ms_per_token": round(t, 4), "tokens_per_s": round(1e3 / t, 1)}
    del gr
    for m in mods:
        m.__class__ = quant.QuantLinearLUTFused
    run(mods)  # (first call: CSR check, workspace)
    t = timed(lambda: run(mods), 5)
    rec["fused_linear_eager"] = {"launches_per_linear": 1, "ms_per_token": round(t, 4), "tokens_per_s": round(1e3 / t, 1)}
    gr = capture(lambda: run(mods))
    t = timed(gr.replay, 20)
    rec["fused_linear_graph"] = {"launches_per_linear": 1, "ms_per_token": round(t, 4), "tokens_per_s": round(1e3 / t, 1)}
    # the fused linear's KERNEL against the operator's, like for like: both as pre-marshalled C-ABI sequences of one launch
    # per linear (no module, no torch op in the graph), the same decoder layers
    ys16 = [torch.empty(l["N"], device=dev, dtype=torch.float16) for l in sub]
    x16f = [x.reshape(-1) for x in xs16]
    lin_seq = decode.OpSequence(sub, x16f, ys16, fuse_shared_input=False, linear=True)
    g2 = lin_seq.graph(warmup=1)
    t_lin = timed(g2.replay, 20)
    ys32 = [torch.zeros(l["N"], device=dev) for l in sub]
    op_seq = decode.OpSequence(sub, xs[:len(sub)], ys32, fuse_shared_input=False)
    g3 = op_seq.graph(warmup=1)
    t_op = timed(g3.replay, 20)
    rec["fused_linear_kernel_vs_operator"] = {"fused_linear_sequence_ms": round(t_lin, 4), "operator_sequence_ms": round(t_op, 4),
                                              "overhead_pct": round((t_lin / t_op - 1) * 100, 1),
                                              "note": "C-ABI sequences, one launch per linear each, graph replay; fused_linear_graph (the same "
                                                      "kernels captured through the torch module) should read within a few percent of the first"}
    del g2, g3, lin_seq, op_seq, ys16, ys32
    rec["note"] = (f"forward_* and fused_linear_graph: {n_dec} of {model_layers} decoder layers timed, scaled; fp16 activations; "
                   "the headline `value` needs the grouped-launch entry point (sqllm_launch_groups / OpSequence), "
                   "which squeezellm/quant.py unchanged does not call")
    del gr, mods
    torch.cuda.empty_cache()
    return rec


# ---------------------------------------------------------------------------------------------------
# one replica-mode measurement of a config (used for the headline and for the sub-records)
# ---------------------------------------------------------------------------------------------------
def measure_replica(config_name, dev, rank, args, steps, warmup, repeats, sync, want_roofline):
    import torch

    from squeezellm_amd import decode, synth

    cfg = CONFIGS[config_name]
    spec = synth.MODEL_SHAPES[cfg["model"]]
    model_layers = spec["layers"] if args.layers is None else args.layers
    layers = build_layers(cfg, dev, 0, model_layers)
    bytes_per_op = [synth.layer_bytes(l, 1) for l in layers]
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    xs, ys = decoder_inputs(layers, dev, gen)
    seq = decode.OpSequence(layers, xs, ys, fuse_shared_input=not args.no_fuse)
    if args.launch == "graph":
        graph = seq.graph(warmup=1)
        step = graph.replay
    else:
        step = seq.launch
    blocks = time_blocks(step, sync, steps, warmup, repeats)
    out = dict(cfg=cfg, model_layers=model_layers, per_layer=len(spec["linears"]), layers=layers, seq=seq, xs=xs, ys=ys,
               bytes_per_op=bytes_per_op, blocks=blocks, roofline=None, per_layer_us=None)
    if want_roofline:
        out["roofline"], out["per_layer_us"] = roofline_leg(seq, layers, bytes_per_op, cfg, config_name, not args.no_fuse)
    return out


def batch_leg_13b(dev, args, sync):
    """BASELINE.json configs[3]: LLaMA-13B shapes, w4 + 0.45 % sparse + top-10, batch 1..8 through the
    operator the reference would pick (batch 1: the matvec op, quant.py:212; 2..8: the *_batched op),
    over 4 decoder layers of distinct weights (0.7 GB), graph replay, plus per-launch kernel times."""
    import torch

    from squeezellm_amd import decode, synth

    cfg = CONFIGS["13b-w4-s45"]
    n_layers = 4 if args.layers is None else min(args.layers, 4)
    layers = build_layers(cfg, dev, 0, n_layers)
    gen = torch.Generator(device=dev).manual_seed(4321)
    rec = {}
    for B in (1, 2, 4, 8, 16):  # (16 rows: beyond configs[3]; the split matrix-core kernel, one launch per group)
        xs, ys = decoder_inputs(layers, dev, gen, batch=0 if B == 1 else B)
        seq = decode.OpSequence(layers, xs, ys, batched=B > 1, fuse_shared_input=not args.no_fuse)
        graph = seq.graph(warmup=1)
        blocks = time_blocks(graph.replay, sync, 20, 3, 3)
        ms_layer = statistics.median(blocks) / 20 / n_layers * 1e3
        bytes_per_op = [synth.layer_bytes(l, B) for l in layers]
        seq.profile(reps=1)
        us = seq.profile(reps=3)
        rec[f"batch{B}"] = {"ms_per_decoder_layer": round(ms_layer, 4), "launches_per_decoder_layer": seq.n_groups // n_layers,
                            "hbm_frac_wall": round(sum(bytes_per_op) / n_layers / (ms_layer * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "per_layer_us": per_shape_table(seq, layers, bytes_per_op, us)}
        del xs, ys, seq, graph
    # B = 2048: the batch the reference's perplexity evaluation feeds the *_batched ops (and what a prompt looks like) --
    # one decoder layer, eager launches (the wide matrix-core kernel splits vec into stream-ordered scratch), events per op
    B = 2048
    one = layers[:7]
    xs, ys = decoder_inputs(one, dev, gen, batch=B)
    seq = decode.OpSequence(one, xs, ys, batched=True, fuse_shared_input=not args.no_fuse)
    seq.launch()
    sync()
    walls = []
    for _ in range(3):
        t0 = time.perf_counter()
        seq.launch()
        sync()
        walls.append(time.perf_counter() - t0)
    seq.profile(reps=1)
    us = seq.profile(reps=3)
    flops = sum(2.0 * B * l["K"] * l["N"] for l in one)
    rec["rows2048"] = {"ms_per_decoder_layer": round(min(walls) * 1e3, 4), "dense_TFLOPs_wall": round(flops / min(walls) / 1e12, 1),
                       "vec": "fp16-born (as QuantLinearLUT.forward passes it)",
                       "per_layer_us": per_shape_table(seq, one, [synth.layer_bytes(l, B) for l in one], us)}
    del xs, ys, seq
    del layers
    torch.cuda.empty_cache()
    return {"workload": f"llama-13b w4 s45 (0.45% CSR outliers + top-10 rows), {n_layers} decoder layers x 7 linears of distinct weights, "
                        "batch 1 = matvec op, batch 2/4/8/16 = *_batched op, HIP-graph replay; rows2048: one layer at 2048 rows, eager", **rec}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from squeezellm_amd import sharding, synth

    cfg = CONFIGS[args.config]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SQLLM_BENCH_FORCE_DIST=1 with ONE rank and --gpus N > 1: a dry run of the N-GPU command's code path on one GPU (the
    # routing --gpus N takes -- 65B: layer-sharded ring pipeline -- under a real one-rank RCCL group), so that the driver's
    # first N-GPU run is not also the first run of that path; the line says so and reports n_gpus = 1
    dry_world = args.gpus if (os.environ.get("SQLLM_BENCH_FORCE_DIST") == "1" and world == 1 and args.gpus > 1) else 0
    if world != args.gpus and not dry_world:
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
        mode = "pipeline" if ((world > 1 or dry_world) and args.config.startswith("65b")) else "replicas"
    if world == 1 and not dry_world and args.parallel not in ("pipeline", "columns"):
        mode = "replicas"  # (an explicit --parallel pipeline / columns on one GPU runs the degenerate one-rank form)

    def sync():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    single = world == 1 and rank == 0 and mode == "replicas"
    extra = {}
    if mode == "replicas":
        m = measure_replica(args.config, dev, rank, args, args.steps, args.warmup, args.repeats, sync,
                            want_roofline=single and not args.no_roofline)
        blocks, seq, layers = m["blocks"], m["seq"], m["layers"]
        bytes_per_op = m["bytes_per_op"]
        tokens_per_step = 1
    elif mode == "columns":
        # every rank builds the same full operands (same seeds), keeps its column slices: 1 / world of every layer
        layers = build_layers(cfg, dev, 0, model_layers)
        bytes_per_op = [synth.layer_bytes(l, 1) for l in layers]
        gen = torch.Generator(device=dev).manual_seed(1234)  # the same activations on every rank
        xs, _ = decoder_inputs(layers, dev, gen)
        cpass = sharding.ColumnParallelPass(layers, xs, rank=rank, world_size=world, device=dev)
        del layers
        torch.cuda.empty_cache()
        layers = []
        blocks = time_blocks(cpass.step, sync, args.steps, args.warmup, args.repeats)
        tokens_per_step = 1
        seq, m = None, None
        extra["column_parallel"] = {"launch_groups": len(cpass.groups), "collectives_per_token": len(cpass.groups) if world > 1 else 0,
                                    "graph_captured": cpass.graph is not None}
    else:
        lo, hi = sharding.partition_layers(model_layers, world)[rank]
        layers = build_layers(cfg, dev, lo, hi)
        bytes_per_op = [synth.layer_bytes(l, 1) for l in layers]
        # the whole tick (stage kernels + all-gather + hand-over copy) as ONE captured graph where RCCL allows it;
        # otherwise the stage replays its own graph and the collective runs eagerly
        stage = sharding.DecodeStage(layers, hidden, dev, seed=rank, graph=False)
        g = torch.Generator(device=dev).manual_seed(99 + rank)
        h0 = torch.randn(hidden, device=dev, generator=g, dtype=torch.float16)
        pipe = sharding.RingPipeline(stage, hidden, rank=rank, world_size=world, device=dev, h0=h0)
        # (across several ranks the capture of the collective is opt-in -- SQLLM_PIPELINE_CAPTURE=1: it has only been
        # exercised on one rank, and a capture refused on SOME ranks would desynchronise the ring)
        whole_tick_captured = (world == 1 or os.environ.get("SQLLM_PIPELINE_CAPTURE") == "1") and pipe.capture()
        if not whole_tick_captured:
            stage = sharding.DecodeStage(layers, hidden, dev, seed=rank, graph=True)
            pipe = sharding.RingPipeline(stage, hidden, rank=rank, world_size=world, device=dev, h0=h0)
        extra["pipeline_tick_captured"] = bool(whole_tick_captured)
        step = lambda: pipe.run(world)  # noqa: E731  W ticks = every sequence advances one token
        blocks = time_blocks(step, sync, args.steps, args.warmup, args.repeats)
        tokens_per_step = world
        seq, m = None, None
        extra["pipeline_tick_us"] = round(statistics.median(blocks) / args.steps / world * 1e6, 2)

    if use_dist:  # every block: MAX over ranks
        t = torch.tensor(blocks, device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        blocks = [float(v) for v in t.tolist()]
    elapsed = blocks[0]  # the contract's timed region: exactly --steps steps
    ms_per_step = elapsed / args.steps * 1e3
    # replicas: every rank produced `steps` tokens of its own stream; pipeline: the job as a whole
    # produced `world` tokens per step
    per_job = world if mode == "replicas" else tokens_per_step
    value = per_job * args.steps / elapsed
    ms_all = [b / args.steps * 1e3 for b in blocks]

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
        "scaling": "strong" if mode in ("pipeline", "columns") else "weak",
        "vs_baseline": None,  # BASELINE.md holds no published number for this metric
        "dtype": "f32",  # fp32 LUT values, fp32 activations, fp32 accumulate (3/4-bit integer indices)
        "data": "synthetic",
        "repeats": {"blocks_of_steps": len(ms_all), "ms_per_step": [round(v, 4) for v in ms_all],
                    "ms_per_step_median": round(statistics.median(ms_all), 4), "ms_per_step_min": round(min(ms_all), 4),
                    "value_median": round(per_job * 1e3 / statistics.median(ms_all), 2), "value_best": round(per_job * 1e3 / min(ms_all), 2)},
        "config": {
            "workload": f"{cfg['model']} w{cfg['bits']} " + (f"s{int(round(cfg['sparse'] * 10000))} (0.45% CSR outliers + top-{cfg['topX']} rows)" if cfg["sparse"] else "s0 (dense-only)")
                        + f", batch=1 decode, {model_layers} layers x {per_layer} linears, op {cfg['op']}",
            "config_name": args.config,
            "launch": (args.launch + (", one launch per linear" if args.no_fuse else
                                      ", linears sharing an input (q/k/v, gate/up) fused into one launch each"))
            if mode == "replicas" else ("column-sharded linears, one all-gather of the mul slices per launch group" if mode == "columns"
                                        else "sequence + ring all-gather"),
            "launches_per_token": seq.n_groups if mode == "replicas" else None,
            "parallelism": "single GPU" if (world == 1 and not dry_world) else ((
                f"dp{dry_world or world}: independent token streams, one full model replica per GPU, no data-path collective"
                if mode == "replicas" else
                (f"tp{dry_world or world} by output column: every GPU holds 1/{dry_world or world} of every linear, RCCL all-gather of the mul slices per launch group"
                 if mode == "columns" else
                 f"pp{dry_world or world}: layer-sharded ring pipeline, RCCL all-gather of the hidden state per tick"))
                + (" -- DRY RUN of that path on ONE rank (SQLLM_BENCH_FORCE_DIST=1)" if dry_world else "")),
            "world_size": world,
            "rccl_ranks": dist.get_world_size() if use_dist else 0,  # 0: no process group (single process)
            "ops_per_token": model_layers * per_layer,
            "algorithmic_bytes_per_token": int(sum(bytes_per_op)) if mode in ("replicas", "columns") else None,
        },
    }
    result.update(extra)

    if single:
        pass_bytes = float(sum(bytes_per_op))
        result["hbm_frac_wall"] = round(pass_bytes / (statistics.median(ms_all) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if m["roofline"] is not None:
            result["roofline"] = m["roofline"]
            result["per_layer_us"] = m["per_layer_us"]
            if args.per_shape:
                print(json.dumps(m["per_layer_us"], indent=1), file=sys.stderr)
        headline_default = args.config == "7b-w4-s0" and args.layers is None and not args.no_fuse and args.launch == "graph"
        # (the drop-in legs run first: the CPU legs leave 128-256 OpenMP / torch worker threads spinning for a while,
        # which slows the eager Python path of forward_eager several-fold)
        if headline_default and not args.no_sub_records:
            result["drop_in"] = drop_in_legs(layers, m["xs"], dev, args, sync, model_layers, per_layer)
        if not args.no_cpu_baseline:
            # value = the C port of the kernels' algorithm on the host cores (kind "port"); the reference-style
            # torch paths BASELINE.json names (dequant + matmul, matmul alone) ride along in `paths`
            cp, ref_outs = cpu_baseline_c_port(layers, model_layers, xs=m["xs"])  # first: torch's OpenMP pool would spin beside it
            # the C port has just computed the first decoder layers of the very pass that was timed: check it
            result["parity_spot"] = parity_spot(seq, m["ys"], ref_outs)
            tb = cpu_baseline_torch(layers, model_layers)
            result["cpu_baseline"] = {
                "value": cp["value"], "unit": "tokens/s", "cores": cp["threads"], "kind": "port",
                "sample": "C port of the kernels (oracle/sqllm_oracle.c, OpenMP), " + cp["sample"],
                "host_cores": tb["cores"],
                "cgroup_cpu_quota_cores": cgroup_cpu_quota(),  # None = no quota readable; the torch paths picked their own best thread count
                "affinity_cores": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
                "paths": {"c_port_openmp": cp,
                          "torch_dequant_matmul_f32": dict(tb["dequant_matmul_f32"], sample=tb["sample"], threads=tb["threads"]),
                          "torch_matmul_only_f32": dict(tb["matmul_only_f32"], sample=tb["sample"], threads=tb["threads"])},
            }
            if headline_default:
                result["cpu_baseline"]["config1_opt1.3b_seq128"] = cpu_leg_config1()
        # release the headline model before the sub-records build theirs
        del m, seq, layers
        torch.cuda.empty_cache()
        if headline_default and not args.no_sub_records:
            sub = {}
            for name in SUB_RECORD_CONFIGS:
                s = measure_replica(name, dev, rank, args, 20, 3, 3, sync, want_roofline=not args.no_roofline)
                ms = [b / 20 * 1e3 for b in s["blocks"]]
                pb = float(sum(s["bytes_per_op"]))
                sub[name] = {
                    "workload": f"{s['cfg']['model']} w{s['cfg']['bits']} s45 (0.45% CSR outliers + top-10 rows), batch=1 decode, op {s['cfg']['op']}",
                    "value": round(1e3 / statistics.median(ms), 2), "unit": "tokens/s", "steps": 20, "blocks_of_steps": len(ms),
                    "ms_per_step_median": round(statistics.median(ms), 4), "ms_per_step_min": round(min(ms), 4),
                    "hbm_frac_wall": round(pb / (statistics.median(ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "algorithmic_bytes_per_token": int(pb), "launches_per_token": s["seq"].n_groups,
                    "roofline": s["roofline"], "per_layer_us": s["per_layer_us"],
                }
                del s
                torch.cuda.empty_cache()
            sub["13b-w4-s45-batched"] = batch_leg_13b(dev, args, sync)
            result["sub_records"] = sub

    if rank == 0:
        print(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
