#!/usr/bin/env python3
"""Where do the microseconds of ONE fused small launch go?  (measurement library: SQLLM_LIB=squeezellm_amd/libsqllm_hip_ablation.so)
Every dense workgroup of sqllm_fused_small_split stamps the 100 MHz clock: 0 entry, 1 codebooks staged, 2 wave 0 done decoding
(7: the last wave), 3 CSR share staged, 4 wave 0's groups walked, 5 walk done (all waves), 6 atomics issued.

    SQLLM_LIB=squeezellm_amd/libsqllm_hip_ablation.so python tools/experiments/small_split_timeline.py --rows 16 [--group gate_up]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from squeezellm_amd import _lib, decode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=16)
    ap.add_argument("--no-ws", action="store_true")
    ap.add_argument("--dense-only", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    lib.sqllm_debug_set_timeline.argtypes = [ctypes.c_void_p]
    lib.sqllm_debug_set_timeline.restype = None
    cfg = dict(bench.CONFIGS["13b-w4-s45"])
    if a.dense_only:
        cfg.update(sparse=0.0, topX=0)
    layers = bench.build_layers(cfg, dev, 0, 3)
    gen = torch.Generator(device=dev).manual_seed(1)
    xs, ys = bench.decoder_inputs(layers, dev, gen, batch=a.rows)
    seq = decode.OpSequence(layers, xs, ys, batched=True, fuse_shared_input=True, workspace=not a.no_ws)
    seq.launch()
    torch.cuda.synchronize()
    names = ["qkv", "o", "gate_up", "down"]
    wgs = 4096
    for gi, grp in enumerate(seq.groups):
        if gi < 4:
            continue  # first layer: cold
        buf = torch.zeros((wgs, 8), dtype=torch.int64, device=dev)
        sub = decode.OpSequence([layers[i] for i in grp], [xs[i] for i in grp], [ys[i] for i in grp], batched=True, fuse_shared_input=True,
                                workspace=not a.no_ws)
        sub.launch()
        torch.cuda.synchronize()
        lib.sqllm_debug_set_timeline(ctypes.c_void_p(buf.data_ptr()))
        sub.launch()
        torch.cuda.synchronize()
        lib.sqllm_debug_set_timeline(None)
        us = sub.profile(reps=2)
        raw = buf.cpu().numpy()
        where = (raw[:, 0].astype(np.uint64) >> np.uint64(48)).astype(np.int64)  # XCC id << 8 | HW_ID[15:8]
        raw = raw.copy()
        raw[:, 0] = (raw[:, 0].astype(np.uint64) & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64)
        t = raw.astype(np.float64) / 100.0
        keep = t[:, 0] > 0
        t, where = t[keep], where[keep]
        if not len(t):
            print(names[gi % 4], "no stamps")
            continue
        t0 = t[:, 0].min()
        def med(x):
            return float(np.percentile(x, 50))
        print(f"{names[gi % 4]} rows {a.rows} ws {not a.no_ws}: launch (events) {us.mean():.1f} us, {len(t)} dense workgroups; entry after the first "
              f"{med(t[:, 0] - t0):.2f} (last {float((t[:, 0] - t0).max()):.2f}); entry -> staged {med(t[:, 1] - t[:, 0]):.2f}; -> wave 0 decoded "
              f"{med(t[:, 2] - t[:, 1]):.2f} (last wave {med(t[:, 7] - t[:, 1]):.2f}); -> share staged {med(t[:, 3] - t[:, 2]):.2f}; -> wave 0 walked "
              f"{med(t[:, 4] - t[:, 3]):.2f}; -> all walked {med(t[:, 5] - t[:, 4]):.2f}; -> atomics issued {med(t[:, 6] - t[:, 5]):.2f}; "
              f"life {med(t[:, 6] - t[:, 0]):.2f}, last done at {float((t[:, 6] - t0).max()):.2f}", flush=True)
        life = t[:, 6] - t[:, 0]
        slow = np.argsort(life)[-5:]
        dec = t[:, 7] - t[:, 1]
        xcc = where >> 8
        cu = where  # (XCC, SE, SH, CU) as one key
        per_cu = {}
        for i, c in enumerate(cu):
            per_cu.setdefault(int(c), []).append(i)
        share = {1: [], 2: [], 3: []}
        for c, idx in per_cu.items():
            share.setdefault(len(idx), []).extend(dec[idx])
        print("   decode time by dense workgroups stamped on the same CU: " + ", ".join(f"{k}: n={len(v)} mean {np.mean(v):.2f}" for k, v in sorted(share.items()) if v) +
              "; by XCC: " + " ".join(f"{x}:{dec[xcc == x].mean():.1f}" for x in sorted(set(xcc.tolist()))) + f"; CUs seen {len(per_cu)}", flush=True)
        print("   life p90 %.2f p99 %.2f max %.2f; decode (staged -> last wave) p50 %.2f p99 %.2f max %.2f; walk (staged share -> all walked) p50 %.2f p99 %.2f max %.2f; slowest workgroups (index in the stamped set, life): %s" % (
            np.percentile(life, 90), np.percentile(life, 99), life.max(), med(t[:, 7] - t[:, 1]), np.percentile(t[:, 7] - t[:, 1], 99), (t[:, 7] - t[:, 1]).max(),
            med(t[:, 5] - t[:, 3]), np.percentile(t[:, 5] - t[:, 3], 99), (t[:, 5] - t[:, 3]).max(), [(int(i), round(float(life[i]), 1)) for i in slow]), flush=True)
        if gi >= 7:
            break


if __name__ == "__main__":
    main()
