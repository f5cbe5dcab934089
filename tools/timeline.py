#!/usr/bin/env python3
"""Where do the microseconds of ONE launch go?  (measurement build: SQLLM_ABLATION=1)

Every dense workgroup stamps the 100 MHz real-time clock at entry, after the codebook barrier, at
the end of its decode loop and after its atomics; this prints the distribution of those stamps
relative to the first workgroup's entry, next to the kernel's own duration.

    python -m squeezellm_amd.build --ablation
    SQLLM_LIB=squeezellm_amd/libsqllm_hip_ablation.so [SQLLM_OPTIONS=stream=0] python tools/timeline.py --shape 4096x4096 --bits 4 [--group 3]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch

    from squeezellm_amd import _lib, decode, synth

    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="4096x4096")
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--group", type=int, default=1)
    ap.add_argument("--sparse", type=float, default=0.0)
    ap.add_argument("--topx", type=int, default=0)
    ap.add_argument("--copies", type=int, default=24)
    ap.add_argument("--batch", type=int, default=0, help="rows of a *_batched op (the wide-batch sparse launch stamps its chunk workgroups)")
    a = ap.parse_args()
    K, N = map(int, a.shape.split("x"))
    dev = torch.device("cuda:0")
    lib = _lib.load()
    lib.sqllm_debug_set_timeline.argtypes = [ctypes.c_void_p]
    lib.sqllm_debug_set_timeline.restype = None
    layers = [synth.make_layer(K, N, a.bits, sparse_frac=a.sparse, topX=a.topx, heavy_rows=10 if a.sparse else 0, device=dev, seed=i)
              for i in range(a.copies * a.group)]
    x = torch.randn((a.batch, K) if a.batch else K, device=dev)
    ys = [torch.zeros((a.batch, N) if a.batch else N, device=dev) for _ in layers]
    seq = decode.OpSequence(layers, [x] * len(layers), ys, batched=a.batch > 0, fuse_shared_input=a.group > 1)
    plan = _lib.plan_query(a.bits, K, N, nnz=0 if not a.sparse else layers[0]["vals"].numel(), topX=a.topx)
    wgs = max(4096, a.group * ((plan["grid_x"] + 7) // 8 * 8))  # (the streaming kernel plans its own grid: be generous)
    buf = torch.zeros((len(seq.groups), wgs, 8), dtype=torch.int64, device=dev)
    seq.launch()  # warm (no probe)
    torch.cuda.synchronize()
    us = seq.profile(reps=2)
    # one launch per group with its own probe buffer
    for gi, grp in enumerate(seq.groups):
        lib.sqllm_debug_set_timeline(ctypes.c_void_p(buf[gi].data_ptr()))
        sub = decode.OpSequence([layers[i] for i in grp], [x] * len(grp), [ys[i] for i in grp], batched=a.batch > 0, fuse_shared_input=a.group > 1)
        sub.launch()
        torch.cuda.synchronize()
    lib.sqllm_debug_set_timeline(None)
    t = buf.cpu().numpy().astype(np.float64) / 100.0  # 100 MHz ticks -> us
    raw = buf.cpu().numpy()
    rows, crows = [], []
    for gi in range(2, len(seq.groups)):  # skip the first two (cold)
        g = t[gi]
        c = raw[gi][raw[gi][:, 0] < 0].astype(np.float64) / 100.0  # CSR chunk workgroups stamp their entry negated
        g = g[g[:, 0] > 0]  # dense workgroups only
        t0 = g[:, 0].min() if len(g) else np.inf
        if len(c):
            c[:, 0] = -c[:, 0]
            t0 = min(t0, c[:, 0].min())
            d = np.diff(c[:, :6], axis=1)
            crows.append([np.percentile(c[:, 0] - t0, 50)] + [np.percentile(d[:, i], 50) for i in range(5)] +
                         [np.percentile(c[:, 5] - c[:, 0], 50), (c[:, 5] - t0).max(), len(c)])
        if len(g):
          rows.append([np.percentile(g[:, 0] - t0, 50), (g[:, 0] - t0).max(), np.percentile(g[:, 1] - g[:, 0], 50),
                     np.percentile(g[:, 2] - g[:, 1], 50), np.percentile(g[:, 3] - g[:, 2], 50),
                     np.percentile(g[:, 3] - t0, 50), (g[:, 3] - t0).max(), (g[:, 4:8].max(axis=1) - g[:, 2]).mean(), len(g)])
    print(f"shape {a.shape} x{a.group} w{a.bits} sparse {a.sparse} rows {max(a.batch, 1)}: launches per op (events) {us[2:].mean():.2f} us")
    if rows:
        r = np.array(rows).mean(axis=0)
        print(f"  dense workgroups {int(r[8])}: entry after the first one: median {r[0]:.2f} us, last {r[1]:.2f} us")
        print(f"  entry -> codebook barrier passed        : median {r[2]:.2f} us")
        print(f"  barrier -> wave 0 done decoding         : median {r[3]:.2f} us   (slowest of waves 0-3 finishes {r[7]:+.2f} us later)")
        print(f"  wave 0 decode end -> atomics issued     : median {r[4]:.2f} us")
        print(f"  first entry -> atomics issued           : median {r[5]:.2f} us, last workgroup {r[6]:.2f} us")
    if crows:
        c = np.array(crows).mean(axis=0)
        print(f"  CSR chunk workgroups ({int(c[8])}): entry {c[0]:.2f} us after the first workgroup; entry -> round 1 landed + counts {c[1]:.2f}, "
              f"-> row pointers staged {c[2]:.2f}, -> rows found {c[3]:.2f}, -> x gathered + sums in LDS {c[4]:.2f}, -> atomics issued {c[5]:.2f}; "
              f"life {c[6]:.2f} us (median), last one done at {c[7]:.2f} us")


if __name__ == "__main__":
    main()
