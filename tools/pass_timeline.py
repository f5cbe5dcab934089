#!/usr/bin/env python3
"""Timeline of the gated pass (measurement build): per work item, the 100 MHz stamps item-begun / gate-passed /
decode-done / arrival-issued; printed per group as offsets from the pass's first stamp.

    python -m squeezellm_amd.build --ablation
    SQLLM_LIB=squeezellm_amd/libsqllm_hip_ablation.so python tools/pass_timeline.py [--config 7b-w4-s0] [--layers 4]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import numpy as np
    import torch

    from squeezellm_amd import _lib, decode, experimental

    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="7b-w4-s0")
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--groups", type=int, default=12, help="groups to print")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = experimental.load()
    cfg = bench.CONFIGS[a.config]
    layers = bench.build_layers(cfg, dev, 0, a.layers)
    gen = torch.Generator(device=dev).manual_seed(1)
    xs, ys = bench.decoder_inputs(layers, dev, gen)
    seq = decode.OpSequence(layers, xs, ys, fuse_shared_input=True)
    p0 = experimental.GatedPass(seq)
    buf = torch.zeros((p0.n_items, 4), dtype=torch.int64, device=dev)
    lib.sqllm_debug_set_timeline(ctypes.c_void_p(buf.data_ptr()))
    p = experimental.GatedPass(seq)
    lib.sqllm_debug_set_timeline(None)
    for _ in range(3):
        p.launch()
    torch.cuda.synchronize()
    buf.zero_()
    p.launch()
    torch.cuda.synchronize()
    print("status", p.status(), "items", p.n_items, "grid", p.grid, "kernel_us", round(p.profile(2), 1))
    t = buf.cpu().numpy().astype(np.float64) / 100.0
    # group of every item: recompute from the group sizes (items are in group order; dense-only configs: count per op)
    img = p.workspace.cpu().numpy()
    items = img[p.desc.items_offset:p.desc.items_offset + 16 * p.n_items].view(np.int32).reshape(-1, 4)
    segs = img[p.desc.segs_offset:p.desc.segs_offset + 128 * p.desc.n_ops].view(np.int32).reshape(-1, 32)
    group_of_seg = segs[:, 16 + 15]  # PassSegSparse.group: the last dword
    grp = group_of_seg[items[:, 0] & 0xffffff]
    dense = (items[:, 0] >> 24) == 0
    t0 = t[dense, 0][t[dense, 0] > 0].min()
    print("group  items |  begun: first / median / last | gate passed: first / median / last | decode done: median / last | arrival: median / last | wait at gate median | decode median | ack median")
    for g in range(min(a.groups, int(grp.max()) + 1)):
        m = dense & (grp == g)
        b, gp, dd, ar = (t[m, i] - t0 for i in range(4))
        print(f"{g:5d} {m.sum():6d} | {b.min():7.2f} {np.median(b):7.2f} {b.max():7.2f} | {gp.min():7.2f} {np.median(gp):7.2f} {gp.max():7.2f} | "
              f"{np.median(dd):7.2f} {dd.max():7.2f} | {np.median(ar):7.2f} {ar.max():7.2f} | {np.median(gp - b):6.2f} | {np.median(dd - gp):6.2f} | {np.median(ar - dd):6.2f}")


if __name__ == "__main__":
    main()
