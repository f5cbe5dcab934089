#!/usr/bin/env python3
"""Debug aid: the gated pass beside competing kernels; on a timeout, dump the status words and the arrival state."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from squeezellm_amd import decode, experimental
from tests import test_gpu_pass as T

gpu = torch.device("cuda:0")
layers, xs, ys = T._chain(T.SMALL, 6, 4, 0.0045, 10, gpu, seed0=1100, scale=T._flat)
ys0_t = [y.clone() for y in ys]
seq = decode.OpSequence(layers, xs, ys, fuse_shared_input=True)
p = experimental.GatedPass(seq)
print("items", p.n_items, "grid", p.grid, "groups", seq.n_groups)
img = p.workspace.cpu().numpy()
items = img[p.desc.items_offset:p.desc.items_offset + 16 * p.n_items].view(np.int32).reshape(-1, 4)
segs = img[p.desc.segs_offset:p.desc.segs_offset + 128 * p.desc.n_ops].view(np.int32).reshape(-1, 32)
grp = segs[:, 31][items[:, 0] & 0xffffff]
expected = np.bincount(grp, minlength=seq.n_groups)
noise = torch.empty(64 << 20, device=gpu)
side = torch.cuda.Stream(gpu)
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    torch._foreach_copy_(ys, ys0_t)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(r % 6):
            noise.add_(1.0)
    t0 = time.perf_counter()
    p.launch()
    st = p.status()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    if st != (0, 0):
        w = p.workspace[:p.desc.state_bytes].view(torch.int32).cpu().numpy()
        sums = w[16:16 + 128 * seq.n_groups].reshape(-1, 128)[:, ::16].sum(axis=1)
        it = st[1]
        print(f"round {r}: status {st} words {w[:8]} (error, item, sc1-load sum, total, workgroup, RMW sum, sc1-load sum again) dt {dt:.3f}s; item {it}: role {items[it, 0] >> 24} seg {items[it, 0] & 0xffffff} group {grp[it]}")
        bad = [(g, int(a), int(b)) for g, (a, b) in enumerate(zip(sums, expected)) if a != b]
        print("   groups short of their arrivals (group, have, expected):", bad[:8])
    else:
        print(f"round {r}: ok dt {dt * 1e3:.2f} ms")
