#!/usr/bin/env python3
"""Debug aid (measurement library): when did each workgroup of the gated pass START, beside competing kernels?"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from squeezellm_amd import _lib, decode, experimental
from tests import test_gpu_pass as T

gpu = torch.device("cuda:0")
lib = experimental.load()
layers, xs, ys = T._chain(T.SMALL, 6, 4, 0.0, 0, gpu, seed0=1100, scale=T._flat)  # dense only: every item stamps
ys0_t = [y.clone() for y in ys]
seq = decode.OpSequence(layers, xs, ys, fuse_shared_input=True)
p0 = experimental.GatedPass(seq)
buf = torch.zeros((p0.n_items, 4), dtype=torch.int64, device=gpu)
lib.sqllm_debug_set_timeline(ctypes.c_void_p(buf.data_ptr()))
p = experimental.GatedPass(seq)
lib.sqllm_debug_set_timeline(None)
print("items", p.n_items, "grid", p.grid, "groups", seq.n_groups)
noise = torch.empty(64 << 20, device=gpu)
side = torch.cuda.Stream(gpu)
for r in range(12):
    torch._foreach_copy_(ys, ys0_t)
    buf.zero_()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        ev0.record()
        for _ in range(r % 4):
            noise.add_(1.0)
        ev1.record()
    p.launch()
    st = p.status()
    torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(np.float64) / 100.0  # us
    first = t[:p.grid, 0]
    t0 = first[first > 0].min()
    late = np.nonzero(first - t0 > 1e5)[0]  # started more than 0.1 s after the first
    print(f"round {r}: noise kernels {r % 4} ({ev0.elapsed_time(ev1) * 1e3:.0f} us) status {st}; workgroups started: within 100 us {int((first - t0 < 100).sum())}, "
          f"within 10 ms {int((first - t0 < 1e4).sum())}, later than 0.1 s {late.size}" + (f" (ids {late[:6].tolist()} .. {late[-3:].tolist()}, at +{(first[late] - t0).min() / 1e6:.2f} s)" if late.size else ""))
