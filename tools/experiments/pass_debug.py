#!/usr/bin/env python3
"""Debug aid for the gated pass: status words and arrival shards after eager launches and graph replays."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from squeezellm_amd import decode, experimental, synth
from tests import test_gpu_pass as T

gpu = torch.device("cuda:0")
layers, xs, ys = T._chain(synth.MODEL_SHAPES["llama-7b"]["linears"], 3, 4, 0.0, 0, gpu, seed0=704, scale=T._flat)
ys0_t = [y.clone() for y in ys]
seq = decode.OpSequence(layers, xs, ys, fuse_shared_input=True)
p = experimental.GatedPass(seq)
print("items", p.n_items, "grid", p.grid, "state_bytes", p.desc.state_bytes, "ws", hex(p.workspace.data_ptr()))


def dump(tag):
    torch.cuda.synchronize()
    st = p.workspace[:p.desc.state_bytes].view(torch.int32).cpu().numpy()
    sums = st[16:16 + 128 * p.desc.n_groups].reshape(-1, 128)[:, ::16].sum(axis=1)
    print(tag, "status", st[:4], "arrivals per group", sums.tolist())


def step_foreach():
    torch._foreach_copy_(ys, ys0_t)
    p.launch()


def step_loop():
    for a, b in zip(ys, ys0_t):
        a.copy_(b)
    p.launch()


p.launch(); dump("eager1")
p.launch(); dump("eager2")
for name, fn in (("launch-only", p.launch), ("foreach", step_foreach), ("loop", step_loop)):
    side = torch.cuda.Stream(gpu)
    side.wait_stream(torch.cuda.current_stream(gpu))
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream(gpu).wait_stream(side)
    dump(name + " warm")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    dump(name + " captured")
    for r in range(3):
        g.replay()
        dump(f"{name} replay{r}")
        print("   status()", p.status())
