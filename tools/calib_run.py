#!/usr/bin/env python3
"""Launch calibration / ablation kernels WITHOUT per-dispatch events (plain sqllm_launch_sequence),
meant to be run under `rocprofv3 --kernel-trace` (needs the SQLLM_ABLATION=1 build)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from squeezellm_amd import _lib, decode, synth  # noqa: E402

dev = torch.device("cuda:0")
modes = [int(m) for m in (sys.argv[1] if len(sys.argv) > 1 else "0,100,113,14").split(",")]
for shp in ("4096x4096", "4096x11008"):
    K, N = map(int, shp.split("x"))
    copies = max(4, int(700e6 / synth.algorithmic_bytes(K, N, 4)))
    layers = [synth.make_layer(K, N, 4, device=dev, seed=i) for i in range(copies)]
    xs = [torch.randn(K, device=dev) for _ in layers]
    ys = [torch.zeros(N, device=dev) for _ in layers]
    for m in modes:
        _lib.set_option("ablate", m)
        seq = decode.OpSequence(layers, xs, ys)
        for _ in range(3):
            seq.launch()
        torch.cuda.synchronize()
