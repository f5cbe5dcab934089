#!/usr/bin/env python3
"""Workload for the counter passes on the wide-batch dense kernel (13B gate/up shape, 4-bit): three variants told apart by
their grids -- 2048 rows split in registers, 1984 rows on fp32 planes (hi, mid, lo), 1920 rows on fp16-born planes (hi, mid).
Run under rocprofv3 --pmc ... --kernel-trace (tools/sessions/r04_s22.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from squeezellm_amd import _lib, decode, synth

dev = torch.device("cuda:0")
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K, N = 5120, 13824
lay = synth.make_layer(K, N, bits, device=dev, seed=1)
for name, B, planes, half in (("regs", 2048, 1 << 30, False), ("planes32", 1984, 1, False), ("planes16", 1920, 1, True)):
    x = torch.randn((B, K), device=dev)
    if half:
        x = x.half().float()
    y = torch.zeros((B, N), device=dev)
    _lib.set_option("split_planes_min_batch", planes)
    seq = decode.OpSequence([lay], [x], [y], batched=True)
    for _ in range(3):
        seq.launch()
    torch.cuda.synchronize()
_lib.set_option("split_planes_min_batch", 0)
