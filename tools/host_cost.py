#!/usr/bin/env python3
"""What one operator call costs on the HOST (eager, no graph): microseconds per call of the quant_cuda names, of the
C function behind them called straight through ctypes with everything pre-marshalled (the floor of this binding),
and of QuantLinearLUT.forward (the reference's four launches per linear).  Tiny layer: the kernel is short, the loop
is host-bound, so wall time / calls = host cost per call.

    python tools/host_cost.py
"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from squeezellm_amd import _lib, quant, quant_cuda, synth  # noqa: E402

dev = torch.device("cuda:0")


def per_call(fn, n=20000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()  # (host time only: the queue drains afterwards)
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6


out = {}
for tag, sparse, topX in (("dense", 0.0, 0), ("hybrid", 0.0045, 10)):
    lay = synth.make_layer(256, 256, 4, sparse_frac=sparse, topX=topX, device=dev, seed=1)
    x, y = torch.randn(256, device=dev), torch.zeros(256, device=dev)
    if sparse:
        f = lambda: quant_cuda.vecquant4matmul_spmv_hybrid_nuq_perchannel(lay["rows"], lay["cols"], lay["vals"], x, lay["full_rows"],  # noqa: E731
                                                                          lay["full_row_indices"], y, 256, lay["qweight"], lay["lookup_table"])
    else:
        f = lambda: quant_cuda.vecquant4matmul_nuq_perchannel(x, lay["qweight"], y, lay["lookup_table"])  # noqa: E731
    out[f"op_call_{tag}_us"] = round(per_call(f), 2)
    lib = _lib.load()
    op = _lib.SqllmOp(bits=4, batch=0, K=256, N=256)
    op.vec, op.qweight, op.mul, op.lookup_table = x.data_ptr(), lay["qweight"].data_ptr(), y.data_ptr(), lay["lookup_table"].data_ptr()
    if sparse:
        op.rows, op.cols, op.vals, op.nnz = lay["rows"].data_ptr(), lay["cols"].data_ptr(), lay["vals"].data_ptr(), lay["vals"].numel()
        op.full_rows, op.full_row_indices, op.topX = lay["full_rows"].data_ptr(), lay["full_row_indices"].data_ptr(), 10
    stream = torch.cuda.current_stream(dev).cuda_stream
    ref = ctypes.byref(op)
    out[f"c_launch_{tag}_us"] = round(per_call(lambda: lib.sqllm_launch(ref, stream)), 2)
    mod = quant.QuantLinearLUT.from_operands(lay)
    x16 = x.half().reshape(1, 1, -1)
    with torch.no_grad():
        out[f"forward_{tag}_us"] = round(per_call(lambda: mod(x16), n=5000), 2)
e = torch.empty(256, device=dev)
out["torch_zero_us"] = round(per_call(lambda: e.zero_()), 2)
out["torch_empty_launch_floor_us"] = round(per_call(lambda: torch.zeros(256, device=dev), n=5000), 2)
print(json.dumps(out))
