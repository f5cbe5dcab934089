#!/usr/bin/env python3
"""Record outputs of the REFERENCE kernels on an MI355X as golden vectors.

oracle/_ref/libsqllm_ref.so is the reference's squeezellm/quant_cuda_kernel.cu compiled unmodified
by hipcc (oracle/build_ref.sh, dev container).  This script runs its twelve launchers on seeded
operands on the GPU box and stores operands + outputs in gpurun_out/refkernel_w{3,4}.npz; the
files are then committed under tests/golden/ and pin oracle/sqllm_oracle.{py,c} on CPU
(tests/test_oracle_cpu.py) wherever the tests run.

    gpurun -- python tests/golden/make_refkernel_golden.py        # then: cp gpurun_out/refkernel_*.npz tests/golden/
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests import helpers as H  # noqa: E402


def main():
    import torch

    ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsqllm_ref.so"))
    dev = torch.device("cuda:0")
    P = ctypes.c_void_p
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for bits in (3, 4):
        K, N, topX = 256, 128, 10  # the reference needs K % 128 == 0 and N % 128 == 0
        case = H.make_case(bits, K, N, sparse=0.02, topX=topX, heavy_rows=1, empty_rows=(5,), seed=100 + bits)
        rng = np.random.default_rng(200 + bits)
        x1 = rng.normal(size=K).astype(np.float32)
        xb = rng.normal(size=(3, K)).astype(np.float32)
        mul1 = rng.normal(0, 0.5, N).astype(np.float32)
        mulb = rng.normal(0, 0.5, (3, N)).astype(np.float32)
        t = H.to_torch(case, dev)
        out = {k: v for k, v in case.items() if isinstance(v, np.ndarray)}
        out.update(bits=np.int32(bits), K=np.int32(K), N=np.int32(N), x1=x1, xb=xb, mul1=mul1, mulb=mulb)
        for batch, x, mul, tag in ((0, x1, mul1, "1"), (3, xb, mulb, "b")):
            xt = torch.from_numpy(x).to(dev)
            args_sp = (P(t["rows"].data_ptr()), P(t["cols"].data_ptr()), P(t["vals"].data_ptr()), case["vals"].size)
            y = torch.from_numpy(mul).to(dev)
            assert ref.refk_dense(bits, batch, P(xt.data_ptr()), P(t["qweight"].data_ptr()), P(y.data_ptr()),
                                  P(t["lookup_table"].data_ptr()), K, N) == 0
            out[f"y_dense_{tag}"] = y.cpu().numpy()
            y = torch.from_numpy(mul).to(dev)
            assert ref.refk_spmv(bits, batch, *args_sp, P(xt.data_ptr()), P(y.data_ptr()), N, P(t["qweight"].data_ptr()),
                                 P(t["lookup_table"].data_ptr()), K, N) == 0
            out[f"y_spmv_{tag}"] = y.cpu().numpy()
            y = torch.from_numpy(mul).to(dev)
            assert ref.refk_hybrid(bits, batch, *args_sp, P(xt.data_ptr()), P(t["full_rows"].data_ptr()),
                                   P(t["full_row_indices"].data_ptr()), topX, P(y.data_ptr()), N,
                                   P(t["qweight"].data_ptr()), P(t["lookup_table"].data_ptr()), K, N) == 0
            out[f"y_hybrid_{tag}"] = y.cpu().numpy()
        path = os.path.join(ROOT, "gpurun_out", f"refkernel_w{bits}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path)


if __name__ == "__main__":
    main()
