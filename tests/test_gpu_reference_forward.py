"""The UNMODIFIED reference `QuantLinearLUT.forward` (squeezellm/quant.py:211-383) on the real HIP
kernels.

`oracle/build_ref.sh` stages the reference's quant.py, byte for byte, into the git-ignored
`oracle/_ref/reference_py/` (it travels to the GPU box with the snapshot, like the reference-kernel
.so).  Here it is imported with this repository's root on sys.path, so its `import quant_cuda`
(quant.py:5) resolves to the MI355X implementation; modules are built with the reference's own
constructor, filled with seeded operands, and their forward() -- zeros / bias.clone(), x.float(),
the operator chosen by the reference's own if-ladder, the cast back, the bias add -- is compared
with the oracle's restatement of it.  Every arm: matvec + batched x dense / spmv / hybrid /
balanced x 3 / 4 bit x fp16 / fp32 input.
"""
import contextlib
import importlib.util
import io
import os
import sys

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

REF_QUANT = os.path.join(H.ROOT, "oracle", "_ref", "reference_py", "quant.py")


@pytest.fixture(scope="module")
def refquant():
    if not os.path.exists(REF_QUANT):
        pytest.skip("oracle/_ref/reference_py/quant.py not staged (oracle/build_ref.sh needs /root/reference)")
    if H.ROOT not in sys.path:
        sys.path.insert(0, H.ROOT)
    import quant_cuda  # the root shim -> squeezellm_amd.quant_cuda

    spec = importlib.util.spec_from_file_location("reference_squeezellm_quant", REF_QUANT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.quant_cuda is quant_cuda
    return mod


def build_module(refquant, case, kind, gpu, bias):
    import torch

    K, N, bits = case["K"], case["N"], case["bits"]
    numvals = case["vals"].size if kind != "dense" else 0
    with contextlib.redirect_stdout(io.StringIO()):  # the constructor prints its buffers
        m = refquant.QuantLinearLUT(bits, K, N, bias, include_sparse=kind != "dense", numvals=numvals,
                                    topX=case["full_rows"].shape[1] if kind == "hybrid" else 0, balanced=kind == "balanced")
    m = m.to(gpu)
    m.qweight.copy_(torch.from_numpy(case["qweight"]))
    m.lookup_table.copy_(torch.from_numpy(case["lookup_table"]))
    if bias:
        m.bias.copy_(torch.from_numpy(case["bias"]))
    if kind != "dense":
        m.rows.copy_(torch.from_numpy(case["rows"]))
        m.cols.copy_(torch.from_numpy(case["cols"]))
        m.vals.copy_(torch.from_numpy(case["vals"]))
    if kind == "hybrid":
        m.full_rows.copy_(torch.from_numpy(case["full_rows"]))
        m.full_row_indices.copy_(torch.from_numpy(case["full_row_indices"]))
    if kind == "balanced":
        sr, nt, _ = H.oracle.startrows_balanced(case["rows"], N, numvals)
        assert nt == m.num_threads
        m.startrows.copy_(torch.from_numpy(sr))
    return m


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("kind", ["dense", "spmv", "hybrid", "balanced"])
@pytest.mark.parametrize("bias", [False, True])
def test_reference_forward_on_hip_kernels(refquant, gpu, bits, kind, bias):
    import torch

    K, N = 512, 384
    case = H.make_case(bits, K, N, sparse=0.02 if kind != "dense" else 0, topX=10 if kind == "hybrid" else 0,
                       heavy_rows=2 if kind != "dense" else 0, seed=bits * 10 + len(kind))
    rng = np.random.default_rng(7)
    case["bias"] = rng.normal(0, 0.05, N).astype(np.float32) if bias else None
    layer = {k: case[k] for k in ("bits", "qweight", "lookup_table", "bias")}
    if kind != "dense":
        layer.update(rows=case["rows"], cols=case["cols"], vals=case["vals"])
    if kind == "hybrid":
        layer.update(full_rows=case["full_rows"], full_row_indices=case["full_row_indices"])
    m = build_module(refquant, case, kind, gpu, bias)
    with torch.cuda.device(gpu):  # quant.py:218 / :317 allocate on "cuda" = the current device
        for dtype, tol in ((np.float32, 2e-5), (np.float16, 2e-3)):
            # matvec branch (x.numel() == x.shape[-1]) with leading 1-dims, then batched shapes incl.
            # 9 and 40 rows (the matrix-core kernel) and a 3-D input (quant.py:313-321)
            for shape in ((K,), (1, 1, K), (3, K), (2, 4, K), (9, K), (40, K)):
                x = rng.normal(size=shape).astype(dtype)
                y = m(torch.from_numpy(x).to(gpu))
                torch.cuda.synchronize()
                ref = H.oracle.quantlinear_forward(x, layer)
                assert tuple(y.shape) == ref.shape and y.dtype == torch.from_numpy(ref[:0]).dtype
                assert H.rel_err(y.float().cpu().numpy(), ref.astype(np.float32)) <= tol, (dtype, shape)
