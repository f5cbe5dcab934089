"""Randomised GPU parity: hypothesis draws (bits, K, N, batch, sparsity, top-X, geometry knobs) and
every draw is checked against the fp64 oracle, through the operator names (by one of the three entries of
tests/helpers.py:call_op, drawn too) and through the fused fp16 linear.  Shapes cover everything the C ABI accepts (K % 32 == 0, N % 4 == 0): ragged last
column tiles, K slices with ragged ends, single-step and many-step slices, empty CSR, batch tiles
with a ragged last tile."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from tests import helpers as H

pytestmark = pytest.mark.gpu

CASE = st.fixed_dictionaries(dict(
    bits=st.sampled_from([3, 4]),
    K=st.integers(1, 40).map(lambda v: 32 * v),
    N=st.integers(1, 160).map(lambda v: 4 * v),
    batch=st.sampled_from([0, 1, 2, 3, 4, 5, 6, 7, 8, 11, 13, 20, 40]),
    sparse=st.sampled_from([0.0, 0.0, 0.002, 0.02, 0.3]),
    topX=st.sampled_from([0, 0, 1, 3, 10]),
    target_wgs=st.sampled_from([0, 0, 1, 7, 64, 4096]),
    seed=st.integers(0, 2**16),
    entry=st.sampled_from(H.ENTRIES),  # the Python module, the header's named symbol, sqllm_launch_ws(NULL): tests/helpers.py
))


def _npl(lay):
    import torch

    return {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in lay.items()}


# SQLLM_PROPERTY_EXAMPLES=<n> (bug hunts: tools/sessions/r06.sh fuzz) draws n fresh, non-derandomised examples instead of the suite's fixed 150
_N_EXAMPLES = int(os.environ.get("SQLLM_PROPERTY_EXAMPLES", "0"))


@settings(max_examples=_N_EXAMPLES or 150, deadline=None, derandomize=not _N_EXAMPLES, database=None, suppress_health_check=list(HealthCheck))
@given(case=CASE)
def test_random_shapes_operator_and_fused_linear(gpu, case):
    import torch

    import quant_cuda as qc
    from squeezellm_amd import _lib, quant, synth

    bits, K, N, batch = case["bits"], case["K"], case["N"], case["batch"]
    sparse = case["sparse"]
    topX = min(case["topX"], N) if sparse > 0 else 0
    lay = synth.make_layer(K, N, bits, sparse_frac=sparse, topX=topX, heavy_rows=1 if sparse > 0 else 0,
                           bias=True, device=gpu, seed=case["seed"])
    if sparse > 0 and lay["vals"].numel() == 0:
        sparse, topX = 0.0, 0
        lay.update(rows=None, cols=None, vals=None, full_rows=None, full_row_indices=None)
    kind = "hybrid" if topX else ("spmv" if sparse > 0 else "dense")
    npl = _npl(lay)
    g = torch.Generator(device=gpu).manual_seed(case["seed"])
    rows = max(batch, 1)
    x16 = torch.randn((rows, K), device=gpu, generator=g).half()
    x = x16.float()
    mul0 = torch.randn((rows, N), device=gpu, generator=g)
    xa, ma = (x, mul0) if batch else (x[0], mul0[0])
    ref = np.asarray(H.oracle_ref(npl, xa.cpu().numpy(), ma.cpu().numpy(), kind), np.float64).reshape(rows, N)
    _lib.set_option("target_wgs", case["target_wgs"])
    try:
        y = ma.clone()
        H.call_op(qc, lay, xa, y, kind, batch > 0, entry=case["entry"])
        assert H.rel_err(y.cpu().numpy().reshape(rows, N), ref) < 3e-5, case
        # fused linear on the same fp16 activations: fp16(W x + bias) within one fp16 ulp
        mod = quant.QuantLinearLUT.from_operands(lay)
        mod.__class__ = quant.QuantLinearLUTFused
        out = mod(x16 if batch else x16.reshape(1, 1, K)).reshape(rows, N).cpu().numpy().astype(np.float64)
        exact = ref - mul0.cpu().numpy().astype(np.float64) + npl["bias"].astype(np.float64)
        tol = np.maximum(np.abs(exact), 2.0**-14) * 2.0**-10 + 2e-6 + 3e-5 * np.abs(mul0.cpu().numpy())
        assert (np.abs(out - exact) <= tol).all(), case
        assert int(next(iter(mod._ws.values())).count_nonzero()) == 0
    finally:
        _lib.set_option("target_wgs", 0)
