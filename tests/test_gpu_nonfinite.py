"""Non-finite operands on the *_batched operators' default routes.

The reference's batched kernels are fp32 FMA chains (/root/reference/squeezellm/quant_cuda_kernel.cu:1011-1036 dense,
:1061-1089 CSR, :1127-1164 top-X): an infinite operand gives +-inf (NaN only for inf - inf and 0 x inf), a NaN gives NaN,
and nothing else is touched.  From 5 rows (4-bit) / 9 rows (3-bit) the dense term runs on the bf16 matrix instructions
with operands split three ways, where inf - inf inside the split turns an infinity into NaN; the kernels therefore
recompute every non-finite sum of the matrix instructions as the reference's fp32 chain (csrc/sqllm_split_common.h:
dense_term_fp32).  These tests pin that: the PATTERN of NaN / +inf / -inf of every route equals the oracle's, and the
finite outputs keep their tolerance -- at 5, 16, 64 and 256 rows, on the fused small launch, the tile form and the wide form.
"""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

TOL_FP64 = 2e-5


@pytest.fixture(scope="module")
def qc():
    from squeezellm_amd import quant_cuda

    return quant_cuda


def _run(qc, gpu, case, x, mul, entry="module"):
    import torch

    t = H.to_torch(case, gpu)
    yt = torch.from_numpy(mul.copy()).to(gpu)
    H.call_op(qc, t, torch.from_numpy(x).to(gpu), yt, "hybrid", True, entry=entry)
    torch.cuda.synchronize()
    return yt.cpu().numpy()


def _same_pattern(got, ref):
    assert np.array_equal(np.isnan(got), np.isnan(ref)), ("NaN pattern", int(np.isnan(got).sum()), int(np.isnan(ref).sum()))
    assert np.array_equal(np.isposinf(got), np.isposinf(ref)), ("+inf pattern", int(np.isposinf(got).sum()), int(np.isposinf(ref).sum()))
    assert np.array_equal(np.isneginf(got), np.isneginf(ref)), ("-inf pattern", int(np.isneginf(got).sum()), int(np.isneginf(ref).sum()))
    fin = np.isfinite(ref)
    if fin.any():
        scale = np.abs(ref[fin]).max() or 1.0
        assert np.abs(got[fin] - ref[fin]).max() / scale <= TOL_FP64


POISONS = ["vec+inf", "vec-inf", "vec-nan", "vec-both-infs", "codebook-inf", "codebook-nan", "vals-inf", "vals-nan"]


@pytest.mark.parametrize("poison", POISONS)
@pytest.mark.parametrize("form,batch", [("default", 5), ("default", 16), ("default", 64), ("default", 256), ("wide", 64), ("wide", 256)])
@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("entry", H.ENTRIES)
def test_nonfinite_operands_follow_the_reference(qc, gpu, bits, batch, poison, form, entry):
    from squeezellm_amd import _lib

    K, N = 512, 320
    case = H.make_case(bits, K, N, sparse=0.02, topX=3, heavy_rows=1, seed=10 * bits + batch)
    rng = np.random.default_rng(batch)
    x = rng.normal(size=(batch, K)).astype(np.float16).astype(np.float32)
    mul = rng.normal(0, 0.5, size=(batch, N)).astype(np.float32)
    with np.errstate(all="ignore"):
        if poison == "vec+inf":
            x[batch // 2, 37] = np.inf
        elif poison == "vec-inf":
            x[0, K - 1] = -np.inf
        elif poison == "vec-nan":
            x[batch - 1, 100] = np.nan
        elif poison == "vec-both-infs":  # inf - inf in one row: NaN there in the reference too
            x[1, 3] = np.inf
            x[1, 200] = -np.inf
        elif poison.startswith("codebook"):
            lut = case["lookup_table"].copy()
            lut[N // 3, 2] = np.inf if poison.endswith("inf") else np.nan  # (an index every 512-k column uses)
            case["lookup_table"] = lut
        else:
            vals = case["vals"].copy()
            vals[len(vals) // 2] = np.inf if poison.endswith("inf") else np.nan
            case["vals"] = vals
        ref = H.oracle_ref(case, x, mul, "hybrid")
    assert not np.isfinite(ref).all(), "the poison must reach an output"
    try:
        _lib.set_option("mfma_min_batch", 5)  # (5 rows on the fused small launch: the default keeps them on the batch tiles, plain fp32 chains)
        if form == "wide":
            _lib.set_option("mfma_wide_min_batch", 64)
        got = _run(qc, gpu, case, x, mul, entry)
    finally:
        _lib.set_option("mfma_wide_min_batch", 0)
        _lib.set_option("mfma_min_batch", 0)
    _same_pattern(got, ref)
