"""GPU parity tests: every quant_cuda operator, called through the C ABI on an MI355X, against the
CPU oracle on the same seeded inputs.

Tolerances (BASELINE.json north_star / SURVEY.md 8(c)):
  * unpacked indices: bit-exact (probed through the op with an identity codebook, see below);
  * fp32 outputs vs the fp64 oracle: max|y - y_ref| / max|y_ref| <= 2e-5 (the kernels, like the
    reference, sum in fp32 with atomics, so the order of additions is unspecified);
  * vs the "fp16 dequant-then-matmul" path the north_star quotes: <= 1e-3.
"""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

TOL_FP64 = 2e-5
TOL_FP16_PATH = 1e-3


@pytest.fixture(scope="module")
def qc():
    from squeezellm_amd import quant_cuda

    return quant_cuda


def run_op(qc, gpu, case, kind, batch, seed=1, mul_init="random", entry="module"):
    import torch

    rng = np.random.default_rng(seed)
    K, N = case["K"], case["N"]
    batched = batch > 0
    x = rng.normal(size=(batch, K) if batched else (K,)).astype(np.float16).astype(np.float32)
    if mul_init == "random":  # bias-like pre-load: the op must ADD to it (quant.py:214-216)
        mul = rng.normal(0, 0.5, size=(batch, N) if batched else (N,)).astype(np.float32)
    else:
        mul = np.zeros((batch, N) if batched else (N,), np.float32)
    t = H.to_torch(case, gpu)
    xt, yt = torch.from_numpy(x).to(gpu), torch.from_numpy(mul).to(gpu)
    H.call_op(qc, t, xt, yt, kind, batched, entry=entry)
    torch.cuda.synchronize()
    return x, mul, yt.cpu().numpy()


SMALL = [(4, 128, 128), (3, 128, 128), (4, 256, 384), (3, 96 * 2, 260), (4, 32, 4), (3, 32, 8), (4, 1024, 132)]


@pytest.mark.parametrize("bits,K,N", SMALL)
@pytest.mark.parametrize("kind", ["dense", "spmv", "hybrid"])
@pytest.mark.parametrize("batch", [0, 1, 2, 3, 8, 9])
@pytest.mark.parametrize("entry", H.ENTRIES)
def test_small_shapes_vs_oracle(qc, gpu, bits, K, N, kind, batch, entry):
    """Every op at every small batch through the Python module, through the header's NAMED C symbol for it (workspace-less) and
    through sqllm_launch_ws with a NULL workspace (tests/helpers.py: _call_c_abi)."""
    case = H.make_case(bits, K, N, sparse=0.03 if kind != "dense" else 0, topX=3 if kind == "hybrid" else 0,
                       heavy_rows=1 if kind != "dense" and N >= 8 else 0, seed=bits * 1000 + K + N)
    x, mul, got = run_op(qc, gpu, case, kind, batch, entry=entry)
    ref = H.oracle_ref(case, x, mul, kind)
    assert got.shape == ref.shape
    assert H.rel_err(got, ref) <= TOL_FP64


LLAMA7B = [(4096, 4096), (4096, 11008), (11008, 4096)]


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("K,N", LLAMA7B)
@pytest.mark.parametrize("kind", ["dense", "hybrid"])
def test_llama7b_shapes_vs_oracle(qc, gpu, bits, K, N, kind):
    """BASELINE configs 2 and 3 at full size (C oracle: a few seconds per case)."""
    case = H.make_case(bits, K, N, sparse=0.0045 if kind == "hybrid" else 0, topX=10 if kind == "hybrid" else 0,
                       heavy_rows=10 if kind == "hybrid" else 0, seed=7)
    x, mul, got = run_op(qc, gpu, case, kind, 0)
    ref = H.c_matvec(H.c_oracle(), case, x, mul, batched=False)
    assert H.rel_err(got, ref) <= TOL_FP64
    # the tolerance the north_star states, against the fp16 dequant-then-matmul path
    sparse = {k: case[k] for k in ("rows", "cols", "vals", "full_rows", "full_row_indices")} if kind == "hybrid" else {}
    ref16 = H.oracle.matvec_fp16_path(x, case["qweight"], mul, case["lookup_table"], bits, **sparse)
    assert H.rel_err(got, ref16) <= TOL_FP16_PATH


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("batch", [2, 5, 8])
def test_llama13b_batched_hybrid(qc, gpu, bits, batch):
    """BASELINE config 4: 13B shapes, 0.45 % sparse + top-10, batch 1..8."""
    K, N = 5120, 5120
    case = H.make_case(bits, K, N, sparse=0.0045, topX=10, heavy_rows=10, seed=13)
    x, mul, got = run_op(qc, gpu, case, "hybrid", batch)
    ref = H.c_matvec(H.c_oracle(), case, x, mul, batched=True)
    assert H.rel_err(got, ref) <= TOL_FP64


@pytest.mark.parametrize("bits", [3, 4])
def test_unpack_indices_bit_exact(qc, gpu, bits):
    """Index decode is bit-exact: with the identity codebook LUT[n][v] = v and one-hot inputs
    x = e_k, the op returns y[n] = idx(k, n) exactly (small integers are exact in fp32).  The batched
    op with x = I_K reads out the whole index matrix."""
    import torch

    K, N = 256, 512
    case = H.make_case(bits, K, N, seed=99)
    idx_ref = H.oracle.unpack_indices(case["qweight"], bits)  # [K, N], pinned to pack2 in the CPU tests
    lut = np.tile(np.arange(1 << bits, dtype=np.float32), (N, 1))
    t = H.to_torch(dict(case, lookup_table=lut), gpu)
    x = torch.eye(K, device=gpu, dtype=torch.float32)
    y = torch.zeros((K, N), device=gpu, dtype=torch.float32)
    H.call_op(qc, t, x, y, "dense", True)
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy().astype(np.uint8), idx_ref)
    # and through the un-batched op for a few k's, including the 3-bit straddlers k = 10, 21
    for k in (0, 9, 10, 11, 21, 22, 31, 32 + 10, K - 1):
        xk = torch.zeros(K, device=gpu)
        xk[k] = 1.0
        yk = torch.zeros(N, device=gpu)
        H.call_op(qc, t, xk, yk, "dense", False)
        assert np.array_equal(yk.cpu().numpy().astype(np.uint8), idx_ref[k])


@pytest.mark.parametrize("bits", [3, 4])
def test_sparse_edge_cases(qc, gpu, bits):
    K, N = 512, 256
    # (a) empty CSR (nnz == 0)
    case = H.make_case(bits, K, N, seed=3)
    case.update(rows=np.zeros(N + 1, np.int32), cols=np.zeros(0, np.int32), vals=np.zeros(0, np.float32))
    x, mul, got = run_op(qc, gpu, case, "spmv", 0)
    assert H.rel_err(got, H.oracle_ref(case, x, mul, "spmv")) <= TOL_FP64
    # (b) empty rows at both ends + in the middle, one row holding most of the non-zeros
    case = H.make_case(bits, K, N, sparse=0.01, heavy_rows=2, empty_rows=(0, 1, 100, N - 1), seed=4)
    assert case["rows"][-1] == case["vals"].size
    for batch in (0, 3):
        x, mul, got = run_op(qc, gpu, case, "spmv", batch)
        assert H.rel_err(got, H.oracle_ref(case, x, mul, "spmv")) <= TOL_FP64
    # (c) duplicate full_row_indices accumulate (each column adds on its own, kernel.cu:1120-1121)
    case = H.make_case(bits, K, N, sparse=0.01, topX=4, dup_topx=True, seed=5)
    x, mul, got = run_op(qc, gpu, case, "hybrid", 0)
    assert H.rel_err(got, H.oracle_ref(case, x, mul, "hybrid")) <= TOL_FP64
    # (d) a single dense CSR row longer than one chunk and a matrix whose nnz spans several chunks
    case = H.make_case(bits, 4096, 128, sparse=0.02, heavy_rows=3, seed=6)
    x, mul, got = run_op(qc, gpu, case, "spmv", 0)
    assert H.rel_err(got, H.oracle_ref(case, x, mul, "spmv")) <= TOL_FP64
    # (e) very sparse: more rows than non-zeros per chunk span (exercises the wide-span fallback)
    case = H.make_case(bits, 128, 8192, sparse=0.0008, seed=8)
    x, mul, got = run_op(qc, gpu, case, "spmv", 0)
    assert H.rel_err(got, H.oracle_ref(case, x, mul, "spmv")) <= TOL_FP64


@pytest.mark.parametrize("bits", [3, 4])
def test_accumulates_into_mul_and_zero_input(qc, gpu, bits):
    """mul is added to, never overwritten; x = 0 leaves mul untouched bit-for-bit."""
    import torch

    case = H.make_case(bits, 256, 256, sparse=0.02, topX=2, seed=11)
    t = H.to_torch(case, gpu)
    y0 = torch.randn(256, device=gpu)
    y = y0.clone()
    H.call_op(qc, t, torch.zeros(256, device=gpu), y, "hybrid", False)
    assert torch.equal(y, y0)
    x = torch.randn(256, device=gpu)
    H.call_op(qc, t, x, y, "hybrid", False)
    H.call_op(qc, t, x, y, "hybrid", False)  # twice: y0 + 2 * op(x)
    ref = H.oracle_ref(case, x.cpu().numpy(), np.zeros(256, np.float32), "hybrid")
    assert H.rel_err(y.cpu().numpy(), y0.cpu().numpy() + 2 * ref) <= TOL_FP64


@pytest.mark.parametrize("bits", [3, 4])
def test_full_size_properties(qc, gpu, bits):
    """Size-independent properties at a BASELINE-size shape (65B down_proj, 22016 -> 8192):
    linearity in x and agreement of the batched op with per-row matvec calls."""
    import torch

    from squeezellm_amd import synth

    lay = synth.make_layer(22016, 8192, bits, sparse_frac=0.0045, topX=10, heavy_rows=10, device=gpu, seed=5)
    lay = dict(lay)
    g = torch.Generator(device=gpu).manual_seed(1)
    x1 = torch.randn(22016, device=gpu, generator=g)
    x2 = torch.randn(22016, device=gpu, generator=g)

    def op(x):
        y = torch.zeros(8192, device=gpu)
        H.call_op(qc, lay, x, y, "hybrid", False)
        return y

    y1, y2, y12 = op(x1), op(x2), op(2.0 * x1 - 0.5 * x2)
    lin = 2.0 * y1 - 0.5 * y2
    assert float((y12 - lin).abs().max() / lin.abs().max()) <= 1e-4
    xb = torch.stack([x1, x2, x1 + x2])
    yb = torch.zeros((3, 8192), device=gpu)
    H.call_op(qc, lay, xb, yb, "hybrid", True)
    ref = torch.stack([y1, y2, op(x1 + x2)])
    assert float((yb - ref).abs().max() / ref.abs().max()) <= 1e-5


def test_balanced_names_equal_spmv(qc, gpu):
    import torch

    for bits in (3, 4):
        case = H.make_case(bits, 256, 128, sparse=0.03, seed=21)
        t = H.to_torch(case, gpu)
        x = torch.randn(256, device=gpu)
        y = torch.zeros(128, device=gpu)
        sr, nt, _ = H.oracle.startrows_balanced(case["rows"], 128, case["vals"].size)
        fn = getattr(qc, f"vecquant{bits}matmul_spmv_balanced_nuq_perchannel")
        fn(t["rows"], t["cols"], torch.from_numpy(sr).to(gpu), t["vals"], x, y, t["qweight"], t["lookup_table"],
           128, nt, case["vals"].size)
        ref = H.oracle_ref(case, x.cpu().numpy(), np.zeros(128, np.float32), "spmv")
        assert H.rel_err(y.cpu().numpy(), ref) <= TOL_FP64


def test_argument_validation_on_gpu(qc, gpu):
    import torch

    case = H.make_case(4, 128, 128, seed=2)
    t = H.to_torch(case, gpu)
    x, y = torch.randn(128, device=gpu), torch.zeros(128, device=gpu)
    with pytest.raises(TypeError):
        qc.vecquant4matmul_nuq_perchannel(x.half(), t["qweight"], y, t["lookup_table"])
    with pytest.raises(ValueError):
        qc.vecquant4matmul_nuq_perchannel(x[:64], t["qweight"], y, t["lookup_table"])
    with pytest.raises(ValueError):
        qc.vecquant3matmul_nuq_perchannel(x, t["qweight"], y, t["lookup_table"])  # 16 rows % 3 != 0
    with pytest.raises(RuntimeError):
        qc.vecquant4matmul_nuq_perchannel(x.cpu(), t["qweight"], y, t["lookup_table"])
    with pytest.raises(ValueError):
        qc.vecquant4matmul_nuq_perchannel(x, t["qweight"].t().contiguous().t(), y, t["lookup_table"])


def test_against_reference_kernels_live(gpu):
    """When oracle/_ref/libsqllm_ref.so (the reference's own kernels, compiled unmodified by
    oracle/build_ref.sh) travelled to this box, run it on the same operands: our result, the
    reference's result and the oracle must agree."""
    import ctypes
    import os

    import torch

    from squeezellm_amd import quant_cuda as qcm

    path = os.path.join(H.ROOT, "oracle", "_ref", "libsqllm_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libsqllm_ref.so not built (needs /root/reference at build time)")
    ref = ctypes.CDLL(path)
    P = ctypes.c_void_p
    for bits in (3, 4):
        for batch in (0, 3):
            K, N = 1024, 512  # reference needs K % 128 == 0 and N % 128 == 0
            case = H.make_case(bits, K, N, sparse=0.01, topX=10, heavy_rows=2, seed=31 + bits)
            t = H.to_torch(case, gpu)
            rng = np.random.default_rng(5)
            x = torch.from_numpy(rng.normal(size=(batch, K) if batch else (K,)).astype(np.float32)).to(gpu)
            mul0 = torch.from_numpy(rng.normal(size=(batch, N) if batch else (N,)).astype(np.float32)).to(gpu)
            ours, theirs = mul0.clone(), mul0.clone()
            H.call_op(qcm, t, x, ours, "hybrid", batch > 0)
            rc = ref.refk_hybrid(bits, batch, P(t["rows"].data_ptr()), P(t["cols"].data_ptr()), P(t["vals"].data_ptr()),
                                 case["vals"].size, P(x.data_ptr()), P(t["full_rows"].data_ptr()),
                                 P(t["full_row_indices"].data_ptr()), 10, P(theirs.data_ptr()), N,
                                 P(t["qweight"].data_ptr()), P(t["lookup_table"].data_ptr()), K, N)
            assert rc == 0
            torch.cuda.synchronize()
            want = H.oracle_ref(case, x.cpu().numpy(), mul0.cpu().numpy(), "hybrid")
            assert H.rel_err(theirs.cpu().numpy(), want) <= TOL_FP64  # pins the oracle to the reference
            assert H.rel_err(ours.cpu().numpy(), want) <= TOL_FP64
            assert H.rel_err(ours.cpu().numpy(), theirs.cpu().numpy()) <= TOL_FP64


def _csr_case(bits, K, N, row_lengths, seed):
    """A dense term plus a CSR with exactly the given number of non-zeros in each row (columns sorted per row)."""
    case = H.make_case(bits, K, N, seed=seed)
    rng = np.random.default_rng(seed + 1)
    rows = np.zeros(N + 1, np.int32)
    rows[1:] = np.cumsum(row_lengths)
    cols = np.concatenate([np.sort(rng.choice(K, size=int(n), replace=False)) for n in row_lengths] or [np.zeros(0)]).astype(np.int32)
    vals = rng.normal(0, 0.1, cols.size).astype(np.float32)
    case.update(rows=rows, cols=cols, vals=vals)
    return case


@pytest.mark.parametrize("bits", [3, 4])
def test_csr_row_segment_structures(qc, gpu, bits):
    """The CSR role keeps two consecutive non-zeros per lane and scans the lanes' open row segments:
    every way a row can meet a lane, a 16-lane DPP row, a wave and a chunk boundary -- rows of one non-zero (a row
    end inside every lane), rows of two that start on odd positions, one row longer than several chunks, runs of
    empty rows, and every total around the 64 / 128 / 1024 marks (the last lane half filled)."""
    K, N = 2048, 192
    patterns = {
        "one each": np.ones(N, int),
        "two each, shifted by one": np.r_[1, np.full(N - 1, 2)],
        "three each": np.full(N, 3),
        "one row holds 2500": np.r_[np.zeros(7, int), 2500 if K >= 2500 else K, np.zeros(N - 8, int)],
        "heavy row between singles": np.r_[np.ones(50, int), 1500, np.ones(N - 51, int)],
        "empty runs": np.where(np.arange(N) % 5 == 0, 37, 0),
        "ramp": np.arange(N) % 40,
    }
    for total in (1, 2, 63, 64, 65, 127, 128, 129, 255, 257, 1023, 1024, 1025, 2047, 2049):
        lens = np.zeros(N, int)
        lens[: total // 11] = 11
        lens[total // 11] = total % 11
        patterns[f"{total} non-zeros in rows of 11"] = lens
    patterns["one row holds 2500"][7] = min(2500, K)
    for name, lens in patterns.items():
        case = _csr_case(bits, K, N, lens, seed=len(name))
        for batch in (0, 3, 8, 16, 40):  # fused kernel (1 / 4 / 8-row tiles), then the wide-batch sparse launch (lane groups / scalar walk)
            x, mul, got = run_op(qc, gpu, case, "spmv", batch)
            assert H.rel_err(got, H.oracle_ref(case, x, mul, "spmv")) <= TOL_FP64, (name, batch)


@pytest.mark.parametrize("K,N", [(4096, 4096), (4096, 11008), (11008, 4096)], ids=["4096x4096", "4096x11008", "11008x4096"])
@pytest.mark.parametrize("bits", [3, 4])
def test_all_reference_launchers_at_baseline_shapes(gpu, bits, K, N):
    """The three LLaMA-7B shapes of BASELINE configs[1] / [2] through ALL TWELVE reference launchers
    (squeezellm/quant_cuda_kernel.cu:132-738: dense / spmv / hybrid x matvec / batched x 3 / 4 bit; the shapes satisfy the
    reference's % 128 limits, :754, :841), compiled unmodified into oracle/_ref: the reference's kernels, ours and the C
    oracle on the same operands.  Pins the oracle to the reference at full size (the golden vectors are K = 256, N = 128),
    and is the correctness half of the same-box timing in profiles/r04_ref_vs_ours_*.json."""
    import ctypes
    import os

    import torch

    from squeezellm_amd import quant_cuda as qcm

    path = os.path.join(H.ROOT, "oracle", "_ref", "libsqllm_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libsqllm_ref.so not built (needs /root/reference at build time)")
    ref = ctypes.CDLL(path)
    ref.refk_set_sync(1)
    P = ctypes.c_void_p
    lib = H.c_oracle()
    case = H.make_case(bits, K, N, sparse=0.0045, topX=10, heavy_rows=10, seed=1000 + bits + K % 97)
    t = H.to_torch(case, gpu)
    rng = np.random.default_rng(17)
    for batch in (0, 3):
        x = torch.from_numpy(rng.normal(size=(batch, K) if batch else (K,)).astype(np.float32)).to(gpu)
        mul0 = torch.from_numpy((rng.normal(size=(batch, N) if batch else (N,)) * 0.01).astype(np.float32)).to(gpu)
        for kind in ("dense", "spmv", "hybrid"):
            ours, theirs = mul0.clone(), mul0.clone()
            H.call_op(qcm, t, x, ours, kind, batch > 0)
            if kind == "dense":
                rc = ref.refk_dense(bits, batch, P(x.data_ptr()), P(t["qweight"].data_ptr()), P(theirs.data_ptr()),
                                    P(t["lookup_table"].data_ptr()), K, N)
            elif kind == "spmv":
                rc = ref.refk_spmv(bits, batch, P(t["rows"].data_ptr()), P(t["cols"].data_ptr()), P(t["vals"].data_ptr()),
                                   case["vals"].size, P(x.data_ptr()), P(theirs.data_ptr()), N, P(t["qweight"].data_ptr()),
                                   P(t["lookup_table"].data_ptr()), K, N)
            else:
                rc = ref.refk_hybrid(bits, batch, P(t["rows"].data_ptr()), P(t["cols"].data_ptr()), P(t["vals"].data_ptr()),
                                     case["vals"].size, P(x.data_ptr()), P(t["full_rows"].data_ptr()),
                                     P(t["full_row_indices"].data_ptr()), 10, P(theirs.data_ptr()), N,
                                     P(t["qweight"].data_ptr()), P(t["lookup_table"].data_ptr()), K, N)
            assert rc == 0, (kind, batch, rc)
            torch.cuda.synchronize()
            sub = dict(case)
            if kind == "dense":
                sub.update(rows=None, cols=None, vals=None)
            if kind != "hybrid":
                sub.update(full_rows=None, full_row_indices=None)
            want = H.c_matvec(lib, sub, x.cpu().numpy(), mul0.cpu().numpy(), batched=batch > 0)
            what = f"w{bits} {K}x{N} {kind} batch {batch}"
            assert H.rel_err(theirs.cpu().numpy(), want) <= TOL_FP64, f"{what}: reference kernels vs oracle"
            assert H.rel_err(ours.cpu().numpy(), want) <= TOL_FP64, f"{what}: ours vs oracle"
            assert H.rel_err(ours.cpu().numpy(), theirs.cpu().numpy()) <= TOL_FP64, f"{what}: ours vs reference kernels"
