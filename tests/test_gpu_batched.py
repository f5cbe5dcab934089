"""GPU parity of the *_batched operators beyond the 8-row batch tiles: the matrix-core kernel
(batches of `mfma_min_batch` = 9 rows and more), at the batch sizes the reference's callers use --
PPL evaluation runs the batched ops with B = 2048 (/root/reference/llama.py:91-103 ->
squeezellm/quant.py:313-383) -- and at the BASELINE.json configurations that name a batch:
config 1 (OPT-1.3B shapes with bias, batch 1 x seq 128 -> B = 128) and config 4 (LLaMA-13B shapes,
0.45 % sparse + top-10, batch 1..8 and beyond).  Same tolerances as tests/test_gpu_parity.py.
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


def run_batched(qc, gpu, case, kind, batch, seed=1, entry="module"):
    import torch

    rng = np.random.default_rng(seed)
    x = rng.normal(size=(batch, case["K"])).astype(np.float16).astype(np.float32)
    mul = rng.normal(0, 0.5, size=(batch, case["N"])).astype(np.float32)
    t = H.to_torch(case, gpu)
    yt = torch.from_numpy(mul).to(gpu)
    H.call_op(qc, t, torch.from_numpy(x).to(gpu), yt, kind, True, entry=entry)
    torch.cuda.synchronize()
    return x, mul, yt.cpu().numpy()


@pytest.mark.parametrize("bits,K,N", [(4, 256, 192), (3, 96 * 2, 260), (4, 1024, 132), (3, 1024, 776), (4, 32, 4), (3, 32, 8)])
@pytest.mark.parametrize("kind", ["dense", "spmv", "hybrid"])
@pytest.mark.parametrize("batch", [9, 16, 17, 33, 64, 65, 130])
@pytest.mark.parametrize("entry", H.ENTRIES)
def test_wide_batches_vs_oracle(qc, gpu, bits, K, N, kind, batch, entry):
    """Through the module (caller workspace), the header's named `*_batched` symbols and sqllm_launch_ws(NULL) (both: the
    library's stream-ordered scratch).  One, two and four row blocks of 16, several passes of 64 rows, ragged ends in every dimension
    (N not a multiple of the 64-column tile, K = 32: a single unit, fewer units than lane rows)."""
    case = H.make_case(bits, K, N, sparse=0.03 if kind != "dense" else 0, topX=3 if kind == "hybrid" else 0,
                       heavy_rows=1 if kind != "dense" and N >= 8 else 0, seed=bits * 1000 + K + N)
    x, mul, got = run_batched(qc, gpu, case, kind, batch, entry=entry)
    assert H.rel_err(got, H.oracle_ref(case, x, mul, kind)) <= TOL_FP64


@pytest.mark.parametrize("options", [dict(mfma_split=0), dict(mfma_fuse_small=0), dict(mfma_split=0, mfma_fuse_small=0), dict(mfma_fuse_sparse=0)],
                         ids=["fp32-instruction", "split-unfused", "fp32-unfused", "sparse-launch-of-its-own"])
@pytest.mark.parametrize("bits,K,N", [(4, 1024, 132), (3, 1024, 776)])
@pytest.mark.parametrize("batch", [9, 16, 40, 130])
def test_wide_batch_routes_behind_options(qc, gpu, options, bits, K, N, batch):
    """The routes the defaults no longer take stay correct: the fp32 matrix instruction (option mfma_split = 0: bit for bit an
    fp32 FMA chain) and one launch per op + its sparse launch up to 16 rows (mfma_fuse_small = 0).  Also: the split kernel's
    result agrees with the fp32 instruction's to fp32 round-off -- its operands are split exactly, six of nine partial
    products kept (csrc/sqllm_mfma_split.hip)."""
    from squeezellm_amd import _lib

    case = H.make_case(bits, K, N, sparse=0.03, topX=3, heavy_rows=1, seed=bits * 100 + batch)
    x, mul, want = run_batched(qc, gpu, case, "hybrid", batch)  # the default route
    try:
        for k, v in options.items():
            _lib.set_option(k, v)
        _, _, got = run_batched(qc, gpu, case, "hybrid", batch)
    finally:
        for k in options:
            _lib.set_option(k, 1)
    ref = H.oracle_ref(case, x, mul, "hybrid")
    assert H.rel_err(got, ref) <= TOL_FP64
    assert H.rel_err(got, want) <= 2e-6, "split operands vs fp32 matrix instruction / fused vs unfused launch"


@pytest.mark.parametrize("bits", [3, 4])
def test_ppl_eval_batch_2048(qc, gpu, bits):
    """B = 2048, the batch the reference's perplexity evaluation feeds the batched ops."""
    case = H.make_case(bits, 512, 320, sparse=0.01, topX=4, heavy_rows=2, seed=77)
    x, mul, got = run_batched(qc, gpu, case, "hybrid", 2048)
    assert H.rel_err(got, H.oracle_ref(case, x, mul, "hybrid")) <= TOL_FP64


@pytest.mark.parametrize("vec", ["fp16-born", "fp32"])
@pytest.mark.parametrize("planes", [True, False], ids=["planes", "in-register-split"])
@pytest.mark.parametrize("bits,K,N", [(4, 1024, 1092), (3, 1024, 776), (4, 4160, 516), (3, 2080, 1028), (4, 32, 4), (3, 32, 8)])
@pytest.mark.parametrize("batch", [17, 64, 130, 700])
def test_wide_form_vs_oracle(qc, gpu, vec, planes, bits, K, N, batch):
    """The wide form of the split matrix-core kernel (csrc/sqllm_mfma_wide.hip: sqllm_fused_wide; default once batch * K * N >= 5.7e9),
    forced from 17 rows up: workgroups of eight column tiles, vec from bf16 planes in fragment order (sqllm_split_vec) or --
    without scratch -- split in registers; fp16-born vec takes the five-product path (no lo plane), fp32 vec all six.
    Shapes: a last column group of one tile and 4 columns (1092), K that is not a whole number of 32-k steps times
    anything convenient (4160 = 130 x 32), a single unit, row blocks with 1..64 live rows.  On the 256-CU part the units
    (64 rows x 8 column tiles) of these shapes fit one round, so every workgroup is a K slice that adds atomically;
    test_wide_form_whole_rounds covers the read-add-write epilogue of unsliced units."""
    import torch

    from squeezellm_amd import _lib

    case = H.make_case(bits, K, N, sparse=0.02, topX=3, heavy_rows=1 if N >= 8 else 0, seed=bits * 7 + K + N + batch)
    rng = np.random.default_rng(batch)
    x = rng.normal(size=(batch, K)).astype(np.float32)
    if vec == "fp16-born":
        x = x.astype(np.float16).astype(np.float32)
    mul = rng.normal(0, 0.5, size=(batch, N)).astype(np.float32)
    t = H.to_torch(case, gpu)
    yt = torch.from_numpy(mul).to(gpu)
    try:
        _lib.set_option("mfma_wide_min_batch", 17)
        _lib.set_option("split_planes_min_batch", 1 if planes else 1 << 30)
        H.call_op(qc, t, torch.from_numpy(x).to(gpu), yt, "hybrid", True)
        torch.cuda.synchronize()
    finally:
        _lib.set_option("mfma_wide_min_batch", 0)
        _lib.set_option("split_planes_min_batch", 0)
    assert H.rel_err(yt.cpu().numpy(), H.oracle_ref(case, x, mul, "hybrid")) <= TOL_FP64


@pytest.mark.parametrize("bits", [3, 4])
def test_wide_form_many_row_blocks(qc, gpu, bits):
    """4100 rows (65 blocks of 64, the last with 4 live rows) x 5 column groups = 325 units on the real CU count: a whole round
    of unsliced units and a sliced tail; hybrid op, so the sparse launch takes its 128-row blocks too."""
    from squeezellm_amd import _lib

    case = H.make_case(bits, 512, 2500, sparse=0.01, topX=4, heavy_rows=2, seed=91 + bits)
    try:
        _lib.set_option("mfma_wide_min_batch", 64)
        x, mul, got = run_batched(qc, gpu, case, "hybrid", 4100)
    finally:
        _lib.set_option("mfma_wide_min_batch", 0)
    assert H.rel_err(got, H.oracle_ref(case, x, mul, "hybrid")) <= TOL_FP64


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("cus", [8, 24])
def test_wide_form_whole_rounds(qc, gpu, bits, cus):
    """Planned for a part of 8 / 24 CUs (option cu_count), 3 row blocks x 9 column groups = 27 units make whole rounds of
    unsliced units (16-byte read-add-write into mul, which starts non-zero) plus a sliced, atomically adding tail."""
    import torch

    from squeezellm_amd import _lib

    K, N, batch = 2048, 4608 - 60, 150
    case = H.make_case(bits, K, N, seed=bits + cus)
    try:
        _lib.set_option("cu_count", cus)
        _lib.set_option("mfma_wide_min_batch", 17)
        p = _lib.plan_query(bits, K, N, batch=batch)
        assert p["k_slices"] > 1 and p["dense_blocks"] == 27 // cus * cus + (27 % cus) * p["k_slices"]
        x, mul, got = run_batched(qc, gpu, case, "dense", batch)
    finally:
        _lib.set_option("cu_count", 0)
        _lib.set_option("mfma_wide_min_batch", 0)
    assert H.rel_err(got, H.oracle_ref(case, x, mul, "dense")) <= TOL_FP64


@pytest.mark.parametrize("mode", ["caller-workspace", "scratch-nodes", "no-scratch"])
def test_wide_form_in_a_captured_graph(gpu, mode):
    """A wide-form hybrid op captured into a graph and replayed twice.  "caller-workspace": the pass owns a workspace
    (sqllm_launch_groups_ws) -- kernel nodes only; "scratch-nodes": the workspace-less entry point with scratch_in_capture = 1
    -- the graph carries the group's scratch block as an allocation and a free node around the transpose / split / sparse /
    dense / reduce launches (node types asserted in tests/test_gpu_workspace.py); "no-scratch": workspace-less with
    scratch_in_capture = 0 -- allocation-free: the CSR role gathers from vec, the wide form (forced here) splits vec in
    registers and its K slices add atomically."""
    import torch

    from squeezellm_amd import _lib, decode

    K, N, batch = 1024, 1092, 130
    case = H.make_case(4, K, N, sparse=0.02, topX=3, heavy_rows=1, seed=5)
    lay = dict(H.to_torch(case, gpu), K=K, N=N, bits=4)
    rng = np.random.default_rng(3)
    x = rng.normal(size=(batch, K)).astype(np.float16).astype(np.float32)
    mul = rng.normal(0, 0.5, size=(batch, N)).astype(np.float32)
    xt, y0 = torch.from_numpy(x).to(gpu), torch.from_numpy(mul).to(gpu)
    yt = y0.clone()
    try:
        _lib.set_option("mfma_wide_min_batch", 17)
        _lib.set_option("scratch_in_capture", 0 if mode == "no-scratch" else 1)
        seq = decode.OpSequence([lay], [xt], [yt], batched=True, workspace=mode == "caller-workspace")
        assert (seq._ws is not None) == (mode == "caller-workspace")
        g = seq.graph(warmup=1)
        for _ in range(2):
            yt.copy_(y0)
            g.replay()
            torch.cuda.synchronize()
            assert H.rel_err(yt.cpu().numpy(), H.oracle_ref(case, x, mul, "hybrid")) <= TOL_FP64
    finally:
        _lib.set_option("mfma_wide_min_batch", 0)
        _lib.set_option("scratch_in_capture", 1)


@pytest.mark.parametrize("vec", ["fp16-born", "fp32"])
@pytest.mark.parametrize("bits", [3, 4])
def test_wide_form_at_13b_size_agrees_with_the_batch1_operator(qc, gpu, bits, vec):
    """BASELINE configs[3] shape at a size the CPU oracle cannot check in seconds: LLaMA-13B gate_proj (5120 x 13824), w + 0.45 %
    CSR outliers + top-10 rows, 600 rows -- the wide form by the default routing (ten 64-row blocks, the last with 24 live
    rows; whole rounds + a sliced tail with slabs; sparse terms as launches of their own).  Sampled rows against the
    batch-1 hybrid operator on the same operands, which the oracle pins at this shape (test_gpu_parity.py)."""
    import torch

    from squeezellm_amd import _lib, synth

    K, N, B = 5120, 13824, 600
    assert _lib.plan_query(bits, K, N, batch=B)["grid_y"] == 1  # (the wide form's 1-D grid)
    lay = synth.make_layer(K, N, bits, sparse_frac=0.0045, topX=10, heavy_rows=10, device=gpu, seed=31 + bits)
    g = torch.Generator(device=gpu).manual_seed(5)
    x = torch.randn((B, K), device=gpu, generator=g, dtype=torch.float16).float() if vec == "fp16-born" else torch.randn((B, K), device=gpu, generator=g)
    y0 = torch.randn((B, N), device=gpu, generator=g) * 0.01
    y = y0.clone()
    name = f"vecquant{bits}matmul_spmv_hybrid_nuq_perchannel"
    args = (lay["rows"], lay["cols"], lay["vals"])
    getattr(qc, name + "_batched")(*args, x, lay["full_rows"], lay["full_row_indices"], y, N, lay["qweight"], lay["lookup_table"])
    sample = (0, 63, 64, 300, 575, 599)
    for r in sample:
        yr = y0[r].clone()
        getattr(qc, name)(*args, x[r].contiguous(), lay["full_rows"], lay["full_row_indices"], yr, N, lay["qweight"], lay["lookup_table"])
        torch.cuda.synchronize()
        err = float((y[r] - yr).abs().max() / yr.abs().max())
        assert err <= 2e-5, (r, err)
    # ... and the same rows against the ORACLE itself (the C restatement, fp64 accumulation: seconds for six rows)
    case = dict(K=K, N=N, bits=bits, **{k: lay[k].cpu().numpy() for k in ("qweight", "lookup_table", "rows", "cols", "vals", "full_rows", "full_row_indices")})
    idx = list(sample)
    ref = H.c_matvec(H.c_oracle(), case, x[idx].cpu().numpy(), y0[idx].cpu().numpy(), True)
    assert H.rel_err(y[idx].cpu().numpy(), ref) <= TOL_FP64


@pytest.mark.parametrize("vec", ["fp32", "fp16-born", "one-fp32-value"])
@pytest.mark.parametrize("bits,K,N", [(4, 1024, 132), (3, 1024, 776), (4, 5120, 320)])
@pytest.mark.parametrize("batch", [5, 9, 16])
def test_fused_small_launch_with_planes(qc, gpu, bits, K, N, batch, vec):
    """The fused small launch with a workspace loads vec already split into bf16 planes (written with the transposed copy by the
    kernel in front): full-precision vec, fp16-born vec (all lo parts zero) and a mix, at 5 / 9 / 16 rows, incl. K = 5120 (161 k
    blocks) -- against the oracle.  (A five-product path for fp16-born vec, switched by a flag from that kernel, was built on this
    test and measured slower than six products: profiles/r05_small_planes_five_products.txt.)"""
    import torch

    case = H.make_case(bits, K, N, sparse=0.02, topX=3, heavy_rows=1, seed=bits * 7 + batch)
    rng = np.random.default_rng(batch + K)
    x = rng.normal(size=(batch, K)).astype(np.float32)
    if vec != "fp32":
        x = x.astype(np.float16).astype(np.float32)
    if vec == "one-fp32-value":
        x[batch - 1, K - 3] = np.float32(0.123456789)
    mul = rng.normal(0, 0.5, size=(batch, N)).astype(np.float32)
    from squeezellm_amd import _lib

    t = H.to_torch(case, gpu)
    yt = torch.from_numpy(mul.copy()).to(gpu)
    try:
        _lib.set_option("mfma_min_batch", 5)  # (explicit: by default such small single ops stay on the batch tiles up to 8 rows)
        H.call_op(qc, t, torch.from_numpy(x).to(gpu), yt, "hybrid", True)
        torch.cuda.synchronize()
    finally:
        _lib.set_option("mfma_min_batch", 0)
    assert H.rel_err(yt.cpu().numpy(), H.oracle_ref(case, x, mul, "hybrid")) <= TOL_FP64


@pytest.fixture
def small_launch_from_5_rows():
    """the fused small launch from 5 rows whatever the shape (the default keeps 5 / 6 rows, and small single ops up to 8, on the batch tiles)"""
    from squeezellm_amd import _lib

    _lib.set_option("mfma_min_batch", 5)
    yield
    _lib.set_option("mfma_min_batch", 0)


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("batch", [5, 8, 16])
def test_fused_small_launch_sparse_edge_cases(qc, gpu, bits, batch, small_launch_from_5_rows):
    """The edge cases of tests/test_gpu_parity.py::test_sparse_edge_cases on the fused small launch, whose dense workgroups
    walk the CSR rows of their own tile (csr_tile_fold_staged): an empty CSR, empty rows at both ends and in the middle beside
    rows that hold most of the non-zeros, duplicate top-X indices, one tile's share larger than a staging pass (4096 non-zeros:
    several passes), a matrix with more rows than non-zeros (most tiles' shares empty), and accumulation into a non-zero mul."""
    K, N = 512, 256
    case = H.make_case(bits, K, N, seed=3)  # (a) nnz == 0 with the CSR operands present
    case.update(rows=np.zeros(N + 1, np.int32), cols=np.zeros(0, np.int32), vals=np.zeros(0, np.float32))
    x, mul, got = run_batched(qc, gpu, case, "spmv", batch)
    assert H.rel_err(got, H.oracle_ref(case, x, mul, "spmv")) <= TOL_FP64
    case = H.make_case(bits, K, N, sparse=0.01, heavy_rows=2, empty_rows=(0, 1, 100, N - 1), seed=4)  # (b)
    for kind in ("spmv", "hybrid") if case.get("full_rows") is not None else ("spmv",):
        x, mul, got = run_batched(qc, gpu, case, kind, batch)
        assert H.rel_err(got, H.oracle_ref(case, x, mul, kind)) <= TOL_FP64
    case = H.make_case(bits, K, N, sparse=0.01, topX=4, dup_topx=True, seed=5)  # (c) duplicate full_row_indices accumulate
    x, mul, got = run_batched(qc, gpu, case, "hybrid", batch)
    assert H.rel_err(got, H.oracle_ref(case, x, mul, "hybrid")) <= TOL_FP64
    case = H.make_case(bits, 4096, 128, sparse=0.05, heavy_rows=3, seed=6)  # (d) ~26 k non-zeros in two tiles: several staging passes each
    assert case["vals"].size > 2 * 2 * 4096
    x, mul, got = run_batched(qc, gpu, case, "spmv", batch)
    assert H.rel_err(got, H.oracle_ref(case, x, mul, "spmv")) <= TOL_FP64
    case = H.make_case(bits, 128, 8192, sparse=0.0008, topX=2, seed=8)  # (e) ~840 non-zeros over 128 tiles
    x, mul, got = run_batched(qc, gpu, case, "hybrid", batch)
    assert H.rel_err(got, H.oracle_ref(case, x, mul, "hybrid")) <= TOL_FP64


def _routing(mfma_min, cols_min, cols_max):
    from squeezellm_amd import _lib

    _lib.set_option("mfma_min_batch", mfma_min)
    _lib.set_option("cols_min_batch", cols_min)
    _lib.set_option("cols_max_batch", cols_max)


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("kind", ["dense", "hybrid"])
@pytest.mark.parametrize("shape", [(256, 192), (1024, 776), (2048, 1024), (32, 4), (96, 68)])
def test_column_lane_kernel_vs_oracle(qc, gpu, bits, kind, shape):
    """The column-lane kernel (default for 2..4 rows) forced for every batch size, ragged shapes included
    (N not a multiple of the 64-column tile, K of one 3-bit unit, ranges that cross tile boundaries)."""
    import torch

    K, N = shape
    case = H.make_case(bits, K, N, sparse=0.02 if kind == "hybrid" else 0, topX=3 if kind == "hybrid" else 0,
                       heavy_rows=1 if kind == "hybrid" else 0, seed=K + N + bits)
    t = H.to_torch(case, gpu)
    try:
        _routing(1 << 30, 1, 1 << 30)
        for B in (1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 13, 14, 15, 16, 23):  # passes of every width 1..8, alone and behind an 8-row pass
            rng = np.random.default_rng(B)
            x = rng.normal(size=(B, K)).astype(np.float32)
            mul = rng.normal(size=(B, N)).astype(np.float32)
            y = torch.from_numpy(mul.copy()).to(gpu)
            H.call_op(qc, t, torch.from_numpy(x).to(gpu), y, kind, True)
            torch.cuda.synchronize()
            assert H.rel_err(y.cpu().numpy(), H.oracle_ref(case, x, mul, kind)) <= TOL_FP64, (B, K, N)
    finally:
        _routing(0, 0, 0)


@pytest.mark.parametrize("entry", ["module", "named"])
@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("kind", ["dense", "spmv", "hybrid"])
@pytest.mark.parametrize("shape", [(256, 192), (1024, 776), (32, 4), (96, 68)])
def test_batch_tiles_every_row_count(qc, gpu, bits, kind, shape, entry):
    """The batch tiles of the batch-1 kernel (sqllm_fused_matvec<BITS, BT>) forced for every batch size: tiles of exactly
    1, 2, 3, 4, 5, 6 and 8 rows (7 rows ride the 8-row tile, 9+ go through in passes of 8) -- round 6 added the 3-, 5- and
    6-row instantiations -- ragged shapes included, accumulating into a non-zero mul."""
    import torch

    K, N = shape
    case = H.make_case(bits, K, N, sparse=0.03 if kind != "dense" else 0, topX=3 if kind == "hybrid" else 0,
                       heavy_rows=1 if kind != "dense" and N >= 8 else 0, seed=K + N + bits)
    t = H.to_torch(case, gpu)
    try:
        _routing(1 << 30, 1 << 30, 1 << 30)
        for B in (1, 2, 3, 4, 5, 6, 7, 8, 11, 13, 14):
            rng = np.random.default_rng(B)
            x = rng.normal(size=(B, K)).astype(np.float32)
            mul = rng.normal(size=(B, N)).astype(np.float32)
            y = torch.from_numpy(mul.copy()).to(gpu)
            H.call_op(qc, t, torch.from_numpy(x).to(gpu), y, kind, True, entry=entry)
            torch.cuda.synchronize()
            assert H.rel_err(y.cpu().numpy(), H.oracle_ref(case, x, mul, kind)) <= TOL_FP64, (B, K, N)
    finally:
        _routing(0, 0, 0)


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("kind", ["dense", "hybrid"])
def test_column_lane_kernel_runs_groups(gpu, bits, kind):
    """q/k/v-style groups (one vec, ops of DIFFERENT, ragged widths) as ONE launch of the column-lane kernel -- the
    workgroups divided between the ops through the block table -- against the oracle, and with the option that
    keeps groups on the batch tiles."""
    import torch

    from squeezellm_amd import _lib, decode

    K, Ns = 1024, (776, 132, 260, 64)
    cases = [H.make_case(bits, K, N, sparse=0.02 if kind == "hybrid" else 0, topX=3 if kind == "hybrid" else 0,
                         heavy_rows=1 if kind == "hybrid" else 0, seed=N + bits) for N in Ns]
    lays = []
    for c in cases:
        t = H.to_torch(c, gpu)
        lays.append(dict(bits=bits, K=K, N=c["N"], qweight=t["qweight"], lookup_table=t["lookup_table"], rows=t.get("rows"),
                         cols=t.get("cols"), vals=t.get("vals"), full_rows=t.get("full_rows"), full_row_indices=t.get("full_row_indices")))
    try:
        _routing(1 << 30, 1, 1 << 30)
        for B in (1, 2, 3, 4, 5, 6, 7, 8, 9, 16):
            rng = np.random.default_rng(B)
            x = rng.normal(size=(B, K)).astype(np.float32)
            xt = torch.from_numpy(x).to(gpu)
            refs = [H.oracle_ref(c, x, np.zeros((B, c["N"]), np.float32), kind) for c in cases]
            for groups in (1, 0):
                _lib.set_option("cols_groups", groups)
                for n in (2, 3, 4):
                    ys = [torch.zeros((B, c["N"]), device=gpu) for c in cases[:n]]
                    seq = decode.OpSequence(lays[:n], [xt] * n, ys, batched=True, fuse_shared_input=True)
                    assert [len(g) for g in seq.groups] == [n]
                    seq.launch()
                    torch.cuda.synchronize()
                    for y, r in zip(ys, refs):
                        assert H.rel_err(y.cpu().numpy(), r) <= TOL_FP64, (B, groups, n)
    finally:
        _lib.set_option("cols_groups", 1)
        _routing(0, 0, 0)


@pytest.mark.parametrize("bits", [3, 4])
def test_three_batched_paths_agree(qc, gpu, bits):
    """Batch tiles, column-lane kernel and matrix-core kernel are three implementations of one operator."""
    import torch

    case = H.make_case(bits, 2048, 1024, sparse=0.0045, topX=10, heavy_rows=4, seed=5)
    t = H.to_torch(case, gpu)
    for B in (3, 24):
        x = torch.randn((B, 2048), device=gpu)
        ref = H.oracle_ref(case, x.cpu().numpy(), np.zeros((B, 1024), np.float32), "hybrid")
        outs = []
        try:
            for routing in ((1 << 30, 1 << 30, 1 << 30), (1 << 30, 1, 1 << 30), (1, 1 << 30, 1 << 30)):
                _routing(*routing)
                y = torch.zeros((B, 1024), device=gpu)
                H.call_op(qc, t, x, y, "hybrid", True)
                torch.cuda.synchronize()
                outs.append(y.cpu().numpy())
                assert H.rel_err(outs[-1], ref) <= TOL_FP64, (B, routing)  # every leg against the oracle, not against leg 0
        finally:
            _routing(0, 0, 0)
        assert H.rel_err(outs[1], outs[0]) <= 1e-5 and H.rel_err(outs[2], outs[0]) <= 1e-5


@pytest.mark.parametrize("bits", [3, 4])
def test_both_batched_paths_agree(qc, gpu, bits):
    """The 8-row batch tiles and the matrix-core kernel are two implementations of one operator."""
    import torch

    from squeezellm_amd import _lib

    case = H.make_case(bits, 2048, 1024, sparse=0.0045, topX=10, heavy_rows=4, seed=5)
    t = H.to_torch(case, gpu)
    x = torch.randn((24, 2048), device=gpu)
    ref = H.oracle_ref(case, x.cpu().numpy(), np.zeros((24, 1024), np.float32), "hybrid")
    outs = []
    try:
        for min_batch in (1 << 30, 1):
            _lib.set_option("mfma_min_batch", min_batch)
            y = torch.zeros((24, 1024), device=gpu)
            H.call_op(qc, t, x, y, "hybrid", True)
            torch.cuda.synchronize()
            outs.append(y.cpu().numpy())
            assert H.rel_err(outs[-1], ref) <= TOL_FP64, min_batch  # each leg against the oracle
    finally:
        _lib.set_option("mfma_min_batch", 0)
    assert H.rel_err(outs[1], outs[0]) <= 1e-5


LLAMA13B = [(5120, 5120), (5120, 13824), (13824, 5120)]


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("K,N", LLAMA13B)
@pytest.mark.parametrize("batch", [2, 5, 8, 16])
@pytest.mark.parametrize("entry", H.ENTRIES)
def test_llama13b_shapes_batched_hybrid(qc, gpu, bits, K, N, batch, entry):
    """BASELINE config 4 at full size on all three 13B shapes, against the C oracle (batch 2: batch tiles / column-lane
    kernel; 5: the 5-row batch tile; 8: the fused small launch at 4 bits -- the 8-row tile for 5120 x 5120, a small op alone in
    its launch --, tiles / column-lane kernel at 3 bits; 16: the fused small launch), through all three entries."""
    case = H.make_case(bits, K, N, sparse=0.0045, topX=10, heavy_rows=10, seed=13)
    x, mul, got = run_batched(qc, gpu, case, "hybrid", batch, entry=entry)
    ref = H.c_matvec(H.c_oracle(), case, x, mul, batched=True)
    assert H.rel_err(got, ref) <= TOL_FP64


OPT13 = [(2048, 2048), (2048, 8192), (8192, 2048)]


@pytest.mark.parametrize("K,N", OPT13)
def test_opt13b_config1_batch128_with_bias(gpu, K, N):
    """BASELINE config 1: OPT-1.3B w4 dense-only, batch 1 x seq 128 -> the batched branch of
    QuantLinearLUT.forward with B = 128 and a bias added AFTER the cast back to the input dtype
    (squeezellm/quant.py:313-321, :380-383; shapes and bias: models/opt-1.3b/config.json)."""
    import torch

    from squeezellm_amd import quant

    case = H.make_case(4, K, N, seed=K + N)
    rng = np.random.default_rng(3)
    case["bias"] = rng.normal(0, 0.01, N).astype(np.float32)
    mod = quant.QuantLinearLUT.from_operands(H.to_torch(case, gpu))
    x = rng.normal(size=(1, 128, K)).astype(np.float16)
    y = mod(torch.from_numpy(x).to(gpu))
    torch.cuda.synchronize()
    ref = H.oracle.quantlinear_forward(x, case)
    # `out.to(fp16) + bias` with the fp32 bias buffer: fp32 result carrying one fp16 rounding (quant.py:380-382)
    assert y.shape == (1, 128, N) and y.dtype == torch.float32 and ref.dtype == np.float32
    assert H.rel_err(y.float().cpu().numpy(), ref.astype(np.float32)) <= 2e-3


LLAMA65B = [(8192, 8192), (8192, 22016), (22016, 8192)]


@pytest.mark.parametrize("K,N", LLAMA65B)
def test_llama65b_shapes_vs_c_oracle(qc, gpu, K, N):
    """BASELINE config 5 (single-GPU leg of the sharded model): the three 65B shapes, w3 + 0.45 %
    sparse + top-10, batch-1 hybrid op against the C oracle."""
    import torch

    case = H.make_case(3, K, N, sparse=0.0045, topX=10, heavy_rows=10, seed=65)
    rng = np.random.default_rng(2)
    x = rng.normal(size=K).astype(np.float16).astype(np.float32)
    mul = rng.normal(0, 0.5, N).astype(np.float32)
    t = H.to_torch(case, gpu)
    yt = torch.from_numpy(mul).to(gpu)
    H.call_op(qc, t, torch.from_numpy(x).to(gpu), yt, "hybrid", False)
    torch.cuda.synchronize()
    ref = H.c_matvec(H.c_oracle(), case, x, mul, batched=False)
    assert H.rel_err(yt.cpu().numpy(), ref) <= TOL_FP64


@pytest.mark.parametrize("bits", [3, 4])
def test_wide_batch_sparse_term_transposed_or_gathered(qc, gpu, bits):
    """The CSR term of a wide-batch op reads a transposed copy of vec (default) or gathers from vec
    itself (option off; also what a capturing stream gets): same result, heavy rows spanning several
    waves and chunks included."""
    import torch

    from squeezellm_amd import _lib

    case = H.make_case(bits, 1024, 776, sparse=0.03, topX=6, heavy_rows=3, seed=11)
    t = H.to_torch(case, gpu)
    for B in (9, 64, 70, 200):
        rng = np.random.default_rng(B)
        x = rng.normal(size=(B, 1024)).astype(np.float32)
        mul = rng.normal(size=(B, 776)).astype(np.float32)
        want = H.oracle_ref(case, x, mul, "hybrid")
        xt = torch.from_numpy(x).to(gpu)
        try:
            for flag in (1, 0):
                _lib.set_option("sparse_transpose", flag)
                y = torch.from_numpy(mul.copy()).to(gpu)
                H.call_op(qc, t, xt, y, "hybrid", True)
                torch.cuda.synchronize()
                assert H.rel_err(y.cpu().numpy(), want) <= TOL_FP64, (B, flag)
        finally:
            _lib.set_option("sparse_transpose", 1)
        # captured through the header's NAMED entry point (workspace-less), with the scratch as graph memory nodes (default)
        # and without (the role gathers); and through the module, whose captured call takes its workspace as a temporary of
        # the captured region (kernel nodes only)
        try:
            for entry, in_capture in (("named", 1), ("named", 0), ("module", 1)):
                _lib.set_option("scratch_in_capture", in_capture)
                y = torch.from_numpy(mul.copy()).to(gpu)
                ystat = y.clone()
                s = torch.cuda.Stream(device=gpu)
                s.wait_stream(torch.cuda.current_stream(gpu))
                with torch.cuda.stream(s):
                    H.call_op(qc, t, xt, ystat.clone(), "hybrid", True, entry=entry)  # warm-up outside the capture
                torch.cuda.current_stream(gpu).wait_stream(s)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    H.call_op(qc, t, xt, ystat, "hybrid", True, entry=entry)
                ystat.copy_(y)
                g.replay()
                g.replay()  # (the graph owns its scratch: replays must not interfere; mul accumulates twice)
                torch.cuda.synchronize()
                want2 = H.oracle_ref(case, x, want, "hybrid")
                assert H.rel_err(ystat.cpu().numpy(), want2) <= TOL_FP64, (B, "graph", entry, in_capture)
        finally:
            _lib.set_option("scratch_in_capture", 1)
