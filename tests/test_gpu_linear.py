"""GPU tests of the fused linear (sqllm_linear_f16): fp16 in / fp16 out, bias, self-cleaning
workspace.  Checked against the oracle's fp64 matvec on the same fp16-rounded activations; the
result must equal the exact value rounded to fp16 up to a few fp32 accumulation ulps, i.e. within
one fp16 ulp of it."""
import ctypes

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _npl(lay):
    import torch

    return {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in lay.items()}


def _exact(npl, x16, kind):
    """fp64 result of the three terms + bias for fp16 activations x16 [rows, K]."""
    x = x16.astype(np.float64)
    ref = H.oracle_ref(npl, x if x.shape[0] > 1 else x[0], np.zeros((x.shape[0], npl["N"])) if x.shape[0] > 1 else np.zeros(npl["N"]), kind)
    ref = np.asarray(ref, np.float64).reshape(x.shape[0], npl["N"])
    if npl.get("bias") is not None:
        ref = ref + npl["bias"].astype(np.float64)
    return ref


def _check_fp16(got16, exact):
    got = got16.astype(np.float64)
    # one fp16 ulp at the magnitude of each element (fp16 has 11 significant bits) + tiny absolute slack
    tol = np.maximum(np.abs(exact), 2.0**-14) * 2.0**-10 + 1e-6
    bad = np.abs(got - exact) > tol
    assert not bad.any(), f"{bad.sum()} of {bad.size} outputs off by more than 1 fp16 ulp; worst {np.abs(got - exact).max()}"


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("kind", ["dense", "spmv", "hybrid"])
@pytest.mark.parametrize("rows", [1, 2, 5, 8, 19])
@pytest.mark.parametrize("bias", [False, True])
def test_fused_forward_matches_oracle_and_cleans_up(gpu, bits, kind, rows, bias):
    import torch

    from squeezellm_amd import quant, synth

    K, N = 1024, 456  # ragged last column tile (456 = 7 * 64 + 8)
    lay = synth.make_layer(K, N, bits, sparse_frac=0.0 if kind == "dense" else 0.01, topX=3 if kind == "hybrid" else 0,
                           heavy_rows=2 if kind != "dense" else 0, bias=bias, device=gpu, seed=31 * bits + rows)
    mod = quant.QuantLinearLUT.from_operands(lay)
    assert quant.fuse_quant_lut(mod) == 1 and type(mod) is quant.QuantLinearLUTFused
    g = torch.Generator(device=gpu).manual_seed(rows)
    x = torch.randn((rows, K), device=gpu, generator=g).half()
    npl = _npl(lay)
    exact = _exact(npl, x.cpu().numpy(), kind)
    for rep in range(3):  # the second and third call run on the workspace the previous one left behind
        y = mod(x if rows > 1 else x.reshape(1, 1, K))
        assert y.dtype == torch.float16 and y.shape[-1] == N
        _check_fp16(y.reshape(rows, N).cpu().numpy(), exact)
    ws = next(iter(mod._ws.values()))
    assert int(ws.count_nonzero()) == 0, "workspace must be left zero-filled"


@pytest.mark.parametrize("bits", [3, 4])
def test_fused_forward_equals_the_four_launch_path(gpu, bits):
    """Same layer through QuantLinearLUT.forward (zeros + float + op + cast) and through the fused
    kernel: both are fp16 roundings of fp32 sums of the same terms (summation order differs)."""
    import torch

    from squeezellm_amd import quant, synth

    K, N = 4096, 4096
    lay = synth.make_layer(K, N, bits, sparse_frac=0.0045, topX=10, heavy_rows=10, device=gpu, seed=5)
    ref_mod = quant.QuantLinearLUT.from_operands(lay)
    fused = quant.QuantLinearLUT.from_operands(lay)
    fused.__class__ = quant.QuantLinearLUTFused
    x = torch.randn((1, 1, K), device=gpu).half()
    a, b = ref_mod(x).float(), fused(x).float()
    assert a.shape == b.shape
    assert float((a - b).abs().max()) <= 2.0**-10 * float(a.abs().max()) * 1.01


def test_fuse_quant_lut_switches_children(gpu):
    import torch

    from squeezellm_amd import quant, synth

    lays = [synth.make_layer(256, 128, 4, device=gpu, seed=s) for s in range(2)]
    model = torch.nn.Sequential(*[quant.QuantLinearLUT.from_operands(l) for l in lays])
    keys = list(model.state_dict().keys())
    assert quant.fuse_quant_lut(model) == 2 and list(model.state_dict().keys()) == keys
    x = torch.randn((1, 256), device=gpu).half()
    y = model[0](x)
    assert y.dtype == torch.float16 and y.shape == (1, 128)
    # fp32 activations keep the operator path
    assert model[0](x.float()).dtype == torch.float32


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("batched", [False, True])
@pytest.mark.parametrize("fold", [True, False])
def test_linear_sequence_groups_and_graph(gpu, bits, batched, fold):
    """A decoder layer's seven linears as fused-linear groups (q/k/v and gate/up share a launch),
    eager and replayed from a HIP graph."""
    import torch

    from squeezellm_amd import decode, synth

    hidden, inter, B = 512, 1408, 4
    shapes = [(hidden, hidden)] * 4 + [(hidden, inter)] * 2 + [(inter, hidden)]
    lays = [synth.make_layer(K, N, bits, sparse_frac=0.005, topX=2, heavy_rows=1, bias=(i % 2 == 0), device=gpu, seed=i)
            for i, (K, N) in enumerate(shapes)]
    rows = B if batched else 1
    xh = torch.randn((rows, hidden), device=gpu).half()
    xi = torch.randn((rows, inter), device=gpu).half()
    xo = torch.randn((rows, hidden), device=gpu).half()
    xs = [xh, xh, xh, xo, xh, xh, xi]
    ys = [torch.full((rows, N), 7.0, device=gpu, dtype=torch.float16) for _, N in shapes]  # must be overwritten
    seq = decode.OpSequence(lays, xs, ys, batched=batched, fuse_shared_input=True, linear=True, fold_topx=fold)
    assert [len(g) for g in seq.groups] == [3, 1, 2, 1]
    assert all((seq.ops[i].topX == 0) == fold for i in range(7))  # top-X rows inside the CSR, or passed separately
    exact = [_exact(_npl(l), x.cpu().numpy(), "hybrid") for l, x in zip(lays, xs)]
    seq.launch()
    torch.cuda.synchronize()
    for y, e in zip(ys, exact):
        _check_fp16(y.cpu().numpy(), e)
    g = seq.graph()
    for y in ys:
        y.fill_(-3.0)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    for y, e in zip(ys, exact):
        _check_fp16(y.cpu().numpy(), e)
    with pytest.raises(NotImplementedError):
        seq.profile()


def test_linear_full_size_hybrid_w3(gpu):
    """LLaMA-7B down_proj shape (K = 11008), w3 s45 + top-X, through the C ABI directly."""
    import torch

    from squeezellm_amd import _lib, synth

    K, N = 11008, 4096
    lay = synth.make_layer(K, N, 3, sparse_frac=0.0045, topX=10, heavy_rows=10, bias=True, device=gpu, seed=9)
    x = torch.randn(K, device=gpu).half()
    out = torch.empty(N, device=gpu, dtype=torch.float16)
    ws = torch.zeros(_lib.linear_workspace_bytes(N, 0), dtype=torch.uint8, device=gpu)
    lin = _lib.SqllmLinear()
    o = lin.op
    o.bits, o.batch, o.K, o.N = 3, 0, K, N
    o.vec, o.qweight, o.mul, o.lookup_table = x.data_ptr(), lay["qweight"].data_ptr(), out.data_ptr(), lay["lookup_table"].data_ptr()
    o.rows, o.cols, o.vals, o.nnz = lay["rows"].data_ptr(), lay["cols"].data_ptr(), lay["vals"].data_ptr(), lay["vals"].numel()
    o.full_rows, o.full_row_indices, o.topX = lay["full_rows"].data_ptr(), lay["full_row_indices"].data_ptr(), 10
    lin.bias, lin.workspace = lay["bias"].data_ptr(), ws.data_ptr()
    lib = _lib.load()
    exact = _exact(_npl(lay), x.cpu().numpy().reshape(1, K), "hybrid")
    for _ in range(2):
        assert lib.sqllm_linear_f16(ctypes.byref(lin), torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        _check_fp16(out.cpu().numpy().reshape(1, N), exact)
    assert int(ws.count_nonzero()) == 0
    # rejected arguments enqueue nothing
    lin.workspace = None
    assert lib.sqllm_linear_f16(ctypes.byref(lin), None) == -3
    lin.workspace = ws.data_ptr() + 4
    assert lib.sqllm_linear_f16(ctypes.byref(lin), None) == -4


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("rows", [1, 3])
def test_linear_counts_every_kind_of_sparse_contribution(gpu, bits, rows):
    """Shapes chosen so that completion counting sees every case: rows spread over several CSR
    chunks (heavy rows), chunks that span more rows than fit in LDS (very sparse region, uncounted
    adds + a counting pass), empty rows, more than 64 top-X columns, and repeated top-X indices."""
    import torch

    from squeezellm_amd import quant, synth

    K, N = 512, 8192
    lay = synth.make_layer(K, N, bits, sparse_frac=1.0 / (4 * K), topX=70, heavy_rows=3, heavy_frac=0.9,
                           bias=True, device=gpu, seed=77 + bits)
    # rows 100..5000 get no outliers at all except three heavy ones -> one chunk spans > 2048 rows
    r = lay["rows"].cpu().numpy().astype(np.int64)
    cnt = np.diff(r)
    keep = np.ones(N, bool)
    keep[100:5000] = cnt[100:5000] > 100
    sel = np.repeat(keep, cnt)
    lay["cols"] = lay["cols"][torch.from_numpy(sel).to(gpu)].contiguous()
    lay["vals"] = lay["vals"][torch.from_numpy(sel).to(gpu)].contiguous()
    new_r = np.zeros(N + 1, np.int32)
    new_r[1:] = np.cumsum(np.where(keep, cnt, 0))
    lay["rows"] = torch.from_numpy(new_r).to(gpu)
    assert int(new_r[-1]) == lay["vals"].numel()
    fi = lay["full_row_indices"].clone()
    fi[1] = fi[0]  # the same column twice
    fi[69] = N - 1
    lay["full_row_indices"] = fi
    mod = quant.QuantLinearLUT.from_operands(lay)
    mod.__class__ = quant.QuantLinearLUTFused
    x = torch.randn((rows, K), device=gpu).half()
    exact = _exact(_npl(lay), x.cpu().numpy(), "hybrid")
    for _ in range(2):
        _check_fp16(mod(x).reshape(rows, N).cpu().numpy(), exact)
    assert int(next(iter(mod._ws.values())).count_nonzero()) == 0


def test_linears_on_one_stream_may_share_a_workspace(gpu):
    """include/sqllm_hip.h: "launches on one stream may share it".  Two different layers (same N)
    alternate on one workspace, back to back, no synchronisation in between."""
    import torch

    from squeezellm_amd import _lib, synth

    K, N = 2048, 1024
    lays = [synth.make_layer(K, N, b, sparse_frac=0.004, topX=4, heavy_rows=2, bias=True, device=gpu, seed=40 + b) for b in (3, 4)]
    xs = [torch.randn(K, device=gpu).half() for _ in lays]
    outs = [torch.empty(N, device=gpu, dtype=torch.float16) for _ in lays]
    ws = torch.zeros(_lib.linear_workspace_bytes(N, 0), dtype=torch.uint8, device=gpu)
    lins = []
    for lay, x, out in zip(lays, xs, outs):
        lin = _lib.SqllmLinear()
        o = lin.op
        o.bits, o.batch, o.K, o.N = lay["bits"], 0, K, N
        o.vec, o.qweight, o.mul, o.lookup_table = x.data_ptr(), lay["qweight"].data_ptr(), out.data_ptr(), lay["lookup_table"].data_ptr()
        o.rows, o.cols, o.vals, o.nnz = lay["rows"].data_ptr(), lay["cols"].data_ptr(), lay["vals"].data_ptr(), lay["vals"].numel()
        o.full_rows, o.full_row_indices, o.topX = lay["full_rows"].data_ptr(), lay["full_row_indices"].data_ptr(), 4
        lin.bias, lin.workspace = lay["bias"].data_ptr(), ws.data_ptr()
        lins.append(lin)
    lib = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        for lin in lins:
            assert lib.sqllm_linear_f16(ctypes.byref(lin), stream) == 0
    torch.cuda.synchronize()
    for lay, x, out in zip(lays, xs, outs):
        _check_fp16(out.cpu().numpy().reshape(1, N), _exact(_npl(lay), x.cpu().numpy().reshape(1, K), "hybrid"))
    assert int(ws.count_nonzero()) == 0


def test_fused_forward_keeps_one_workspace_across_batch_sizes(gpu):
    """Variable prompt lengths must not grow a cache: one zero-filled buffer per device, sized for
    the largest batch seen, whose prefix serves the smaller ones."""
    import torch

    from squeezellm_amd import quant, synth

    K, N = 512, 200
    lay = synth.make_layer(K, N, 4, sparse_frac=0.01, topX=3, heavy_rows=1, bias=True, device=gpu, seed=123)
    mod = quant.QuantLinearLUT.from_operands(lay)
    mod.__class__ = quant.QuantLinearLUTFused
    npl = _npl(lay)
    sizes = []
    for rows in (8, 3, 1, 12, 5, 12, 1):
        x = torch.randn((rows, K), device=gpu).half()
        y = mod(x if rows > 1 else x.reshape(1, 1, K)).reshape(rows, N)
        _check_fp16(y.cpu().numpy(), _exact(npl, x.cpu().numpy(), "hybrid"))
        assert sorted(k[1] == "graph" for k in mod._ws if k != "retired") == [False, True]  # one eager buffer (per device and stream) + the graph-only one
        assert len(mod._desc) == 1  # (ONE descriptor per device and stream, not one per row count pinning a superseded workspace each)
        sizes.append(next(iter(mod._ws.values())).numel())
        assert int(next(iter(mod._ws.values())).count_nonzero()) == 0
    assert sizes == sorted(sizes) and sizes[-1] == 8 * 12 * N


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("kind", ["dense", "hybrid"])
@pytest.mark.parametrize("rows", [1, 3])
@pytest.mark.parametrize("poison", ["nan", "+inf", "-inf", "+inf,-inf", "nan,+inf"])
def test_fused_linear_propagates_nan_and_inf_like_the_operator_path(gpu, bits, kind, rows, poison):
    """Non-finite activations: the reference's path (fp32 atomics into `mul`, squeezellm/quant.py:214-223) carries NaN
    and +-inf through to the output; the fused linear accumulates in integer fixed point and carries them in two
    sticky flag bits of the word.  Same non-finite pattern as the four-launch path (column by column: NaN where
    it is NaN, +-inf with the same sign), finite columns / rows unchanged, workspace left clean."""
    import torch

    from squeezellm_amd import quant, synth

    K, N = 1024, 264
    lay = synth.make_layer(K, N, bits, sparse_frac=0.0 if kind == "dense" else 0.01, topX=3 if kind == "hybrid" else 0,
                           heavy_rows=2 if kind != "dense" else 0, bias=True, device=gpu, seed=77 + bits)
    plain = quant.QuantLinearLUT.from_operands(lay)
    fused = quant.QuantLinearLUT.from_operands(lay)
    quant.fuse_quant_lut(fused)
    g = torch.Generator(device=gpu).manual_seed(5)
    x = torch.randn((rows, K), device=gpu, generator=g).half()
    vals = {"nan": float("nan"), "+inf": float("inf"), "-inf": float("-inf")}
    for j, name in enumerate(poison.split(",")):
        x[rows - 1, 37 + 500 * j] = vals[name]  # (two poisons land in different K slices of the tile)
    xin = x if rows > 1 else x.reshape(1, 1, K)
    want = plain(xin).reshape(rows, N).float().cpu()
    for rep in range(2):  # second call: the flags of the first must not linger in the workspace
        got = fused(xin).reshape(rows, N).float().cpu()
        assert torch.equal(torch.isnan(got), torch.isnan(want)), f"NaN pattern differs ({int(torch.isnan(got).sum())} vs {int(torch.isnan(want).sum())})"
        assert torch.equal(torch.isposinf(got), torch.isposinf(want)) and torch.equal(torch.isneginf(got), torch.isneginf(want))
        fin = torch.isfinite(want)
        assert fin[: rows - 1].all(), "rows without poison stay finite"
        assert torch.allclose(got[fin], want[fin], rtol=2e-3, atol=2e-3)
    assert (~torch.isfinite(want[rows - 1])).all(), "a poisoned row is non-finite in every column"
    ws = next(iter(fused._ws.values()))
    assert int(ws.count_nonzero()) == 0, "workspace must be left zero-filled"


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("rows", [1, 3])
def test_fused_module_with_and_without_folded_topx(gpu, bits, rows):
    """QuantLinearLUTFused hands the kernel one CSR that contains the top-X rows (default) or both terms separately:
    the same fp16 result within an ulp of the oracle either way, including a layer whose outliers are ALL in the
    top-X rows (empty CSR) and repeated top-X indices."""
    import torch

    from squeezellm_amd import quant, synth

    K, N = 1024, 456
    for sparse, dup in ((0.01, False), (0.0, False), (0.01, True)):
        lay = synth.make_layer(K, N, bits, sparse_frac=sparse, topX=4, heavy_rows=2 if sparse else 0, bias=True, device=gpu, seed=bits + rows)
        if dup:
            lay["full_row_indices"][1] = lay["full_row_indices"][0]
        x = torch.randn((rows, K), device=gpu).half()
        exact = _exact(_npl(lay), x.cpu().numpy(), "hybrid")
        outs = []
        for fold in (True, False):
            mod = quant.QuantLinearLUT.from_operands(lay)
            mod.__class__ = quant.QuantLinearLUTFused
            mod.fold_topx = fold
            y = mod(x if rows > 1 else x.reshape(1, 1, K)).reshape(rows, N)
            assert ("_folded" in mod.__dict__) == fold
            _check_fp16(y.cpu().numpy(), exact)
            outs.append(y)


def test_fused_linear_sees_operands_refreshed_in_place(gpu):
    """a module that ran one forward and then had its outlier buffers refreshed IN PLACE (what
    load_state_dict does: same storage, new contents) must not keep computing with the folded CSR it built from the
    old contents."""
    import torch

    from squeezellm_amd import quant, synth

    lay = synth.make_layer(512, 256, 4, sparse_frac=0.01, topX=6, heavy_rows=2, device=gpu, seed=77)
    mod = quant.QuantLinearLUT.from_operands(lay)
    mod.__class__ = quant.QuantLinearLUTFused
    x = torch.randn(1, 512, device=gpu, dtype=torch.float16)
    y0 = mod(x).float()
    with torch.no_grad():
        mod.vals.mul_(-3.0)       # in place: same pointers
        mod.full_rows.mul_(0.0)
    y1 = mod(x).float()
    ref = quant.QuantLinearLUT.from_operands(lay)  # the reference-style forward reads the live buffers
    want = ref(x).float()
    torch.cuda.synchronize()
    assert not torch.allclose(y0, y1, atol=1e-3)
    assert torch.allclose(y1, want, atol=2e-3 * float(want.abs().max())), float((y1 - want).abs().max())


def test_fused_descriptor_follows_routing_attributes_and_replaced_storage(gpu):
    """the cached descriptor is keyed on the buffers' identity, STORAGE and version and on the routing attributes --
    `mod.fold_topx = False` after a first forward takes effect, and `buf.data = other` (same object, same version counter,
    new storage) does not leave the kernel reading the old allocation."""
    import torch

    from squeezellm_amd import quant, synth

    lay = synth.make_layer(512, 256, 4, sparse_frac=0.01, topX=6, heavy_rows=2, device=gpu, seed=78)
    mod = quant.QuantLinearLUT.from_operands(lay)
    mod.__class__ = quant.QuantLinearLUTFused
    x = torch.randn(3, 512, device=gpu, dtype=torch.float16)
    y0 = mod(x).float()
    lin0 = next(iter(mod._desc.values()))[1][0]
    assert lin0.op.topX == 0  # folded: one CSR term
    mod.fold_topx = False
    y1 = mod(x).float()
    lin1 = next(iter(mod._desc.values()))[1][0]
    assert lin1.op.topX == 6 and lin1.op.full_rows == mod.full_rows.data_ptr()  # rebuilt: the top-X rows passed separately
    assert torch.allclose(y0, y1, atol=2e-3 * float(y0.abs().max()))
    new_lut = (mod.lookup_table * 2.0).clone()
    mod.lookup_table.data = new_lut  # same tensor object and version, another allocation
    y2 = mod(x).float()
    lin2 = next(iter(mod._desc.values()))[1][0]
    assert lin2.op.lookup_table == new_lut.data_ptr()
    ref = quant.QuantLinearLUT.from_operands(dict(lay, lookup_table=new_lut))
    want = ref(x).float()
    torch.cuda.synchronize()
    assert torch.allclose(y2, want, atol=2e-3 * float(want.abs().max())), float((y2 - want).abs().max())


def test_captured_module_forwards_hold_their_kernels_only(gpu):
    """seven fused linears captured through the torch MODULE must give the graph a C-ABI sequence gives --
    one kernel node per linear.  (Until round 6 the module's workspace was keyed on the stream, the capture stream is never
    the warm-up stream, and every linear of the graph got a `torch.zeros` fill kernel in front of it, replayed every time:
    484 against 603 tokens/s on the 7B pass.)  Two replays against the eager result; a capture WITHOUT a warm-up call falls
    back to an in-graph zero-filled temporary (two nodes per linear) and is still right."""
    import ctypes

    import torch

    from squeezellm_amd import quant, synth

    hip = ctypes.CDLL("libamdhip64.so")

    def node_types(g):
        raw = ctypes.c_void_p(g.raw_cuda_graph())
        n = ctypes.c_size_t(0)
        assert hip.hipGraphGetNodes(raw, None, ctypes.byref(n)) == 0
        nodes = (ctypes.c_void_p * n.value)()
        assert hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n)) == 0
        out = []
        for nd in nodes:
            ty = ctypes.c_int(-1)
            assert hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(ty)) == 0
            out.append(ty.value)
        return out

    K, N = 512, 328
    mods = []
    for i in range(7):
        lay = synth.make_layer(K, N, 4 if i % 2 else 3, sparse_frac=0.01, topX=3, heavy_rows=1, bias=bool(i % 3), device=gpu, seed=50 + i)
        m = quant.QuantLinearLUT.from_operands(lay)
        m.__class__ = quant.QuantLinearLUTFused
        mods.append(m)
    x = torch.randn((1, 1, K), device=gpu).half()
    for warm in (True, False):
        if not warm:
            for m in mods:
                m.__dict__.pop("_ws", None)  # as if never called eagerly
        outs = []

        def run():
            outs.clear()
            with torch.no_grad():
                for m in mods:
                    outs.append(m(x))

        if warm:
            side = torch.cuda.Stream(gpu)
            side.wait_stream(torch.cuda.current_stream(gpu))
            with torch.cuda.stream(side):
                run()
            torch.cuda.current_stream(gpu).wait_stream(side)
            torch.cuda.synchronize()
            want = [o.clone() for o in outs]
        g = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(g):
            run()
        types = node_types(g)
        if warm:
            assert types == [0] * len(mods), types  # hipGraphNodeTypeKernel x 7: nothing but the linears
        else:
            assert len(types) == 2 * len(mods) and 10 not in types and 11 not in types, types  # + one zero fill each (kernel or memset node)
        g.instantiate()
        for _ in range(2):
            for o in outs:
                o.zero_()
            g.replay()
            torch.cuda.synchronize()
            for o, w in zip(outs, want):
                assert torch.equal(o, w)  # (the fused linear is bit-reproducible: integer accumulation)
