"""GPU tests of the host-side pieces around the operator: QuantLinearLUT.forward (both branches,
every op kind), the whole-pass launchers (one FFI crossing / HIP-graph replay / per-kernel
timing), and the committed golden vectors (reference-kernel outputs recorded on an MI355X)."""
import glob
import os

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("cfg", ["dense", "spmv", "hybrid", "balanced"])
@pytest.mark.parametrize("dtype", ["float16", "float32"])
def test_quantlinear_forward_vs_oracle(gpu, bits, cfg, dtype):
    import torch

    from squeezellm_amd import quant, synth

    K, N = 512, 384
    lay = synth.make_layer(K, N, bits, sparse_frac=0.0 if cfg == "dense" else 0.01, topX=4 if cfg == "hybrid" else 0,
                           heavy_rows=2 if cfg != "dense" else 0, bias=True, device=gpu, seed=bits * 7)
    mod = quant.QuantLinearLUT.from_operands(lay, balanced=(cfg == "balanced"))
    assert mod.op_kind(False) == {"dense": "dense", "spmv": "spmv", "hybrid": "spmv_hybrid", "balanced": "spmv_balanced"}[cfg]
    npl = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in lay.items()}
    tdt = getattr(torch, dtype)
    tol = 2e-3 if dtype == "float16" else 2e-5
    g = torch.Generator(device=gpu).manual_seed(3)
    for shape in [(K,), (1, 1, K), (5, K), (2, 3, K)]:  # matvec branch x2, batched branch x2
        x = torch.randn(shape, device=gpu, generator=g).to(tdt)
        y = mod(x)
        ref = H.oracle.quantlinear_forward(x.cpu().numpy(), npl)
        # (matvec branch: the input dtype; batched branch: `out.to(dtype) + bias`, promoted by the fp32 bias)
        assert y.shape == ref.shape and y.dtype == torch.from_numpy(ref[:0]).dtype
        assert y.dtype == (tdt if len(shape) == 1 or shape[:-1] == (1, 1) else torch.float32)
        assert H.rel_err(y.float().cpu().numpy(), ref.astype(np.float32)) <= tol


def test_make_quant_lut_swaps_linears(gpu):
    import torch
    import torch.nn as nn

    from squeezellm_amd import quant

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj = nn.Linear(64, 96, bias=False)
            self.mlp = nn.Sequential(nn.Linear(96, 128), nn.ReLU())
            self.lm_head = nn.Linear(128, 10)

    m = Block()
    quant.make_quant_lut(m, {"q_proj", "mlp.0"}, 4)
    assert isinstance(m.q_proj, quant.QuantLinearLUT) and isinstance(m.mlp[0], quant.QuantLinearLUT)
    assert isinstance(m.lm_head, nn.Linear) and m.mlp[0].bias is not None and m.q_proj.bias is None
    assert m.q_proj.qweight.shape == (64 // 32 * 4, 96)


@pytest.mark.parametrize("batched", [False, True])
def test_op_sequence_launch_graph_and_profile(gpu, batched):
    import torch

    from squeezellm_amd import decode, synth

    layers = [synth.make_layer(K, N, bits, sparse_frac=sp, topX=tx, heavy_rows=1 if sp else 0, device=gpu, seed=i)
              for i, (bits, K, N, sp, tx) in enumerate([(4, 256, 512, 0, 0), (3, 512, 256, 0.01, 3), (4, 512, 512, 0.01, 0), (3, 256, 256, 0, 0)])]
    B = 3 if batched else 0
    g = torch.Generator(device=gpu).manual_seed(0)
    xs = [torch.randn((B, l["K"]) if batched else (l["K"],), device=gpu, generator=g) for l in layers]
    ys = [torch.zeros((B, l["N"]) if batched else (l["N"],), device=gpu) for l in layers]
    seq = decode.OpSequence(layers, xs, ys, batched=batched)
    seq.launch()
    torch.cuda.synchronize()
    refs = []
    for l, x in zip(layers, xs):
        npl = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in l.items()}
        kind = "hybrid" if l["full_rows"] is not None else ("spmv" if l["vals"] is not None else "dense")
        refs.append(H.oracle_ref(npl, x.cpu().numpy(), np.zeros(tuple(ys[0].shape[:-1]) + (l["N"],), np.float32), kind))
    for y, r in zip(ys, refs):
        assert H.rel_err(y.cpu().numpy(), r) <= 2e-5
    # graph replay accumulates once more per replay (mul += ...)
    graph = seq.graph(warmup=1)  # warm-up launch: ys = 2 * ref; capture itself does not execute
    graph.replay()
    graph.replay()
    torch.cuda.synchronize()
    for y, r in zip(ys, refs):
        assert H.rel_err(y.cpu().numpy(), 4 * r) <= 2e-5
    us = seq.profile(reps=2)  # +2 passes
    assert us.shape == (4,) and (us > 0).all() and (us < 1e4).all()
    torch.cuda.synchronize()
    for y, r in zip(ys, refs):
        assert H.rel_err(y.cpu().numpy(), 6 * r) <= 2e-5
    with pytest.raises(ValueError):
        decode.OpSequence(layers, [x.double() for x in xs], ys, batched=batched)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(H.GOLDEN, "refkernel_w*.npz"))), ids=os.path.basename)
def test_against_committed_reference_kernel_vectors(gpu, path):
    """Our kernels vs what the reference's kernels produced for the same operands (recorded by
    tests/golden/make_refkernel_golden.py; the fixtures travel, /root/reference does not)."""
    import torch

    from squeezellm_amd import quant_cuda as qc

    g = np.load(path)
    case = {k: g[k] for k in ("qweight", "lookup_table", "rows", "cols", "vals", "full_rows", "full_row_indices")}
    case.update(bits=int(g["bits"]), K=int(g["K"]), N=int(g["N"]))
    t = H.to_torch(case, gpu)
    for kind in ("dense", "spmv", "hybrid"):
        for tag, x, mul in (("1", g["x1"], g["mul1"]), ("b", g["xb"], g["mulb"])):
            y = torch.from_numpy(mul.copy()).to(gpu)
            H.call_op(qc, t, torch.from_numpy(x).to(gpu), y, kind, tag == "b")
            assert H.rel_err(y.cpu().numpy(), g[f"y_{kind}_{tag}"]) <= 2e-5


def test_launch_is_on_the_current_stream_and_capturable(gpu):
    """The reference launches on the legacy default stream; ours must follow torch's current stream."""
    import torch

    from squeezellm_amd import quant_cuda as qc

    case = H.make_case(4, 1024, 512, seed=4)
    t = H.to_torch(case, gpu)
    x = torch.randn(1024, device=gpu)
    y = torch.zeros(512, device=gpu)
    s = torch.cuda.Stream(gpu)
    s.wait_stream(torch.cuda.current_stream(gpu))
    with torch.cuda.stream(s):
        H.call_op(qc, t, x, y, "dense", False)  # warm-up on the side stream
    torch.cuda.current_stream(gpu).wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    y.zero_()
    with torch.cuda.graph(graph):
        H.call_op(qc, t, x, y, "dense", False)
    assert float(y.abs().sum()) == 0.0  # capture enqueues nothing
    graph.replay()
    torch.cuda.synchronize()
    ref = H.oracle_ref(case, x.cpu().numpy(), np.zeros(512, np.float32), "dense")
    assert H.rel_err(y.cpu().numpy(), ref) <= 2e-5


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("batched", [False, True])
def test_grouped_launch_same_input(gpu, bits, batched):
    """Ops that share their input vector (q/k/v, gate/up) as ONE kernel: same results as separate
    launches; mixed dense / sparse members; mismatching members are rejected."""
    import ctypes

    import torch

    from squeezellm_amd import _lib, decode, synth

    K = 1024
    specs = [(384, 0.0, 0), (132, 0.01, 3), (1024, 0.0, 0), (64, 0.02, 0)]  # N, sparse, topX
    layers = [synth.make_layer(K, N, bits, sparse_frac=sp, topX=tx, heavy_rows=1 if sp else 0, device=gpu, seed=10 + i)
              for i, (N, sp, tx) in enumerate(specs)]
    B = 5 if batched else 0
    x = torch.randn((B, K) if batched else (K,), device=gpu)
    other = synth.make_layer(512, 256, bits, device=gpu, seed=99)
    x_other = torch.randn((B, 512) if batched else (512,), device=gpu)
    all_layers = layers + [other]
    xs = [x] * len(layers) + [x_other]

    def run(fuse):
        ys = [torch.zeros((B, l["N"]) if batched else (l["N"],), device=gpu) for l in all_layers]
        seq = decode.OpSequence(all_layers, xs, ys, batched=batched, fuse_shared_input=fuse)
        seq.launch()
        torch.cuda.synchronize()
        return seq, ys

    seq1, y_sep = run(False)
    seqf, y_fus = run(True)
    assert seq1.n_groups == 5 and seqf.groups == [[0, 1, 2, 3], [4]]
    for l, a, b in zip(all_layers, y_sep, y_fus):
        npl = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in l.items()}
        kind = "hybrid" if l["full_rows"] is not None else ("spmv" if l["vals"] is not None else "dense")
        xin = (x_other if l is other else x).cpu().numpy()
        ref = H.oracle_ref(npl, xin, np.zeros(tuple(a.shape), np.float32), kind)
        assert H.rel_err(a.cpu().numpy(), ref) <= 2e-5
        assert H.rel_err(b.cpu().numpy(), ref) <= 2e-5
    us = seqf.profile(reps=1)
    assert us.shape == (2,)
    # a group whose members disagree is rejected before anything is enqueued
    lib = _lib.load()
    bad = (_lib.SqllmOp * 2)(seq1.ops[0], seq1.ops[4])
    assert lib.sqllm_launch_group(bad, 2, None) == -8
    assert lib.sqllm_launch_group(seq1.ops, 0, None) == -8 and lib.sqllm_launch_group(seq1.ops, 5, None) == -8


@pytest.mark.parametrize("bits", [3, 4])
def test_gpu_packed_layer_runs_through_the_module(gpu, bits):
    """pack.pack_layer on device tensors -> QuantLinearLUT -> hybrid op; reference result is the
    fp64 product with the effective weight lut[idx] + (outlier - zero centroid)."""
    import torch

    from squeezellm_amd import pack
    from squeezellm_amd.quant import QuantLinearLUT

    N, K, topX = 512, 1024, 10
    rng = np.random.default_rng(100 + bits)
    idx_nk = rng.integers(0, 1 << bits, size=(N, K))
    lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float32), axis=1)
    outl = np.where(rng.random((N, K)) < 0.005, rng.normal(0, 0.2, (N, K)), 0).astype(np.float32)
    outl[[5, 99]] = rng.normal(0, 0.2, (2, K)).astype(np.float32)
    dev = gpu
    layer = pack.pack_layer(torch.from_numpy(idx_nk).to(dev), torch.from_numpy(lut).to(dev), bits,
                            torch.from_numpy(outl).to(dev), topX=topX)
    assert layer["qweight"].is_cuda and layer["full_rows"].shape == (K, topX)
    assert np.array_equal(layer["qweight"].cpu().numpy(), H.oracle.pack_indices(idx_nk.T, bits))
    m = QuantLinearLUT.from_operands(layer)
    x = torch.from_numpy(rng.normal(size=(1, 1, K)).astype(np.float32)).to(dev)
    y = m(x).reshape(-1).double().cpu().numpy()
    zero = lut[np.arange(N), np.abs(lut).argmin(1)]
    W = lut[np.arange(N)[:, None], idx_nk].astype(np.float64) + np.where(outl != 0, outl.astype(np.float64) - zero[:, None], 0)
    ref = W @ x.reshape(-1).double().cpu().numpy()
    assert H.rel_err(y, ref) < 1e-5


def test_c_abi_demo_builds_and_runs_without_python_in_the_loop(gpu, tmp_path):
    """examples/c_abi_demo.c: plain C against include/sqllm_hip.h + libsqllm_hip.so (operator and
    fused linear), compiled here with gcc and run as its own process."""
    import shutil
    import subprocess

    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if not shutil.which("gcc") or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("gcc / ROCm headers not available on this box")
    libdir = os.path.join(H.ROOT, "squeezellm_amd")
    exe = str(tmp_path / "c_abi_demo")
    subprocess.run(["gcc", "-std=c99", "-O2", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include", f"-I{os.path.join(H.ROOT, 'include')}",
                    os.path.join(H.ROOT, "examples", "c_abi_demo.c"), f"-L{libdir}", "-lsqllm_hip", f"-L{rocm}/lib", "-lamdhip64",
                    f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{rocm}/lib", "-lm", "-o", exe], check=True, capture_output=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "ok (ABI version 1)" in res.stdout and "more than one fp16 ulp: 0 of" in res.stdout


@pytest.mark.parametrize("bits", [3, 4])
def test_all_outliers_in_full_rows_keeps_the_topx_term(gpu, bits):
    """A layer whose outliers all moved into full_rows has an EMPTY CSR (nnz == 0) but a top-X term:
    it must still dispatch the hybrid op, through the operator path and through the fused linear
    (round-1 advisor finding: from_operands gated the top-X term on nnz > 0)."""
    import torch

    from squeezellm_amd import quant

    K, N = 256, 192
    case = H.make_case(bits, K, N, topX=3, seed=17)
    case.update(rows=np.zeros(N + 1, np.int32), cols=np.zeros(0, np.int32), vals=np.zeros(0, np.float32))
    lay = H.to_torch(case, gpu)
    lay["bias"] = None
    mod = quant.QuantLinearLUT.from_operands(lay)
    assert mod.include_sparse and mod.op_kind(False) == "spmv_hybrid" and mod.op_kind(True) == "spmv_hybrid"
    layer = dict(case, bias=None)
    rng = np.random.default_rng(4)
    for shape in ((K,), (5, K), (12, K)):
        x = rng.normal(size=shape).astype(np.float16)
        ref = H.oracle.quantlinear_forward(x, layer).astype(np.float32)
        assert np.abs(ref - H.oracle.quantlinear_forward(x, dict(layer, full_rows=None, full_row_indices=None, rows=None)).astype(np.float32)).max() > 0
        y = mod(torch.from_numpy(x).to(gpu))
        assert H.rel_err(y.float().cpu().numpy(), ref) <= 2e-3
    fused = quant.QuantLinearLUT.from_operands(lay)
    fused.__class__ = quant.QuantLinearLUTFused
    x = rng.normal(size=(3, K)).astype(np.float16)
    y = fused(torch.from_numpy(x).to(gpu))
    assert H.rel_err(y.float().cpu().numpy(), H.oracle.quantlinear_forward(x, layer).astype(np.float32)) <= 2e-3


def test_inconsistent_csr_fails_loudly(gpu):
    """The fused linear counts contributions from `rows`; an inconsistent CSR must raise instead of
    leaving columns unfinished and the workspace dirty: once per module in QuantLinearLUTFused, and
    on every launch behind the C ABI's "validate_csr" debug option."""
    import torch

    from squeezellm_amd import _lib, quant, quant_cuda as qc, synth

    lay = synth.make_layer(256, 128, 4, sparse_frac=0.02, heavy_rows=1, device=gpu, seed=3)
    good_rows = lay["rows"].clone()
    x = torch.randn(1, 256, device=gpu, dtype=torch.float16)
    mod = quant.QuantLinearLUT.from_operands(lay)
    mod.__class__ = quant.QuantLinearLUTFused
    mod(x)  # consistent: fine
    bad = dict(lay)
    bad["rows"] = good_rows.clone()
    bad["rows"][5] = bad["rows"][6] + 1  # not non-decreasing
    mod2 = quant.QuantLinearLUT.from_operands(bad)
    mod2.__class__ = quant.QuantLinearLUTFused
    with pytest.raises(ValueError, match="inconsistent CSR"):
        mod2(x)
    # the C ABI's debug option, through an operator name
    y = torch.zeros(128, device=gpu)
    qc.vecquant4matmul_spmv_nuq_perchannel(good_rows, lay["cols"], lay["vals"], x.float().reshape(-1), y, 128, lay["qweight"], lay["lookup_table"])
    _lib.set_option("validate_csr", 1)
    try:
        qc.vecquant4matmul_spmv_nuq_perchannel(good_rows, lay["cols"], lay["vals"], x.float().reshape(-1), y, 128, lay["qweight"], lay["lookup_table"])
        with pytest.raises(ValueError, match="sparse"):
            qc.vecquant4matmul_spmv_nuq_perchannel(bad["rows"], lay["cols"], lay["vals"], x.float().reshape(-1), y, 128, lay["qweight"], lay["lookup_table"])
        short = good_rows.clone()
        short[-1] -= 1  # rows[N] != nnz
        with pytest.raises(ValueError, match="sparse"):
            qc.vecquant4matmul_spmv_nuq_perchannel(short, lay["cols"], lay["vals"], x.float().reshape(-1), y, 128, lay["qweight"], lay["lookup_table"])
    finally:
        _lib.set_option("validate_csr", 0)
    torch.cuda.synchronize()
