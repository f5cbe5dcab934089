"""Shared test helpers: seeded numpy cases, the C oracle binding, torch <-> numpy plumbing."""
import ctypes
import os

import numpy as np

from oracle import sqllm_oracle as oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def c_oracle():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libsqllm_oracle.so"))
    lib.sqo_matvec.restype = ctypes.c_int
    lib.sqo_unpack.restype = ctypes.c_int
    return lib


def _p(a, t):
    return None if a is None else np.ascontiguousarray(a).ctypes.data_as(ctypes.POINTER(t))


def c_matvec(lib, case, x, mul, batched):
    K, N, bits = case["K"], case["N"], case["bits"]
    B = x.shape[0] if batched else 1
    out = np.zeros((B, N), np.float64)
    topX = 0 if case.get("full_rows") is None else case["full_rows"].shape[1]
    keep = [np.ascontiguousarray(a) if a is not None else None for a in
            (x, case["qweight"], mul, case["lookup_table"], case.get("rows"), case.get("cols"), case.get("vals"),
             case.get("full_rows"), case.get("full_row_indices"))]
    rc = lib.sqo_matvec(bits, B if batched else 0, _p(keep[0], ctypes.c_float), _p(keep[1], ctypes.c_int32),
                        _p(keep[2], ctypes.c_float), _p(keep[3], ctypes.c_float), K, N,
                        _p(keep[4], ctypes.c_int32), _p(keep[5], ctypes.c_int32), _p(keep[6], ctypes.c_float),
                        _p(keep[7], ctypes.c_float), _p(keep[8], ctypes.c_int32), topX, _p(out, ctypes.c_double))
    assert rc == 0
    return out if batched else out[0]


def make_case(bits, K, N, *, sparse=0.0, topX=0, heavy_rows=0, empty_rows=(), dup_topx=False, seed=0):
    """Seeded numpy operands (same distributions as squeezellm_amd.synth)."""
    rng = np.random.default_rng(seed)
    q = rng.integers(-(2**31), 2**31, size=(K // 32 * bits, N), dtype=np.int64).astype(np.int32)
    lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float32), axis=1)
    case = dict(bits=bits, K=K, N=N, qweight=q, lookup_table=lut, rows=None, cols=None, vals=None,
                full_rows=None, full_row_indices=None)
    if sparse > 0 or heavy_rows or len(empty_rows):
        mask = rng.random((N, K)) < sparse
        for r in rng.choice(N, size=heavy_rows, replace=False) if heavy_rows else []:
            mask[r] = rng.random(K) < 0.3
        for r in empty_rows:
            mask[r] = False
        counts = mask.sum(axis=1)
        rows = np.zeros(N + 1, np.int32)
        rows[1:] = np.cumsum(counts)
        cols = np.nonzero(mask)[1].astype(np.int32)
        vals = rng.normal(0, 0.1, cols.size).astype(np.float32)
        case.update(rows=rows, cols=cols, vals=vals)
    if topX:
        case["full_rows"] = rng.normal(0, 0.02, (K, topX)).astype(np.float32)
        idx = rng.choice(N, size=topX, replace=False).astype(np.int32)
        if dup_topx and topX > 1:
            idx[1] = idx[0]
        case["full_row_indices"] = idx
    return case


def to_torch(case, device):
    import torch

    out = {}
    for k, v in case.items():
        out[k] = torch.from_numpy(np.ascontiguousarray(v)).to(device) if isinstance(v, np.ndarray) else v
    return out


def rel_err(got, ref):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))


ENTRIES = ("module", "named", "ws-null")


def _call_c_abi(t, x, y, kind, batched, entry):
    """The same op straight through the C ABI (include/sqllm_hip.h), on torch's current stream:
    "named"   -- the header's `sqllm_vecquant{3,4}matmul*` symbol for (bits, kind, batched): what a C / C++ caller binds in
                 place of quant_cuda_kernel.cu:132-738; workspace-less, so wider batches take the library's stream-ordered
                 scratch (or, inside a capture, memory nodes / the gathering fallback);
    "ws-null" -- sqllm_launch_ws with a NULL workspace (the no-workspace contract of the `_ws` entry points)."""
    import ctypes

    import torch

    from squeezellm_amd import _lib

    lib = _lib.load()
    b, K, N = t["bits"], t["K"], t["N"]
    stream = torch.cuda.current_stream(x.device).cuda_stream
    q, lut = t["qweight"], t["lookup_table"]
    height = q.shape[0]
    batch = x.shape[0] if batched else 0
    assert x.dtype == torch.float32 and y.dtype == torch.float32 and x.is_contiguous() and y.is_contiguous()
    sparse = kind in ("spmv", "hybrid")
    hybrid = kind == "hybrid"
    nnz = t["vals"].numel() if sparse else 0
    topX = t["full_rows"].shape[1] if hybrid else 0
    with torch.cuda.device(x.device):
        if entry == "ws-null":
            op = _lib.SqllmOp(bits=b, batch=batch, K=K, N=N, vec=x.data_ptr(), qweight=q.data_ptr(), mul=y.data_ptr(), lookup_table=lut.data_ptr())
            if sparse:
                op.rows, op.cols, op.vals, op.nnz = t["rows"].data_ptr(), t["cols"].data_ptr(), t["vals"].data_ptr(), nnz
            if hybrid:
                op.full_rows, op.full_row_indices, op.topX = t["full_rows"].data_ptr(), t["full_row_indices"].data_ptr(), topX
            rc = lib.sqllm_launch_ws(ctypes.byref(op), None, 0, stream)
            what = "sqllm_launch_ws(NULL)"
        else:
            sfx = "_batched" if batched else ""
            tail = (batch, K, stream) if batched else (stream,)
            if kind == "dense":
                what = f"sqllm_vecquant{b}matmul_nuq_perchannel{sfx}"
                args = (x.data_ptr(), q.data_ptr(), y.data_ptr(), lut.data_ptr(), height, N)
            elif kind == "spmv":
                what = f"sqllm_vecquant{b}matmul_spmv_nuq_perchannel{sfx}"
                args = (t["rows"].data_ptr(), t["cols"].data_ptr(), t["vals"].data_ptr(), x.data_ptr(), y.data_ptr(), N, q.data_ptr(), lut.data_ptr(),
                        height, N, nnz)
            elif kind == "hybrid":
                what = f"sqllm_vecquant{b}matmul_spmv_hybrid_nuq_perchannel{sfx}"
                args = (t["rows"].data_ptr(), t["cols"].data_ptr(), t["vals"].data_ptr(), x.data_ptr(), t["full_rows"].data_ptr(),
                        t["full_row_indices"].data_ptr(), y.data_ptr(), N, q.data_ptr(), lut.data_ptr(), height, N, nnz, topX)
            else:
                raise ValueError(kind)
            rc = getattr(lib, what)(*args, *tail)
    _lib.check(rc, what)


def call_op(qc, t, x, y, kind, batched, entry="module"):
    """Dispatch to the quant_cuda name for (bits, kind, batched) with the reference argument order (entry "module"), or to
    the C ABI directly ("named", "ws-null": see _call_c_abi)."""
    if entry != "module":
        return _call_c_abi(t, x, y, kind, batched, entry)
    b = t["bits"]
    sfx = "_batched" if batched else ""
    if kind == "dense":
        getattr(qc, f"vecquant{b}matmul_nuq_perchannel{sfx}")(x, t["qweight"], y, t["lookup_table"])
    elif kind == "spmv":
        getattr(qc, f"vecquant{b}matmul_spmv_nuq_perchannel{sfx}")(
            t["rows"], t["cols"], t["vals"], x, y, t["N"], t["qweight"], t["lookup_table"])
    elif kind == "hybrid":
        getattr(qc, f"vecquant{b}matmul_spmv_hybrid_nuq_perchannel{sfx}")(
            t["rows"], t["cols"], t["vals"], x, t["full_rows"], t["full_row_indices"], y, t["N"],
            t["qweight"], t["lookup_table"])
    else:
        raise ValueError(kind)


def oracle_ref(case, x, mul, kind):
    kw = {}
    if kind in ("spmv", "hybrid"):
        kw.update(rows=case["rows"], cols=case["cols"], vals=case["vals"])
    if kind == "hybrid":
        kw.update(full_rows=case["full_rows"], full_row_indices=case["full_row_indices"])
    return oracle.matvec(x, case["qweight"], mul, case["lookup_table"], case["bits"], **kw)
