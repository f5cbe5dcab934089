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


def call_op(qc, t, x, y, kind, batched):
    """Dispatch to the quant_cuda name for (bits, kind, batched) with the reference argument order."""
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
