"""`quant_cuda` for MI355X: the operator module `squeezellm/quant.py` imports.

Host-side mirror of the reference's pybind11 extension (/root/reference/squeezellm/quant_cuda.cpp:
the 12 functions exported at :257-270, forwarding wrappers :112-255) plus the two `balanced`
names that `QuantLinearLUT.forward` calls (squeezellm/quant.py:237-250, :281-294) but the
reference never defined.  Same names, same positional arguments, same in-place
accumulate-into-`mul` semantics, `None` return.

Differences from the reference, all on the safe side:
  * arguments are validated (dtype / device / contiguity / shapes) and errors raise -- the
    reference validated nothing and read out of bounds;
  * the launch goes to torch's CURRENT stream (the reference used the legacy default stream), so
    calls are ordered with surrounding torch ops and can be captured in a HIP graph;
  * one fused kernel per call instead of 1-3 dependent launches.

Everything is delegated to libsqllm_hip.so through its C ABI (include/sqllm_hip.h).  There is no
CPU / eager fallback: CPU tensors or a missing library raise.
"""
from __future__ import annotations

import torch

from . import _lib

__all__ = [
    f"vecquant{b}matmul{kind}_nuq_perchannel{suffix}"
    for b in (3, 4)
    for kind, suffix in (("", ""), ("", "_batched"), ("_spmv", ""), ("_spmv", "_batched"),
                         ("_spmv_hybrid", ""), ("_spmv_hybrid", "_batched"), ("_spmv_balanced", ""))
]

_F32 = torch.float32
_I32 = torch.int32


def _dev_ptr(t, dtype, name: str) -> int:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t).__name__}")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor (quant_cuda has no CPU path), got device {t.device}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t.data_ptr()


def _same_device(ref: torch.Tensor, *others) -> None:
    for t in others:
        if t.device != ref.device:
            raise RuntimeError(f"all operands must be on {ref.device}, found {t.device}")


class _on_device_of:
    """`const at::cuda::OptionalCUDAGuard device_guard(device_of(vec))` (quant_cuda.cpp:116):
    make vec's device current for the launch; a no-op in the common single-device case."""

    __slots__ = ("idx", "prev")

    def __init__(self, t: torch.Tensor):
        self.idx = t.device.index

    def __enter__(self):
        self.prev = torch.cuda.current_device()
        if self.prev != self.idx:
            torch.cuda.set_device(self.idx)
        return torch.cuda.current_stream(self.idx).cuda_stream

    def __exit__(self, *exc):
        if self.prev != self.idx:
            torch.cuda.set_device(self.prev)
        return False


def _dense_shapes(vec, mat, mul, lookup_table, bits: int, batched: bool):
    if mat.dim() != 2:
        raise ValueError("qweight must be 2-D [K/32*bits, N]")
    height, width = mat.shape
    if height % bits:
        raise ValueError(f"qweight has {height} rows, not a multiple of {bits}")
    K = height // bits * 32
    if tuple(lookup_table.shape) != (width, 1 << bits):
        raise ValueError(f"lookup_table must be [{width}, {1 << bits}], got {tuple(lookup_table.shape)}")
    if batched:
        if vec.dim() != 2 or vec.shape[1] != K:
            raise ValueError(f"vec must be [batch, {K}], got {tuple(vec.shape)}")
        if mul.dim() != 2 or tuple(mul.shape) != (vec.shape[0], width):
            raise ValueError(f"mul must be [{vec.shape[0]}, {width}], got {tuple(mul.shape)}")
        return height, width, K, vec.shape[0]
    if vec.numel() != K:
        raise ValueError(f"vec must have {K} elements, got {vec.numel()}")
    if mul.numel() != width:
        raise ValueError(f"mul must have {width} elements, got {mul.numel()}")
    return height, width, K, 0


def _csr_shapes(rows, cols, vals, num_rows: int, width: int) -> int:
    if int(num_rows) != width:
        raise ValueError(f"num_rows ({num_rows}) must equal outfeatures ({width})")
    if rows.numel() != width + 1:
        raise ValueError(f"rows must have {width + 1} entries, got {rows.numel()}")
    if cols.numel() != vals.numel():
        raise ValueError("cols and vals must have the same length")
    return cols.numel()


def _dense(bits, batched, vec, mat, mul, lookup_table):
    height, width, K, batch = _dense_shapes(vec, mat, mul, lookup_table, bits, batched)
    pv, pq = _dev_ptr(vec, _F32, "vec"), _dev_ptr(mat, _I32, "mat")
    pm, pl = _dev_ptr(mul, _F32, "mul"), _dev_ptr(lookup_table, _F32, "lookup_table")
    _same_device(vec, mat, mul, lookup_table)
    lib = _lib.load()
    with _on_device_of(vec) as stream:
        if batched:
            fn = getattr(lib, f"sqllm_vecquant{bits}matmul_nuq_perchannel_batched")
            rc = fn(pv, pq, pm, pl, height, width, batch, K, stream)
        else:
            fn = getattr(lib, f"sqllm_vecquant{bits}matmul_nuq_perchannel")
            rc = fn(pv, pq, pm, pl, height, width, stream)
    _lib.check(rc, fn.__name__)


def _spmv(bits, batched, rows, cols, mat, vec, mul, num_rows, matq, lookup_table):
    height, width, K, batch = _dense_shapes(vec, matq, mul, lookup_table, bits, batched)
    nnz = _csr_shapes(rows, cols, mat, num_rows, width)
    pr, pc, pvl = _dev_ptr(rows, _I32, "rows"), _dev_ptr(cols, _I32, "cols"), _dev_ptr(mat, _F32, "mat (csr values)")
    pv, pq = _dev_ptr(vec, _F32, "vec"), _dev_ptr(matq, _I32, f"mat{bits}")
    pm, pl = _dev_ptr(mul, _F32, "mul"), _dev_ptr(lookup_table, _F32, "lookup_table")
    _same_device(vec, rows, cols, mat, matq, mul, lookup_table)
    lib = _lib.load()
    with _on_device_of(vec) as stream:
        if batched:
            fn = getattr(lib, f"sqllm_vecquant{bits}matmul_spmv_nuq_perchannel_batched")
            rc = fn(pr, pc, pvl, pv, pm, int(num_rows), pq, pl, height, width, nnz, batch, K, stream)
        else:
            fn = getattr(lib, f"sqllm_vecquant{bits}matmul_spmv_nuq_perchannel")
            rc = fn(pr, pc, pvl, pv, pm, int(num_rows), pq, pl, height, width, nnz, stream)
    _lib.check(rc, fn.__name__)


def _hybrid(bits, batched, rows, cols, mat, vec, full_rows, full_row_indices, mul, num_rows, matq, lookup_table):
    height, width, K, batch = _dense_shapes(vec, matq, mul, lookup_table, bits, batched)
    nnz = _csr_shapes(rows, cols, mat, num_rows, width)
    if full_rows.dim() != 2 or full_rows.shape[0] != K:
        raise ValueError(f"full_rows must be [{K}, topX], got {tuple(full_rows.shape)}")
    topX = full_rows.shape[1]
    if full_row_indices.numel() != topX:
        raise ValueError(f"full_row_indices must have {topX} entries, got {full_row_indices.numel()}")
    pr, pc, pvl = _dev_ptr(rows, _I32, "rows"), _dev_ptr(cols, _I32, "cols"), _dev_ptr(mat, _F32, "mat (csr values)")
    pv, pq = _dev_ptr(vec, _F32, "vec"), _dev_ptr(matq, _I32, f"mat{bits}")
    pm, pl = _dev_ptr(mul, _F32, "mul"), _dev_ptr(lookup_table, _F32, "lookup_table")
    pfr, pfi = _dev_ptr(full_rows, _F32, "full_rows"), _dev_ptr(full_row_indices, _I32, "full_row_indices")
    _same_device(vec, rows, cols, mat, matq, mul, lookup_table, full_rows, full_row_indices)
    lib = _lib.load()
    with _on_device_of(vec) as stream:
        if batched:
            fn = getattr(lib, f"sqllm_vecquant{bits}matmul_spmv_hybrid_nuq_perchannel_batched")
            rc = fn(pr, pc, pvl, pv, pfr, pfi, pm, int(num_rows), pq, pl, height, width, nnz, topX, batch, K, stream)
        else:
            fn = getattr(lib, f"sqllm_vecquant{bits}matmul_spmv_hybrid_nuq_perchannel")
            rc = fn(pr, pc, pvl, pv, pfr, pfi, pm, int(num_rows), pq, pl, height, width, nnz, topX, stream)
    _lib.check(rc, fn.__name__)


def _balanced(bits, rows, cols, startrows, vals, vec, mul, matq, lookup_table, outfeatures, num_threads, numvals):
    height, width, K, _ = _dense_shapes(vec, matq, mul, lookup_table, bits, False)
    nnz = _csr_shapes(rows, cols, vals, outfeatures, width)
    if int(numvals) != nnz:
        raise ValueError(f"numvals ({numvals}) != number of stored values ({nnz})")
    pr, pc, pvl = _dev_ptr(rows, _I32, "rows"), _dev_ptr(cols, _I32, "cols"), _dev_ptr(vals, _F32, "vals")
    psr = _dev_ptr(startrows, _I32, "startrows") if startrows is not None else None
    pv, pq = _dev_ptr(vec, _F32, "vec"), _dev_ptr(matq, _I32, f"mat{bits}")
    pm, pl = _dev_ptr(mul, _F32, "mul"), _dev_ptr(lookup_table, _F32, "lookup_table")
    _same_device(vec, rows, cols, vals, matq, mul, lookup_table)
    lib = _lib.load()
    fn = getattr(lib, f"sqllm_vecquant{bits}matmul_spmv_balanced_nuq_perchannel")
    with _on_device_of(vec) as stream:
        rc = fn(pr, pc, psr, pvl, pv, pm, pq, pl, int(outfeatures), int(num_threads), nnz, height, width, stream)
    _lib.check(rc, fn.__name__)


# ---- the reference names (quant_cuda.cpp:257-270) ----------------------------------------------


def vecquant3matmul_nuq_perchannel(vec, mat, mul, lookup_table):
    """mul += W3(mat, lookup_table) . vec   (quant_cuda.cpp:112-118)"""
    _dense(3, False, vec, mat, mul, lookup_table)


def vecquant4matmul_nuq_perchannel(vec, mat, mul, lookup_table):
    """mul += W4(mat, lookup_table) . vec   (quant_cuda.cpp:119-125)"""
    _dense(4, False, vec, mat, mul, lookup_table)


def vecquant3matmul_nuq_perchannel_batched(vec, mat, mul, lookup_table):
    """mul[b] += W3 . vec[b]   (quant_cuda.cpp:126-132)"""
    _dense(3, True, vec, mat, mul, lookup_table)


def vecquant4matmul_nuq_perchannel_batched(vec, mat, mul, lookup_table):
    """mul[b] += W4 . vec[b]   (quant_cuda.cpp:133-139)"""
    _dense(4, True, vec, mat, mul, lookup_table)


def vecquant3matmul_spmv_nuq_perchannel(rows, cols, mat, vec, mul, num_rows, mat3, lookup_table):
    """mul += W3 . vec + CSR(rows, cols, mat) . vec   (quant_cuda.cpp:141-153)"""
    _spmv(3, False, rows, cols, mat, vec, mul, num_rows, mat3, lookup_table)


def vecquant4matmul_spmv_nuq_perchannel(rows, cols, mat, vec, mul, num_rows, mat4, lookup_table):
    """mul += W4 . vec + CSR . vec   (quant_cuda.cpp:154-166)"""
    _spmv(4, False, rows, cols, mat, vec, mul, num_rows, mat4, lookup_table)


def vecquant3matmul_spmv_nuq_perchannel_batched(rows, cols, mat, vec, mul, num_rows, mat3, lookup_table):
    """batched form   (quant_cuda.cpp:168-180)"""
    _spmv(3, True, rows, cols, mat, vec, mul, num_rows, mat3, lookup_table)


def vecquant4matmul_spmv_nuq_perchannel_batched(rows, cols, mat, vec, mul, num_rows, mat4, lookup_table):
    """batched form   (quant_cuda.cpp:181-193)"""
    _spmv(4, True, rows, cols, mat, vec, mul, num_rows, mat4, lookup_table)


def vecquant3matmul_spmv_hybrid_nuq_perchannel(rows, cols, mat, vec, full_rows, full_row_indices, mul, num_rows, mat3, lookup_table):
    """mul += W3 . vec + CSR . vec + full_rows^T . vec scattered to full_row_indices   (quant_cuda.cpp:195-209)"""
    _hybrid(3, False, rows, cols, mat, vec, full_rows, full_row_indices, mul, num_rows, mat3, lookup_table)


def vecquant4matmul_spmv_hybrid_nuq_perchannel(rows, cols, mat, vec, full_rows, full_row_indices, mul, num_rows, mat4, lookup_table):
    """(quant_cuda.cpp:210-224)"""
    _hybrid(4, False, rows, cols, mat, vec, full_rows, full_row_indices, mul, num_rows, mat4, lookup_table)


def vecquant3matmul_spmv_hybrid_nuq_perchannel_batched(rows, cols, mat, vec, full_rows, full_row_indices, mul, num_rows, mat3, lookup_table):
    """(quant_cuda.cpp:226-240)"""
    _hybrid(3, True, rows, cols, mat, vec, full_rows, full_row_indices, mul, num_rows, mat3, lookup_table)


def vecquant4matmul_spmv_hybrid_nuq_perchannel_batched(rows, cols, mat, vec, full_rows, full_row_indices, mul, num_rows, mat4, lookup_table):
    """(quant_cuda.cpp:241-255)"""
    _hybrid(4, True, rows, cols, mat, vec, full_rows, full_row_indices, mul, num_rows, mat4, lookup_table)


# ---- the two names quant.py calls for balanced=True layers (never defined by the reference) -----


def vecquant3matmul_spmv_balanced_nuq_perchannel(rows, cols, startrows, vals, vec, mul, mat3, lookup_table, outfeatures, num_threads, numvals):
    """Call site squeezellm/quant.py:237-250.  Same result as the spmv op; `startrows` /
    `num_threads` (the reference's intended per-thread partition, quant.py:139-169) are accepted
    and ignored because the kernel balances by nnz itself."""
    _balanced(3, rows, cols, startrows, vals, vec, mul, mat3, lookup_table, outfeatures, num_threads, numvals)


def vecquant4matmul_spmv_balanced_nuq_perchannel(rows, cols, startrows, vals, vec, mul, mat4, lookup_table, outfeatures, num_threads, numvals):
    """Call site squeezellm/quant.py:281-294."""
    _balanced(4, rows, cols, startrows, vals, vec, mul, mat4, lookup_table, outfeatures, num_threads, numvals)
