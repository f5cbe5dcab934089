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

import weakref

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

# ---------------------------------------------------------------------------------------------------
# Host path.  The reference's wrapper is a device guard and a call (quant_cuda.cpp:112-125); this one is Python over
# ctypes, so what a call costs on the host decides the eager decode rate (224 calls per 7B token).  Three things keep
# it short:
#   * the PERSISTENT operands of a layer (qweight, lookup_table, rows / cols / vals, full_rows / full_row_indices --
#     the same tensor objects call after call) are validated once and remembered by object identity
#     (`_persistent`): pointer, shape facts and device come out of a dict; only vec and mul, which are new tensors
#     on every call, are checked every time;
#   * the current stream is read as a raw handle (torch._C._cuda_getCurrentRawStream) instead of building a
#     torch.cuda.Stream object, and the device guard is two integer compares unless vec lives on another device;
#   * the C functions are looked up once per (bits, op kind) and kept with their prototypes.
# Measured per call on the MI355X host (tools/host_cost.py): see DESIGN.md, "host path".
# ---------------------------------------------------------------------------------------------------
_raw_stream = torch._C._cuda_getCurrentRawStream
_get_device = torch._C._cuda_getDevice
_set_device = torch._C._cuda_setDevice

_fn_cache = {}


def _fn(name: str):
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(_lib.load(), name)
    return fn


def _dev_ptr(t, dtype, name: str) -> int:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t).__name__}")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor (quant_cuda has no CPU path), got device {t.device}")
    return t.data_ptr()


# id(tensor) -> (weakref to it, data_ptr, device index, shape) for operands that were validated once
_persistent = {}


def _persist(t, dtype, name: str):
    """(data_ptr, device index, shape) of a layer's persistent operand, validated on first sight.  The entry is
    dropped when the tensor dies (weakref callback) and re-validated if its storage moved."""
    e = _persistent.get(id(t))
    # a hit needs the same object, storage, ROLE (the dtype this operand must have) and shape: a tensor validated as int32
    # `cols` and later passed as `vals`, or resized in place (same data_ptr), is validated again
    if e is not None and e[0]() is t and e[1] == t.data_ptr() and e[4] is dtype and t.shape == e[3]:
        return e
    ptr = _dev_ptr(t, dtype, name)
    key = id(t)
    e = (weakref.ref(t, lambda _r, k=key: _persistent.pop(k, None)), ptr, t.get_device(), tuple(t.shape), dtype)
    _persistent[key] = e
    return e


# ---- caller workspace of the batched operators (include/sqllm_hip.h: sqllm_launch_ws) ----
# The reference's launchers allocate nothing (quant_cuda_kernel.cu:580-657); here a batched op with sparse terms reads a
# transposed copy of vec (and, wider, bf16 planes / slabs) out of a workspace.  This module owns ONE buffer per (device,
# stream), grown on demand, and launches through sqllm_launch_ws: the library then allocates nothing and never touches
# the default memory pool.  (A buffer is replaced, not freed early: torch's caching allocator keeps a block that was
# allocated on a stream alive for the work already enqueued there.)
# While the stream is CAPTURING the cached buffer is neither used nor grown: a graph bakes the raw address in, so the
# buffer of a capture must live exactly as long as the graph does.  The workspace of a captured call is therefore a
# temporary like any other tensor of the captured region -- torch's allocator takes it from the graph's private pool, which
# keeps it for the graph's replays and orders its reuse inside the capture -- and is not remembered here (a remembered
# one would be handed to the next graph captured on the same stream and dangle once the first graph is destroyed).
_workspaces = {}
_ws_need = {}
_tls = __import__("threading").local()  # one sqllm_op descriptor per Python thread (ctypes releases the GIL during the call)
_is_capturing = torch.cuda.is_current_stream_capturing


def _workspace(dev: int, stream: int, need: int):
    if _is_capturing():
        return torch.empty(need, dtype=torch.uint8, device=torch.device("cuda", dev))
    ws = _workspaces.get((dev, stream))
    if ws is None or ws.numel() < need:
        ws = _workspaces[(dev, stream)] = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=torch.device("cuda", dev))
    return ws


def _launch_ws(dev: int, bits, batch, K, width, pv, pq, pm, pl, csr=None, topx=None):
    """A batched op through sqllm_launch_ws with this module's workspace of (device, current stream)."""
    import ctypes

    lib = _lib.load()
    o = getattr(_tls, "op", None)
    if o is None:
        o = _tls.op = _lib.SqllmOp()
    o.bits, o.batch, o.K, o.N = bits, batch, K, width
    o.vec, o.qweight, o.mul, o.lookup_table = pv, pq, pm, pl
    o.rows, o.cols, o.vals, o.nnz = csr if csr is not None else (None, None, None, 0)
    o.full_rows, o.full_row_indices, o.topX = topx if topx is not None else (None, None, 0)
    prev = _get_device()
    if prev != dev:
        _set_device(dev)
    try:
        if _ws_need.get("epoch") != _lib.option_epoch:  # a library option changed: the routes (and what they can use) may have
            _ws_need.clear()
            _ws_need["epoch"] = _lib.option_epoch
        key = (dev, bits, batch, K, width, csr is not None and csr[3] > 0, topx is not None and topx[2] > 0)
        need = _ws_need.get(key)
        if need is None:
            need = _ws_need[key] = int(lib.sqllm_workspace_bytes(ctypes.byref(o), 1))
        stream = _raw_stream(dev)
        if need:
            ws = _workspace(dev, stream, need)
            rc = lib.sqllm_launch_ws(ctypes.byref(o), ws.data_ptr(), ws.numel(), stream)
        else:
            rc = lib.sqllm_launch_ws(ctypes.byref(o), None, 0, stream)
    finally:
        if prev != dev:
            _set_device(prev)
    if rc:
        _lib.check(rc, "sqllm_launch_ws")


def _io_type(vec, mul) -> None:
    """vec / mul are new tensors on every call, so every call checks them -- type and dtype first (as the reference's
    data_ptr<float>() would throw first), the rest in _io_ptr once the layer's device is known"""
    if not isinstance(vec, torch.Tensor) or not isinstance(mul, torch.Tensor):
        raise TypeError(f"vec and mul must be torch.Tensors, got {type(vec).__name__} and {type(mul).__name__}")
    if vec.dtype is not _F32 or mul.dtype is not _F32:
        raise TypeError(f"vec and mul must be {_F32}, got {vec.dtype} and {mul.dtype}")


def _io_ptr(t, name: str, dev: int) -> int:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor (quant_cuda has no CPU path), got device {t.device}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if t.get_device() != dev:
        raise RuntimeError(f"all operands must be on cuda:{dev}, found {name} on {t.device}")
    return t.data_ptr()


def _launch(fn, dev: int, args: tuple):
    """`const at::cuda::OptionalCUDAGuard device_guard(device_of(vec))` (quant_cuda.cpp:116) + the call: vec's
    device is made current for the launch (two integer compares when it already is), the launch goes to torch's
    current stream of that device."""
    prev = _get_device()
    if prev != dev:
        _set_device(dev)
        try:
            rc = fn(*args, _raw_stream(dev))
        finally:
            _set_device(prev)
    else:
        rc = fn(*args, _raw_stream(dev))
    if rc:
        _lib.check(rc, fn.__name__)


def _qweight(mat, bits: int, name: str):
    e = _persist(mat, _I32, name)
    shape = e[3]
    if len(shape) != 2:
        raise ValueError("qweight must be 2-D [K/32*bits, N]")
    if shape[0] % bits:
        raise ValueError(f"qweight has {shape[0]} rows, not a multiple of {bits}")
    return e[1], e[2], shape[0], shape[1], shape[0] // bits * 32


def _lut(lookup_table, width: int, bits: int, dev: int) -> int:
    e = _persist(lookup_table, _F32, "lookup_table")
    if e[3] != (width, 1 << bits):
        raise ValueError(f"lookup_table must be [{width}, {1 << bits}], got {e[3]}")
    if e[2] != dev:
        raise RuntimeError(f"all operands must be on cuda:{dev}, found lookup_table on cuda:{e[2]}")
    return e[1]


def _vec_mul(vec, mul, K: int, width: int, batched: bool, dev: int):
    pv, pm = _io_ptr(vec, "vec", dev), _io_ptr(mul, "mul", dev)
    if batched:
        if vec.dim() != 2 or vec.shape[1] != K:
            raise ValueError(f"vec must be [batch, {K}], got {tuple(vec.shape)}")
        batch = vec.shape[0]
        if mul.dim() != 2 or mul.shape[0] != batch or mul.shape[1] != width:
            raise ValueError(f"mul must be [{batch}, {width}], got {tuple(mul.shape)}")
        return pv, pm, batch
    if vec.numel() != K:
        raise ValueError(f"vec must have {K} elements, got {vec.numel()}")
    if mul.numel() != width:
        raise ValueError(f"mul must have {width} elements, got {mul.numel()}")
    return pv, pm, 0


def _csr(rows, cols, vals, num_rows, width: int, dev: int, vals_name: str = "mat (csr values)"):
    if int(num_rows) != width:
        raise ValueError(f"num_rows ({num_rows}) must equal outfeatures ({width})")
    er, ec, ev = _persist(rows, _I32, "rows"), _persist(cols, _I32, "cols"), _persist(vals, _F32, vals_name)
    n_rows = 1
    for d in er[3]:
        n_rows *= d
    if n_rows != width + 1:
        raise ValueError(f"rows must have {width + 1} entries, got {n_rows}")
    nnz = 1
    for d in ec[3]:
        nnz *= d
    n_vals = 1
    for d in ev[3]:
        n_vals *= d
    if nnz != n_vals:
        raise ValueError("cols and vals must have the same length")
    for e, nm in ((er, "rows"), (ec, "cols"), (ev, vals_name)):
        if e[2] != dev:
            raise RuntimeError(f"all operands must be on cuda:{dev}, found {nm} on cuda:{e[2]}")
    return er[1], ec[1], ev[1], nnz


_SFX = ("", "_batched")


def _dense(bits, batched, vec, mat, mul, lookup_table):
    _io_type(vec, mul)
    pq, dev, height, width, K = _qweight(mat, bits, "mat")
    pl = _lut(lookup_table, width, bits, dev)
    pv, pm, batch = _vec_mul(vec, mul, K, width, batched, dev)
    if batched and batch >= 64:  # (the wide form's planes / slabs; narrower dense ops need no workspace)
        return _launch_ws(dev, bits, batch, K, width, pv, pq, pm, pl)
    fn = _fn(f"sqllm_vecquant{bits}matmul_nuq_perchannel{_SFX[batched]}")
    _launch(fn, dev, (pv, pq, pm, pl, height, width, batch, K) if batched else (pv, pq, pm, pl, height, width))


def _spmv(bits, batched, rows, cols, mat, vec, mul, num_rows, matq, lookup_table):
    _io_type(vec, mul)
    pq, dev, height, width, K = _qweight(matq, bits, f"mat{bits}")
    pl = _lut(lookup_table, width, bits, dev)
    pr, pc, pvl, nnz = _csr(rows, cols, mat, num_rows, width, dev)
    pv, pm, batch = _vec_mul(vec, mul, K, width, batched, dev)
    if batched and batch >= 2:  # (what a batch needs beside its operands comes out of this module's workspace)
        return _launch_ws(dev, bits, batch, K, width, pv, pq, pm, pl, csr=(pr, pc, pvl, nnz))
    fn = _fn(f"sqllm_vecquant{bits}matmul_spmv_nuq_perchannel{_SFX[batched]}")
    head = (pr, pc, pvl, pv, pm, int(num_rows), pq, pl, height, width, nnz)
    _launch(fn, dev, head + (batch, K) if batched else head)


def _hybrid(bits, batched, rows, cols, mat, vec, full_rows, full_row_indices, mul, num_rows, matq, lookup_table):
    _io_type(vec, mul)
    pq, dev, height, width, K = _qweight(matq, bits, f"mat{bits}")
    pl = _lut(lookup_table, width, bits, dev)
    pr, pc, pvl, nnz = _csr(rows, cols, mat, num_rows, width, dev)
    efr, efi = _persist(full_rows, _F32, "full_rows"), _persist(full_row_indices, _I32, "full_row_indices")
    if len(efr[3]) != 2 or efr[3][0] != K:
        raise ValueError(f"full_rows must be [{K}, topX], got {efr[3]}")
    topX = efr[3][1]
    n_idx = 1
    for d in efi[3]:
        n_idx *= d
    if n_idx != topX:
        raise ValueError(f"full_row_indices must have {topX} entries, got {n_idx}")
    if efr[2] != dev or efi[2] != dev:
        raise RuntimeError(f"all operands must be on cuda:{dev}, found full_rows / full_row_indices elsewhere")
    pv, pm, batch = _vec_mul(vec, mul, K, width, batched, dev)
    if batched and batch >= 2:
        return _launch_ws(dev, bits, batch, K, width, pv, pq, pm, pl, csr=(pr, pc, pvl, nnz), topx=(efr[1], efi[1], topX))
    fn = _fn(f"sqllm_vecquant{bits}matmul_spmv_hybrid_nuq_perchannel{_SFX[batched]}")
    head = (pr, pc, pvl, pv, efr[1], efi[1], pm, int(num_rows), pq, pl, height, width, nnz, topX)
    _launch(fn, dev, head + (batch, K) if batched else head)


def _balanced(bits, rows, cols, startrows, vals, vec, mul, matq, lookup_table, outfeatures, num_threads, numvals):
    _io_type(vec, mul)
    pq, dev, height, width, K = _qweight(matq, bits, f"mat{bits}")
    pl = _lut(lookup_table, width, bits, dev)
    pr, pc, pvl, nnz = _csr(rows, cols, vals, outfeatures, width, dev, "vals")
    if int(numvals) != nnz:
        raise ValueError(f"numvals ({numvals}) != number of stored values ({nnz})")
    psr = _persist(startrows, _I32, "startrows")[1] if startrows is not None else None
    pv, pm, _ = _vec_mul(vec, mul, K, width, False, dev)
    fn = _fn(f"sqllm_vecquant{bits}matmul_spmv_balanced_nuq_perchannel")
    _launch(fn, dev, (pr, pc, psr, pvl, pv, pm, pq, pl, int(outfeatures), int(num_threads), nnz, height, width))


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
