"""ctypes binding of libsqllm_hip.so (C ABI: include/sqllm_hip.h).

The library is the product; there is NO CPU or PyTorch fallback.  If it has not been built
(`python -m squeezellm_amd.build`) loading raises, and every operator call raises with it.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_int, c_int32, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
# SQLLM_LIB: measurement aid -- load a variant build of the same library (tools/ab_libs.sh) instead
LIB_PATH = os.environ.get("SQLLM_LIB") or os.path.join(HERE, "libsqllm_hip.so")


class SqllmOp(ctypes.Structure):
    """struct sqllm_op (include/sqllm_hip.h)."""

    _fields_ = [
        ("bits", c_int32),
        ("batch", c_int32),
        ("K", c_int32),
        ("N", c_int32),
        ("vec", c_void_p),
        ("qweight", c_void_p),
        ("mul", c_void_p),
        ("lookup_table", c_void_p),
        ("rows", c_void_p),
        ("cols", c_void_p),
        ("vals", c_void_p),
        ("nnz", c_int32),
        ("topX", c_int32),
        ("full_rows", c_void_p),
        ("full_row_indices", c_void_p),
    ]


class SqllmLinear(ctypes.Structure):
    """struct sqllm_linear (include/sqllm_hip.h): op.vec / op.mul carry fp16 pointers."""

    _fields_ = [("op", SqllmOp), ("bias", c_void_p), ("workspace", c_void_p)]


class SqllmPlan(ctypes.Structure):
    """struct sqllm_plan (include/sqllm_hip.h)."""

    _fields_ = [(n, c_int32) for n in (
        "col_tiles", "k_slices", "groups_per_wave", "dense_blocks", "csr_blocks", "topx_blocks", "grid_x", "grid_y")]


P = c_void_p  # every device pointer crosses as void*

_DENSE = [P, P, P, P, c_int, c_int, P]
_DENSE_B = [P, P, P, P, c_int, c_int, c_int, c_int, P]
_SPMV = [P, P, P, P, P, c_int, P, P, c_int, c_int, c_int, P]
_SPMV_B = [P, P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, P]
_HYB = [P, P, P, P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, P]
_HYB_B = [P, P, P, P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]
_BAL = [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]

# every symbol include/sqllm_hip.h declares -> argtypes (restype is int unless noted)
SIGNATURES = {
    "sqllm_launch": [POINTER(SqllmOp), P],
    "sqllm_launch_sequence": [POINTER(SqllmOp), c_int32, P, POINTER(c_int32)],
    "sqllm_profile_sequence": [POINTER(SqllmOp), c_int32, P, c_int32, POINTER(ctypes.c_float)],
    "sqllm_launch_group": [POINTER(SqllmOp), c_int32, P],
    "sqllm_launch_groups": [POINTER(SqllmOp), POINTER(c_int32), c_int32, P, POINTER(c_int32)],
    "sqllm_profile_groups": [POINTER(SqllmOp), POINTER(c_int32), c_int32, P, c_int32, POINTER(ctypes.c_float)],
    "sqllm_workspace_bytes": [POINTER(SqllmOp), c_int32],
    "sqllm_launch_ws": [POINTER(SqllmOp), P, ctypes.c_int64, P],
    "sqllm_launch_group_ws": [POINTER(SqllmOp), c_int32, P, ctypes.c_int64, P],
    "sqllm_launch_groups_ws": [POINTER(SqllmOp), POINTER(c_int32), c_int32, P, ctypes.c_int64, P, POINTER(c_int32)],
    "sqllm_profile_groups_ws": [POINTER(SqllmOp), POINTER(c_int32), c_int32, P, ctypes.c_int64, P, c_int32, POINTER(ctypes.c_float)],
    "sqllm_linear_workspace_bytes": [POINTER(SqllmOp)],
    "sqllm_linear_f16": [POINTER(SqllmLinear), P],
    "sqllm_linear_f16_groups": [POINTER(SqllmLinear), POINTER(c_int32), c_int32, P, POINTER(c_int32)],
    "sqllm_abi_version": [],
    "sqllm_error_string": [c_int],
    "sqllm_set_option": [c_char_p, c_int],
    "sqllm_get_option": [c_char_p, POINTER(c_int)],
    "sqllm_plan_query": [POINTER(SqllmOp), POINTER(SqllmPlan)],
}
for _b in (3, 4):
    SIGNATURES[f"sqllm_vecquant{_b}matmul_nuq_perchannel"] = _DENSE
    SIGNATURES[f"sqllm_vecquant{_b}matmul_nuq_perchannel_batched"] = _DENSE_B
    SIGNATURES[f"sqllm_vecquant{_b}matmul_spmv_nuq_perchannel"] = _SPMV
    SIGNATURES[f"sqllm_vecquant{_b}matmul_spmv_nuq_perchannel_batched"] = _SPMV_B
    SIGNATURES[f"sqllm_vecquant{_b}matmul_spmv_hybrid_nuq_perchannel"] = _HYB
    SIGNATURES[f"sqllm_vecquant{_b}matmul_spmv_hybrid_nuq_perchannel_batched"] = _HYB_B
    SIGNATURES[f"sqllm_vecquant{_b}matmul_spmv_balanced_nuq_perchannel"] = _BAL

_lib = None


def load() -> ctypes.CDLL:
    """dlopen the library once and attach prototypes.  Raises if it is missing: by design."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64; it must be the HIP runtime this library binds to
    # (device pointers and streams come from torch), so it has to be loaded first
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -m squeezellm_amd.build). "
            "There is no CPU/PyTorch fallback for the quant_cuda operators."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        if os.environ.get("SQLLM_LIB") and not hasattr(lib, name):
            continue  # (measurement aid: an older build of the library under SQLLM_LIB lacks the newer entry points)
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.argtypes = argtypes
        fn.restype = (c_char_p if name == "sqllm_error_string" else
                      ctypes.c_int64 if name in ("sqllm_linear_workspace_bytes", "sqllm_workspace_bytes") else c_int)
    if lib.sqllm_abi_version() != 1:
        raise RuntimeError(f"libsqllm_hip.so ABI {lib.sqllm_abi_version()} != 1 expected by this package")
    _lib = lib
    # SQLLM_OPTIONS="name=value,..." (measurement aid): library options applied to the current device at load
    for item in filter(None, os.environ.get("SQLLM_OPTIONS", "").split(",")):
        name, _, value = item.partition("=")
        rc = lib.sqllm_set_option(name.strip().encode(), int(value))
        if rc != 0:
            raise RuntimeError(f"SQLLM_OPTIONS: {item!r} rejected ({rc})")
    return lib


def error_string(code: int) -> str:
    return load().sqllm_error_string(int(code)).decode()


def check(code: int, what: str) -> None:
    if code != 0:
        kind = ValueError if code < 0 else RuntimeError
        raise kind(f"{what}: {error_string(code)} (code {code})")


option_epoch = 0  # bumped by every set_option: sizes cached on the Python side (quant_cuda's workspace needs) are per epoch


def set_option(name: str, value: int) -> None:
    global option_epoch
    check(load().sqllm_set_option(name.encode(), int(value)), f"sqllm_set_option({name})")
    option_epoch += 1


def get_option(name: str) -> int:
    v = c_int(0)
    check(load().sqllm_get_option(name.encode(), ctypes.byref(v)), f"sqllm_get_option({name})")
    return v.value


def linear_workspace_bytes(N: int, batch: int = 0) -> int:
    """Bytes of zero-filled device memory one fused linear of this shape needs (no GPU needed)."""
    op = SqllmOp(N=N, batch=batch)
    return int(load().sqllm_linear_workspace_bytes(ctypes.byref(op)))


def workspace_bytes(bits: int, K: int, N: int, batch: int, nnz: int = 0, topX: int = 0, n_ops: int = 1) -> int:
    """Bytes of caller workspace a batched op (or a group of `n_ops` such ops over one vec) can use (no GPU needed)."""
    ops = (SqllmOp * n_ops)()
    for op in ops:
        op.bits, op.batch, op.K, op.N, op.nnz, op.topX = bits, batch, K, N, nnz, topX
        op.vec = op.qweight = op.mul = op.lookup_table = 16  # (sizing looks at shapes and at which pointers are non-NULL)
        if nnz:
            op.rows = op.cols = op.vals = 16
        if topX:
            op.full_rows = op.full_row_indices = 16
    return int(load().sqllm_workspace_bytes(ops, n_ops))


def plan_query(bits: int, K: int, N: int, batch: int = 0, nnz: int = 0, topX: int = 0) -> dict:
    """Launch geometry the library would use (no GPU needed)."""
    op = SqllmOp(bits=bits, batch=batch, K=K, N=N, nnz=nnz, topX=topX)
    # planning only looks at shapes and at whether the sparse pointers are non-NULL
    if nnz:
        op.rows = op.cols = op.vals = 1
    if topX:
        op.full_rows = op.full_row_indices = 1
    plan = SqllmPlan()
    check(load().sqllm_plan_query(ctypes.byref(op), ctypes.byref(plan)), "sqllm_plan_query")
    return {n: getattr(plan, n) for n, _ in SqllmPlan._fields_}
