"""Python side of the MEASUREMENT library (libsqllm_hip_ablation.so: python -m squeezellm_amd.build --ablation).

Nothing here is product: these are the kernels that were built, parity-tested, measured on MI355X and not adopted,
kept loadable so that the numbers quoted for them (profiles/) stay reproducible.  The measurement library
is a second shared object with the same C ABI plus the experimental entry points; it is loaded BESIDE the product
library (its own handle, its own option state) and never by the operator module.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_int, c_int32, c_void_p

import torch

from . import _lib, build
from ._lib import SqllmOp

P = c_void_p


class SqllmPass(ctypes.Structure):
    """struct sqllm_pass (include/sqllm_hip.h): plain data describing a built pass workspace."""

    _fields_ = [("workspace", c_void_p), ("workspace_bytes", ctypes.c_int64), ("segs_offset", ctypes.c_int64),
                ("items_offset", ctypes.c_int64), ("state_bytes", c_int32), ("bits", c_int32), ("n_groups", c_int32),
                ("n_ops", c_int32), ("n_items", c_int32), ("grid", c_int32), ("poll_sleep", c_int32), ("timeout_ms", c_int32)]



_SIGNATURES = {
    "sqllm_pass_workspace_bytes": [POINTER(SqllmOp), POINTER(c_int32), c_int32],
    "sqllm_pass_plan": [POINTER(SqllmOp), POINTER(c_int32), c_int32, P, ctypes.c_int64, P, POINTER(SqllmPass)],
    "sqllm_pass_build": [POINTER(SqllmOp), POINTER(c_int32), c_int32, P, ctypes.c_int64, POINTER(SqllmPass)],
    "sqllm_pass_launch": [POINTER(SqllmPass), P],
    "sqllm_pass_status": [POINTER(SqllmPass), P, POINTER(c_int32), POINTER(c_int32)],
    "sqllm_pass_profile": [POINTER(SqllmPass), P, c_int32, POINTER(ctypes.c_float)],
}
_xlib = None


def load() -> ctypes.CDLL:
    """dlopen the measurement library (building it first where hipcc is present and it is stale)."""
    global _xlib
    if _xlib is not None:
        return _xlib
    _lib.load()  # (torch's HIP runtime first, and the product's prototypes)
    try:
        build.build_ablation(force=False)
    except build.HipccMissing:
        pass
    if not os.path.exists(build.ABLATION_LIB_PATH):
        raise RuntimeError(f"{build.ABLATION_LIB_PATH} is missing: python -m squeezellm_amd.build --ablation")
    lib = ctypes.CDLL(build.ABLATION_LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int64 if name == "sqllm_pass_workspace_bytes" else c_int
    for name in ("sqllm_set_option", "sqllm_get_option", "sqllm_error_string"):
        fn = getattr(lib, name)
        fn.argtypes = _lib.SIGNATURES[name]
        fn.restype = ctypes.c_char_p if name == "sqllm_error_string" else c_int
    lib.sqllm_debug_set_timeline.argtypes = [c_void_p]
    lib.sqllm_debug_set_timeline.restype = None
    _xlib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        kind = ValueError if code < 0 else RuntimeError
        raise kind(f"{what}: {load().sqllm_error_string(int(code)).decode()} (code {code})")


def set_option(name: str, value: int) -> None:
    check(load().sqllm_set_option(name.encode(), int(value)), f"sqllm_set_option({name})")


class GatedPass:
    """A decode pass as one persistent launch (include/sqllm_hip.h, "Dependency-gated pass"): the groups of an
    OpSequence in pass order, every group gated on the completion of the one before it exactly where it first
    reads vec.  Batch-1 operator sequences only."""

    def __init__(self, seq: OpSequence):
        if seq.linear:
            raise ValueError("the gated pass runs the fp32 operator ABI (not the fused fp16 linear)")
        self.seq = seq  # (keeps descriptors and tensors alive)
        self.device = seq.device
        self._lib = load()
        need = self._lib.sqllm_pass_workspace_bytes(seq.ops, seq._sizes, seq.n_groups)
        if need < 0:
            check(int(need), "sqllm_pass_workspace_bytes")
        self.workspace = torch.zeros(int(need), dtype=torch.uint8, device=self.device)
        self.desc = SqllmPass()
        with torch.cuda.device(self.device):
            rc = self._lib.sqllm_pass_build(seq.ops, seq._sizes, seq.n_groups, self.workspace.data_ptr(), int(need), ctypes.byref(self.desc))
        check(rc, "sqllm_pass_build")
        self.n_items, self.grid = self.desc.n_items, self.desc.grid

    def launch(self) -> None:
        """Enqueue the pass on the current stream: one memset node + one kernel."""
        rc = self._lib.sqllm_pass_launch(ctypes.byref(self.desc), torch.cuda.current_stream(self.device).cuda_stream)
        if rc != 0:
            check(rc, "sqllm_pass_launch")

    def status(self):
        """(error, item) of the last launch; synchronises the current stream.  error 0 = every gate opened."""
        err, item = ctypes.c_int32(0), ctypes.c_int32(0)
        rc = self._lib.sqllm_pass_status(ctypes.byref(self.desc), torch.cuda.current_stream(self.device).cuda_stream,
                                         ctypes.byref(err), ctypes.byref(item))
        check(rc, "sqllm_pass_status")
        return err.value, item.value

    def profile(self, reps: int = 3) -> float:
        """Average device-side duration of the pass kernel in microseconds (its own start / stop events)."""
        out = ctypes.c_float(0.0)
        rc = self._lib.sqllm_pass_profile(ctypes.byref(self.desc), torch.cuda.current_stream(self.device).cuda_stream, int(reps), ctypes.byref(out))
        check(rc, "sqllm_pass_profile")
        return float(out.value)

    def graph(self, warmup: int = 1) -> "torch.cuda.CUDAGraph":
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.launch()
        torch.cuda.current_stream(self.device).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.launch()
        return g
