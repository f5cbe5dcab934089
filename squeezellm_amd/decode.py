"""Whole-pass launchers: a decode step over a stack of QuantLinearLUT layers as ONE host call.

The reference pays a Python -> pybind -> launch round trip per operator (and 1-3 launches inside
each, quant_cuda_kernel.cu:439-506); at batch 1 the 7B model's per-linear budget is ~1-4 us, far
below an eager launch (~3-4 us of host time).  `OpSequence` pre-marshals every operand once into a
C array of `sqllm_op` descriptors (include/sqllm_hip.h) and then
  * `launch()`  enqueues the whole pass through one FFI crossing (sqllm_launch_sequence), or
  * `graph()`   captures that into a HIP graph for replay.
With `linear=True` the ops are fused linears (sqllm_linear_f16): fp16 activations in, fp16 out,
bias included, no zero-fill / cast launches around them -- the whole matvec branch of
QuantLinearLUT.forward (squeezellm/quant.py:211-312) per kernel.
All launches go to torch's current stream; nothing here synchronises.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


def _ptr(t):
    return None if t is None else t.data_ptr()


def fold_topx_into_csr(lay):
    """(rows, cols, vals) of a layer's CSR with its top-X dense rows folded back in as ordinary CSR rows (an entry
    present in both is summed), on the layer's device; cached in the layer dict under "csr_with_topx", keyed on the
    buffers it was built from (pointers, sizes, in-place version counters).

    The reference keeps the few densest outlier rows as dense `full_rows` because its SpMV walks one row per
    thread (quant_cuda_kernel.cu:1049-1058); the CSR role here is balanced by non-zeros, so a heavy row costs
    nothing extra, and the FUSED LINEAR -- which has to fold the top-X rows into its dense tiles, its most
    expensive term -- is faster with them in the CSR.  The operator path keeps the reference's
    operands as they are."""
    full_rows, full_idx = lay["full_rows"], lay["full_row_indices"]
    # the cache entry is only good for the very buffers (and contents) it was built from: a column shard that was
    # copied from this dict, or an in-place edit of the outlier values, must not see it
    src = [full_rows, full_idx] + ([lay["rows"], lay["cols"], lay["vals"]] if lay.get("vals") is not None else [])
    cache_key = (lay["N"],) + tuple((t.data_ptr(), t.numel(), t._version) for t in src)
    hit = lay.get("csr_with_topx")
    if hit is not None and hit[0] == cache_key:
        return hit[1]
    dev = full_rows.device
    N, K = lay["N"], full_rows.shape[0]
    if lay.get("vals") is not None and lay["vals"].numel():
        rows, cols, vals = lay["rows"], lay["cols"], lay["vals"]
        counts = (rows[1:] - rows[:-1]).to(torch.int64)
        rid = torch.repeat_interleave(torch.arange(N, device=dev), counts)
        cols = cols.to(torch.int64)
    else:
        rid = torch.zeros(0, dtype=torch.int64, device=dev)
        cols = torch.zeros(0, dtype=torch.int64, device=dev)
        vals = torch.zeros(0, dtype=full_rows.dtype, device=dev)
    nz = (full_rows != 0).nonzero()  # (k, slot)
    key = torch.cat([rid * K + cols, full_idx.to(torch.int64)[nz[:, 1]] * K + nz[:, 0]])
    v = torch.cat([vals, full_rows[nz[:, 0], nz[:, 1]]])
    order = torch.argsort(key, stable=True)
    uniq, inv = torch.unique_consecutive(key[order], return_inverse=True)  # duplicates (CSR and top-X, or repeated indices): summed
    v = torch.zeros(uniq.numel(), dtype=v.dtype, device=dev).index_add_(0, inv, v[order])
    new_rows = torch.zeros(N + 1, dtype=torch.int32, device=dev)
    new_rows[1:] = torch.bincount(uniq // K, minlength=N).cumsum(0).to(torch.int32)
    out = (new_rows, (uniq % K).to(torch.int32).contiguous(), v.contiguous())
    lay["csr_with_topx"] = (cache_key, out)
    return out


class OpSequence:
    """A fixed list of ops `ys[i] += layer_i(xs[i])` with all pointers resolved up front."""

    def __init__(self, layers, xs, ys, batched: bool = False, fuse_shared_input: bool = False,
                 linear: bool = False, fold_topx: bool = True, workspace: bool = True):
        """fuse_shared_input: consecutive ops that read the SAME x tensor (and agree in K, bits,
        batch) are enqueued as one kernel (sqllm_launch_group), up to 4 per launch -- q/k/v and
        gate/up of a decoder layer.
        linear: `ys[i] = fp16(layer_i(xs[i]) + bias_i)` with fp16 xs / ys (ys overwritten) instead
        of the operator semantics `ys[i] += layer_i(xs[i])` on fp32.
        fold_topx (fused linears only): hand the kernel a CSR that contains the layer's top-X rows
        (`fold_topx_into_csr`, built once per layer) instead of the separate dense rows.
        workspace (batched operator sequences): own ONE workspace buffer for the pass, sized by
        sqllm_workspace_bytes, and launch through the `_ws` entry points -- the library then allocates
        nothing (False: the workspace-less names)."""
        if not (len(layers) == len(xs) == len(ys)):
            raise ValueError("layers, xs, ys must have equal length")
        self.n = len(layers)
        self.linear = bool(linear)
        self._keep = [layers, xs, ys]  # keep the tensors alive as long as the descriptors
        self.device = xs[0].device if self.n else torch.device("cuda")
        io = torch.float16 if linear else torch.float32
        if linear:
            self.lins = (_lib.SqllmLinear * self.n)()
            self.ops = [self.lins[i].op for i in range(self.n)]
        else:
            self.ops = (_lib.SqllmOp * self.n)()
        for i, (lay, x, y) in enumerate(zip(layers, xs, ys)):
            for name, t, dt in (("x", x, io), ("y", y, io),
                                ("qweight", lay["qweight"], torch.int32),
                                ("lookup_table", lay["lookup_table"], torch.float32)):
                if t.dtype != dt or not t.is_cuda or not t.is_contiguous():
                    raise ValueError(f"op {i}: {name} must be a contiguous {dt} GPU tensor")
            K, N = lay["K"], lay["N"]
            batch = x.shape[0] if batched else 0
            if x.numel() != max(batch, 1) * K or y.numel() != max(batch, 1) * N:
                raise ValueError(f"op {i}: x/y sizes do not match K={K}, N={N}, batch={batch}")
            o = self.ops[i]
            o.bits, o.batch, o.K, o.N = lay["bits"], batch, K, N
            o.vec, o.qweight, o.mul, o.lookup_table = x.data_ptr(), lay["qweight"].data_ptr(), y.data_ptr(), lay["lookup_table"].data_ptr()
            if linear and fold_topx and lay.get("full_rows") is not None and lay["full_rows"].shape[1] > 0:
                rows, cols, vals = fold_topx_into_csr(lay)  # (kept alive by the layer dict in _keep)
                if vals.numel():
                    o.rows, o.cols, o.vals, o.nnz = rows.data_ptr(), cols.data_ptr(), vals.data_ptr(), vals.numel()
            else:
                if lay.get("vals") is not None:
                    o.rows, o.cols, o.vals = _ptr(lay["rows"]), _ptr(lay["cols"]), _ptr(lay["vals"])
                    o.nnz = lay["vals"].numel()
                if lay.get("full_rows") is not None:
                    o.full_rows, o.full_row_indices = _ptr(lay["full_rows"]), _ptr(lay["full_row_indices"])
                    o.topX = lay["full_rows"].shape[1]
            if linear:
                bias = lay.get("bias")
                if bias is not None and (bias.dtype != torch.float32 or not bias.is_cuda or bias.numel() != N):
                    raise ValueError(f"op {i}: bias must be an fp32 GPU tensor of {N} elements")
                ws = torch.zeros(_lib.linear_workspace_bytes(N, batch), dtype=torch.uint8, device=self.device)
                self._keep.append(ws)
                self.lins[i].bias = _ptr(bias)
                self.lins[i].workspace = ws.data_ptr()
        # launch groups: lists of consecutive op indices sharing one input vector
        self.groups = []
        for i in range(self.n):
            o = self.ops[i]
            if fuse_shared_input and self.groups and len(self.groups[-1]) < 4:
                p = self.ops[self.groups[-1][0]]
                if (p.vec, p.K, p.bits, p.batch) == (o.vec, o.K, o.bits, o.batch):
                    self.groups[-1].append(i)
                    continue
            self.groups.append([i])
        self.n_groups = len(self.groups)
        self._sizes = (ctypes.c_int32 * max(self.n_groups, 1))(*[len(g) for g in self.groups])
        self._lib = _lib.load()
        self._done = ctypes.c_int32(0)
        # one workspace for the whole pass (its groups run one after the other on one stream)
        self._ws = None
        if workspace and batched and not linear and self.n and hasattr(self._lib, "sqllm_workspace_bytes"):
            need, at = 0, 0
            for g in self.groups:
                first = ctypes.cast(ctypes.byref(self.ops, at * ctypes.sizeof(_lib.SqllmOp)), ctypes.POINTER(_lib.SqllmOp))
                need = max(need, int(self._lib.sqllm_workspace_bytes(first, len(g))))
                at += len(g)
            if need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)

    def launch(self) -> None:
        """Enqueue the whole pass on the current stream of the sequence's device (one FFI crossing)."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if self.linear:
            rc = self._lib.sqllm_linear_f16_groups(self.lins, self._sizes, self.n_groups, stream, ctypes.byref(self._done))
        elif self._ws is not None:
            rc = self._lib.sqllm_launch_groups_ws(self.ops, self._sizes, self.n_groups, self._ws.data_ptr(), self._ws.numel(), stream,
                                                  ctypes.byref(self._done))
        else:
            rc = self._lib.sqllm_launch_groups(self.ops, self._sizes, self.n_groups, stream, ctypes.byref(self._done))
        if rc != 0:
            _lib.check(rc, f"launch of group {self._done.value} of {self.n_groups}")

    def profile(self, reps: int = 3):
        """Per-LAUNCH kernel durations in microseconds (device-side begin->end of each dispatch, as
        a profiler would report them), one per entry of `self.groups`, averaged over `reps`
        passes.  Synchronises."""
        import numpy as np

        if self.linear:
            raise NotImplementedError("per-launch profiling is provided for operator sequences only")
        out = (ctypes.c_float * max(self.n_groups, 1))()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if self._ws is not None:
            rc = self._lib.sqllm_profile_groups_ws(self.ops, self._sizes, self.n_groups, self._ws.data_ptr(), self._ws.numel(), stream,
                                                   int(reps), out)
        else:
            rc = self._lib.sqllm_profile_groups(self.ops, self._sizes, self.n_groups, stream, int(reps), out)
        _lib.check(rc, "sqllm_profile_groups")
        out = (ctypes.c_float * self.n_groups).from_buffer(out)
        return np.ctypeslib.as_array(out).astype(np.float64).copy()

    def graph(self, warmup: int = 1) -> "torch.cuda.CUDAGraph":
        """Capture one pass into a HIP graph (replay with .replay())."""
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

