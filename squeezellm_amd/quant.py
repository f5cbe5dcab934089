"""QuantLinearLUT for MI355X -- host-side mirror of the reference layer
(/root/reference/squeezellm/quant.py:28-383): same constructor signature, same buffer names /
shapes / dtypes (so reference checkpoints load with `load_state_dict`), same choice of operator
per configuration and the same pre/post-processing around it.  The reference's own quant.py also
runs unchanged on top of the top-level `quant_cuda` shim; this module exists so
that the path can be exercised where the reference tree is absent, and to host the MI355X-only
conveniences (`from_operands`, `operands`).

Packing (`pack2`, quant.py:97-208) is offline tooling and out of scope here; see oracle/ for the
format restatement used by the tests.
"""
from __future__ import annotations

import ctypes
import math

import torch
import torch.nn as nn

from . import _lib, quant_cuda


class QuantLinearLUT(nn.Module):
    """Drop-in for the reference class of the same name (quant.py:28-95 buffers, :211-383 forward)."""

    def __init__(self, bits, infeatures, outfeatures, bias, include_sparse=False, numvals=0, topX=0,
                 balanced=False, num_nonzero_per_thread=10):
        super().__init__()
        if bits not in (3, 4):
            raise NotImplementedError("Only 3 and 4 bits is supported.")  # quant.py:42-43
        self.bits, self.infeatures, self.outfeatures = bits, infeatures, outfeatures
        self.include_sparse, self.numvals, self.topX, self.balanced = include_sparse, numvals, topX, balanced
        i32, f32 = torch.int32, torch.float32
        self.register_buffer("qweight", torch.zeros((infeatures // 32 * bits, outfeatures), dtype=i32))
        self.include_bias = bool(bias)
        if self.include_bias:
            self.register_buffer("bias", torch.zeros(outfeatures, dtype=f32))
        else:
            self.bias = None
        self.register_buffer("lookup_table", torch.zeros((outfeatures, 2**bits), dtype=f32))
        if numvals > 0:  # quant.py:66-71
            self.register_buffer("rows", torch.zeros(outfeatures + 1, dtype=i32))
            self.register_buffer("cols", torch.zeros(numvals, dtype=i32))
            self.register_buffer("vals", torch.zeros(numvals, dtype=f32))
        if topX > 0:  # quant.py:74-80
            self.register_buffer("full_rows", torch.zeros((infeatures, topX), dtype=f32))
            self.register_buffer("full_row_indices", torch.zeros(topX, dtype=i32))
        if include_sparse and balanced and numvals > 0:  # quant.py:84-95
            nt = int((numvals + num_nonzero_per_thread - 1) / num_nonzero_per_thread)
            self.num_threads = 128 * math.ceil(nt / 128)
            self.register_buffer("startrows", torch.zeros(self.num_threads, dtype=i32))

    # -- which operator a configuration maps to: hybrid > balanced > spmv > dense (quant.py:224-265;
    #    the batched branch has no balanced arm, :322-349)
    def op_kind(self, batched: bool) -> str:
        if self.include_sparse and self.topX > 0:
            return "spmv_hybrid"
        if self.include_sparse and self.balanced and not batched:
            return "spmv_balanced"
        if self.include_sparse:
            return "spmv"
        return "dense"

    def _call(self, x32: torch.Tensor, y: torch.Tensor, batched: bool) -> None:
        kind = self.op_kind(batched)
        sfx = "_batched" if batched else ""
        b = self.bits
        if kind == "dense":
            getattr(quant_cuda, f"vecquant{b}matmul_nuq_perchannel{sfx}")(x32, self.qweight, y, self.lookup_table)
        elif kind == "spmv":
            getattr(quant_cuda, f"vecquant{b}matmul_spmv_nuq_perchannel{sfx}")(
                self.rows, self.cols, self.vals, x32, y, self.outfeatures, self.qweight, self.lookup_table)
        elif kind == "spmv_hybrid":
            getattr(quant_cuda, f"vecquant{b}matmul_spmv_hybrid_nuq_perchannel{sfx}")(
                self.rows, self.cols, self.vals, x32, self.full_rows, self.full_row_indices, y,
                self.outfeatures, self.qweight, self.lookup_table)
        else:
            getattr(quant_cuda, f"vecquant{b}matmul_spmv_balanced_nuq_perchannel")(
                self.rows, self.cols, self.startrows, self.vals, x32, y, self.qweight, self.lookup_table,
                self.outfeatures, self.num_threads, self.numvals)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        dtype = x.dtype
        if x.shape[-1] == x.numel():  # single token: the matvec ops (quant.py:212-312)
            y = self.bias.clone() if self.bias is not None else torch.zeros(
                self.outfeatures, device=x.device, dtype=torch.float32)
            self._call(x.float().contiguous(), y, batched=False)
            return y.to(dtype).reshape(*x.shape[:-1], self.outfeatures)
        # several rows: the *_batched ops; bias is added AFTER the cast back (quant.py:313-383)
        x2 = x.reshape(-1, x.shape[-1])
        out = torch.zeros((x2.shape[0], self.outfeatures), device=x.device, dtype=torch.float32)
        self._call(x2.float().contiguous(), out, batched=True)
        out = out.to(dtype).reshape(*x.shape[:-1], self.outfeatures)
        # `out + self.bias` exactly as the reference writes it (quant.py:382): with the fp32 bias buffer
        # the constructor registers, an fp16 result is promoted to fp32 (checked against the unmodified
        # reference module in tests/test_gpu_reference_forward.py)
        return out + self.bias if self.bias is not None else out

    # -- MI355X-side conveniences -------------------------------------------------------------
    @classmethod
    def from_operands(cls, layer: dict, balanced: bool = False) -> "QuantLinearLUT":
        """Wrap already-packed operands (e.g. squeezellm_amd.synth.make_layer) without copying."""
        nnz = 0 if layer.get("vals") is None else layer["vals"].numel()
        topX = 0 if layer.get("full_rows") is None else layer["full_rows"].shape[1]
        # a layer whose outliers ALL moved into full_rows (pack_layer with every outlier in <= topX
        # rows) has an empty CSR but still needs the hybrid op: the sparse path is on whenever either
        # term exists
        m = cls(layer["bits"], layer["K"], layer["N"], layer.get("bias") is not None,
                include_sparse=nnz > 0 or topX > 0, numvals=nnz, topX=topX, balanced=balanced)
        m.qweight, m.lookup_table = layer["qweight"], layer["lookup_table"]
        if layer.get("bias") is not None:
            m.bias = layer["bias"]
        if nnz:
            m.rows, m.cols, m.vals = layer["rows"], layer["cols"], layer["vals"]
            if balanced:
                m.startrows = torch.zeros(m.num_threads, dtype=torch.int32, device=layer["vals"].device)
        elif topX:  # empty CSR operands for the hybrid op (the constructor registers none for numvals == 0)
            dev = layer["qweight"].device
            m.rows = layer["rows"] if layer.get("rows") is not None else torch.zeros(layer["N"] + 1, dtype=torch.int32, device=dev)
            m.cols = torch.zeros(0, dtype=torch.int32, device=dev)
            m.vals = torch.zeros(0, dtype=torch.float32, device=dev)
        if topX:
            m.full_rows, m.full_row_indices = layer["full_rows"], layer["full_row_indices"]
        return m


_is_capturing = torch.cuda.is_current_stream_capturing


class QuantLinearLUTFused(QuantLinearLUT):
    """Opt-in forward for fp16 activations: ONE kernel per call (sqllm_linear_f16) instead of the
    reference's four (`zeros`/`bias.clone()`, `x.float()`, the op, `y.to(fp16)`; quant.py:214-223,
    :311-312 and :314-321, :380-383).  Same buffers and state dict as QuantLinearLUT -- switch an
    existing model over with `fuse_quant_lut(model)`.  Result = fp16(fp32 accumulation + bias);
    the reference's batched branch rounds to fp16 before adding the bias (and then promotes to
    fp32), so the two differ by at most one fp16 rounding of the output.  Other dtypes take the
    parent's path."""

    GRAPH_WS_MAX_BYTES = 4 << 20  # eager calls keep a second, graph-only workspace ready up to this size (decode batches)

    def _workspace(self, batch: int, device) -> torch.Tensor:
        """ONE zero-filled workspace per (device, stream), sized for the largest batch seen so far: a
        launch uses the first 8 * batch * N bytes and leaves them zero-filled, so smaller batches
        reuse the same buffer (a cache keyed by batch size would grow without bound under variable
        prompt lengths).  Two launches of one module that may overlap -- i.e. on different streams --
        must not share the accumulator words, hence the stream in the key.

        Captured calls: a graph bakes the workspace's address in, and the capture stream is never the stream of the
        warm-up calls, so a buffer keyed on it would be allocated AND zero-filled inside the capture -- a fill kernel in
        front of every linear of the graph, replayed every time (that was 25 % of a replayed 7B pass: 484 against 603
        tokens/s for the same kernels as a C-ABI sequence, BENCH_r05 `drop_in`).  Every eager call therefore also keeps
        a graph-only workspace of its size ready (key (device, "graph"); up to GRAPH_WS_MAX_BYTES: decode batches), a
        captured call takes that one -- the graph then holds the linear's kernel and nothing else -- and superseded
        buffers are retired, not freed (older graphs may still point at them).  All graphs captured from this module
        share it: replaying two of them CONCURRENTLY on different streams is not supported (set GRAPH_WS_MAX_BYTES = 0
        on the module for private, in-graph workspaces).  Without a prepared buffer (no eager call of this size before
        the capture) the workspace is a zero-filled temporary of the captured region."""
        cache = self.__dict__.setdefault("_ws", {})
        need = _lib.linear_workspace_bytes(self.outfeatures, batch)
        if _is_capturing():
            ws = cache.get((device, "graph"))
            if ws is not None and ws.numel() >= need:
                return ws
            return torch.zeros(need, dtype=torch.uint8, device=device)  # (not remembered: it belongs to this graph's pool)
        key = (device, quant_cuda._raw_stream(device.index if device.index is not None else torch.cuda.current_device()))
        ws = cache.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.zeros(need, dtype=torch.uint8, device=device)  # zero-filled once
            cache[key] = ws
        if need <= self.GRAPH_WS_MAX_BYTES:
            gws = cache.get((device, "graph"))
            if gws is None or gws.numel() < need:
                if gws is not None:
                    cache.setdefault("retired", []).append(gws)
                cache[(device, "graph")] = torch.zeros(need, dtype=torch.uint8, device=device)
        return ws

    def _check_csr_once(self) -> None:
        """The fused kernel detects completion by COUNTING the contributions `rows` announces: an
        inconsistent CSR (rows not non-decreasing, rows[N] != nnz -- e.g. buffers not loaded yet)
        would leave columns unfinished and the shared workspace dirty for every later call.  Checked
        once per module and CSR buffer (one device round trip), so that it fails loudly instead."""
        key = (self.rows.data_ptr(), self.rows._version, self.vals.data_ptr(), self.vals.numel())
        if self.__dict__.get("_csr_ok") == key:
            return
        if torch.cuda.is_current_stream_capturing():
            return  # the check reads back from the device, which would invalidate the capture: deferred to the next eager call
        r = self.rows
        ok = r.numel() == self.outfeatures + 1 and bool((r[1:] >= r[:-1]).all()) and int(r[0]) == 0 and int(r[-1]) == self.vals.numel()
        if not ok:
            raise ValueError("QuantLinearLUTFused: inconsistent CSR operands (rows must be non-decreasing with rows[0] == 0 and "
                             "rows[N] == vals.numel()); the fused kernel counts contributions from `rows`")
        self.__dict__["_csr_ok"] = key

    fold_topx = True  # fold the top-X dense rows into the CSR the fused kernel is given (set False to pass them separately)

    def _csr_with_topx(self):
        """(rows, cols, vals) with the top-X rows folded in (decode.fold_topx_into_csr), rebuilt whenever one of the
        buffers it was built from is replaced OR written in place (load_state_dict copies into the same storage: the
        key carries the tensors' version counters); not built while the stream is capturing -- that call then passes
        the top-X rows separately."""
        from . import decode

        has_csr = self.numvals > 0
        src = [self.full_rows, self.full_row_indices] + ([self.rows, self.cols, self.vals] if has_csr else [])
        key = tuple((t.data_ptr(), t._version) for t in src) + (self.numvals,)
        hit = self.__dict__.get("_folded")
        if hit is not None and hit[0] == key:
            return hit[1]
        if torch.cuda.is_current_stream_capturing():
            return None
        lay = dict(N=self.outfeatures, full_rows=self.full_rows, full_row_indices=self.full_row_indices)
        if has_csr:
            lay.update(rows=self.rows, cols=self.cols, vals=self.vals)
        out = decode.fold_topx_into_csr(lay)
        self.__dict__["_folded"] = (key, out)
        return out

    def _descriptor(self, dev: int, stream: int):
        """The pre-marshalled sqllm_linear of this module for (device, stream): every persistent field filled in once --
        weights, codebook, the (folded) sparse operands, bias -- and re-used call after call; vec / mul, the batch and the
        workspace pointer are set per call (ONE entry per device and stream whatever row counts arrive: an entry per
        batch would pin a superseded workspace each and grow without bound under variable prompt lengths).  Rebuilt when
        a buffer is replaced, moved or written in place (identity, storage and version counter of every buffer it was
        built from) or when a routing attribute changes (fold_topx, include_sparse, topX, numvals)."""
        # (straight from the module's buffer dict: nn.Module.__getattr__ costs ~0.5 us per buffer, ten times a dict lookup)
        bufs = self.__dict__["_buffers"]
        key = (tuple((id(t), t.data_ptr(), t._version) for t in bufs.values() if t is not None),
               self.fold_topx, self.include_sparse, self.topX, self.numvals)
        cache = self.__dict__.setdefault("_desc", {})
        hit = cache.get((dev, stream))
        if hit is not None and hit[0] == key:
            return hit[1]
        capturing = torch.cuda.is_current_stream_capturing()
        K, N = self.infeatures, self.outfeatures
        lin = _lib.SqllmLinear()
        o = lin.op
        o.bits, o.K, o.N = self.bits, K, N
        o.qweight, o.lookup_table = self.qweight.data_ptr(), self.lookup_table.data_ptr()
        if self.include_sparse and self.numvals > 0:
            self._check_csr_once()
        folded = self._csr_with_topx() if self.include_sparse and self.topX > 0 and self.fold_topx else None
        keep = [folded]
        if folded is not None:  # one CSR term that contains the top-X rows (decode.fold_topx_into_csr)
            if folded[2].numel():
                o.rows, o.cols, o.vals, o.nnz = folded[0].data_ptr(), folded[1].data_ptr(), folded[2].data_ptr(), folded[2].numel()
        else:
            if self.include_sparse and self.numvals > 0:
                o.rows, o.cols, o.vals, o.nnz = self.rows.data_ptr(), self.cols.data_ptr(), self.vals.data_ptr(), self.vals.numel()
            if self.include_sparse and self.topX > 0:  # independent of the CSR term (which may be empty)
                o.full_rows, o.full_row_indices, o.topX = self.full_rows.data_ptr(), self.full_row_indices.data_ptr(), self.topX
        lin.bias = None if self.bias is None else self.bias.data_ptr()
        entry = (key, (lin, ctypes.byref(lin), keep))
        # (while a stream is capturing, the folded CSR and the CSR check are deferred: do not pin that state)
        if not (capturing and self.include_sparse):
            cache[(dev, stream)] = entry
        return entry[1]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dtype is not torch.float16 or not x.is_cuda:
            return super().forward(x)
        K, N = self.infeatures, self.outfeatures
        if x.shape[-1] != K:
            raise ValueError(f"last dimension of x must be {K}, got {tuple(x.shape)}")
        x2 = x if x.dim() == 2 else x.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        rows = x2.shape[0]
        dev = x.get_device()
        out = torch.empty((rows, N), dtype=torch.float16, device=x.device)
        lin, ref, _keep = self._descriptor(dev, quant_cuda._raw_stream(dev))
        batch = 0 if rows == 1 else rows
        ws = self._workspace(batch, self.qweight.device)  # (one buffer per device and stream, grown to the largest batch seen)
        o = lin.op
        o.batch, o.vec, o.mul = batch, x2.data_ptr(), out.data_ptr()
        lin.workspace = ws.data_ptr()
        quant_cuda._launch(quant_cuda._fn("sqllm_linear_f16"), dev, (ref,))
        return out.reshape(*x.shape[:-1], N)


def fuse_quant_lut(module: nn.Module) -> int:
    """Switch every QuantLinearLUT under `module` to the fused fp16 forward (in place, buffers and
    state dict untouched).  Returns the number of layers switched."""
    n = 0
    for m in module.modules():
        if type(m) is QuantLinearLUT:
            m.__class__ = QuantLinearLUTFused
            n += 1
    return n


def make_quant_lut(module, names, bits, name="", include_sparse=False, numvals=None, topX=0, balanced=False,
                   num_nonzero_per_thread=10):
    """Swap the nn.Linear children listed in `names` for QuantLinearLUT, recursively
    (reference: quant.py:386-435)."""
    if isinstance(module, QuantLinearLUT):
        return
    for child_name, child in list(module.named_children()):
        full = f"{name}.{child_name}" if name else child_name
        if full in names:
            setattr(module, child_name, QuantLinearLUT(
                bits, child.in_features, child.out_features, child.bias is not None,
                include_sparse=include_sparse, numvals=(numvals[full] if numvals is not None else 0),
                topX=topX, balanced=balanced, num_nonzero_per_thread=num_nonzero_per_thread))
        else:
            make_quant_lut(child, names, bits, full, include_sparse=include_sparse, numvals=numvals, topX=topX,
                           balanced=balanced, num_nonzero_per_thread=num_nonzero_per_thread)
