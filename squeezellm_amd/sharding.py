"""Layer-sharded decode over the GPUs of one node: one process per GPU, `torch.distributed` with
the "nccl" backend (= RCCL over xGMI on ROCm).

The reference has no multi-GPU code.  The quantised-linear path shards naturally
BY LAYER: every QuantLinearLUT is an independent unit, and the only thing that crosses a stage
boundary is the hidden state of the token being decoded (2 * hidden bytes in fp16: 8 KiB for 7B,
16 KiB for 65B).  BASELINE.json's north_star names an RCCL all-gather for this exchange.

Scheme (`RingPipeline`): rank r owns the contiguous layer range `partition_layers(L, W)[r]`.  W
independent sequences are decoded at once, one per stage, rotating around the ring:

    every tick, on every rank:   h_out = stage_r(h_in)                 # this rank's layers
                                 all_gather(buf[W, hidden], h_out)      # ONE small collective
                                 h_in  = buf[(r - 1) mod W]             # predecessor's output

After W ticks every sequence has passed through all W stages once, i.e. advanced by one token; the
output of the last stage wraps around to stage 0 as the next token's input (autoregressive ring).
So the job produces one token per tick, tick = (single-GPU pass time) / W + all-gather latency.
The payload is tiny (W * hidden * 2 bytes <= 128 KiB), hence latency-bound; xGMI bandwidth is
irrelevant here and a ring collective's per-link limit never shows.  Every rank executes the
identical sequence of collectives, so the schedule cannot deadlock.

The exchange logic is backend-agnostic and is covered on CPU with gloo (tests/test_sharding_cpu.py)
by injecting a CPU stage function; the product stage (`DecodeStage`) launches the HIP kernels.

Second split (path 2): BY OUTPUT COLUMN inside a layer.  `qweight[:, n0:n1]`,
`lookup_table[n0:n1]`, the CSR rows n0..n1 and the top-X rows whose column falls in the range
partition by N with NO reduction: every rank computes its slice of `mul` and ONE all-gather of the
slices (N * 4 bytes <= 88 KiB) rebuilds the vector (`shard_layer_columns`, `ColumnParallelOp`).  This
is the split that shortens a single token's latency (every GPU streams 1 / W of every layer), at the
price of one latency-bound collective per linear.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def partition_layers(n_layers: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced [start, end) layer ranges; the first n_layers % world_size ranks get
    one extra layer.  Ranks beyond n_layers get empty ranges."""
    if n_layers < 0 or world_size < 1:
        raise ValueError("need n_layers >= 0 and world_size >= 1")
    base, extra = divmod(n_layers, world_size)
    out, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < extra else 0)
        out.append((start, start + n))
        start += n
    return out


class RingPipeline:
    """The tick loop described in the module docstring.

    stage_fn(h_in) -> h_out : both 1-D tensors of `hidden` elements, dtype `dtype`, on `device`;
    it must enqueue its work on the current stream and must not synchronise."""

    def __init__(self, stage_fn: Callable[[torch.Tensor], torch.Tensor], hidden: int, *, rank: int, world_size: int,
                 device, dtype=torch.float16, group=None, h0: torch.Tensor | None = None):
        self.stage_fn, self.hidden, self.rank, self.world = stage_fn, hidden, rank, world_size
        self.group = group
        self.buf = torch.zeros((world_size, hidden), device=device, dtype=dtype)
        self.h_in = torch.zeros(hidden, device=device, dtype=dtype) if h0 is None else h0.to(device=device, dtype=dtype).clone()
        self._chunks = list(self.buf.unbind(0))
        self._use_flat = hasattr(dist, "all_gather_into_tensor") and (dist.get_backend(group) != "gloo" if dist.is_initialized() else True)
        self.ticks = 0
        self._graph, self._out = None, None

    def _tick_body(self) -> torch.Tensor:
        h_out = self.stage_fn(self.h_in)
        if self.world == 1:
            self.buf[0].copy_(h_out)
        elif self._use_flat:
            dist.all_gather_into_tensor(self.buf, h_out.contiguous(), group=self.group)
        else:  # gloo (CPU tests): list form
            dist.all_gather(self._chunks, h_out.contiguous(), group=self.group)
        self.h_in.copy_(self.buf[(self.rank - 1) % self.world])
        return h_out

    def capture(self) -> bool:
        """Capture one whole tick -- the stage's kernels, the all-gather (RCCL collectives can be captured) and
        the hand-over copy -- into ONE HIP graph: a tick then costs a single graph launch on the host instead of a
        graph launch + an eager collective + an eager copy (host-bound for short stages, e.g. 4 layers of a 7B
        model per GPU).  The stage function must be capturable as plain launches (a DecodeStage built with
        graph=False; a graph replay cannot be nested in a capture).  Returns False -- and leaves the eager tick in
        place -- if the backend refuses the capture."""
        if self._graph is not None:
            return True
        dev = self.buf.device
        if dev.type != "cuda":
            return False
        try:
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            keep = self.h_in.clone()
            with torch.cuda.stream(side):
                self._tick_body()  # warm-up (allocations, communicator set-up) outside the capture
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.h_in.copy_(keep)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._out = self._tick_body()
            self.h_in.copy_(keep)  # (capturing does not execute)
            self._graph = g
            return True
        except RuntimeError:
            self._graph = None
            torch.cuda.synchronize(dev)
            return False

    def tick(self) -> torch.Tensor:
        if self._graph is not None:
            self._graph.replay()
            h_out = self._out
        else:
            h_out = self._tick_body()
        self.ticks += 1
        return h_out

    def run(self, n_ticks: int) -> None:
        for _ in range(n_ticks):
            self.tick()


def column_ranges(N: int, world_size: int, align: int = 64) -> List[Tuple[int, int]]:
    """[n0, n1) of every rank: contiguous, multiples of `align` columns (the kernels' tile width; N % 4 == 0 is
    what they require), as equal as that allows; trailing ranks may be empty when N is small."""
    if N < 0 or world_size < 1 or align < 4 or align % 4:
        raise ValueError("need N >= 0, world_size >= 1, align a positive multiple of 4")
    blocks = -(-N // align)
    out = []
    for a, b in partition_layers(blocks, world_size):
        out.append((min(a * align, N), min(b * align, N)))
    return out


def shard_layer_columns(layer: dict, rank: int, world_size: int, align: int = 64) -> dict:
    """This rank's column slice of one quantised linear's operands (a dict as squeezellm_amd.synth / checkpoint
    produce them): `qweight[:, n0:n1]`, `lookup_table[n0:n1]`, bias, the CSR rows n0..n1 re-based to start at 0,
    and the top-X rows whose output column falls inside the range (indices re-based).  Pure tensor slicing: works
    on CPU and GPU tensors alike; the result is a valid operand set for every operator of the library."""
    n0, n1 = column_ranges(layer["N"], world_size, align)[rank]
    out = dict(layer)
    out.pop("csr_with_topx", None)  # (a derived entry of the unsharded layer: decode.fold_topx_into_csr rebuilds it for the shard)
    out["N"] = n1 - n0
    out["col_range"] = (n0, n1)
    out["qweight"] = layer["qweight"][:, n0:n1].contiguous()
    out["lookup_table"] = layer["lookup_table"][n0:n1].contiguous()
    if layer.get("bias") is not None:
        out["bias"] = layer["bias"][n0:n1].contiguous()
    if layer.get("vals") is not None:
        rows = layer["rows"]
        e0, e1 = int(rows[n0]), int(rows[n1])
        out["rows"] = (rows[n0:n1 + 1] - e0).contiguous()
        out["cols"] = layer["cols"][e0:e1].contiguous()
        out["vals"] = layer["vals"][e0:e1].contiguous()
    if layer.get("full_rows") is not None:
        idx = layer["full_row_indices"]
        keep = ((idx >= n0) & (idx < n1)).nonzero().flatten()
        if keep.numel():
            out["full_rows"] = layer["full_rows"][:, keep].contiguous()
            out["full_row_indices"] = (idx[keep] - n0).to(idx.dtype).contiguous()
        else:  # no top-X row lands in this range
            out["full_rows"], out["full_row_indices"] = None, None
    return out


class ColumnParallelOp:
    """One quantised linear split by output column over the ranks of a process group: `local_fn(x) -> y_slice`
    computes this rank's columns (the HIP kernels on the sharded operands, or a test double), one all-gather
    rebuilds the full `mul`.  Slices are padded to the widest one for the collective (ranges are equal up to
    one 64-column block)."""

    def __init__(self, local_fn: Callable[[torch.Tensor], torch.Tensor], N: int, *, rank: int, world_size: int, device,
                 dtype=torch.float32, group=None, align: int = 64):
        self.local_fn, self.N, self.rank, self.world, self.group = local_fn, N, rank, world_size, group
        self.ranges = column_ranges(N, world_size, align)
        self.width = max(b - a for a, b in self.ranges) if self.ranges else 0
        self.buf = torch.zeros((world_size, max(self.width, 1)), device=device, dtype=dtype)
        self.mine = torch.zeros(max(self.width, 1), device=device, dtype=dtype)
        self._chunks = list(self.buf.unbind(0))
        self._use_flat = hasattr(dist, "all_gather_into_tensor") and (dist.get_backend(group) != "gloo" if dist.is_initialized() else True)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        n0, n1 = self.ranges[self.rank]
        if n1 > n0:
            self.mine[: n1 - n0].copy_(self.local_fn(x).reshape(-1))
        if self.world == 1:
            self.buf[0].copy_(self.mine)
        elif self._use_flat:
            dist.all_gather_into_tensor(self.buf, self.mine, group=self.group)
        else:
            dist.all_gather(self._chunks, self.mine, group=self.group)
        return torch.cat([self.buf[r, : b - a] for r, (a, b) in enumerate(self.ranges)])


class ColumnParallelPass:
    """A whole decode pass with every linear split by output column over the ranks (the latency split): per
    launch group (q/k/v, o, gate/up, down) this rank's column slices run as ONE kernel launch into a
    contiguous arena slice, then ONE all-gather of that slice rebuilds the group's `mul` vectors on every rank.
    `layers`: the FULL operand dicts (identical on every rank: same seeds / same checkpoint); `xs`: one input
    tensor per linear (linears sharing an input share the tensor, as in decode.OpSequence).
    graph=True captures the pass -- kernels AND collectives -- into one HIP graph where the backend allows it
    (RCCL collectives are capturable); otherwise the pass runs eagerly, one FFI crossing + one collective per
    group."""

    def __init__(self, layers: Sequence[dict], xs: Sequence[torch.Tensor], *, rank: int, world_size: int, device, group=None,
                 graph: bool = True):
        from .decode import OpSequence

        self.rank, self.world, self.group, self.device = rank, world_size, group, device
        self._flat = hasattr(dist, "all_gather_into_tensor") and (dist.get_backend(group) != "gloo" if dist.is_initialized() else True)
        # launch groups = runs of consecutive linears reading the same input tensor (up to 4)
        groups, cur = [], []
        for i, x in enumerate(xs):
            if cur and len(cur) < 4 and xs[cur[0]] is x and layers[cur[0]]["K"] == layers[i]["K"]:
                cur.append(i)
            else:
                if cur:
                    groups.append(cur)
                cur = [i]
        if cur:
            groups.append(cur)
        self.groups = groups
        # Every group's gather buffer [world, W] is a view of ONE arena, and this rank's kernels accumulate straight into
        # ITS row of that buffer (the in-place form of the all-gather: input = output[rank]): no per-group staging
        # buffer, no per-group memset, no hand-over copy -- one memset of the arena per pass is all the `mul += ...`
        # semantics need.  (With a staging buffer, a memset and a copy per group the one-rank pass ran 30 % below the
        # replica line before any collective: 626 vs 897 tokens/s on 7b-w4-s0, round 3.)
        self.widths = [[max(b - a for a, b in column_ranges(layers[i]["N"], world_size)) for i in grp] for grp in groups]
        sizes = [world_size * sum(w) for w in self.widths]
        self.arena = torch.zeros(max(sum(sizes), 1), device=device, dtype=torch.float32)
        self.local, self.gathered, self.seqs = [], [], []
        at = 0
        for grp, widths, size in zip(groups, self.widths, sizes):
            full = self.arena[at:at + size].view(world_size, sum(widths))
            at += size
            mine = full[rank]
            shards = [shard_layer_columns(layers[i], rank, world_size) for i in grp]
            ys, off = [], 0
            for sh, w in zip(shards, widths):  # (every rank's slice of a linear is padded to the widest rank's: the collective is uniform)
                ys.append(mine[off:off + sh["N"]])
                off += w
            live = [(sh, xs[i], y) for sh, i, y in zip(shards, grp, ys) if sh["N"] > 0]
            self.seqs.append(OpSequence([t[0] for t in live], [t[1] for t in live], [t[2] for t in live], fuse_shared_input=True) if live else None)
            self.local.append(mine)
            self.gathered.append(full)
        self.graph = None
        if graph and torch.device(device).type == "cuda":
            try:
                side = torch.cuda.Stream(device)
                side.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(side):
                    self._eager()
                torch.cuda.current_stream(device).wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._eager()
                self.graph = g
            except RuntimeError:
                self.graph = None
                torch.cuda.synchronize(device)

    def _eager(self) -> None:
        self.arena.zero_()  # (operator semantics: mul += ...; once per pass, every group's slices at once)
        for seq, mine, full in zip(self.seqs, self.local, self.gathered):
            if seq is not None:
                seq.launch()
            if self.world > 1:
                if self._flat:
                    dist.all_gather_into_tensor(full, mine, group=self.group)  # in place: mine IS full[rank]
                else:  # gloo: list form (it copies the input into its own slot: a no-op move onto itself)
                    dist.all_gather(list(full.unbind(0)), mine.clone(), group=self.group)

    def step(self) -> None:
        if self.graph is not None:
            self.graph.replay()
        else:
            self._eager()

    def result(self, gi: int, j: int, N: int) -> torch.Tensor:
        """The full `mul` of the j-th linear of launch group gi (N columns), assembled from the gathered slices."""
        off = sum(self.widths[gi][:j])
        parts = [self.gathered[gi][r, off:off + (b - a)] for r, (a, b) in enumerate(column_ranges(N, self.world))]
        return torch.cat(parts)


class DecodeStage:
    """This rank's slice of a synthetic decoder stack as a stage function: hidden in -> the stage's
    quantised linears (one FFI crossing, squeezellm_amd.decode.OpSequence) -> hidden out.

    Dataflow of the synthetic stage: every linear whose K equals `hidden` reads the received hidden
    state (fp32 copy), the others read a fixed random activation of their own width; the stage
    output is the last linear whose N equals `hidden`, RMS-normalised so that values stay O(1)
    around the ring.  All `mul` buffers live in one arena that is zeroed once per tick
    (the `torch.zeros` of QuantLinearLUT.forward, quant.py:218)."""

    def __init__(self, layers: Sequence[dict], hidden: int, device, seed: int = 0, graph: bool = True,
                 dtype=torch.float16):
        """graph=True (GPU stages): the whole stage -- input cast, arena clear, every kernel of the
        pass, the norm and the output cast -- is captured ONCE into a HIP graph and replayed per
        tick from static input / output buffers, so a tick costs one graph launch on the host
        instead of ~10 eager torch calls (which would be host-bound at 4-10 layers per GPU).
        NOTE the aliasing that comes with it: with a graph, every call returns THE SAME output buffer
        (`out_static`), overwritten by the next call -- clone it to keep a tick's result (RingPipeline
        copies it into its gather buffer at once).  If the capture fails the stage runs eagerly."""
        from .decode import OpSequence

        self.hidden, self.device = hidden, device
        g = torch.Generator(device=device).manual_seed(seed)
        self.x_hidden = torch.zeros(hidden, device=device, dtype=torch.float32)
        total_n = sum(l["N"] for l in layers)
        self.arena = torch.zeros(max(total_n, 1), device=device, dtype=torch.float32)
        xs, ys, off = [], [], 0
        self.out_idx = None
        for i, l in enumerate(layers):
            xs.append(self.x_hidden if l["K"] == hidden else torch.randn(l["K"], device=device, generator=g, dtype=torch.float32))
            ys.append(self.arena[off:off + l["N"]])
            off += l["N"]
            if l["N"] == hidden:
                self.out_idx = i
        self.ys = ys
        self.seq = OpSequence(list(layers), xs, ys, fuse_shared_input=True) if layers else None
        self.graph = None
        if graph and self.seq is not None and self.out_idx is not None and torch.device(device).type == "cuda":
            self.h_static = torch.zeros(hidden, device=device, dtype=dtype)
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                self._eager(self.h_static)  # warm-up outside the capture
            torch.cuda.current_stream(device).wait_stream(side)
            try:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self.out_static = self._eager(self.h_static)
            except RuntimeError:  # capture not possible here (e.g. an allocator / library that cannot be captured): eager stage
                self.graph = None
                torch.cuda.synchronize(device)

    def _eager(self, h_in: torch.Tensor) -> torch.Tensor:
        self.x_hidden.copy_(h_in)  # fp16 -> fp32 (the x.float() of quant.py:223)
        self.arena.zero_()
        self.seq.launch()
        y = self.ys[self.out_idx]
        y = y * torch.rsqrt(y.pow(2).mean() + 1e-6)
        return y.to(h_in.dtype)

    def __call__(self, h_in: torch.Tensor) -> torch.Tensor:
        if self.seq is None or self.out_idx is None:
            return h_in  # a rank without layers passes the hidden state through
        if self.graph is not None and h_in.dtype == self.h_static.dtype:
            self.h_static.copy_(h_in)
            self.graph.replay()
            return self.out_static
        return self._eager(h_in)
