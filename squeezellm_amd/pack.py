"""Operand packer for QuantLinearLUT (torch, CPU or GPU tensors; vectorised).

The reference packs with `QuantLinearLUT.pack2` (/root/reference/squeezellm/quant.py:97-208): a
Python loop over qweight rows on the CPU, and it never produces the `full_rows` /
`full_row_indices` operands that the hybrid operator it dispatches to consumes (SURVEY.md 8(f)3:
checkpoints packed by the reference run the hybrid op with all-zero full rows).  This module
produces the same buffers bit-for-bit (checked against pack2's output in tests/test_pack_cpu.py)
and adds the missing top-X extraction.  It is host-side tooling around the hot path, not part of it.
"""
from __future__ import annotations

import torch


def pack_qweight(idx_kn: torch.Tensor, bits: int) -> torch.Tensor:
    """Integer indices [K, N] (values < 2**bits) -> int32 qweight [K // 32 * bits, N].

    4-bit (quant.py:180-184): row r, bits [4j, 4j+4) = index of k = 8r + j.
    3-bit (quant.py:185-203): each group of 3 rows is one little-endian 96-bit stream in which
    weight k of the group occupies bits [3k, 3k+3)."""
    if bits not in (3, 4):
        raise NotImplementedError("Only 3 and 4 bits is supported.")
    K, N = idx_kn.shape
    if K % 32:
        raise ValueError("K must be a multiple of 32")
    idx = idx_kn.to(torch.int64)
    if bits == 4:
        g = idx.reshape(K // 8, 8, N)
        shifts = (4 * torch.arange(8, device=idx.device, dtype=torch.int64)).view(1, 8, 1)
        word = (g << shifts).sum(dim=1)  # fields do not overlap: sum == or
    else:
        g = idx.reshape(K // 32, 32, N)
        shifts = (3 * torch.arange(32, device=idx.device, dtype=torch.int64)).view(1, 32, 1)
        lo = (g[:, :21] << shifts[:, :21]).sum(dim=1)               # bits 0..62 of the stream
        hi = (g[:, 21:] << (shifts[:, 21:] - 63)).sum(dim=1)        # bits 63..95, re-based at 63
        m32 = (1 << 32) - 1
        r0 = lo & m32
        r1 = ((lo >> 32) & 0x7FFFFFFF) | ((hi & 1) << 31)
        r2 = (hi >> 1) & m32
        word = torch.stack([r0, r1, r2], dim=1).reshape(K // 32 * 3, N)
    word = torch.where(word >= (1 << 31), word - (1 << 32), word)  # reinterpret as int32
    return word.to(torch.int32).contiguous()


def unpack_qweight(qweight: torch.Tensor, bits: int) -> torch.Tensor:
    """int32 qweight -> uint8 indices [K, N] (inverse of pack_qweight)."""
    q = qweight.to(torch.int64) & 0xFFFFFFFF
    R, N = q.shape
    if bits == 4:
        shifts = (4 * torch.arange(8, device=q.device, dtype=torch.int64)).view(1, 8, 1)
        return ((q.unsqueeze(1) >> shifts) & 0xF).reshape(R * 8, N).to(torch.uint8)
    if bits != 3 or R % 3:
        raise ValueError("3-bit qweight needs a multiple of 3 rows")
    g = q.reshape(R // 3, 3, N)
    lo = g[:, 0] | ((g[:, 1] & 0x7FFFFFFF) << 32)        # stream bits 0..62
    hi = (g[:, 1] >> 31) | (g[:, 2] << 1)                # stream bits 63..95, re-based
    k = torch.arange(32, device=q.device, dtype=torch.int64).view(1, 32, 1)
    out_lo = (lo.unsqueeze(1) >> (3 * k[:, :21])) & 7
    out_hi = (hi.unsqueeze(1) >> (3 * k[:, 21:] - 63)) & 7
    return torch.cat([out_lo, out_hi], dim=1).reshape(R // 3 * 32, N).to(torch.uint8)


def outliers_to_csr(outliers_nk: torch.Tensor, lookup_table: torch.Tensor):
    """Dense outlier matrix [N, K] (0 = no outlier) -> (rows int32 [N+1], cols int32, vals fp32) as
    pack2 stores them: every outlier minus the centroid nearest zero of its channel (the dense part
    holds that centroid's index at outlier positions, quant.py:117-123), then CSR over output
    channels (quant.py:126-131).  Entries that become exactly 0 are dropped, as to_sparse_csr does."""
    out = outliers_nk.to(torch.float32)
    lut = lookup_table.to(torch.float32)
    zero_map = lut.gather(1, lut.abs().argmin(dim=1, keepdim=True)).squeeze(1)  # ties -> lowest index
    shifted = torch.where(out != 0, out - zero_map[:, None], out)
    keep = shifted != 0
    rows = torch.zeros(out.shape[0] + 1, dtype=torch.int32, device=out.device)
    rows[1:] = keep.sum(dim=1).cumsum(0).to(torch.int32)
    nz = keep.nonzero()
    return rows, nz[:, 1].to(torch.int32).contiguous(), shifted[keep].contiguous()


def extract_topx_rows(rows: torch.Tensor, cols: torch.Tensor, vals: torch.Tensor, K: int, topX: int):
    """Move the topX output channels with the most outliers out of the CSR into the dense
    `full_rows` [K, topX] / `full_row_indices` [topX] operands of the hybrid op
    (quant_cuda_kernel.cu:1101-1121 adds full_rows[:, c] . vec to mul[full_row_indices[c]]).
    Returns (rows', cols', vals', full_rows, full_row_indices); the op's result is unchanged."""
    N = rows.numel() - 1
    counts = (rows[1:] - rows[:-1]).to(torch.int64)
    topX = min(int(topX), N)
    if topX <= 0:
        return rows, cols, vals, None, None
    top = torch.topk(counts, topX).indices.sort().values
    rid = torch.repeat_interleave(torch.arange(N, device=rows.device), counts)
    is_top = torch.zeros(N, dtype=torch.bool, device=rows.device)
    is_top[top] = True
    moved = is_top[rid]
    slot = torch.full((N,), -1, dtype=torch.int64, device=rows.device)
    slot[top] = torch.arange(topX, device=rows.device)
    full_rows = torch.zeros((K, topX), dtype=torch.float32, device=rows.device)
    full_rows.index_put_((cols[moved].to(torch.int64), slot[rid[moved]]), vals[moved], accumulate=True)
    new_counts = torch.where(is_top, torch.zeros_like(counts), counts)
    new_rows = torch.zeros(N + 1, dtype=torch.int32, device=rows.device)
    new_rows[1:] = new_counts.cumsum(0).to(torch.int32)
    return new_rows, cols[~moved].contiguous(), vals[~moved].contiguous(), full_rows, top.to(torch.int32)


def pack_layer(idx_nk: torch.Tensor, lookup_table: torch.Tensor, bits: int, outliers_nk: torch.Tensor | None = None,
               topX: int = 0, bias: torch.Tensor | None = None) -> dict:
    """All operands of one QuantLinearLUT from per-channel indices [N, K], codebooks [N, 2**bits]
    and an optional dense outlier matrix [N, K] -- the dict squeezellm_amd.decode / quant consume."""
    N, K = idx_nk.shape
    layer = dict(bits=bits, K=K, N=N, qweight=pack_qweight(idx_nk.t().contiguous(), bits),
                 lookup_table=lookup_table.to(torch.float32).contiguous(), bias=bias,
                 rows=None, cols=None, vals=None, full_rows=None, full_row_indices=None)
    if outliers_nk is not None:
        rows, cols, vals = outliers_to_csr(outliers_nk, lookup_table)
        fr = fi = None
        if topX > 0:
            rows, cols, vals, fr, fi = extract_topx_rows(rows, cols, vals, K, topX)
        layer.update(rows=rows, cols=cols, vals=vals, full_rows=fr, full_row_indices=fi)
    return layer
