"""Synthetic operands for the quant_cuda ops at the reference's model shapes.

No checkpoints or datasets are needed: any int32 bit pattern is a valid 3-/4-bit packing
(squeezellm/quant.py:171-208), codebooks are per-channel sorted N(0, 0.02^2) (k-means-like and
distinct per channel, so channel mix-ups are visible), outliers are a Bernoulli mask plus a few
planted heavy rows (the skew that motivated the reference's top-X rows).  Everything is random and
non-symmetric; nothing is zero-filled.
"""
from __future__ import annotations

import torch

# (K, N) of the quantised linears of one decoder layer, in call order, with multiplicity
MODEL_SHAPES = {
    # models/llama-2-7b/config.json: hidden 4096, intermediate 11008, 32 layers
    "llama-7b": dict(layers=32, linears=[("q_proj", 4096, 4096), ("k_proj", 4096, 4096), ("v_proj", 4096, 4096),
                                         ("o_proj", 4096, 4096), ("gate_proj", 4096, 11008),
                                         ("up_proj", 4096, 11008), ("down_proj", 11008, 4096)]),
    # models/llama-2-13b/config.json: hidden 5120, intermediate 13824, 40 layers
    "llama-13b": dict(layers=40, linears=[("q_proj", 5120, 5120), ("k_proj", 5120, 5120), ("v_proj", 5120, 5120),
                                          ("o_proj", 5120, 5120), ("gate_proj", 5120, 13824),
                                          ("up_proj", 5120, 13824), ("down_proj", 13824, 5120)]),
    # LLaMA-65B: hidden 8192, intermediate 22016, 80 layers
    "llama-65b": dict(layers=80, linears=[("q_proj", 8192, 8192), ("k_proj", 8192, 8192), ("v_proj", 8192, 8192),
                                          ("o_proj", 8192, 8192), ("gate_proj", 8192, 22016),
                                          ("up_proj", 8192, 22016), ("down_proj", 22016, 8192)]),
    # models/opt-1.3b/config.json: hidden 2048, ffn 8192, 24 layers, linears have bias
    "opt-1.3b": dict(layers=24, linears=[("q_proj", 2048, 2048), ("k_proj", 2048, 2048), ("v_proj", 2048, 2048),
                                         ("out_proj", 2048, 2048), ("fc1", 2048, 8192), ("fc2", 8192, 2048)]),
}


def algorithmic_bytes(K: int, N: int, bits: int, batch: int = 1, nnz: int = 0, topX: int = 0) -> int:
    """Bytes one op call must move: qweight + LUT + x + mul read/write
    (+ CSR vals/cols/rows) (+ full_rows and its indices)."""
    b = K * N * bits // 8 + N * (1 << bits) * 4 + batch * K * 4 + 2 * batch * N * 4
    if nnz:
        b += 8 * nnz + 4 * (N + 1)
    if topX:
        b += 4 * K * topX + 4 * topX
    return b


def make_layer(K: int, N: int, bits: int, *, sparse_frac: float = 0.0, topX: int = 0, heavy_rows: int = 0,
               heavy_frac: float = 0.2, bias: bool = False, device="cuda", seed: int = 0) -> dict:
    """Operands of one QuantLinearLUT (buffer names as in squeezellm/quant.py:48-95)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rows_q = K // 32 * bits
    qweight = torch.randint(-(2**31), 2**31, (rows_q, N), dtype=torch.int64, device=device, generator=g).to(torch.int32)
    lut = torch.randn((N, 1 << bits), device=device, generator=g, dtype=torch.float32) * 0.02
    lut, _ = torch.sort(lut, dim=1)
    layer = dict(bits=bits, K=K, N=N, qweight=qweight, lookup_table=lut.contiguous(), bias=None,
                 rows=None, cols=None, vals=None, full_rows=None, full_row_indices=None)
    if bias:
        layer["bias"] = torch.randn(N, device=device, generator=g, dtype=torch.float32) * 0.01
    if sparse_frac > 0:
        # built one row block at a time so the N x K Bernoulli mask never exists whole
        counts = torch.zeros(N, dtype=torch.int64, device=device)
        col_chunks = []
        heavy = set(torch.randperm(N, generator=g, device=device)[:heavy_rows].tolist()) if heavy_rows else set()
        blk = max(1, (1 << 24) // K)
        for r0 in range(0, N, blk):
            r1 = min(N, r0 + blk)
            m = torch.rand((r1 - r0, K), device=device, generator=g) < sparse_frac
            for r in heavy:
                if r0 <= r < r1:
                    m[r - r0] = torch.rand(K, device=device, generator=g) < heavy_frac
            counts[r0:r1] = m.sum(dim=1)
            col_chunks.append(m.nonzero()[:, 1].to(torch.int32))  # row-major -> cols sorted per row
        cols = torch.cat(col_chunks) if col_chunks else torch.zeros(0, dtype=torch.int32, device=device)
        rows = torch.zeros(N + 1, dtype=torch.int32, device=device)
        rows[1:] = torch.cumsum(counts, 0).to(torch.int32)
        vals = torch.randn(cols.numel(), device=device, generator=g, dtype=torch.float32) * 0.1
        layer.update(rows=rows, cols=cols.contiguous(), vals=vals)
    if topX > 0:
        layer["full_rows"] = (torch.randn((K, topX), device=device, generator=g, dtype=torch.float32) * 0.02).contiguous()
        layer["full_row_indices"] = torch.randperm(N, generator=g, device=device)[:topX].to(torch.int32).contiguous()
    return layer


def layer_bytes(layer: dict, batch: int = 1) -> int:
    nnz = 0 if layer["vals"] is None else layer["vals"].numel()
    topX = 0 if layer["full_rows"] is None else layer["full_rows"].shape[1]
    return algorithmic_bytes(layer["K"], layer["N"], layer["bits"], batch, nnz, topX)


def make_model(name: str, bits: int, *, sparse_frac: float = 0.0, topX: int = 0, heavy_rows: int = 10,
               n_layers: int | None = None, device="cuda", seed: int = 0) -> list[dict]:
    """All quantised linears of a model, in execution order, each with its own weights (so a pass
    streams distinct bytes from HBM -- 3.3 GB for 7B w4 -- and cannot be served by the 256 MiB
    Infinity Cache)."""
    spec = MODEL_SHAPES[name]
    L = spec["layers"] if n_layers is None else n_layers
    out = []
    s = seed
    for li in range(L):
        for lname, K, N in spec["linears"]:
            lay = make_layer(K, N, bits, sparse_frac=sparse_frac, topX=topX if sparse_frac > 0 else 0,
                             heavy_rows=heavy_rows if sparse_frac > 0 else 0, bias=name.startswith("opt"),
                             device=device, seed=s)
            lay["name"] = f"layers.{li}.{lname}"
            out.append(lay)
            s += 1
    return out
