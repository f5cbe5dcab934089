"""Reading and writing SqueezeLLM checkpoints (`sq-*.pt`).

Format (written by /root/reference/quantization/pack.py:173-181, read by
/root/reference/llama.py:157-182): one flat `torch.save`d state dict holding, for every quantised
linear `<name>` (e.g. `model.layers.0.self_attn.q_proj`),
    <name>.qweight int32 [K/32*bits, N]     <name>.lookup_table fp32 [N, 2**bits]
    <name>.bias fp32 [N] (models with bias)
    <name>.rows / .cols / .vals             CSR outliers (dense-and-sparse checkpoints)
    <name>.startrows                        (packed with --balance)
    sparse_threshold.<name> = nnz           so the loader can size cols / vals before loading
next to the unquantised tensors (embeddings, norms, lm_head).  `full_rows` / `full_row_indices`
are never in a checkpoint: pack2 does not produce them (quant.py:97-208) and llama.py loads with
strict=False, so the reference's hybrid operator runs with all-zero full rows.  `topX > 0` here
builds the real thing: the topX densest outlier rows move out of the CSR into `full_rows`
(squeezellm_amd.pack.extract_topx_rows), which leaves the layer's result unchanged.

Host-side tooling around the hot path; tensors only, no model code.
"""
from __future__ import annotations

import re
from collections import OrderedDict

import torch

from . import pack

_ORDER = {"q_proj": 0, "k_proj": 1, "v_proj": 2, "o_proj": 3, "out_proj": 3, "gate_proj": 4, "fc1": 4,
          "up_proj": 5, "down_proj": 6, "fc2": 6}
_FIELDS = ("qweight", "lookup_table", "bias", "rows", "cols", "vals", "startrows", "full_rows", "full_row_indices")


def _exec_key(name: str):
    m = re.search(r"layers\.(\d+)\.", name)
    return (int(m.group(1)) if m else 1 << 30, _ORDER.get(name.rsplit(".", 1)[-1], 99), name)


def quantized_names(state_dict) -> list[str]:
    """Names of the quantised linears in a checkpoint, in execution order (q, k, v, o, gate, up,
    down inside each decoder layer: squeezellm/model_parse.py:53-61)."""
    names = [k[: -len(".qweight")] for k in state_dict if k.endswith(".qweight")]
    return sorted(names, key=_exec_key)


def layer_operands(state_dict, name: str, topX: int = 0, device=None) -> dict:
    """The operand dict (as squeezellm_amd.decode / quant.QuantLinearLUT.from_operands take it) of
    one quantised linear of a checkpoint."""
    t = {f: state_dict.get(f"{name}.{f}") for f in _FIELDS}
    if t["qweight"] is None or t["lookup_table"] is None:
        raise KeyError(f"{name}: qweight / lookup_table missing from the checkpoint")
    entries = t["lookup_table"].shape[1]
    if entries not in (8, 16):
        raise ValueError(f"{name}: lookup_table has {entries} entries per channel, expected 8 or 16")
    bits = 3 if entries == 8 else 4
    rows_q, N = t["qweight"].shape
    if rows_q % bits or t["lookup_table"].shape[0] != N:
        raise ValueError(f"{name}: qweight {tuple(t['qweight'].shape)} does not match {bits}-bit / N={N}")
    K = rows_q // bits * 32
    nnz_key = f"sparse_threshold.{name}"
    if t["vals"] is not None and nnz_key in state_dict and int(state_dict[nnz_key]) != t["vals"].numel():
        raise ValueError(f"{name}: sparse_threshold says {int(state_dict[nnz_key])} outliers, vals has {t['vals'].numel()}")
    mv = (lambda x: x) if device is None else (lambda x: None if x is None else x.to(device))
    lay = dict(name=name, bits=bits, K=K, N=N, qweight=mv(t["qweight"].to(torch.int32)).contiguous(),
               lookup_table=mv(t["lookup_table"].to(torch.float32)).contiguous(),
               bias=None if t["bias"] is None else mv(t["bias"].to(torch.float32)),
               rows=None, cols=None, vals=None, full_rows=None, full_row_indices=None)
    if t["vals"] is not None and t["vals"].numel() > 0:
        rows, cols, vals = mv(t["rows"].to(torch.int32)), mv(t["cols"].to(torch.int32)), mv(t["vals"].to(torch.float32))
        if rows.numel() != N + 1 or int(rows[-1]) != vals.numel() or cols.numel() != vals.numel():
            raise ValueError(f"{name}: inconsistent CSR (rows[{rows.numel()}], last {int(rows[-1])}, nnz {vals.numel()})")
        fr = fi = None
        stored = t["full_rows"]
        if stored is not None and bool((stored != 0).any()):  # a checkpoint that does carry real full rows
            fr, fi = mv(stored.to(torch.float32)).contiguous(), mv(t["full_row_indices"].to(torch.int32)).contiguous()
        elif topX > 0:
            rows, cols, vals, fr, fi = pack.extract_topx_rows(rows, cols, vals, K, topX)
        lay.update(rows=rows.contiguous(), cols=cols.contiguous(), vals=vals.contiguous(), full_rows=fr, full_row_indices=fi)
    return lay


def load_layers(checkpoint, topX: int = 0, device=None) -> "OrderedDict[str, dict]":
    """Every quantised linear of a checkpoint (a path or an already loaded state dict), in
    execution order, as operand dicts.  Unquantised tensors are ignored."""
    sd = torch.load(checkpoint, map_location="cpu") if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__") else checkpoint
    return OrderedDict((n, layer_operands(sd, n, topX=topX, device=device)) for n in quantized_names(sd))


def to_state_dict(layers: dict, extra: dict | None = None) -> dict:
    """The inverse: operand dicts by name -> a flat state dict in the reference's checkpoint format
    (loadable by the reference's llama.py).  `full_rows`, which the reference's loader would have no
    matching CSR for, are folded back into the CSR so the checkpoint stays self-consistent for it."""
    sd = dict(extra or {})
    for name, lay in layers.items():
        sd[f"{name}.qweight"] = lay["qweight"].cpu()
        sd[f"{name}.lookup_table"] = lay["lookup_table"].cpu()
        if lay.get("bias") is not None:
            sd[f"{name}.bias"] = lay["bias"].cpu()
        if lay.get("vals") is not None:
            rows, cols, vals = lay["rows"].cpu(), lay["cols"].cpu(), lay["vals"].cpu()
            if lay.get("full_rows") is not None:
                rows, cols, vals = _fold_full_rows(rows, cols, vals, lay["full_rows"].cpu(), lay["full_row_indices"].cpu())
            sd[f"{name}.rows"], sd[f"{name}.cols"], sd[f"{name}.vals"] = rows, cols, vals
            sd[f"sparse_threshold.{name}"] = vals.numel()
    return sd


def _fold_full_rows(rows, cols, vals, full_rows, full_idx):
    N = rows.numel() - 1
    K = full_rows.shape[0]
    counts = (rows[1:] - rows[:-1]).to(torch.int64)
    rid = torch.repeat_interleave(torch.arange(N), counts)
    nz = (full_rows != 0).nonzero()  # (k, slot), slot-major after the sort below
    r_new = full_idx.to(torch.int64)[nz[:, 1]]
    all_r = torch.cat([rid, r_new])
    all_c = torch.cat([cols.to(torch.int64), nz[:, 0]])
    all_v = torch.cat([vals, full_rows[nz[:, 0], nz[:, 1]]])
    key = all_r * K + all_c
    order = torch.argsort(key, stable=True)
    key, all_v = key[order], all_v[order]
    uniq, inv = torch.unique_consecutive(key, return_inverse=True)  # an entry present in both: summed
    v = torch.zeros(uniq.numel(), dtype=all_v.dtype).index_add_(0, inv, all_v)
    r = uniq // K
    new_rows = torch.zeros(N + 1, dtype=torch.int32)
    new_rows[1:] = torch.bincount(r, minlength=N).cumsum(0).to(torch.int32)
    return new_rows, (uniq % K).to(torch.int32), v
