"""CPU tests of squeezellm_amd.checkpoint against the reference's checkpoint conventions
(quantization/pack.py:173-181 writes, llama.py:157-182 reads), using tensors produced by the
UNMODIFIED reference packer (tests/golden/pack2_*.npz)."""
import os

import numpy as np
import pytest
import torch

from squeezellm_amd import checkpoint, quant
from tests import helpers as H


def _golden(name):
    g = np.load(os.path.join(H.GOLDEN, name))
    return {k: g[k] for k in g.files}


def _fake_checkpoint():
    """A two-decoder-layer 'model' whose q_proj / down_proj tensors are the reference packer's output."""
    w4 = _golden("pack2_w4_sparse.npz")
    sd = {"model.embed_tokens.weight": torch.zeros(4, 4), "model.norm.weight": torch.ones(4)}
    names = ["model.layers.1.mlp.down_proj", "model.layers.0.self_attn.q_proj", "model.layers.1.self_attn.q_proj",
             "model.layers.0.mlp.down_proj"]  # deliberately out of order
    for n in names:
        for f in ("qweight", "lookup_table", "rows", "cols", "vals"):
            sd[f"{n}.{f}"] = torch.from_numpy(w4[f].copy())
        sd[f"sparse_threshold.{n}"] = int(w4["vals"].size)
    return sd, w4, names


def test_load_layers_orders_infers_and_validates(tmp_path):
    sd, w4, names = _fake_checkpoint()
    path = tmp_path / "sq-fake-w4-s45.pt"
    torch.save(sd, path)
    layers = checkpoint.load_layers(str(path))
    assert list(layers) == ["model.layers.0.self_attn.q_proj", "model.layers.0.mlp.down_proj",
                            "model.layers.1.self_attn.q_proj", "model.layers.1.mlp.down_proj"]
    lay = layers["model.layers.0.self_attn.q_proj"]
    assert (lay["bits"], lay["K"], lay["N"]) == (4, int(w4["K"]), int(w4["N"]))
    assert np.array_equal(lay["qweight"].numpy(), w4["qweight"]) and np.array_equal(lay["vals"].numpy(), w4["vals"])
    assert lay["full_rows"] is None and lay["bias"] is None
    bad = dict(sd)
    bad["sparse_threshold.model.layers.0.self_attn.q_proj"] = 1
    with pytest.raises(ValueError, match="sparse_threshold"):
        checkpoint.load_layers(bad)
    bad = dict(sd)
    bad["model.layers.0.self_attn.q_proj.rows"] = sd["model.layers.0.self_attn.q_proj.rows"][:-1]
    with pytest.raises(ValueError, match="inconsistent CSR"):
        checkpoint.load_layers(bad)


def test_topx_regeneration_keeps_the_result_and_round_trips():
    sd, w4, _ = _fake_checkpoint()
    name = "model.layers.0.self_attn.q_proj"
    plain = checkpoint.layer_operands(sd, name)
    hyb = checkpoint.layer_operands(sd, name, topX=4)
    assert hyb["full_rows"].shape == (plain["K"], 4) and hyb["vals"].numel() < plain["vals"].numel()
    rng = np.random.default_rng(0)
    x = rng.normal(size=plain["K"]).astype(np.float32)
    mul = np.zeros(plain["N"], np.float32)
    npy = lambda d: {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}
    a = H.oracle_ref(npy(plain), x, mul, "spmv")
    b = H.oracle_ref(npy(hyb), x, mul, "hybrid")
    assert np.allclose(a, b, rtol=0, atol=1e-6)
    # writing the hybrid layer back gives the reference-format tensors we started from
    back = checkpoint.to_state_dict({name: hyb})
    assert back[f"sparse_threshold.{name}"] == plain["vals"].numel()
    for f in ("rows", "cols"):
        assert torch.equal(back[f"{name}.{f}"], plain[f])
    assert torch.allclose(back[f"{name}.vals"], plain["vals"], rtol=0, atol=0)
    assert f"{name}.full_rows" not in back


def test_checkpoint_loads_into_the_module_tree_like_llama_py():
    """make_quant_lut + load_state_dict(strict=False), the reference's loading sequence (llama.py:160-180)."""
    sd, w4, _ = _fake_checkpoint()
    K, N = int(w4["K"]), int(w4["N"])

    class Attn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj = torch.nn.Linear(K, N, bias=False)

    class Mlp(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.down_proj = torch.nn.Linear(K, N, bias=False)

    class Layer(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.self_attn, self.mlp = Attn(), Mlp()

    class Inner(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.layers = torch.nn.ModuleList([Layer(), Layer()])

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.model = Inner()

    model = Model()
    names = checkpoint.quantized_names(sd)
    numvals = {k.replace("sparse_threshold.", ""): v for k, v in sd.items() if k.startswith("sparse_threshold.")}
    tensors = {k: v for k, v in sd.items() if not k.startswith("sparse_threshold.")}
    quant.make_quant_lut(model, names, 4, include_sparse=True, numvals=numvals, topX=10)
    res = model.load_state_dict(tensors, strict=False)
    assert all("full_row" in k for k in res.missing_keys) and len(res.missing_keys) == 8
    q = model.model.layers[0].self_attn.q_proj
    assert isinstance(q, quant.QuantLinearLUT) and np.array_equal(q.vals.numpy(), w4["vals"])
    assert q.op_kind(False) == "spmv_hybrid" and not bool(q.full_rows.any())
