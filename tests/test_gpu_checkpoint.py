"""SURVEY.md 8(f3) on the GPU: a checkpoint in the reference's format -- the flat state dict quantization/pack.py:173-181
writes (per quantised linear `qweight`, `lookup_table`, `rows`, `cols`, `vals`, plus `sparse_threshold.<name>` = nnz) and
llama.py:157-182 reads -- taken all the way to the kernels, two ways:

  (a) squeezellm_amd.checkpoint.load_layers(path, topX=10): operand dicts with the ten densest outlier rows moved out of the
      CSR into `full_rows` (which the reference's packer never produces), run as one decoder layer of grouped launches;
  (b) the reference's own loader path: the UNMODIFIED squeezellm/quant.py staged by oracle/build_ref.sh, its
      `make_quant_lut(model, layers, wbits, include_sparse=True, numvals=..., topX=10)` + `load_state_dict(strict=False)`
      exactly as llama.py:172-181 does it (so `full_rows` stay all-zero and the hybrid operator adds zeros to mul[0]),
      every QuantLinearLUT.forward on the HIP kernels.

Both against the C oracle fed the ORIGINAL operands (the checkpoint's CSR, no extraction): the result must be the same
linear map whichever way the outliers are split.
"""
import contextlib
import importlib.util
import io
import os
import sys

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

HIDDEN, INTER = 512, 1408
LINEARS = [("self_attn.q_proj", HIDDEN, HIDDEN), ("self_attn.k_proj", HIDDEN, HIDDEN), ("self_attn.v_proj", HIDDEN, HIDDEN),
           ("self_attn.o_proj", HIDDEN, HIDDEN), ("mlp.gate_proj", HIDDEN, INTER), ("mlp.up_proj", HIDDEN, INTER),
           ("mlp.down_proj", INTER, HIDDEN)]
REF_QUANT = os.path.join(H.ROOT, "oracle", "_ref", "reference_py", "quant.py")


def _checkpoint(bits, tmp_path):
    """Synthetic operands of one decoder layer -> a state dict with the reference's keys, saved and re-read."""
    import torch

    from squeezellm_amd import synth

    originals, sd = {}, {}
    for j, (lname, K, N) in enumerate(LINEARS):
        name = f"model.layers.0.{lname}"
        lay = synth.make_layer(K, N, bits, sparse_frac=0.01, heavy_rows=6, heavy_frac=0.3, device="cpu", seed=40 + 10 * bits + j)
        originals[name] = lay
        for f in ("qweight", "lookup_table", "rows", "cols", "vals"):
            sd[f"{name}.{f}"] = lay[f]
        sd[f"sparse_threshold.{name}"] = lay["vals"].numel()  # quantization/pack.py:176-178
    sd["model.norm.weight"] = torch.ones(HIDDEN, dtype=torch.float16)  # an unquantised tensor rides along
    path = tmp_path / f"sq-toy-w{bits}-s45.pt"
    torch.save(sd, path)
    return originals, path


def _np(lay):
    import torch

    return {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in lay.items()}


@pytest.mark.parametrize("bits", [3, 4])
def test_checkpoint_to_kernels_via_load_layers(gpu, bits, tmp_path):
    import torch

    from squeezellm_amd import checkpoint, decode

    originals, path = _checkpoint(bits, tmp_path)
    layers = checkpoint.load_layers(str(path), topX=10, device=gpu)
    assert list(layers) == list(originals)  # execution order q, k, v, o, gate, up, down
    lays = list(layers.values())
    for lay, orig in zip(lays, originals.values()):
        assert lay["full_rows"] is not None and lay["full_rows"].shape == (lay["K"], 10)
        assert lay["vals"].numel() < orig["vals"].numel()  # the densest rows left the CSR
    g = torch.Generator(device=gpu).manual_seed(8)
    h = torch.randn(HIDDEN, device=gpu, generator=g, dtype=torch.float16).float()
    m = torch.randn(HIDDEN, device=gpu, generator=g, dtype=torch.float16).float()
    d = torch.randn(INTER, device=gpu, generator=g, dtype=torch.float16).float()
    xs = [h, h, h, m, m, m, d]
    xs[3] = torch.randn(HIDDEN, device=gpu, generator=g, dtype=torch.float16).float()  # o_proj has its own input
    ys = [torch.zeros(l["N"], device=gpu) for l in lays]
    seq = decode.OpSequence(lays, xs, ys, fuse_shared_input=True)
    assert seq.groups == [[0, 1, 2], [3], [4, 5], [6]]
    seq.launch()
    torch.cuda.synchronize()
    lib = H.c_oracle()
    for (name, orig), x, y in zip(originals.items(), xs, ys):
        ref = H.c_matvec(lib, _np(orig), x.cpu().numpy(), np.zeros(orig["N"], np.float32), batched=False)
        err = H.rel_err(y.cpu().numpy(), ref)
        assert err <= 2e-5, f"{name}: {err:.2e}"


@pytest.fixture(scope="module")
def refquant():
    if not os.path.exists(REF_QUANT):
        pytest.skip("oracle/_ref/reference_py/quant.py not staged (oracle/build_ref.sh needs /root/reference)")
    if H.ROOT not in sys.path:
        sys.path.insert(0, H.ROOT)
    import quant_cuda  # noqa: F401  the root shim -> squeezellm_amd.quant_cuda

    spec = importlib.util.spec_from_file_location("reference_squeezellm_quant_ckpt", REF_QUANT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("bits", [3, 4])
def test_checkpoint_to_kernels_via_the_reference_loader(gpu, refquant, bits, tmp_path):
    """llama.py:157-182 line for line on a toy decoder layer: thresholds out of the state dict, make_quant_lut with
    include_sparse / numvals / topX, load_state_dict(strict=False); then the reference's forward on every linear."""
    import torch
    from torch import nn

    originals, path = _checkpoint(bits, tmp_path)

    class Attn(nn.Module):
        def __init__(self):
            super().__init__()
            for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
                setattr(self, n, nn.Linear(HIDDEN, HIDDEN, bias=False))

    class Mlp(nn.Module):
        def __init__(self):
            super().__init__()
            self.gate_proj, self.up_proj = nn.Linear(HIDDEN, INTER, bias=False), nn.Linear(HIDDEN, INTER, bias=False)
            self.down_proj = nn.Linear(INTER, HIDDEN, bias=False)

    class Layer(nn.Module):
        def __init__(self):
            super().__init__()
            self.self_attn, self.mlp = Attn(), Mlp()

    class Model(nn.Module):
        def __init__(self):
            super().__init__()
            self.layers = nn.ModuleList([Layer()])
            self.norm = nn.LayerNorm(HIDDEN, bias=False)

    class Top(nn.Module):
        def __init__(self):
            super().__init__()
            self.model = Model()

    model = Top().half().eval()
    names = {n for n, m in model.named_modules() if isinstance(m, nn.Linear)}  # find_layers (llama.py:161)
    state_dict = torch.load(path)
    num_vals = {k.replace("sparse_threshold.", ""): v for k, v in state_dict.items() if "sparse_threshold." in k}
    for k in num_vals:
        del state_dict["sparse_threshold." + k]
    with contextlib.redirect_stdout(io.StringIO()):
        refquant.make_quant_lut(model, names, bits, include_sparse=True, numvals=num_vals, topX=10)
    missing = model.load_state_dict(state_dict, strict=False)
    assert all(k.endswith(("full_rows", "full_row_indices")) for k in missing.missing_keys), missing.missing_keys
    model = model.to(gpu)
    g = torch.Generator(device=gpu).manual_seed(9)
    for name, orig in originals.items():
        mod = model.get_submodule(name)
        assert type(mod).__name__ == "QuantLinearLUT" and mod.topX == 10 and not bool(mod.full_rows.any())
        x = torch.randn(1, 1, orig["K"], device=gpu, generator=g, dtype=torch.float16)
        with torch.no_grad():
            y = mod(x)
        torch.cuda.synchronize()
        assert y.dtype == torch.float16 and tuple(y.shape) == (1, 1, orig["N"])
        ref = H.oracle.quantlinear_forward(x.cpu().numpy(), _np(orig))
        err = H.rel_err(y.float().cpu().numpy(), ref)
        assert err <= 1e-3, f"{name}: {err:.2e}"  # fp16 output rounding (BASELINE.json north_star tolerance)
