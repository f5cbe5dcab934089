"""The dependency-gated pass (sqllm_pass_*, squeezellm_amd/csrc/experimental/sqllm_pass.hip): consecutive groups of a
decode pass as ONE persistent launch, each group gated on the completion of the one before it where it first reads
vec.  MEASUREMENT LIBRARY: built, measured 2.3-4x slower than one launch per group (profiles/r04_pass_*.txt) and not
adopted; these tests keep it parity-green so that the measurement stays reproducible.

The tests chain the groups for real: o_proj's input IS q_proj's output buffer, gate/up read o_proj's output,
down_proj reads gate_proj's, the next layer's q/k/v read down_proj's -- so an op that consumed its vec before the
producing group was complete (or from a stale cache line) computes from a partial vector and fails the per-op
comparison with the C oracle, which is fed the FINAL contents of that vector.  Reference launch structure this
replaces: one to three dependent launches per op, squeezellm/quant_cuda_kernel.cu:157-179, :510-577; arithmetic
:741-880, :1040-1164.
"""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

TOL_FP64 = 2e-5  # fp32 accumulation in unspecified (atomic) order vs the fp64 oracle, max-norm relative


def _np_layer(lay):
    import torch

    return {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in lay.items()}


def _chain(linears, n_layers, bits, sparse, topX, gpu, seed0, scale=None):
    """`n_layers` decoder layers of the given (name, K, N) linears with CHAINED activations:
    q/k/v <- hidden, o_proj <- y(q_proj), gate/up <- y(o_proj), down_proj <- y(gate_proj), next hidden = y(down_proj)."""
    import torch

    from squeezellm_amd import synth

    g = torch.Generator(device=gpu)
    g.manual_seed(seed0)
    layers, xs, ys = [], [], []
    hidden = torch.randn(linears[0][1], device=gpu, generator=g, dtype=torch.float16).float()
    for li in range(n_layers):
        out = {}
        for j, (name, K, N) in enumerate(linears):
            lay = synth.make_layer(K, N, bits, sparse_frac=sparse, topX=topX, heavy_rows=4 if sparse else 0, device=gpu,
                                   seed=seed0 + 17 * li + j)
            if scale is not None:  # keep the chain's magnitude flat over many layers
                lay["lookup_table"] = (lay["lookup_table"] * scale(K)).contiguous()
                if lay["vals"] is not None:
                    lay["vals"] = (lay["vals"] * scale(K)).contiguous()
                if lay["full_rows"] is not None:
                    lay["full_rows"] = (lay["full_rows"] * scale(K)).contiguous()
            lay["name"] = f"layers.{li}.{name}"
            x = {"q_proj": hidden, "k_proj": hidden, "v_proj": hidden, "o_proj": out.get("q_proj"),
                 "gate_proj": out.get("o_proj"), "up_proj": out.get("o_proj"), "down_proj": out.get("gate_proj")}[name]
            y = torch.randn(N, device=gpu, generator=g) * 0.01  # accumulate semantics: mul starts non-zero
            out[name] = y
            layers.append(lay)
            xs.append(x)
            ys.append(y)
        hidden = out["down_proj"]
    return layers, xs, ys


def _check_ops(layers, xs, ys0, ys, what):
    """every op against the C oracle fed the FINAL contents of its input vector"""
    lib = H.c_oracle()
    x_final = {id(x): x.cpu().numpy() for x in xs}
    for l, x, y0, y in zip(layers, xs, ys0, ys):
        ref = H.c_matvec(lib, _np_layer(l), x_final[id(x)], y0, batched=False)
        err = H.rel_err(y.cpu().numpy(), ref)
        assert err <= TOL_FP64, f"{what}: {l['name']} {l['K']}x{l['N']} w{l['bits']}: rel err {err:.2e}"


SMALL = [("q_proj", 1024, 1024), ("k_proj", 1024, 1024), ("v_proj", 1024, 1024), ("o_proj", 1024, 1024),
         ("gate_proj", 1024, 2816), ("up_proj", 1024, 2816), ("down_proj", 2816, 1024)]
CONFIGS = [(4, 0.0, 0), (4, 0.0045, 10), (3, 0.0045, 10), (3, 0.0, 0)]
IDS = ["w4-s0", "w4-s45", "w3-s45", "w3-s0"]


def _flat(K):
    return 1.0 / (0.02 * np.sqrt(K))


@pytest.mark.parametrize("bits,sparse,topX", CONFIGS, ids=IDS)
def test_chained_pass_small_shapes(gpu, bits, sparse, topX):
    """12 chained decoder layers of small linears: 48 dependent groups, far more work items than resident
    workgroups, every workgroup walks several items and most items wait at a gate."""
    import torch

    from squeezellm_amd import decode, experimental

    layers, xs, ys = _chain(SMALL, 12, bits, sparse, topX, gpu, seed0=500 + bits, scale=_flat)
    ys0 = [y.cpu().numpy().copy() for y in ys]
    seq = decode.OpSequence(layers, xs, ys, fuse_shared_input=True)
    assert seq.groups[:4] == [[0, 1, 2], [3], [4, 5], [6]]
    p = experimental.GatedPass(seq)
    assert p.n_items > p.grid > 0
    p.launch()
    assert p.status() == (0, 0)
    _check_ops(layers, xs, ys0, ys, "eager")


@pytest.mark.parametrize("bits,sparse,topX", CONFIGS[:3], ids=IDS[:3])
def test_chained_pass_llama7b_graph_replay(gpu, bits, sparse, topX):
    """BASELINE configs[1] / [2] shapes, three chained decoder layers, the pass captured in a HIP graph together with
    the re-initialisation of its outputs and replayed several times: every replay must reproduce the oracle (the
    arrival counters are re-zeroed by the launch's own memset node, consumers are L1-warm from the replay before)."""
    import torch

    from squeezellm_amd import decode, experimental, synth

    layers, xs, ys = _chain(synth.MODEL_SHAPES["llama-7b"]["linears"], 3, bits, sparse, topX, gpu, seed0=700 + bits, scale=_flat)
    ys0_t = [y.clone() for y in ys]
    ys0 = [y.cpu().numpy().copy() for y in ys]
    seq = decode.OpSequence(layers, xs, ys, fuse_shared_input=True)
    p = experimental.GatedPass(seq)

    def step():
        torch._foreach_copy_(ys, ys0_t)
        p.launch()

    side = torch.cuda.Stream(gpu)
    side.wait_stream(torch.cuda.current_stream(gpu))
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream(gpu).wait_stream(side)
    assert p.status() == (0, 0)
    _check_ops(layers, xs, ys0, ys, "warm-up")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for r in range(4):
        g.replay()
        assert p.status() == (0, 0)
        _check_ops(layers, xs, ys0, ys, f"replay {r}")


def test_pass_equals_grouped_launches(gpu):
    """Same semantics as sqllm_launch_groups: the chained pass and the one-launch-per-group sequence produce the same
    vectors (to summation order) -- including the accumulate into a non-zero mul and the in-order visibility of vec."""
    import torch

    from squeezellm_amd import decode, experimental, synth

    layers, xs, ys = _chain(synth.MODEL_SHAPES["llama-7b"]["linears"], 2, 4, 0.0045, 10, gpu, seed0=900, scale=_flat)
    ys0_t = [y.clone() for y in ys]
    seq = decode.OpSequence(layers, xs, ys, fuse_shared_input=True)
    seq.launch()
    torch.cuda.synchronize()
    want = [y.clone() for y in ys]
    torch._foreach_copy_(ys, ys0_t)
    p = experimental.GatedPass(seq)
    p.launch()
    assert p.status() == (0, 0)
    for l, a, b in zip(layers, ys, want):
        err = H.rel_err(a.cpu().numpy(), b.cpu().numpy())
        assert err <= 2 * TOL_FP64, f"{l['name']}: pass vs grouped launches {err:.2e}"


def test_pass_under_uneven_load(gpu):
    """Hand-offs must hold when the producers are slowed down unevenly: a second stream keeps a bandwidth-hungry
    kernel running beside the pass (it takes memory bandwidth and CU slots away from SOME workgroups), many launches,
    every word of every output checked."""
    import torch

    from squeezellm_amd import decode, experimental

    layers, xs, ys = _chain(SMALL, 6, 4, 0.0045, 10, gpu, seed0=1100, scale=_flat)
    ys0_t = [y.clone() for y in ys]
    ys0 = [y.cpu().numpy().copy() for y in ys]
    seq = decode.OpSequence(layers, xs, ys, fuse_shared_input=True)
    p = experimental.GatedPass(seq)
    noise = torch.empty(64 << 20, device=gpu)
    side = torch.cuda.Stream(gpu)
    for r in range(6):
        torch._foreach_copy_(ys, ys0_t)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(r):  # 0..5 competing kernels in flight
                noise.add_(1.0)
        p.launch()
        assert p.status() == (0, 0)
        torch.cuda.synchronize()
        _check_ops(layers, xs, ys0, ys, f"round {r}")


def test_pass_rejects_what_it_cannot_run(gpu):
    import torch

    from squeezellm_amd import decode, experimental, synth

    lay = synth.make_layer(256, 128, 4, device=gpu, seed=1)
    x = torch.randn(2, 256, device=gpu)
    y = torch.zeros(2, 128, device=gpu)
    seq = decode.OpSequence([lay], [x], [y], batched=True)
    with pytest.raises(ValueError):
        experimental.GatedPass(seq)  # batched ops are not part of a gated pass
    lay3 = synth.make_layer(256, 128, 3, device=gpu, seed=2)
    seq = decode.OpSequence([lay, lay3], [x[0], x[1]], [y[0], y[1]])
    with pytest.raises(ValueError):
        experimental.GatedPass(seq)  # one bit width per pass
