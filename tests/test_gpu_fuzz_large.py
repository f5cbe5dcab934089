"""Randomised GPU parity at LARGE shapes: launches whose plans leave the small-shape property test's reach -- more workgroups
than the chip holds at once, hundreds of sparse workgroups in front of the grid (the role priorities and the 2048-non-zero CSR
chunks of sqllm_capi.hip: set_role_priority / widen_csr_chunks switch on there), K slices of many steps, groups of 1-4 ops sharing
their input as ONE launch.  Every op of every draw is checked against the C oracle (oracle/sqllm_oracle.c, the restatement of
squeezellm/quant_cuda_kernel.cu:741-1164), through `OpSequence(fuse_shared_input=True)` -- the entry bench.py times -- with a
caller workspace or without one.

The suite runs a fixed, seeded set of draws; SQLLM_FUZZ_LARGE=<n> (tools/sessions/r06.sh fuzz_large) runs n fresh ones.
"""
import os

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

TOL_FP64 = 2e-5  # fp32 accumulation in unspecified (atomic) order vs the fp64 oracle, max-norm relative
_N_FRESH = int(os.environ.get("SQLLM_FUZZ_LARGE", "0"))
_SEEDS = list(np.random.SeedSequence().generate_state(_N_FRESH)) if _N_FRESH else list(range(24))


def _np_layer(lay):
    import torch

    return {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in lay.items()}


def _draw(seed):
    rng = np.random.default_rng(int(seed))
    bits = int(rng.choice([3, 4]))
    K = 32 * int(rng.integers(32, 449))  # 1024 ... 14336
    n_ops = int(rng.choice([1, 1, 2, 3, 4]))
    budget = 1.4e8 / K  # columns of the whole launch: the C oracle stays within a second or two
    Ns = [4 * int(rng.integers(128, max(129, int(budget / n_ops) // 4))) for _ in range(n_ops)]
    batch = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 5, 6, 7, 8, 12, 16]))
    sparse = float(rng.choice([0.0, 0.0045, 0.0045, 0.0045, 0.012]))
    topX = int(rng.choice([0, 10, 10])) if sparse > 0 else 0
    heavy = int(rng.choice([0, 3, 10])) if sparse > 0 else 0
    return dict(bits=bits, K=K, Ns=Ns, batch=batch, sparse=sparse, topX=topX, heavy=heavy, workspace=bool(rng.integers(0, 2)),
                graph=bool(rng.integers(0, 2)), seed=int(seed) & 0xFFFF)


@pytest.mark.parametrize("seed", _SEEDS)
def test_large_random_launches_vs_c_oracle(gpu, seed):
    import torch

    from squeezellm_amd import decode, synth

    c = _draw(seed)
    bits, K, batch = c["bits"], c["K"], c["batch"]
    layers = [synth.make_layer(K, N, bits, sparse_frac=c["sparse"], topX=c["topX"], heavy_rows=c["heavy"], device=gpu, seed=c["seed"] + j)
              for j, N in enumerate(c["Ns"])]
    g = torch.Generator(device=gpu).manual_seed(c["seed"])
    x = torch.randn((batch, K) if batch else (K,), device=gpu, generator=g, dtype=torch.float16).float()
    xs = [x] * len(layers)  # one tensor: the ops share their input and become ONE launch
    ys0 = [torch.randn((batch, l["N"]) if batch else (l["N"],), device=gpu, generator=g) * 0.01 for l in layers]
    ys = [y.clone() for y in ys0]
    seq = decode.OpSequence(layers, xs, ys, batched=batch > 0, fuse_shared_input=True, workspace=c["workspace"])
    assert seq.groups == [list(range(len(layers)))], c
    if c["graph"] and (c["workspace"] or batch <= 1):  # (a workspace-less batched capture takes scratch as graph memory nodes: tests/test_gpu_workspace.py)
        gr = seq.graph(warmup=0)
        for y, y0 in zip(ys, ys0):
            y.copy_(y0)
        gr.replay()
    else:
        seq.launch()
    torch.cuda.synchronize()
    lib = H.c_oracle()
    for l, y0, y in zip(layers, ys0, ys):
        ref = H.c_matvec(lib, _np_layer(l), x.cpu().numpy(), y0.cpu().numpy(), batched=batch > 0)
        err = H.rel_err(y.cpu().numpy(), ref)
        assert err <= TOL_FP64, f"{c}: op {l['K']}x{l['N']}: rel err {err:.2e}"
