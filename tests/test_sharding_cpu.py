"""The N > 1 path on CPU: world_size-2 gloo processes run the layer-sharded ring pipeline
(squeezellm_amd/sharding.py) with an injected CPU stage and must reproduce the single-process
result; plus the layer partition arithmetic."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from squeezellm_amd import sharding
from tests import helpers as H


def test_partition_layers():
    assert sharding.partition_layers(32, 8) == [(4 * i, 4 * i + 4) for i in range(8)]
    assert sharding.partition_layers(80, 8)[0] == (0, 10)
    p = sharding.partition_layers(10, 4)
    assert p == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert sharding.partition_layers(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    with pytest.raises(ValueError):
        sharding.partition_layers(4, 0)
    for L in range(0, 40):
        for W in range(1, 9):
            parts = sharding.partition_layers(L, W)
            assert parts[0][0] == 0 and parts[-1][1] == L
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


HIDDEN, LAYERS, TICKS = 64, 4, 8


def _make_layers(n_layers=LAYERS):
    """A tiny 'decoder stack': n_layers square quantised linears with bias-free hybrid operands."""
    return [H.make_case(4 if i % 2 else 3, HIDDEN, HIDDEN, sparse=0.05, topX=2, seed=100 + i) for i in range(n_layers)]


def _stage_fn(layers):
    """CPU stage: the oracle stands in for the GPU kernels (test double), then an RMS-norm so values
    stay O(1) around the ring, like sharding.DecodeStage."""
    def fn(h):
        v = h.double().numpy()
        for lay in layers:
            v = H.oracle_ref(lay, v.astype(np.float32), np.zeros(HIDDEN, np.float32), "hybrid")
            v = v / np.sqrt((v * v).mean() + 1e-6)
        return torch.from_numpy(v).to(h.dtype)
    return fn


def _h0(r):
    return torch.from_numpy(np.random.default_rng(500 + r).normal(size=HIDDEN)).float()


def _reference_ring(world, n_layers=LAYERS):
    """Single-process emulation of the ring schedule for `world` stages."""
    layers = _make_layers(n_layers)
    parts = sharding.partition_layers(n_layers, world)
    stages = [_stage_fn(layers[a:b]) for a, b in parts]
    h_in = [_h0(r) for r in range(world)]
    outs = None
    for _ in range(TICKS):
        outs = [stages[r](h_in[r]) for r in range(world)]
        h_in = [outs[(r - 1) % world] for r in range(world)]
    return torch.stack(outs)


def _worker(rank, world, port, q, n_layers=LAYERS):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    layers = _make_layers(n_layers)
    a, b = sharding.partition_layers(n_layers, world)[rank]
    pipe = sharding.RingPipeline(_stage_fn(layers[a:b]), HIDDEN, rank=rank, world_size=world, device="cpu",
                                 dtype=torch.float32, h0=_h0(rank))
    out = None
    for _ in range(TICKS):
        out = pipe.tick()
    q.put((rank, out.numpy(), pipe.buf.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


# world 2: even split; world 4 over 6 layers: uneven ranges (2, 2, 1, 1); world 4 over 3 layers: the last
# rank owns NO layer and passes the hidden state through
@pytest.mark.parametrize("world,n_layers", [(2, LAYERS), (4, 6), (4, 3)])
def test_ring_pipeline_gloo_matches_single_process(world, n_layers):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7 * world + n_layers
    assert [b - a for a, b in sharding.partition_layers(n_layers, world)] == {(2, 4): [2, 2], (4, 6): [2, 2, 1, 1], (4, 3): [1, 1, 1, 0]}[(world, n_layers)]
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, n_layers)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _reference_ring(world, n_layers).numpy()
    for rank, out, buf in res:
        assert np.allclose(out, want[rank], rtol=1e-5, atol=1e-6)
        assert np.allclose(buf, want, rtol=1e-5, atol=1e-6)  # everyone gathered everyone's output


def test_ring_pipeline_world1_is_a_plain_loop():
    layers = _make_layers()
    pipe = sharding.RingPipeline(_stage_fn(layers), HIDDEN, rank=0, world_size=1, device="cpu", dtype=torch.float32, h0=_h0(0))
    pipe.run(3)
    h = _h0(0)
    f = _stage_fn(layers)
    for _ in range(3):
        h = f(h)
    assert torch.allclose(pipe.h_in, h)


# ---- column (N) sharding inside a layer: SURVEY.md 8(e) path 2 ----
def _torch_case(bits, K, N, seed):
    case = H.make_case(bits, K, N, sparse=0.03, topX=5, heavy_rows=2, seed=seed)
    return case, {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k, v in case.items()}


def _oracle_on(layer_t, x):
    npl = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in layer_t.items()}
    kind = "hybrid" if npl.get("full_rows") is not None else ("spmv" if npl.get("vals") is not None else "dense")
    return np.asarray(H.oracle_ref(npl, x, np.zeros(npl["N"], np.float32), kind))


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shard_layer_columns_partitions_the_result(bits, world):
    """Slicing qweight / lookup_table / CSR rows / top-X rows by output column and concatenating the slices'
    results reproduces the unsharded op exactly (no reduction across ranks)."""
    K, N = 256, 328  # 5 blocks of 64 + a ragged one: uneven and (for world 8) empty ranges
    case, lay = _torch_case(bits, K, N, seed=40 + bits)
    x = np.random.default_rng(3).normal(size=K).astype(np.float32)
    want = np.asarray(H.oracle_ref(case, x, np.zeros(N, np.float32), "hybrid"))
    ranges = sharding.column_ranges(N, world)
    assert ranges[0][0] == 0 and ranges[-1][1] == N and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    assert all(a % 64 == 0 for a, b in ranges if b > a)
    parts, topx_seen = [], 0
    for r in range(world):
        sh = sharding.shard_layer_columns(lay, r, world)
        assert sh["N"] == ranges[r][1] - ranges[r][0] and sh["qweight"].shape == (K // 32 * bits, sh["N"])
        assert int(sh["rows"][0]) == 0 and int(sh["rows"][-1]) == sh["vals"].numel() == sh["cols"].numel()
        topx_seen += 0 if sh["full_rows"] is None else sh["full_rows"].shape[1]
        parts.append(_oracle_on(sh, x) if sh["N"] else np.zeros(0))
    assert topx_seen == 5  # every top-X row belongs to exactly one rank
    assert np.allclose(np.concatenate(parts), want, rtol=1e-12, atol=1e-14)  # (fp64 oracle; only the order of the terms differs)


def _col_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, lay = _torch_case(4, 256, 328, seed=44)
    sh = sharding.shard_layer_columns(lay, rank, world)
    op = sharding.ColumnParallelOp(lambda x: torch.from_numpy(_oracle_on(sh, x.numpy())).float(), 328, rank=rank,
                                   world_size=world, device="cpu")
    g = torch.Generator().manual_seed(9)
    outs = [op(torch.randn(256, generator=g)).numpy().copy() for _ in range(3)]
    q.put((rank, outs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_column_parallel_op_gloo_matches_single_process(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 100 + world
    procs = [ctx.Process(target=_col_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    case, _ = _torch_case(4, 256, 328, seed=44)
    g = torch.Generator().manual_seed(9)
    for i in range(3):
        x = torch.randn(256, generator=g).numpy()
        want = np.asarray(H.oracle_ref(case, x, np.zeros(328, np.float32), "hybrid"))
        for rank, outs in res:
            assert outs[i].shape == (328,) and np.allclose(outs[i], want, rtol=1e-6, atol=1e-7), f"rank {rank} call {i}"
