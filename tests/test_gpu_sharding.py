"""GPU side of the column (N) split (squeezellm_amd/sharding.py, SURVEY.md 8(e) path 2): the HIP kernels on
column-sharded operands, every shard's result concatenated, against the unsharded op and the oracle; and the
one-rank form of the column-parallel pass (the collective path itself is covered with gloo on CPU)."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _npl(lay):
    import torch

    return {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in lay.items()}


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_column_shards_run_on_the_kernels(gpu, bits, world):
    import torch

    from squeezellm_amd import decode, sharding, synth

    K, N = 2048, 1096  # 17 blocks of 64 + a ragged one
    lay = synth.make_layer(K, N, bits, sparse_frac=0.01, topX=6, heavy_rows=3, device=gpu, seed=21 + bits)
    x = torch.randn(K, device=gpu)
    ref = H.c_matvec(H.c_oracle(), _npl(lay), x.cpu().numpy(), np.zeros(N, np.float32), batched=False)
    parts = []
    for r in range(world):
        sh = sharding.shard_layer_columns(lay, r, world)
        y = torch.zeros(sh["N"], device=gpu)
        if sh["N"]:
            decode.OpSequence([sh], [x], [y]).launch()
        parts.append(y)
    torch.cuda.synchronize()
    got = torch.cat(parts).cpu().numpy()
    assert got.shape == (N,) and H.rel_err(got, ref) <= 2e-5


@pytest.mark.parametrize("graph", [False, True])
def test_column_parallel_pass_one_rank(gpu, graph):
    import torch

    from squeezellm_amd import sharding, synth

    spec = [("q", 1024, 512), ("k", 1024, 512), ("v", 1024, 512), ("o", 512, 1024), ("g", 1024, 1408), ("u", 1024, 1408), ("d", 1408, 1024)]
    layers = [synth.make_layer(K, N, 4, sparse_frac=0.01, topX=4, heavy_rows=1, device=gpu, seed=60 + i) for i, (_, K, N) in enumerate(spec)]
    xh, xo, xm = torch.randn(1024, device=gpu), torch.randn(512, device=gpu), torch.randn(1408, device=gpu)
    xs = [xh, xh, xh, xo, xh.clone(), None, xm]
    xs[5] = xs[4]
    cp = sharding.ColumnParallelPass(layers, xs, rank=0, world_size=1, device=gpu, graph=graph)
    assert cp.groups == [[0, 1, 2], [3], [4, 5], [6]] and (cp.graph is not None) == graph
    cp.step()
    cp.step()  # (a second pass starts from zeroed slices again)
    torch.cuda.synchronize()
    lib = H.c_oracle()
    for gi, grp in enumerate(cp.groups):
        for j, i in enumerate(grp):
            ref = H.c_matvec(lib, _npl(layers[i]), xs[i].cpu().numpy(), np.zeros(layers[i]["N"], np.float32), batched=False)
            assert H.rel_err(cp.result(gi, j, layers[i]["N"]).cpu().numpy(), ref) <= 2e-5


def test_ring_pipeline_whole_tick_capture_matches_eager(gpu):
    """One rank: the tick captured as ONE graph (stage kernels + hand-over; the collective joins it at world > 1)
    reproduces the eager ticks bit for bit over several turns of the ring."""
    import torch

    from squeezellm_amd import sharding, synth

    hidden = 512
    layers = [synth.make_layer(K, N, 4, sparse_frac=0.01, topX=2, heavy_rows=1, device=gpu, seed=80 + i)
              for i, (K, N) in enumerate([(512, 512), (512, 1024), (1024, 512)])]
    h0 = torch.randn(hidden, device=gpu).half()
    outs = []
    for captured in (False, True):
        stage = sharding.DecodeStage(layers, hidden, gpu, seed=3, graph=False)
        pipe = sharding.RingPipeline(stage, hidden, rank=0, world_size=1, device=gpu, h0=h0)
        if captured:
            assert pipe.capture() and pipe.capture()  # (idempotent)
        seq = []
        for _ in range(4):
            seq.append(pipe.tick().clone())
        torch.cuda.synchronize()
        assert pipe.ticks == 4
        outs.append(torch.stack(seq).float().cpu())
    assert torch.isfinite(outs[0]).all() and float(outs[0].abs().max()) > 0
    # fp32 atomics: summation order may differ between the two runs; one fp16 ulp of slack
    assert torch.allclose(outs[0], outs[1], rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("mode,env", [("columns", {}), ("pipeline", {}), ("pipeline", {"SQLLM_PIPELINE_CAPTURE": "1"})],
                         ids=["columns", "pipeline-eager-collective", "pipeline-tick-captured"])
def test_bench_distributed_modes_run_on_one_rccl_rank(gpu, mode, env):
    """bench.py's N > 1 modes with a real RCCL process group of ONE rank (SQLLM_BENCH_FORCE_DIST=1): init, the collective
    per launch group / per tick (captured into the tick's graph, and eager), barrier + MAX reduction of the timings, the
    JSON line -- so that the driver's first 8-GPU run is not also the first run of this code."""
    import json
    import os
    import subprocess
    import sys

    e = dict(os.environ, SQLLM_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + hash((mode, tuple(env))) % 300),
             RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", **env)
    cmd = [sys.executable, os.path.join(H.ROOT, "bench.py"), "--gpus", "1", "--parallel", mode, "--config", "7b-w4-s45", "--layers", "3",
           "--steps", "4", "--warmup", "2", "--repeats", "2", "--no-cpu-baseline", "--no-roofline", "--no-sub-records"]
    r = subprocess.run(cmd, cwd=H.ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["rccl_ranks"] == 1
    assert line["scaling"] == "strong" and line["steps"] == 4
    if mode == "pipeline":
        assert line["pipeline_tick_captured"] is True  # (one rank: the whole tick is always captured)
    else:
        assert line["column_parallel"]["graph_captured"] in (True, False)


def test_bench_8gpu_command_dry_run_on_one_rank(gpu):
    """`python bench.py --gpus 8 --config 65b-w3-s45` under SQLLM_BENCH_FORCE_DIST=1 on ONE rank runs the code
    path the 8-GPU command takes (layer-sharded ring pipeline, one RCCL all-gather per tick, the tick captured) and says so."""
    import json
    import os
    import subprocess
    import sys

    e = dict(os.environ, SQLLM_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29977", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    cmd = [sys.executable, os.path.join(H.ROOT, "bench.py"), "--gpus", "8", "--config", "65b-w3-s45", "--layers", "2", "--steps", "3", "--warmup", "1",
           "--repeats", "2", "--no-cpu-baseline", "--no-roofline", "--no-sub-records"]
    r = subprocess.run(cmd, cwd=H.ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["parallelism"].startswith("pp8") and "DRY RUN" in line["config"]["parallelism"]
    assert line["config"]["rccl_ranks"] == 1 and line["pipeline_tick_captured"] is True and line["value"] > 0
