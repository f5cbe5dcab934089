"""Full-size parity of the path bench.py times: one whole decoder layer enqueued through
`OpSequence(fuse_shared_input=True)` -- q/k/v as ONE launch of three ops, o_proj alone, gate/up as ONE
launch of two, down_proj alone -- every op against the C oracle.

The grouped launches are planned differently from single-op launches (sqllm_capi.hip: make_plan with
ops_in_launch > 1: other workgroup counts, other K slices, a block-prefix table over the segments), so
the single-op full-size tests of test_gpu_parity.py do not cover them.  Reference arithmetic:
squeezellm/quant_cuda_kernel.cu:831-880 (w4), :741-828 (w3), :1040-1089 (CSR), :1092-1164 (top-X),
batched :884-1038; the forward that issues them squeezellm/quant.py:211-383.
"""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

TOL_FP64 = 2e-5  # fp32 accumulation in unspecified (atomic) order vs the fp64 oracle, max-norm relative


def _np_layer(lay):
    import torch

    return {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in lay.items()}


def _decoder_layer(model, bits, sparse, topX, gpu, seed0):
    from squeezellm_amd import synth

    spec = synth.MODEL_SHAPES[model]["linears"]
    return [dict(synth.make_layer(K, N, bits, sparse_frac=sparse, topX=topX, heavy_rows=10 if sparse else 0, device=gpu,
                                  seed=seed0 + j), name=name) for j, (name, K, N) in enumerate(spec)]


def _inputs(layers, gpu, batch):
    """q/k/v share one tensor, gate/up another (squeezellm/model_parse.py:53-61)."""
    import torch

    g = torch.Generator(device=gpu)
    g.manual_seed(1234)
    shared = {"k_proj": "q_proj", "v_proj": "q_proj", "up_proj": "gate_proj"}
    xs, last = [], {}
    for l in layers:
        src = shared.get(l["name"])
        if src in last:
            xs.append(last[src])
        else:
            shape = (batch, l["K"]) if batch else (l["K"],)
            xs.append(torch.randn(shape, device=gpu, generator=g, dtype=torch.float16).float())
        last[l["name"]] = xs[-1]
    return xs


def _check_layer(layers, gpu, batch, graph):
    import torch

    from squeezellm_amd import decode

    xs = _inputs(layers, gpu, batch)
    g = torch.Generator(device=gpu)
    g.manual_seed(99)
    # accumulate semantics: mul starts from a non-zero value (squeezellm/quant.py:214-219: bias.clone())
    ys0 = [torch.randn((batch, l["N"]) if batch else (l["N"],), device=gpu, generator=g) * 0.01 for l in layers]
    ys = [y.clone() for y in ys0]
    seq = decode.OpSequence(layers, xs, ys, batched=batch > 0, fuse_shared_input=True)
    assert seq.groups == [[0, 1, 2], [3], [4, 5], [6]]
    if graph:
        gr = seq.graph(warmup=0)
        for y, y0 in zip(ys, ys0):
            y.copy_(y0)
        gr.replay()
    else:
        seq.launch()
    torch.cuda.synchronize()
    lib = H.c_oracle()
    for l, x, y0, y in zip(layers, xs, ys0, ys):
        ref = H.c_matvec(lib, _np_layer(l), x.cpu().numpy(), y0.cpu().numpy(), batched=batch > 0)
        err = H.rel_err(y.cpu().numpy(), ref)
        assert err <= TOL_FP64, f"{l['name']} {l['K']}x{l['N']} w{l['bits']} batch {batch}: rel err {err:.2e}"


@pytest.mark.parametrize("bits,sparse,topX", [(4, 0.0, 0), (4, 0.0045, 10), (3, 0.0045, 10), (3, 0.0, 0)],
                         ids=["w4-s0", "w4-s45", "w3-s45", "w3-s0"])
def test_llama7b_decoder_layer_grouped(gpu, bits, sparse, topX):
    """BASELINE configs[1] / [2] exactly as bench.py enqueues them (graph replay of the grouped pass)."""
    layers = _decoder_layer("llama-7b", bits, sparse, topX, gpu, seed0=100 * bits)
    _check_layer(layers, gpu, batch=0, graph=True)


@pytest.mark.parametrize("bits,sparse,topX", [(4, 0.0, 0), (3, 0.0045, 10), (3, 0.0, 0)], ids=["w4-s0", "w3-s45", "w3-s0"])
@pytest.mark.parametrize("route", ["fused-only", "column-lane-everything"])
def test_llama7b_decoder_layer_both_batch1_routes(gpu, bits, sparse, topX, route):
    """Batch 1 is routed per launch shape since round 6 (sqllm_capi.hip: cols_pays_batch1 -- the dense-only q/k/v group and down_proj, the 3-bit
    gate/up ... on the column-lane kernel, the rest on the fused kernel).  The default routing is test_llama7b_decoder_layer_grouped above; here
    the same layers with EVERY launch on the fused kernel (cols_min_batch huge: the route those shapes had until then) and with every launch on the
    column-lane kernel (cols_min_batch = cols_max_batch = 1), against the C oracle."""
    from squeezellm_amd import _lib

    layers = _decoder_layer("llama-7b", bits, sparse, topX, gpu, seed0=100 * bits + 7)
    opts = {"cols_min_batch": 1 << 30} if route == "fused-only" else {"cols_min_batch": 1, "cols_max_batch": 1}
    for k, v in opts.items():
        _lib.set_option(k, v)
    try:
        _check_layer(layers, gpu, batch=0, graph=False)
        _check_layer(layers, gpu, batch=1, graph=False)  # (the *_batched operators with one row take the same routes)
    finally:
        for k in opts:
            _lib.set_option(k, 0)


@pytest.mark.parametrize("batch", [1, 2, 3, 4, 5, 6, 7, 8, 12, 16])
def test_llama13b_decoder_layer_grouped_batched(gpu, batch):
    """BASELINE configs[3]: 13B shapes, w4 s45, the *_batched operators at every row count of "batch 1..8" and beyond, grouped, by
    the default routing: batch tiles of exactly that many rows / column-lane passes up to 6 rows, the fused small launch from 7
    (o_proj alone on the 7- / 8-row tile)."""
    layers = _decoder_layer("llama-13b", 4, 0.0045, 10, gpu, seed0=1300)
    _check_layer(layers, gpu, batch=batch, graph=False)


@pytest.mark.parametrize("batch", [3, 5, 7])
def test_llama13b_decoder_layer_w3_odd_rows(gpu, batch):
    """3-bit 13B layer at 3 / 5 / 7 rows: the column-lane kernel's passes of exactly that many rows (q/k/v group, gate/up, down_proj)
    and the batch tile of that width (o_proj), grouped, against the C oracle."""
    layers = _decoder_layer("llama-13b", 3, 0.0045, 10, gpu, seed0=1360)
    _check_layer(layers, gpu, batch=batch, graph=False)


def test_llama13b_decoder_layer_w3_batch4(gpu):
    layers = _decoder_layer("llama-13b", 3, 0.0045, 10, gpu, seed0=1350)
    _check_layer(layers, gpu, batch=4, graph=False)


@pytest.mark.parametrize("bits", [4, 3])
@pytest.mark.parametrize("batch", [5, 8, 13])
def test_llama13b_decoder_layer_small_split_groups(gpu, bits, batch):
    """Up to 16 rows a GROUP of ops (q/k/v, gate/up) is one launch of the split matrix-core kernel with the CSR chunks
    and top-X slabs of every op in the same grid (csrc/sqllm_mfma_split.hip: sqllm_fused_small_split); forced from 5
    rows up here, whatever the router's default switch-over is."""
    from squeezellm_amd import _lib

    layers = _decoder_layer("llama-13b", bits, 0.0045, 10, gpu, seed0=1400 + bits)
    before = _lib.get_option("mfma_min_batch")
    _lib.set_option("mfma_min_batch", 5)
    try:
        _check_layer(layers, gpu, batch=batch, graph=False)
    finally:
        _lib.set_option("mfma_min_batch", before)
