"""How much of a decode pass is spent AROUND the operator?  (report the op-only
number and the number through QuantLinearLUT.forward.)

Times one pass over every quantised linear of a model (batch 1, fp16 activations, distinct weights
per layer) six ways and prints one JSON line:

  forward_eager        QuantLinearLUT.forward per linear, eager   (4 launches per linear + Python)
  forward_graph        the same calls captured in one HIP graph   (4 launches per linear)
  fused_eager          QuantLinearLUTFused.forward per linear     (1 launch per linear + Python)
  fused_graph          the same captured in one HIP graph         (1 launch per linear)
  linear_groups_graph  OpSequence(linear=True, fuse_shared_input) (4 launches per decoder layer)
  op_groups_graph      OpSequence operator-only (bench.py's path; fp32 in, accumulate, no cast)

    python tools/forward_bench.py --config 7b-w4-s0 [--layers 32] [--reps 20]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from squeezellm_amd import decode, quant, synth  # noqa: E402

CONFIGS = {"7b-w4-s0": ("llama-7b", 4, 0.0, 0), "7b-w3-s45": ("llama-7b", 3, 0.0045, 10),
           "7b-w4-s45": ("llama-7b", 4, 0.0045, 10), "13b-w4-s45": ("llama-13b", 4, 0.0045, 10)}


def timed(fn, reps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="7b-w4-s0", choices=sorted(CONFIGS))
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    name, bits, frac, topX = CONFIGS[a.config]
    dev = torch.device("cuda:0")
    layers = synth.make_model(name, bits, sparse_frac=frac, topX=topX, n_layers=a.layers, device=dev)
    n_dec = len(layers) // len(synth.MODEL_SHAPES[name]["linears"])
    full = synth.MODEL_SHAPES[name]["layers"]
    scale = full / n_dec
    mods = [quant.QuantLinearLUT.from_operands(l) for l in layers]
    fmods = [quant.QuantLinearLUT.from_operands(l) for l in layers]
    for m in fmods:
        m.__class__ = quant.QuantLinearLUTFused
    # one input per distinct K; linears of a decoder layer that share an input share the tensor
    xin = {}
    xs16 = []
    for i, l in enumerate(layers):
        lname = l["name"].split(".")[-1]
        key = (i // 7, "h" if lname in ("q_proj", "k_proj", "v_proj") else "m" if lname in ("gate_proj", "up_proj") else lname)
        if key not in xin:
            xin[key] = torch.randn((1, 1, l["K"]), device=dev).half()
        xs16.append(xin[key])

    def run(ms):
        with torch.no_grad():
            for m, x in zip(ms, xs16):
                m(x)

    res = {}
    res["forward_eager"] = timed(lambda: run(mods), max(3, a.reps // 4))
    g = capture(lambda: run(mods))
    res["forward_graph"] = timed(g.replay, a.reps)
    res["fused_eager"] = timed(lambda: run(fmods), max(3, a.reps // 4))
    g2 = capture(lambda: run(fmods))
    res["fused_graph"] = timed(g2.replay, a.reps)
    xs2 = [x.reshape(-1) for x in xs16]
    ys16 = [torch.empty(l["N"], device=dev, dtype=torch.float16) for l in layers]
    seq = decode.OpSequence(layers, xs2, ys16, fuse_shared_input=True, linear=True)
    g3 = seq.graph()
    res["linear_groups_graph"] = timed(g3.replay, a.reps)
    x32 = {id(x): x.float().reshape(-1) for x in xs16}
    xs32 = [x32[id(x)] for x in xs16]
    ys32 = [torch.zeros(l["N"], device=dev) for l in layers]
    seq2 = decode.OpSequence(layers, xs32, ys32, fuse_shared_input=True)
    g4 = seq2.graph()
    res["op_groups_graph"] = timed(g4.replay, a.reps)
    out = {"config": a.config, "decoder_layers_timed": n_dec, "ms_per_pass_full_model": {k: round(v * scale, 4) for k, v in res.items()},
           "tokens_per_s": {k: round(1e3 / (v * scale), 1) for k, v in res.items()}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
