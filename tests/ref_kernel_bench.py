#!/usr/bin/env python3
"""Head-to-head on one MI355X: the REFERENCE's own kernels (squeezellm/quant_cuda_kernel.cu
compiled unmodified for gfx950 by oracle/build_ref.sh -> oracle/_ref/libsqllm_ref.so) against this
repo's kernel, over one full pass of a model's quantised linears (batch 1, distinct weights).

Test infrastructure (it executes oracle/_ref), not product.  Prints one JSON line and writes it to
gpurun_out/ref_vs_ours_<config>.json.  For per-kernel device time run it under
`rocprofv3 --kernel-trace --stats` (the reference's kernels are named VecQuant*/SPMV*/DenseMatVec*).

The reference side is timed the way the reference runs -- eager launches on the legacy default
stream, 1-3 kernels per operator (quant_cuda_kernel.cu:439-506) -- but driven from a C loop
through ctypes with no Python between the kernels of an op and no per-op synchronisation, which
flatters it relative to its real PyTorch call path.  Ours: HIP-graph replay of the grouped pass
(bench.py's path) and, for a like-for-like launch discipline, eager per-op launches.

    python tests/ref_kernel_bench.py --config 7b-w4-s0 [--layers 32] [--reps 10]
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {"7b-w4-s0": ("llama-7b", 4, 0.0, 0), "7b-w3-s45": ("llama-7b", 3, 0.0045, 10),
           "7b-w4-s45": ("llama-7b", 4, 0.0045, 10), "7b-w3-s0": ("llama-7b", 3, 0.0, 0)}


def main():
    import torch

    import quant_cuda as qc
    from squeezellm_amd import decode, synth
    from tests import helpers as H

    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="7b-w4-s0", choices=sorted(CONFIGS))
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    name, bits, frac, topX = CONFIGS[a.config]
    dev = torch.device("cuda:0")
    layers = synth.make_model(name, bits, sparse_frac=frac, topX=topX, n_layers=a.layers, device=dev)
    n_dec = len(layers) // len(synth.MODEL_SHAPES[name]["linears"])
    scale = synth.MODEL_SHAPES[name]["layers"] / n_dec
    ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsqllm_ref.so"))
    P = ctypes.c_void_p
    xin, xs = {}, []
    for i, l in enumerate(layers):
        ln = l["name"].split(".")[-1]
        key = (i // 7, "h" if ln in ("q_proj", "k_proj", "v_proj") else "m" if ln in ("gate_proj", "up_proj") else ln)
        if key not in xin:
            xin[key] = torch.randn(l["K"], device=dev)
        xs.append(xin[key])
    ys_ref = [torch.zeros(l["N"], device=dev) for l in layers]
    ys = [torch.zeros(l["N"], device=dev) for l in layers]
    kind = "hybrid" if topX else ("spmv" if frac else "dense")

    def ref_pass():
        for l, x, y in zip(layers, xs, ys_ref):
            if kind == "dense":
                rc = ref.refk_dense(bits, 0, P(x.data_ptr()), P(l["qweight"].data_ptr()), P(y.data_ptr()),
                                    P(l["lookup_table"].data_ptr()), l["K"], l["N"])
            else:
                rc = ref.refk_hybrid(bits, 0, P(l["rows"].data_ptr()), P(l["cols"].data_ptr()), P(l["vals"].data_ptr()),
                                     l["vals"].numel(), P(x.data_ptr()), P(l["full_rows"].data_ptr()),
                                     P(l["full_row_indices"].data_ptr()), topX, P(y.data_ptr()), l["N"],
                                     P(l["qweight"].data_ptr()), P(l["lookup_table"].data_ptr()), l["K"], l["N"])
            assert rc == 0, rc

    def ours_eager():
        for l, x, y in zip(layers, xs, ys):
            H.call_op(qc, l, x, y, kind, False)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    # same results first (one pass each from zero)
    ref.refk_set_sync(1)
    ref_pass()
    ours_eager()
    torch.cuda.synchronize()
    worst = max(H.rel_err(a_.cpu().numpy(), b_.cpu().numpy()) for a_, b_ in zip(ys[:14], ys_ref[:14]))
    assert worst < 2e-5, worst
    ref.refk_set_sync(0)
    res = {"reference_kernels_eager_c_loop": timed(ref_pass, a.reps), "ours_eager_per_op_python": timed(ours_eager, a.reps)}
    seq = decode.OpSequence(layers, xs, ys, fuse_shared_input=True)
    res["ours_one_call_grouped"] = timed(seq.launch, a.reps)
    g = seq.graph()
    res["ours_graph_grouped"] = timed(g.replay, a.reps)
    out = {"config": a.config, "decoder_layers_timed": n_dec, "max_rel_diff_first_14_ops": worst,
           "ms_per_pass_full_model": {k: round(v * scale, 4) for k, v in res.items()},
           "tokens_per_s": {k: round(1e3 / (v * scale), 1) for k, v in res.items()}}
    out["speedup_graph_vs_reference"] = round(res["reference_kernels_eager_c_loop"] / res["ours_graph_grouped"], 2)
    line = json.dumps(out)
    print(line)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"ref_vs_ours_{a.config}.json"), "w") as f:
        f.write(line + "\n")


if __name__ == "__main__":
    main()
