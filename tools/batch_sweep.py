#!/usr/bin/env python3
"""Batched operators: parity against the numpy oracle on a small shape and device time per launch by
batch size, through the 8-row batch tiles ("tile"), the column-lane kernel ("cols") and the
matrix-core kernel ("mfma"), on distinct weight copies (HBM-resident stream).

    python tools/batch_sweep.py [--shape 5120x13824] [--bits 4] [--batches 1,2,4,8,16] [--check]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="5120x13824")
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--sparse", type=float, default=0.0045)
    ap.add_argument("--topx", type=int, default=10)
    ap.add_argument("--batches", default="1,2,4,8,16,32,64,128")
    ap.add_argument("--paths", default="tile,cols,mfma")
    ap.add_argument("--sparse-transpose", type=int, default=1, help="0: the wide-batch CSR role gathers from vec itself")
    ap.add_argument("--total-mb", type=float, default=600.0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--vec", default="fp16", choices=["fp16", "fp32"], help="fp16: vec values born in fp16 (what forward passes); fp32: full-mantissa values")
    ap.add_argument("--check", action="store_true", help="parity of the matrix-core path vs the oracle on a small shape")
    args = ap.parse_args()
    import numpy as np
    import torch

    from squeezellm_amd import _lib, decode, quant_cuda as qc, synth

    dev = torch.device("cuda:0")
    _lib.set_option("sparse_transpose", args.sparse_transpose)
    if args.check:
        import helpers as H

        worst = 0.0
        for path in ("mfma", "cols"):
          _lib.set_option("mfma_min_batch", 1 if path == "mfma" else 1 << 30)
          _lib.set_option("cols_min_batch", 1)
          _lib.set_option("cols_max_batch", 1 << 30)
          for bits in (3, 4):
              for kind in ("dense", "spmv", "hybrid"):
                  for (K, N) in ((256, 192), (1024, 776)):
                      case = H.make_case(bits, K, N, sparse=0.02 if kind != "dense" else 0, topX=5 if kind == "hybrid" else 0,
                                         heavy_rows=2 if kind != "dense" else 0, seed=K + bits)
                      t = H.to_torch(case, dev)
                      for B in (1, 2, 3, 4, 5, 8, 9, 16, 17, 33, 64, 65, 130):
                          rng = np.random.default_rng(B)
                          x = rng.normal(size=(B, K)).astype(np.float32)
                          mul = rng.normal(size=(B, N)).astype(np.float32)
                          y = torch.from_numpy(mul.copy()).to(dev)
                          H.call_op(qc, t, torch.from_numpy(x).to(dev), y, kind, True)
                          torch.cuda.synchronize()
                          err = H.rel_err(y.cpu().numpy(), H.oracle_ref(case, x, mul, kind))
                          worst = max(worst, err)
                          if err > 2e-5:
                              print("MISMATCH", path, bits, kind, K, N, B, err, flush=True)
        print(json.dumps(dict(check="mfma and cols paths vs oracle", worst_rel_err=worst, ok=bool(worst <= 2e-5))), flush=True)

    K, N = map(int, args.shape.split("x"))
    one = synth.algorithmic_bytes(K, N, args.bits)
    copies = max(4, int(args.total_mb * 1e6 / one))
    layers = [synth.make_layer(K, N, args.bits, sparse_frac=args.sparse, topX=args.topx if args.sparse > 0 else 0,
                               heavy_rows=10 if args.sparse > 0 else 0, device=dev, seed=i) for i in range(copies)]
    for B in map(int, args.batches.split(",")):
        # (fp16-born vec, as QuantLinearLUT.forward passes it: x.float() of a half tensor; --vec fp32 for full-mantissa values)
        xs = [torch.randn((B, K), device=dev) if args.vec == "fp32" else torch.randn((B, K), device=dev, dtype=torch.float16).float() for _ in layers]
        ys = [torch.zeros((B, N), device=dev) for _ in layers]
        nbytes = synth.layer_bytes(layers[0], B)
        for path in args.paths.split(","):
            _lib.set_option("mfma_min_batch", 1 if path == "mfma" else 1 << 30)
            _lib.set_option("cols_min_batch", 1 if path == "cols" else 1 << 30)
            _lib.set_option("cols_max_batch", 1 << 30)
            seq = decode.OpSequence(layers, xs, ys, batched=True)
            seq.profile(reps=1)
            us = seq.profile(reps=args.reps)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(args.reps):
                seq.launch()  # wall time per op, everything the call enqueues included (vec transpose, scratch)
            ev1.record()
            torch.cuda.synchronize()
            wall = ev0.elapsed_time(ev1) * 1e3 / (args.reps * len(layers))
            plan = _lib.plan_query(args.bits, K, N, B, nnz=layers[0]["vals"].numel() if args.sparse else 0, topX=args.topx if args.sparse else 0)
            print(json.dumps(dict(shape=args.shape, bits=args.bits, batch=B, path=path, grid=[plan["grid_x"], plan["grid_y"]], k_slices=plan["k_slices"],
                                  us_mean=round(float(us.mean()), 2), us_min=round(float(us.min()), 2), wall_us=round(wall, 2), GBps=round(nbytes / us.mean() / 1e3, 1),
                                  TFLOPs=round(2.0 * B * K * N / us.mean() / 1e6, 2))), flush=True)
        del xs, ys


if __name__ == "__main__":
    main()
