#!/usr/bin/env python3
"""Generate tests/golden/pack2_*.npz by running the UNMODIFIED reference packer
(`QuantLinearLUT.pack2`, /root/reference/squeezellm/quant.py:97-208) on seeded inputs.

Runs only in the dev container (needs /root/reference); the .npz files are committed so the
parity tests can pin oracle/sqllm_oracle.py's format restatement anywhere.

`squeezellm/quant.py:5` does `import quant_cuda` at module top, and the reference's CUDA extension
cannot be built here, so an empty stub module of that name is injected first -- pack2 never calls
into it.

    python tests/golden/make_pack2_golden.py
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_quant():
    sys.modules.setdefault("quant_cuda", types.ModuleType("quant_cuda"))  # stub, never called
    sys.path.insert(0, REF)
    from squeezellm import quant as refquant  # noqa: E402

    return refquant


def make_case(refquant, bits, K, N, sparse, balanced, seed):
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, 1 << bits, size=(N, K), dtype=np.int64)
    # k-means-like per-channel centroids; one channel gets an exact |tie| and one an exact zero
    lut = np.sort(rng.normal(0, 0.02, size=(N, 1 << bits)).astype(np.float32), axis=1)
    lut[3, 0], lut[3, 1] = -0.005, 0.005
    lut[3, 2:] = np.abs(lut[3, 2:]) + 0.01
    lut[5, 2] = 0.0
    outl = np.zeros((N, K), np.float32)
    if sparse:
        mask = rng.random((N, K)) < 0.02
        mask[7, :] = rng.random(K) < 0.5  # one heavy row
        mask[11, :] = False  # one empty row
        outl[mask] = rng.normal(0, 0.3, size=int(mask.sum())).astype(np.float32)
        # an outlier exactly equal to the zero-mapping centroid vanishes in to_sparse_csr
        zm = lut[np.arange(N), np.argmin(np.abs(lut), axis=1)]
        outl[13, 5] = zm[13]

    linear = torch.nn.Linear(K, N, bias=True)
    with torch.no_grad():
        linear.weight.zero_()
        linear.bias.copy_(torch.from_numpy(rng.normal(0, 0.01, N).astype(np.float32)))
    lut_arg = [[(lut[c].copy(), idx[c].copy())] for c in range(N)]
    outl_t = torch.from_numpy(outl.copy()).to_sparse() if sparse else None

    with contextlib.redirect_stdout(io.StringIO()):  # the reference prints its buffers
        layer = refquant.QuantLinearLUT(
            bits, K, N, True, include_sparse=sparse, numvals=int((outl != 0).sum()), topX=0,
            balanced=balanced, num_nonzero_per_thread=10,
        )
        layer.pack2(linear, (lut_arg, outl_t), include_sparse=sparse, num_nonzero_per_thread=10)

    out = dict(
        bits=np.int32(bits), K=np.int32(K), N=np.int32(N),
        idx_nk=idx.astype(np.uint8), lut=lut, outliers_nk=outl,
        qweight=layer.qweight.numpy().astype(np.int32),
        lookup_table=layer.lookup_table.numpy().astype(np.float32),
        bias=layer.bias.detach().numpy().astype(np.float32),
    )
    if sparse:
        out.update(
            rows=layer.rows.numpy().astype(np.int32),
            cols=layer.cols.numpy().astype(np.int32),
            vals=layer.vals.numpy().astype(np.float32),
        )
        if balanced:
            out.update(startrows=layer.startrows.numpy().astype(np.int32), num_threads=np.int32(layer.num_threads))
    return out


def main():
    refquant = load_reference_quant()
    cases = [
        ("pack2_w4_dense", 4, 256, 128, False, False, 1),
        ("pack2_w3_dense", 3, 256, 128, False, False, 2),
        ("pack2_w4_sparse", 4, 128, 128, True, False, 3),
        ("pack2_w3_sparse_balanced", 3, 128, 128, True, True, 4),
    ]
    for name, bits, K, N, sparse, balanced, seed in cases:
        case = make_case(refquant, bits, K, N, sparse, balanced, seed)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **case)
        print(f"wrote {path}: qweight {case['qweight'].shape}" + (f", nnz {len(case['vals'])}" if sparse else ""))


if __name__ == "__main__":
    main()
