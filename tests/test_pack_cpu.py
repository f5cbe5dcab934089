"""CPU tests of squeezellm_amd.pack against the UNMODIFIED reference packer's output
(tests/golden/pack2_*.npz) and of the top-X extraction against the oracle."""
import glob
import os

import numpy as np
import pytest
import torch

from squeezellm_amd import pack
from tests import helpers as H

PACK2 = sorted(glob.glob(os.path.join(H.GOLDEN, "pack2_*.npz")))


@pytest.mark.parametrize("path", PACK2, ids=os.path.basename)
def test_pack_matches_reference_pack2_bit_exactly(path):
    g = np.load(path)
    bits = int(g["bits"])
    idx_nk = torch.from_numpy(g["idx_nk"].astype(np.int64))
    q = pack.pack_qweight(idx_nk.t().contiguous(), bits)
    assert q.dtype == torch.int32 and np.array_equal(q.numpy(), g["qweight"])
    assert np.array_equal(pack.unpack_qweight(torch.from_numpy(g["qweight"]), bits).numpy(), g["idx_nk"].T)
    if "rows" in g.files:
        rows, cols, vals = pack.outliers_to_csr(torch.from_numpy(g["outliers_nk"]), torch.from_numpy(g["lookup_table"]))
        assert np.array_equal(rows.numpy(), g["rows"]) and np.array_equal(cols.numpy(), g["cols"])
        assert np.array_equal(vals.numpy(), g["vals"])


@pytest.mark.parametrize("bits", [3, 4])
def test_pack_roundtrip_and_any_bit_pattern(bits):
    gen = torch.Generator().manual_seed(bits)
    idx = torch.randint(0, 1 << bits, (96, 37), generator=gen)
    q = pack.pack_qweight(idx, bits)
    assert torch.equal(pack.unpack_qweight(q, bits).to(torch.int64), idx)
    assert np.array_equal(q.numpy(), H.oracle.pack_indices(idx.numpy(), bits))
    raw = torch.randint(-(2**31), 2**31, q.shape, generator=gen, dtype=torch.int64).to(torch.int32)
    assert torch.equal(pack.pack_qweight(pack.unpack_qweight(raw, bits).to(torch.int64), bits), raw)


@pytest.mark.parametrize("bits", [3, 4])
def test_topx_extraction_preserves_the_result(bits):
    N, K, topX = 64, 128, 5
    rng = np.random.default_rng(bits)
    idx_nk = rng.integers(0, 1 << bits, size=(N, K))
    lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float32), axis=1)
    outl = np.where(rng.random((N, K)) < 0.03, rng.normal(0, 0.2, (N, K)), 0).astype(np.float32)
    outl[[3, 17, 40]] = np.where(rng.random((3, K)) < 0.6, rng.normal(0, 0.2, (3, K)), 0).astype(np.float32)  # heavy rows
    plain = pack.pack_layer(torch.from_numpy(idx_nk), torch.from_numpy(lut), bits, torch.from_numpy(outl), topX=0)
    hyb = pack.pack_layer(torch.from_numpy(idx_nk), torch.from_numpy(lut), bits, torch.from_numpy(outl), topX=topX)
    assert hyb["full_rows"].shape == (K, topX) and hyb["full_row_indices"].dtype == torch.int32
    assert {3, 17, 40} <= set(hyb["full_row_indices"].tolist())
    assert hyb["vals"].numel() < plain["vals"].numel() and int(hyb["rows"][-1]) == hyb["vals"].numel()
    x = rng.normal(size=K).astype(np.float32)
    mul = np.zeros(N, np.float32)
    npy = lambda d: {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}
    a = H.oracle_ref(npy(plain), x, mul, "spmv")
    b = H.oracle_ref(npy(hyb), x, mul, "hybrid")
    assert np.allclose(a, b, rtol=0, atol=1e-6)
    # and the dense part alone equals lut[idx] @ x
    W = lut[np.arange(N)[:, None], idx_nk]
    assert np.allclose(H.oracle_ref(npy(plain), x, mul, "dense"), W.astype(np.float64) @ x, atol=1e-9)
