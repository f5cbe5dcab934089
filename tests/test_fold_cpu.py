"""decode.fold_topx_into_csr on CPU tensors: the folded CSR is the same matrix as CSR + scattered top-X rows."""
import numpy as np
import torch

from squeezellm_amd import decode


def _dense(rows, cols, vals, N, K):
    m = np.zeros((N, K))
    for r in range(N):
        for e in range(int(rows[r]), int(rows[r + 1])):
            m[r, int(cols[e])] += float(vals[e])
    return m


def test_folded_csr_is_the_same_matrix():
    rng = np.random.default_rng(0)
    N, K, topX = 37, 64, 5
    for trial in range(6):
        mask = rng.random((N, K)) < (0.1 if trial % 3 else 0.0)  # trial 0, 3: empty CSR
        counts = mask.sum(axis=1)
        rows = np.zeros(N + 1, np.int32)
        rows[1:] = np.cumsum(counts)
        cols = np.nonzero(mask)[1].astype(np.int32)
        vals = rng.normal(size=cols.size).astype(np.float32)
        full = rng.normal(size=(K, topX)).astype(np.float32)
        full[rng.random((K, topX)) < 0.3] = 0.0
        idx = rng.choice(N, size=topX, replace=trial < 4).astype(np.int32)  # trials 4, 5 may repeat an index
        lay = dict(N=N, full_rows=torch.from_numpy(full), full_row_indices=torch.from_numpy(idx))
        if cols.size:
            lay.update(rows=torch.from_numpy(rows), cols=torch.from_numpy(cols), vals=torch.from_numpy(vals))
        r2, c2, v2 = decode.fold_topx_into_csr(lay)
        assert decode.fold_topx_into_csr(lay)[0] is r2  # cached
        ref = _dense(rows, cols, vals, N, K)
        for j in range(topX):
            ref[idx[j], :] += full[:, j]
        got = _dense(r2.numpy(), c2.numpy(), v2.numpy(), N, K)
        assert np.allclose(got, ref, atol=1e-6)
        assert r2.dtype == torch.int32 and c2.dtype == torch.int32 and int(r2[-1]) == v2.numel()
        # sorted columns inside every row, no duplicates
        for r in range(N):
            seg = c2[int(r2[r]):int(r2[r + 1])].numpy()
            assert (np.diff(seg) > 0).all()


def test_fold_cache_is_keyed_on_the_buffers_it_was_built_from():
    """the folded CSR cached in the layer dict must not leak into a column shard copied from that dict, and
    must be rebuilt after an in-place edit of the operands."""
    import torch

    from squeezellm_amd import decode, sharding, synth

    lay = synth.make_layer(256, 256, 4, sparse_frac=0.02, topX=4, heavy_rows=2, device="cpu", seed=3)
    full = decode.fold_topx_into_csr(lay)
    assert full[0].numel() == 257
    shard = sharding.shard_layer_columns(lay, 1, 2)
    sh = decode.fold_topx_into_csr(shard)
    assert sh is not full and sh[0].numel() == shard["N"] + 1 and int(sh[0][-1]) == sh[2].numel()
    assert decode.fold_topx_into_csr(lay) is full  # unchanged operands: the cached result
    lay["vals"].mul_(2.0)  # in-place edit
    again = decode.fold_topx_into_csr(lay)
    assert again is not full and not torch.equal(again[2], full[2])
