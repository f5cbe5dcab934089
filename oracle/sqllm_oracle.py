"""CPU ORACLE (test infrastructure, NOT product code) for SqueezeLLM's dense-and-sparse
LUT-quantised matvec hot path.

This module is a numpy restatement of the arithmetic that lives in the reference's
`squeezellm/quant_cuda_kernel.cu` (kernels) and `squeezellm/quant.py` (packing format and the
`QuantLinearLUT.forward` pre/post-processing).  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it; the product path (`squeezellm_amd/`) never does.

Parity pinning: the reference ships no tests / golden vectors, so the
oracle is pinned two ways:
  1. packing format  -- against the UNMODIFIED reference `QuantLinearLUT.pack2`, run in the dev
     container with a stub `quant_cuda`; vectors in `tests/golden/pack2_*.npz`
     (generator: `tests/golden/make_pack2_golden.py`).
  2. kernel outputs  -- against the UNMODIFIED reference kernels compiled with hipcc from where
     they lie (`oracle/build_ref.sh` -> `oracle/_ref/libsqllm_ref.so`) and run on an MI355X;
     vectors in `tests/golden/refkernel_*.npz` (generator: `tests/golden/make_refkernel_golden.py`).

Every function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------------------------
# packing (format spec) -- squeezellm/quant.py:171-208
# ----------------------------------------------------------------------------------------------


def pack_indices(idx_kn: np.ndarray, bits: int) -> np.ndarray:
    """Pack an integer index matrix `idx_kn` [K, N] (values < 2**bits) into the reference's
    int32 `qweight` [K // 32 * bits, N].

    Follows `QuantLinearLUT.pack2`, squeezellm/quant.py:171-208 (the weight matrix has already
    been transposed to [K, N] at quant.py:172).
      4-bit (quant.py:180-184): row r, bits [4j, 4j+4) = index of k = 8r + j.
      3-bit (quant.py:185-203): per 3 rows / 32 k's --
          row0 bits 3j (j<10) = k0..9,  bits 30-31 = low 2 bits of k10
          row1 bit 0 = bit 2 of k10,    bits 3j+1  = k11..20, bit 31 = bit 0 of k21
          row2 bits 0-1 = bits 1-2 of k21, bits 3j+2 = k22..31
    """
    idx = np.ascontiguousarray(idx_kn).astype(np.uint32)
    K, N = idx.shape
    if K % 32:
        raise ValueError("K must be a multiple of 32")
    q = np.zeros((K // 32 * bits, N), dtype=np.uint32)
    if bits == 4:
        for j in range(8):
            q |= idx[j::8] << np.uint32(4 * j)
    elif bits == 3:
        g = idx.reshape(K // 32, 32, N)
        r0 = np.zeros((K // 32, N), np.uint32)
        r1 = np.zeros_like(r0)
        r2 = np.zeros_like(r0)
        for j in range(10):
            r0 |= g[:, j] << np.uint32(3 * j)
        r0 |= g[:, 10] << np.uint32(30)  # uint32 wrap keeps the low 2 bits (quant.py:189)
        r1 |= (g[:, 10] >> np.uint32(2)) & np.uint32(1)
        for j in range(10):
            r1 |= g[:, 11 + j] << np.uint32(3 * j + 1)
        r1 |= g[:, 21] << np.uint32(31)
        r2 |= (g[:, 21] >> np.uint32(1)) & np.uint32(3)
        for j in range(10):
            r2 |= g[:, 22 + j] << np.uint32(3 * j + 2)
        q[0::3], q[1::3], q[2::3] = r0, r1, r2
    else:
        raise NotImplementedError("Only 3 and 4 bits are on the hot path (quant.py:42-43)")
    return q.view(np.int32)


# ----------------------------------------------------------------------------------------------
# unpacking -- the inverse the kernels perform
# ----------------------------------------------------------------------------------------------


def unpack_indices(qweight: np.ndarray, bits: int) -> np.ndarray:
    """int32 `qweight` [K // 32 * bits, N] -> uint8 indices [K, N], bit-exactly what the kernels
    decode.

    4-bit: quant_cuda_kernel.cu:863-877   `(tmp >> 4j) & 0xf`, k advances 8 per row.
    3-bit: quant_cuda_kernel.cu:776-825   10 direct fields, the straddlers
           `(tmp1 >> 30) | ((tmp2 << 2) & 0x4)` (:792) and `(tmp2 >> 30) | ((tmp1 << 1) & 0x6)`
           (:809, after `tmp2 >>= 1` at :793), k advances 11 + 11 + 10.
    """
    q = np.ascontiguousarray(qweight).view(np.uint32)
    R, N = q.shape
    if bits == 4:
        K = R * 8
        out = np.empty((K, N), np.uint8)
        for j in range(8):
            out[j::8] = (q >> np.uint32(4 * j)) & np.uint32(0xF)
        return out
    if bits == 3:
        if R % 3:
            raise ValueError("3-bit qweight needs a multiple of 3 rows")
        G = R // 3
        K = G * 32
        t1, t2, t3 = q[0::3], q[1::3], q[2::3]  # tmp1, tmp2, (second) tmp1 in the kernel
        out = np.empty((G, 32, N), np.uint8)
        for j in range(10):  # :779-788
            out[:, j] = (t1 >> np.uint32(3 * j)) & np.uint32(7)
        out[:, 10] = ((t1 >> np.uint32(30)) | ((t2 << np.uint32(2)) & np.uint32(4))) & np.uint32(7)  # :792
        t2s = t2 >> np.uint32(1)  # :793
        for j in range(10):  # :796-805
            out[:, 11 + j] = (t2s >> np.uint32(3 * j)) & np.uint32(7)
        out[:, 21] = ((t2s >> np.uint32(30)) | ((t3 << np.uint32(1)) & np.uint32(6))) & np.uint32(7)  # :809
        t3s = t3 >> np.uint32(2)  # :810
        for j in range(10):  # :813-822
            out[:, 22 + j] = (t3s >> np.uint32(3 * j)) & np.uint32(7)
        return out.reshape(K, N)
    raise NotImplementedError("Only 3 and 4 bits are on the hot path")


def dequantize(qweight: np.ndarray, lookup_table: np.ndarray, bits: int, dtype=np.float32) -> np.ndarray:
    """Dense weight matrix W [K, N] with W[k, n] = lookup_table[n, idx[k, n]].

    LUT indexing `lookup_table[col * 2**bits + val]`: quant_cuda_kernel.cu:759-762 (3-bit),
    :849-852 (4-bit)."""
    idx = unpack_indices(qweight, bits)  # [K, N]
    lut = np.asarray(lookup_table)
    N = idx.shape[1]
    if lut.shape != (N, 1 << bits):
        raise ValueError(f"lookup_table must be [N, {1 << bits}], got {lut.shape}")
    W = lut[np.arange(N)[None, :], idx]
    return W.astype(dtype, copy=False)


# ----------------------------------------------------------------------------------------------
# the three terms of the op
# ----------------------------------------------------------------------------------------------


def dense_term(x2d: np.ndarray, qweight, lookup_table, bits: int, acc=np.float64) -> np.ndarray:
    """sum_k LUT[n][idx(k, n)] * vec[b, k]  -> [B, N].
    quant_cuda_kernel.cu:831-880 (w4), :741-828 (w3); batched indexing :923/:977, :1017/:1036."""
    W = dequantize(qweight, lookup_table, bits, dtype=acc)
    return np.asarray(x2d, dtype=acc) @ W


def csr_term(x2d: np.ndarray, rows, cols, vals, num_rows: int, acc=np.float64) -> np.ndarray:
    """CSR SpMV: out[b, r] = sum_{i in [rows[r], rows[r+1])} vals[i] * vec[b, cols[i]].
    SPMV_ATOMIC quant_cuda_kernel.cu:1049-1058; batched :1072-1088."""
    x2d = np.asarray(x2d, dtype=acc)
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    vals = np.asarray(vals, dtype=acc)
    B = x2d.shape[0]
    out = np.zeros((B, num_rows), dtype=acc)
    nnz = int(rows[num_rows]) if len(rows) > num_rows else 0
    if nnz == 0:
        return out
    counts = np.diff(rows[: num_rows + 1])
    rid = np.repeat(np.arange(num_rows), counts)
    start = int(rows[0])
    prod = x2d[:, cols[start : start + len(rid)]] * vals[start : start + len(rid)][None, :]
    for b in range(B):
        out[b] = np.bincount(rid, weights=prod[b], minlength=num_rows)
    return out


def topx_term(x2d: np.ndarray, full_rows, full_row_indices, n_out: int, acc=np.float64) -> np.ndarray:
    """"top-X full rows": out[b, full_row_indices[c]] += sum_k full_rows[k, c] * vec[b, k].
    DenseMatVecKernel quant_cuda_kernel.cu:1101-1121; batched :1139-1162.  Duplicate indices
    accumulate (each column does its own atomicAdd, :1120-1121)."""
    x2d = np.asarray(x2d, dtype=acc)
    fr = np.asarray(full_rows, dtype=acc)  # [K, topX]
    idx = np.asarray(full_row_indices, dtype=np.int64)
    out = np.zeros((x2d.shape[0], n_out), dtype=acc)
    if fr.size == 0:
        return out
    contrib = x2d @ fr  # [B, topX]
    for c in range(fr.shape[1]):
        out[:, idx[c]] += contrib[:, c]
    return out


def matvec(
    vec,
    qweight,
    mul,
    lookup_table,
    bits: int,
    rows=None,
    cols=None,
    vals=None,
    full_rows=None,
    full_row_indices=None,
    acc=np.float64,
):
    """The whole op with the reference's accumulate-into-`mul` semantics
    (`mul` arrives holding bias or zeros: squeezellm/quant.py:214-219, :316-318).

    `vec` is [K] (matvec ops) or [B, K] (`*_batched` ops); `mul` is [N] / [B, N].
    Returns a NEW array mul + dense (+ csr) (+ top-X) in `acc` precision (the reference updates
    `mul` in place with fp32 atomics, whose summation order is unspecified; the oracle returns
    the exactly-rounded target instead)."""
    vec = np.asarray(vec)
    mul = np.asarray(mul)
    batched = vec.ndim == 2 and mul.ndim == 2
    x2d = vec.reshape(1, -1) if not batched else vec
    N = np.asarray(qweight).shape[1]
    out = np.asarray(mul, dtype=acc).reshape(x2d.shape[0], N).copy()
    out += dense_term(x2d, qweight, lookup_table, bits, acc)
    if rows is not None:
        out += csr_term(x2d, rows, cols, vals, N, acc)
    if full_rows is not None:
        out += topx_term(x2d, full_rows, full_row_indices, N, acc)
    return out if batched else out.reshape(mul.shape)


def matvec_fp16_path(vec, qweight, mul, lookup_table, bits: int, **sparse):
    """The "reference fp16 dequant-then-matmul path" BASELINE.json's tolerance is quoted against:
    W dequantised to fp16, x in fp16, products accumulated in fp32, sparse terms in fp32."""
    vec = np.asarray(vec)
    batched = vec.ndim == 2
    x2d = (vec if batched else vec.reshape(1, -1)).astype(np.float16).astype(np.float32)
    W = dequantize(qweight, lookup_table, bits, dtype=np.float16).astype(np.float32)
    N = W.shape[1]
    out = np.asarray(mul, dtype=np.float32).reshape(x2d.shape[0], N) + x2d @ W
    if sparse.get("rows") is not None:
        out = out + csr_term(x2d, sparse["rows"], sparse["cols"], sparse["vals"], N, np.float32)
    if sparse.get("full_rows") is not None:
        out = out + topx_term(x2d, sparse["full_rows"], sparse["full_row_indices"], N, np.float32)
    return out if batched else out.reshape(np.asarray(mul).shape)


# ----------------------------------------------------------------------------------------------
# QuantLinearLUT.forward pre/post-processing -- squeezellm/quant.py:211-383
# ----------------------------------------------------------------------------------------------


def quantlinear_forward(x, layer: dict, acc=np.float64):
    """Restates `QuantLinearLUT.forward` around the op.

    matvec branch (x.shape[-1] == x.numel(), quant.py:212): y = bias.clone() or zeros (:214-219),
    x.float() (:223/:267), op, y.to(x.dtype).reshape(outshape) (:311-312).
    batched branch (:313-383): x.reshape(-1, K), out = zeros [B, N] fp32, op, out.to(dtype),
    reshape, + bias AFTER the cast (:380-383; the result takes the promoted dtype of out and bias).

    `layer` keys: bits, qweight, lookup_table, bias (or None), and optionally rows/cols/vals,
    full_rows/full_row_indices.  Op selection order hybrid -> spmv -> dense follows :224-265.
    """
    x = np.asarray(x)
    dtype = x.dtype
    N = layer["qweight"].shape[1]
    sparse = {}
    if layer.get("rows") is not None:
        sparse.update(rows=layer["rows"], cols=layer["cols"], vals=layer["vals"])
        if layer.get("full_rows") is not None:
            sparse.update(full_rows=layer["full_rows"], full_row_indices=layer["full_row_indices"])
    bias = layer.get("bias")
    if x.shape[-1] == x.size:
        y0 = np.zeros(N, np.float32) if bias is None else np.asarray(bias, np.float32).copy()
        y = matvec(x.reshape(-1).astype(np.float32), layer["qweight"], y0, layer["lookup_table"], layer["bits"], acc=acc, **sparse)
        return y.astype(dtype).reshape(x.shape[:-1] + (N,))
    x2 = x.reshape(-1, x.shape[-1]).astype(np.float32)
    out = matvec(x2, layer["qweight"], np.zeros((x2.shape[0], N), np.float32), layer["lookup_table"], layer["bits"], acc=acc, **sparse)
    out = out.astype(dtype).reshape(x.shape[:-1] + (N,))
    if bias is not None:
        # `out + self.bias` (quant.py:382): a plain add, so an fp32 bias buffer promotes an fp16 `out`
        # to fp32 -- torch's and numpy's type promotion agree here
        out = out + np.asarray(bias)
    return out


# ----------------------------------------------------------------------------------------------
# sparse-value convention of pack2 -- squeezellm/quant.py:117-131
# ----------------------------------------------------------------------------------------------


def outliers_to_csr(outliers_nk: np.ndarray, lookup_table: np.ndarray):
    """Dense outlier matrix [N, K] (0 = not an outlier) -> (rows, cols, vals) exactly as pack2
    stores them: each outlier value MINUS the centroid nearest to zero of its channel
    (`round_to_nearest_pole_sim(zeros(1), centroid)`, quant.py:8-24 / :117-123; the dense part
    holds that centroid's index at outlier positions), then CSR over output channels with int32
    crow/col and fp32 values (quant.py:126-131).  Ties in |centroid| resolve to the lowest index
    (torch.argmin), as in quant.py:20."""
    out = np.array(outliers_nk, dtype=np.float32, copy=True)
    lut = np.asarray(lookup_table, dtype=np.float32)
    N, K = out.shape
    zero_map = lut[np.arange(N), np.argmin(np.abs(lut), axis=1)]
    nzmask = out != 0
    out = np.where(nzmask, out - zero_map[:, None], out)
    # to_sparse_csr keeps entries that are non-zero AFTER the subtraction (quant.py:126)
    keep = out != 0
    counts = keep.sum(axis=1)
    rows = np.zeros(N + 1, np.int32)
    rows[1:] = np.cumsum(counts)
    r, c = np.nonzero(keep)
    return rows, c.astype(np.int32), out[r, c].astype(np.float32)


def startrows_balanced(rows: np.ndarray, outfeatures: int, numvals: int, num_nonzero_per_thread: int = 10):
    """`startrows` of the `balanced` packing, quant.py:139-169 (loop restated literally)."""
    import math

    num_threads = int((numvals + num_nonzero_per_thread - 1) / num_nonzero_per_thread)
    num_threads = 128 * math.ceil(num_threads / 128)
    nnz_per_thread = int((numvals + num_threads - 1) / num_threads)
    start_rows = np.zeros(num_threads, np.int32)
    minidx = 0
    for i in range(num_threads):
        tmpmin = minidx
        for j in range(minidx, outfeatures):
            if nnz_per_thread * i > numvals:
                start_rows[i] = -1
                break
            elif rows[j] < nnz_per_thread * i:
                start_rows[i] = j
                tmpmin = j
            else:
                break
        minidx = tmpmin
    return start_rows, num_threads, nnz_per_thread
