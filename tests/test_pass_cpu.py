"""Planner of the dependency-gated pass (measurement library, csrc/experimental/sqllm_experimental.hip: sqllm_pass_plan),
without a GPU: the workspace image must cover every work item of every op exactly once, in group order, with
consistent gates.  (The kernel itself: tests/test_gpu_pass.py.  Reference launch structure it replaces:
squeezellm/quant_cuda_kernel.cu:157-179, :510-577.)"""
import ctypes
import os

import numpy as np
import pytest

from squeezellm_amd import _lib, build

pytestmark = pytest.mark.skipif(not os.path.exists(build.LIB_PATH), reason="libsqllm_hip.so not built")

SHAPES_7B = [(4096, 4096)] * 4 + [(4096, 11008)] * 2 + [(11008, 4096)]
GROUPS = [3, 1, 2, 1]


def _ops(bits, sparse, n_layers=2):
    ops = (_lib.SqllmOp * (7 * n_layers))()
    for li in range(n_layers):
        for j, (K, N) in enumerate(SHAPES_7B):
            o = ops[7 * li + j]
            o.bits, o.K, o.N, o.batch = bits, K, N, 0
            base = 0x100000000 * (7 * li + j + 1)
            grp = [0, 0, 0, 1, 2, 2, 3][j]
            o.vec = 0x7000000000 + 0x100000 * (4 * li + grp)  # members of a group share vec
            o.qweight, o.mul, o.lookup_table = base, base + 0x40000000, base + 0x50000000
            if sparse:
                o.rows, o.cols, o.vals, o.nnz = base + 0x60000000, base + 0x61000000, base + 0x62000000, int(0.0045 * K * N)
                o.full_rows, o.full_row_indices, o.topX = base + 0x63000000, base + 0x64000000, 10
    sizes = (ctypes.c_int32 * (4 * n_layers))(*(GROUPS * n_layers))
    return ops, sizes


@pytest.fixture(scope="module")
def xlib():
    from squeezellm_amd import experimental

    lib = experimental.load()
    assert lib.sqllm_set_option(b"cu_count", 256) == 0
    return lib


@pytest.mark.parametrize("bits,sparse", [(4, False), (4, True), (3, True)])
def test_plan_covers_every_item_once(xlib, bits, sparse):
    from squeezellm_amd import experimental

    ops, sizes = _ops(bits, sparse)
    need = xlib.sqllm_pass_workspace_bytes(ops, sizes, len(sizes))
    assert need > 0 and need % 128 == 0
    img = (ctypes.c_char * need)()
    p = experimental.SqllmPass()
    ws = 0x7F0000000000
    assert xlib.sqllm_pass_plan(ops, sizes, len(sizes), ws, need, img, ctypes.byref(p)) == 0
    assert (p.n_groups, p.n_ops, p.bits) == (len(sizes), len(ops), bits) and 0 < p.grid <= 4 * 256
    raw = np.frombuffer(img, dtype=np.uint8)
    assert not raw[:p.state_bytes].any()  # the state region starts (and is re-) zeroed
    items = raw[p.items_offset:p.items_offset + 16 * p.n_items].view(np.int32).reshape(-1, 4)
    segs32 = raw[p.segs_offset:p.segs_offset + 128 * p.n_ops].view(np.int32).reshape(-1, 32)
    segs64 = raw[p.segs_offset:p.segs_offset + 128 * p.n_ops].view(np.int64).reshape(-1, 16)
    seg, role = items[:, 0] & 0xFFFFFF, items[:, 0] >> 24
    group = segs32[:, 31]
    assert (np.diff(group[seg]) >= 0).all()  # items come in group order
    per_group = np.bincount(group[seg], minlength=len(sizes))
    kk = 8 if bits == 4 else 32
    for i in range(len(ops)):
        K, N = ops[i].K, ops[i].N
        hot = segs32[i, :16]
        assert (hot[10], hot[11]) == (K, N)
        g = group[i]
        assert hot[12] == g - 1 and hot[13] == (per_group[g - 1] if g else 0)  # gate on the group before, with its item count
        assert int(segs64[i, 4]) == ws + 4 * (16 + 128 * int(g))  # arrival shards of its own group
        mine = items[seg == i]
        dense = mine[(mine[:, 0] >> 24) == 0]
        # every (column tile, unit) exactly once
        cover = np.zeros((N // 64 + (N % 64 > 0), K // kk), np.int32)
        for _, col0, ub, ue in dense:
            assert col0 % 64 == 0 and 0 <= ub < ue <= K // kk
            cover[col0 // 64, ub:ue] += 1
        assert (cover == 1).all()
        n_csr = ((mine[:, 0] >> 24) == 1).sum()
        n_topx = ((mine[:, 0] >> 24) == 2).sum()
        if sparse:
            assert n_csr == -(-ops[i].nnz // 1024) and sorted(mine[(mine[:, 0] >> 24) == 1][:, 1]) == list(range(n_csr))
            assert n_topx == -(-K // 256)
        else:
            assert n_csr == 0 and n_topx == 0
    # within a group the sparse items are dealt first
    for g in range(len(sizes)):
        r = role[group[seg] == g]
        assert (np.diff((r == 0).astype(int)) >= 0).all()


def test_plan_rejects(xlib):
    from squeezellm_amd import experimental

    ops, sizes = _ops(4, False, n_layers=1)
    need = xlib.sqllm_pass_workspace_bytes(ops, sizes, 4)
    img = (ctypes.c_char * need)()
    p = experimental.SqllmPass()
    assert xlib.sqllm_pass_plan(ops, sizes, 4, 0x7F0000000000, need - 128, img, ctypes.byref(p)) == -9  # SQLLM_E_WORKSPACE
    assert xlib.sqllm_pass_plan(ops, sizes, 4, 0x7F0000000040, need, img, ctypes.byref(p)) == -9       # misaligned
    ops[1].batch = 2
    assert xlib.sqllm_pass_workspace_bytes(ops, sizes, 4) == -6  # SQLLM_E_BATCH
    ops[1].batch = 0
    ops[3].bits = 3
    assert xlib.sqllm_pass_plan(ops, sizes, 4, 0x7F0000000000, need, img, ctypes.byref(p)) == -8  # one bit width per pass
    ops[3].bits = 4
    ops[1].vec += 64
    assert xlib.sqllm_pass_plan(ops, sizes, 4, 0x7F0000000000, need, img, ctypes.byref(p)) == -8  # members of a group share vec


def test_product_library_has_no_pass_symbols():
    lib = _lib.load()
    for name in ("sqllm_pass_launch", "sqllm_pass_plan", "sqllm_debug_set_timeline"):
        assert not hasattr(lib, name), f"{name} belongs to the measurement library"
