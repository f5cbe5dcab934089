"""CPU tests of the drop-in boundary (no GPU, no compute calls): the C-ABI library loads, exports
every symbol include/sqllm_hip.h declares, plans launches, rejects bad arguments before touching
the device, and the Python operator module refuses CPU tensors instead of falling back."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers as H

HEADER = os.path.join(H.ROOT, "include", "sqllm_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sqllm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from squeezellm_amd import _lib

    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 22
    for name in syms:
        assert hasattr(lib, name), f"{name} declared in sqllm_hip.h but not exported"
    assert set(syms) == set(_lib.SIGNATURES), "ctypes prototypes out of sync with the header"
    assert lib.sqllm_abi_version() == 1
    # the 12 reference names (quant_cuda.cpp:257-270) + the 2 balanced ones
    for b in (3, 4):
        for kind in ("", "_spmv", "_spmv_hybrid"):
            for sfx in ("", "_batched"):
                assert f"sqllm_vecquant{b}matmul{kind}_nuq_perchannel{sfx}" in syms
        assert f"sqllm_vecquant{b}matmul_spmv_balanced_nuq_perchannel" in syms


def test_header_compiles_as_plain_c(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "sqllm_hip.h"\nint main(void){ sqllm_op op; (void)op; return SQLLM_ABI_VERSION - 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", f"-I{os.path.dirname(HEADER)}", str(c), "-o", str(tmp_path / "t")], check=True)
    assert subprocess.run([str(tmp_path / "t")]).returncode == 0


def test_python_operator_module_mirrors_reference_names():
    import quant_cuda  # the top-level shim squeezellm/quant.py imports
    from squeezellm_amd import quant_cuda as impl

    names = [f"vecquant{b}matmul{k}_nuq_perchannel{s}" for b in (3, 4) for k in ("", "_spmv", "_spmv_hybrid") for s in ("", "_batched")]
    names += [f"vecquant{b}matmul_spmv_balanced_nuq_perchannel" for b in (3, 4)]
    for n in names:
        assert callable(getattr(quant_cuda, n)) and getattr(quant_cuda, n) is getattr(impl, n)
    assert sorted(impl.__all__) == sorted(names)


def test_plan_geometry():
    from squeezellm_amd import _lib

    _lib.set_option("cu_count", 256)
    p = _lib.plan_query(4, 4096, 4096)
    assert p["col_tiles"] == 64 and p["dense_blocks"] == p["col_tiles"] * p["k_slices"]
    assert p["dense_blocks"] == 512   # a small op alone in its launch: two workgroups per CU
    assert _lib.plan_query(4, 11008, 4096, nnz=202_899, topX=10)["dense_blocks"] == 960   # a large one alone (fused kernel: it has sparse terms): ~4 per CU
    # ... dense-only, tall and >= 16 MB it takes the column-lane kernel at batch 1 too (round 6): ranges of the flattened (tile, unit) space, ~3 per CU
    assert _lib.plan_query(4, 11008, 4096)["dense_blocks"] == 768
    assert p["csr_blocks"] == 0 and p["topx_blocks"] == 0 and p["grid_y"] == 1
    assert p["groups_per_wave"] % 32 == 0  # a K slice is whole workgroup steps (8 waves x 4 units)
    # every unit is covered exactly once
    assert p["k_slices"] * p["groups_per_wave"] >= 4096 // 8 > (p["k_slices"] - 1) * p["groups_per_wave"]
    p3 = _lib.plan_query(3, 11008, 4096, nnz=202_899, topX=10)
    assert p3["csr_blocks"] == -(-202_899 // 1024) and p3["topx_blocks"] == 11008 // 256  # slabs of 256 rows
    assert p3["grid_x"] >= p3["dense_blocks"] + p3["csr_blocks"] + p3["topx_blocks"]
    assert (p3["grid_x"] - p3["dense_blocks"]) % 8 == 0  # dense ids stay XCD-aligned
    assert _lib.plan_query(4, 4096, 4096, batch=3)["grid_y"] == 1
    # the column-lane kernel (2 .. 4 rows 4-bit, 2 .. 16 rows 3-bit): equal ranges of the flattened (tile, unit) space,
    # ~3 per CU.  With the routing options at their defaults (0) a small single 4-bit op stays on the batch tiles (round
    # 3: they fit three workgroups per CU and measure faster), large ops take the column-lane kernel
    assert _lib.get_option("cols_min_batch") == 0 and _lib.get_option("cols_max_batch") == 0
    pt = _lib.plan_query(4, 5120, 5120, batch=4)
    # batch tiles: column tiles x K slices; an op of > 12 MB alone on a tile that holds three workgroups per CU is planned as ONE resident round (2.5 per CU, round 6)
    assert pt["dense_blocks"] == pt["col_tiles"] * pt["k_slices"] and pt["dense_blocks"] == 560
    assert _lib.plan_query(4, 5120, 5120, batch=2)["dense_blocks"] == 800 == _lib.plan_query(4, 5120, 5120, batch=0)["dense_blocks"]  # four per CU there
    pc = _lib.plan_query(3, 5120, 13824, batch=4)
    total = pc["col_tiles"] * (5120 // 32)
    assert pc["grid_y"] == 1 and pc["groups_per_wave"] % 8 == 0 and 700 <= pc["dense_blocks"] <= 768
    assert pc["dense_blocks"] * pc["groups_per_wave"] >= total > (pc["dense_blocks"] - 1) * pc["groups_per_wave"]
    assert _lib.plan_query(3, 4096, 4096, batch=4)["dense_blocks"] == 64 * _lib.plan_query(3, 4096, 4096, batch=4)["k_slices"]  # small: tiles
    _lib.set_option("cols_max_batch", 4)  # an explicit range is taken at its word, whatever the shape
    p4 = _lib.plan_query(4, 5120, 5120, batch=4)
    assert 700 <= p4["dense_blocks"] <= 768  # (80 tiles x 9 tile-aligned ranges)
    _lib.set_option("cols_max_batch", 0)
    _lib.set_option("cols_min_batch", 1 << 30)  # switched off: the batch tiles of the batch-1 kernel
    assert _lib.plan_query(3, 5120, 13824, batch=4)["dense_blocks"] == pc["col_tiles"] * _lib.plan_query(3, 5120, 13824, batch=4)["k_slices"]
    _lib.set_option("cols_min_batch", 0)
    # batches from `mfma_min_batch` (4-bit 7 -- 9 for an op of <= 16 MB alone in its launch --, 3-bit 9) rows up take the matrix-core
    # kernel: passes of 16 / 32 / 64 rows; below, batch tiles of exactly 1, 2, 3, 4, 5, 6 or 8 rows (round 6: 3, 5, 6)
    assert _lib.get_option("mfma_min_batch") == 0  # = the measured default
    assert _lib.plan_query(4, 4096, 11008, batch=8)["grid_y"] == 1 and _lib.plan_query(3, 4096, 4096, batch=16)["grid_y"] == 1  # (3-bit: matrix cores from 9 rows since round 4)
    for b, tiles in ((3, 3), (5, 5), (6, 6), (7, 7)):  # (batch, rows per pass) on the batch tiles: one pass, K slices per column tile
        pt = _lib.plan_query(4, 4096, 4096, batch=b)
        assert pt["grid_y"] == 1 and pt["dense_blocks"] == pt["col_tiles"] * pt["k_slices"], (b, pt)
    assert _lib.plan_query(4, 4096, 4096, batch=8)["dense_blocks"] == 64 * _lib.plan_query(4, 4096, 4096, batch=8)["k_slices"]  # 8.4 MB alone: 8-row tile
    assert _lib.plan_query(4, 4096, 8192, batch=6)["dense_blocks"] == 128 * _lib.plan_query(4, 4096, 8192, batch=6)["k_slices"]  # 6 rows, 16.8 MB: the 6-row tile
    # (a single op of >= 20 MB takes the column-lane kernel up to 6 rows: ranges of the flattened (tile, unit) space, three workgroups per CU)
    assert _lib.plan_query(4, 4096, 11008, batch=6)["grid_y"] == 1 and _lib.plan_query(4, 4096, 11008, batch=6)["dense_blocks"] <= 3 * 256
    assert _lib.plan_query(3, 4096, 4096, batch=17)["grid_y"] == 1
    assert _lib.plan_query(4, 4096, 4096, batch=9)["grid_y"] == 1
    assert _lib.plan_query(4, 4096, 4096, batch=33)["grid_y"] == 1
    assert _lib.plan_query(4, 4096, 4096, batch=320)["grid_y"] == 5  # batch * K * N = 5.4e9: still the tile kernel, passes of 64 rows
    # from batch * K * N = 5.7e9 (4-bit; 3-bit 4e9) the WIDE form: a 1-D grid over units of 64 rows x 8 column tiles, whole rounds of one unit per
    # CU over all of K, the last partial round cut into K slices (csrc/sqllm_capi.hip: takes_wide_path, make_plan_wide)
    assert _lib.plan_query(4, 4096, 4096, batch=340)["grid_y"] == 1 and _lib.plan_query(3, 4096, 4096, batch=240)["grid_y"] == 1
    pq = _lib.plan_query(4, 4096, 4096, batch=2048)  # 32 row blocks x 8 column groups = 256 units: exactly one round
    assert pq["grid_y"] == 1 and pq["dense_blocks"] == 256 and pq["k_slices"] == 1
    pm = _lib.plan_query(4, 5120, 13824, batch=16)
    # ... whose dense work is the flattened (column tile, unit) space in equal ranges, one round of workgroups
    assert pm["groups_per_wave"] % 32 == 0 and pm["dense_blocks"] <= 512
    # the fused small launch (matrix cores up to 16 rows): the CSR term is walked by the dense workgroups themselves
    # (csr_tile_fold: one wave of each) -- no chunk workgroups in the grid, only the top-X slabs; batch 1, the batch tiles
    # and 17+ rows keep the chunk role
    # (planned as launched with a workspace: vec transposed, the top-X slabs shared by few workgroups of several slabs each --
    # an op alone in its launch: up to 24, or 16 from 25 slabs; in a group 8 / 16 per op -- and the dense ranges cut for the
    # slots they leave: ONE round of workgroups in all)
    for b in (7, 8, 16):
        pf = _lib.plan_query(4, 5120, 13824, nnz=330_000, topX=10, batch=b)
        assert pf["csr_blocks"] == 0 and pf["topx_blocks"] == 20 and pf["grid_x"] == 24 + pf["dense_blocks"] <= 512, (b, pf)
    assert _lib.plan_query(4, 13824, 5120, nnz=330_000, topX=10, batch=8)["topx_blocks"] == 16
    assert _lib.plan_query(4, 5120, 13824, nnz=330_000, topX=10, batch=17)["csr_blocks"] == -(-330_000 // 1024)
    # batch 1, a launch of more workgroups than the chip holds with >= 1.25 x CUs of them sparse: CSR chunks of 2048 non-zeros (round 6) ...
    assert _lib.plan_query(4, 5120, 13824, nnz=330_000, topX=10, batch=1)["csr_blocks"] == -(-330_000 // 2048)
    # ... a launch that is resident at once keeps 1024 (7B o_proj: 512 dense + 74 + 16), and so does one with few sparse workgroups
    assert _lib.plan_query(4, 4096, 4096, nnz=75_000, topX=10)["csr_blocks"] == 74
    assert _lib.plan_query(4, 11008, 4096, nnz=203_000, topX=10)["csr_blocks"] == 199  # (7B down_proj: 1202 workgroups, 242 of them sparse < 320)
    assert _lib.plan_query(4, 5120, 5120, nnz=128_000, topX=10, batch=2)["csr_blocks"] == 125
    total = pm["col_tiles"] * (5120 // 8)
    assert pm["dense_blocks"] * pm["groups_per_wave"] >= total > (pm["dense_blocks"] - 1) * pm["groups_per_wave"]
    pw = _lib.plan_query(4, 5120, 13824, batch=2048)  # 32 x 27 = 864 units: 3 whole rounds + 96 units in two K slices of 320
    assert pw["grid_y"] == 1 and pw["k_slices"] == 2 and pw["groups_per_wave"] == 320 and pw["dense_blocks"] == 768 + 2 * 96
    _lib.set_option("mfma_wide_min_batch", 1 << 30)  # ... unless switched off: 32 passes of 64 rows share the chip, 256 / 32 ranges
    pw = _lib.plan_query(4, 5120, 13824, batch=2048)
    assert pw["grid_y"] == 32 and pw["dense_blocks"] == 8
    _lib.set_option("mfma_wide_min_batch", 0)
    _lib.set_option("mfma_min_batch", 1 << 30)  # ... unless switched off: batch tiles of 8
    assert _lib.plan_query(4, 4096, 4096, batch=9)["grid_y"] == 2
    _lib.set_option("mfma_min_batch", 0)
    # options round-trip and steer the plan
    _lib.set_option("target_wgs", 1024)
    assert _lib.get_option("target_wgs") == 1024
    assert _lib.plan_query(4, 4096, 4096)["dense_blocks"] == 1024
    _lib.set_option("target_wgs", 0)
    with pytest.raises(ValueError):
        _lib.set_option("no_such_option", 1)
    with pytest.raises(ValueError):
        _lib.plan_query(5, 4096, 4096)
    with pytest.raises(ValueError):
        _lib.plan_query(4, 4100, 4096)  # K % 32
    with pytest.raises(ValueError):
        _lib.plan_query(4, 4096, 4098)  # N % 4


def test_launch_rejects_bad_arguments_before_touching_the_device():
    from squeezellm_amd import _lib

    lib = _lib.load()
    op = _lib.SqllmOp(bits=4, batch=0, K=128, N=128)
    assert lib.sqllm_launch(ctypes.byref(op), None) == -3  # SQLLM_E_NULL: no operand pointers
    op.vec = op.qweight = op.mul = op.lookup_table = 16
    op.bits = 5
    assert lib.sqllm_launch(ctypes.byref(op), None) == -1  # SQLLM_E_BITS
    op.bits, op.K = 4, 100
    assert lib.sqllm_launch(ctypes.byref(op), None) == -2  # SQLLM_E_SHAPE
    op.K, op.qweight = 128, 20
    assert lib.sqllm_launch(ctypes.byref(op), None) == -4  # SQLLM_E_ALIGN
    op.qweight, op.batch = 16, -1
    assert lib.sqllm_launch(ctypes.byref(op), None) == -6  # SQLLM_E_BATCH
    # named entry points: height must be K/32*bits, num_rows must equal N
    f = lib.sqllm_vecquant3matmul_nuq_perchannel
    assert f(16, 16, 16, 16, 16, 128, None) == -2  # 16 rows is not a multiple of 3
    g = lib.sqllm_vecquant4matmul_spmv_nuq_perchannel
    assert g(16, 16, 16, 16, 16, 64, 16, 16, 16, 128, 10, None) == -5  # num_rows != width
    assert lib.sqllm_vecquant4matmul_nuq_perchannel_batched(16, 16, 16, 16, 16, 128, 0, 128, None) == -6
    assert lib.sqllm_vecquant4matmul_nuq_perchannel_batched(16, 16, 16, 16, 16, 128, 2, 64, None) == -6
    assert b"bits" in lib.sqllm_error_string(-1) and lib.sqllm_error_string(0) == b"ok"
    n_done = ctypes.c_int32(-1)
    assert lib.sqllm_launch_sequence(None, 0, None, ctypes.byref(n_done)) == 0 and n_done.value == 0


def test_linear_entry_points_validate_and_size_their_workspace():
    from squeezellm_amd import _lib

    lib = _lib.load()
    # 64-bit accumulator words [batch, N]
    assert _lib.linear_workspace_bytes(4096, 0) == 8 * 4096
    assert _lib.linear_workspace_bytes(456, 5) == 8 * 5 * 456
    lin = _lib.SqllmLinear()
    lin.op.bits, lin.op.K, lin.op.N = 4, 128, 128
    lin.op.vec = lin.op.qweight = lin.op.mul = lin.op.lookup_table = 16
    assert lib.sqllm_linear_f16(None, None) == -3
    assert lib.sqllm_linear_f16(ctypes.byref(lin), None) == -3  # no workspace
    lin.workspace = 8
    assert lib.sqllm_linear_f16(ctypes.byref(lin), None) == -4  # workspace not 16-byte aligned
    lin.workspace, lin.op.bits = 16, 2
    assert lib.sqllm_linear_f16(ctypes.byref(lin), None) == -1
    sizes = (ctypes.c_int32 * 1)(0)
    done = ctypes.c_int32(-1)
    assert lib.sqllm_linear_f16_groups(ctypes.byref(lin), sizes, 1, None, ctypes.byref(done)) == -8 and done.value == 0
    assert lib.sqllm_linear_f16_groups(None, None, 0, None, None) == 0
    # the operator entry points refuse matrices their 32-bit offsets cannot address
    op = _lib.SqllmOp(bits=4, batch=0, K=65536, N=131072)
    op.vec = op.qweight = op.mul = op.lookup_table = 16
    assert lib.sqllm_launch(ctypes.byref(op), None) == -2


def test_no_cpu_fallback_in_the_operator_module():
    import torch

    from squeezellm_amd import quant_cuda as qc

    case = H.make_case(4, 128, 128, seed=0)
    t = H.to_torch(case, "cpu")
    x, y = torch.randn(128), torch.zeros(128)
    with pytest.raises(RuntimeError, match="no CPU path"):
        qc.vecquant4matmul_nuq_perchannel(x, t["qweight"], y, t["lookup_table"])
    with pytest.raises(TypeError):
        qc.vecquant4matmul_nuq_perchannel(x.double(), t["qweight"], y, t["lookup_table"])
    with pytest.raises(ValueError):
        qc.vecquant4matmul_nuq_perchannel(x, t["qweight"][:, :64], y, t["lookup_table"])


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under squeezellm_amd/ (nor the quant_cuda shim)
    may import or reference it."""
    pkg = os.path.join(H.ROOT, "squeezellm_amd")
    files = [os.path.join(dp, f) for dp, _, fs in os.walk(pkg) for f in fs if f.endswith((".py", ".hip", ".h"))]
    files.append(os.path.join(H.ROOT, "quant_cuda.py"))
    for f in files:
        src = open(f).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
        assert "sqllm_oracle" not in src and "libsqllm_ref" not in src, f


def test_quantlinear_module_schema_matches_reference():
    import torch

    from squeezellm_amd.quant import QuantLinearLUT

    m = QuantLinearLUT(3, 256, 128, True, include_sparse=True, numvals=77, topX=10, balanced=True)
    sd = m.state_dict()
    assert sd["qweight"].shape == (256 // 32 * 3, 128) and sd["qweight"].dtype == torch.int32
    assert sd["lookup_table"].shape == (128, 8) and sd["bias"].shape == (128,)
    assert sd["rows"].shape == (129,) and sd["cols"].shape == (77,) and sd["vals"].dtype == torch.float32
    assert sd["full_rows"].shape == (256, 10) and sd["full_row_indices"].dtype == torch.int32
    assert sd["startrows"].shape == (128,)  # ceil(77/10) -> 8 -> rounded up to 128 threads
    assert m.op_kind(False) == "spmv_hybrid" and QuantLinearLUT(4, 64, 64, False).op_kind(True) == "dense"
    assert QuantLinearLUT(4, 64, 64, False, include_sparse=True, numvals=3, balanced=True).op_kind(False) == "spmv_balanced"
    assert QuantLinearLUT(4, 64, 64, False, include_sparse=True, numvals=3, balanced=True).op_kind(True) == "spmv"
    with pytest.raises(NotImplementedError):
        QuantLinearLUT(2, 64, 64, False)


@pytest.mark.skipif(not os.path.exists("/root/reference/squeezellm/quant.py"), reason="reference tree only exists in the dev container")
def test_unmodified_reference_quant_py_runs_on_this_quant_cuda():
    """`import quant_cuda` at squeezellm/quant.py:5 must resolve to this repo's module, and the
    reference's own QuantLinearLUT.forward must reach our operators with its own argument order
    (checked with a recording double; the real call needs a GPU and is covered by -m gpu)."""
    code = r'''
import sys, contextlib, io
sys.path.insert(0, %r); sys.path.insert(1, "/root/reference")
import torch, quant_cuda
import squeezellm_amd.quant_cuda as impl
assert quant_cuda.vecquant4matmul_nuq_perchannel is impl.vecquant4matmul_nuq_perchannel
from squeezellm import quant as refquant          # the UNMODIFIED reference module
assert refquant.quant_cuda is quant_cuda
calls = []
for name in impl.__all__:
    setattr(quant_cuda, name, (lambda n: (lambda *a: calls.append((n, len(a)))))(name))
torch.zeros_orig = torch.zeros
torch.zeros = lambda *a, **k: torch.zeros_orig(*a, **{**k, "device": "cpu"})   # quant.py:218 hard-codes "cuda"
with contextlib.redirect_stdout(io.StringIO()):
    dense = refquant.QuantLinearLUT(4, 128, 64, False)
    hyb = refquant.QuantLinearLUT(3, 128, 64, False, include_sparse=True, numvals=5, topX=10)
    bal = refquant.QuantLinearLUT(4, 128, 64, False, include_sparse=True, numvals=5, balanced=True)
dense(torch.zeros(1, 1, 128)); dense(torch.zeros(3, 128)); hyb(torch.zeros(1, 128)); hyb(torch.zeros(2, 128)); bal(torch.zeros(128))
assert calls == [("vecquant4matmul_nuq_perchannel", 4), ("vecquant4matmul_nuq_perchannel_batched", 4),
                 ("vecquant3matmul_spmv_hybrid_nuq_perchannel", 10), ("vecquant3matmul_spmv_hybrid_nuq_perchannel_batched", 10),
                 ("vecquant4matmul_spmv_balanced_nuq_perchannel", 11)], calls
print("OK")
''' % H.ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stderr[-2000:]


def test_every_documented_option_round_trips():
    """Each option name the header documents is accepted by set / get (per device: slot 0 without a GPU),
    and the defaults are the documented ones."""
    import re

    from squeezellm_amd import _lib

    hdr = open(HEADER).read()
    block = hdr[hdr.index("Launch-geometry knobs"):hdr.index("int sqllm_set_option")]
    names = sorted(set(re.findall(r'"([a-z_]+)"', block)))
    assert {"target_wgs", "groups_per_wave", "sparse_last", "cu_count", "mfma_min_batch", "cols_min_batch", "cols_max_batch",
            "sparse_transpose", "scratch_in_capture", "validate_csr"} <= set(names)
    defaults = {"mfma_min_batch": 0, "cols_min_batch": 0, "cols_max_batch": 0, "sparse_transpose": 1, "scratch_in_capture": 1,
                "validate_csr": 0, "sparse_last": 0, "target_wgs": 0, "groups_per_wave": 0, "cols_groups": 1}
    for n in names:
        before = _lib.get_option(n)
        if n in defaults:
            assert before == defaults[n], (n, before)
        _lib.set_option(n, 1)
        assert _lib.get_option(n) == 1, n
        _lib.set_option(n, before)
        assert _lib.get_option(n) == before, n


def test_option_values_are_range_checked():
    """No option value switches the product library to anything but a documented route (`sparse_transpose = 2`
    was stored as it came and made the fused small launch read an unwritten workspace): switches take 0 / 1 only, counts
    their documented range, and a rejected value leaves the stored one untouched."""
    from squeezellm_amd import _lib

    lib = _lib.load()
    switches = ["sparse_last", "cols_groups", "sparse_transpose", "scratch_in_capture", "validate_csr", "mfma_split", "mfma_fuse_small",
                "mfma_fuse_sparse", "scratch_pool_threshold", "small_reserve_topx", "small_planes"]
    for n in switches:
        before = _lib.get_option(n)
        for bad in (2, 3, 1 << 30, -1):
            assert lib.sqllm_set_option(n.encode(), bad) == -7, (n, bad)  # SQLLM_E_OPTION
        assert _lib.get_option(n) == before, n
    for n, top in (("small_wgs_per_cu", 8), ("cu_count", 1 << 16), ("target_wgs", 1 << 24), ("groups_per_wave", 1 << 24)):
        before = _lib.get_option(n)
        assert lib.sqllm_set_option(n.encode(), top + 1) == -7 and lib.sqllm_set_option(n.encode(), -1) == -7, n
        assert lib.sqllm_set_option(n.encode(), top) == 0
        _lib.set_option(n, before)
    for n in ("mfma_min_batch", "cols_min_batch", "cols_max_batch", "split_planes_min_batch", "mfma_wide_min_batch"):
        assert lib.sqllm_set_option(n.encode(), 0x7fffffff) == 0 and lib.sqllm_set_option(n.encode(), -1) == -7, n
        _lib.set_option(n, 0)


def test_product_library_carries_no_measurement_variants():
    """The ablation kernel instantiations, timeline probes, calibration kernels and the measured-and-not-adopted
    kernels (streaming, column-pair tables) live in libsqllm_hip_ablation.so only; variant switches cannot be
    compiled into the product (squeezellm_amd/build.py, csrc/sqllm_kernels.h)."""
    import re
    import subprocess

    from squeezellm_amd import build as B

    syms = subprocess.run(["nm", "-D", "--defined-only", B.LIB_PATH], check=True, capture_output=True, text=True).stdout
    for needle in ("sqllm_calib", "sqllm_debug_", "stream_matvec", "pair4_matvec"):
        assert needle not in syms, needle
    # sqllm_fused_matvec<BITS, BT, WAVES, ABL, LIN>: only ABL == 0 instantiations
    abl = set(re.findall(r"sqllm_fused_matvecILi\dELi\dELi\d+ELi(\d+)E", syms))
    assert abl == {"0"}, abl
    # a variant switch without the measurement guard does not compile
    import shutil

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if os.path.exists(hipcc):
        r = subprocess.run([hipcc, f"--offload-arch={B.ARCH}", "-std=c++17", "-DSQLLM_PAIR3=0", f"-I{B.INCLUDE}", f"-I{B.CSRC}",
                            "-fsyntax-only", "--cuda-device-only", os.path.join(B.CSRC, "sqllm_kernels.hip")], capture_output=True, text=True)
        assert r.returncode != 0 and "SQLLM_ABLATION_BUILD" in r.stderr
    with pytest.raises(ValueError):
        B.build_ablation(out=B.LIB_PATH)


def test_range_plans_cover_every_unit_exactly_once():
    """The column-lane and matrix-core kernels cut the flattened (column tile, unit) space into ranges: contiguous
    ones (the last may be short) or -- where a whole number per tile fits -- tile-aligned ones (the last of every
    tile may be short; the kernels tell the two apart by dense_blocks == col_tiles * k_slices).  Either way the
    ranges must cover all units and no range may start beyond the end."""
    from squeezellm_amd import _lib

    _lib.set_option("cu_count", 256)
    try:
        _lib.set_option("cols_min_batch", 1)
        _lib.set_option("cols_max_batch", 16)  # explicit: every shape below 17 rows plans the column-lane kernel
        seen_aligned = seen_flat = 0
        for bits in (3, 4):
            for K, N in ((4096, 4096), (5120, 5120), (5120, 13824), (13824, 5120), (11008, 4096), (4096, 11008), (8192, 8192),
                         (8192, 22016), (22016, 8192), (1024, 776), (96, 68), (32, 4)):
                if K % 32:
                    continue
                for batch in (2, 4, 8, 16, 17, 64, 300, 2048):  # 17+ (and 4-bit 9+ without the options): matrix cores
                    p = _lib.plan_query(bits, K, N, batch=batch)
                    U = K // (8 if bits == 4 else 32)
                    upw, blocks, tiles, ks = p["groups_per_wave"], p["dense_blocks"], p["col_tiles"], p["k_slices"]
                    assert upw >= 1 and blocks >= 1
                    units = -(-tiles // 8) * -(-batch // 64)
                    if batch >= 64 and batch * K * N >= (5.7e9 if bits == 4 else 4e9):  # the wide form: whole rounds unsliced, the rest in ks slices
                        full = units // 256 * 256
                        assert blocks == full + (units - full) * ks and upw % 4 == 0, (bits, K, N, batch, p)
                        assert ks * upw >= U > (ks - 1) * upw, (bits, K, N, batch, p)
                        assert ks == 1 or (units - full) * ks <= 256, (bits, K, N, batch, p)  # the sliced tail fits one round
                        continue
                    if blocks == tiles * ks:  # tile-aligned (or a contiguous cut that happens to be)
                        assert ks * upw >= U > (ks - 1) * upw, (bits, K, N, batch, p)
                        seen_aligned += 1
                    else:
                        assert blocks * upw >= tiles * U > (blocks - 1) * upw, (bits, K, N, batch, p)
                        seen_flat += 1
        assert seen_aligned > 20 and seen_flat > 20
    finally:
        _lib.set_option("cols_min_batch", 0)
        _lib.set_option("cols_max_batch", 0)


def test_wide_plan_whole_rounds_and_sliced_tail_for_any_cu_count():
    """make_plan_wide (csrc/sqllm_capi.hip) on parts of 8 / 104 / 256 / 304 CUs: units of 64 rows x 8 column tiles; whole rounds
    of one unit per CU run unsliced, the remaining units are cut into as many K slices as fit the idle CUs (each slice a whole
    number of 4-unit groups, all of K covered, at most kMaxSlices = 120 slices, none shorter than 8 steps unless K is)."""
    from squeezellm_amd import _lib

    try:
        for cus in (8, 104, 256, 304):
            _lib.set_option("cu_count", cus)
            for bits in (3, 4):
                for K, N in ((5120, 13824), (13824, 5120), (4096, 4096), (8192, 22016), (1024, 776), (32, 4)):
                    for batch in (64, 100, 512, 2048, 5000):
                        _lib.set_option("mfma_wide_min_batch", 64)  # whatever the routing rule says about this shape
                        p = _lib.plan_query(bits, K, N, batch=batch)
                        U = K // (8 if bits == 4 else 32)
                        units = -(-p["col_tiles"] // 8) * -(-batch // 64)
                        full, ks, upw = units // cus * cus, p["k_slices"], p["groups_per_wave"]
                        assert p["grid_y"] == 1 and p["dense_blocks"] == full + (units - full) * ks, (cus, bits, K, N, batch, p)
                        assert upw % 4 == 0 and ks * upw >= U > (ks - 1) * upw and 1 <= ks <= 120, (cus, bits, K, N, batch, p)
                        if ks > 1:
                            assert (units - full) * ks <= cus, (cus, bits, K, N, batch, p)  # the sliced tail is one round
                            assert upw >= (32 if bits == 4 else 8) or ks * upw < U + upw, (cus, bits, K, N, batch, p)
    finally:
        _lib.set_option("cu_count", 0)
        _lib.set_option("mfma_wide_min_batch", 0)


def test_product_sources_carry_no_measurement_routing():
    """The measured-and-not-adopted kernels, their options and their routing live in csrc/experimental/ (measurement
    library) and reach the host layer through the hooks of sqllm_host.h: the product's host source has no
    measurement-build branch at all, and the product library leaves every hook null (unknown options are rejected)."""
    from squeezellm_amd import _lib
    from squeezellm_amd import build as B

    src = open(os.path.join(B.CSRC, "sqllm_capi.hip")).read()
    assert "SQLLM_ABLATION_BUILD" not in src
    assert not any(os.path.basename(s) in ("sqllm_stream.hip", "sqllm_pair.hip", "sqllm_pass.hip") for s in B.SOURCES)
    lib = _lib.load()
    for name in (b"stream", b"pair4", b"ablate", b"lds_pad", b"pass_poll_sleep", b"skip_prepare_small"):
        assert lib.sqllm_set_option(name, 1) == -7, name  # SQLLM_E_OPTION
    assert "TIMING" not in src and "skip_prepare_small()" in src  # the one timing-only mode sits behind a (null) hook
