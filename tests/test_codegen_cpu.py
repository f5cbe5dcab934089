"""Guards on the generated gfx950 code of the fused kernel (hipcc cross-compiles without a GPU).

Each of these was a measured regression at some point of the build:
  * a FLAT memory instruction anywhere in the kernel makes the compiler treat vmcnt as out of order,
    and every later wait for a load degrades to vmcnt(0) -- including the dense role's codebook
    staging wait, which must leave the already issued weight loads in flight (+0.3-0.6 us/launch);
  * register spills / scratch in the decode loop;
  * more than 64 VGPRs in the batch-1 kernels (four 8-wave workgroups per CU no longer fit: -18 % on
    7B w3 s45) or more than 128 in the wider batch tiles.
"""
import os
import re
import shutil
import subprocess

import pytest

from squeezellm_amd import build as B


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("asm") / "k.s"
    cmd = [hipcc, f"--offload-arch={B.ARCH}", *[f for f in B.FLAGS if f != "-fPIC"], "-S", "--cuda-device-only",
           f"-I{B.INCLUDE}", f"-I{B.CSRC}", os.path.join(B.CSRC, "sqllm_kernels.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True)
    return out.read_text()


def _strip_preload_preamble(body):
    """Kernels built with kernel-argument preload (-amdgpu-kernarg-preload-count, squeezellm_amd/build.py) begin with a
    compatibility stub -- load the preloaded arguments the old way, wait, branch over padding to the 256-byte-aligned real
    entry -- that firmware with preload support skips.  The guards below are about the real entry."""
    for i, l in enumerate(body[:12]):
        if re.match(r"\s+\.p2align\s+8", l) and any("s_branch" in b for b in body[:i]):
            return body[:1] + body[i + 1:]
    return body


def _kernels(asm):
    out = {}
    for m in re.finditer(r"^(_ZN5sqllm18sqllm_fused_matvec\w+):.*?^\.Lfunc_end", asm, re.S | re.M):
        out[m.group(1)] = _strip_preload_preamble(m.group(0).split("\n"))
    return out


def test_all_instantiations_present(asm):
    ks = _kernels(asm)
    # {3,4} bits x batch tile {1,2,4,8} x {operator, fused linear} + the operator's tiles of exactly 3 / 5 / 6 / 7 rows
    # + the 4-bit batch-1 operator kernel with two-step chunks (launches whose K slices are all at most two steps per wave)
    assert len(ks) == 25


def test_no_flat_memory_instructions(asm):
    for name, body in _kernels(asm).items():
        flat = [l.strip() for l in body if re.match(r"\s+flat_", l)]
        assert not flat, f"{name}: {flat[:3]}"


def test_codebook_staging_wait_leaves_the_weight_loads_in_flight(asm):
    for name, body in _kernels(asm).items():
        i0 = next(i for i, l in enumerate(body) if "global_load_dwordx4" in l and " nt" in l)  # dense role's first weight load
        i1 = next(i for i in range(i0, len(body)) if "s_barrier" in body[i])                 # the staging barrier
        waits = [int(m.group(1)) for l in body[i0:i1] for m in [re.search(r"s_waitcnt vmcnt\((\d+)\)", l)] if m]
        assert waits and min(waits) >= 3, f"{name}: waits {waits} between the first weight load and the staging barrier"


def test_no_spills_and_occupancy_targets(asm):
    meta = re.findall(r"\.name:\s+(_ZN5sqllm18sqllm_fused_matvec\w+).*?\.private_segment_fixed_size:\s+(\d+).*?"
                      r"\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", asm, re.S)
    assert len(meta) == 25  # (+ the 4-bit batch-1 operator kernel with two-step chunks for launches of short K slices)
    for name, scratch, sspill, vgpr, vspill in meta:
        # (a few SGPRs parked in VGPR lanes are tolerated: no memory traffic; scratch is not.  The whole
        # segment descriptor is held in SGPRs from the prologue on -- one round of scalar loads instead of a
        # dependent chain -- which costs the widest fused-linear tiles a few more parked SGPRs)
        assert int(scratch) == 0 and int(vspill) == 0 and int(sspill) <= 32, (name, scratch, sspill, vspill)
        batch1 = re.search(r"matvecILi[34]ELi1E", name) is not None
        assert int(vgpr) <= (64 if batch1 else 128), (name, vgpr)  # four / two 8-wave workgroups per CU
        # three per CU (80 VGPRs) for the operator's 2- / 3-row tiles and its 4-bit 4- / 5- / 6-row tiles (round 3: -6...-13 % per launch)
        m = re.search(r"matvecILi([34])ELi(\d)ELi8ELi0ELb0", name)
        if m and (m.group(2) in "23" or (m.group(1) == "4" and m.group(2) in "456")):
            assert int(vgpr) <= 80, (name, vgpr)
        # four per CU for the 4-bit 2-row tile, operator and fused linear (round 6: half stages, two steps per chunk)
        if re.search(r"matvecILi4ELi2E", name):
            assert int(vgpr) <= 64, (name, vgpr)


def test_prologue_reads_the_argument_block_in_one_round(asm):
    """A dependent scalar load costs 0.15 us (tools/experiments/dispatch_ramp.hip): everything a workgroup
    of segment 0 needs -- vec, the block table, the whole segment descriptor -- must be requested before
    the first wait, and the descriptor of another segment in ONE further round."""
    for name, body in _kernels(asm).items():
        first_wait = next(i for i, l in enumerate(body) if "s_waitcnt" in l)
        loads = [i for i, l in enumerate(body) if re.match(r"\s+s_load_", l)]
        early = [i for i in loads if i < first_wait]
        assert len(early) >= 6, f"{name}: only {len(early)} scalar loads before the first wait"
        # the next batch (descriptor of a later segment) is contiguous and followed by one wait
        after = [i for i in loads if i > first_wait]
        later = after[:1]
        for i in after[1:]:  # the contiguous run of scalar loads that follows
            if i - later[-1] > 2:
                break
            later.append(i)
        assert 4 <= len(later) <= 8 and later[-1] - later[0] <= 8, f"{name}: descriptor reload is not one batch: lines {later}"
        # ... and, in the batch-1 operator kernels (the decode path), nothing else is read from the argument
        # block before the first vector load (wider tiles may re-read vec's address under register pressure)
        if not re.search(r"matvecILi[34]ELi1ELi8ELi0ELb0E", name):
            continue
        first_vmem = next(i for i, l in enumerate(body) if re.match(r"\s+(global|buffer)_load", l))
        stray = [i for i in loads if later[-1] < i < first_vmem]
        assert not stray, f"{name}: scalar loads at lines {stray} between the prologue and the first vector load"


def test_three_bit_batch1_decode_uses_pair_lookups(asm):
    """The 3-bit batch-1 kernels look two weights up with ONE ds_read_b64 (64-entry pair tables,
    DESIGN.md 4.1): per 32-k unit and column 16 eight-byte lookups and 16 packed FMAs, and no
    four-byte lookups left in the decode."""
    for name, body in _kernels(asm).items():
        if not re.search(r"matvecILi3ELi1E", name):
            continue
        b64 = sum(1 for l in body if re.match(r"\s+ds_read_b64", l))
        b32 = sum(1 for l in body if re.match(r"\s+ds_read_b32", l))
        pk = sum(1 for l in body if re.match(r"\s+v_pk_fma_f32", l))
        assert b64 >= 128 and pk >= 128, (name, b64, pk)  # two copies of the step (first chunk + loop) x 64
        assert b32 <= 64, (name, b32)                       # epilogue / sparse roles (row searches; the operator kernel carries the CSR role at two chunk sizes) only


# ---- round 2: the batched kernel families ----

def _family(asm, mangled_prefix):
    out = {}
    for m in re.finditer(r"^(" + mangled_prefix + r"\w+):.*?^\.Lfunc_end", asm, re.S | re.M):
        out[m.group(1)] = m.group(0).split("\n")
    return out


def _meta(asm, mangled_prefix):
    return re.findall(r"\.name:\s+(" + mangled_prefix + r"\w+).*?\.private_segment_fixed_size:\s+(\d+).*?"
                      r"\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", asm, re.S)


def test_column_lane_kernels(asm):
    """lane = column, vec in SGPRs: the x operands of the packed FMAs must be SGPR pairs fed by scalar loads
    with an SGPR offset (the compiler's own pointer arithmetic once cost two v_readlane per load), no
    FLAT, no scratch, <= 80 VGPRs (three 8-wave workgroups per CU)."""
    ks = _family(asm, "_ZN5sqllm16sqllm_fused_cols")
    assert len(ks) == 16  # {3,4} bits x rows per pass {1,2,3,4,5,6,7,8}
    for name, body in ks.items():
        assert not [l for l in body if re.match(r"\s+flat_", l)], name
        pk = [l for l in body if re.match(r"\s+v_pk_fma_f32", l)]
        assert pk and all(re.search(r"v_pk_fma_f32 v\[\d+:\d+\], v\[\d+:\d+\], s\[\d+:\d+\], v\[\d+:\d+\]", l) for l in pk), name
        sload = [l for l in body if re.match(r"\s+s_load_dwordx[48] s\[\d+:\d+\], s\[\d+:\d+\], s\d+", l)]
        bt = int(re.search(r"colsILi[34]ELi(\d)E", name).group(1))
        assert len(sload) >= 2 * bt, (name, len(sload))
        assert [l for l in body if "buffer_load_dword" in l], name  # range-checked weight loads
    for name, scratch, sspill, vgpr, vspill in _meta(asm, "_ZN5sqllm16sqllm_fused_cols"):
        assert int(scratch) == 0 and int(vspill) == 0 and int(vgpr) <= 80, (name, scratch, vspill, vgpr)


def test_matrix_core_kernels(asm):
    """fp32 MFMA kernels: 32 matrix instructions per decoded group and row block, no FLAT, no LDS float
    atomics in the epilogue (ds_add_f32 executes lane by lane: 50 us per launch once), register budgets of
    the measured occupancy (two workgroups per CU up to 32 rows for 4-bit)."""
    ks = _family(asm, "_ZN5sqllm19sqllm_fused_batched")
    assert len(ks) == 6  # {3,4} bits x {1,2,4} row blocks
    for name, body in ks.items():
        assert not [l for l in body if re.match(r"\s+flat_", l)], name
        assert not [l for l in body if re.match(r"\s+ds_add_(rtn_)?f32", l)], name
        mb = int(re.search(r"batchedILi[34]ELi(\d)E", name).group(1))
        n_mfma = sum(1 for l in body if "v_mfma_f32_16x16x4_f32" in l or "v_mfma_f32_16x16x4f32" in l)
        assert n_mfma >= 64 * mb, (name, n_mfma)
    budget = {(4, 1): 104, (4, 2): 128, (4, 4): 224, (3, 1): 128, (3, 2): 176, (3, 4): 256}
    for name, scratch, sspill, vgpr, vspill in _meta(asm, "_ZN5sqllm19sqllm_fused_batched"):
        bits, mb = map(int, re.search(r"batchedILi([34])ELi(\d)E", name).groups())
        assert int(vgpr) <= budget[(bits, mb)], (name, vgpr)
        if (bits, mb) not in ((4, 2), (3, 4)):  # (these two trade a few spilled dwords for their occupancy)
            assert int(scratch) == 0 and int(vspill) == 0, (name, scratch, vspill)


def test_wide_batch_sparse_kernel(asm):
    ks = _family(asm, "_ZN5sqllm20sqllm_sparse_batched")
    assert len(ks) == 1
    for name, body in ks.items():
        assert not [l for l in body if re.match(r"\s+flat_", l)], name
        assert [l for l in body if "v_readlane_b32" in l], name  # scalar walk over the non-zeros
    for name, scratch, sspill, vgpr, vspill in _meta(asm, "_ZN5sqllm20sqllm_sparse_batched"):
        assert int(scratch) == 0 and int(vspill) == 0 and int(vgpr) <= 128, (name, scratch, vspill, vgpr)


@pytest.fixture(scope="module")
def asm_wide(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("asm_wide") / "w.s"
    cmd = [hipcc, f"--offload-arch={B.ARCH}", *[f for f in B.FLAGS if f != "-fPIC"], "-S", "--cuda-device-only",
           f"-I{B.INCLUDE}", f"-I{B.CSRC}", os.path.join(B.CSRC, "sqllm_mfma_wide.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True)
    return out.read_text()


def _loops(body):
    """(start, end) line ranges of the backward branches of a kernel body."""
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if l.startswith(".LBB")}
    out = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            out.append((labels[m.group(1)], i))
    return out


def test_wide_form_kernels(asm_wide):
    """The wide form of the split matrix-core kernel (csrc/sqllm_mfma_wide.hip): two waves per SIMD (<= 256 registers), no
    FLAT, no scratch traffic inside a loop, no barrier at all; the planes kernels address their fragment loads with a
    scalar base + one 32-bit offset; and the HAND SCHEDULE of the phase survives the compiler: in the main loops (two
    steps = 160 matrix instructions with five partial products, 192 with six) a handful of matrix instructions (at most 6, with six products 8)
    stand back to back -- the slots the next column's lookups leave free -- and the column's eight LDS reads sit between
    them.  (Without the pins in wide_phase the packing of a phase's last column sank behind the phase: 19 in a row.)"""
    ks = _family(asm_wide, "_ZN5sqllm16sqllm_fused_wide")
    assert len(ks) == 4  # {3,4} bits x {planes, fp32 vec}
    for name, scratch, sspill, vgpr, vspill in _meta(asm_wide, "_ZN5sqllm16sqllm_fused_wide"):
        assert int(vgpr) <= 256, (name, vgpr)
        if "ILi4E" in name:
            assert int(scratch) == 0 and int(vspill) == 0, (name, scratch, vspill)
    for name, body in ks.items():
        assert not [l for l in body if re.match(r"\s+flat_", l)], name
        assert not [l for l in body if re.match(r"\s+s_barrier", l)], name
        planes = "ELb1E" in name
        if planes:
            assert [l for l in body if re.search(r"global_load_dwordx4 v\[\d+:\d+\], v\d+, s\[\d+:\d+\] offset:1024", l)], name
        main = [(a, b) for a, b in _loops(body) if sum("v_mfma_f32_16x16x32_bf16" in l for l in body[a:b]) >= 160]
        assert main, name
        counts = set()
        for a, b in main:
            loop = body[a:b]
            assert not [l for l in loop if "scratch_" in l], name
            n_mfma = sum("v_mfma_f32_16x16x32_bf16" in l for l in loop)
            if n_mfma not in (160, 192, 640, 768):  # (an enclosing loop: counted once through its inner ones)
                continue
            counts.add(n_mfma)
            run = longest = 0
            for l in loop:
                t = l.strip()
                if not t or t.startswith(";") or t.startswith("."):
                    continue
                if t.startswith("v_mfma"):
                    run += 1
                    longest = max(longest, run)
                elif not t.startswith("s_"):  # (scalar bookkeeping and waits do not take a vector issue slot)
                    run = 0
            assert longest <= (6 if n_mfma % 160 == 0 else 8), (name, n_mfma, longest)  # (free slots 1, 6, 7, 20-23 and, at 3 bits, a slot 0 without instructions)
            per_col = 20 if n_mfma % 160 == 0 else 24
            assert sum(l.strip().startswith("ds_read_b64") for l in loop) * per_col >= n_mfma * 8, name  # (at least) 8 lookups per column of 20 / 24 matrix instructions
        assert counts, name
        if planes:
            assert len(counts) == 2, (name, counts)  # the five- and the six-product role
