"""The caller-workspace entry points (include/sqllm_hip.h: sqllm_launch_ws, sqllm_launch_group(s)_ws, sqllm_workspace_bytes).

The reference's launchers allocate nothing (/root/reference/squeezellm/quant_cuda_kernel.cu:580-657).  With a workspace
of sqllm_workspace_bytes the batched ops keep that contract at every batch: a stream capture of a 64- or 256-row hybrid
op contains kernel nodes only (no memory-allocation / free nodes), and the default memory pool's release threshold is left
as the application set it; with a NULL workspace nothing is allocated up to 16 rows either (the sparse terms gather).  The
workspace-less names keep their stream-ordered scratch as the fallback (and only they touch the pool): checked against the
oracle eagerly and as a captured graph with memory nodes, replayed twice.  Runs in a fresh process: the pool threshold is
process-wide state other tests may have raised.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import ctypes, json, sys
import numpy as np
import torch
sys.path.insert(0, %r)
from squeezellm_amd import _lib, quant_cuda
from tests import helpers as H

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)

def threshold():
    pool = ctypes.c_void_p()
    assert hip.hipDeviceGetDefaultMemPool(ctypes.byref(pool), 0) == 0
    v = ctypes.c_uint64(0)
    assert hip.hipMemPoolGetAttribute(pool, 4, ctypes.byref(v)) == 0  # hipMemPoolAttrReleaseThreshold
    return v.value

out = {"threshold_start": threshold()}
lib = _lib.load()
_lib.set_option("mfma_min_batch", 5)  # (explicit: these small shapes would otherwise stay on the batch tiles up to 8 rows, which need no workspace)
for batch in (8, 64, 256):
    case = H.make_case(4, 512, 320, sparse=0.02, topX=3, heavy_rows=1, seed=batch)
    t = H.to_torch(case, dev)
    rng = np.random.default_rng(batch)
    x = rng.normal(size=(batch, 512)).astype(np.float32)
    mul = rng.normal(size=(batch, 320)).astype(np.float32)
    xt, yt = torch.from_numpy(x).to(dev), torch.from_numpy(mul).to(dev)
    # (a) the module's batched name: launches through sqllm_launch_ws with the module's workspace
    H.call_op(quant_cuda, t, xt, yt, "hybrid", True)
    torch.cuda.synchronize()
    out[f"err_{batch}"] = float(H.rel_err(yt.cpu().numpy(), H.oracle_ref(case, x, mul, "hybrid")))
    # (b) a raw capture of sqllm_launch_ws: node types of the graph
    op = _lib.SqllmOp(bits=4, batch=batch, K=512, N=320, vec=xt.data_ptr(), qweight=t["qweight"].data_ptr(), mul=yt.data_ptr(),
                      lookup_table=t["lookup_table"].data_ptr(), rows=t["rows"].data_ptr(), cols=t["cols"].data_ptr(), vals=t["vals"].data_ptr(),
                      nnz=t["vals"].numel(), topX=3, full_rows=t["full_rows"].data_ptr(), full_row_indices=t["full_row_indices"].data_ptr())
    need = int(lib.sqllm_workspace_bytes(ctypes.byref(op), 1))
    out[f"need_{batch}"] = need
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    assert hip.hipStreamBeginCapture(ctypes.c_void_p(s.cuda_stream), 0) == 0  # hipStreamCaptureModeGlobal
    rc = lib.sqllm_launch_ws(ctypes.byref(op), ws.data_ptr(), ws.numel(), s.cuda_stream)
    g = ctypes.c_void_p()
    assert hip.hipStreamEndCapture(ctypes.c_void_p(s.cuda_stream), ctypes.byref(g)) == 0 and rc == 0, rc
    n = ctypes.c_size_t(0)
    assert hip.hipGraphGetNodes(g, None, ctypes.byref(n)) == 0
    nodes = (ctypes.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(g, nodes, ctypes.byref(n)) == 0
    types = []
    for nd in nodes:
        ty = ctypes.c_int(-1)
        assert hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(ty)) == 0
        types.append(ty.value)
    out[f"node_types_{batch}"] = types
    hip.hipGraphDestroy(g)
out["threshold_after_ws"] = threshold()
# (c) sqllm_launch_ws with a NULL workspace at 8 rows (fused small launch): no allocation either -- the sparse terms gather
# from vec -- so the pool is still as it was, and the result is the oracle's
case8 = H.make_case(4, 512, 320, sparse=0.02, topX=3, heavy_rows=1, seed=8)
t8 = H.to_torch(case8, dev)
x8 = np.random.default_rng(8).normal(size=(8, 512)).astype(np.float32)
m8 = np.random.default_rng(9).normal(size=(8, 320)).astype(np.float32)
y8 = torch.from_numpy(m8).to(dev)
H.call_op(quant_cuda, t8, torch.from_numpy(x8).to(dev), y8, "hybrid", True, entry="ws-null")
torch.cuda.synchronize()
out["err_ws_null_8"] = float(H.rel_err(y8.cpu().numpy(), H.oracle_ref(case8, x8, m8, "hybrid")))
out["threshold_after_ws_null"] = threshold()
# (d) the workspace-less name at 64 rows: stream-ordered scratch, the fallback -- it may raise the pool's threshold
op.batch = 64
x64 = np.random.default_rng(64).normal(size=(64, 512)).astype(np.float32)
m64 = np.random.default_rng(65).normal(size=(64, 320)).astype(np.float32)
xt64, yt64 = torch.from_numpy(x64).to(dev), torch.from_numpy(m64).to(dev)
op.vec, op.mul = xt64.data_ptr(), yt64.data_ptr()
assert lib.sqllm_launch(ctypes.byref(op), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
out["threshold_after_fallback"] = threshold()
out["err_fallback_64"] = float(H.rel_err(yt64.cpu().numpy(), H.oracle_ref(case, x64, m64, "hybrid")))  # (`case`: the 256-row loop's last)
# (e) the same workspace-less op CAPTURED with scratch_in_capture = 1 (default): the graph carries the scratch as a
# memory-allocation and a memory-free node; instantiated and replayed twice it accumulates the oracle's result twice
yt64.copy_(torch.from_numpy(m64))
s = torch.cuda.Stream(dev)
torch.cuda.synchronize()
assert hip.hipStreamBeginCapture(ctypes.c_void_p(s.cuda_stream), 0) == 0
rc = lib.sqllm_launch(ctypes.byref(op), s.cuda_stream)
g = ctypes.c_void_p()
assert hip.hipStreamEndCapture(ctypes.c_void_p(s.cuda_stream), ctypes.byref(g)) == 0 and rc == 0, rc
n = ctypes.c_size_t(0)
assert hip.hipGraphGetNodes(g, None, ctypes.byref(n)) == 0
nodes = (ctypes.c_void_p * n.value)()
assert hip.hipGraphGetNodes(g, nodes, ctypes.byref(n)) == 0
types = []
for nd in nodes:
    ty = ctypes.c_int(-1)
    assert hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(ty)) == 0
    types.append(ty.value)
out["node_types_scratch_capture"] = types
ge = ctypes.c_void_p()
assert hip.hipGraphInstantiate(ctypes.byref(ge), g, None, None, 0) == 0
ref1 = H.oracle_ref(case, x64, m64, "hybrid")
for rep in (1, 2):
    assert hip.hipGraphLaunch(ge, ctypes.c_void_p(s.cuda_stream)) == 0
    assert hip.hipStreamSynchronize(ctypes.c_void_p(s.cuda_stream)) == 0
    want = ref1 if rep == 1 else H.oracle_ref(case, x64, ref1.astype(np.float32), "hybrid")
    out[f"err_scratch_capture_replay{rep}"] = float(H.rel_err(yt64.cpu().numpy(), want))
hip.hipGraphExecDestroy(ge)
hip.hipGraphDestroy(g)
print("RESULT " + json.dumps(out))
"""


def test_ws_launches_allocate_nothing_and_leave_the_pool_alone():
    import json

    p = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    for batch in (8, 64, 256):
        assert out[f"err_{batch}"] <= 2e-5, out
        assert out[f"need_{batch}"] > 0, out
        types = out[f"node_types_{batch}"]
        assert types and all(t == 0 for t in types), (batch, types)  # hipGraphNodeTypeKernel only: no MemAlloc (10) / MemFree (11)
    assert out["threshold_after_ws"] == out["threshold_start"], out  # the default pool is left as it was
    # a NULL workspace at 8 rows: still nothing allocated, and the gathered result is right
    assert out["threshold_after_ws_null"] == out["threshold_start"] and out["err_ws_null_8"] <= 2e-5, out
    # the workspace-less name: scratch from the pool (threshold raised), result checked against the oracle
    assert out["threshold_after_fallback"] >= out["threshold_after_ws"] and out["err_fallback_64"] <= 2e-5, out
    # ... and captured: MemAlloc (10) and MemFree (11) nodes around the kernels, two replays right
    types = out["node_types_scratch_capture"]
    assert 10 in types and 11 in types and types.count(0) >= 2, types
    assert out["err_scratch_capture_replay1"] <= 2e-5 and out["err_scratch_capture_replay2"] <= 2e-5, out


def test_workspace_sizes():
    from squeezellm_amd import _lib

    # batch 1 and the batch tiles need none; 5..16 rows the transposed vec (rows rounded up to 8 / 16) and its bf16 planes in
    # fragment order (K / 32 + 1 k blocks of 3 KB); wider what the stream-ordered scratch would hold
    planes = (5120 // 32 + 1) * 3072
    assert _lib.workspace_bytes(4, 5120, 13824, 1, nnz=330_000, topX=10) == 0
    assert _lib.workspace_bytes(4, 5120, 13824, 4, nnz=330_000, topX=10) == 0
    assert _lib.workspace_bytes(4, 5120, 13824, 8, nnz=330_000, topX=10) == 5120 * 8 * 4 + planes
    assert _lib.workspace_bytes(4, 5120, 13824, 16, nnz=330_000, topX=10, n_ops=2) == 5120 * 16 * 4 + planes
    assert _lib.workspace_bytes(4, 5120, 13824, 16) == 0  # dense-only: nothing to transpose for
    assert _lib.workspace_bytes(4, 5120, 13824, 64, nnz=330_000, topX=10) >= 5120 * 64 * 4
