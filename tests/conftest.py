import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


NEEDS_LIB = ("test_capi_cpu.py", "test_codegen_cpu.py")  # CPU-side tests that load libsqllm_hip.so
_lib_problem = None


def pytest_configure(config):
    """Make sure the built artefacts exist (both are git-ignored).  On a host without ROCm the HIP
    library cannot be built: the pure-CPU tests (oracle, packer, checkpoint format) still run, the
    tests that need the library are skipped with the reason."""
    global _lib_problem
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run under gpurun)")
    from squeezellm_amd import build as sq_build

    try:
        sq_build.build()
    except sq_build.HipccMissing as e:
        # No ROCm toolchain on this host.  A library that is already there AND newer than its sources
        # (e.g. it travelled with the tree) is used; anything else skips the tests that need it.
        if not os.path.exists(sq_build.LIB_PATH):
            _lib_problem = str(e).splitlines()[0]
        elif sq_build.is_stale():
            _lib_problem = "libsqllm_hip.so is older than its sources and hipcc is not here to rebuild it"
    # (a BuildError -- hipcc ran and the sources do not compile -- propagates: never test a stale binary)
    if not os.path.exists(os.path.join(ROOT, "oracle", "libsqllm_oracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libsqllm_oracle.so"], check=True,
                       capture_output=True)


def pytest_collection_modifyitems(config, items):
    if _lib_problem is None:
        return
    skip = pytest.mark.skip(reason=f"libsqllm_hip.so unavailable: {_lib_problem}")
    for item in items:
        if item.get_closest_marker("gpu") or os.path.basename(str(item.fspath)) in NEEDS_LIB:
            item.add_marker(skip)


POISON_PATTERNS = (0x7FC00000, 0x7F7FFFFF)  # a quiet NaN; FLT_MAX (finite: survives a sum that would drop a NaN through a select)
_lds_poison = None


def _lds_poison_lib():
    """tests/native/lds_poison.hip built in place (hipcc) and loaded: fills every CU's 160 KB of LDS with a pattern."""
    global _lds_poison
    if _lds_poison is None:
        import ctypes
        import shutil

        import torch  # noqa: F401  (its libamdhip64 must be the HIP runtime the helper binds to)

        src = os.path.join(ROOT, "tests", "native", "lds_poison.hip")
        out = os.path.join(ROOT, "tests", "native", "liblds_poison.so")
        try:
            if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
                hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
                subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", out], check=True, capture_output=True)
            lib = ctypes.CDLL(out)
            lib.lds_poison.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
            lib.lds_poison.restype = ctypes.c_int
            _lds_poison = lib
        except (OSError, subprocess.CalledProcessError) as e:  # no hipcc on this host and no prebuilt helper: the tests still run, unpoisoned
            import warnings

            warnings.warn(f"tests/native/lds_poison.hip could not be built or loaded ({e}): LDS is not poisoned in this run")
            _lds_poison = False
    return _lds_poison


@pytest.fixture(autouse=True)
def poisoned_workspaces(request, monkeypatch):
    """Every caller workspace handed to the library in a GPU test is filled with a poison pattern right before the launch
    (the Python module's per-stream buffer: quant_cuda._workspace; a pass's buffer: decode.OpSequence.launch / .profile), so
    a kernel that reads workspace bytes which no kernel of THIS launch wrote turns its output into NaN / 1e38 instead of
    passing on the zeros of a fresh allocation.  The LDS of every CU gets the same treatment at the start of the test and
    in front of those launches (tests/native/lds_poison.hip): LDS is not cleared between kernels, so a ticket or a slab read
    before it is written would otherwise see whatever the previous kernel left -- right or wrong by the luck of placement.
    The pattern is picked per test (hash of its node id): both get used."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import zlib

    import torch

    from squeezellm_amd import decode, quant_cuda

    if not torch.cuda.is_available():  # (the `gpu` fixture reports that)
        yield
        return
    upattern = POISON_PATTERNS[zlib.crc32(request.node.nodeid.encode()) & 1]
    pattern = upattern - ((1 << 32) if upattern >= (1 << 31) else 0)
    lds = _lds_poison_lib()
    sink = torch.zeros(4, dtype=torch.int32, device="cuda:0")

    def poison_lds():
        if lds and not torch.cuda.is_current_stream_capturing():
            rc = lds.lds_poison(torch.cuda.current_stream().cuda_stream, upattern, sink.data_ptr())
            assert rc == 0, rc

    def fill(ws):
        n = ws.numel() // 4 * 4
        if n:
            ws[:n].view(torch.int32).fill_(pattern)

    real_workspace = quant_cuda._workspace

    def workspace(dev, stream, need):
        ws = real_workspace(dev, stream, need)
        if ws is not None:
            with torch.cuda.device(dev):
                fill(ws)
                poison_lds()
        return ws

    real_launch, real_profile = decode.OpSequence.launch, decode.OpSequence.profile

    def launch(self):
        if self._ws is not None:
            fill(self._ws)
        with torch.cuda.device(self.device):
            poison_lds()
        return real_launch(self)

    def profile(self, reps=3):
        if self._ws is not None:
            fill(self._ws)
        return real_profile(self, reps)

    monkeypatch.setattr(quant_cuda, "_workspace", workspace)
    monkeypatch.setattr(decode.OpSequence, "launch", launch)
    monkeypatch.setattr(decode.OpSequence, "profile", profile)
    with torch.cuda.device(0):
        poison_lds()
    yield
    assert int(sink[0]) == 0, "the LDS poison kernel read back something it had not written"


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible")
    return torch.device("cuda:0")
