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


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible")
    return torch.device("cuda:0")
