import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run under gpurun)")
    # make sure the built artefacts exist (both are git-ignored)
    from squeezellm_amd import build as sq_build

    sq_build.build()
    if not os.path.exists(os.path.join(ROOT, "oracle", "libsqllm_oracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libsqllm_oracle.so"], check=True,
                       capture_output=True)


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible")
    return torch.device("cuda:0")
