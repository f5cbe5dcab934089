"""Build libsqllm_hip.so for gfx950 with hipcc -- the replacement for the reference's
`squeezellm/setup_cuda.py` (torch CUDAExtension, /root/reference/squeezellm/setup_cuda.py:4-12).

No torch headers, no pybind11: the product is a plain C-ABI shared library (include/sqllm_hip.h)
built in-tree next to this file, so it travels with the source tree.

    python -m squeezellm_amd.build [--force] [--verbose]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB_NAME = "libsqllm_hip.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
SOURCES = ["sqllm_kernels.hip", "sqllm_capi.hip"]
HEADERS = [os.path.join(CSRC, "sqllm_kernels.h"), os.path.join(INCLUDE, "sqllm_hip.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X kernels cannot be built (no fallback exists)")
    return exe


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip -> squeezellm_amd/libsqllm_hip.so (cross-compiles without a GPU).
    SQLLM_ABLATION=1 in the environment adds the measurement-only ablation kernel variants."""
    if not force and not is_stale():
        return LIB_PATH
    extra = ["-DSQLLM_ABLATION_BUILD"] if os.environ.get("SQLLM_ABLATION") == "1" else []
    if os.environ.get("SQLLM_WAVES"):  # measurement builds: waves per workgroup (default 8)
        extra.append("-DSQLLM_WAVES=" + str(int(os.environ["SQLLM_WAVES"])))
    if os.environ.get("SQLLM_PAIR3"):  # measurement builds: 0 = 3-bit decode with one lookup per weight
        extra.append("-DSQLLM_PAIR3=" + str(int(os.environ["SQLLM_PAIR3"])))
    if os.environ.get("SQLLM_HALF_STAGES"):  # measurement builds: 0 = whole-stage decode (32 live lookups)
        extra.append("-DSQLLM_HALF_STAGES=" + str(int(os.environ["SQLLM_HALF_STAGES"])))
    if os.environ.get("SQLLM_PAIR3_NOCONFLICT"):  # measurement builds (wrong results): 3-bit pair lookups without bank conflicts
        extra.append("-DSQLLM_PAIR3_NOCONFLICT=" + str(int(os.environ["SQLLM_PAIR3_NOCONFLICT"])))
    if os.environ.get("SQLLM_PIPE"):  # measurement builds: software-pipelined 4-bit chunk decode
        extra.append("-DSQLLM_PIPE=" + str(int(os.environ["SQLLM_PIPE"])))
    if os.environ.get("SQLLM_SCHED_PATTERN"):  # measurement builds: fixed decode-stage schedules
        extra.append("-DSQLLM_SCHED_PATTERN=" + str(int(os.environ["SQLLM_SCHED_PATTERN"])))
    extra += os.environ.get("SQLLM_EXTRA_DEFINES", "").split()  # measurement builds: e.g. "-DSQLLM_MFMA_VAR=4"
    cmd = [hipcc(), f"--offload-arch={ARCH}", *FLAGS, *extra, "-shared", f"-I{INCLUDE}", f"-I{CSRC}",
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed ({res.returncode}):\n{res.stdout}\n{res.stderr}")
    if verbose and res.stderr.strip():
        print(res.stderr, file=sys.stderr)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)
    print(path)
