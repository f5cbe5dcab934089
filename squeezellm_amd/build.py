"""Build libsqllm_hip.so for gfx950 with hipcc -- the replacement for the reference's
`squeezellm/setup_cuda.py` (torch CUDAExtension, /root/reference/squeezellm/setup_cuda.py:4-12).

No torch headers, no pybind11: the product is a plain C-ABI shared library (include/sqllm_hip.h)
built in-tree next to this file, so it travels with the source tree.

    python -m squeezellm_amd.build [--force] [--verbose]
    python -m squeezellm_amd.build --ablation [--out PATH]     # measurement build, separate library
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
SOURCES = ["sqllm_kernels.hip", "sqllm_mfma_split.hip", "sqllm_mfma_wide.hip", "sqllm_capi.hip"]
# measured-and-not-adopted kernels (round 3: the streaming batch-1 kernel, the column-pair-table kernel; round 4: the
# dependency-gated persistent pass) and the host code that routes to them: csrc/experimental/, part of the MEASUREMENT
# library only (options "stream" / "pair4", entry points sqllm_pass_*)
EXPERIMENTAL = os.path.join(CSRC, "experimental")
EXPERIMENT_SOURCES = ["experimental/sqllm_ablation.hip", "experimental/sqllm_stream.hip", "experimental/sqllm_pair.hip",
                      "experimental/sqllm_pass.hip", "experimental/sqllm_experimental.hip"]
HEADERS = [os.path.join(CSRC, h) for h in ("sqllm_kernels.h", "sqllm_decode.h", "sqllm_roles.h", "sqllm_fused.h", "sqllm_split_common.h", "sqllm_host.h", "sqllm_probe.h")] + [os.path.join(INCLUDE, "sqllm_hip.h")]
EXPERIMENT_HEADERS = [os.path.join(EXPERIMENTAL, h) for h in ("sqllm_pass.h", "sqllm_pass_api.h")]
ARCH = "gfx950"
# -amdgpu-kernarg-preload-count: the kernels' leading explicit arguments (the vec pointer) arrive in SGPRs instead of
# through the first scalar load (gfx950 preloads up to 16 dwords; the by-value descriptor block cannot be preloaded):
# mean kernel duration of the 7B batch-1 launches -2.5 %, same-box A/B profiles/r05_kernarg_preload_ab.txt
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]


class HipccMissing(RuntimeError):
    """No hipcc on this host: the library cannot be built here (and there is no fallback)."""


class BuildError(RuntimeError):
    """hipcc ran and failed: the sources do not compile."""


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise HipccMissing("hipcc not found: the MI355X kernels cannot be built (no fallback exists)")
    return exe


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


ABLATION_LIB_NAME = "libsqllm_hip_ablation.so"
ABLATION_LIB_PATH = os.path.join(HERE, ABLATION_LIB_NAME)


def ablation_is_stale() -> bool:
    if not os.path.exists(ABLATION_LIB_PATH):
        return True
    t = os.path.getmtime(ABLATION_LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + EXPERIMENT_SOURCES] + HEADERS + EXPERIMENT_HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)
# measurement-only preprocessor switches (environment variable -> macro).  They select kernel variants that
# are slower or deliberately WRONG (ablations); a build that sets any of them is an ablation build and can
# only be written to libsqllm_hip_ablation.so, never to the product library.
# (SQLLM_WAVES=4, round 2's 4-wave workgroups, is gone from the list: the column-lane kernel, the top-X role and the CSR
# role's lane runs are written for 8 waves and say so in static_asserts)
VARIANT_ENV = ("SQLLM_PAIR3", "SQLLM_HALF_STAGES", "SQLLM_PAIR3_NOCONFLICT", "SQLLM_MFMA_VAR", "SQLLM_MFMA_FAKE", "SQLLM_TOPX_ROWS", "SQLLM_CSR_CHUNK")


def _compile(out: str, extra, verbose: bool, sources=None) -> str:
    cmd = [hipcc(), f"--offload-arch={ARCH}", *FLAGS, *extra, "-shared", f"-I{INCLUDE}", f"-I{CSRC}", f"-I{EXPERIMENTAL}",
           *[os.path.join(CSRC, s) for s in (sources or SOURCES)], "-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise BuildError(f"hipcc failed ({res.returncode}):\n{res.stdout}\n{res.stderr}")
    if verbose and res.stderr.strip():
        print(res.stderr, file=sys.stderr)
    os.replace(out + ".tmp", out)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip -> squeezellm_amd/libsqllm_hip.so (cross-compiles without a GPU).  The product
    build takes NO variant switches: the measurement variants live in build_ablation()."""
    if not force and not is_stale():
        return LIB_PATH
    return _compile(LIB_PATH, [], verbose)


def build_ablation(verbose: bool = False, out: str | None = None, force: bool = True) -> str:
    """Measurement build (ablation kernels, timeline probes, calibration kernels, and whatever variant
    switches the environment names) -> libsqllm_hip_ablation.so (or `out`, which must not be the product
    library).  Load it with SQLLM_LIB=<path>."""
    out = os.path.abspath(out or ABLATION_LIB_PATH)
    if out == os.path.abspath(LIB_PATH):
        raise ValueError("an ablation build must not overwrite the product library")
    if not force and out == os.path.abspath(ABLATION_LIB_PATH) and not ablation_is_stale():
        return out
    extra = ["-DSQLLM_ABLATION_BUILD"]
    for name in VARIANT_ENV:
        if os.environ.get(name):
            extra.append(f"-D{name}={int(os.environ[name])}")
    extra += os.environ.get("SQLLM_EXTRA_DEFINES", "").split()
    return _compile(out, extra, verbose, SOURCES[:-1] + EXPERIMENT_SOURCES + SOURCES[-1:])


if __name__ == "__main__":
    _verbose = "--verbose" in sys.argv or "-v" in sys.argv
    if "--ablation" in sys.argv or os.environ.get("SQLLM_ABLATION") == "1":
        _out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
        print(build_ablation(verbose=_verbose, out=_out))
    else:
        print(build(force="--force" in sys.argv, verbose=_verbose))
