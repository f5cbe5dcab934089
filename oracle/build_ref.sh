#!/usr/bin/env bash
# ORACLE TEST INFRASTRUCTURE.  Compile the reference's kernel file UNMODIFIED, from where it lies
# under /root/reference, into oracle/_ref/libsqllm_ref.so (gfx950).  Nothing from the reference is
# copied into this repo: ref_shim/ref_capi.hip #includes squeezellm/quant_cuda_kernel.cu via -I.
# The reference's own build (setup_cuda.py -> torch CUDAExtension/hipify) is NOT run: it fails on
# torch 2.10 (probe table).  Only the dev container has /root/reference; the built .so
# travels to the GPU box with the gpurun snapshot (it is git-ignored, not gpurun-ignored).
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
ref="${SQLLM_REFERENCE_DIR:-/root/reference}"
if [ ! -f "$ref/squeezellm/quant_cuda_kernel.cu" ]; then
  echo "build_ref.sh: $ref not present -- keeping any prebuilt oracle/_ref" >&2
  exit 0
fi
mkdir -p "$here/_ref"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC \
  -I"$here/ref_shim" -I"$ref" \
  -Wno-unused-value -Wno-deprecated-declarations \
  "$here/ref_shim/ref_capi.hip" -o "$here/_ref/libsqllm_ref.so"
echo "built $here/_ref/libsqllm_ref.so"
# Stage the reference's UNMODIFIED QuantLinearLUT module next to it (git-ignored like the .so, and
# travelling with the gpurun snapshot like the .so): tests/test_gpu_reference_forward.py runs ITS
# forward() -- the caller the drop-in claim is about -- on the real kernels of this repository.
mkdir -p "$here/_ref/reference_py"
cp "$ref/squeezellm/quant.py" "$here/_ref/reference_py/quant.py"
echo "staged $here/_ref/reference_py/quant.py (unmodified copy of $ref/squeezellm/quant.py)"
