#!/bin/bash
# build_variant.sh <csrc dir> <name>: all product sources of <csrc dir> with build.py's flags -> squeezellm_amd/ab/lib<name>.so
set -e
C=$1; N=$2; R=/root/repo
mkdir -p $R/squeezellm_amd/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 -shared -I$R/include -I$C -I$C/experimental $C/sqllm_kernels.hip $C/sqllm_mfma_split.hip $C/sqllm_mfma_wide.hip $C/sqllm_capi.hip -o $R/squeezellm_amd/ab/lib$N.so
ls -la $R/squeezellm_amd/ab/lib$N.so
