// ORACLE TEST INFRASTRUCTURE -- not product code.
// Minimal stand-in for <torch/all.h> so that the reference's squeezellm/quant_cuda_kernel.cu can be
// compiled UNMODIFIED, from where it lies, by hipcc (oracle/build_ref.sh).  It provides only what
// that file touches: torch::Tensor::{size, data_ptr<T>, data<T>, type} and
// AT_DISPATCH_FLOATING_TYPES.  No libtorch is linked; tensors are (pointer, shape) views built by
// oracle/ref_shim/ref_capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace torch {
struct Tensor {
  void* ptr = nullptr;
  int64_t shape[2] = {0, 0};
  Tensor() = default;
  Tensor(const void* p, int64_t s0, int64_t s1 = 1) : ptr(const_cast<void*>(p)) { shape[0] = s0; shape[1] = s1; }
  int64_t size(int i) const { return shape[i]; }
  template <typename T> T* data_ptr() const { return static_cast<T*>(ptr); }
  template <typename T> T* data() const { return static_cast<T*>(ptr); }
  int type() const { return 0; }  // only ever fed to AT_DISPATCH_FLOATING_TYPES below
};
}  // namespace torch

// The reference dispatches SPMV_ATOMIC on vals' dtype but then reads vec/mul as that same
// scalar_t and the dense kernels as float (quant_cuda_kernel.cu:261-280), so float is the only
// instantiation that can work; the stand-in pins scalar_t = float.
#define AT_DISPATCH_FLOATING_TYPES(TYPE, NAME, ...) \
  do {                                              \
    (void)(TYPE);                                   \
    using scalar_t = float;                         \
    __VA_ARGS__();                                  \
  } while (0)
