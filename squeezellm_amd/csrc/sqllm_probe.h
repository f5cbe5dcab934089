// sqllm_probe.h -- the ONE place where measurement builds differ from the product inside the kernel templates: per-workgroup
// timeline stamps (tools/timeline.py, tools/experiments/small_split_timeline.py).  In the product library every macro here
// expands to nothing; in the measurement library (python -m squeezellm_amd.build --ablation, -DSQLLM_ABLATION_BUILD) a launch
// whose segments carry a buffer in Segment::bias (operator launches do not use it otherwise; sqllm_debug_set_timeline) gets
// eight 64-bit words per workgroup, stamped from the 100 MHz constant clock (s_memrealtime: comparable across CUs).
#pragma once

#ifdef SQLLM_ABLATION_BUILD
// the workgroup's eight words, or null
#define SQLLM_PROBE_PTR(sg)                                                                                                   \
  ((sg).bias ? reinterpret_cast<unsigned long long*>(const_cast<float*>((sg).bias)) +                                        \
                   8ull * (blockIdx.x + (unsigned long long)gridDim.x * blockIdx.y)                                           \
             : nullptr)
#define SQLLM_PROBE(tl, I, COND)                                 \
  do {                                                           \
    if ((tl) && (COND)) (tl)[I] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
// a stamp stored NEGATED (the CSR chunk workgroups' entry: tells them from the dense ones)
#define SQLLM_PROBE_NEG(tl, I, COND)                                     \
  do {                                                                   \
    if ((tl) && (COND)) (tl)[I] = 0ull - __builtin_amdgcn_s_memrealtime(); \
  } while (0)
// ablation bits of a role (measurement library: option ablate_csr): 0 in the product, where whatever they guard folds away
#define SQLLM_ABLATION_BITS(x) (x)
// the entry stamp with the workgroup's place in its top 16 bits: XCC id (4 bits) above HW_ID[15:8] (CU, SH, SE)
#define SQLLM_PROBE_ENTRY(tl, COND)                                                                                          \
  do {                                                                                                                        \
    if ((tl) && (COND))                                                                                                       \
      (tl)[0] = (__builtin_amdgcn_s_memrealtime() & 0xFFFFFFFFFFFFull) |                                                      \
                ((unsigned long long)(((__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u) << 8) |                   \
                                      ((__builtin_amdgcn_s_getreg(4 | (8 << 6) | (7 << 11))) & 255u))                        \
                 << 48);                                                                                                      \
  } while (0)
#else
#define SQLLM_PROBE_PTR(sg) nullptr
#define SQLLM_PROBE(tl, I, COND) \
  do {                           \
  } while (0)
#define SQLLM_PROBE_ENTRY(tl, COND) \
  do {                              \
  } while (0)
#define SQLLM_PROBE_NEG(tl, I, COND) \
  do {                               \
  } while (0)
#define SQLLM_ABLATION_BITS(x) 0
#endif
