#!/usr/bin/env python3
"""Build (here, no GPU needed) or run (on the GPU box) the ablation binaries of the wide matrix-core kernel.
    python tools/experiments/wide_ablate/run.py build     # -> tools/experiments/wide_ablate/bin/<variant>
    python tools/experiments/wide_ablate/run.py run       # prints one line per variant and configuration
Each variant is a text patch of a COPY of csrc/sqllm_mfma_wide.hip."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
CSRC = os.path.join(ROOT, "squeezellm_amd", "csrc")
SRC = os.path.join(CSRC, "sqllm_mfma_wide.hip")
BIN = os.path.join(HERE, "bin")

LOADX_HEAD = "  auto load_x = [&](int g, int ph, u32x4 (&dx)[NX]) {\n    const int gu = group_unit(g);"
LOADW_HEAD = "  auto load_w = [&](int g, u32x4 (&dw)[R]) {\n    const int u = clamp_unit(group_unit(g));"


def patch(text, name):
    wide = text.index("dense_role_mfma_wide(const void*")  # patches below apply from the wide role on (load_x / load_w) or to split_phase
    head, tail = text[:wide], text[wide:]
    if "noA" in name:  # vec values loaded for group 0 only
        tail = tail.replace(LOADX_HEAD, LOADX_HEAD.replace("{\n", "{\n    if (g > 0 || ph > 0) return;\n", 1), 1)
    if "noW" in name:  # weights loaded for groups 0 and 1 only
        tail = tail.replace(LOADW_HEAD, LOADW_HEAD.replace("{\n", "{\n    if (g > 1) return;\n", 1), 1)
    if "noLDS" in name:  # entries made up from the addresses instead of looked up
        old = "            e[i] = lds_read_u32x2(ad + jn * kColStride<BITS>);"
        assert old in head + tail
        head, tail = head.replace(old, "            e[i] = u32x2{ad, ad ^ wmask};", 1), tail.replace(old, "            e[i] = u32x2{ad, ad ^ wmask};", 1)
    if "halfMFMA" in name:  # three of the partial products dropped
        old = "        acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, Bx, acc[mb][j], 0, 0, 0);"
        assert old in head + tail
        new_ = "        if (pp == 2 || pp >= 4) acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, Bx, acc[mb][j], 0, 0, 0);"
        head, tail = head.replace(old, new_, 1), tail.replace(old, new_, 1)
    if "noLut" in name:  # codebook values made up instead of gathered
        old = "    for (int i = 0; i < NLV; ++i) lv[i] = lp[i];"
        assert old in tail
        tail = tail.replace(old, "    for (int i = 0; i < NLV; ++i) lv[i] = f32x4{1e-3f * c, 2e-3f * i, 3e-3f, 4e-3f}; (void)lp;", 1)
    if "wt4" in name:  # four column tiles (waves) per workgroup, two workgroups per CU
        for old, new in (("constexpr int kWideTiles = 8;", "constexpr int kWideTiles = 4;"),
                         ("__global__ void __launch_bounds__(kWaves * 64, 2)\nsqllm_fused_wide", "__global__ void __launch_bounds__(kWideTiles * 64, 2)\nsqllm_fused_wide"),
                         ('  static_assert(kWaves == kWideTiles, "one column tile per wave");\n', "")):
            assert old in head + tail, old
            head, tail = head.replace(old, new, 1), tail.replace(old, new, 1)
    if "noEpi" in name:
        tail = tail.replace("  if (c0 < N) {  // (N is a multiple of 4", "  if (c0 < N && batch < 0) {  // (N is a multiple of 4", 1)
    return head + tail


VARIANTS = ["base", "wt4"]


def build():
    os.makedirs(BIN, exist_ok=True)
    text = open(SRC).read()
    for v in VARIANTS:
        src = os.path.join(BIN, f"kernel_{v}.hip")
        open(src, "w").write(patch(text, v))
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fno-gpu-rdc", f"-I{ROOT}/include", f"-I{CSRC}",
               f'-DKERNEL_SOURCE="{src}"', f'-DVARIANT="{v}"', f'-DWT={4 if "wt4" in v else 8}', os.path.join(HERE, "harness.hip"), "-o", os.path.join(BIN, v)]
        subprocess.check_call(cmd)
        os.remove(src)
        print("built", v, flush=True)


def run():
    for args in (["4", "128", "0", "0"], ["4", "256", "0", "0"], ["4", "512", "0", "0"], ["4", "2048", "0", "0"], ["4", "2048", "1", "0"], ["3", "2048", "0", "0"]):
        for v in VARIANTS:
            subprocess.call([os.path.join(BIN, v)] + args)
        print(flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
