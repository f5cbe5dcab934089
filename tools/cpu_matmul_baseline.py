#!/usr/bin/env python3
"""The CPU baseline BASELINE.md section 3 plans next to the GPU numbers: the "reference CPU
torch.matmul path" on the host cores of the GPU box.

  path A  dequant + matmul : unpack qweight -> indices, W = lookup_table.gather(idx) -> [N, K], y = W @ x
  path B  matmul only      : W pre-dequantised once; only torch.matmul is timed (fp32 and fp16/bf16)

for the LLaMA-7B linear shapes, batch 1, summed to tokens/s over the 224 linears of the model.
Uses the product's own tensor-level unpacker (squeezellm_amd.pack), no oracle.  Prints one JSON line.

    python tools/cpu_matmul_baseline.py [--bits 4] [--reps 10]
"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from squeezellm_amd import pack  # noqa: E402

SHAPES = [("q/k/v/o_proj", 4096, 4096, 4), ("gate/up_proj", 4096, 11008, 2), ("down_proj", 11008, 4096, 1)]


def med(fn, reps):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return statistics.median(ts), min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    g = torch.Generator().manual_seed(0)
    out = {"cores": os.cpu_count(), "threads": torch.get_num_threads(), "bits": a.bits, "per_shape_ms": {}}
    tot = {"dequant_matmul_f32": 0.0, "matmul_f32": 0.0, "matmul_bf16": 0.0}
    for name, K, N, count in SHAPES:
        rows_q = K // 32 * a.bits
        q = torch.randint(-2**31, 2**31, (rows_q, N), dtype=torch.int64, generator=g).to(torch.int32)
        lut = torch.sort(torch.randn((N, 1 << a.bits), generator=g) * 0.02, dim=1).values
        x = torch.randn(K, generator=g)

        def dequant():
            idx = pack.unpack_qweight(q, a.bits).to(torch.int64)  # [K, N]
            return lut.gather(1, idx.t().contiguous())            # [N, K]

        W = dequant()
        Wb, xb = W.to(torch.bfloat16), x.to(torch.bfloat16)
        tA, _ = med(lambda: dequant() @ x, max(3, a.reps // 3))
        tB, tBmin = med(lambda: W @ x, a.reps)
        tC, _ = med(lambda: Wb @ xb, a.reps)
        assert torch.allclose(dequant() @ x, W @ x)
        out["per_shape_ms"][f"{K}x{N}"] = {"dequant+matmul f32": round(tA * 1e3, 3), "matmul f32": round(tB * 1e3, 3),
                                          "matmul f32 min": round(tBmin * 1e3, 3), "matmul bf16": round(tC * 1e3, 3)}
        tot["dequant_matmul_f32"] += count * tA
        tot["matmul_f32"] += count * tB
        tot["matmul_bf16"] += count * tC
    out["tokens_per_s_llama7b"] = {k: round(1.0 / (32 * v), 2) for k, v in tot.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
