#!/usr/bin/env python3
"""Markdown rows for DESIGN.md / BASELINE.md from the bench lines `tools/collect_profiles.sh <tag>` left under profiles/.

    python tools/bench_table.py r03
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = [("7b-w4-s0", "7B w4 s0, fused q/k/v + gate/up"), ("7b-w4-s0_unfused", "7B w4 s0, one launch per linear"),
        ("7b-w3-s0", "7B w3 s0"), ("7b-w4-s45", "7B w4 s45"), ("7b-w3-s45", "7B w3 s45"), ("13b-w4-s45", "13B w4 s45"),
        ("65b-w3-s45", "65B w3 s45, one GPU")]


def load(tag, name):
    with open(os.path.join(ROOT, "profiles", f"{tag}_bench_{name}.json")) as f:
        return json.loads([l for l in f if l.startswith("{")][-1])


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    print("| config | launches/token | tokens/s (first block; median of 5) | avg kernel us | roofline.frac | wall frac |")
    print("|---|---|---|---|---|---|")
    for name, label in ROWS:
        d = load(tag, name)
        r = d["roofline"]
        print(f"| {label} | {d['config']['launches_per_token']} | {d['value']:.0f}; {d['repeats']['value_median']:.0f} | "
              f"{r['avg_kernel_us']:.2f} | {r['frac']:.3f} | {d['hbm_frac_wall']:.3f} |")
    d = load(tag, "7b-w4-s0")
    print("\nper launch (w4 s0, in-bench events):", {k: v["us_mean"] for k, v in d["per_layer_us"].items()})
    print("traffic per launch:", d["roofline"]["traffic"], "algorithmic:", d["roofline"]["algorithmic_bytes_per_launch"],
          {k: (v.get("roofline") or {}).get("traffic") for k, v in d["sub_records"].items()})
    print("13B s45 layer, us at 1/2/4/8 rows:", {b: round(1e3 * v["ms_per_decoder_layer"], 1)
                                                  for b, v in d["sub_records"]["13b-w4-s45-batched"].items() if b.startswith("batch")})
    print("drop_in:", {k: v.get("tokens_per_s") for k, v in d["drop_in"].items() if isinstance(v, dict)})
    s = load(tag, "samebox_7b-w4-s0")
    print("same-box un-profiled default line:", s["value"], s["roofline"]["frac"])


if __name__ == "__main__":
    main()
