"""The product's four batch-1 launch shapes of a LLaMA-7B decoder layer and loads-only kernels over the same packed
bytes, on ONE clock: wall time per launch inside a replayed HIP graph of launches of that one shape over distinct
weights (~600 MB per shape, so nothing is served by the 256 MiB Infinity Cache), same box, same session.

VERDICT r5 item 3(a): BASELINE.md section 7 divided the product's in-process EVENT time per launch (no launch boundary)
by the loads-only kernels' GRAPH WALL time (boundary included).  This tool times both the second way.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/stream_patterns.hip -o /tmp/sp
    python tools/ceiling_same_clock.py --loads-only /tmp/sp        # prints one JSON line per (bits, sparsity) + a table

Loads-only figures: the best pattern / geometry of stream_patterns per shape.  Its o_proj / qkv / gate+up / down rows
are the 4-BIT byte counts: the 3-bit product is shown beside the same figure, which is then no floor for it (3/4 of the
bytes) -- the ratio to read for 3 bits is product 3-bit / product 4-bit.
"""
import argparse
import json
import os
import re
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def product_shapes(bits, sparse, dev, budget_mb):
    """{shape name: (us per launch by graph wall, us per launch by per-dispatch events, algorithmic MB per launch)}"""
    import torch

    import bench
    from squeezellm_amd import decode, synth

    shapes = {"o_proj": [(4096, 4096)], "qkv": [(4096, 4096)] * 3, "gate+up": [(4096, 11008)] * 2, "down": [(11008, 4096)]}
    out = {}
    for name, ops in shapes.items():
        mb = sum(K * N * bits / 8 for K, N in ops) / 1e6
        copies = max(4, min(64, int(budget_mb / mb)))
        layers, xs, ys = [], [], []
        for c in range(copies):
            x = torch.randn(ops[0][0], device=dev).half().float()
            for i, (K, N) in enumerate(ops):
                lay = synth.make_layer(K, N, bits, sparse_frac=0.0045 if sparse else 0.0, topX=10 if sparse else 0,
                                       heavy_rows=10 if sparse else 0, device=dev, seed=1000 * c + i + bits)
                layers.append(lay)
                xs.append(x)
                ys.append(torch.zeros(N, device=dev))
        seq = decode.OpSequence(layers, xs, ys, fuse_shared_input=True)
        assert seq.n_groups == copies, (seq.n_groups, copies)
        g = seq.graph(warmup=1)
        blocks = bench.time_blocks(g.replay, torch.cuda.synchronize, 20, 3, 5)
        wall = statistics.median(blocks) / 20 / copies * 1e6
        ev = float(seq.profile(reps=3).mean())
        alg = sum(synth.layer_bytes(l) for l in layers[:len(ops)]) / 1e6
        out[name] = (round(wall, 3), round(ev, 3), round(alg, 3), copies)
        del g, seq, layers, xs, ys
        torch.cuda.empty_cache()
    return out


def loads_only(binary):
    """best (minimum) graph-wall microseconds per launch of stream_patterns per shape, and the line it came from"""
    p = subprocess.run([binary], capture_output=True, text=True, timeout=900)
    best = {}
    for line in p.stdout.splitlines():
        m = re.match(r"(\S+)\s+([\d.]+) MB (pattern.*?):\s+([\d.]+) us/launch", line)
        if not m:
            continue
        name, us = m.group(1), float(m.group(4))
        if name not in best or us < best[name][0]:
            best[name] = (us, float(m.group(2)), m.group(3).strip())
    return best, p.stdout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--loads-only", default="/tmp/sp", help="the compiled tools/experiments/stream_patterns.hip")
    ap.add_argument("--budget-mb", type=float, default=600.0)
    ap.add_argument("--raw-out", default=None, help="where to keep stream_patterns' full output")
    a = ap.parse_args()
    import torch

    dev = torch.device("cuda:0")
    lo, raw = loads_only(a.loads_only)
    if a.raw_out:
        open(a.raw_out, "w").write(raw)
    rows = []
    for bits, sparse in ((4, False), (4, True), (3, False), (3, True)):
        prod = product_shapes(bits, sparse, dev, a.budget_mb)
        rec = {"config": "7b-w%d-%s" % (bits, "s45" if sparse else "s0"), "shapes": {}}
        layer_wall = layer_ev = layer_lo = layer_mb = 0.0
        for name, (wall, ev, alg, copies) in prod.items():
            l = lo.get(name)
            rec["shapes"][name] = {"product_graph_wall_us": wall, "product_event_us": ev, "algorithmic_MB": alg, "copies": copies,
                                   "loads_only_graph_wall_us": l[0] if l else None, "loads_only_MB": l[1] if l else None,
                                   "loads_only_variant": l[2] if l else None,
                                   "product_over_loads_only_same_clock": round(wall / l[0], 3) if l else None}
            layer_wall += wall
            layer_ev += ev
            layer_mb += alg
            layer_lo += l[0] if l else 0.0
        rec["decoder_layer"] = {"product_graph_wall_us": round(layer_wall, 2), "product_event_us": round(layer_ev, 2),
                                "loads_only_graph_wall_us": round(layer_lo, 2), "algorithmic_MB": round(layer_mb, 2),
                                "product_frac_of_8TBps_wall": round(layer_mb / layer_wall / 8.0, 4) if layer_wall else None,  # (MB / us = TB/s)
                                "loads_only_over_product_same_clock": round(layer_lo / layer_wall, 3)}
        rows.append(rec)
        print(json.dumps(rec), flush=True)
    # 4-bit ceiling as a fraction of 8 TB/s: the 4-bit algorithmic bytes of the layer over the loads-only wall
    w4 = rows[0]
    mb4 = w4["decoder_layer"]["algorithmic_MB"]
    lo_us = w4["decoder_layer"]["loads_only_graph_wall_us"]
    print(json.dumps({"summary": "7b-w4-s0 decoder layer", "loads_only_frac_of_8TBps": round(mb4 / lo_us / 8.0, 4),
                      "product_frac_of_8TBps_graph_wall": round(mb4 / w4["decoder_layer"]["product_graph_wall_us"] / 8.0, 4),
                      "product_frac_of_8TBps_events": round(mb4 / w4["decoder_layer"]["product_event_us"] / 8.0, 4)}))


if __name__ == "__main__":
    main()
