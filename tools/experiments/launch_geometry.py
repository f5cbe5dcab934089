"""Does another cut of a launch help?  Generalises small_op_geometry.py to any shape, group size, batch and sparsity: graph wall and per-dispatch
events per launch of ONE shape (~600 MB of distinct weights, tools/ceiling_same_clock.py's protocol) under the planner's default and under explicit
`target_wgs` (dense workgroups PER OP of the launch).  Every set is measured twice, alternating; the first measurement after new inputs is dropped.

    python tools/experiments/launch_geometry.py --shapes "o13:5120x5120x1,qkv13:5120x5120x3,gateup13:5120x13824x2" --batch 4 --sparse 0.0045 --topx 10 \
        --targets 256,384,512,640,768,1024
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch

    import bench
    from squeezellm_amd import _lib, decode, synth

    ap = argparse.ArgumentParser()
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--budget-mb", type=float, default=600.0)
    ap.add_argument("--shapes", default="o13:5120x5120x1")
    ap.add_argument("--batch", type=int, default=0, help="0: the matvec operator; B >= 1: the *_batched operator with B rows")
    ap.add_argument("--sparse", type=float, default=0.0)
    ap.add_argument("--topx", type=int, default=0)
    ap.add_argument("--targets", default="256,384,512,640,768,1024")
    ap.add_argument("--options", default="", help="library options held for every set: name=value,...")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    held = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.options.split(",") if kv}
    sets = [("default", {})] + [("target_wgs=%d" % int(t), {"target_wgs": int(t)}) for t in a.targets.split(",") if t]
    for spec in a.shapes.split(","):
        name, dims = spec.split(":")
        K, N, group = (int(v) for v in dims.split("x"))
        mb = K * N * a.bits / 8 / 1e6 * group
        copies = max(4, min(64, int(a.budget_mb / mb)))
        layers = [synth.make_layer(K, N, a.bits, sparse_frac=a.sparse, topX=a.topx, heavy_rows=10 if a.sparse else 0, device=dev, seed=100 * c + a.bits)
                  for c in range(copies * group)]
        xs = []
        for c in range(copies):
            x = torch.randn((a.batch, K) if a.batch else (K,), device=dev).half().float()
            xs += [x] * group
        ys = [torch.zeros((a.batch, N) if a.batch else (N,), device=dev) for _ in range(copies * group)]
        seen = {}
        for k, v in held.items():
            _lib.set_option(k, v)
        try:
            for tag, opts in [sets[0]] + sets + sets:
                for k, v in opts.items():
                    _lib.set_option(k, v)
                try:
                    seq = decode.OpSequence(layers, xs, ys, batched=a.batch > 0, fuse_shared_input=group > 1)
                    assert seq.n_groups == copies
                    g = seq.graph(warmup=1)
                    blocks = bench.time_blocks(g.replay, torch.cuda.synchronize, 20, 3, 5)
                    wall = statistics.median(blocks) / 20 / copies * 1e6
                    ev = float(seq.profile(reps=3).mean())
                finally:
                    for k in opts:
                        _lib.set_option(k, 0)
                seen.setdefault(tag, []).append((wall, ev))
                del g, seq
        finally:
            for k in held:
                _lib.set_option(k, 0)
        for tag, _ in sets:
            runs = seen[tag][1:] if tag == "default" else seen[tag]  # (drop the warm-up run of the default)
            print(json.dumps({"shape": name, "K": K, "N": N, "ops": group, "bits": a.bits, "batch": a.batch, "sparse": a.sparse, "set": tag,
                              "graph_wall_us": [round(r[0], 3) for r in runs], "event_us": [round(r[1], 3) for r in runs]}), flush=True)
        del layers, xs, ys
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
