"""does another cut of a SMALL batch-1 launch (o_proj: 8.7 MB, 0.215 of the roofline by rocprof) help?
Graph wall and per-dispatch events per launch of one shape (tools/ceiling_same_clock.py's protocol: ~600 MB of distinct
weights per shape) under the planner's default and under explicit geometries: target_wgs (dense workgroups per op) and
groups_per_wave (K units each wave walks).  The loads-only winner for o_proj is 256 workgroups x 8 waves x 8 loads in flight
(profiles/r06_stream_patterns.txt: 2.89 us).

    python tools/experiments/small_op_geometry.py [--bits 4]
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
    ap.add_argument("--groups", action="store_true", help="the q/k/v and gate/up GROUP launches instead of the single-op ones")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    # (name: K, N, ops per launch -- a group shares vec; target_wgs is PER OP of the launch)
    shapes = {"o_proj": (4096, 4096, 1), "down": (11008, 4096, 1)}
    targets = (256, 384, 512, 640, 768, 1024, 1536, 2048)
    if a.groups:
        shapes = {"qkv": (4096, 4096, 3), "gate+up": (4096, 11008, 2)}
        targets = (128, 192, 256, 320, 344, 384, 512, 688, 768, 860, 1032)
    sets = [("default", {})] + [("target_wgs=%d" % t, {"target_wgs": t}) for t in targets]
    for name, (K, N, group) in shapes.items():
        mb = K * N * a.bits / 8 / 1e6 * group
        copies = max(4, min(64, int(a.budget_mb / mb)))
        layers = [synth.make_layer(K, N, a.bits, device=dev, seed=100 * c + a.bits) for c in range(copies * group)]
        xs = []
        for c in range(copies):
            x = torch.randn(K, device=dev).half().float()
            xs += [x] * group
        ys = [torch.zeros(N, device=dev) for _ in range(copies * group)]
        # warm the box with the default first, then measure every set twice, alternating (the first set after new inputs reads high)
        order = sets + sets
        seen = {}
        for tag, opts in [sets[0]] + order:
            for k, v in opts.items():
                _lib.set_option(k, v)
            try:
                seq = decode.OpSequence(layers, xs, ys, fuse_shared_input=group > 1)
                assert seq.n_groups == copies
                g = seq.graph(warmup=1)
                blocks = bench.time_blocks(g.replay, torch.cuda.synchronize, 20, 3, 5)
                wall = statistics.median(blocks) / 20 / copies * 1e6
                ev = float(seq.profile(reps=3).mean())
                plan = _lib.plan_query(a.bits, K, N)
            finally:
                for k in opts:
                    _lib.set_option(k, 0)
            seen.setdefault(tag, []).append((wall, ev, plan["dense_blocks"] if group == 1 else None, plan["k_slices"] if group == 1 else None))
            del g, seq
        for tag, _ in sets:
            runs = seen[tag][1:] if tag == "default" else seen[tag]  # (drop the warm-up run of the default)
            print(json.dumps({"shape": name, "bits": a.bits, "set": tag, "dense_workgroups": runs[0][2], "k_slices": runs[0][3],
                              "graph_wall_us": [round(r[0], 3) for r in runs], "event_us": [round(r[1], 3) for r in runs]}), flush=True)
        del layers, xs, ys
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
