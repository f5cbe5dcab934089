"""Fused linear vs bare operator, whole 7B model (batch 1, graph replay), by term: dense only, + CSR, + top-X rows.

    python tools/linear_overhead.py
"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from squeezellm_amd import decode, synth
dev = torch.device("cuda:0")
def timed(fn, reps=30, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for bits in (3, 4):
    for frac, topX in ((0.0, 0), (0.0045, 0), (0.0045, 10)):
        layers = synth.make_model("llama-7b", bits, sparse_frac=frac, topX=topX, n_layers=None, device=dev)
        xin = {}; xs16 = []
        for i, l in enumerate(layers):
            lname = l["name"].split(".")[-1]
            key = (i // 7, "h" if lname in ("q_proj", "k_proj", "v_proj") else "m" if lname in ("gate_proj", "up_proj") else lname)
            if key not in xin: xin[key] = torch.randn((l["K"],), device=dev).half()
            xs16.append(xin[key])
        ys16 = [torch.empty(l["N"], device=dev, dtype=torch.float16) for l in layers]
        g3 = decode.OpSequence(layers, xs16, ys16, fuse_shared_input=True, linear=True, fold_topx=os.environ.get("FOLD_TOPX", "1") != "0").graph()
        x32 = {id(x): x.float() for x in xs16}
        ys32 = [torch.zeros(l["N"], device=dev) for l in layers]
        g4 = decode.OpSequence(layers, [x32[id(x)] for x in xs16], ys32, fuse_shared_input=True).graph()
        a = timed(g3.replay); b = timed(g4.replay); a2 = timed(g3.replay); b2 = timed(g4.replay)
        print(json.dumps(dict(bits=bits, sparse=frac, topX=topX, linear_ms=round(min(a, a2), 4), op_ms=round(min(b, b2), 4), overhead_pct=round((min(a, a2) / min(b, b2) - 1) * 100, 1))), flush=True)
        del layers, g3, g4, ys16, ys32, x32, xin, xs16
        torch.cuda.empty_cache()
