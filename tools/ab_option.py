#!/usr/bin/env python3
"""Same-process A/B of a library option over bench.py configs:
    python tools/ab_option.py half_stage 2,1,0 7b-w4-s0,7b-w3-s0 [reps]"""
import io
import json
import os
import runpy
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from squeezellm_amd import _lib  # noqa: E402

name, values, configs = sys.argv[1], [int(v) for v in sys.argv[2].split(",")], sys.argv[3].split(",")
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
for rep in range(reps):
    for v in values:
        for c in configs:
            _lib.set_option(name, v)
            sys.argv = ["bench.py", "--config", c, "--no-cpu-baseline", "--no-sub-records", "--steps", "40"]
            buf = io.StringIO()
            try:
                with redirect_stdout(buf):
                    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
            except SystemExit:
                pass
            d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
            print(name, v, c, d["value"], {k: x["us_mean"] for k, x in d["per_layer_us"].items()}, flush=True)
