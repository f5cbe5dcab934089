import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezellm_amd import _lib
for v in (0, 1, 0, 1):
    _lib.set_option("topx_fold", v)
    sys.argv = ["bench.py", "--config", "7b-w3-s45", "--no-cpu-baseline", "--steps", "30"]
    print("topx_fold", v, flush=True)
    try:
        runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
    except SystemExit:
        pass
