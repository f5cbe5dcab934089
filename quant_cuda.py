"""Top-level `quant_cuda` module: what `squeezellm/quant.py:5` (`import quant_cuda`) resolves to
when this repository root is on sys.path.  It re-exports the MI355X implementation so the
reference's quant.py / llama.py run unchanged."""
from squeezellm_amd.quant_cuda import *  # noqa: F401,F403
from squeezellm_amd.quant_cuda import __all__  # noqa: F401
