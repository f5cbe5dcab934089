"""squeezellm_amd -- MI355X (gfx950) implementation of SqueezeLLM's dense-and-sparse
LUT-quantised matvec hot path, behind the reference's `quant_cuda` operator API.

    squeezellm_amd.quant_cuda   the 12 reference operator names (+ the 2 `balanced` ones)
    squeezellm_amd.quant        QuantLinearLUT mirror (buffer schema + forward dispatch)
    squeezellm_amd.decode       whole-pass launchers (one FFI crossing / HIP-graph replay)
    squeezellm_amd.sharding     layer-sharded pipeline over the GPUs of a node (RCCL)
    squeezellm_amd.synth        synthetic operands at the reference's model shapes
    squeezellm_amd.build        hipcc build of libsqllm_hip.so (replaces setup_cuda.py)

The compute lives in libsqllm_hip.so (C ABI: include/sqllm_hip.h).  Nothing here falls back to
CPU or eager PyTorch.
"""
__version__ = "0.1.0"
