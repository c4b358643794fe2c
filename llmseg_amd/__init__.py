"""llmseg_amd -- MI355X-native (gfx950) implementation of LLM-Seg's `model_forward` hot path.

`csrc/` holds the HIP kernels and the C ABI (`include/llmseg_hip.h`); `ops.py` binds them to torch tensors;
`lisa.py` mirrors the reference's `LISAForCausalLM` interface on top of them.
"""
__version__ = "0.1.0"
