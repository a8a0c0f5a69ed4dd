"""anyedit_amd — MI355X-native (gfx950) implementation of AnyEdit's diffusion denoising hot path.

Python host code mirroring the reference's `ldm/`, `segment_anything` image-encoder and AnySD operator API,
calling hand-written HIP kernels through the C ABI in include/anyedit_hip.h (libanyedit_hip.so).
PyTorch is used for device memory, streams, RNG and torch.distributed (RCCL) only.
"""
__version__ = "0.1.0"
