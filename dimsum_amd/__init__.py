"""dimsum_amd -- MI355X (gfx950) native implementation of the DiMSUM denoiser hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); every hot op runs a hand-written
HIP kernel from libdimsum_hip.so through the C ABI declared in include/dimsum_hip.h. There is NO CPU fallback:
ops raise RuntimeError when the library is missing or a tensor is not on the GPU.
"""
__version__ = "0.1.0"
