"""pasero_amd — MI355X-native (gfx950) implementation of Pasero's Transformer encoder-decoder training hot path.

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed over RCCL).
Compute:   hand-written HIP kernels behind the C ABI of include/pasero_hip.h (libpasero_hip.so).
"""
__version__ = '0.1.0'
