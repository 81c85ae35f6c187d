"""diffews_amd: MI355X-native (gfx950) implementation of the DiffewS hot path.

Python host code on PyTorch-ROCm (device memory, streams, torch.distributed)
calling hand-written HIP kernels through a C-ABI shared library
(include/diffews_hip.h, diffews_amd/csrc/).  See DESIGN.md.
"""
__version__ = "0.1.0"
