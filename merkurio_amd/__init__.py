"""merkurio_amd -- MI355X-native drop-in for MerKurio's pattern_matching hot path.

Product code lives in csrc/ (hand-written gfx950 HIP kernels + C++ host code behind the C ABI
of include/merkurio_hip.h).  `native` is a thin ctypes mirror of the reference's matcher
surface used by tests and bench.py; PyTorch appears only in bench.py / the multi-GPU host
for device memory, streams and torch.distributed (RCCL).
"""
from . import build  # noqa: F401

__all__ = ["build", "native"]
