"""cropsr_amd -- MI355X-native PAM-scan + on-target-score engine for CROPSR.

Drop-in accelerator for the inner loop of H2muller/CROPSR (CROPSR.py:409-474):
hand-written HIP kernels for gfx950 behind a C ABI (include/cropsr_hip.h),
bound here with ctypes.  See DESIGN.md and INTEGRATION.md.
"""
from .engine import Arena, Engine, Genome, Hits, pack_ascii  # noqa: F401
from .node import Node, NodeHits  # noqa: F401  (one process over N GPUs: the library's node handle)
from ._native import CropsrHipError  # noqa: F401

__all__ = ["Engine", "Arena", "Genome", "Hits", "Node", "NodeHits", "pack_ascii", "CropsrHipError"]
