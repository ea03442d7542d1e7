"""femasr_amd — MI355X-native (gfx950) SR-inference hot path of FeMaSR.

Python host (this package) -> ctypes -> `csrc/libfemasr_hip.so` (flat C ABI,
`include/femasr_hip.h`) -> hand-written HIP kernels.  The only model surface is
`FeMaSRNet`, registered in `ARCH_REGISTRY` exactly as the reference registers
its own (basicsr/archs/femasr_arch.py:214-215).
"""
from .registry import ARCH_REGISTRY, Registry  # noqa: F401
