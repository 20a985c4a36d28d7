"""sygnals_amd -- MI355X (gfx950) backend for the sygnals feature-extraction hot path.

Host side is Python mirroring the reference's function / plugin surface; arithmetic runs
in hand-written HIP kernels behind the C ABI of include/sygnals_hip.h.
"""
__version__ = "0.1.0"
