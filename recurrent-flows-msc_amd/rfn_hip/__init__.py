"""rfn_hip — ctypes binding + autograd glue for librfn_hip.so (gfx950 kernels of the RFN hot path)."""
from . import lib, ops  # noqa: F401
