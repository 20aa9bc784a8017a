"""rfn_hip — ctypes binding + autograd glue for librfn_hip.so (gfx950 kernels of the RFN hot path)."""
import os

# ROCm 7.2: with graph packet capture on, hipGraph memset nodes (PyTorch multi-block reductions zero their semaphores
# with one) race with neighbouring kernel nodes on replay.  Effective only when set before the HIP runtime initialises,
# so the entry points (bench.py, main_rfn.py, tests/conftest.py) also set it before importing torch;
# Solver.capture_graph refuses to capture without it.  librfn_hip itself never enqueues memsets (csrc/common.h).
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

from . import lib, ops  # noqa: E402,F401
