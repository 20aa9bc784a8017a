"""rfn_hip — ctypes binding + autograd glue for librfn_hip.so (gfx950 kernels of the RFN hot path)."""
import os

# ROCm 7.2: with graph packet capture on, hipGraph memset nodes (PyTorch multi-block reductions zero their semaphores
# with one) race with neighbouring kernel nodes on replay.  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 avoids it, but only when it
# is in the environment BEFORE the HIP runtime initialises.  The entry points (bench.py, main_rfn.py, __graft_entry__.py,
# tests/conftest.py) export it before importing torch; this package does NOT set it (setting it here, possibly after an
# integrator already touched torch.cuda, would make the guard below pass although the flag has no effect) -- it records
# what it finds at import time, and Solver.capture_graph refuses to capture unless graph_capture_safe().
# librfn_hip itself never enqueues memsets (csrc/common.h).
GRAPH_ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"


def _initial_environ_value(name):
    """the value `name` had when the process was exec'ed (None if absent / unreadable)"""
    try:
        with open("/proc/self/environ", "rb") as f:
            for item in f.read().split(b"\0"):
                if item.startswith(name.encode() + b"="):
                    return item.split(b"=", 1)[1].decode()
    except OSError:
        pass
    return None


def _capture_safe(initial_value, value_at_import, hip_up_at_import, set_before_torch=False):
    """pure decision logic (unit tested): the flag certainly preceded HIP initialisation when the process was started
    with it, or when it was already '0' while this package was imported and the runtime had not been brought up yet, or
    when an entry point of this repository set it before importing torch and says so (RFN_GRAPH_ENV_BEFORE_TORCH=1)"""
    if initial_value == "0":
        return True
    return value_at_import == "0" and (not hip_up_at_import or set_before_torch)


def _hip_up():
    try:
        import torch
        return bool(torch.cuda.is_initialized())
    except Exception:  # noqa: BLE001
        return False


_STATE = (_initial_environ_value(GRAPH_ENV), os.environ.get(GRAPH_ENV), _hip_up(),
          os.environ.get("RFN_GRAPH_ENV_BEFORE_TORCH") == "1")


def graph_capture_safe():
    """True when hipGraph replays of the training step can be trusted in this process (see the note above)."""
    return _capture_safe(*_STATE)


from . import lib, ops  # noqa: E402,F401
