import os
# ROCm 7.2: with graph packet capture on, hipGraph memset nodes (PyTorch multi-block reductions zero their semaphores
# with one) race with neighbouring kernel nodes on replay; must be set before the HIP runtime initialises.
import sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
if "torch" not in sys.modules:  # the flag certainly precedes the HIP runtime: tell rfn_hip.graph_capture_safe()
    os.environ.setdefault("RFN_GRAPH_ENV_BEFORE_TORCH", "1")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "recurrent-flows-msc_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import torch

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = torch.load(os.path.join(GOLDEN, name), weights_only=True)
        return cache[name]

    return load
