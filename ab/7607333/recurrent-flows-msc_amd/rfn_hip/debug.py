"""Finite-value tracing that also works inside a captured hipGraph (RFN_DEBUG_FINITE=1): every check() writes
isfinite(t).all() into a slot of one persistent device tensor, report() reads them back after a step / replay.
Development aid only; a no-op unless the environment variable is set."""
import os

import torch

ENABLED = os.environ.get("RFN_DEBUG_FINITE") == "1"
_names, _flags, _idx = [], None, 0


def begin():
    """start of a step: slot names are re-recorded on every python execution (a replay keeps the capture's names)"""
    global _idx
    _idx = 0
    del _names[:]


def check(name, t):
    global _flags, _idx
    if not ENABLED or t is None or not torch.is_tensor(t):
        return
    if _flags is None:
        _flags = torch.ones(8192, device=t.device)
    _names.append(name)
    with torch.no_grad():
        _flags[_idx:_idx + 1].copy_(torch.isfinite(t.detach()).all().float().reshape(1))
    _idx += 1


def report():
    if _flags is None:
        return []
    vals = _flags[:len(_names)].tolist()
    return [n for n, v in zip(_names, vals) if v == 0]
