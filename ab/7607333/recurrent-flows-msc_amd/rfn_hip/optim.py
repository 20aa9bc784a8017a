"""Adam for the training step (RFN/trainer.py:96 of the reference: torch.optim.Adam(model.parameters(), lr)) on the
one-launch HIP kernel rfn_adam_step_f32.  A subclass of torch.optim.Adam: constructor, param_groups, state layout
({"step", "exp_avg", "exp_avg_sq"} per parameter) and state_dict()/load_state_dict() are torch's, so checkpoints written
by either load into the other (the reference's rfn.pt holds `optimizer_state_dict`); only step() differs.

The kernel reads a device table of (p, g, m, v, numel, step offset) per tensor.  The table is rebuilt whenever a pointer
changed: never in hipGraph mode (the gradient tensors are the graph's static outputs), every step in eager mode (autograd
allocates fresh gradients), where the rebuild is one small host-to-device copy next to ~7000 eager launches."""
import ctypes

import numpy as np
import torch

from . import lib as L


class HipAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, foreach=False,
                         fused=False)
        self._key = None
        self._keep = None
        self._t = 0           # kernel step counter; a tensor's own count is _t - its step_offset
        self._steps = {}      # id(p) -> step count as of the last step() (host ints; state["step"] is synced lazily)
        self._dirty = False

    # ---- torch-visible state -------------------------------------------------------------------------------------
    def _sync_step_tensors(self):
        if self._dirty:
            for group in self.param_groups:
                for p in group["params"]:
                    st = self.state.get(p)
                    if st is not None and id(p) in self._steps:
                        st["step"] = torch.tensor(float(self._steps[id(p)]))
            self._dirty = False

    def state_dict(self):
        self._sync_step_tensors()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._key, self._steps, self._dirty = None, {}, False

    # ---- the step ------------------------------------------------------------------------------------------------
    def _build(self, group, params, key):
        dev = params[0].device
        chunk = int(L.load().rfn_adam_chunk_elems())
        ent = np.zeros((len(params), 6), dtype=np.int64)   # rfn_adam_entry: 4 pointers, long n, (int offset, int pad)
        chunks = []
        for i, p in enumerate(params):
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if id(p) not in self._steps:
                self._steps[id(p)] = int(float(st["step"]))
            m, v, g = st["exp_avg"], st["exp_avg_sq"], p.grad
            for name, t in (("parameter", p), ("gradient", g), ("exp_avg", m), ("exp_avg_sq", v)):
                if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
                    raise RuntimeError("HipAdam: %s must be a dense fp32 device tensor (got %s %s, contiguous=%s)" %
                                       (name, t.dtype, t.device, t.is_contiguous()))
            n = p.numel()
            for name, t in (("gradient", g), ("exp_avg", m), ("exp_avg_sq", v)):
                if t.numel() != n:   # e.g. moments of another batch size loaded from a checkpoint: the kernel indexes by n
                    raise RuntimeError("HipAdam: %s has %d elements, its parameter %d (shape %s)" %
                                       (name, t.numel(), n, tuple(p.shape)))
            off = self._t - self._steps[id(p)]
            ent[i] = (p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, off & 0xFFFFFFFF)
            nck = (n + chunk - 1) // chunk
            chunks.append(np.stack([np.full(nck, i, dtype=np.int32), np.arange(nck, dtype=np.int32)], 1))
        chunks = np.concatenate(chunks, 0) if chunks else np.zeros((0, 2), np.int32)
        tab_d = torch.from_numpy(ent.view(np.uint8).reshape(-1)).to(dev)
        chk_d = torch.from_numpy(np.ascontiguousarray(chunks).reshape(-1)).to(dev)
        self._keep = (tab_d, chk_d, int(chunks.shape[0]), list(params), 28.0 * sum(p.numel() for p in params))
        self._key = key

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if len(self.param_groups) != 1:
            raise NotImplementedError("HipAdam: one parameter group (what the reference's trainer builds)")
        group = self.param_groups[0]
        if group.get("amsgrad") or group.get("maximize"):
            raise NotImplementedError("HipAdam: amsgrad / maximize are not implemented")
        params = [p for p in group["params"] if p.grad is not None]
        if not params:
            return loss
        for p in params:
            if not p.grad.is_contiguous():
                p.grad = p.grad.contiguous()
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in params)
        if key != self._key or any(id(p) not in self._steps for p in params):
            # a parameter that skipped steps keeps its own count through its step_offset
            self._build(group, params, key)
        tab_d, chk_d, nck, plist, nbytes = self._keep
        # every listed tensor advances by one; the kernel computes bias corrections from t - step_offset
        self._t += 1
        for p in plist:
            self._steps[id(p)] += 1
        self._dirty = True
        beta1, beta2 = group["betas"]
        L.call("rfn_adam_step_f32", ctypes.c_void_p(tab_d.data_ptr()), ctypes.c_void_p(chk_d.data_ptr()), ctypes.c_int(nck),
               ctypes.c_double(float(group["lr"])), ctypes.c_double(beta1), ctypes.c_double(beta2),
               ctypes.c_double(group["eps"]), ctypes.c_double(group["weight_decay"]), ctypes.c_int(self._t),
               meta=("shell", "adam", 0.0, "%d tensors" % len(plist), nbytes))
        # the kernel wrote through raw pointers: tell autograd's version counters, which everything keyed on
        # `p._version` relies on (ListGlow._reverse_cache, RFN._gen_graph: cached inverse matrices / packs / hipGraph)
        torch.autograd.graph.increment_version(plist)
        return loss
