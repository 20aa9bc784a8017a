"""Data parallelism for the RFN training step: one process per GPU, the sequence batch sharded on dim 0, gradients
all-reduced with RCCL over xGMI (torch.distributed backend "nccl" IS RCCL on ROCm; "gloo" is used by the CPU tests).

The reference has no working multi-GPU path (its nn.DataParallel wrapper is broken and bypassed), so
the contract here is "same mathematics as one process running the global batch":
  * shared parameters are broadcast from rank 0 at start and their gradients are averaged every step;
  * the batch-shaped learnable initial states (z_0, z_0x, h_0, c_0, a_0, ca_0 — RFN_new.py:69-76) are SHARDED: each
    rank owns the rows of its local sequences, so they are never reduced; their gradients come from a LOCAL batch mean
    and are divided by the world size (a single process takes the mean over the global batch); checkpoints hold the
    gathered global rows (`gather_sharded_state` / `load_sharded_state`);
  * data dependent ActNorm initialisation happens on rank 0's first batch and is broadcast
    (`broadcast_module_state`, called by `Solver.train_step` after the first forward);
  * gradient buckets are reduced asynchronously while backward is still running (flow parameters become ready first),
    in a few large buckets: xGMI is point-to-point, fewer / larger collectives amortise the per-link latency.
"""
import torch
import torch.distributed as dist

SHARDED_PARAM_NAMES = ("z_0", "z_0x", "h_0", "c_0", "a_0", "ca_0")


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def collective_device(group=None):
    """where a tensor must live to go through `group`'s backend: the current GPU for nccl (= RCCL, device memory only),
    the host for gloo"""
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


class GradBucketReducer:
    """Flat-bucket asynchronous gradient averaging driven by post-accumulate-grad hooks."""

    def __init__(self, named_params, bucket_bytes=64 << 20, group=None, sharded_names=SHARDED_PARAM_NAMES):
        self.group = group
        self.world = dist.get_world_size(group) if is_dist() else 1
        # top-level batch-shaped initial states are sharded over ranks, everything else is replicated
        named_params = list(named_params)
        params = [(n, p) for n, p in named_params if p.requires_grad and not ("." not in n and n in sharded_names)]
        self.sharded = [p for n, p in named_params if p.requires_grad and "." not in n and n in sharded_names]
        self.params = [p for _, p in params]
        self.names = [n for n, _ in params]
        # buckets in REVERSE registration order: backward produces gradients roughly last-layer-first
        self.buckets, cur, cur_bytes = [], [], 0
        for p in reversed(self.params):
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self._bucket_of[id(p)] = bi
        self._pending = [0] * len(self.buckets)
        self._works = []
        self._flat = [None] * len(self.buckets)
        self._hooks = []
        self.add_hooks()
        self.reset()

    def add_hooks(self):
        if self.world > 1 and not self._hooks:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def remove_hooks(self):
        """no overlap with backward: finish() then reduces every bucket after the fact (hipGraph-replayed backward)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._works = []

    def _flat_views(self, bi):
        """the bucket's persistent flat buffer (allocated once, on the gradients' device) and one view per parameter"""
        if self._flat[bi] is None:
            b = self.buckets[bi]
            flat = torch.empty(sum(p.numel() for p in b), device=b[0].device, dtype=b[0].dtype)
            views, off = [], 0
            for p in b:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            self._flat[bi] = (flat, views)
        return self._flat[bi]

    def _launch(self, bi):
        """gather the bucket's gradients into its flat buffer (one multi-tensor copy, no allocation) and start ONE
        in-place all-reduce on it"""
        b = self.buckets[bi]
        flat, views = self._flat_views(bi)
        have = [(v, p.grad) for v, p in zip(views, b) if p.grad is not None]
        if len(have) < len(b):
            flat.zero_()   # parameters that got no gradient on this rank contribute zeros (others may have one)
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        if flat.is_cuda and dist.get_backend(self.group) == "gloo":
            # CPU-side collective on device memory (the one-GPU rehearsal of the N>1 path): without this the ranks sharing a
            # GPU were observed to stall for tens of seconds inside gloo's own stream hand-over
            torch.cuda.current_stream().synchronize()
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append((bi, work))

    def _on_grad(self, p):
        bi = self._bucket_of[id(p)]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def finish(self):
        """call after backward(): flush buckets whose parameters got no gradient, wait, write averages back."""
        if self.world == 1:
            return
        grads = [p.grad for p in self.sharded if p.grad is not None]
        if grads:
            torch._foreach_div_(grads, float(self.world))   # local-batch mean -> global-batch mean
        for bi, n in enumerate(self._pending):
            if n > 0:
                self._pending[bi] = 0
                self._launch(bi)
        for bi, work in self._works:
            work.wait()
            flat, views = self._flat[bi]
            flat.div_(self.world)
            dst, src = [], []
            for p, g in zip(self.buckets[bi], views):
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    dst.append(p.grad)
                    src.append(g)
            if dst:
                torch._foreach_copy_(dst, src)  # one multi-tensor launch per bucket instead of one copy per parameter
        self.reset()


def broadcast_module_state(module, src=0, group=None, sharded_names=SHARDED_PARAM_NAMES):
    """parameters (except the sharded batch-shaped ones) and buffers <- rank `src`."""
    if not is_dist():
        return
    with torch.no_grad():
        for n, p in module.named_parameters():
            if "." not in n and n in sharded_names:
                continue
            dist.broadcast(p.data, src=src, group=group)
        for _, b in module.named_buffers():
            if b.dtype == torch.uint8:  # ActNorm.initialized flags
                t = b.to(torch.int32)
                dist.broadcast(t, src=src, group=group)
                b.copy_(t.to(torch.uint8))
            else:
                dist.broadcast(b, src=src, group=group)


def gather_sharded_state(module, group=None, sharded_names=SHARDED_PARAM_NAMES):
    """state_dict of `module` with the batch-sharded initial states concatenated over ranks (rank order = row order of
    the global batch).  Collective; every rank gets the full dict (only rank 0 writes it)."""
    sd = module.state_dict()
    if not is_dist():
        return sd
    world = dist.get_world_size(group)
    out = {}
    for k, v in sd.items():
        if "." not in k and k in sharded_names:
            parts = [torch.empty_like(v) for _ in range(world)]
            dist.all_gather(parts, v.contiguous(), group=group)
            out[k] = torch.cat(parts, 0)
        else:
            out[k] = v
    return out


def load_sharded_state(module, state_dict, group=None, sharded_names=SHARDED_PARAM_NAMES):
    """load_state_dict where a batch-sharded initial state in the file holds the GLOBAL rows: each rank takes its own"""
    if is_dist():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        own = dict(module.named_parameters())
        sd = {}
        for k, v in state_dict.items():
            if "." not in k and k in sharded_names and k in own and v.shape[0] == own[k].shape[0] * world:
                b = own[k].shape[0]
                v = v[rank * b:(rank + 1) * b]
            sd[k] = v
        state_dict = sd
    module.load_state_dict(state_dict)


def all_reduce_mean_scalars(*vals, group=None):
    """average a few 0-d loss tensors over ranks for logging."""
    if not is_dist():
        return [float(v) for v in vals]
    t = torch.stack([v.detach().float().reshape(()) for v in vals])
    t = t.to(collective_device(group))   # host scalars cannot go through an RCCL-only group
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t /= dist.get_world_size(group)
    return [float(x) for x in t]


def _sharded_param_indices(module, sharded_names=SHARDED_PARAM_NAMES):
    """positions, in module.parameters() order (= the optimizer's parameter indices), of the batch-sharded states"""
    return [i for i, (n, _) in enumerate(module.named_parameters()) if "." not in n and n in sharded_names]


def gather_sharded_optimizer_state(optimizer, module, group=None, sharded_names=SHARDED_PARAM_NAMES):
    """optimizer.state_dict() with the Adam moments of the batch-sharded initial states concatenated over ranks, like
    `gather_sharded_state` does for the parameters themselves: the file then describes ONE process on the global batch
    (rows in rank order).  Collective; every rank gets the dict."""
    sd = optimizer.state_dict()
    if not is_dist():
        return sd
    world = dist.get_world_size(group)
    dev = collective_device(group)
    state = dict(sd["state"])
    for i in _sharded_param_indices(module, sharded_names):
        st = state.get(i)
        # every rank must enter the same collectives: a rank whose state has not been created yet contributes zeros
        p = list(module.parameters())[i]
        new = dict(st) if st is not None else {}
        for k in ("exp_avg", "exp_avg_sq"):
            v = (st[k] if st is not None and k in st else torch.zeros_like(p)).detach().to(dev).contiguous()
            parts = [torch.empty_like(v) for _ in range(world)]
            dist.all_gather(parts, v, group=group)
            new[k] = torch.cat(parts, 0).to(p.device)
        if st is not None:
            state[i] = new
    out = dict(sd)
    out["state"] = state
    return out


def shard_optimizer_state(state_dict, module, group=None, sharded_names=SHARDED_PARAM_NAMES):
    """the inverse on load: moments saved for the GLOBAL rows of a batch-sharded state are cut to this rank's rows.
    Moments whose leading dimension fits neither the local nor the global batch are dropped (they restart from zero)
    rather than handed to the kernel with a wrong size."""
    world = dist.get_world_size(group) if is_dist() else 1
    rank = dist.get_rank(group) if is_dist() else 0
    params = list(module.parameters())
    state = dict(state_dict["state"])
    for i in _sharded_param_indices(module, sharded_names):
        st = state.get(i)
        if st is None:
            continue
        b = params[i].shape[0]
        new = dict(st)
        ok = True
        for k in ("exp_avg", "exp_avg_sq"):
            v = st.get(k)
            if v is None:
                continue
            if v.shape[0] == b * world and tuple(v.shape[1:]) == tuple(params[i].shape[1:]):
                new[k] = v[rank * b:(rank + 1) * b].clone()
            elif tuple(v.shape) != tuple(params[i].shape):
                ok = False
        if ok:
            state[i] = new
        else:
            del state[i]
    out = dict(state_dict)
    out["state"] = state
    return out


# ---- synchronised BatchNorm --------------------------------------------------------------------
# Off by default: the bench replays a captured hipGraph per rank and a collective per BatchNorm layer (2 x 20 per step)
# cannot live inside it; with local statistics the data-parallel run is a documented deviation from "one process on the
# global batch".  RFN_SYNC_BN=1 (or set_sync_batchnorm(True)) makes every per-step BatchNorm of the extractor / upscaler
# use the global batch's statistics: eager launches only (Solver.capture_graph refuses), exact parity with a single
# process (tests/test_hip_modules.py::test_data_parallel_rfn_equals_single_process_global_batch).
_SYNC_BN = None


def set_sync_batchnorm(on):
    global _SYNC_BN
    _SYNC_BN = bool(on)


def sync_batchnorm_on():
    import os
    on = _SYNC_BN if _SYNC_BN is not None else os.environ.get("RFN_SYNC_BN") == "1"
    return bool(on) and is_dist()


def all_gather_cat(t, group=None):
    """[1, ...] per rank -> [world, ...] (rank order), through the group's device"""
    dev = collective_device(group)
    src = t.detach().to(dev).contiguous()
    if src.is_cuda and dist.get_backend(group) == "gloo":
        torch.cuda.current_stream().synchronize()
    parts = [torch.empty_like(src) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, src, group=group)
    return torch.cat(parts, 0).to(t.device)


def all_reduce_sum_(t, group=None):
    """in-place sum over ranks of a device tensor"""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        torch.cuda.current_stream().synchronize()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
