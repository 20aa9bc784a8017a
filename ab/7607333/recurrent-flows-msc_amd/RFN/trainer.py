"""Solver — training driver with the reference's surface (RFN/trainer.py of the reference): `Solver(args).build();
.train(); .load(ckpt)`, `preprocess`, `compute_loss` (bits/dim bookkeeping), β annealing, linear LR decay, checkpoint
dict layout.  Plotting (matplotlib PNG panels) and the file-backed datasets are outside the hot-path scope; a synthetic
SM-MNIST-shaped loader is built in (`--synthetic_data`).  Multi-GPU = one process per GPU (rfn_hip/dist.py)."""
import math
import os

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader

from data_generators import SyntheticMovingMNIST
from rfn_hip import dist as rdist
from rfn_hip import ops as K
from Utils import set_gpu
from .RFN_new import RFN


class EarlyStopping:
    """RFN/trainer.py:18-44."""

    def __init__(self, min_delta=0, patience=50, verbose=True):
        self.min_delta, self.patience, self.verbose = min_delta, patience, verbose
        self.wait, self.best_loss, self.stop_training = 0, 1e15, False

    def step(self, epoch, loss):
        if loss is None:
            return False
        if (loss - self.best_loss) < -self.min_delta:
            self.best_loss, self.wait = loss, 1
            return False
        if self.wait >= self.patience:
            self.stop_training = True
            if self.verbose:
                print("STOP! Criterion met at epoch %d" % epoch)
            return True
        self.wait += 1
        return False


class Solver(object):
    def __init__(self, args):
        self.args = args
        for k in ("n_bits", "n_epochs", "learning_rate", "verbose", "batch_size", "patience_lr", "factor_lr", "min_lr",
                  "patience_es", "beta_max", "beta_min", "beta_steps", "choose_data", "n_frames", "digit_size",
                  "step_length", "num_digits", "image_size", "preprocess_range", "preprocess_scale", "num_workers",
                  "multigpu", "n_predictions", "n_conditions", "scheduler_type", "use_validation_set"):
            setattr(self, k, getattr(args, k))
        self.path = str(os.path.abspath(os.getcwd())) + args.path
        self.plot_counter, self.epoch_i, self.counter = 0, 0, 0
        self.losses, self.kl_loss, self.recon_loss, self.bits = [], [], [], []
        self.best_loss, self.beta, self.stop = 1e15, args.beta_min, False
        self.rank = int(os.environ.get("RANK", 0))
        self.world = int(os.environ.get("WORLD_SIZE", 1))
        self.device = set_gpu(True)

    # ---------------------------------------------------------------------------------------------- setup
    def build(self):
        if self.multigpu and self.world > 1 and not dist.is_initialized():
            local = int(os.environ.get("LOCAL_RANK", 0))
            if torch.cuda.is_available():
                if os.environ.get("RFN_SINGLE_GPU"):
                    local = 0
                torch.cuda.set_device(local)
                self.device = torch.device("cuda", local)
            # (RFN_DIST_BACKEND=gloo: rehearsal with several ranks on one GPU, as in bench.py and the tests)
            dist.init_process_group(os.environ.get("RFN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo"))
        self.train_loader, self.test_loader = self.create_loaders()
        if self.rank == 0:
            os.makedirs(self.path + "png_folder", exist_ok=True)
            os.makedirs(self.path + "model_folder", exist_ok=True)
        self.model = RFN(self.args).to(self.device)
        rdist.broadcast_module_state(self.model)
        self.reducer = rdist.GradBucketReducer(list(self.model.named_parameters()))
        self.optimizer = self.make_optimizer(self.model.parameters(), self.learning_rate)
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, "min", patience=self.patience_lr,
                                                                    factor=self.factor_lr, min_lr=self.min_lr)
        self.earlystopping = EarlyStopping(min_delta=0, patience=self.patience_es, verbose=self.verbose)
        self.counter, self.stop = 0, False

    @staticmethod
    def make_optimizer(params, lr):
        """RFN/trainer.py:96: Adam with torch's defaults.  On the GPU the whole update is one launch of the HIP kernel
        (rfn_hip.optim.HipAdam, same state layout as torch.optim.Adam); CPU parameters only occur in host-logic tests."""
        params = list(params)
        if params and params[0].is_cuda:
            from rfn_hip.optim import HipAdam
            return HipAdam(params, lr=lr)
        return torch.optim.Adam(params, lr=lr)

    def create_loaders(self):
        if not getattr(self.args, "synthetic_data", False):
            raise RuntimeError("the file-backed datasets of the reference (MNIST download, BAIR, KTH) are outside this "
                               "implementation's scope; pass --synthetic_data for SM-MNIST-shaped synthetic video")
        c = self.args.x_dim[1]
        mk = lambda seed: SyntheticMovingMNIST(seq_len=self.n_frames, image_size=self.image_size,
                                               digit_size=self.digit_size, num_digits=self.num_digits,
                                               step_length=self.step_length, channels=c,
                                               seed=seed * self.world + self.rank)
        kw = dict(batch_size=self.batch_size, num_workers=self.num_workers, shuffle=True, drop_last=True)
        return DataLoader(mk(0), **kw), DataLoader(mk(1), **kw)

    # ---------------------------------------------------------------------------------------------- arithmetic
    def preprocess(self, x, reverse=False):
        """RFN/trainer.py:165-188."""
        n_bins = 2 ** self.n_bits
        if not reverse:
            x = x * self.preprocess_scale
            if self.n_bits < 8:
                x = torch.floor(x / 2 ** (8 - self.n_bits))
            x = x / n_bins
            return x - 0.5 if self.preprocess_range == "0.5" else x
        if self.preprocess_range == "0.5":
            x = x + 0.5
        x = x * n_bins
        return torch.clamp(torch.floor(x) * (256. / n_bins), 0, 255).byte()

    def adjust_learning_rate(self, batch):
        """RFN/trainer.py:190-204 — linear decay to zero over 150k steps after step 100k."""
        startbatch, num_steps = 100000, 150000
        if batch > startbatch:
            lr = self.learning_rate - (batch - startbatch) * self.learning_rate / num_steps
            for g in self.optimizer.param_groups:
                g["lr"] = lr
        if batch == (startbatch + num_steps - 5):
            self.stop = True

    def compute_loss(self, nll, kl_free_bit, kl, dims, t=10):
        """RFN/trainer.py:206-219 — loss = nll + β·kl_fb ; bits/dim = (kl+nll)/(ln2 · C·H·W · t)."""
        loss = nll + self.beta * kl_free_bit
        kl_store, nll_store = kl.detach(), nll.detach()
        bits = (kl_store + nll_store) / (np.log(2.) * float(np.prod(list(dims))) * t)
        self.bits.append(float(bits))
        self.losses.append(float(loss.detach()) / t)
        self.kl_loss.append(float(kl_store) / t)
        self.recon_loss.append(float(nll_store) / t)
        return loss

    # ---------------------------------------------------------------------------------------------- loop
    def train_step(self, image):
        """one optimizer step on an already-resident [B,T,C,H,W] batch in [0,1] (RFN/trainer.py:237-250)."""
        self.beta = min(self.beta_max, self.beta_min + self.counter * (self.beta_max - self.beta_min) / self.beta_steps)
        if self._graph is not None:
            return self._graphed_step(image)
        image = self.preprocess(image)
        first = self.counter == 0 and self.world > 1
        kl_free_bit, kl, nll = self.model.loss(image, 0)
        if first:  # replicas must share rank 0's data dependent ActNorm init: broadcast, then redo the step's forward
            rdist.broadcast_module_state(self.model)
            kl_free_bit, kl, nll = self.model.loss(image, 0)
        loss = self.compute_loss(nll, kl_free_bit, kl, image.shape[2:], t=image.shape[1] - 1)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.reducer.finish()
        self.optimizer.step()
        if self.scheduler_type == "linear":
            self.adjust_learning_rate(self.counter)
        self.counter += 1
        return loss

    # ---- hipGraph mode: preprocess + loss forward + backward of one step are captured once and replayed, which removes
    # the ~7000 per-step launches' host cost (the step is launch-bound once the batch is sharded over several GPUs).
    # Gradient all-reduce, Adam, the beta/LR schedules and the loss bookkeeping stay outside the graph.
    _graph = None

    def capture_graph(self, example_image, static_draws=None):
        """Call after a few eager steps (ActNorm initialised, MIOpen solvers chosen).  Returns True on success; on
        any capture failure the solver silently stays in eager mode.  `static_draws` (tests only) pins the noise.
        The caller must not hold tensors of an earlier eager step's autograd graph (e.g. a returned loss): their
        AccumulateGrad nodes are bound to the default stream and would pull it into the capture."""
        if not torch.cuda.is_available():
            return False
        import rfn_hip
        if rdist.sync_batchnorm_on():
            self._graph_error = "synchronised BatchNorm issues a collective per layer: eager launches only"
            return False
        if not rfn_hip.graph_capture_safe():
            # see rfn_hip/__init__.py: replays are not trustworthy with packet capture on (memset nodes race), and the
            # flag only counts when it was in the environment before the HIP runtime initialised
            self._graph_error = "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 must be exported before the HIP runtime starts"
            return False
        try:
            self._g_draws = static_draws
            self._g_in = example_image.clone()
            self._g_beta = torch.zeros((), device=example_image.device)
            self.reducer.remove_hooks()           # reductions run after the replay, on the static gradient tensors
            self.optimizer.zero_grad(set_to_none=True)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                # three warm-ups of exactly the captured callable on a side stream: MIOpen / the autograd engine still
                # initialise lazily on the 2nd-3rd execution of a backward, and doing that under capture crashes
                for _ in range(int(os.environ.get("RFN_CAPTURE_WARMUPS", "3"))):  # (developer knob: DESIGN.md §3)
                    self._graph_body()
                    self.optimizer.zero_grad(set_to_none=True)
            torch.cuda.current_stream().wait_stream(side)
            self.optimizer.zero_grad(set_to_none=True)
            K.flush_packs()   # (nothing queued outside may be launched -- and replayed -- inside the capture)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._graph_body()
            self._graph = g
            return True
        except Exception as e:  # noqa: BLE001 - any failure means "no graph", never a dead trainer
            import traceback
            self._graph = None
            self._graph_error = repr(e)[:200] + " | " + " <- ".join(
                "%s:%d %s" % (f.filename.split("/")[-1], f.lineno, f.name) for f in traceback.extract_tb(e.__traceback__)[-6:])
            torch.cuda.synchronize()
            self.reducer.add_hooks()
            self.optimizer.zero_grad(set_to_none=True)
            return False

    def _graph_body(self):
        from rfn_hip import debug as D
        D.begin()
        image = self.preprocess(self._g_in)
        D.check("image", image)
        kl_free_bit, kl, nll = self.model.loss(image, 0, draws=getattr(self, "_g_draws", None))
        loss = nll + self._g_beta * kl_free_bit
        loss.backward()
        self._g_out = torch.stack([loss.detach(), kl_free_bit.detach(), kl.detach(), nll.detach()])

    def _graphed_step(self, image):
        if os.environ.get("RFN_STEP_TRACE") == "1":  # developer aid: wall time of the phases of a replayed step
            import time
            ts = []
            def mark():
                torch.cuda.synchronize()
                ts.append(time.perf_counter())
            mark(); self._g_in.copy_(image, non_blocking=True); self._g_beta.fill_(self.beta)
            mark(); self._graph.replay()
            mark(); self.reducer.finish()
            mark(); self.optimizer.step()
            mark()
            print("[step %d rank %d] copy %.3f replay %.3f reduce %.3f adam %.3f s" % (
                self.counter, self.rank, ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3]), flush=True)
        else:
            self._g_in.copy_(image, non_blocking=True)
            self._g_beta.fill_(self.beta)
            self._graph.replay()
            self.reducer.finish()
            self.optimizer.step()
        if self.scheduler_type == "linear":
            self.adjust_learning_rate(self.counter)
        self.counter += 1
        self._pending_log = (self._g_out, tuple(image.shape))
        return self._g_out[0]

    def flush_log(self):
        """bits/dim bookkeeping of the last graphed step (one device->host read, kept off the per-step path)."""
        if getattr(self, "_pending_log", None) is None:
            return
        out, shape = self._pending_log
        loss, _, kl, nll = [float(v) for v in out.tolist()]
        t = shape[1] - 1
        self.bits.append((kl + nll) / (np.log(2.) * float(np.prod(shape[2:])) * t))
        self.losses.append(loss / t)
        self.kl_loss.append(kl / t)
        self.recon_loss.append(nll / t)
        self._pending_log = None

    def train(self):
        max_steps = getattr(self.args, "max_steps", 0)
        for _ in range(self.n_epochs):
            self.model.train()
            self.epoch_i += 1
            for image in self.train_loader:
                image = image[0] if self.choose_data == "bair" and isinstance(image, (list, tuple)) else image
                self.train_step(image.to(self.device, non_blocking=True))
                if max_steps and self.counter >= max_steps:
                    self.stop = True
                    break
            self.flush_log()   # graph mode keeps the last step's scalars on the device until asked
            epoch_loss = float(np.mean(self.losses)) if self.losses else math.nan
            # every rank must take the same decisions (checkpoint value, early stop, plateau scheduler): rank-local
            # losses would let learning rates diverge or leave one rank waiting in an all-reduce the others never enter
            # (host scalars: all_reduce_mean_scalars moves them to the GPU when the group is RCCL-only)
            epoch_loss, stop_flag = rdist.all_reduce_mean_scalars(torch.tensor(epoch_loss), torch.tensor(float(self.stop)))
            self.stop = stop_flag > 0.0
            self.checkpoint("rfn.pt", self.epoch_i, epoch_loss)   # (collective: gathers the sharded initial states)
            stop = self.earlystopping.step(self.epoch_i, epoch_loss)
            if stop or self.stop:
                break
            if self.earlystopping.best_loss < self.best_loss and self.epoch_i > 50:
                self.best_loss = self.earlystopping.best_loss
                self.checkpoint("rfn_best_model.pt", self.epoch_i, epoch_loss)
            if self.scheduler_type == "plateau":
                self.scheduler.step(epoch_loss)
            if self.verbose:
                print("Epoch {} Loss: {:.2f}".format(self.epoch_i, epoch_loss))
            elif self.rank == 0:
                self.status()

    # ---------------------------------------------------------------------------------------------- state
    def checkpoint(self, model_name, epoch, loss):
        """same dict layout as RFN/trainer.py:277-300 (model/optimizer state, histories, counters, args); `args_dict` is
        the same Namespace as a plain dict.  Collective under data parallelism: the batch-sharded initial states are
        gathered so that the file holds the GLOBAL batch rows (a single process can resume it); rank 0 writes."""
        state = rdist.gather_sharded_state(self.model)
        opt_state = rdist.gather_sharded_optimizer_state(self.optimizer, self.model)   # (moments of the sharded rows too)
        if self.rank != 0:
            return
        # `args.batch_size` is the per-rank batch; the file describes the GLOBAL batch (rows in rank order)
        common = {"epoch": epoch, "loss": loss, "kl_loss": self.kl_loss, "recon_loss": self.recon_loss,
                  "losses": self.losses, "bits_per_dim": self.bits, "annealing_counter": self.counter,
                  "args": self.args, "args_dict": dict(vars(self.args)), "world_size": self.world,
                  "global_batch_size": int(self.batch_size) * self.world}
        full = dict(common)
        full.update({"model_state_dict": state, "optimizer_state_dict": opt_state,
                     "plot_counter": self.plot_counter})
        torch.save(full, self.path + "model_folder/" + model_name)
        torch.save(common, self.path + "model_folder/eval_dict.pt")

    @staticmethod
    def args_for_world(ckpt, world):
        """the Namespace to rebuild a Solver from `ckpt` on `world` ranks: the stored `batch_size` is per rank of the
        run that wrote the file; the global batch is what is kept (files without `global_batch_size` -- the reference's
        own -- are single-process: global = stored)."""
        import copy
        args = copy.copy(ckpt["args"])
        gb = int(ckpt.get("global_batch_size", args.batch_size))
        if gb % world:
            raise ValueError("checkpoint global batch %d does not divide over %d ranks" % (gb, world))
        args.batch_size = gb // world
        for k in ("x_dim", "condition_dim"):   # [B, C, H, W] lists carry the batch too
            v = getattr(args, k, None)
            if isinstance(v, (list, tuple)) and len(v) == 4:
                setattr(args, k, [gb // world] + list(v[1:]))
        return args

    @staticmethod
    def read_checkpoint(path):
        """load an rfn.pt written by this Solver or by the reference WITHOUT executing anything from the file: tensors,
        containers and numbers plus the one class the layout needs (argparse.Namespace)."""
        import argparse
        with torch.serialization.safe_globals([argparse.Namespace]):
            return torch.load(path, map_location="cpu", weights_only=True)

    def load(self, load_model):
        rdist.load_sharded_state(self.model, load_model["model_state_dict"])
        # the file holds the moments of the GLOBAL rows of the batch-sharded initial states: every rank takes its own
        # (torch's load_state_dict does not compare shapes, and HipAdam indexes the moments by the parameter's numel)
        self.optimizer.load_state_dict(rdist.shard_optimizer_state(load_model["optimizer_state_dict"], self.model))
        self.epoch_i += load_model["epoch"]
        loss = load_model["loss"]
        self.kl_loss, self.recon_loss = load_model["kl_loss"], load_model["recon_loss"]
        self.losses, self.plot_counter = load_model["losses"], load_model["plot_counter"]
        self.counter, self.bits = load_model["annealing_counter"], load_model["bits_per_dim"]
        self.best_loss = loss
        self.model.to(self.device)
        return self.epoch_i, loss

    def status(self):
        lr = self.optimizer.param_groups[0]["lr"]
        with open(self.path + "model_folder/status.txt", "a") as f:
            print("STATUS:", file=f)
            if self.kl_loss:
                print("\tKL and Reconstruction loss: {:.4f}, {:.4f}".format(self.kl_loss[-1], self.recon_loss[-1]),
                      file=f)
            print(f"\tEpoch {self.epoch_i}, Beta value {self.beta:.4f}, Learning rate {lr}", file=f)
