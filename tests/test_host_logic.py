"""CPU-only tests of the host side: the C-ABI library loads and exports every symbol include/rfn_hip.h declares (with
the argument lists the ctypes binding assumes), the drop-in modules build with the reference's state_dict layout,
trainer arithmetic, CLI flags, the synthetic data contract, loud failure without a GPU, and the data-parallel
gradient reducer over a 2-rank gloo group."""
import os
import json
import re
import subprocess
import sys
from argparse import Namespace

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _parse_header():
    txt = open(os.path.join(ROOT, "include", "rfn_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"(?:int|long|const char\*)\s+(rfn_\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        protos[name] = [] if args in ("void", "") else [a.strip() for a in args.split(",")]
    return protos


def test_c_abi_exports_every_declared_symbol():
    import ctypes
    from rfn_hip import lib
    protos = _parse_header()
    assert len(protos) >= 20
    L = lib.load()
    cmap = {"const float*": ctypes.c_void_p, "float*": ctypes.c_void_p, "const void*": ctypes.c_void_p, "long": ctypes.c_long, "int": ctypes.c_int,
            "float": ctypes.c_float, "rfn_stream_t": ctypes.c_void_p,
            "const float* const*": ctypes.c_void_p, "float* const*": ctypes.c_void_p,
            "long long*": ctypes.c_void_p, "const rfn_adam_entry*": ctypes.c_void_p, "const int*": ctypes.c_void_p,
            "double": ctypes.c_double}  # host arrays of device pointers
    for name, args in protos.items():
        assert hasattr(L, name), "librfn_hip.so does not export %s" % name
        assert name in lib.SIGNATURES, "ctypes binding lacks %s" % name
        want = []
        for a in args:
            ty = a.rsplit(" ", 1)[0].strip() if not a.endswith("*") else a
            ty = re.sub(r"\s+", " ", ty)
            ty = ty.replace("float *", "float*")
            want.append(cmap[ty])
        assert want == lib.SIGNATURES[name], "%s: header %s vs binding %s" % (name, want, lib.SIGNATURES[name])
    assert set(lib.SIGNATURES) == set(protos), set(lib.SIGNATURES) ^ set(protos)
    assert L.rfn_abi_version() == 1


def test_product_fails_loudly_without_gpu_tensors():
    from rfn_hip import ops
    with pytest.raises(RuntimeError, match="device tensors"):
        ops.squeeze2d_raw(torch.zeros(1, 1, 2, 2))
    from Flow import Squeeze2d
    with pytest.raises(RuntimeError):
        Squeeze2d()(torch.zeros(1, 1, 4, 4), undo_squeeze=False)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "recurrent-flows-msc_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith(".py"):
                src = open(os.path.join(dp, fn)).read()
                assert "oracle" not in src.replace("rfn_oracle", "oracle") or "import" not in "".join(
                    l for l in src.splitlines() if "oracle" in l), fn


def test_state_dict_layout_matches_reference_fixture(golden):
    from RFN import RFN
    fx = golden("rfn_loss.pt")
    for name, f in fx.items():
        m = RFN(Namespace(**f["args"]))
        missing, unexpected = m.load_state_dict(f["sd"], strict=False)
        assert not missing and not unexpected, (name, missing, unexpected)
        for k, v in m.state_dict().items():
            assert tuple(v.shape) == tuple(f["sd"][k].shape), k


def test_canonical_config_parameter_count():
    import main_rfn
    from RFN import RFN
    args = main_rfn.build_parser().parse_args(main_rfn.canonical_smmnist_argv(32, 20))
    m = RFN(args)
    assert len(m.state_dict()) == 1269            # 1269 entries at the canonical config
    assert sum(p.numel() for p in m.parameters()) == 36498112
    assert args.extractor_structure[0] == [16, 16, "pool", 32] and args.upscaler_structure[1] == ["upsample", 128, 128]


def test_trainer_arithmetic_matches_reference(golden):
    import main_rfn
    from RFN.trainer import Solver
    f = golden("trainer.pt")
    for nb in (5, 8):
        for rng in ("0.5", "1.0"):
            a = main_rfn.build_parser().parse_args(main_rfn.canonical_smmnist_argv(2, 4) +
                                                   ["--n_bits", str(nb), "--preprocess_range", rng])
            s = Solver(a)
            g = f["preprocess_%d_%s" % (nb, rng)]
            assert torch.equal(s.preprocess(g["x"]), g["y"])
            assert torch.equal(s.preprocess(g["y"], reverse=True), g["y_back"])
    g = f["compute_loss"]
    s = Solver(main_rfn.build_parser().parse_args(main_rfn.canonical_smmnist_argv(2, 4)))
    s.beta = g["beta"]
    loss = s.compute_loss(g["nll"], g["kl_fb"], g["kl"], torch.Size(g["dims"]), t=g["t"])
    torch.testing.assert_close(loss, g["loss"])
    assert abs(s.bits[-1] - g["bits"]) < 1e-6 * abs(g["bits"])
    assert abs(s.losses[-1] - g["losses"]) < 1e-5 and abs(s.kl_loss[-1] - g["kl_loss"]) < 1e-6


def test_synthetic_data_contract():
    from data_generators import SyntheticMovingMNIST
    d = SyntheticMovingMNIST(seq_len=7, seed=3)
    x = d[5]
    assert x.shape == (7, 1, 64, 64) and x.dtype == torch.float32
    assert float(x.min()) >= 0.0 and float(x.max()) <= 1.0 and float(x.max()) > 0.5
    assert torch.equal(x, d[5]) and not torch.equal(x, d[6])
    assert not torch.equal(x[0], x[3])  # things move


_DP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[1], "recurrent-flows-msc_amd"))
from rfn_hip import dist as rdist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.manual_seed(100 + rank)          # different init per rank: broadcast must fix it

class Toy(torch.nn.Module):
    def __init__(self, B):
        super().__init__()
        self.h_0 = torch.nn.Parameter(torch.randn(B, 3))            # batch-shaped, sharded (never reduced)
        self.lin = torch.nn.Linear(3, 4)
        self.big = torch.nn.Parameter(torch.randn(300, 70))         # forces several buckets
        self.unused = torch.nn.Parameter(torch.randn(5))            # never gets a gradient
        self.register_buffer("initialized", torch.tensor(rank, dtype=torch.uint8))
    def forward(self, x):
        return ((self.lin(x + self.h_0) ** 2).sum(1) + (self.big ** 2).sum() * x.mean(1)).mean()

B_local = 2
m = Toy(B_local)
rdist.broadcast_module_state(m)
assert int(m.initialized) == 0
red = rdist.GradBucketReducer(list(m.named_parameters()), bucket_bytes=64)
assert len(red.buckets) >= 2
g = torch.Generator().manual_seed(7)
X = torch.randn(world * B_local, 3, generator=g)
H = torch.randn(world * B_local, 3, generator=g)
with torch.no_grad():
    m.h_0.copy_(H[rank * B_local:(rank + 1) * B_local])
for step in range(2):
    m.zero_grad(set_to_none=True)
    m(X[rank * B_local:(rank + 1) * B_local]).backward()
    red.finish()
# single-process reference on the global batch
torch.manual_seed(100)
ref = Toy(world * B_local)
with torch.no_grad():
    for (n, p), (_, q) in zip(ref.named_parameters(), m.named_parameters()):
        if n != "h_0":
            p.copy_(q)
    ref.h_0.copy_(H)
ref(X).backward()
for (n, p), (_, q) in zip(ref.named_parameters(), m.named_parameters()):
    if n == "h_0":
        # sharded rows: the local-batch mean is rescaled to the global-batch mean (same as the single process)
        torch.testing.assert_close(q.grad, p.grad[rank * B_local:(rank + 1) * B_local], rtol=1e-5, atol=1e-6)
    elif n == "unused":
        assert q.grad is None or float(q.grad.abs().sum()) == 0.0
    else:
        torch.testing.assert_close(q.grad, p.grad, rtol=1e-5, atol=1e-6)
vals = rdist.all_reduce_mean_scalars(torch.tensor(float(rank)), torch.tensor(2.0))
assert abs(vals[0] - (world - 1) / 2) < 1e-6 and abs(vals[1] - 2.0) < 1e-6
# a stop decision taken by ONE rank reaches all of them (Solver.train: stop flag = mean over ranks > 0)
stop = rdist.all_reduce_mean_scalars(torch.tensor(1.0 if rank == 1 else 0.0))[0] > 0.0
assert stop
# checkpoints hold the GLOBAL rows of the sharded initial states; loading gives every rank its own rows back
full = rdist.gather_sharded_state(m)
assert tuple(full["h_0"].shape) == (world * B_local, 3)
assert torch.equal(full["h_0"], H)
with torch.no_grad():
    m.h_0.zero_()
rdist.load_sharded_state(m, full)
assert torch.equal(m.h_0.detach(), H[rank * B_local:(rank + 1) * B_local])
# the Adam moments of the sharded rows travel the same way (ADVICE r2: a checkpoint must describe ONE process on the
# global batch, and a loaded moment must have its parameter's size): gather -> global rows, shard -> my rows back
opt = torch.optim.Adam(m.parameters(), lr=1e-2)
opt.step()
osd = rdist.gather_sharded_optimizer_state(opt, m)
names = [n for n, _ in m.named_parameters()]
i_h = names.index("h_0")
assert tuple(osd["state"][i_h]["exp_avg"].shape) == (world * B_local, 3)
parts = [torch.empty_like(opt.state[m.h_0]["exp_avg"]) for _ in range(world)]
dist.all_gather(parts, opt.state[m.h_0]["exp_avg"])
assert torch.equal(osd["state"][i_h]["exp_avg"], torch.cat(parts, 0))
assert torch.equal(osd["state"][names.index("big")]["exp_avg"], opt.state[m.big]["exp_avg"])   # replicated: untouched
back = rdist.shard_optimizer_state(osd, m)
assert torch.equal(back["state"][i_h]["exp_avg"], opt.state[m.h_0]["exp_avg"])
assert torch.equal(back["state"][i_h]["exp_avg_sq"], opt.state[m.h_0]["exp_avg_sq"])
opt2 = torch.optim.Adam(m.parameters(), lr=1e-2)
opt2.load_state_dict(back)
assert opt2.state[m.h_0]["exp_avg"].shape == m.h_0.shape
# moments that fit neither the local nor the global batch are dropped, not handed on with a wrong size
bad = {"state": {i_h: {"step": torch.tensor(1.0), "exp_avg": torch.zeros(7, 3), "exp_avg_sq": torch.zeros(7, 3)}},
       "param_groups": osd["param_groups"]}
assert i_h not in rdist.shard_optimizer_state(bad, m)["state"]
assert rdist.collective_device().type == "cpu"   # gloo group: host tensors
dist.destroy_process_group()
print("rank %d ok" % rank)
"""


def test_data_parallel_reducer_world2_gloo(tmp_path):
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "rank %d ok" % r in o


def test_graph_capture_guard_logic():
    """hipGraph replays are only trusted when DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 preceded HIP initialisation
    (rfn_hip/__init__.py): started with it -> safe; set by an entry point before the runtime came up -> safe; set (or
    still unset) when the runtime was already up at import -> refuse."""
    import rfn_hip
    f = rfn_hip._capture_safe
    assert f("0", "0", True) and f("0", "0", False)
    assert f(None, "0", False)
    assert not f(None, "0", True)          # an integrator touched torch.cuda first, then imported the package
    assert f(None, "0", True, True)        # ... unless an entry point set it before importing torch and says so
    assert not f(None, "1", True, True)
    assert not f(None, None, False) and not f("1", "1", False)
    assert rfn_hip.graph_capture_safe() in (True, False)
    src = open(os.path.join(ROOT, "recurrent-flows-msc_amd", "rfn_hip", "__init__.py")).read()
    assert "setdefault" not in src         # the package must not set the flag itself


def test_convlstm_state_dict_keys_do_not_change_after_forward_bookkeeping():
    """the lazily 'created' peephole tensors are not registered, so a checkpoint written after training loads into a model
    that has not run yet (and into the reference's GPU model); reference CPU checkpoints that carry them still load."""
    from Utils import ConvLSTM
    m = ConvLSTM(in_channels=4, hidden_channels=3, kernel_size=[3, 3], bias=True)
    keys0 = set(m.state_dict())
    m.LSTMlayer._peephole_tensors(2, 2, torch.device("cpu"))
    assert set(m.state_dict()) == keys0
    sd = dict(m.state_dict())
    for n in ("Wci", "Wcf", "Wco"):
        sd["LSTMlayer." + n] = torch.zeros(1, 3, 2, 2)
    m2 = ConvLSTM(in_channels=4, hidden_channels=3, kernel_size=[3, 3], bias=True)
    m2.load_state_dict(sd, strict=True)
    assert "LSTMlayer.Wci" in m2.state_dict()


def test_checkpoint_file_is_read_without_executing_it(tmp_path):
    """Solver.read_checkpoint: weights_only load with argparse.Namespace as the only allow-listed class"""
    from argparse import Namespace
    from RFN.trainer import Solver
    f = tmp_path / "rfn.pt"
    torch.save({"epoch": 3, "loss": 1.5, "args": Namespace(K=2, x_dim=[1, 1, 8, 8], structure=[[4, "pool", 8]]),
                "model_state_dict": {"w": torch.ones(2)}, "losses": [1.0, 2.0]}, f)
    ck = Solver.read_checkpoint(str(f))
    assert ck["args"].K == 2 and ck["args"].structure == [[4, "pool", 8]] and ck["epoch"] == 3

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    torch.save({"args": Evil()}, f)
    with pytest.raises(Exception):
        Solver.read_checkpoint(str(f))


def test_per_step_batchnorm_equals_sequential_calls():
    """time-batched BatchNorm with per-step statistics == one nn.BatchNorm2d call per step (outputs, grads, EMA)."""
    from Utils.modules import per_step_batchnorm
    g = torch.Generator().manual_seed(0)
    S, B, C = 5, 3, 4
    x = torch.randn(S * B, C, 6, 6, generator=g) * 2 + 1
    bn_a, bn_b = torch.nn.BatchNorm2d(C), torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn_a.weight.uniform_(0.5, 1.5); bn_a.bias.uniform_(-1, 1)
        bn_b.load_state_dict(bn_a.state_dict())
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    ya = torch.cat([bn_a(xa[t * B:(t + 1) * B]) for t in range(S)])
    yb = per_step_batchnorm(bn_b, xb, S)
    torch.testing.assert_close(yb, ya, rtol=1e-5, atol=1e-5)
    w = torch.randn(ya.shape, generator=g)
    (ya * w).sum().backward()
    (yb * w).sum().backward()
    torch.testing.assert_close(xb.grad, xa.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(bn_b.weight.grad, bn_a.weight.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(bn_b.running_mean, bn_a.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn_b.running_var, bn_a.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn_b.num_batches_tracked) == int(bn_a.num_batches_tracked) == S


def test_recurrent_param_net_deferred_weight_gradients():
    """SimpleParamNet.recurrent(): per-step calls whose weight gradients are computed once, time-batched, must give the
    same outputs and gradients as calling the module every step (RFN_new.py:167-179 calls prior/encoder per timestep)."""
    import copy
    from Utils.modules import SimpleParamNet
    torch.manual_seed(0)
    net = SimpleParamNet([12, 10], in_channels=7, out_channels=3, norm_type="none", non_lin="leakyrelu").double()
    ref = copy.deepcopy(net)
    x0 = torch.randn(2, 4, 2, 2, dtype=torch.double, requires_grad=True)
    x1 = x0.detach().clone().requires_grad_(True)
    h = [torch.randn(2, 3, 2, 2, dtype=torch.double) for _ in range(5)]

    def roll(call, x):
        out, z = 0, x
        for t in range(5):
            r = call(torch.cat((z, h[t]), 1))
            out = out + (r * (t + 1)).sum()
            z = torch.tanh(r[:, :4])
        return out
    f = net.recurrent(force=True)
    assert f != net.raw
    a, b = roll(f, x0), roll(ref.raw, x1)
    assert torch.allclose(a, b, rtol=1e-12)
    a.backward(); b.backward()
    assert torch.allclose(x0.grad, x1.grad, rtol=1e-10, atol=1e-12)
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-10, atol=1e-12), n
    # norm layers other than "none" keep the plain path
    bn = SimpleParamNet([8], in_channels=4, out_channels=2, norm_type="batchnorm")
    assert bn.recurrent(force=True) == bn.raw


def test_bench_roofline_groups_by_kernel_symbol():
    """bench.kernel_roofline: the roofline line is per kernel SYMBOL (epilogue variants such as "+actbwd" are merged, as
    rocprofv3 --stats groups them), the dominant kernel is the one with the most time among the MFMA kernels, and the
    fractions are algorithmic bytes / FLOPs over time against the 8 TB/s and split-precision MFMA roofs."""
    import importlib
    bench = importlib.import_module("bench")

    class Ev:
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t

    def rec(name, kind, sym, flops, nbytes, ms, shape="s"):
        return (name, (kind, sym, flops, shape, nbytes), Ev(0.0), Ev(ms))
    records = []
    for _ in range(2):  # two "steps"
        records += [rec("rfn_conv2d_fwd_bf16x3", "conv", "kernA", 1e12, 2e9, 1.0),
                    rec("rfn_conv2d_dgrad_act_bf16x3", "conv", "kernA+actbwd", 1e12, 4e9, 1.5),
                    rec("rfn_gemm_wgrad_bf16x3", "wgrad", "kernB<1,2>", 1.5e12, 0.5e9, 1.0),
                    rec("rfn_gemm_wgrad_bf16x3", "wgrad", "kernB<grouped 3,4>", 1.5e12, 0.5e9, 1.0),
                    rec("rfn_conv2d_fwd_bf16x6", "conv", "kernB<1,2> x6", 1e12, 1e9, 0.3),   # other arithmetic: apart
                    ("rfn_squeeze2d_f32", None, Ev(0.0), Ev(0.25))]
    roof, table = bench.kernel_roofline(records, 2)
    # instantiations of one template are ONE kernel: kernB = 2.0 ms/step, kernA (with its +actbwd mode) 2.5
    assert roof["kernel"] == "kernA" and roof["launches_per_step"] == 2
    assert abs(roof["avg_launch_us"] - 1250.0) < 1e-6
    assert abs(roof["hbm"]["achieved_GBps"] - 6e9 / 2.5e-3 / 1e9) < 1e-6      # (2e9 + 4e9) B in 2.5 ms
    assert abs(roof["mfma"]["achieved_TFLOPs_fp32_equiv"] - 2e12 / 2.5e-3 / 1e12) < 1e-6
    assert roof["bound"] in ("hbm", "mfma") and 0 < roof["frac"] <= 1.5
    assert abs(roof["hip_kernel_ms_per_step"] - 5.05) < 1e-9
    names = [r[0] for r in table["kernels"]]
    assert "kernA" in names and "kernA+actbwd" in names and "rfn_squeeze2d_f32" in names and "kernB<1,2>" in names
    # with the template's instantiations summed it can become the dominant kernel
    records += [rec("rfn_gemm_wgrad_bf16x3", "wgrad", "kernB<9>", 1e12, 1e9, 1.2) for _ in range(2)]
    roof2, _ = bench.kernel_roofline(records, 2)
    assert roof2["kernel"] == "kernB" and roof2["launches_per_step"] == 3


_RANK_STUB = """
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
mode = sys.argv[1]
if mode == "ok":
    if rank == 0:
        print("noise line")
        print(json.dumps({"n_gpus": world, "argv": sys.argv[2:], "port": os.environ["MASTER_PORT"]}))
    sys.exit(0)
if mode == "fail1":          # rank 1 dies; the others would wait for a rendezvous for ever
    if rank == 1:
        sys.exit(7)
    time.sleep(120)
"""


def test_bench_self_launches_its_ranks(tmp_path, capsys):
    """`python bench.py --gpus N` without a launcher: the GPU-less parent starts N ranks with the torchrun environment,
    passes rank 0's JSON line through, and turns a failing rank into a non-zero exit code without hanging (VERDICT r2
    item 2; the rank body is a stub here -- the real one needs GPUs)."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    stub = tmp_path / "rank_stub.py"
    stub.write_text(_RANK_STUB)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    rc = bench.launch_ranks(3, ["ok", "--gpus", "3"], env=env, script=str(stub))
    out = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert rc == 0 and len(out) == 1
    rec = json.loads(out[0])
    assert rec["n_gpus"] == 3 and rec["argv"] == ["--gpus", "3"]
    t0 = time.time()
    rc = bench.launch_ranks(2, ["fail1"], env=env, script=str(stub))
    assert rc == 7 and time.time() - t0 < 60
    assert bench._gpus_arg(["--steps", "3", "--gpus", "4"]) == 4 and bench._gpus_arg(["--gpus=2"]) == 2
    assert bench._gpus_arg(["--steps", "3"]) == 1


def test_checkpoint_args_follow_the_world_size():
    """a data-parallel checkpoint stores the per-rank batch in `args` and the global batch beside it; resuming on
    another number of ranks (or on one process) rebuilds the Solver with the global batch divided over the new ranks
    (the file must describe ONE process on the global batch)."""
    from RFN.trainer import Solver
    args = Namespace(batch_size=4, x_dim=[4, 1, 64, 64], condition_dim=[4, 1, 64, 64], multigpu=True)
    ckpt = {"args": args, "global_batch_size": 32, "world_size": 8}
    a1 = Solver.args_for_world(ckpt, 1)
    assert a1.batch_size == 32 and a1.x_dim == [32, 1, 64, 64] and a1.condition_dim[0] == 32
    a2 = Solver.args_for_world(ckpt, 2)
    assert a2.batch_size == 16 and a2.x_dim[0] == 16
    assert args.batch_size == 4 and args.x_dim[0] == 4          # the stored Namespace is not modified
    with pytest.raises(ValueError):
        Solver.args_for_world(ckpt, 5)
    # a reference checkpoint (single process, no global_batch_size entry): the stored batch is the global one
    assert Solver.args_for_world({"args": args}, 2).batch_size == 2
