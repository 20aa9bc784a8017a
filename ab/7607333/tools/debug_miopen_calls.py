"""developer probe: which convolutions of one RFN.loss call still go to MIOpen? (MIOPEN_ENABLE_LOGGING_CMD=1)"""
import os, sys
os.environ["MIOPEN_ENABLE_LOGGING_CMD"] = "1"
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recurrent-flows-msc_amd"))
import torch
import main_rfn
from RFN import RFN
B, T = 2, 4
args = main_rfn.build_parser().parse_args(main_rfn.canonical_smmnist_argv(B, T))
torch.manual_seed(71)
m = RFN(args).cuda().train()
x = (torch.rand(B, T, 1, 64, 64) - 0.5).cuda()
with torch.no_grad():
    m.loss(x, 0)
torch.cuda.synchronize()
print("==== second call (no_grad)", file=sys.stderr, flush=True)
with torch.no_grad():
    m.loss(x, 0)
torch.cuda.synchronize()
print("==== third call (grad + backward)", file=sys.stderr, flush=True)
kl_fb, kl, nll = m.loss(x, 0)
(nll + kl_fb).backward()
torch.cuda.synchronize()
print("==== done", file=sys.stderr, flush=True)
