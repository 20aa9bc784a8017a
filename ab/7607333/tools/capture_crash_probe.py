#!/usr/bin/env python3
"""Developer tool: which lazily created state makes hipStreamEndCapture crash when the captured step has not been
warmed up on the capture path?  Runs `Solver.capture_graph` with RFN_CAPTURE_WARMUPS side-stream warm-ups (0..3) in a
child process each and reports whether the child survived; with AMD_LOG_LEVEL=3 the tail of the HIP API log of a dying
child is kept (what the runtime was doing when it fell over)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "recurrent-flows-msc_amd"))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch, bench
solver, args = bench.build_solver(4, 6, torch.device("cuda"))
x = bench.make_batch(4, 6, 5, "cuda")
for _ in range(int(os.environ.get("EAGER_STEPS", "2"))):
    solver.train_step(x)
torch.cuda.synchronize()
print("eager ok", flush=True)
ok = solver.capture_graph(x)
print("capture returned", ok, getattr(solver, "_graph_error", ""), flush=True)
if ok:
    solver.train_step(x); torch.cuda.synchronize(); print("replay ok", flush=True)
''' % (ROOT, ROOT)
for warm in (0, 1, 2, 3):
    for eager in (2,):
        env = dict(os.environ, RFN_CAPTURE_WARMUPS=str(warm), EAGER_STEPS=str(eager))
        if warm == 0:
            env["AMD_LOG_LEVEL"] = "3"
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
        out = [l for l in r.stdout.splitlines() if l.strip()]
        print("warmups=%d eager=%d rc=%d stdout=%s" % (warm, eager, r.returncode, out[-2:]), flush=True)
        if r.returncode != 0 or warm == 0:
            tail = r.stderr.splitlines()[-40:]
            print("  --- stderr tail ---")
            for l in tail:
                print("  " + l[:220])
