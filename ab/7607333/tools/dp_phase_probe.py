"""Developer probe: per-phase wall time of the graphed train step under torch.distributed (2 ranks, gloo, one GPU)."""
import os, sys, time, datetime
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench
dist.init_process_group(os.environ.get("RFN_DIST_BACKEND", "gloo"), timeout=datetime.timedelta(seconds=300))
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
B = 32 // world
solver, args = bench.build_solver(B, 20, dev)
batches = [bench.make_batch(B, 20, 100 + rank * 17 + i, dev) for i in range(2)]
def T(): torch.cuda.synchronize(); return time.time()
for i in range(3):
    t = T(); solver.train_step(batches[i % 2]); 
    if rank == 0: print("eager step %d: %.3f s" % (i, T() - t), flush=True)
ok = solver.capture_graph(batches[0])
if rank == 0: print("capture", ok, getattr(solver, "_graph_error", ""), flush=True)
for i in range(3):
    t0 = T(); solver._g_in.copy_(batches[i % 2]); solver._g_beta.fill_(solver.beta); solver._graph.replay(); t1 = T()
    solver.reducer.finish(); t2 = T()
    solver.optimizer.step(); t3 = T()
    if rank == 0: print("graph step %d: replay %.3f  reduce %.3f  adam %.3f" % (i, t1 - t0, t2 - t1, t3 - t2), flush=True)
