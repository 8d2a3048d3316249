"""per-level phase times of the one-rank partitioned BFS for given root indices (GPU box)"""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cugraph_amd as cg
from cugraph_amd import mg_traversal as mt
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
scale = 24
h = cg.ResourceHandle()
nv, ne = 1 << scale, 16 << scale
src, dst = cg.generate_rmat_edgelist(h, scale, ne)
t = mt.MGTraversal(src, dst, nv, None, "bfs")
outdeg = torch.bincount(src.to(torch.int64), minlength=nv)
cand = torch.nonzero(outdeg > 0).flatten().cpu()
perm = torch.randperm(cand.numel(), generator=torch.Generator().manual_seed(0))[:8]
roots = cand[perm].tolist()
times = []
for k in range(24):
    torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
    t.run([roots[k % 8]])
    torch.cuda.synchronize(); dist.barrier()
    times.append(round(1e3 * (time.perf_counter() - t0), 2))
print("24 runs over the 8 roots, ms:", times, flush=True)
e = t.engine
for name in ("expand", "apply", "bottom_up", "merge_visited", "frontier_bits", "reset", "results"):
    f = getattr(e, name)
    def wrap(f=f, name=name):
        def g(*a, **k):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize()
            print(f"    {name:14s} {1e6 * (time.perf_counter() - t0):9.1f} us  -> {r if isinstance(r, int) else ''}", flush=True)
            return r
        return g
    setattr(e, name, wrap())
ex = t._exchange
def ex2(send, counts):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = ex(send, counts); torch.cuda.synchronize()
    print(f"    exchange       {1e6 * (time.perf_counter() - t0):9.1f} us  tuples {sum(counts)}", flush=True); return r
t._exchange = ex2
for i in (0, 7, 7, 3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t.run([roots[i]])
    torch.cuda.synchronize()
    print(f"root {i} ({roots[i]}): {1e3 * (time.perf_counter() - t0):.2f} ms, levels {t.levels}, bottom-up {t.bottom_up_levels}", flush=True)
