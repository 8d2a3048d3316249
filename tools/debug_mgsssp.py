"""per-round phase times of the one-rank partitioned SSSP (GPU box): where its 36 ms at RMAT-24 go"""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cugraph_amd as cg
from cugraph_amd import mg_traversal as mt
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
scale = 24
h = cg.ResourceHandle()
nv, ne = 1 << scale, 16 << scale
src, dst = cg.generate_rmat_edgelist(h, scale, ne)
w = torch.randint(1, 256, (ne,), generator=torch.Generator(device="cuda").manual_seed(1), device="cuda").to(torch.float32)
t = mt.MGTraversal(src, dst, nv, w, "sssp")
outdeg = torch.bincount(src.to(torch.int64), minlength=nv)
cand = torch.nonzero(outdeg > 0).flatten().cpu()
roots = cand[torch.randperm(cand.numel(), generator=torch.Generator().manual_seed(0))[:4]].tolist()
t.run([roots[0]]); t.run([roots[1]])
acc = {}
e = t.engine
for name in ("expand", "apply", "reset", "results"):
    f = getattr(e, name)
    def wrap(f=f, name=name):
        def g(*a, **k):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize()
            dt = 1e6 * (time.perf_counter() - t0); acc[name] = acc.get(name, 0.0) + dt
            extra = f"tuples {sum(r[1])}" if name == "expand" else (f"next {r}" if name == "apply" else "")
            print(f"    {name:8s} {dt:9.1f} us  {extra}", flush=True)
            return r
        return g
    setattr(e, name, wrap())
ex = t._exchange
def ex2(send, counts):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = ex(send, counts); torch.cuda.synchronize()
    dt = 1e6 * (time.perf_counter() - t0); acc["exchange"] = acc.get("exchange", 0.0) + dt
    print(f"    exchange {dt:9.1f} us", flush=True); return r
t._exchange = ex2
torch.cuda.synchronize(); t0 = time.perf_counter()
t.run([roots[2]])
torch.cuda.synchronize()
print(f"root {roots[2]}: {1e3 * (time.perf_counter() - t0):.2f} ms, rounds {t.levels}; per phase (us): " + ", ".join(f"{k} {v:.0f}" for k, v in acc.items()), flush=True)
