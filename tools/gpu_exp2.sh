#!/usr/bin/env bash
# GPU box, round 2 experiment 2: ablations of phase 1, phase-2 tile heights, and a check of the one-rank multi-GPU plan at RMAT-26
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
LIBS="base nostore nogather nodelta loadonly p2r8k p2r16k" STEPS=20 BENCH_EXTRA="--no-check" bash tools/gpu_ab.sh > "$O/exp2_ab.log" 2>&1
cat "$O/ab.log"
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -30
import os, sys, torch
sys.path.insert(0, os.getcwd())
import cugraph_amd as cg
h = cg.ResourceHandle()
scale = 26; nv, ne = 1 << scale, 16 << scale
src, dst = cg.generate_rmat_edgelist(h, scale, ne)
s64 = src.to(torch.int64)
live = torch.zeros(nv, dtype=torch.bool, device="cuda"); live[s64] = True
print("distinct sources (scatter):", int(live.sum()))
u, inv = torch.unique(s64, sorted=True, return_inverse=True)
print("torch.unique:", int(u.numel()), "inverse ok:", bool((u[inv] == s64).all()), "sorted:", bool((u[1:] > u[:-1]).all()))
indeg = torch.zeros(nv, dtype=torch.int64, device="cuda").index_add_(0, dst.to(torch.int64), torch.ones(ne, dtype=torch.int64, device="cuda"))
bc = torch.bincount(dst.to(torch.int64), minlength=nv)
print("bincount == index_add:", bool((bc == indeg).all()))
_, order = torch.sort(indeg, descending=True, stable=True)
print("sort is a permutation:", int(torch.unique(order).numel()) == nv, "descending:", bool((indeg[order][1:] <= indeg[order][:-1]).all()))
PY
