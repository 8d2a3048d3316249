#!/usr/bin/env bash
# round 6: Louvain with the reference's numbering of contracted levels (degree order) and the sweep-local direction flip: every Louvain test but the
# RMAT fixtures (regenerated separately), single GPU and partitioned
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r6ae}
timeout 1500 python -m pytest tests -m gpu -q -k "louvain and not rmat_golden and not rmat22_golden" 2>&1 | tail -15 | tee "$O/${TAG}_pytest_louvain.log"
