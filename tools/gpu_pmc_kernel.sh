#!/usr/bin/env bash
# GPU box: PMC passes restricted to the kernels matching KRE while running CMD; per-dispatch counters summarised into gpurun_out/pmc_<TAG>/summary.txt
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"
TAG="${TAG:-k}"; KRE="${KRE:-bottom_up}"
CMD="${CMD:-python $R/bench_traversal.py --scale 24 --roots 4 --no-sssp --no-cpu-baseline}"
P="$O/pmc_$TAG"; rm -rf "$P"; mkdir -p "$P"
cd /tmp && export TMPDIR=/tmp
i=0
for set in ${PMC_SETS:-"GRBM_GUI_ACTIVE:SQ_WAVES:SQ_BUSY_CYCLES:SQ_WAVE_CYCLES:SQ_WAIT_INST_ANY:SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD:SQ_INSTS_VALU:SQ_INSTS_SALU:SQ_INSTS_LDS:SQ_ACTIVE_INST_VMEM:SQ_WAIT_ANY" "FETCH_SIZE:TCC_HIT_sum:TCC_MISS_sum:TCC_REQ_sum" "TCP_TCC_READ_REQ_sum:TCP_TOTAL_CACHE_ACCESSES_sum:TCP_PENDING_STALL_CYCLES_sum:TCP_TCC_READ_REQ_LATENCY_sum" "WRITE_SIZE:TCC_EA_RDREQ_sum:TCC_EA_RDREQ_32B_sum"}; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $(echo $set | tr ':' ' ') --kernel-trace --kernel-include-regex "$KRE" -d "$P/pmc$i" -o run -- $CMD > "$P/pmc$i.log" 2>&1 || echo "pmc set failed: $set" >> "$P/fail.log"
done
python "$R/tools/rocpd_summary.py" "$P" --per-dispatch "$KRE" > "$P/summary.txt" 2>&1
find "$P" -name "*.db" -delete
cat "$P/summary.txt" | head -150; cat "$P/fail.log" 2>/dev/null
