#!/usr/bin/env bash
# GPU box: kernel-trace stats + PMC passes (separate runs) of bench.py at one scale; summaries into gpurun_out/prof_<tag>/
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"
SCALE="${SCALE:-26}"; TAG="${TAG:-r1}"; STEPS="${STEPS:-5}"
P="$O/prof_$TAG"; rm -rf "$P"; mkdir -p "$P"
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --scale $SCALE --steps $STEPS --warmup 1 --no-cpu-baseline ${EXTRA:-}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$P/stats" -o run -- $CMD > "$P/stats.log" 2>&1
i=0
for set in ${PMC_SETS:-"FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES:SQ_WAVE_CYCLES:SQ_BUSY_CYCLES:SQ_WAIT_ANY:SQ_WAIT_INST_ANY:SQ_ACTIVE_INST_ANY:SQ_ACTIVE_INST_VALU:SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU:SQ_INSTS_SALU:SQ_INSTS_LDS:SQ_INSTS_VMEM_RD:SQ_INSTS_VMEM_WR:SQ_LDS_BANK_CONFLICT:SQ_ACTIVE_INST_VMEM:SQ_ACTIVE_INST_SCA" "TCC_HIT_sum:TCC_MISS_sum:TCC_REQ_sum" "GRBM_GUI_ACTIVE:TCP_TCC_READ_REQ_sum:TCP_TCC_WRITE_REQ_sum"}; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $(echo $set | tr ':' ' ') --kernel-trace -d "$P/pmc$i" -o run -- $CMD > "$P/pmc$i.log" 2>&1 || echo "pmc set failed: $set" >> "$P/fail.log"
done
python "$R/tools/rocpd_summary.py" "$P" > "$P/summary.txt" 2>&1
find "$P" -name "*.db" -delete   # keep the text only (the databases are large)
grep -v "^cga::(anonymous namespace)::k_\(rs_\|scan\|mark\|lookup\)" "$P/summary.txt" | head -120
