#!/usr/bin/env bash
# Runs on the MI355X box (via gpurun): GPU tests, hot-tile sweep, kernel-trace stats, PMC passes.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out"
mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q > "$O/pytest_gpu.log" 2>&1; echo "pytest_rc=$?" >> "$O/pytest_gpu.log"
SCALE="${SCALE:-22}"
for hot in 0 4096 16384 32768 39936; do
  echo "== hot=$hot" >> "$O/sweep.log"
  timeout 120 python bench.py --scale "$SCALE" --steps 20 --warmup 3 --no-cpu-baseline --hot-tile $hot 2>/dev/null | tail -1 >> "$O/sweep.log"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$O/counters_list.txt" 2>&1 || true
rocprofv3 --kernel-trace --stats -d "$O/prof_stats" -o run -- python "$R/bench.py" --scale "$SCALE" --steps 20 --warmup 3 --no-cpu-baseline > "$O/prof_stats.log" 2>&1
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_BUSY_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo "$set" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace -d "$O/pmc_$tag" -o run -- python "$R/bench.py" --scale "$SCALE" --steps 5 --warmup 1 --no-cpu-baseline > "$O/pmc_$tag.log" 2>&1 || echo "pmc set failed: $set" >> "$O/pmc_fail.log"
done
find "$O" -name "*.csv" | head -50 > "$O/csv_files.txt"
tail -3 "$O/pytest_gpu.log"; cat "$O/sweep.log" | cut -c1-400
