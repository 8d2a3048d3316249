#!/usr/bin/env bash
# round 6, session n: PMC passes on k_sssp_sweep: the kernel as shipped against its streamed part alone (-DCGA_ABL_SWEEP_NODRAIN): what the drains add besides their own time
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; cd "$R"
for l in new sweep_nodrain; do
  cp gpurun_libs/$l.so cugraph_amd/lib/libcugraph_c.so
  TAG=sweep_$l KRE="k_sssp_sweep" CMD="python $R/bench_traversal.py --scale 24 --roots 1 --weights int --no-cpu-baseline --no-check --single-variant" \
    PMC_SETS="GRBM_GUI_ACTIVE:SQ_WAVES:SQ_BUSY_CYCLES:SQ_WAVE_CYCLES:SQ_WAIT_INST_ANY:SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD:SQ_INSTS_VMEM_WR:SQ_INSTS_VALU:SQ_INSTS_LDS:SQ_WAIT_ANY:SQ_INSTS_SALU FETCH_SIZE:TCC_HIT_sum:TCC_MISS_sum:TCC_REQ_sum WRITE_SIZE:TCC_EA0_RDREQ_sum:TCC_EA0_WRREQ_sum:TCC_ATOMIC_sum TCP_TCC_READ_REQ_sum:TCP_TOTAL_CACHE_ACCESSES_sum:TCP_PENDING_STALL_CYCLES_sum:TCP_TCC_ATOMIC_WITH_RET_REQ_sum" \
    bash tools/gpu_pmc_kernel.sh > "$O/r6n_pmc_$l.txt" 2>&1
done
cp gpurun_libs/new.so cugraph_amd/lib/libcugraph_c.so
