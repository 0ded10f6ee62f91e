#!/bin/bash
# Runs on the GPU box (via gpurun): PMC passes for the dominant kernel, each in its own rocprofv3 run
# (only --kernel-trace next to --pmc, as the pool requires).  Usage: tools/pmc_profile.sh <outdir> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--hay-count 2048 --steps 2 --warmup 1 --no-cpu-baseline $*"
pass() {
  name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" -o "$name" -- python "$R/bench.py" $ARGS > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
pass tcc_fetch FETCH_SIZE TCC_HIT_sum
pass tcc_write WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
pass tcc_ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_READ_sum
pass sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
pass sq_b SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
pass ta TA_BUSY_sum TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE GRBM_COUNT
