R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05v; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--workload natural_100k_10GiB --hay-count 2048 --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-h2d --workloads none"
pass() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" -o "$name" -- python "$R/bench.py" $ARGS > "$OUT/$name.log" 2>&1; echo "pass $name rc=$?"; python $R/tools/pmc_summary.py $OUT/$name "k_sf<" 2>&1 | tail -12; }
pass ta TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
