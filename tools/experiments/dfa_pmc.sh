R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05w; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; timeout 240 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" -o "$name" -- python $R/tools/experiments/dfa_probe.py natural_100k_10GiB 2 > "$OUT/$name.log" 2>&1; echo "pass $name rc=$?"; python $R/tools/pmc_summary.py $OUT/$name "k_dfa<0" 2>&1 | tail -8; }
pass ta TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass tcc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass sq SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM
