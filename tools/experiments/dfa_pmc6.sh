# Round 6: counters of k_dfa on natural text (2 GiB per launch), one rocprofv3 --pmc pass per group.  usage: dfa_pmc6.sh <out-name> [tune value]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r06pmc}; mkdir -p $OUT
TUNE=${2:-0}
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" -o "$name" -- python $R/tools/experiments/dfa_tune.py 2 v=$TUNE > "$OUT/$name.log" 2>&1; echo "pass $name rc=$?"; python $R/tools/pmc_summary.py $OUT/$name "k_dfa" 2>&1 | grep -v "^  *$" | tail -40; rm -rf $OUT/$name/*/*.db; }
pass ta TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pass sq2 SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT SQ_INSTS_SMEM
