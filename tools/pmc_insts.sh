#!/bin/bash
# dynamic instruction counts of k_sf per ablation mode (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for a in 1 3 0; do
  OUT=$R/gpurun_out/pmc_insts_$a; mkdir -p $OUT
  AM_SF_ABLATE=$a timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR --kernel-trace -d $OUT -o p -- python $R/bench.py --hay-count 2048 --steps 2 --warmup 1 --no-cpu-baseline > $OUT.log 2>&1
  echo "== ablate $a"; python $R/tools/pmc_summary.py $OUT "k_sf<true, 1" | grep -E "INSTS|WAVE_CYC|WAIT_ANY|ACTIVE"
done
