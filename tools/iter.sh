#!/bin/bash
# One k_sf optimisation iteration on the GPU box (via gpurun): parity subset, bench line, instruction-count PMC pass.
# Usage: tools/iter.sh <tag> [quick]     -> gpurun_out/<tag>/{tests.log,bench.log,insts.txt}
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; MODE=${2:-full}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
if [ "$MODE" = "full" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
else
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "golden or edge or fragment or synthetic or full_size or soak or cfg4 or fold_hash or pool" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
fi
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench.log 2>&1; echo "bench rc=$?"
python - <<PY
import json
for l in open("$OUT/bench.log"):
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print("BENCH value %.1f GiB/s ms/step %.3f k_sf %.3f ms frac %.4f count_only %.1f parity %s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], d["count_only_gibps"], d.get("parity",{}).get("kernels_agree")))
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR --kernel-trace -d $OUT/pmc -o p -- python $R/bench.py --hay-count 2048 --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $OUT/pmc.log 2>&1
python $R/tools/pmc_summary.py $OUT/pmc "k_sf<true, 1" > $OUT/insts.txt 2>&1
python - <<PY
vals={}
for l in open("$OUT/insts.txt"):
    p=l.split()
    if len(p)>=3 and p[2].startswith("avg="): vals[p[0]]=float(p[2][4:])
ch=2097152.0
if vals: print("PER CHUNK: VALU %.0f SALU %.0f LDS %.1f VMEM_RD %.1f | wave cycles/chunk %.0f (x4 quad) wait_any %.0f%%" % (vals["SQ_INSTS_VALU"]/ch, vals["SQ_INSTS_SALU"]/ch, vals["SQ_INSTS_LDS"]/ch, vals["SQ_INSTS_VMEM_RD"]/ch, vals["SQ_WAVE_CYCLES"]/ch, 100*vals["SQ_WAIT_ANY"]/vals["SQ_WAVE_CYCLES"]))
PY
rm -rf $OUT/pmc
