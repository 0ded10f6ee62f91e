#!/bin/bash
# Runs on the GPU box (via gpurun): everything the profiles/ summaries of a round are made from.
# Usage: tools/profile_round.sh <tag> [core]   (core: default bench, kernel trace and PMC passes only)
# Usage: tools/profile_round.sh <tag>   -> gpurun_out/<tag>/{bench_default.log,kt.log,kt_summary.md,pmc_summary.txt,bench_<workload>.log}
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.log 2>$OUT/bench_default.err; echo "bench default rc=$?"
[ "${2:-all}" = "core" ] && WL="" || WL="cfg2_runText_10k_1GiB cfg2_single_1GiB cfg4_100k_1M_haystacks cfg5_replacer_50k_1GiB natural_100k_10GiB"
for w in $WL; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.log 2>$OUT/bench_$w.err; echo "bench $w rc=$?"
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-h2d > $OUT/kt.log 2>&1; echo "kernel trace rc=$?"
python $R/tools/rocprof_summary.py $OUT/kt > $OUT/kt_summary.md 2>&1
[ "${2:-all}" = "core" ] || timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt5 -o kt5 -- python $R/bench.py --workload cfg5_replacer_50k_1GiB --steps 3 --warmup 1 --no-cpu-baseline > $OUT/kt5.log 2>&1; echo "kernel trace cfg5 rc=$?"
python $R/tools/rocprof_summary.py $OUT/kt5 > $OUT/kt5_summary.md 2>&1
bash $R/tools/pmc_profile.sh $OUT/pmc --no-parity
python $R/tools/pmc_summary.py $OUT/pmc "k_sf" > $OUT/pmc_summary.txt 2>&1
[ "${2:-all}" = "core" ] || bash $R/tools/pmc_traffic.sh $OUT/traffic
rm -rf $OUT/kt $OUT/kt5 $OUT/pmc/*/
ls $OUT
