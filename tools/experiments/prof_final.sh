# round-5 final evidence: the default bench line (the driver's command), its kernel trace, the natural-text traffic of the final k_dfa
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.out 2> $OUT/bench_default.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/kt.log 2>&1; echo "kernel trace rc=$?"
python $R/tools/rocprof_summary.py $OUT/kt > $OUT/kt_summary.md 2>&1
rm -rf $OUT/kt
bash $R/tools/pmc_traffic.sh $OUT/traffic natural_100k_10GiB
cd $R; python -c "import __graft_entry__ as g; g.smoke()"
