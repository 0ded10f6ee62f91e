# round-5 final evidence: the GPU test suite, the default bench line (the driver's command), its kernel trace, the natural-text traffic of k_dfa, smoke()
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd $R
(timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log); tail -3 $OUT/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.out 2> $OUT/bench_default.err; echo "bench rc=$?"
bash tools/experiments/prof_trace.sh $1
bash $R/tools/pmc_traffic.sh $OUT/traffic natural_100k_10GiB
cd $R; python -c "import __graft_entry__ as g; g.smoke()"
