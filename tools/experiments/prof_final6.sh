# round-6 final evidence on the GPU box (via gpurun): the GPU test suite, the driver's bench command, its kernel trace, the HBM traffic of every workload's dominant kernel,
# k_dfa's counters, the robustness sweep, the index assertions, the soaks, smoke().  usage: prof_final6.sh <tag>      -> gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd $R
(timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log); tail -3 $OUT/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.log 2> $OUT/bench_default.err; echo "bench rc=$?"
bash tools/experiments/prof_trace.sh $1
bash $R/tools/pmc_traffic.sh $OUT/traffic
bash tools/experiments/dfa_pmc6.sh $1/dfa_pmc 0 > $OUT/dfa_pmc_summary.txt 2>&1; rm -rf $OUT/dfa_pmc/*/; tail -2 $OUT/dfa_pmc_summary.txt
cd $R
(timeout 1200 python tests/measure/robustness_sweep.py 2 > $OUT/robustness.md 2> $OUT/robustness.err; echo "robustness rc=$?")
(timeout 1200 bash tools/bounds_check.sh > $OUT/bounds.log 2>&1; echo "bounds rc=$?"; tail -3 $OUT/bounds.log)
(timeout 900 python tests/measure/soak_dfa.py 20000 > $OUT/soak_dfa.log 2>&1; tail -1 $OUT/soak_dfa.log)
(timeout 400 python tests/measure/soak.py 120 > $OUT/soak.log 2>&1; tail -1 $OUT/soak.log)
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
