# the kernel trace of the driver's bench command (rocprofv3 --kernel-trace --stats), summarised per kernel and per cluster of launch durations
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/kt.log 2>&1; echo "kernel trace rc=$?"
python $R/tools/rocprof_summary.py $OUT/kt > $OUT/kt_summary.md 2>&1
grep "^{" $OUT/kt.log | tail -1 > $OUT/kt_bench_line.json
rm -rf $OUT/kt
