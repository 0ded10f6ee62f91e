R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05y; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/kt.log 2>&1; echo "kernel trace rc=$?"
python $R/tools/rocprof_summary.py $OUT/kt > $OUT/kt_summary.md 2>&1
rm -rf $OUT/kt
bash $R/tools/pmc_traffic.sh $OUT/traffic
ls $OUT $OUT/traffic
