# natural-text throughput for a few settings of the walker queue (round 5, after the one-trip walk steps)
for v in "AM_SF_WQ_ITERS=1" "AM_SF_WQ_ITERS=2" "AM_SF_WQ_ITERS=3" "AM_SF_WQ_ITERS=4" "AM_SF_WQ=0"; do
  echo "== $v"
  env $v python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-parity --workload natural_100k_10GiB 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
