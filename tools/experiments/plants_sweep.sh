# robustness sweep: needles planted per 1-KiB cell of the cfg3 haystacks (BASELINE = 1)
for p in 0 1 8 64; do
  echo "== plants $p"
  python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --plants $p --parity-oracle-mib 64 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('count_only_gibps'), (d.get('parity') or {}).get('kernels_agree'))"
done
