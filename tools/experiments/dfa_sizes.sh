# k_dfa against k_sf on natural text for batches of 4 MiB ... 2 GiB (where does the table walk start to win?)
for g in 0.00390625 0.015625 0.03125 0.0625 0.125 0.25 0.5 2; do
  echo "== $g GiB"
  timeout 300 python tools/experiments/dfa_probe.py natural_100k_10GiB $g 2>&1 | grep "sf count\|sf emit\|dfa count\|dfa emit"
done
