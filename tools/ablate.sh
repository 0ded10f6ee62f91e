#!/bin/bash
# timing experiment: AM_SF_ABLATE 0 = full, 1 = filter only (no candidates), 5 = filter + compaction, 2 = + probe without its loads,
# 3 = + bucket loads (nothing deferred), 4 = full probe but no resolve
R=${GRAFT_REPO_ROOT:-/root/repo}
for a in $1; do
  AM_SF_ABLATE=$a timeout 300 python $R/bench.py --hay-count 4096 --steps 3 --no-cpu-baseline --no-parity 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('ablate $a: count-only %.1f GiB/s  k_sf %.3f ms/launch' % (d['count_only_gibps'], r['avg_launch_ms']))
"
done
