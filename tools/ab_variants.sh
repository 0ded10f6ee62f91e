#!/bin/bash
# A/B of k_sf tuning variants on the GPU box: AM_SF_VARIANT = ilp*10 + nt.  Usage: tools/ab_variants.sh "10 11 20 21 40 41" [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
VARS=$1; shift
for v in $VARS; do
  AM_SF_VARIANT=$v timeout 300 python $R/bench.py --hay-count 4096 --steps 3 --no-cpu-baseline "$@" 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('variant $v: emit %.1f GiB/s  count-only %.1f GiB/s  k_sf %.3f ms/launch (%.0f GB/s, frac %.4f)' % (d['value'], d['count_only_gibps'], r['avg_launch_ms'], r['achieved'], r['frac']))
"
done
