#!/bin/bash
# Runs on the GPU box (via gpurun): HBM traffic of the dominant kernel (k_sf; natural text: k_dfa) per workload -- FETCH_SIZE and WRITE_SIZE, each in its own rocprofv3 pass (only
# --kernel-trace next to --pmc), on a ~2-GiB launch of every workload.  Usage: tools/pmc_traffic.sh <outdir>   -> <outdir>/<workload>.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1
ONLY=${2:-}            # optional: only this workload
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for spec in cfg3_runLower_100k_10GiB:2048 cfg2_runText_10k_1GiB:32768 cfg4_100k_1M_haystacks:20480 natural_100k_10GiB:2048; do
  W=${spec%%:*}; N=${spec##*:}
  [ -n "$ONLY" ] && [ "$ONLY" != "$W" ] && continue
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace -d "$OUT/$W/$C" -o p -- python "$R/bench.py" --workload $W --hay-count $N --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-h2d > "$OUT/$W.$C.log" 2>&1
    echo "$W $C rc=$?"
  done
  K=k_sf; [ "$W" = natural_100k_10GiB ] && K=k_dfa        # the dictionary takes the table-walk route (csrc/am_dfa.hip): k_dfa<...> and k_dfa_place both match
  python "$R/tools/pmc_summary.py" "$OUT/$W" "$K" > "$OUT/$W.txt" 2>&1
  rm -rf "$OUT/$W"
done
# the Replacer's one-kernel loop (k_rp_lds: a haystack's lists in LDS) on config 5: 4000 haystacks = 250 MiB of input -- nearly every wavefront slot busy once, and
# below the 4096 from which results that go to the host are cut into groups (every k_rp_lds launch of the run then covers the same 4000 haystacks)
W=cfg5_replacer_50k_1GiB
[ -n "$ONLY" ] && [ "$ONLY" != "$W" ] && exit 0
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d "$OUT/$W/$C" -o p -- python "$R/bench.py" --workload $W --hay-count 4000 --steps 2 --warmup 1 --no-cpu-baseline --no-parity > "$OUT/$W.$C.log" 2>&1
  echo "$W $C rc=$?"
done
python "$R/tools/pmc_summary.py" "$OUT/$W" "k_rp_lds" > "$OUT/$W.txt" 2>&1
rm -rf "$OUT/$W"
