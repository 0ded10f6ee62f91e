#!/usr/bin/env python3
"""Turns one gpurun_out/<tag>/ directory made by tools/profile_round.sh (bench logs, rocprofv3 kernel-trace summary, PMC summary)
into profiles/<round>_kernel_trace_<tag>.md, profiles/<round>_pmc_<tag>.md and profiles/<round>_replacer_trace_<tag>.md; the bench logs go to
profiles/history/ (profiles/pmc_traffic.json: tools/pmc_traffic.py).
Usage: python tools/write_profiles.py r02 r02b "514 (r01 v15), 478 (v16e)" """
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND, V = sys.argv[1], sys.argv[2]
hist = sys.argv[3] if len(sys.argv) > 3 else ""
D = os.path.join(ROOT, "gpurun_out", V)
P = os.path.join(ROOT, "profiles")
line = lambda f: json.loads([l for l in open(os.path.join(D, f)) if l.startswith("{")][-1])
kt = open(os.path.join(D, "kt_summary.md")).read().split("| kernel | grid")[0].rstrip()
grid = [l for l in open(os.path.join(D, "kt_summary.md")).read().split("| kernel | grid")[1].split("\n") if "k_sf<true, 1" in l and "1024" in l.split("|")[3]][0].split("|")
d, plain = line("kt.log"), line("bench_default.log")
ksf = [l for l in kt.split("\n") if "k_sf<true, 1" in l][0].split("|")
os.makedirs(os.path.join(P, "history"), exist_ok=True)
for f in os.listdir(D):
    if f.startswith("bench_") and f.endswith(".log") and os.path.getsize(os.path.join(D, f)):
        shutil.copy(os.path.join(D, f), os.path.join(P, "history", "%s_%s_%s" % (RND, V, f)))
cb = plain["cpu_baseline"]
open(os.path.join(P, "%s_kernel_trace_%s.md" % (RND, V)), "w").write('''# %s -- k_sf (%s), rocprofv3 --kernel-trace --stats

Command on the MI355X box (tools/profile_round.sh): `cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d gpurun_out/%s/kt -o kt -- python bench.py --no-cpu-baseline`
(default workload: cfg3, 100k-needle IgnoreCase automaton, 10240 x 1 MiB = 10 GiB per step, 1 warm-up + 5 timed steps + 1 count-only call, then the
parity gate: both kernels over every haystack + k_fold_hash -- the two k_ac launches below belong to the gate, not to the timed steps)

%s

`k_sf<IC=true, MODE=1 (emit), ILP=2, LW=15, SHORT=false, DBG=false, NT=1024>`: grid %s threads = 256 workgroups x 1024 threads, %s B of LDS (2 KiB mask table + 128 KiB
filter + staged chunks + queues), %s VGPRs, no scratch.  rocprofv3 average %.3f ms per 10 GiB launch; bench.py's HIP events on the launch stream in the same
run: %.4f ms.

bench.py JSON line of this run (under the profiler):

```
%s
```

Same build without the profiler (profiles/history/%s_%s_bench_default.log): %.1f GiB/s, k_sf %.3f ms/launch (roofline.frac %.4f), count-only %.0f GiB/s.
CPU oracle by the reference's protocol (pinned core, 5 repetitions): run %.3f +- %.3f s (min %.3f) for %s = %.4f GiB/s on 1 core;
build + run %.3f s; %.3f GiB/s with one task per usable host core (%d).
''' % (RND, V, V, kt, grid[2].strip(), grid[4].strip(), grid[5].strip(), float(ksf[4]) / 1e3, d["roofline"]["avg_launch_ms"], json.dumps(d), RND, V, plain["value"],
       plain["roofline"]["avg_launch_ms"], plain["roofline"]["frac"], plain["count_only_gibps"], cb["run_s"]["mean"], cb["run_s"]["stdev"], cb["run_s"]["min"], cb["sample"].split(" of ")[0],
       cb["value"], cb["build_plus_run_s"]["mean"], cb["all_cores"]["value"], cb["all_cores"]["cores"]))

pm = open(os.path.join(D, "pmc_summary.txt")).read()
emit = pm[pm.index("### void am::dev::k_sf<true, 1"):].split("\n###")[0]
vals = {}
for l in emit.split("\n"):
    p = l.split()
    if len(p) >= 3 and p[2].startswith("avg="):
        vals[p[0]] = float(p[2][4:])
chunks, scanned = 2097152, 2147483648
stream_req = scanned / 128.0
other = vals["TCC_EA0_RDREQ_sum"] - stream_req
upper = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
split = scanned + other * 64 + vals["WRITE_SIZE"] * 1024
open(os.path.join(P, "%s_pmc_%s.md" % (RND, V)), "w").write('''# %s -- k_sf (%s) PMC counters (rocprofv3 --pmc, one pass per counter group, 2 GiB per launch)

Command: `tools/pmc_profile.sh gpurun_out/%s/pmc --no-parity` = bench.py --hay-count 2048 --steps 2 (cfg3 automaton), six rocprofv3 --pmc passes with --kernel-trace only.
Averages per k_sf launch (emit mode, 2 GiB of haystack = 2,097,152 1-KiB chunks, 4096 wavefronts):

```
%s
```

HBM traffic per launch (2.147 GB scanned; algorithmic 2.147 GB + 0.10 GB of records):
  as MI355X_MICROARCH.md prescribes (FETCH_SIZE tallies the 128-B requests of a wide coalesced stream at 64 B: double it):
    2 x FETCH_SIZE + WRITE_SIZE = (2 x %.4e + %.4e) KiB = %.3f GB = %.2fx the scanned bytes.  This doubles EVERY request, also the
    random 64-B lines of phase 2, so it is an upper bound; it is the figure bench.py scales into roofline.traffic.
  split by request count: TCC_EA0_RDREQ = %.3e requests, of which 2 GiB / 128 B = 1.678e7 are the haystack stream (2.147 GB) and the other
    %.3e are single lines of the probe and resolve tables (hot / cold cuckoo buckets, trie nodes, haystack index): x 64 B = %.2f GB.
    Stream + lines + writes = %.2f GB = %.2fx.  The stream itself is read exactly once (no re-reads: the probe takes its bytes from the
    chunk staged in LDS); what exceeds 1.05x are ~%.0f table lines per 1-KiB chunk that miss the 4-MiB L2 (the tables are 10 MB, the
    256-MiB Infinity Cache holds them: these requests are counted although they do not reach HBM).
  L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS) = %.0f %%.
Per 1-KiB chunk per wavefront: **%.0f VALU** (before: %s), %.0f SALU, %.1f LDS, %.1f VMEM-read instructions;
SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.0f %%; VALU-active share = SQ_ACTIVE_INST_VALU x 4 / SQ_WAVE_CYCLES... see DESIGN.md "where the cycles go";
LDS bank-conflict cycles / LDS active cycles = %.0f %%.
''' % (RND, V, V, emit.strip(), vals["FETCH_SIZE"], vals["WRITE_SIZE"], upper / 1e9, upper / scanned, vals["TCC_EA0_RDREQ_sum"], other, other * 64 / 1e9,
       split / 1e9, split / scanned, other / chunks, 100 * vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), vals["SQ_INSTS_VALU"] / chunks, hist,
       vals["SQ_INSTS_SALU"] / chunks, vals["SQ_INSTS_LDS"] / chunks, vals["SQ_INSTS_VMEM_RD"] / chunks, 100 * vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"],
       100 * vals["SQ_LDS_BANK_CONFLICT"] / vals["SQ_LDS_IDX_ACTIVE"]))
# (profiles/pmc_traffic.json -- the figure bench.py scales into roofline.traffic, per workload -- is written by tools/pmc_traffic.py)

# Replacer (config 5)
if os.path.exists(os.path.join(D, "bench_cfg5_replacer_50k_1GiB.log")) and os.path.getsize(os.path.join(D, "bench_cfg5_replacer_50k_1GiB.log")):
    k5 = open(os.path.join(D, "kt5_summary.md")).read().split("| kernel | grid")[0].rstrip()
    r5 = line("bench_cfg5_replacer_50k_1GiB.log")
    open(os.path.join(P, "%s_replacer_trace_%s.md" % (RND, V)), "w").write('''# %s -- Replacer.run on config 5 (%s), rocprofv3 --kernel-trace --stats

    Command: `rocprofv3 --kernel-trace --stats -- python bench.py --workload cfg5_replacer_50k_1GiB --steps 3 --warmup 1 --no-cpu-baseline`
    (16384 x 64 KiB, 50 000 pairs; every step is one full scan, ONE k_rp_loop launch that runs all ~160 passes of every haystack, and one materialise launch; the trace covers the device-resident steps, the
    host-result steps (am_replacer_run_batch: the `__amd_rocclr_copyBuffer` rows are their device-to-host copies) and one profiled step)

    %s

    bench.py line without the profiler (profiles/history/%s_%s_bench_cfg5_replacer_50k_1GiB.log):

    ```
    %s
    ```
    ''' % (RND, V, k5, RND, V, json.dumps(r5)))

print("VALU/chunk %.0f SALU %.0f LDS %.1f traffic upper %.2fx split %.2fx" % (vals["SQ_INSTS_VALU"] / chunks, vals["SQ_INSTS_SALU"] / chunks, vals["SQ_INSTS_LDS"] / chunks, upper / scanned, split / scanned))
