#!/usr/bin/env python3
"""Turns one gpurun_out/<tag>/ directory (bench logs, rocprofv3 kernel-trace summary, PMC summary; see the commands in the
generated files) into profiles/r01_kernel_trace_<tag>.md, profiles/r01_pmc_<tag>.md and profiles/pmc_traffic.json.
Usage: python tools/write_profiles.py v15 "805, v12: 695, v13: 660, v14: 550" """
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V = sys.argv[1]
hist = sys.argv[2] if len(sys.argv) > 2 else ""
D = os.path.join(ROOT, "gpurun_out", V)
kt = open(os.path.join(D, "kt_summary.md")).read().split("| kernel | grid")[0].rstrip()
bench = [l for l in open(os.path.join(D, "kt.log")) if l.startswith("{")][-1].strip()
d = json.loads(bench)
plain = json.loads([l for l in open(os.path.join(D, "bench_default.log")) if l.startswith("{")][-1])
ksf = [l for l in kt.split("\n") if "k_sf<true, 1" in l][0].split("|")
open(os.path.join(ROOT, "profiles", "r01_kernel_trace_%s.md" % V), "w").write('''# Round 1 — k_sf %s, rocprofv3 --kernel-trace --stats

Command on the MI355X box: `cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d gpurun_out/%s/kt -o kt -- python bench.py --no-cpu-baseline`
(default workload: cfg3, 100k-needle IgnoreCase automaton, 10240 x 1 MiB = 10 GiB per step, 1 warm-up + 5 timed steps + 1 count-only call)

%s

`k_sf<IC=true, MODE=1 (emit), ILP=2, LW=15, SHORT=false, DBG=false>`: grid 256 workgroups x 1024 threads, 156.5 KiB LDS (128 KiB filter + staged
chunks + queues), 90 VGPRs, no scratch.  rocprofv3 average %.3f ms per 10 GiB launch; bench.py's HIP events on the launch stream in the same
run: %.4f ms.

bench.py JSON line of this run (under the profiler):

```
%s
```

Same build without the profiler (profiles/history/r01_bench_%s_default.log): %.1f GiB/s, k_sf %.2f ms/launch, count-only %.0f GiB/s,
CPU oracle %.4f GiB/s on 1 core, %.3f GiB/s with one task per usable host core (%d).
''' % (V, V, kt, float(ksf[4]) / 1e3, d["roofline"]["avg_launch_ms"], bench, V, plain["value"], plain["roofline"]["avg_launch_ms"], plain["count_only_gibps"],
       plain["cpu_baseline"]["value"], plain["cpu_baseline"]["all_cores"]["value"], plain["cpu_baseline"]["all_cores"]["cores"]))
pm = open(os.path.join(D, "pmc_summary.txt")).read()
emit = pm[pm.index("### void am::dev::k_sf<true, 1"):].split("\n###")[0]
vals = {}
for l in emit.split("\n"):
    p = l.split()
    if len(p) >= 3 and p[2].startswith("avg="):
        vals[p[0]] = float(p[2][4:])
chunks = 2097152
open(os.path.join(ROOT, "profiles", "r01_pmc_%s.md" % V), "w").write('''# Round 1 — k_sf %s PMC counters (rocprofv3 --pmc, one pass per counter group, 2 GiB per launch)

Command: `tools/pmc_profile.sh gpurun_out/%s/pmc` = bench.py --hay-count 2048 --steps 2 (cfg3 automaton), six rocprofv3 --pmc passes with --kernel-trace only.
Averages per k_sf launch (emit mode, 2 GiB of haystack = 2,097,152 1-KiB chunks, 4096 wavefronts):

```
%s
```

HBM traffic per launch, corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE counts the wide coalesced stream at half):
  2 x FETCH_SIZE + WRITE_SIZE = (2 x %.4e + %.4e) KiB = %.3f GB (upper bound) per 2.147 GB scanned; algorithmic 2.147 GB + 0.10 GB of records.
  By TCC_EA0_RDREQ (%.3e requests): 1.68e7 x 128 B = 2.15 GB haystack stream + the rest x 64 B of random lines (phase 2: cold buckets, trie nodes, label compares).
  L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS) = %.0f %%.
Per 1-KiB chunk per wavefront: **%.0f VALU** (v11: %s), %.0f SALU, %.1f LDS, %.1f VMEM-read instructions;
SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.0f %%; LDS bank-conflict cycles / LDS active cycles = %.0f %%.
VALU issue model (tools/microbench/valu_rates*.hip, 4 waves per SIMD): plain VOP2 with VGPR/immediate operands (add, and, or, xor, shift by constant,
mov) issues every ~2.8 cycles per wave64 instruction, everything else (VOP3 encodings, SGPR operands, mul, bfe, alignbyte, cndmask, compares) every ~4.7.
''' % (V, V, emit.strip(), vals["FETCH_SIZE"], vals["WRITE_SIZE"], (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024 / 1e9, vals["TCC_EA0_RDREQ_sum"],
       100 * vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), vals["SQ_INSTS_VALU"] / chunks, hist, vals["SQ_INSTS_SALU"] / chunks,
       vals["SQ_INSTS_LDS"] / chunks, vals["SQ_INSTS_VMEM_RD"] / chunks, 100 * vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"],
       100 * vals["SQ_LDS_BANK_CONFLICT"] / vals["SQ_LDS_IDX_ACTIVE"]))
tr = {"source": "profiles/r01_pmc_%s.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, cfg3 automaton, 2 GiB launch)" % V, "kernel": "k_sf", "workload": "cfg3_runLower_100k_10GiB",
      "fetch_size_kib": vals["FETCH_SIZE"], "write_size_kib": vals["WRITE_SIZE"], "scanned_bytes": 2147483648,
      "hbm_bytes_per_scanned_byte": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024 / 2147483648,
      "correction": "2 x FETCH_SIZE (gfx950 counts 128-B streaming requests at 64 B) + WRITE_SIZE, per MI355X_MICROARCH.md"}
json.dump(tr, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print("VALU/chunk %.0f SALU %.0f traffic/byte %.2f" % (vals["SQ_INSTS_VALU"] / chunks, vals["SQ_INSTS_SALU"] / chunks, tr["hbm_bytes_per_scanned_byte"]))
