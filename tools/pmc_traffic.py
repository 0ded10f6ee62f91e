#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the per-workload summaries tools/pmc_traffic.sh leaves (gpurun_out/<tag>/traffic/<workload>.txt): HBM bytes per scanned
byte of k_sf's match-emitting launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB / bytes of the launch (gfx950 counts a 128-byte streaming request at 64 bytes:
MI355X_MICROARCH.md), keyed by workload; bench.py scales the entry of the workload it runs to its launch and refuses entries of another image version.
Usage: python tools/pmc_traffic.py <gpurun_out/tag/traffic> <round tag>"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
D, TAG = sys.argv[1], sys.argv[2]
SPEC = {"cfg5_replacer_50k_1GiB": (4000, 64 << 10), "cfg3_runLower_100k_10GiB": (2048, 1 << 20), "cfg2_runText_10k_1GiB": (32768, 64 << 10), "cfg4_100k_1M_haystacks": (20480, 100 << 10), "natural_100k_10GiB": (2048, 1 << 20)}
version = int(re.search(r"kImageVersion\s*=\s*(\d+)", open(os.path.join(ROOT, "alfred-margaret_amd", "csrc", "am_image.h")).read()).group(1))
out = {"correction": "2 x FETCH_SIZE (gfx950 counts 128-B streaming requests at 64 B) + WRITE_SIZE, per MI355X_MICROARCH.md; one rocprofv3 --pmc pass per counter",
       "image_version": version, "source": "tools/pmc_traffic.sh on the MI355X box (%s)" % TAG, "workloads": {}}
for w, (n_hay, hb) in SPEC.items():
    f = os.path.join(D, w + ".txt")
    if not os.path.exists(f):
        continue
    blocks = open(f).read().split("### ")
    kernel = "k_rp_lds" if w.startswith("cfg5") else "k_dfa + k_dfa_place" if w.startswith("natural") else "k_sf"
    # k_dfa: MODE = 16, records in one walk (tokens) -- and k_dfa_place, which turns the tokens into the records: the step's bytes are the pair's; k_sf: MODE = 1, the match-emitting instantiation
    patterns = {"k_rp_lds": [r"void am::dev::k_rp_lds<false, false"], "k_dfa + k_dfa_place": [r"void am::dev::k_dfa<16,", r"am::dev::k_dfa_place"],
                "k_sf": [r"void am::dev::k_sf<(true|false), 1,"]}[kernel]
    vals, per_kernel = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}, {}
    for pattern in patterns:
        emit = [b for b in blocks if re.match(pattern, b)]
        if not emit:
            vals = None
            break
        one = {}
        for l in emit[0].split("\n"):
            p = l.split()
            if len(p) >= 3 and p[2].startswith("avg="):
                one[p[0]] = float(p[2][4:])
        if "FETCH_SIZE" not in one or "WRITE_SIZE" not in one:
            vals = None
            break
        per_kernel[emit[0].split("(")[0].split("::")[-1].split("<")[0]] = {"fetch_size_kib": one["FETCH_SIZE"], "write_size_kib": one["WRITE_SIZE"]}
        vals["FETCH_SIZE"] += one["FETCH_SIZE"]; vals["WRITE_SIZE"] += one["WRITE_SIZE"]
    if not vals:
        continue
    scanned = n_hay * hb
    out["workloads"][w] = {"kernel": kernel, "launch_bytes": scanned, "fetch_size_kib": vals["FETCH_SIZE"], "write_size_kib": vals["WRITE_SIZE"],
                           "hbm_bytes_per_scanned_byte": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024 / scanned,
                           "fetch_size_bytes_per_scanned_byte_as_counted": vals["FETCH_SIZE"] * 1024 / scanned,
                           **({"per_kernel": per_kernel} if len(per_kernel) > 1 else {}), "profile": "profiles/%s_pmc_traffic.md" % TAG}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
lines = ["# %s -- HBM traffic of the dominant kernel per workload (k_sf; natural text: k_dfa + k_dfa_place, the pair that makes the records; config 5: k_rp_lds, per byte of INPUT text) (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each, ~2-GiB launches; tools/pmc_traffic.sh)" % TAG, "",
         "| workload | launch | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes per scanned byte (2 x FETCH + WRITE) |", "|---|---|---|---|---|"]
for w, e in out["workloads"].items():
    lines.append("| %s | %.2f GiB | %.0f | %.0f | %.3f |" % (w, e["launch_bytes"] / 2**30, e["fetch_size_kib"], e["write_size_kib"], e["hbm_bytes_per_scanned_byte"]))
lines += ["", "Raw per-kernel averages: profiles/history/%s_pmc_traffic_<workload>.txt.  The correction (a 128-byte streaming request is counted as 64 bytes by FETCH_SIZE on gfx950)" % TAG,
          "is the one /opt/skills/guides/MI355X_MICROARCH.md prescribes; `bench.py` puts `hbm_bytes_per_scanned_byte` x the bytes of its launch into `roofline.traffic`.", ""]
open(os.path.join(ROOT, "profiles", "%s_pmc_traffic.md" % TAG), "w").write("\n".join(lines))
print(json.dumps(out["workloads"], indent=1))
