#!/usr/bin/env python3
"""CPU replay of k_sf's filter -> probe -> resolve decisions on a BASELINE workload (no GPU needed): how many positions per
KiB pass the Bloom filter (real 4-byte-suffix hits vs false positives), how many the probe defers to phase 2 and by which
kind of hot slot, and how many of those are real matches.  Usage: python tools/replay_stats.py [workload] [KiB of text]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import alfred_margaret_amd as am
from alfred_margaret_amd import synth

workload = sys.argv[1] if len(sys.argv) > 1 else "cfg3_runLower_100k_10GiB"
kib = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
w = synth.WORKLOADS[workload]
needles = synth.needles_for(workload)
a = am.Automaton(needles)
with tempfile.TemporaryDirectory() as d:
    a.transitions().tofile(os.path.join(d, "tr.bin")); a.offsets().tofile(os.path.join(d, "of.bin")); a.root_ascii().tofile(os.path.join(d, "ra.bin"))
    np.diff(a.values_off()).astype(np.uint32).tofile(os.path.join(d, "vl.bin"))
    synth.haystacks_host(needles, w["mixed"], 0, kib, natural=bool(w.get("natural"))).tofile(os.path.join(d, "text.bin"))
    exe = os.path.join(d, "replay")
    csrc = os.path.join(ROOT, "alfred-margaret_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-w", "-DAM_REPLAY_CASE=%d" % w["case"], "-I", csrc, os.path.join(ROOT, "tools", "replay_stats.cpp"),
                           os.path.join(csrc, "am_flatten.cpp"), "-o", exe])
    subprocess.check_call([exe], cwd=d)
