#!/usr/bin/env python3
"""GPU-box measurements quoted in DESIGN.md that are NOT the headline metric:
  (1) PCIe-inclusive rate of the one-shot host entry point am_run (staging + H2D + scan + D2H of records),
  (2) BASELINE config 5: Replacer.run over a batch (50k pairs; GPU scan per pass + host splice), reduced size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import alfred_margaret_amd as am
from alfred_margaret_amd import synth
from oracle import oracle

# (1) host-slice path, cfg2 automaton, 512 x 1 MiB
needles = synth.needles_for("cfg2_runText_10k_1GiB")
a = am.Automaton(needles)
n_hay, cells = 512, 1024
text = synth.haystacks_host(needles, False, 0, n_hay * cells)
hays = [text[i * cells * 1024:(i + 1) * cells * 1024] for i in range(n_hay)]
a.run_records(0, hays)      # warm-up: allocates the pinned staging buffer
t0 = time.perf_counter(); recs = a.run_records(0, hays); dt = time.perf_counter() - t0
print("am_run on host slices (PCIe inclusive, pinned staging): %d MiB, %d records, %.3f s -> %.2f GiB/s" % (text.size >> 20, len(recs), dt, text.size / dt / 2**30))
t0 = time.perf_counter(); c = a.count_matches(0, hays); dt = time.perf_counter() - t0
print("am_count on host slices: %.3f s -> %.2f GiB/s (total %d)" % (dt, text.size / dt / 2**30, int(c.sum())))

# (2) Replacer, cfg5 shape at reduced size: 50k pairs, 2048 x 64 KiB = 128 MiB
rng = np.random.default_rng(5)
pair_needles = synth.make_needles(50_000, False, seed=synth.NEEDLE_SEED + 5)
repls = ["".join(chr(ord("A") + int(x)) for x in rng.integers(0, 26, size=int(rng.integers(0, 17)))) for _ in pair_needles]
pairs = list(zip(pair_needles, repls))
t0 = time.perf_counter(); r = am.Replacer(0, pairs); build_s = time.perf_counter() - t0
n_hay, cells = 2048, 64
text = synth.haystacks_host(pair_needles, False, 0, n_hay * cells)
hays = [bytes(text[i * cells * 1024:(i + 1) * cells * 1024]) for i in range(n_hay)]
r.run_batch(hays[:8])
t0 = time.perf_counter(); out = r.run_batch(hays); dt = time.perf_counter() - t0
print("Replacer.run batch (50k pairs, %d MiB): build %.2f s, run %.3f s -> %.3f GiB/s input" % (text.size >> 20, build_s, dt, text.size / dt / 2**30))
# parity on a sample against the oracle's Replacer
t0 = time.perf_counter(); orc = oracle.Replacer(0, pairs); k = 4
exp = [orc.run(h) for h in hays[:k]]; dt_o = time.perf_counter() - t0
assert out[:k] == exp, "Replacer parity failure"
print("  parity vs oracle on %d haystacks OK; oracle %.3f s for %d KiB -> %.5f GiB/s" % (k, dt_o, k * cells, k * cells * 1024 / dt_o / 2**30))
