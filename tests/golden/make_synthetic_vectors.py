#!/usr/bin/env python3
"""Writes tests/golden/synthetic_vectors.json: seeded fragment-pool cases (the generator idea of the
reference's tests/Data/Text/TestInstances.hs:46-93) with the fold sequences [(haystack, matchPos,
needle index)] computed by the pinned oracle (oracle/am_oracle.c) and, where no empty needle is
involved, cross-checked against the independent naive oracle at generation time.  The vectors let the
GPU parity tests run against committed data."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import naive, oracle  # noqa: E402
from tests.helpers import fragment_case, oracle_triples  # noqa: E402

rng = random.Random(20260926)
cases = []
for i in range(80):
    needles, hays = fragment_case(rng, n_hay_max=4, hay_frags=60)
    for case in (0, 1):
        ns = [oracle.lower_utf8(n).decode("utf-8") for n in needles] if (case and i % 5) else needles
        m = oracle.Machine(ns)
        triples = oracle_triples(m, case, hays)
        if "" not in ns:
            ref = []
            for h_i, h in enumerate(hays):
                ref += [(h_i, e, idx) for e, idx in naive.all_matches(ns, h, bool(case))]
            assert ref == triples, (ns, hays, case)
        cases.append({"case": case, "needles": ns, "haystacks": hays, "triples": triples})
with open(os.path.join(ROOT, "tests", "golden", "synthetic_vectors.json"), "w") as f:
    json.dump({"generator": "tests/golden/make_synthetic_vectors.py (seed 20260926)", "cases": cases}, f, ensure_ascii=True, separators=(",", ":"))
print(len(cases), "cases,", sum(len(c["triples"]) for c in cases), "matches")
