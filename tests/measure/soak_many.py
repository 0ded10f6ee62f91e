#!/usr/bin/env python3
"""GPU-box soak beyond the four seeds of tests/test_gpu_parity.py::test_soak_random_automata: random automata and haystacks, both case modes, batch calls and
one-document calls (the one-synchronisation am_run path) against the oracle.  Usage: python tests/measure/soak_many.py <first seed> <last seed + 1>
(round 3, final tree: seeds 0..149, no mismatch; ~2 s per seed, most of it the oracle)."""
import os, sys, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import alfred_margaret_amd as am
from oracle import oracle
from tests.helpers import expand_records, oracle_triples
from tests.test_gpu_parity import _soak_case
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = random.Random(20000 + seed)
    needles, hays = _soak_case(rng)
    # also single-document calls (the one-synchronisation am_run path) on each haystack
    for case in (0, 1):
        ns = [oracle.lower_utf8(n).decode() for n in needles] if case else needles
        o = oracle.Machine(ns); a = am.Automaton(ns)
        exp = oracle_triples(o, case, hays)
        recs = a.run_records(case, hays)
        ok = expand_records(o.values_off(), o.values(), recs["haystack"], recs["state"], recs["end_pos"]) == exp
        ok = ok and [int(c) for c in a.count_matches(case, hays)] == [o.count_matches(case, h) for h in hays]
        for h in hays[:6]:
            r1 = a.run_records(case, [h])
            ok = ok and expand_records(o.values_off(), o.values(), r1["haystack"], r1["state"], r1["end_pos"]) == oracle_triples(o, case, [h])
        if not ok:
            bad += 1; print("MISMATCH seed", seed, "case", case, flush=True)
print("soak seeds", sys.argv[1], sys.argv[2], "bad", bad)
