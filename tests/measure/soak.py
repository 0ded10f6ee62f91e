#!/usr/bin/env python3
"""Randomised differential soak on the MI355X: many seeds of the fragment-pool generator (tests/helpers.py) through every
entry point and both kernels against the oracle, plus random Replacer / containsAll / Splitter cases.  Not collected by
pytest (takes minutes): python tests/measure/soak.py [seconds] [first_seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import alfred_margaret_amd as am
am.api.load_check()          # k_ac (set_kernel(1)) is test infrastructure: libam_check.so
from oracle import oracle
from tests.helpers import expand_records, fragment_case, oracle_triples

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
t_end = time.time() + budget
n_cases = n_rep = 0
while time.time() < t_end:
    rng = random.Random(seed)
    needles, hays = fragment_case(rng, n_hay_max=8, hay_frags=rng.choice((10, 60, 400)))
    for case in (0, 1):
        ns = [oracle.lower_utf8(n).decode() for n in needles] if (case and rng.random() < 0.8) else needles
        o = oracle.Machine(ns)
        a = am.Automaton(ns)
        exp = oracle_triples(o, case, hays)
        for k in (0, 1):
            a.set_kernel(k)
            recs = a.run_records(case, hays)
            got = expand_records(o.values_off(), o.values(), recs["haystack"], recs["state"], recs["end_pos"])
            assert got == exp, ("records", seed, case, k, ns, hays)
            assert [int(c) for c in a.count_matches(case, hays)] == [o.count_matches(case, h) for h in hays], ("count", seed, case, k)
        s = am.Searcher(case, ns)
        assert [bool(x) for x in s.contains_any_batch(hays)] == [o.contains_any(case, h) for h in hays], ("any", seed, case)
        assert [bool(x) for x in s.contains_all_batch(hays)] == [o.contains_all(case, h) for h in hays], ("all", seed, case)
        n_cases += 1
        if "" not in ns:
            pairs = [(n, "".join(rng.choice("xyzİ" + n[:2]) for _ in range(rng.randint(0, 4)))) for n in needles[:rng.randint(1, 8)]]
            oo = oracle.Replacer(case, pairs)
            r = am.Replacer(case, pairs)
            lim = rng.choice((-1, -1, 30, 200))
            expr = [oo.run(h, lim) for h in hays]
            for forced in (None, "1"):
                for pieces in (None, "1"):                  # small batches take the splicing loop by default: also the piece-table loop
                    if forced: os.environ["AM_RP_PARALLEL_FOLD"] = forced
                    if pieces: os.environ["AM_RP_PIECES"] = pieces
                    got = r.run_batch(hays, lim)
                    os.environ.pop("AM_RP_PARALLEL_FOLD", None); os.environ.pop("AM_RP_PIECES", None)
                    assert got == expr, ("replacer", seed, case, forced, pieces, pairs, hays, lim)
            n_rep += 1
    seed += 1
print("soak ok: %d automaton cases, %d replacer cases, seeds up to %d" % (n_cases, n_rep, seed))
