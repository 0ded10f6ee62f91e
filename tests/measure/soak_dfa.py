#!/usr/bin/env python3
"""GPU-box soak of the table-walk route (k_dfa, csrc/am_dfa.hip): random fragment automata (Unicode case variants, duplicates, shared prefixes and suffixes), every one
forced to carry a DFA section (AM_DFA=1) and forced onto the route (am_automaton_set_kernel(a, 3)), with random unit sizes, the walk's variants (AM_DFA_TUNE) and,
every other case, most bytes made rare; records, counts and flags against the oracle.  Usage: python tests/measure/soak_dfa.py [cases, default 1500] [seed]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

import alfred_margaret_amd as am
from oracle import oracle
from tests.helpers import expand_records, fragment_case, oracle_triples

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20260928)
am.debug_set("AM_DFA", 1)
t0, done, records = time.time(), 0, 0
for it in range(N):
    needles, hays = fragment_case(rng, n_hay_max=rng.choice((1, 4, 12)), hay_frags=rng.choice((20, 120, 400)))
    if "" in needles or not any(needles):
        continue
    case = rng.randrange(2)
    ns = [oracle.lower_utf8(n).decode() for n in needles] if (case and rng.random() < 0.8) else needles
    am.debug_set("AM_DFA_CHUNK", rng.choice((-1, 64, 80, 256, 4096, 131072)))
    am.debug_set("AM_DFA_RARE_PERMILLE", rng.choice((-1, 400)))
    am.debug_set("AM_DFA_NO_CHAINS", rng.choice((-1, -1, 1)))
    am.debug_set("AM_DFA_TUNE", rng.choice((-1, -1, 3, 1, 0x1000000)))      # the default walk; lanes out of step; 16 bytes of text per request; no records in LDS
    hays = hays + rng.choice(([], [""], ["", hays[0][:5] if hays else ""]))
    o = oracle.Machine(ns)
    a = am.Automaton(ns)
    a.set_kernel(3)
    exp = oracle_triples(o, case, hays)
    recs = a.run_records(case, hays)
    assert expand_records(o.values_off(), o.values(), recs["haystack"], recs["state"], recs["end_pos"]) == exp, (it, case, ns, hays)
    assert [int(c) for c in a.count_matches(case, hays)] == [o.count_matches(case, h) for h in hays], (it, case, ns, hays)
    s = am.api._Slices(hays)
    flags = np.zeros(max(s.n, 1), np.uint8)
    am.api.check(am.api.libam().am_contains_any(a.device, case, s.arr, s.n, flags.ctypes.data))
    assert [bool(x) for x in flags[:s.n]] == [o.contains_any(case, h) for h in hays], (it, case, ns, hays)
    done += 1
    records += len(recs)
print("soak_dfa: %d automata x (records, counts, flags) == oracle; %d records; %.1f s" % (done, records, time.time() - t0))
