"""Soak with the mid-sized adversarial automata of tests/test_gpu_parity.py::_soak_case (1500 long needles sharing stems, every UTF-8 length, near misses): records and Replacer output against the oracle for many seeds.  python tests/measure/soak_large.py [seconds] [first_seed]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import alfred_margaret_amd as am
from oracle import oracle
from tests.helpers import expand_records, oracle_triples
from tests.test_gpu_parity import _soak_case
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 150.0); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 70000; n = 0
while time.time() < t_end:
    rng = random.Random(seed)
    needles, hays = _soak_case(rng)
    for case in (0, 1):
        ns = [oracle.lower_utf8(x).decode() for x in needles] if case else needles
        o = oracle.Machine(ns); a = am.Automaton(ns)
        exp = oracle_triples(o, case, hays)
        recs = a.run_records(case, hays)
        assert expand_records(o.values_off(), o.values(), recs["haystack"], recs["state"], recs["end_pos"]) == exp, (seed, case)
        pairs = [(x, "<%d>" % i) for i, x in enumerate(needles[:120])]
        oo = oracle.Replacer(case, pairs); r = am.Replacer(case, pairs)
        sub = hays[:12]
        assert r.run_batch(sub) == [oo.run(h) for h in sub], ("replacer", seed, case)
        n += 1
    seed += 1
print("soak2 ok:", n, "cases")
