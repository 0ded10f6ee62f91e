#!/usr/bin/env python3
"""GPU-box soak of the Replacer beyond the test suite: random pair sets and batches, both case modes, with and without a length limit; every
second seed forces the piece-table loop (AM_RP_PIECES is read per call).  Usage: python tests/measure/soak_replacer.py <first seed> <last seed + 1>
(round 3, final tree: seeds 0..199, no mismatch; 2 s per seed, most of it the oracle)."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import alfred_margaret_amd as am
from oracle import oracle

bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = random.Random(31000 + seed)
    alpha = rng.choice(["abcde ", "ab", "abİKß ", "xyzXYZ.", "0123456789"])
    n_pairs = rng.choice((1, 3, 10, 40, 150))
    pairs = [("".join(rng.choice(alpha) for _ in range(rng.randint(1, 5))), "".join(rng.choice(alpha + "Q") for _ in range(rng.randint(0, 6)))) for _ in range(n_pairs)]
    sizes = rng.choice(((0, 1, 5, 30), (0, 3, 50, 800), (10, 2000, 9000), (1, 2, 3, 4, 5, 6, 7, 8)))
    hays = ["".join(rng.choice(alpha) for _ in range(rng.choice(sizes))) for _ in range(rng.choice((1, 7, 60, 500)))]
    if seed & 1: os.environ["AM_RP_PIECES"] = "1"
    else: os.environ.pop("AM_RP_PIECES", None)
    for case in (0, 1):
        r = am.Replacer(case, pairs); o = oracle.Replacer(case, pairs)
        lim = rng.choice((-1, -1, 20, 300))
        got = r.run_batch(hays, lim)
        exp = [o.run(h, lim) for h in hays]
        if got != exp:
            bad += 1; print("MISMATCH seed", seed, "case", case, "limit", lim, [i for i, (g, e) in enumerate(zip(got, exp)) if g != e][:5], flush=True)
print("replacer soak seeds", sys.argv[1], sys.argv[2], "bad", bad)
