#!/usr/bin/env python3
"""GPU-box check of the Replacer's piece-table loop under its A/B switches (read once per process, hence a script: tests/test_gpu_parity.py runs it in
subprocesses with AM_RP_NO_FUSE / AM_RP_NO_SPIN / AM_RP_MAT_MAIN / AM_RP_NO_RANGE_REUSE set): batches whose passes alternate between the window
re-scan with merge (record ranges handed from pass to pass) and the whole-text re-scan of tiny texts, haystacks that finish at different passes, the
length limit -- every result equals the oracle's.  Prints "replacer toggles OK"."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import alfred_margaret_amd as am
from oracle import oracle

os.environ["AM_RP_PIECES"] = "1"                     # the piece-table loop whatever the batch shape
rng = random.Random(77)
alpha = "abcde "
n_checked = 0
for trial in range(6):
    n_pairs = rng.randint(8, 60)
    pairs = [("".join(rng.choice(alpha) for _ in range(rng.randint(2, 4))), "".join(rng.choice("ABC" + alpha) for _ in range(rng.randint(0, 5)))) for _ in range(n_pairs)]
    # a few long haystacks (their windows are small next to the text: window re-scan + merge) among many tiny ones (windows larger than the texts:
    # whole-text re-scan) -- which kind dominates changes as haystacks finish, so the loop switches between the two ways of making the next records
    hays = ["".join(rng.choice(alpha) for _ in range(rng.choice((2000, 6000)))) for _ in range(rng.randint(0, 3))]
    hays += ["".join(rng.choice(alpha) for _ in range(rng.choice((0, 1, 4, 9, 30)))) for _ in range(rng.randint(50, 400))]
    rng.shuffle(hays)
    for case in (0,):
        r = am.Replacer(case, pairs)
        o = oracle.Replacer(case, pairs)
        got = r.run_batch(hays)
        exp = [o.run(h) for h in hays]
        assert got == exp, ("trial", trial, [i for i, (g, e) in enumerate(zip(got, exp)) if g != e][:5])
        passes, _ = r.last_stats()
        n_checked += len(hays)
        lim = 40
        got_l = r.run_batch(hays, lim)
        assert got_l == [o.run(h, lim) for h in hays], ("trial", trial, "limit")
print("replacer toggles OK: %d haystacks, last batch %d passes" % (n_checked, passes))
