#!/usr/bin/env python3
"""Count-all-matches harness with the reference's CLI protocol (benchmark/haskell/app/Main.hs:26-76,
benchmark/README.md:20-32): each input file holds one needle per line, a blank line, then the haystack
(UTF-8).  For every file: 5 repetitions of (build automaton + count all matches) -- construction is
inside the timed region, as in the reference -- nanoseconds per repetition tab-separated on stdout,
the match count of the first repetition on stderr."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import alfred_margaret_amd as am  # noqa: E402


def read_needle_haystack_file(path):
    data = open(path, "rb").read()
    needles, i = [], 0
    while i < len(data):
        if data[i] == 10:                       # empty line: the rest is the haystack
            return needles, data[i + 1:]
        j = data.index(b"\n", i) if b"\n" in data[i:] else len(data)
        needles.append(data[i:j])
        i = j + 1
    return needles, b""


def count_matches(needles, haystack):
    if not needles:
        return 0                                # Main.hs:69
    a = am.Automaton(needles)
    return int(a.count_matches(am.CASE_SENSITIVE, [haystack])[0])


def main(paths):
    for path in paths:
        needles, haystack = read_needle_haystack_file(path)
        times = []
        for i in range(5):
            t0 = time.perf_counter_ns()
            n = count_matches(needles, haystack)
            times.append(time.perf_counter_ns() - t0)
            if i == 0:
                print(n, file=sys.stderr)
        print("\t".join(str(t) for t in times) + "\t")


if __name__ == "__main__":
    main(sys.argv[1:])
