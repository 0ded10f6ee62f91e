"""Round 6 (CPU only): where the table walk's steps go on the natural-text workload, state by state -- what a better choice of the rows in LDS, 16-bit entries,
or chain records read two at a time could buy.  Runs the host interpreter (libam_imgcheck.so) over a few MiB of the workload's text."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import alfred_margaret_amd as am
from alfred_margaret_amd import synth
from tests.helpers import ImgCheck

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = "natural_100k_10GiB"; w = synth.WORKLOADS[wl]
needles = synth.needles_for(wl)
chk = ImgCheck()
a = am.Automaton(needles)
img = chk.flatten(a, w["case"])
hd = chk.dfa_header(img)
print(hd)
text = synth.haystacks_host(needles, w["mixed"], 0, mib * 1024, natural=True)
t = np.frombuffer(text, dtype=np.uint8)
visits = np.zeros(hd["n_states"], np.uint32)
chk.lib.amchk_dfa_visits.restype = C.c_longlong
assert chk.lib.amchk_dfa_visits(img.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), C.c_uint64(len(t)), visits.ctypes.data_as(C.c_void_p)) == 0
n = len(t); nr = hd["n_rows"]
rows, chains = visits[:nr].astype(np.int64), visits[nr:].astype(np.int64)
print("steps %d: at row states %.1f %%, at chain states %.1f %%" % (n, 100.0 * rows.sum() / n, 100.0 * chains.sum() / n))
cur = np.cumsum(rows); best = np.cumsum(np.sort(rows)[::-1])
for k in (256, 512, 1024, 2048, 2304, 4096, 8192, 16384, 32768, 65536):
    if k <= nr:
        print("first %6d rows: current numbering %.1f %% of all steps, by measured frequency %.1f %%" % (k, 100.0 * cur[k - 1] / n, 100.0 * best[k - 1] / n))
# how many row states are visited at all, and the rows it takes to cover 90 / 99 % of the row steps
print("row states visited: %d of %d; rows for 90 %% / 99 %% of the row steps by frequency: %d / %d" % ((rows > 0).sum(), nr, np.searchsorted(best, 0.9 * rows.sum()) + 1, np.searchsorted(best, 0.99 * rows.sum()) + 1))
# proxy the flattener could compute without any text: walk the needles themselves (joined by blanks) and count the visits
corpus = np.frombuffer((" ".join(needles)).encode(), dtype=np.uint8)
pv = np.zeros(hd["n_states"], np.uint32)
assert chk.lib.amchk_dfa_visits(img.ctypes.data_as(C.c_void_p), corpus.ctypes.data_as(C.c_void_p), C.c_uint64(len(corpus)), pv.ctypes.data_as(C.c_void_p)) == 0
order = np.argsort(-pv[:nr].astype(np.int64), kind="stable")
prox = np.cumsum(rows[order])
for k in (256, 512, 1024, 2048, 4096, 16384):
    print("first %6d rows by visits on the needle corpus: %.1f %% of all steps" % (k, 100.0 * prox[k - 1] / n))
# class frequencies of the text (classes are numbered by the number of EDGES that carry them, not by text frequency)
cls = img[hd["off_cls"]:hd["off_cls"] + 256]
cf = np.bincount(cls[t], minlength=256).astype(np.float64) / n
print("text bytes by class: " + " ".join("%d:%.3f" % (c, cf[c]) for c in range(1 << hd["log2_classes"])))
for k in (8, 16, 32, 64):
    print("classes < %d: %.1f %% of the text's bytes; rare (255): %.2f %%" % (k, 100.0 * cf[:k].sum(), 100.0 * cf[255]))
srt = np.sort(cf[:64])[::-1]
print("the 8 / 16 / 32 most frequent classes of THIS text: %.1f / %.1f / %.1f %%" % (100 * srt[:8].sum(), 100 * srt[:16].sum(), 100 * srt[:32].sum()))
# second proxy: every SUFFIX of every needle, separated by a byte no needle contains (text words that are not in the dictionary end in states reached through fallbacks)
parts = []
for w_ in needles:
    bts = w_.encode()
    parts.extend(bts[j:] for j in range(len(bts)))
corpus2 = np.frombuffer(b"\n".join(parts), dtype=np.uint8)
pv2 = np.zeros(hd["n_states"], np.uint32)
assert chk.lib.amchk_dfa_visits(img.ctypes.data_as(C.c_void_p), corpus2.ctypes.data_as(C.c_void_p), C.c_uint64(len(corpus2)), pv2.ctypes.data_as(C.c_void_p)) == 0
for name, v in (("suffix corpus", pv2), ("both", pv2.astype(np.int64) + 4 * pv.astype(np.int64))):
    order2 = np.argsort(-v[:nr].astype(np.int64), kind="stable")
    prox2 = np.cumsum(rows[order2])
    print(name + ": " + ", ".join("%d: %.1f %%" % (k, 100.0 * prox2[k - 1] / n) for k in (256, 512, 1024, 2048, 4096, 16384, 32768)))
# chain records: how concentrated are the visits?  (8 bytes each; a 128-byte line holds 16)
cv = np.sort(chains)[::-1]; cc = np.cumsum(cv); tot = max(1, chains.sum())
lines_cur = chains[: (len(chains) // 16) * 16].reshape(-1, 16).sum(axis=1)
lc_sorted = np.cumsum(np.sort(lines_cur)[::-1])
print("chain states %d (%.1f MB), visited %d; the hottest 32k / 64k / 128k records take %.1f / %.1f / %.1f %% of the chain steps; in the current numbering the hottest 2k / 4k / 8k / 16k LINES take %.1f / %.1f / %.1f / %.1f %%" % (
    len(chains), len(chains) * 8 / 1e6, (chains > 0).sum(), 100.0 * cc[32767] / tot, 100.0 * cc[65535] / tot, 100.0 * cc[131071] / tot,
    100.0 * lc_sorted[2047] / tot, 100.0 * lc_sorted[4095] / tot, 100.0 * lc_sorted[8191] / tot, 100.0 * lc_sorted[16383] / tot))
