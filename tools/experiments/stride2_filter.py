"""Round 6 (CPU only): the arithmetic of a stride-2 suffix filter BEFORE any GPU minute.  k_sf tests the 4-byte window ending at EVERY position against a
blocked Bloom filter in LDS (four bits of one 32-bit word per key; 2^15 words = 128 KiB).  Stride 2 would test only the windows ending at even positions, against a filter that holds
each needle's last window AND the window ending one byte earlier.  This script builds both filters with the kernel's own hash and masks from the ASCII-folded bytes of cfg3's
needles and counts what passes on a MiB of cfg3's text."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from alfred_margaret_amd import synth

wl = "cfg3_runLower_100k_10GiB"; w = synth.WORKLOADS[wl]
needles = synth.needles_for(wl)
LW = 15
MUL = np.uint32(0x9E3779B1)


def mask_entry(i):
    x = ((i + 1) * 0x9E3779B1) & 0xFFFFFFFF
    x ^= x >> 15; x = (x * 0x85EBCA6B) & 0xFFFFFFFF; x ^= x >> 13; x = (x * 0xC2B2AE35) & 0xFFFFFFFF; x ^= x >> 16
    m = 0
    for k in range(4):
        b = (x >> (5 * k)) & 31
        while m & (1 << b): b = (b + 1) & 31
        m |= 1 << b
    return m


MASKS = np.array([mask_entry(i) for i in range(512)], dtype=np.uint32)


def fold(b):
    a = np.frombuffer(b, dtype=np.uint8).copy()
    up = (a >= 0x41) & (a <= 0x5A)
    a[up] += 0x20
    return a


def windows(a):
    """window ending at position i (i >= 3): bytes i-3 .. i, newest byte on top -- as a uint32 per position"""
    a = a.astype(np.uint32)
    return (a[3:] << 24) | (a[2:-1] << 16) | (a[1:-2] << 8) | a[:-3]


def hashes(keys):
    return (keys.astype(np.uint64) * np.uint64(0x9E3779B1) & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def make_filter(keys):
    h = hashes(np.unique(keys))
    f = np.zeros(1 << LW, np.uint32)
    np.bitwise_or.at(f, h >> np.uint32(32 - LW), MASKS[(h >> np.uint32(2)) & np.uint32(511)])
    return f


def passes(f, win):
    h = hashes(win)
    m = MASKS[(h >> np.uint32(2)) & np.uint32(511)]
    return (f[h >> np.uint32(32 - LW)] & m) == m


last, prev = [], []
for n in needles:
    a = fold(n.encode())
    if len(a) >= 4: last.append(windows(a)[-1])
    if len(a) >= 5: prev.append(windows(a)[-2])
last, prev = np.array(last, np.uint32), np.array(prev, np.uint32)
f1, f12 = make_filter(last), make_filter(np.concatenate([last, prev]))
fill = lambda f: float(np.unpackbits(f.view(np.uint8)).mean())
text = fold(bytes(synth.haystacks_host(needles, w["mixed"], 0, 1024)))
win = windows(text)                      # win[j] ends at position j + 3
kib = len(text) / 1024.0
real1 = np.isin(win, last)
p1 = passes(f1, win)
even = (np.arange(len(win)) + 3) % 2 == 0
real12 = np.isin(win, np.concatenate([last, prev]))
p12 = passes(f12, win) & even
print("keys: %d last windows, %d windows one byte earlier (ASCII-folded bytes of %d needles)" % (len(np.unique(last)), len(np.unique(prev)), len(needles)))
print("today    : filter fill %.3f; per KiB %.1f positions tested, %.1f pass (%.1f real window hits, %.1f false)" % (fill(f1), len(win) / kib, p1.sum() / kib, (p1 & real1).sum() / kib, (p1 & ~real1).sum() / kib))
print("stride 2 : filter fill %.3f; per KiB %.1f positions tested, %.1f pass (%.1f real window hits, %.1f false); each stands for TWO end positions the probe must look at" % (
    fill(f12), even.sum() / kib, p12.sum() / kib, (p12 & real12).sum() / kib, (p12 & ~real12).sum() / kib))
