// am_synth.h -- deterministic synthetic haystack generator (benchmark/test input, not the hot path).
// Counter-based: cell g of the batch depends only on (seed, g), so the same bytes are produced by
// the HIP kernel (one lane per 1-KiB cell, inputs born in HBM) and by the host loop (CPU baseline,
// parity tests).  Cells are valid UTF-8 and exactly `cell_bytes` long; haystacks are whole numbers
// of cells.  Mix per SURVEY 8d: 90 % ASCII over the needle alphabet [a-z0-9 ], 8 % two-byte
// (Latin-1 / Greek / Cyrillic letters), 1.5 % three-byte, 0.5 % four-byte; one needle planted per
// cell.  mode 1 (IgnoreCase workloads) upper-cases 30 % of ASCII letters, also inside planted
// needles, and sprinkles U+0130 / U+1E9E / U+212A / U+212B (lower-casing changes their byte length).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SYN_HD __host__ __device__ __forceinline__
#else
#define SYN_HD inline
#endif

namespace amsynth {

struct Params {
    uint64_t seed;
    uint32_t cell_bytes;     // 1024
    uint32_t mode;           // 0: lower-case ASCII alphabet, 1: mixed case + special code points
    uint32_t n_needles;      // 0: plant nothing
    uint32_t plants;         // needles planted per cell: 0 and 1 = one (the BASELINE workloads), 2.. = that many, back to back with
                             // random gaps (robustness sweep: match density)
    // kind 1 ("natural text", robustness sweep): words drawn Zipf-like from a vocabulary through a quantile table, separated
    // by spaces and some punctuation -- the needles of such a workload are vocabulary words and phrases, so suffixes are
    // shared and matches are dense, the regime of the reference's real-world data set (README.md:14-25)
    uint32_t kind;           // 0: random code points, 1: natural text
    uint32_t n_quantile;     // entries of the quantile table (a power of two)
    const uint8_t* vocab_bytes;
    const uint64_t* vocab_offs;
    const uint32_t* quantile;   // quantile[k] = word at quantile (k + 0.5) / n_quantile of the word distribution
};

struct Rng {
    uint64_t s;
    SYN_HD uint64_t next()
    {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
};

SYN_HD uint32_t put_cp(uint8_t* out, uint32_t pos, uint32_t c)
{
    if (c < 0x80) { out[pos] = (uint8_t)c; return pos + 1; }
    if (c < 0x800) { out[pos] = (uint8_t)(0xc0 | (c >> 6)); out[pos + 1] = (uint8_t)(0x80 | (c & 0x3f)); return pos + 2; }
    if (c < 0x10000) { out[pos] = (uint8_t)(0xe0 | (c >> 12)); out[pos + 1] = (uint8_t)(0x80 | ((c >> 6) & 0x3f)); out[pos + 2] = (uint8_t)(0x80 | (c & 0x3f)); return pos + 3; }
    out[pos] = (uint8_t)(0xf0 | (c >> 18)); out[pos + 1] = (uint8_t)(0x80 | ((c >> 12) & 0x3f)); out[pos + 2] = (uint8_t)(0x80 | ((c >> 6) & 0x3f)); out[pos + 3] = (uint8_t)(0x80 | (c & 0x3f));
    return pos + 4;
}

SYN_HD uint32_t alphabet_char(uint32_t i)   // "abcdefghijklmnopqrstuvwxyz0123456789 "
{
    return i < 26 ? 'a' + i : i < 36 ? '0' + (i - 26) : ' ';
}

SYN_HD uint32_t random_cp(Rng& rng, uint32_t mode)
{
    const uint64_t r = rng.next();
    const uint32_t u = (uint32_t)(r % 1000u);
    if (mode == 1 && (uint32_t)((r >> 40) % 10000u) == 0) {
        const uint32_t k = (uint32_t)(r >> 60) & 3u;
        return k == 0 ? 0x130u : k == 1 ? 0x1E9Eu : k == 2 ? 0x212Au : 0x212Bu;
    }
    if (u < 900) {
        uint32_t c = alphabet_char((uint32_t)((r >> 10) % 37u));
        if (mode == 1 && c >= 'a' && c <= 'z' && (uint32_t)((r >> 20) % 10u) < 3) c -= 0x20;
        return c;
    }
    const uint32_t x = (uint32_t)(r >> 12);
    if (u < 980) {
        const uint32_t cls = (uint32_t)((r >> 10) & 3u) % 3u;
        if (cls == 0) return 0xC0u + x % 0x40u;
        if (cls == 1) { const uint32_t c = 0x391u + x % 0x39u; return c == 0x3A2u ? 0x3A3u : c; }
        return 0x410u + x % 0x40u;
    }
    if (u < 995) return 0x4E00u + x % 0x5000u;
    return 0x1F300u + x % 0x300u;
}

// kind 1: a cell of "natural text"
SYN_HD void generate_cell_natural(const Params& p, uint64_t g, uint8_t* out)
{
    Rng rng{p.seed ^ (g * 0xD1342543DE82EF95ull)};
    const uint32_t cell = p.cell_bytes;
    uint32_t pos = 0;
    bool capital = p.mode == 1;                       // mixed-case text: sentences start with a capital
    for (;;) {
        const uint64_t r = rng.next();
        const uint32_t w = p.quantile[(uint32_t)r & (p.n_quantile - 1u)];
        const uint64_t b = p.vocab_offs[w], e = p.vocab_offs[w + 1];
        if (pos + (e - b) + 2 > cell) break;
        const bool shout = p.mode == 1 && (uint32_t)((r >> 40) % 50u) == 0;           // an occasional WORD IN CAPITALS
        for (uint64_t k = b; k < e; k++) {
            uint32_t c = p.vocab_bytes[k];
            if ((shout || (capital && k == b)) && c >= 'a' && c <= 'z') c -= 0x20;
            out[pos++] = (uint8_t)c;
        }
        capital = false;
        const uint32_t sep = (uint32_t)((r >> 32) % 100u);
        if (sep < 6) { out[pos++] = ','; }
        else if (sep < 11) { out[pos++] = '.'; capital = p.mode == 1; }
        out[pos++] = ' ';
    }
    while (pos < cell) out[pos++] = ' ';
}

// Fills out[0 .. cell_bytes) with cell `g`.
SYN_HD void generate_cell(const Params& p, const uint8_t* needle_bytes, const uint64_t* needle_offs, uint64_t g, uint8_t* out)
{
    if (p.kind == 1) { generate_cell_natural(p, g, out); return; }
    Rng rng{p.seed ^ (g * 0xD1342543DE82EF95ull)};
    const uint32_t cell = p.cell_bytes;
    if (p.plants > 1 && p.n_needles) {
        // many needles per cell: needle, gap of random code points, needle, ... with the mean spacing cell / plants
        const uint32_t spacing = cell / p.plants;
        uint32_t pos = 0;
        while (pos + 4 <= cell) {
            const uint64_t r = rng.next();
            const uint32_t idx = (uint32_t)(r % p.n_needles);
            const uint64_t b = needle_offs[idx], e = needle_offs[idx + 1];
            if (pos + (e - b) + 4 > cell) break;
            for (uint64_t k = b; k < e; k++) {
                uint32_t c = needle_bytes[k];
                if (p.mode == 1 && c >= 'a' && c <= 'z' && (uint32_t)(rng.next() % 10u) < 3) c -= 0x20;
                out[pos++] = (uint8_t)c;
            }
            const uint32_t len = (uint32_t)(e - b);
            const uint32_t mean_gap = spacing > len ? spacing - len : 0u;
            uint32_t gap = mean_gap ? (uint32_t)((r >> 33) % (2u * mean_gap + 1u)) : 0u;
            while (gap-- && pos + 4 <= cell) pos = put_cp(out, pos, random_cp(rng, p.mode));
        }
        while (pos + 4 <= cell) pos = put_cp(out, pos, random_cp(rng, p.mode));
        while (pos < cell) out[pos++] = ' ';
        return;
    }
    const uint32_t plant_at = (uint32_t)(rng.next() % (uint64_t)(cell - 96u));
    bool planted = p.n_needles == 0;
    uint32_t pos = 0;
    while (pos + 4 <= cell) {
        if (!planted && pos >= plant_at) {
            planted = true;
            const uint64_t r = rng.next();
            const uint32_t idx = (uint32_t)(r % p.n_needles);
            const uint64_t b = needle_offs[idx], e = needle_offs[idx + 1];
            if (pos + (e - b) <= cell) {
                for (uint64_t k = b; k < e; k++) {
                    uint32_t c = needle_bytes[k];
                    if (p.mode == 1 && c >= 'a' && c <= 'z' && (uint32_t)(rng.next() % 10u) < 3) c -= 0x20;
                    out[pos++] = (uint8_t)c;
                }
            }
            continue;
        }
        pos = put_cp(out, pos, random_cp(rng, p.mode));
    }
    while (pos < cell) out[pos++] = ' ';
}

}  // namespace amsynth
