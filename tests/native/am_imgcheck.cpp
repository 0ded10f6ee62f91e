// am_imgcheck.cpp -- TEST-ONLY host interpreter of the device image (libam_imgcheck.so).
//
// Runs the very same per-position walk code the HIP kernels run (am_image.h) on the CPU, so the
// flattener and the walk logic can be differential-tested against the oracle in the GPU-less
// container.  It is NOT part of libam.so and no product entry point reaches it: the product has no
// CPU execution path at all (am_abi.cpp fails with AM_ERR_NO_DEVICE without a GPU).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "am_config.h"
#include "am_flatten.h"

using namespace am;

// which == 1 scans: candidates that passed the Bloom filter, positions the probe deferred, positions where a needle ends (measurement aid:
// how good the probe is on a given automaton + text, without a GPU)
static uint64_t g_stats[3] = {0, 0, 0};
static uint64_t g_walk_hist[16] = {0};      // walk steps per deferred pair of items (the interpreter resolves two in lock step): [i] = pairs whose deeper walk took i steps (15: or more)

namespace {
struct Rec { uint32_t hay; uint32_t state; uint64_t end_pos; uint32_t vlen; };
struct Collect {
    std::vector<Rec>* out;
    void operator()(uint32_t hay, uint64_t end_pos, uint32_t state, uint32_t vlen) { out->push_back({hay, state, end_pos, vlen}); }
};
}  // namespace

extern "C" {
void amchk_walk_hist(uint64_t* out16, int reset) { for (int i = 0; i < 16; i++) { out16[i] = g_walk_hist[i]; if (reset) g_walk_hist[i] = 0; } }
// a switch of csrc/am_config.h for the flattener compiled into THIS library (0 = done, -1 = no such switch)
int amchk_set(const char* name, long value) { return cfg::set(name, value) ? 0 : -1; }
void amchk_stats(uint64_t* out3, int reset) { for (int i = 0; i < 3; i++) { out3[i] = g_stats[i]; if (reset) g_stats[i] = 0; } }


// Flatten only: returns image size (or -1) -- lets tests inspect the header.
// (lower_from / lower_to / n_pairs: the caller's lower-case table as am_automaton_create_ex takes it; null = built-in)
long long amchk_flatten_ex(const uint64_t* transitions, size_t n_transitions, const uint32_t* offsets, size_t n_states,
                           const uint64_t* root_ascii, const uint32_t* values_len, int case_mode,
                           const uint32_t* lower_from, const uint32_t* lower_to, size_t n_pairs,
                           uint8_t* image_out, size_t image_cap, char* err_out, size_t err_cap)
{
    std::vector<uint8_t> img; std::string err;
    RefArrays ref{transitions, n_transitions, offsets, n_states, root_ascii, values_len};
    LowerTable lt;
    const bool custom = lower_from && lower_to;
    if ((custom && LowerTable::make(lower_from, lower_to, n_pairs, lt, err) != 0) || flatten(ref, case_mode, img, err, custom ? &lt : nullptr) != 0) {
        if (err_out && err_cap) { std::strncpy(err_out, err.c_str(), err_cap - 1); err_out[err_cap - 1] = 0; }
        return -1;
    }
    if (image_out && image_cap >= img.size()) std::memcpy(image_out, img.data(), img.size());
    return (long long)img.size();
}

long long amchk_flatten(const uint64_t* transitions, size_t n_transitions, const uint32_t* offsets, size_t n_states,
                        const uint64_t* root_ascii, const uint32_t* values_len, int case_mode,
                        uint8_t* image_out, size_t image_cap, char* err_out, size_t err_cap)
{
    return amchk_flatten_ex(transitions, n_transitions, offsets, n_states, root_ascii, values_len, case_mode, nullptr, nullptr, 0, image_out, image_cap, err_out, err_cap);
}

// Where do the steps of the table walk go?  One walk over `text` from the root; out[0] = steps at one of the first `hot_rows` row states (k_dfa keeps those rows in LDS),
// [1] = at the other row states (a 4-byte load from the table), [2] = at chain states that answer themselves (8 bytes, shared lines along a path), [3] = at chain states that ask
// their fallback's row (two loads), [4] = rare bytes, [5] = steps that land on a needle end.  -3: no DFA section.
long long amchk_dfa_stats(const uint8_t* image, const uint8_t* text, uint64_t n, uint32_t hot_rows, uint64_t* out6)
{
    ImageHeader h; std::memcpy(&h, image, sizeof(h));
    if (h.magic != kImageMagic || !h.dfa_n_states) return -3;
    const DfaView d = make_dfa_view(image, h);
    for (int i = 0; i < 6; i++) out6[i] = 0;
    uint32_t state = 0;
    for (uint64_t p = 0; p < n; p++) {
        uint32_t byte = text[p];
        const uint32_t cl = d.cls[byte];
        uint32_t e;
        if (cl == kDfaRare) { if (d.ic && byte - 0x41u < 26u) byte += 0x20u; e = dfa_rare_step(d, state, byte); out6[4]++; }
        else {
            if (state < d.n_rows) out6[state < hot_rows ? 0 : 1]++;
            else out6[(d.chain[state - d.n_rows].y >> 24) == cl ? 2 : 3]++;
            e = dfa_common_step(d, state, cl);
        }
        state = e & kDfaStateMask;
        if (e >> kDfaEndShift) out6[5]++;
    }
    return 0;
}

// Which states does the table walk visit?  visits[state] += 1 for the state a step STARTS from (one walk over `text` from the root); what an ordering of the rows by
// measured frequency would buy (tools/experiments/dfa_visits.py).  -3: no DFA section.
long long amchk_dfa_visits(const uint8_t* image, const uint8_t* text, uint64_t n, uint32_t* visits)
{
    ImageHeader h; std::memcpy(&h, image, sizeof(h));
    if (h.magic != kImageMagic || !h.dfa_n_states) return -3;
    const DfaView d = make_dfa_view(image, h);
    uint32_t state = 0;
    for (uint64_t p = 0; p < n; p++) {
        uint32_t byte = text[p];
        const uint32_t cl = d.cls[byte];
        visits[state]++;
        uint32_t e;
        if (cl == kDfaRare) { if (d.ic && byte - 0x41u < 26u) byte += 0x20u; e = dfa_rare_step(d, state, byte); }
        else e = dfa_common_step(d, state, cl);
        state = e & kDfaStateMask;
    }
    return 0;
}

// A model of what ONE XCD's L2 sees of the table walk (tools/experiments/dfa_l2sim.py): `lanes` lanes, lane i walking text[i * unit, (i + 1) * unit) from the root, take
// their steps round-robin; every lane-load that k_dfa would send to L2 -- a chain record, an entry of a row that is not in LDS (hot or cold table), a 64-byte piece of
// text -- looks its 128-byte line up in a set-associative LRU cache of l2_bytes.  out[2 * c] / out[2 * c + 1] = requests / misses of category c: 0 text, 1 chain
// records, 2 hot table, 3 cold columns of the rows, 4 rare-byte walk (hash + fail), 5 steps answered in LDS or without a load (requests only).  -3: no DFA section.
long long amchk_dfa_l2sim(const uint8_t* image, const uint8_t* text, uint64_t lanes, uint32_t unit, uint32_t hot_rows, uint64_t l2_bytes, uint32_t ways, uint32_t hot16_limit, uint64_t* out12)
{
    ImageHeader h; std::memcpy(&h, image, sizeof(h));
    if (h.magic != kImageMagic || !h.dfa_n_states) return -3;
    const DfaView d = make_dfa_view(image, h);
    std::fprintf(stderr, "[l2sim] image version %u: %u states, %u rows (%u columns, %.1f MB; hot table %.1f MB), %u single records, %u double records (%.1f MB)\n", h.version, d.n_states, d.n_rows,
                 1u << d.log2_classes, (double)((uint64_t)d.n_rows << d.log2_classes) * 4e-6, (double)((uint64_t)d.n_rows << d.hot_log2) * 4e-6, d.n_single, d.n_states - d.n_rows - d.n_single,
                 (double)d.n_single * 8e-6 + (double)(d.n_states - d.n_rows - d.n_single) * 16e-6);
    for (int i = 0; i < 12; i++) out12[i] = 0;
    const uint64_t n_sets = l2_bytes / 128u / ways;
    std::vector<uint64_t> tag((size_t)(n_sets * ways), ~0ull);
    std::vector<uint32_t> age((size_t)(n_sets * ways), 0);
    uint32_t clock = 0;
    const bool text_no_alloc = std::getenv("SIM_TEXT_NO_ALLOC") != nullptr;
    const uint32_t what_if_class = std::getenv("SIM_RESET_BYTE") ? d.cls[(uint8_t)std::atoi(std::getenv("SIM_RESET_BYTE"))] : 0u;
    uint64_t wi[2] = {0, 0};
    auto touch = [&](int cat, uint64_t addr) {
        const uint64_t line = addr >> 7, set = (line * 0x9E3779B97F4A7C15ull >> 20) % n_sets;
        out12[2 * cat]++;
        uint64_t* t = &tag[(size_t)(set * ways)]; uint32_t* a = &age[(size_t)(set * ways)];
        uint32_t victim = 0;
        for (uint32_t w = 0; w < ways; w++) { if (t[w] == line) { a[w] = ++clock; return; } if (a[w] < a[victim]) victim = w; }
        out12[2 * cat + 1]++;
        if (cat == 0 && text_no_alloc) return;                  // (what-if: text lines do not stay in the cache)
        t[victim] = line; a[victim] = ++clock;
    };
    // what-if (SIM_SPARSE=k): row states with at most k children of their own (entries that differ from their fallback's row) keep a compact record -- 8 bytes for one
    // child, 16 for two -- instead of a row, and a byte that is no child asks the fallback (which may be such a record again): more requests, fewer lines
    const uint32_t sparse_k = std::getenv("SIM_SPARSE") ? (uint32_t)std::atoi(std::getenv("SIM_SPARSE")) : 0u;
    std::vector<uint8_t> n_own; std::vector<uint64_t> rec_at;
    if (sparse_k) {
        n_own.assign(d.n_rows, 255); rec_at.assign(d.n_rows, 0);
        uint64_t at = 0;
        for (uint32_t r = 1; r < d.n_rows; r++) {
            const uint32_t f = d.fail[r];
            if (f >= d.n_rows) continue;
            uint32_t k = 0;
            for (uint32_t c = 1; c < (1u << d.log2_classes); c++) k += d.next[((uint64_t)r << d.log2_classes) + c] != d.next[((uint64_t)f << d.log2_classes) + c];
            if (k <= sparse_k && r >= hot_rows) { n_own[r] = (uint8_t)k; rec_at[r] = at; at += k <= 1 ? 8 : 16; }
        }
        std::fprintf(stderr, "[l2sim] sparse records: %.2f MB\n", (double)at / 1e6);
    }
    // address spaces: image offsets for the tables, 2^40 + offset for the text
    std::vector<uint32_t> state((size_t)lanes, 0);
    uint64_t trips_hist[16] = {0}, lane_hist[16] = {0}; uint32_t wave_max = 0;
    double rec_stat[6] = {0, 0, 0, 0, 0, 0};
    const uint32_t sim_window = std::getenv("SIM_WINDOW") ? (uint32_t)std::atoi(std::getenv("SIM_WINDOW")) : 0u;
    const uint32_t sim_window_bytes = std::getenv("SIM_WINDOW_BYTES") ? (uint32_t)std::atoi(std::getenv("SIM_WINDOW_BYTES")) : 16u;
    std::vector<uint32_t> win((size_t)lanes, 0), win_at((size_t)lanes, 0); uint64_t wfree = 0;
    for (uint32_t p = 0; p < unit; p++) {
        for (uint64_t l = 0; l < lanes; l++) {
            const uint64_t at = l * unit + p;
            if ((p & 63u) == 0) touch(0, (1ull << 40) + at);
            uint32_t byte = text[at];
            const uint32_t cl = d.cls[byte];
            uint32_t st = state[l], e;
            if ((l & 63u) == 0) { if (p || l) { trips_hist[wave_max < 15u ? wave_max : 15u]++; } wave_max = 0; }
            const uint64_t req_before = out12[2] + out12[4] + out12[6];
            if (cl == kDfaRare) {
                if (d.ic && byte - 0x41u < 26u) byte += 0x20u;
                touch(4, h.off_dfa_rare + (uint64_t)dfa_rare_slot(st, byte, d.rare_log2_cap) * 16u);
                e = dfa_rare_step(d, st, byte);
            } else {
                if (cl == 0u) out12[10]++;
                else {
                    while (st != kNone && st >= d.n_rows) {                                     // record states (image version 17): a child answers, else on to the fallback
                        if (st < d.n_rows + d.n_single) {
                            const u32x2 r = d.chain[st - d.n_rows];
                            // what-if (SIM_WINDOW=k): a single-child record is 16 bytes and also names the classes of the next k states of its path (numbered consecutively):
                            // a lane that came along the path and has the window in registers steps on without a load while the text follows the path
                            if (sim_window) {
                                const bool on_path = (r.x & kDfaStateMask) == st + 1u;
                                if (win[l] > 0 && st == win_at[l]) {
                                    if (on_path && (r.y >> 24) == cl) { win[l]--; win_at[l] = st + 1u; st = kNone; wfree++; continue; }
                                    win[l] = 0;                                                   // the text leaves the path (or the path ends here): this state's own record is read
                                }
                                touch(1, (3ull << 40) + (uint64_t)(st - d.n_rows) * (uint64_t)sim_window_bytes);
                                if ((r.y >> 24) == cl) { if (on_path) { win[l] = sim_window; win_at[l] = st + 1u; } st = kNone; }
                                else st = r.y & 0xFFFFFFu;
                                continue;
                            }
                            touch(1, h.off_dfa_chain + (uint64_t)(st - d.n_rows) * 8u);
                            st = (r.y >> 24) == cl ? kNone : (r.y & 0xFFFFFFu);
                            rec_stat[0]++; if (st != kNone) { rec_stat[1]++; if (st < hot_rows && cl <= 32u) rec_stat[2]++; if (cl == d.cls[0x20]) rec_stat[3]++; if ((r.y >> 24) == kDfaNoChild) rec_stat[4]++; }
                        } else {
                            rec_stat[5]++;
                            touch(1, h.off_dfa_chain2 + (uint64_t)(st - d.n_rows - d.n_single) * 16u);
                            const u32x4 q = d.chain2[st - d.n_rows - d.n_single];
                            st = ((q.y >> 24) == cl || (q.w >> 24) == cl) ? kNone : (q.y & 0xFFFFFFu);
                        }
                    }
                    while (sparse_k && st != kNone && st < d.n_rows && n_own[st] != 255) {      // a compact record: the child, or on to the fallback
                        touch(1, (3ull << 40) + rec_at[st]);
                        const uint32_t f = d.fail[st];
                        if (d.next[((uint64_t)st << d.log2_classes) + cl] != d.next[((uint64_t)f << d.log2_classes) + cl]) st = kNone; else st = f;
                    }
                    if (st != kNone) {
                        if (st < hot_rows && cl <= 32u) out12[10]++;
                        else if (cl <= (1u << d.hot_log2)) {
                            // (what-if, hot16_limit != 0: 16-bit entries for targets below the limit with at most two needle ends, a second request into the 32-bit table otherwise)
                            const uint32_t e32 = d.hot[((uint64_t)st << d.hot_log2) + cl - 1u];
                            if (hot16_limit) {
                                touch(2, (2ull << 40) + (((uint64_t)st << d.hot_log2) + cl - 1u) * 2u);
                                if ((e32 & kDfaStateMask) >= hot16_limit || (e32 >> kDfaEndShift) > 2u) touch(3, h.off_dfa_hot + (((uint64_t)st << d.hot_log2) + cl - 1u) * 4u);
                            } else touch(2, h.off_dfa_hot + (((uint64_t)st << d.hot_log2) + cl - 1u) * 4u);
                        }
                        else touch(3, h.off_dfa_next + (((uint64_t)st << d.log2_classes) + cl) * 4u);
                    }
                }
                e = dfa_common_step(d, state[l], cl);
                if (what_if_class && cl == what_if_class && e == 0u) {      // (what-if: steps on this class that lead to the root, and the requests they cost today)
                    wi[0]++;
                    const uint32_t s0 = state[l];
                    uint32_t req = 0, row = s0;
                    if (s0 >= d.n_rows) { req++; row = d.chain[s0 - d.n_rows].y & 0xFFFFFFu; }
                    if (!(row < hot_rows && cl <= 32u)) req++;
                    wi[1] += req;
                }
            }
            state[l] = e & kDfaStateMask;
            { const uint32_t tr = (uint32_t)(out12[2] + out12[4] + out12[6] - req_before); lane_hist[tr < 15u ? tr : 15u]++; if (tr > wave_max) wave_max = tr; }
        }
    }
    if (sim_window) std::fprintf(stderr, "[l2sim] window of %u: %.4f steps per step follow a path without a load\n", sim_window, (double)wfree / ((double)lanes * unit));
    std::fprintf(stderr, "[l2sim] per step: single-record reads %.4f, of them not answered by the record %.4f (the row it leans on is in LDS: %.4f; the byte is a blank: %.4f; the record has no entry: %.4f); double-record reads %.4f\n",
                 rec_stat[0] / ((double)lanes * unit), rec_stat[1] / ((double)lanes * unit), rec_stat[2] / ((double)lanes * unit), rec_stat[3] / ((double)lanes * unit), rec_stat[4] / ((double)lanes * unit), rec_stat[5] / ((double)lanes * unit));
    if (std::getenv("SIM_TRIPS")) {
        std::fprintf(stderr, "[l2sim] dependent table trips per lane step / max over the 64 lanes of a wavefront step:\n");
        uint64_t a = 0, b = 0; for (int i = 0; i < 16; i++) { a += lane_hist[i]; b += trips_hist[i]; }
        double ma = 0, mb = 0;
        for (int i = 0; i < 16; i++) { std::fprintf(stderr, "  %2d  %.4f  %.4f\n", i, (double)lane_hist[i] / a, (double)trips_hist[i] / b); ma += (double)i * lane_hist[i] / a; mb += (double)i * trips_hist[i] / b; }
        std::fprintf(stderr, "  mean %.3f  %.3f\n", ma, mb);
    }
    if (what_if_class) std::fprintf(stderr, "[l2sim] class %u: %llu steps lead to the root (%.4f per step), costing %llu requests today (%.4f per step)\n", what_if_class,
                                    (unsigned long long)wi[0], (double)wi[0] / ((double)lanes * unit), (unsigned long long)wi[1], (double)wi[1] / ((double)lanes * unit));
    return 0;
}

// Interpret an image over a batch.  which: 0 = AC walk (general kernel's logic), 1 = SF (filter +
// verify, fast kernel's logic), 2 = SF without the Bloom filter (every position verified: separates
// filter bugs from table bugs), 3 = the DFA table walk (k_dfa's logic).  Fills up to cap records sorted by (haystack, end_pos); returns the
// record count, or -2 if the image has no SF section (empty needle present).
long long amchk_scan(const uint8_t* image, int which, const uint8_t* text, const uint64_t* offsets, uint32_t n_hay,
                     uint32_t* hay_out, uint32_t* state_out, uint64_t* end_out, uint32_t* vlen_out, size_t cap)
{
    ImageHeader h; std::memcpy(&h, image, sizeof(h));
    if (h.magic != kImageMagic) return -1;
    const bool ic = h.case_mode == 1;
    const uint64_t total = offsets[n_hay];
    std::vector<Rec> recs;
    if (total > 0) {
        // padded private copy of the text, as the device batch guarantees
        std::vector<uint8_t> padded((size_t)((total + 15) & ~15ull) + 16, 0);
        std::memcpy(padded.data(), text, (size_t)total);
        std::vector<uint32_t> hidx((size_t)(total >> kHidxShift) + 2);
        for (size_t k = 0; k < hidx.size(); k++) {
            uint64_t p = std::min<uint64_t>((uint64_t)k << kHidxShift, total - 1);
            uint32_t lo = 0, hi = n_hay - 1;
            while (lo < hi) { uint32_t mid = lo + (hi - lo + 1) / 2; if (offsets[mid] <= p) lo = mid; else hi = mid - 1; }
            hidx[k] = lo;
        }
        BatchView b{padded.data(), offsets, hidx.data(), total, n_hay, 0};
        if (which == 0) {
            AcView a = make_ac_view(image, h);
            Collect c{&recs};
            const uint64_t units = (total + a.chunk - 1) / a.chunk;
            for (uint64_t u = 0; u < units; u++) { if (ic) ac_scan_unit<true>(a, b, u, c); else ac_scan_unit<false>(a, b, u, c); }
        } else if (which == 3) {
            // the table walk (k_dfa's logic): -3 when the image has no DFA section
            if (!h.dfa_n_states) return -3;
            DfaView d = make_dfa_view(image, h);
            Collect c{&recs};
            const uint64_t units = (total + d.chunk - 1) / d.chunk;
            for (uint64_t u = 0; u < units; u++) dfa_scan_unit(d, b, u, c);
        } else {
            if (!h.sf_enabled) return -2;
            SfView s = make_sf_view(image, h);
            if (s.tiers) {
                // candidates exactly as the kernel finds them, then verified two at a time (which == 1:
                // the ILP path) or one at a time without the Bloom filter (which == 2)
                std::vector<uint64_t> cands;
                for (uint64_t p = 0; p < total; p++) {
                    // window of the 4 bytes ending at p, exactly as the kernel builds it from dwords
                    uint32_t w = 0;
                    for (uint32_t j = 0; j < 4; j++) {
                        uint32_t byte = p >= j ? padded[(size_t)(p - j)] : 0u;
                        w |= byte << (24u - 8u * j);
                    }
                    if (ic) w = fold_dword(w);
                    if (which == 1 && !sf_filter_window(s.bloom, s.bloom_log2_words, s.tiers, w)) continue;
                    cands.push_back(p);
                }
                for (size_t i = 0; i < cands.size(); i += 2) {
                    uint64_t g[2] = {cands[i], i + 1 < cands.size() ? cands[i + 1] : 0}, a[2] = {0, 0};
                    bool valid[2] = {true, i + 1 < cands.size()}, found[2] = {false, false};
                    uint32_t hay[2] = {0, 0}, st[2] = {0, 0}, vl[2] = {0, 0};
                    for (int k = 0; k < 2; k++) if (valid[k]) { hay[k] = find_haystack(b, g[k]); a[k] = g[k] - offsets[hay[k]] + 1; }
                    if (which == 1) {
                        // what the kernel does: w / nb from the (folded) haystack bytes, two candidates per probe call
                        uint32_t wv[2] = {0, 0}, nbv[2] = {0, 0};
                        bool defer[2]; uint32_t hint[2] = {0, 0};
                        for (int k = 0; k < 2; k++) if (valid[k]) {
                            for (uint32_t j = 0; j < 4; j++) { uint32_t byte = g[k] >= j ? padded[(size_t)(g[k] - j)] : 0u; wv[k] |= byte << (24u - 8u * j); }
                            uint32_t b1 = g[k] >= 4 ? padded[(size_t)(g[k] - 4)] : 0u, b2 = g[k] >= 5 ? padded[(size_t)(g[k] - 5)] : 0u, b3 = g[k] >= 6 ? padded[(size_t)(g[k] - 6)] : 0u;
                            if (ic) { wv[k] = fold_dword(wv[k]); b1 = fold_byte(b1); b2 = fold_byte(b2); b3 = fold_byte(b3); }
                            nbv[k] = b1 | (b2 << 8) | (b3 << 16);
                        }
                        sf_probe_n<2>(s, wv, nbv, a, valid, defer, hint);
                        for (int k = 0; k < 2; k++) if (valid[k]) { g_stats[0]++; if (defer[k]) g_stats[1]++; }
                        // phase 2, two items in lock step as in the kernel
                        bool todo[2] = {valid[0] && defer[0], valid[1] && defer[1]};
                        uint64_t it[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        if (ic) sf_resolve_n<true, 2>(s, padded.data(), g, a, todo, hint, found, st, vl, SfNoHook(), it);
                        else sf_resolve_n<false, 2>(s, padded.data(), g, a, todo, hint, found, st, vl, SfNoHook(), it);
                        if (todo[0] || todo[1]) g_walk_hist[it[0] < 15 ? it[0] : 15]++;
                    } else {
                        // every position through the exact phase 2, with a slot hint that is wrong three times out of four: the
                        // probe only filters, so the answer must not depend on it (this is the path of a fingerprint collision)
                        for (int k = 0; k < 2; k++) if (valid[k]) {
                            const uint32_t hint = (uint32_t)(g[k] & 3u);
                            found[k] = ic ? sf_resolve<true>(s, padded.data(), g[k], a[k], st[k], vl[k], hint) : sf_resolve<false>(s, padded.data(), g[k], a[k], st[k], vl[k], hint);
                        }
                    }
                    for (int k = 0; k < 2; k++) if (valid[k] && found[k]) { recs.push_back({hay[k], st[k], a[k], vl[k]}); if (which == 1) g_stats[2]++; }
                }
            }
        }
    }
    if (which != 0 && h.root_vlen > 0 && total > 0) {
        // automaton with the empty needle: the dense part (what k_dense_* does on the device) -- every position where a first
        // code point ends and the suffix filter reported nothing folds the root's values (state 0)
        std::vector<uint8_t> padded((size_t)((total + 15) & ~15ull) + 16, 0);
        std::memcpy(padded.data(), text, (size_t)total);
        AcView a = make_ac_view(image, h);
        std::vector<Rec> merged;
        size_t r = 0;
        for (uint32_t hay = 0; hay < n_hay; hay++) {
            for (uint64_t g = offsets[hay]; g < offsets[hay + 1]; g++) {
                const uint64_t end_pos = g - offsets[hay] + 1;
                if (r < recs.size() && recs[r].hay == hay && recs[r].end_pos == end_pos) { merged.push_back(recs[r++]); continue; }
                if (ends_first_code_point(a, ic, padded.data(), offsets[hay], offsets[hay + 1], g)) merged.push_back({hay, 0u, end_pos, h.root_vlen});
            }
        }
        recs.swap(merged);
    }
    const size_t n = recs.size();
    for (size_t i = 0; i < n && i < cap; i++) {
        hay_out[i] = recs[i].hay; state_out[i] = recs[i].state; end_out[i] = recs[i].end_pos; vlen_out[i] = recs[i].vlen;
    }
    return (long long)n;
}

uint32_t amchk_simple_lower(uint32_t cp) { return simple_lower(cp); }

size_t amchk_unlower(uint32_t cp, uint32_t* out, size_t cap)
{
    std::vector<uint32_t> v; unlower(cp, v);
    for (size_t i = 0; i < v.size() && i < cap; i++) out[i] = v[i];
    return v.size();
}

}  // extern "C"
