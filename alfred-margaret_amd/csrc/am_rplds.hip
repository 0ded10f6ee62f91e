// am_rplds.hip -- Replacer.run, all passes of a haystack in one wavefront WITH THE HAYSTACK'S LISTS IN LDS (round 5; reference:
// src/Data/Text/AhoCorasick/Replacer.hs:203-274).
//
// k_rp_loop (am_rploop.hip, round 4) keeps a haystack's record list and piece list in global memory: every pass re-reads the whole record list,
// rewrites the piece list, moves half the records -- ~13 dependent trips through L2 per pass (2 000-3 000 cycles each under load) and 16 bytes of HBM
// traffic per input byte (profiles/r04_pmc_traffic.md).  But the working set of a haystack is small -- a few hundred records, a few hundred pieces --
// and it lives for ~100 passes: it belongs in LDS.  Here one wavefront (= one workgroup) owns a haystack and 10 KiB of LDS:
//   records   up to 512, as {end position, priority, payload} (12 B; priority and payload looked up ONCE, when a record enters the list -- the
//             fold of a pass is three sweeps over LDS, no table look-up)
//   pieces    up to 448, as {source, logical start} (8 B; sources relative to the haystack / the replacement blob), edited IN PLACE
//   window    the re-scanned stretch of the new text (up to 448 bytes), gathered through the piece list and read back by the verifying lanes
// A pass touches global memory for: the selected payload (one uniform load), the window's bytes (one gather), the window's probe / resolve
// look-ups (sf_verify, as k_sf runs it) and the state entries of the records the window adds.  The kept matches of a pass are applied one at a time,
// from the LAST to the first: replacing one match is a self-contained step on a consistent (text, records) pair -- records ending at or before the match
// stay, those ending more than `ov` bytes behind it move with the text, those in between are dropped and that stretch of the NEW text is scanned again --
// and right-to-left the positions left of the match are still the pass's own.  After the last step the list is the scan of the new text, which is what the
// pass-by-pass loops and k_rp_loop compute.  Records that can never be chosen again (priority at or above the pass's -- the threshold only falls,
// Replacer.hs:236-242) are not entered.
// 16 workgroups per CU by LDS = 4 wavefronts per SIMD, the occupancy sf_verify's registers allow anyway.
// A haystack that does not fit (more records / pieces than the LDS lists hold, positions beyond 2^31, more than 64 new records in one window, a pass
// that keeps more matches than its kept list holds) raises ITS redo flag and k_rp_loop -- launched right behind this kernel -- runs that haystack from
// its first scan; nothing is shared between haystacks, so nothing else repeats.
#include <hip/hip_runtime.h>

#include "am_device.h"
#include "am_bounds.h"
#include "am_config.h"
#include "am_wave.h"

AM_BOUNDS_TU("am_rplds.hip")

namespace am {
namespace dev {

namespace {

constexpr int kWave = 64;
constexpr uint32_t kLdsRec = 512;                         // records a haystack may hold (8 blocks of 64: the fold's sweeps are unrolled over them)
constexpr uint32_t kLdsBlocks = kLdsRec / kWave;
constexpr uint32_t kLdsPc = 448;                          // pieces (+ the sentinel)
constexpr uint32_t kReplBit = 0x80000000u;                // piece source: offset into the replacement blob instead of the haystack
constexpr uint32_t kLdsWin = 448;                         // bytes of a re-scan window (the replacement and `ov` bytes either side of it); a longer one: k_rp_loop

// PLI ("payload implicit", round 6): the replacer's payloads carry priority = -index -- what Replacer.build makes of a needle list (Replacer.hs:100-104) --, so a record's
// payload is its priority negated and the list needs no payload column: 8 160 instead of 10 208 bytes of LDS per haystack and eight registers less in the fold, which is
// what a FIFTH wavefront per SIMD needs (20 haystacks per CU instead of 16).  A caller's own priorities (am_replacer_create takes any distinct ones) keep the column.
template <bool PLI>
struct LpLds {
    uint32_t end[kLdsRec]; int32_t prio[kLdsRec]; uint32_t pl[PLI ? 1 : kLdsRec];
    uint32_t psrc[kLdsPc + 2]; uint32_t pls[kLdsPc + 2];
    alignas(16) uint8_t win[kLdsWin + 16];                // the window's text: gathered through the piece list, read back by the probe / resolve lanes
};

__device__ __forceinline__ int64_t ld_wave_max_i64(int64_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int64_t o = __shfl_xor(v, d, kWave); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ int64_t ld_wave_sum_i64(int64_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
    return v;
}
// maximum over the 64 lanes with DPP (the shape of am_wave.h's prefix sum: inside each row of 16, then across the rows); uniform result
__device__ __forceinline__ int32_t ld_wave_max_i32(int32_t x)
{
    x = [&] { const int32_t o = __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x111, 0xf, 0xf, false); return o > x ? o : x; }();      // row_shr:1
    x = [&] { const int32_t o = __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x112, 0xf, 0xf, false); return o > x ? o : x; }();      // row_shr:2
    x = [&] { const int32_t o = __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x114, 0xf, 0xf, false); return o > x ? o : x; }();      // row_shr:4
    x = [&] { const int32_t o = __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x118, 0xf, 0xf, false); return o > x ? o : x; }();      // row_shr:8
    x = [&] { const int32_t o = __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x142, 0xa, 0xf, false); return o > x ? o : x; }();      // row_bcast:15 -> rows 1, 3
    x = [&] { const int32_t o = __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x143, 0xc, 0xf, false); return o > x ? o : x; }();      // row_bcast:31 -> rows 2, 3
    return __builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ uint32_t ld_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// what the lanes of this wavefront wrote to GLOBAL memory is visible to its other lanes (the window scratch, the kept list)
__device__ __forceinline__ void ld_global_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// elements [from, to) of an LDS array move to [from + g, to + g), `add` added to each: 64 per trip, from the top down when they move up (a trip's
// reads are issued before its writes and DS operations of a wavefront execute in order; across trips the writes land where everything has been read)
__device__ __forceinline__ void ld_move(uint32_t* A, uint32_t from, uint32_t to, int32_t g, uint32_t add, int lane)
{
    if (to <= from || (g == 0 && add == 0)) return;
    const uint32_t nb = (to - from + kWave - 1) / kWave;
    for (uint32_t b = 0; b < nb; b++) {
        const uint32_t blk = g > 0 ? nb - 1 - b : b;
        const uint32_t i = from + blk * kWave + (uint32_t)lane;
        uint32_t x = 0;
        if (i < to) x = A[i];
        wave_lds_fence();
        AM_BOUNDS(i >= to || ((int32_t)i + g >= 0 && (uint32_t)((int32_t)i + g) < kLdsRec));
        if (i < to) A[(uint32_t)((int32_t)i + g)] = x + add;
        wave_lds_fence();
    }
}

constexpr uint64_t kLdMaxTicks = 4000000000ull;           // watchdog, as in k_rp_loop: ~2 s for ONE haystack, then the redo flag

// DBG (AM_RP_TRACE >= 3): s_memtime per phase of every pass, summed over all haystacks into ctrl[8..27] (64-bit: records in + fold, select + payload, overlap removal,
// counts + dead slots, piece list, gather, window scan, inserts, the whole run, passes)
template <bool IC, bool DBG, bool PLI>
__device__ __forceinline__ void ld_run_haystack(const RpLoop& a, LpLds<PLI>& L, const uint32_t h, const int lane)
{
    // the payload of record slot r, given its priority (a state with several values holds its -- positive -- state id there; kDead: no record)
    auto set_pl = [&](uint32_t r, uint32_t v) { if (!PLI) L.pl[r] = v; };
    auto pl_of = [&](int32_t pr, uint32_t column) -> uint32_t { return !PLI ? column : pr > 0 ? kRpWalkList : (uint32_t)(-pr); };
    const uint64_t deadline = __builtin_amdgcn_s_memtime() + kLdMaxTicks;
    uint64_t t_mark = DBG ? __builtin_amdgcn_s_memtime() : 0, t_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t t_begin = t_mark;
    auto tick = [&](int ph) { if (DBG) { __builtin_amdgcn_s_waitcnt(0); const uint64_t now = __builtin_amdgcn_s_memtime(); t_ph[ph] += now - t_mark; t_mark = now; } };
    const uint64_t hoff = uniform_u64(a.offsets[h]);
    const uint64_t len0 = uniform_u64(a.offsets[h + 1]) - hoff;
    const uint64_t rb = uniform_u64(a.rec_base[h]), cap_r = (uniform_u64(a.rec_base[h + 1]) - rb) >> 1;
    const uint64_t pb = uniform_u64(a.pc_base[h]), cap_p = uniform_u64(a.pc_base[h + 1]) - pb;
    const uint64_t rf0 = uniform_u64(a.rec_first0[h]);
    const uint64_t nr0 = uniform_u64(a.rec_first0[h + 1]) - rf0;
    RpKept* const K = a.kept_buf + (rb >> 1);                           // kept matches of a pass that keeps several (cap_r entries)
    const uint8_t* const htext = a.text + hoff;
    bool redo = len0 >= 0x7FFFF000ull || nr0 > kLdsRec || cap_p < 4;
    uint32_t nr = 0, np = 1, passes = 0, status = kRpFinished;
    uint64_t curlen = len0, scanned = 0;
    int64_t threshold = 1;                                               // initialThreshold (Replacer.hs:211)
    auto timed_out = [&](uint32_t code) -> bool {
        if (__builtin_amdgcn_s_memtime() <= deadline) return false;
        if (lane == 0) atomicMax(a.ctrl + 5, 100u + code);
        return true;
    };

    constexpr uint32_t kPcBlocks = (kLdsPc + 1 + kWave - 1) / kWave;      // blocks of 64 piece entries (sentinel included)
    constexpr int32_t kDead = 0x7FFFFFFF;                                  // priority of a record that is gone: never below a threshold (its slot is reused by the next window's records)
    constexpr uint32_t kNoPos = 0xFFFFFFFFu;
    // Invariants that keep the sweeps below free of bounds tests: record slots from nr on hold {end = kNoPos, priority = kDead}, piece slots behind
    // the sentinel (index np: the text's length) hold start = kNoPos.
    bool has_walk = false;                                               // some record's state carries several values (its list is walked: the slow side of the fold)
    if (!redo) {
        // ---- the first scan's records of this haystack enter the LDS lists with their priority and payload
        nr = (uint32_t)nr0;
#pragma unroll
        for (uint32_t b = 0; b < kLdsBlocks; b++) {
            const uint32_t r = b * kWave + (uint32_t)lane;
            uint32_t e = kNoPos, pl = 0; int32_t pr = kDead;
            if (r < nr) {
                const Record rec = a.recs0[rf0 + r];
                const RpStateOne one = a.t.one[rec.state];
                e = (uint32_t)rec.end_pos; pl = one.payload;
                pr = one.payload != kRpWalkList ? one.priority : (int32_t)rec.state;      // several values: the state (> 0, never below a threshold), its list is walked
            }
            L.end[r] = e; L.prio[r] = pr; set_pl(r, pl);
            has_walk = has_walk || __ballot(pl == kRpWalkList) != 0ull;
        }
#pragma unroll
        for (uint32_t b = 0; b < kPcBlocks; b++) { const uint32_t k = b * kWave + (uint32_t)lane; if (k < kLdsPc + 2u) { L.pls[k] = k == 0 ? 0u : k == 1 ? (uint32_t)len0 : kNoPos; L.psrc[k] = 0; } }
        wave_lds_fence();
    }

    // byte at logical position p of the text the LDS piece list describes, piece index known
    auto byte_in = [&](uint32_t idx, uint32_t p) -> uint32_t {
        const uint32_t s = L.psrc[idx];
        const uint8_t* base = (s & kReplBit) ? a.t.repl + (s & ~kReplBit) : htext + s;
        return base[p - L.pls[idx]];
    };

    while (!redo) {
        passes++;
        if ((passes & 15u) == 0u && timed_out(1)) { redo = true; break; }      // (s_memtime is a scalar memory round trip: not in every pass)
        // ---- the record list, ONE batch of LDS reads (24 in flight, one wait): lane l holds records l, 64 + l, ... for the fold, the overlap removal and
        // the counts of the first replacement; nothing below reads a record from LDS again unless a pass keeps several matches
        uint32_t e_[kLdsBlocks], l_[kLdsBlocks]; int32_t p_[kLdsBlocks];
#pragma unroll
        for (uint32_t b = 0; b < kLdsBlocks; b++) { const uint32_t r = b * kWave + (uint32_t)lane; e_[b] = L.end[r]; p_[b] = L.prio[r]; l_[b] = PLI ? 0u : L.pl[r]; }
        // ---- prependMatch, first half (Replacer.hs:255-258): the best priority below the threshold.  Priorities of single-valued states are 32-bit and
        // <= 0, dead and unused slots hold kDead, states with several values hold their (positive) state id: one compare + select + max per block
        const int32_t thr32 = threshold < (int64_t)INT32_MIN ? INT32_MIN : (int32_t)threshold;      // (threshold <= 1)
        int32_t b32 = INT32_MIN;
#pragma unroll
        for (uint32_t b = 0; b < kLdsBlocks; b++) { const int32_t v = p_[b] < thr32 ? p_[b] : INT32_MIN; b32 = v > b32 ? v : b32; }
        b32 = ld_wave_max_i32(b32);
        int64_t best = b32 == INT32_MIN ? INT64_MIN : (int64_t)b32;
        if (has_walk) {                                                  // ... and the value lists of the states that carry several
            int64_t bw = INT64_MIN;
#pragma unroll
            for (uint32_t b = 0; b < kLdsBlocks; b++) {
                if (pl_of(p_[b], l_[b]) == kRpWalkList && p_[b] != kDead) {
                    const uint32_t st = (uint32_t)p_[b];
                    for (uint64_t k = a.t.vals_off[st], ke = a.t.vals_off[st + 1]; k < ke; k++) {
                        const int64_t p = a.t.payloads[a.t.vals[k]].priority;
                        if (p < threshold && p > bw) bw = p;
                    }
                }
            }
            bw = (int64_t)uniform_u64((uint64_t)ld_wave_max_i64(bw));
            if (bw > best) best = bw;
        }
        tick(0);
        if (best == INT64_MIN) { status = kRpFinished; break; }           // no match below the threshold: the text stays (:228-230)

        // ---- which records carry it: priorities are distinct, so every one of them has the same payload
        uint64_t selmask[kLdsBlocks];
        uint32_t payload = 0;
        const bool fast_sel = best >= (int64_t)INT32_MIN + 1 && best <= 0;
        const int32_t best32 = fast_sel ? (int32_t)best : kDead - 1;      // (no slot holds kDead - 1)
#pragma unroll
        for (uint32_t b = 0; b < kLdsBlocks; b++) {
            selmask[b] = __ballot(p_[b] == best32);
            if (selmask[b]) payload = PLI ? (uint32_t)(-best32) : (uint32_t)__builtin_amdgcn_readlane((int)l_[b], __ffsll((unsigned long long)selmask[b]) - 1);
        }
        if (has_walk) {
            uint32_t pw = 0;
#pragma unroll
            for (uint32_t b = 0; b < kLdsBlocks; b++) {
                bool sel = false;
                if (pl_of(p_[b], l_[b]) == kRpWalkList && p_[b] != kDead) {
                    const uint32_t st = (uint32_t)p_[b];
                    for (uint64_t k = a.t.vals_off[st], ke = a.t.vals_off[st + 1]; k < ke; k++) {
                        const uint32_t v = a.t.vals[k];
                        if (a.t.payloads[v].priority == best) { sel = true; pw = v; }
                    }
                }
                selmask[b] |= __ballot(sel);
            }
            pw = ld_u32((uint32_t)ld_wave_max_i64((int64_t)pw));
            if (pw > payload) payload = pw;
        }
        RpPayload pp = a.t.payloads[payload];                             // uniform index: one load for the pass (requesting it a pass ahead on a guess was measured: no gain)
        const uint32_t m_len = ld_u32(pp.len_bytes), m_cps = ld_u32(pp.len_code_points), rl = ld_u32(pp.repl_len);
        const uint32_t repl_off = ld_u32((uint32_t)pp.repl_off);
        tick(1);

        // ---- makeMatch (:264-274) + removeOverlap (:191-198): 64 records at a time, in position order (positions are 32-bit here)
        int32_t delta_sum = 0;                                           // IgnoreCase: the sum of (replacement - match) over this lane's selected records
        uint32_t n_sel = 0;
        uint32_t last_end = 0, k0_start = 0, k0_len = 0, nkept = 0;
#pragma unroll
        for (uint32_t b = 0; b < kLdsBlocks; b++) {
            if (selmask[b] == 0) continue;                                // (uniform)
            n_sel += (uint32_t)__popcll(selmask[b]);
            const bool sel = (selmask[b] >> lane) & 1ull;
            const uint32_t end_pos = sel ? e_[b] : 0u;
            uint32_t len = m_len, start = end_pos - m_len;                // CaseSensitive (:266-267)
            if (IC) {
                if (sel) {                                               // IgnoreCase: as long as its code points are in the haystack (skipCodePointsBackwards, Utf8.hs:256-276)
                    if (m_cps == 0) start = end_pos;
                    else {
                        uint32_t lo = 0, hi = np;                        // last piece that starts at or before the match's last byte
                        const uint32_t index = end_pos - 1u;
                        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (L.pls[mid] <= index) lo = mid; else hi = mid; }
                        uint32_t pi = lo;
                        int64_t i = (int64_t)index; uint32_t n = m_cps - 1u;
                        for (;;) {
                            for (;;) {
                                if (i <= 0) break;
                                while ((uint32_t)i < L.pls[pi]) pi--;     // (pieces may be empty: a loop)
                                if ((byte_in(pi, (uint32_t)i) & 0xC0u) != 0x80u) break;      // atTrailingByte
                                i--;
                            }
                            if (n == 0 || i <= 0) break;
                            i--; n--;
                        }
                        start = (uint32_t)(i < 0 ? 0 : i);
                    }
                    len = end_pos - start;
                    delta_sum += (int32_t)rl - (int32_t)len;
                }
            }
            uint64_t pending = selmask[b];
            bool keep = false;
            while (pending) {
                const uint64_t ok = __ballot(sel && start >= last_end) & pending;
                if (!ok) break;
                const int l = __ffsll((unsigned long long)ok) - 1;
                if (lane == l) keep = true;
                last_end = (uint32_t)__builtin_amdgcn_readlane((int)(start + len), l);
                pending &= l == 63 ? 0ull : ~((2ull << l) - 1ull);
            }
            const uint64_t keepmask = __ballot(keep);
            if (keepmask) {
                const uint32_t nk = (uint32_t)__popcll(keepmask);
                if ((uint64_t)nkept + nk > cap_r) { redo = true; break; }
                if (nkept == 0) {                                        // the first kept match stays in scalar registers (a pass usually keeps one)
                    const int l = __ffsll((unsigned long long)keepmask) - 1;
                    k0_start = (uint32_t)__builtin_amdgcn_readlane((int)start, l); k0_len = (uint32_t)__builtin_amdgcn_readlane((int)len, l);
                }
                if (nkept + nk > 1u && keep) {                           // several: the list goes through the haystack's kept region (global memory)
                    const uint32_t rank = (uint32_t)__popcll(keepmask & ((1ull << lane) - 1ull));
                    RpKept e; e.src_start = start; e.src_len = len; e.dst = 0;
                    K[nkept + rank] = e;
                }
                if (nkept == 1u && lane == 0) { RpKept e; e.src_start = k0_start; e.src_len = k0_len; e.dst = 0; K[0] = e; }      // (the one held in registers joins it)
                nkept += nk;
            }
        }
        if (redo) break;
        int64_t delta_all = (int64_t)n_sel * ((int64_t)rl - (int64_t)m_len);      // replacementLength over ALL matches (:240), before removeOverlap
        if (IC) delta_all = (int64_t)uniform_u64((uint64_t)ld_wave_sum_i64((int64_t)delta_sum));
        const int64_t newlen_all = (int64_t)curlen + delta_all;
        if (newlen_all > 0 && (uint64_t)newlen_all > a.max_len) { status = kRpNothing; break; }
        status = best == a.t.min_priority ? kRpFinished : kRpActive;    // :241-242
        if (nkept > 1) ld_global_sync();                                  // K is read back below
        tick(2);

        // ---- replace (:163-180), one kept match at a time, the last one first
        for (uint32_t jj = nkept; jj-- > 0 && !redo;) {
            if ((jj & 15u) == 15u && timed_out(2)) { redo = true; break; }
            uint32_t ms = k0_start, ml = k0_len;
            if (nkept > 1) { const RpKept k = K[jj]; ms = ld_u32((uint32_t)k.src_start); ml = ld_u32((uint32_t)k.src_len); }
            const uint32_t me = ms + ml;
            const int32_t delta = (int32_t)rl - (int32_t)ml;
            const uint64_t newlen = (uint64_t)((int64_t)curlen + delta);
            if (newlen >= 0x7FFFF000ull) { redo = true; break; }
            const bool last_pass = status == kRpFinished;                 // (:241) nobody looks at the records again: the piece list alone is edited
            uint32_t hi = ms + rl + a.ov;                                 // the window of the replacement in the new text: [ws, hi), its own positions from ms on
            if ((uint64_t)hi > newlen) hi = (uint32_t)newlen;
            const uint32_t ws = ms > a.ov ? ms - a.ov : 0u;
            const uint32_t wlen = hi > ms ? hi - ws : 0u, own_lo = ms - ws;
            if (wlen > kLdsWin) { redo = true; break; }

            // (b) old records: [0, c_before) end at or before the match's start and stay; [c_before, c_gone) end within reach of its end and are gone --
            // their slots are marked dead where they are (the window's records go there); the rest shift with the text.  From the registers of the pass's
            // batch read for the first replacement; a pass that keeps several matches reads the ends again.
            uint32_t c_before = 0, c_gone = 0;
            if (!last_pass) {
                if (jj + 1u != nkept) {
#pragma unroll
                    for (uint32_t b = 0; b < kLdsBlocks; b++) e_[b] = L.end[b * kWave + (uint32_t)lane];
                }
                const uint32_t x2 = me + a.ov;
#pragma unroll
                for (uint32_t b = 0; b < kLdsBlocks; b++) {              // (unused slots hold kNoPos: never counted)
                    c_before += (uint32_t)__popcll(__ballot(e_[b] <= ms));
                    c_gone += (uint32_t)__popcll(__ballot(e_[b] <= x2));
                }
#pragma unroll
                for (uint32_t b = 0; b < kLdsBlocks; b++) {
                    const uint32_t r0 = b * kWave;
                    if (r0 >= nr || r0 + kWave <= c_before) continue;    // (uniform) nothing of this block changes
                    const uint32_t r = r0 + (uint32_t)lane;
                    if (r0 < c_gone) {                                   // the block holds dead slots (end = the window's upper bound: the list stays sorted)
                        if (r >= c_before && r < c_gone) { L.end[r] = hi; L.prio[r] = kDead; set_pl(r, 0u); }
                        else if (r >= c_gone && r < nr) L.end[r] = (uint32_t)((int32_t)e_[b] + delta);
                    } else if (delta != 0) {
                        if (r0 + kWave <= nr) L.end[r] = (uint32_t)((int32_t)e_[b] + delta);
                        else if (r < nr) L.end[r] = (uint32_t)((int32_t)e_[b] + delta);
                    }
                }
            }
            tick(3);

            // (a) the piece list, in place: pieces i .. j2 hold the match; what is left of them is a head, the replacement, a tail.  One batch of reads
            // answers "which piece holds byte x" for the match's first and last byte and for the window's first byte (slots behind the sentinel hold
            // kNoPos, the sentinel the text's length: neither is ever <= a position inside the text).
            uint32_t c_ms = 0, c_me = 0, c_ws = 0;
            {
                uint32_t v[kPcBlocks];
#pragma unroll
                for (uint32_t b = 0; b < kPcBlocks; b++) { const uint32_t k = b * kWave + (uint32_t)lane; v[b] = L.pls[k < kLdsPc + 2u ? k : kLdsPc + 1u]; }
#pragma unroll
                for (uint32_t b = 0; b < kPcBlocks; b++) {
                    if (b * kWave >= np) continue;                        // (uniform)
                    c_ms += (uint32_t)__popcll(__ballot(v[b] <= ms));
                    c_me += (uint32_t)__popcll(__ballot(v[b] <= me - 1u));      // the match is not empty here (automata with the empty needle do not take this route)
                    c_ws += (uint32_t)__popcll(__ballot(v[b] <= ws));
                }
            }
            const uint32_t i = c_ms - 1u, j2 = c_me - 1u, first = c_ws - 1u;     // (>= 0: piece 0 starts at 0; first <= i: ws <= ms)
            const uint32_t pi_ls = ld_u32(L.pls[i]);
            const uint32_t pj_ls = ld_u32(L.pls[j2]), pj_src = ld_u32(L.psrc[j2]), pj_le = ld_u32(L.pls[j2 + 1]);
            const uint32_t keep_head = ms > pi_ls ? 1u : 0u, has_repl = rl ? 1u : 0u, has_tail = me < pj_le ? 1u : 0u;
            const int32_t s = (int32_t)(keep_head + has_repl + has_tail) - (int32_t)(j2 - i + 1u);
            if ((int64_t)np + s + 1 > (int64_t)kLdsPc) { redo = true; break; }
            wave_lds_fence();
            if (s != 0 || delta != 0) {                                   // entries j2 + 1 .. np (the sentinel: the new length) move by s and shift with the text; both arrays per trip
                const uint32_t from = j2 + 1u, to = np + 1u;
                const uint32_t nb = (to - from + kWave - 1) / kWave;
                for (uint32_t b = 0; b < nb; b++) {
                    const uint32_t blk = s > 0 ? nb - 1 - b : b;
                    const uint32_t k = from + blk * kWave + (uint32_t)lane;
                    uint32_t x = 0, y = 0;
                    if (k < to) { x = L.psrc[k]; y = L.pls[k]; }
                    wave_lds_fence();
                    AM_BOUNDS(k >= to || ((int32_t)k + s >= 0 && (uint32_t)((int32_t)k + s) < kLdsPc + 2u));
                    if (k < to) { L.psrc[(uint32_t)((int32_t)k + s)] = x; L.pls[(uint32_t)((int32_t)k + s)] = y + (uint32_t)delta; }
                    wave_lds_fence();
                }
                if (s < 0 && (uint32_t)lane < (uint32_t)(-s)) L.pls[np + 1u - (uint32_t)(-s) + (uint32_t)lane] = kNoPos;      // the slots the list shrank out of (|s| <= pieces of one match <= 64 here, else ...)
                if (s < -(int32_t)kWave) { for (uint32_t k = np + 1u - (uint32_t)(-s) + (uint32_t)lane; k <= np; k += kWave) L.pls[k] = kNoPos; }
            }
            if (lane == 0) {
                uint32_t at = i + keep_head;
                AM_BOUNDS(at + (has_repl ? 1u : 0u) + (has_tail ? 1u : 0u) <= kLdsPc + 2u);
                if (has_repl) { L.psrc[at] = kReplBit | repl_off; L.pls[at] = ms; at++; }
                if (has_tail) { L.psrc[at] = pj_src + (me - pj_ls); L.pls[at] = ms + rl; }
            }
            np = (uint32_t)((int32_t)np + s);
            wave_lds_fence();
            tick(4);
            if (last_pass) { curlen = newlen; continue; }

            // (c) gather the window's bytes through the piece list, scan its own positions
            uint32_t nf = 0;
            if (wlen) {
                {
                    // the six entries from the piece that holds the window's first byte, in scalar registers (one batch of broadcast reads): a window usually
                    // lies in three or four pieces (text, replacement, text); a lane that needs a later one walks the list
                    uint32_t ls[6], sr[6];
#pragma unroll
                    for (int q = 0; q < 6; q++) { const uint32_t k = first + (uint32_t)q <= np ? first + (uint32_t)q : np; ls[q] = L.pls[k]; sr[q] = L.psrc[k]; }
#pragma unroll
                    for (int q = 0; q < 6; q++) { ls[q] = ld_u32(ls[q]); sr[q] = ld_u32(sr[q]); }
                    const bool all_here = first + 5u >= np || ls[5] >= ws + wlen;      // the window ends inside the five pieces
                    for (uint32_t x = (uint32_t)lane; x < wlen; x += kWave) {
                        const uint32_t p = ws + x;
                        uint32_t pls = ls[0], psr = sr[0];
                        if (p >= ls[1] && first + 1u < np) { pls = ls[1]; psr = sr[1]; }
                        if (p >= ls[2] && first + 2u < np) { pls = ls[2]; psr = sr[2]; }
                        if (p >= ls[3] && first + 3u < np) { pls = ls[3]; psr = sr[3]; }
                        if (p >= ls[4] && first + 4u < np) { pls = ls[4]; psr = sr[4]; }
                        if (!all_here && p >= ls[5]) {
                            uint32_t idx = first + 5u;
                            while (idx + 1u < np && L.pls[idx + 1u] <= p) idx++;
                            pls = L.pls[idx]; psr = L.psrc[idx];
                        }
                        const uint8_t* base = (psr & kReplBit) ? a.t.repl + (psr & ~kReplBit) : htext + psr;
                        AM_BOUNDS(x < kLdsWin + 16u);
                        L.win[x] = base[p - pls];
                    }
                }
                wave_lds_fence();
                tick(5);
                scanned += wlen;
                for (uint32_t base = own_lo; base < wlen && !redo; base += kWave) {
                    const uint32_t g = base + (uint32_t)lane;
                    bool found = false; uint32_t state = 0, vlen = 0;
                    if (g < wlen) {
                        // the probe decides exactly whether a needle of >= 4 bytes may end here (its bucket loads go out for every position of the
                        // window at once: one trip); the Bloom filter is a trip of its own and only pays when 1-3-byte needles make every position defer
                        bool look = true;
                        if (a.s.tiers & 7u) {
                            uint32_t w, w2;
                            load_suffix8(L.win, g, w, w2);
                            if (IC) w = fold_dword(w);
                            look = sf_filter_window(a.s.bloom, a.s.bloom_log2_words, a.s.tiers, w);
                        }
                        if (look) found = sf_verify<IC>(a.s, L.win, g, (uint64_t)g + 1, state, vlen);
                    }
                    // a record enters the list with its priority and payload; one that can never be chosen again does not enter at all
                    RpStateOne one{0, kRpWalkList};
                    if (found) {
                        one = a.t.one[state];
                        if (one.payload != kRpWalkList && (int64_t)one.priority >= best) found = false;
                    }
                    const uint64_t fm = __ballot(found);
                    const uint32_t nfb = (uint32_t)__popcll(fm);
                    if (nfb) {
                        if (nf + nfb > (uint32_t)kWave || nr + nf + nfb > kLdsRec) { redo = true; break; }
                        if (found) {                                     // staged behind the list; they move into the dead slots below
                            const uint32_t at = nr + nf + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull));
                            AM_BOUNDS(at < kLdsRec);
                            L.end[at] = ws + g + 1u;
                            L.prio[at] = one.payload != kRpWalkList ? one.priority : (int32_t)state;
                            set_pl(at, one.payload);
                        }
                        has_walk = has_walk || __ballot(found && one.payload == kRpWalkList) != 0ull;
                        nf += nfb;
                    }
                }
                if (redo) break;
                wave_lds_fence();                                         // (the window is rewritten by the next kept match's)
            }
            tick(6);

            // (d) the window's records (rarely any) go where the dead ones lie: [c_before, c_before + nf); more of them than slots: the rest of the list moves up
            if (nf) {
                const uint32_t room = c_gone - c_before, staged = nr;
                uint32_t s_end = 0, s_pl = 0; int32_t s_prio = 0;
                if ((uint32_t)lane < nf) { s_end = L.end[staged + lane]; s_prio = L.prio[staged + lane]; s_pl = PLI ? 0u : L.pl[staged + lane]; }
                wave_lds_fence();
                if ((uint32_t)lane < nf) { L.end[staged + lane] = kNoPos; L.prio[staged + lane] = kDead; set_pl(staged + lane, 0u); }      // (unused slots again, unless the list grows into them)
                wave_lds_fence();
                if (nf > room) {
                    const int32_t g = (int32_t)(nf - room);
                    if ((int64_t)nr + g > (int64_t)kLdsRec) { redo = true; break; }
                    ld_move(L.end, c_gone, nr, g, 0u, lane);
                    ld_move(reinterpret_cast<uint32_t*>(L.prio), c_gone, nr, g, 0u, lane);
                    if (!PLI) ld_move(L.pl, c_gone, nr, g, 0u, lane);
                    nr = (uint32_t)((int32_t)nr + g);
                }
                AM_BOUNDS(c_before + nf <= kLdsRec && staged + nf <= kLdsRec);
                if ((uint32_t)lane < nf) { L.end[c_before + lane] = s_end; L.prio[c_before + lane] = s_prio; set_pl(c_before + lane, s_pl); }
                wave_lds_fence();
            }
            curlen = newlen;
            tick(7);
        }
        if (redo) break;
        if (status == kRpFinished) break;
        threshold = best;
    }

    if (redo) {
        if (lane == 0) a.redo[h] = 1u;                                    // k_rp_loop, launched behind this kernel, runs this haystack from its first scan
        return;
    }
    // the final piece list, in the form k_pt_materialise reads (the haystack's own region of the piece buffer)
    if ((uint64_t)np + 1u > cap_p) { if (lane == 0) a.redo[h] = 1u; return; }
    RpPiece* const P = a.pc_buf + pb;
    for (uint32_t k = (uint32_t)lane; k <= np; k += kWave) {
        const uint32_t s = L.psrc[k];
        RpPiece e; e.lstart = k < np ? L.pls[k] : curlen;
        e.src = k < np ? ((s & kReplBit) ? (kPieceRepl | (uint64_t)(s & ~kReplBit)) : hoff + s) : 0;
        P[k] = e;
    }
    if (lane == 0) {
        RpLoopOut o;
        o.len = status == kRpNothing ? 0 : curlen; o.pieces_at = pb; o.n_pieces = np; o.status = status; o.passes = passes; o.pad = 0;
        a.out[h] = o;
        atomicMax(a.ctrl + 1, passes);
        if (scanned) atomicAdd(reinterpret_cast<unsigned long long*>(a.ctrl + 2), (unsigned long long)scanned);
        atomicAdd(a.ctrl + 7, 1u);                                        // haystacks finished here (the rest: k_rp_loop)
        if (DBG) {
            for (int i = 0; i < 8; i++) atomicAdd(reinterpret_cast<unsigned long long*>(a.ctrl + 8) + i, (unsigned long long)t_ph[i]);
            atomicAdd(reinterpret_cast<unsigned long long*>(a.ctrl + 8) + 8, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin));
            atomicAdd(reinterpret_cast<unsigned long long*>(a.ctrl + 8) + 9, (unsigned long long)passes);
        }
    }
}

}  // namespace

template <bool IC, bool DBG = false, bool PLI = false>
__global__ void __launch_bounds__(64, PLI ? 5 : 4) k_rp_lds(RpLoop a)
{
    __shared__ LpLds<PLI> L;
    const uint32_t h = a.h_first + blockIdx.x;
    ld_run_haystack<IC, DBG, PLI>(a, L, h, (int)(threadIdx.x & (kWave - 1)));
}

hipError_t launch_rp_lds(bool ic, const RpLoop& a, uint32_t n, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const bool pli = a.pl_implicit != 0 && !cfg::on(cfg::kRpNoPli);
    if (a.pad) { if (ic) hipLaunchKernelGGL((k_rp_lds<true, true>), dim3(n), dim3(64), 0, st, a); else hipLaunchKernelGGL((k_rp_lds<false, true>), dim3(n), dim3(64), 0, st, a); }      // per-phase cycle sums (AM_RP_TRACE >= 3)
    else if (ic) { if (pli) hipLaunchKernelGGL((k_rp_lds<true, false, true>), dim3(n), dim3(64), 0, st, a); else hipLaunchKernelGGL((k_rp_lds<true>), dim3(n), dim3(64), 0, st, a); }
    else { if (pli) hipLaunchKernelGGL((k_rp_lds<false, false, true>), dim3(n), dim3(64), 0, st, a); else hipLaunchKernelGGL((k_rp_lds<false>), dim3(n), dim3(64), 0, st, a); }
    return hipGetLastError();
}

}  // namespace dev
}  // namespace am
