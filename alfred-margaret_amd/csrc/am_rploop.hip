// am_rploop.hip -- Replacer.run with ALL passes of a haystack inside one kernel (reference: src/Data/Text/AhoCorasick/Replacer.hs:203-274).
//
// The pass loop of `runWithLimit.go` is per haystack: nothing of one haystack's passes depends on another haystack.  The piece-table path
// (am_replace.hip, replacer_run_pt) nevertheless runs a pass as ~16 dependent launches over ALL active haystacks plus one host look at the
// totals -- on BASELINE config 5 (16 384 haystacks, 159 passes, one replacement per haystack and pass) 159 x 240 us of latency-bound
// kernels.  Here a wavefront takes one haystack and runs its loop to the end:
//   fold      prependMatch / makeMatch / removeOverlap over the haystack's sorted records (as k_rp_pass); lists of up to 256 records are read once per pass
//   pieces    the next piece list (as k_pt_build; with one kept match every piece knows what it becomes from that match alone: one sweep)
//   windows   per kept match: gather the window around the replacement through the new piece list into the haystack's scratch (a small window through six
//             piece entries held in scalar registers), test its own positions against the Bloom filter (read from L2: a window is tens of positions) and
//             verify the survivors exactly (sf_verify: the same probe + resolve k_sf runs), and merge: old records + the window's records -> next list
//             (one kept match: in place -- the records before it stay, those behind move by the difference; several: rebuilt in the second buffer)
// Every haystack owns fixed regions (two record lists, two piece lists, a kept list, a window scratch) sized from its first scan; a
// haystack that outgrows them raises the overflow flag and the host runs the batch through the piece-table path instead.
// Replacers on the suffix-filter route, both case modes (IgnoreCase: makeMatch walks the text backwards through the piece list).
#include <hip/hip_runtime.h>

#include "am_device.h"
#include "am_wave.h"

namespace am {
namespace dev {

namespace {

constexpr int kWave = 64;
constexpr int kFold = 4;                                  // records per lane the fold keeps in registers (lists of up to 256 records are read once per pass)

__device__ __forceinline__ int64_t lp_wave_max_i64(int64_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int64_t o = __shfl_xor(v, d, kWave); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ int64_t lp_wave_sum_i64(int64_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
    return v;
}
__device__ __forceinline__ int64_t lp_wave_incl_i64(int64_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) { const int64_t o = __shfl_up(v, d, kWave); if (lane >= d) v += o; }
    return v;
}

__device__ __forceinline__ uint32_t lp_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }      // a value that is the same in every lane: scalar from here on

// what the lanes of this wavefront wrote to global memory is visible to its other lanes (one L1 per workgroup: a wait, no cache control)
__device__ __forceinline__ void lp_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// number of elements of the sorted key(0..n) that are <= x: a 64-ary search, one trip per factor of 64 (all arguments uniform)
template <class Key>
__device__ __forceinline__ uint64_t lp_count_le(Key key, uint64_t n, uint64_t x, int lane, uint64_t deadline)
{
    uint64_t lo = 0, hi = n;
    while (hi > lo) {
        if (__builtin_amdgcn_s_memtime() > deadline) return lo;       // (watchdog: the caller gives up at its next step)
        const uint64_t m = hi - lo, step = (m + 63) / 64;
        const bool in = (uint64_t)lane * step < m;                       // the lane's block is not empty
        uint64_t i = lo + ((uint64_t)lane + 1) * step - 1;                // its last element
        if (i >= hi) i = hi - 1;
        const bool le = in && key(i) <= x;                                // then the whole block is <= x
        const uint64_t c = (uint64_t)__popcll(__ballot(le));              // such blocks are a prefix
        if (step == 1) return uniform_u64(lo + c);
        const uint64_t nlo = uniform_u64(lo + c * step);
        if (nlo >= hi) return hi;
        lo = nlo; hi = nlo + step < hi ? nlo + step : hi;               // the first element > x lies in block c
    }
    return lo;
}

// bytes [lo, lo + len) of a piece list (P[n] = sentinel) into dst
__device__ __forceinline__ void lp_gather(const RpPiece* __restrict__ P, uint32_t n, const uint8_t* __restrict__ text, const uint8_t* __restrict__ repl,
                                          uint64_t lo, uint64_t len, uint8_t* __restrict__ dst, int lane, uint64_t deadline)
{
    if (len == 0) return;
    const uint64_t cnt = lp_count_le([&](uint64_t i) { return P[i].lstart; }, n, lo, lane, deadline);      // pieces that start at or before lo (>= 1: P[0] starts at 0)
    uint64_t pos = lo;
    const uint64_t end = lo + len;
    for (uint64_t i = cnt - 1; pos < end && i < n; i++) {
        const uint64_t ps = P[i].lstart, pn = P[i + 1].lstart, pe = pn < end ? pn : end, s = P[i].src;
        if (pe <= pos) continue;                                         // (an empty piece)
        if (__builtin_amdgcn_s_memtime() > deadline) return;
        const uint8_t* from = ((s & kPieceRepl) ? repl + (s & ~kPieceRepl) : text + s) + (pos - ps);
        uint8_t* to = dst + (pos - lo);
        for (uint64_t x = lane; x < pe - pos; x += kWave) to[x] = from[x];
        pos = pe;
    }
}

// Utf8.hs:256-276 skipCodePointsBackwards on the text a piece list describes (makeMatch of an IgnoreCase replacer, Replacer.hs:268-274): from
// byte `index` back over n code points, to the first byte of the code point reached; never leaves the text.  One lane, its own cursor into the
// list (the bytes of a match almost always lie in one piece).
__device__ __forceinline__ uint64_t lp_skip_code_points_backwards(const RpPiece* __restrict__ P, uint32_t np, const uint8_t* __restrict__ text, const uint8_t* __restrict__ repl,
                                                               uint64_t index, uint64_t n)
{
    uint32_t lo = 0, hi = np;                                // last piece that starts at or before index (np >= 1 here: the text holds a match)
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (P[mid].lstart <= index) lo = mid; else hi = mid; }
    uint32_t pi = lo;
    uint64_t ls = P[pi].lstart, src = P[pi].src;
    auto byte_at = [&](uint64_t pos) -> uint32_t {
        while (pos < ls) { pi--; ls = P[pi].lstart; src = P[pi].src; }      // (pieces may be empty: a loop)
        const uint8_t* from = (src & kPieceRepl) ? repl + (src & ~kPieceRepl) : text + src;
        return from[pos - ls];
    };
    int64_t i = (int64_t)index;
    for (;;) {
        while (i > 0 && (byte_at((uint64_t)i) & 0xC0u) == 0x80u) i--;      // atTrailingByte
        if (n == 0 || i <= 0) return (uint64_t)(i < 0 ? 0 : i);
        i--; n--;
    }
}

// how many of the sorted key(0 .. m) are <= x1 / <= x2 (x1 <= x2): lists of up to 1024 keys are looked at 256 per trip (four per lane in flight), stopping
// behind x2 -- one or two trips instead of two 64-ary searches of two trips each
template <class Key>
__device__ __forceinline__ void lp_count2_le(Key key, uint64_t m, uint64_t x1, uint64_t x2, uint64_t& c1, uint64_t& c2, int lane, uint64_t deadline)
{
    if (m > 16u * kWave) {
        c1 = lp_count_le(key, m, x1, lane, deadline);
        c2 = c1 + lp_count_le([&](uint64_t i) { return key(c1 + i); }, m - c1, x2, lane, deadline);
        return;
    }
    uint64_t a1 = 0, a2 = 0;
    for (uint64_t g = 0; g < m; g += 4u * kWave) {
        uint64_t e[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint64_t i = g + (uint64_t)u * kWave + lane; e[u] = ~0ull; if (i < m) e[u] = key(i); }
        uint64_t in2 = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) { a1 += (uint64_t)__popcll(__ballot(e[u] <= x1)); in2 += (uint64_t)__popcll(__ballot(e[u] <= x2)); }
        a2 += in2;
        if (in2 < 4u * kWave) break;
    }
    c1 = a1; c2 = a2;
}

// elements [from, to) of A go to [from + g, to + g), each through fix(): 128 per trip, from the end backwards when they move up (what a store overwrites
// has been loaded: within a trip all loads precede the stores, across trips the stores land behind everything still to be read), forwards otherwise
template <class T, class F>
__device__ __forceinline__ void lp_move(T* A, uint64_t from, uint64_t to, int64_t g, F fix, int lane)
{
    if (to <= from) return;
    const uint64_t nb = (to - from + 2 * kWave - 1) / (2 * kWave);
    for (uint64_t b = 0; b < nb; b++) {
        const uint64_t blk = g > 0 ? nb - 1 - b : b;
        const uint64_t i0 = from + blk * 2 * kWave + (uint64_t)lane, i1 = i0 + kWave;
        const bool v0 = i0 < to, v1 = i1 < to;
        T x0{}, x1{};
        if (v0) x0 = A[i0];
        if (v1) x1 = A[i1];
        if (v0) { fix(x0); A[(uint64_t)((int64_t)i0 + g)] = x0; }
        if (v1) { fix(x1); A[(uint64_t)((int64_t)i1 + g)] = x1; }
    }
}

// Watchdog: the loops of a haystack's run look at the clock; a run that lasts longer than kLpMaxTicks (a corrupt table, a bug) gives up with
// the overflow flag and the number of the loop in ctrl[5], and the host takes the pass-by-pass loop: the kernel cannot hang.
constexpr uint64_t kLpMaxTicks = 4000000000ull;           // s_memtime ticks at about the shader clock here (~2 GHz, measured in round 2): ~2 s for ONE haystack
#define LP_STEP(code) do { if (__builtin_amdgcn_s_memtime() > deadline) { if (lane == 0) atomicMax(a.ctrl + 5, (uint32_t)(code)); overflow = true; } } while (0)

// DBG (AM_RP_TRACE >= 3): s_memtime per phase of every pass, summed over all haystacks into ctrl[8..23] (64-bit: fold 1, fold 2, pieces, record copies + searches,
// gather, window scan, the whole run, passes)
template <bool IC, bool DBG>
__device__ void lp_run_haystack(const RpLoop& a, const uint32_t h, const int lane)
{
    const uint64_t deadline = __builtin_amdgcn_s_memtime() + kLpMaxTicks;
    uint64_t t_mark = DBG ? __builtin_amdgcn_s_memtime() : 0, t_ph[7] = {0, 0, 0, 0, 0, 0, 0};
    const uint64_t t_begin = t_mark;
    auto tick = [&](int ph) { if (DBG) { const uint64_t now = __builtin_amdgcn_s_memtime(); t_ph[ph] += now - t_mark; t_mark = now; } };
    // (everything every lane agrees on goes through readfirstlane: the compiler then keeps it in scalar registers and branches on it with
    // scalar branches -- the loops below are uniform by construction, and it cannot know that)
    const uint64_t hoff = uniform_u64(a.offsets[h]);
    uint64_t curlen = uniform_u64(a.offsets[h + 1]) - hoff;
    const uint64_t rb = uniform_u64(a.rec_base[h]), cap_r = (uniform_u64(a.rec_base[h + 1]) - rb) >> 1;
    const uint64_t pb = uniform_u64(a.pc_base[h]), cap_p = (uniform_u64(a.pc_base[h + 1]) - pb) >> 1;
    const uint64_t rf0 = uniform_u64(a.rec_first0[h]);
    const Record* R = a.recs0 + rf0;                                     // pass 1 reads the first scan's records where they lie
    uint64_t nr = uniform_u64(a.rec_first0[h + 1]) - rf0;
    Record* const Rbuf0 = a.rec_buf + rb; Record* const Rbuf1 = Rbuf0 + cap_r;
    RpPiece* P = a.pc_buf + pb; RpPiece* Q = P + cap_p;
    RpKept* const K = a.kept_buf + (rb >> 1);
    uint8_t* const wt = a.wtext + (uint64_t)h * a.wcap;
    if (lane == 0) { P[0] = RpPiece{hoff, 0}; P[1] = RpPiece{0, curlen}; }
    uint32_t np = 1;
    int64_t threshold = 1;                                               // initialThreshold (Replacer.hs:211)
    uint32_t passes = 0, status = kRpFinished;
    uint64_t scanned = 0;
    bool overflow = nr > cap_r || cap_p < 4;
    lp_sync();

    while (!overflow) {
        passes++;
        LP_STEP(1);
        // ---- prependMatch, first half: the best priority below the threshold (Replacer.hs:255-258).
        // The first kFold x 64 records are taken kFold per lane with all their loads in flight together (record -> its state's entry: two trips for
        // the lot instead of two per 64) and stay in registers for the second half; longer lists go on 64 at a time.
        int64_t best = INT64_MIN;
        uint64_t c_end[kFold]; int32_t c_prio[kFold]; uint32_t c_st[kFold], c_pl[kFold]; bool c_valid[kFold];
#pragma unroll
        for (int u = 0; u < kFold; u++) {
            const uint64_t r = (uint64_t)u * kWave + lane;
            c_valid[u] = r < nr;
            Record rec{0, 0, 0};
            if (c_valid[u]) rec = R[r];
            c_st[u] = rec.state; c_end[u] = rec.end_pos;
        }
#pragma unroll
        for (int u = 0; u < kFold; u++) {
            RpStateOne one{0, kRpWalkList};
            if (c_valid[u]) one = a.t.one[c_st[u]];
            c_prio[u] = one.priority; c_pl[u] = one.payload;
        }
        auto best_of_list = [&](uint32_t st) {
            for (uint64_t k = a.t.vals_off[st], ke = a.t.vals_off[st + 1]; k < ke; k++) {
                const int64_t p = a.t.payloads[a.t.vals[k]].priority;
                if (p < threshold && p > best) best = p;
            }
        };
#pragma unroll
        for (int u = 0; u < kFold; u++) {
            if (c_valid[u]) {
                if (c_pl[u] != kRpWalkList) { if ((int64_t)c_prio[u] < threshold && (int64_t)c_prio[u] > best) best = c_prio[u]; }
                else best_of_list(c_st[u]);
            }
        }
        for (uint64_t r = (uint64_t)kFold * kWave + lane; r < nr; r += kWave) {
            const uint32_t st = R[r].state;
            const RpStateOne one = a.t.one[st];
            if (one.payload != kRpWalkList) { if ((int64_t)one.priority < threshold && (int64_t)one.priority > best) best = one.priority; }
            else best_of_list(st);
        }
        best = (int64_t)uniform_u64((uint64_t)lp_wave_max_i64(best));
        tick(0);
        if (best == INT64_MIN) { status = kRpFinished; break; }           // no match below the threshold: the text stays (:228-230)

        // ---- second half: the matches that carry it, makeMatch (:264-267), removeOverlap (:191-198), 64 records at a time in position order
        int64_t delta_all = 0, delta_kept = 0;
        uint64_t last_end = 0;
        uint32_t nkept = 0, payload = 0;
        // the value of the state's list that carries `best` (a state with several values)
        auto pick_of_list = [&](uint32_t st, bool& sel, uint32_t& pl) {
            for (uint64_t k = a.t.vals_off[st], ke = a.t.vals_off[st + 1]; k < ke; k++) {
                const uint32_t v = a.t.vals[k];
                if (a.t.payloads[v].priority == best) { sel = true; pl = v; }
            }
        };
        // one block of 64 records: sel = carries the best priority (then pl, len, cps, rl are its payload's)
        auto block = [&](bool sel, uint32_t pl, uint64_t end_pos) {
            uint64_t len = 0; uint32_t cps = 0, rl = 0;
            if (sel) { const RpPayload pp = a.t.payloads[pl]; len = pp.len_bytes; cps = pp.len_code_points; rl = pp.repl_len; }      // (the few records that carry the best priority)
            uint64_t start = end_pos - len;                              // makeMatch, CaseSensitive (Replacer.hs:266-267)
            if (IC && sel) {                                             // IgnoreCase (:268-274): the match is as long as its code points are in the haystack
                start = cps == 0 ? end_pos : lp_skip_code_points_backwards(P, np, a.text, a.t.repl, end_pos - 1, cps - 1);
                len = end_pos - start;
            }
            const int64_t delta = sel ? (int64_t)rl - (int64_t)len : 0;
            delta_all += delta;
            uint64_t pending = __ballot(sel);
            bool keep = false;
            while (pending) {
                const uint64_t ok = __ballot(sel && start >= last_end) & pending;
                if (!ok) break;
                const int l = __ffsll((unsigned long long)ok) - 1;
                if (lane == l) keep = true;
                last_end = uniform_u64(__shfl(start + len, l, kWave));
                pending &= l == 63 ? 0ull : ~((2ull << l) - 1ull);
            }
            const uint64_t keepmask = __ballot(keep);
            if (keepmask) {
                const int64_t kd = keep ? delta : 0;
                const int64_t incl = lp_wave_incl_i64(kd, lane);
                if (keep) {
                    const uint32_t rank = __popcll(keepmask & ((1ull << lane) - 1ull));
                    RpKept e; e.src_start = start; e.src_len = len; e.dst = (uint64_t)((int64_t)start + delta_kept + (incl - kd));
                    K[nkept + rank] = e;
                    payload = pl;
                }
                delta_kept += (int64_t)uniform_u64((uint64_t)__shfl(incl, kWave - 1, kWave));
                nkept += __popcll(keepmask);
            }
        };
#pragma unroll
        for (int u = 0; u < kFold; u++) {
            if ((uint64_t)u * kWave < nr) {                              // (uniform)
                bool sel = false; uint32_t pl = 0;
                if (c_valid[u]) {
                    if (c_pl[u] != kRpWalkList) { if ((int64_t)c_prio[u] == best) { sel = true; pl = c_pl[u]; } }
                    else pick_of_list(c_st[u], sel, pl);
                }
                block(sel, pl, c_end[u]);
            }
        }
        for (uint64_t base = (uint64_t)kFold * kWave; base < nr && !overflow; base += kWave) {
            const uint64_t r = base + lane;
            LP_STEP(3);
            bool sel = false; uint32_t pl = 0; uint64_t end_pos = 0;
            if (r < nr) {
                const Record rec = R[r];
                end_pos = rec.end_pos;
                const RpStateOne one = a.t.one[rec.state];
                if (one.payload != kRpWalkList) { if ((int64_t)one.priority == best) { sel = true; pl = one.payload; } }
                else pick_of_list(rec.state, sel, pl);
            }
            block(sel, pl, end_pos);
        }
        delta_all = (int64_t)uniform_u64((uint64_t)lp_wave_sum_i64(delta_all));
        payload = lp_u32((uint32_t)lp_wave_max_i64((int64_t)payload));    // uniform: every kept match has the same payload
        const int64_t newlen_all = (int64_t)curlen + delta_all;           // replacementLength over ALL matches (:240)
        if (newlen_all > 0 && (uint64_t)newlen_all > a.max_len) { status = kRpNothing; break; }
        const uint64_t newlen = (uint64_t)((int64_t)curlen + delta_kept);
        status = best == a.t.min_priority ? kRpFinished : kRpActive;    // :241-242
        lp_sync();                                                       // K is read by every lane from here on
        tick(1);

        // ---- replace (:163-180) on the piece list: P -> Q (the scheme of k_pt_build; the two lists of a haystack take turns)
        RpPayload pp = a.t.payloads[payload];
        pp.repl_off = uniform_u64(pp.repl_off); pp.repl_len = lp_u32(pp.repl_len);
        const uint64_t repl_len = nkept ? pp.repl_len : 0;
        if ((uint64_t)np + 2ull * nkept + 2ull > cap_p) { overflow = true; break; }
        uint32_t nq = 0, near = 0;                                       // near: index (new list) of the first entry of the piece the match starts in: where the window's gather begins to look
        if (nkept == 1) {
            // ONE kept match (a pass of a haystack usually makes one replacement): every piece knows what it becomes from K[0] alone -- untouched
            // before the match, moved behind it, cut where it overlaps (at most: head, replacement, tail) -- one sweep, no look at K per piece
            // (matches are not empty here: automata with the empty needle do not take this kernel)
            RpKept k = K[0];
            k.src_start = uniform_u64(k.src_start); k.src_len = uniform_u64(k.src_len); k.dst = uniform_u64(k.dst);
            const uint64_t ms = k.src_start, me = k.src_start + k.src_len;
            const int64_t shift = (int64_t)(k.dst + repl_len) - (int64_t)me;      // new position - old position behind the match
            for (uint32_t r0 = 0; r0 < np && !overflow; r0 += kWave) {
                LP_STEP(5);
                const uint32_t i = r0 + lane;
                RpPiece e[3]; uint32_t c = 0;
                bool has_ms = false;
                if (i < np) {
                    const RpPiece pc = P[i];
                    const uint64_t ls = pc.lstart, le = P[i + 1].lstart;
                    has_ms = ls <= ms && ms < le;
                    if (le > ls) {
                        if (le <= ms) { e[c++] = RpPiece{pc.src, ls}; }                       // ends at or before the match
                        else if (ls >= me) { e[c++] = RpPiece{pc.src, (uint64_t)((int64_t)ls + shift)}; }      // starts behind it
                        else {
                            if (ms > ls) e[c++] = RpPiece{pc.src, ls};                                               // head
                            if (ms >= ls && repl_len) e[c++] = RpPiece{kPieceRepl | pp.repl_off, k.dst};   // the match starts in this piece: its replacement
                            if (me < le) e[c++] = RpPiece{pc.src + (me - ls), (uint64_t)((int64_t)me + shift)};      // tail
                        }
                    }
                }
                const uint32_t incl = wave_inclusive_sum(c, (uint32_t)lane);
                uint32_t at = nq + incl - c;
                if (c > 0) Q[at] = e[0];
                if (c > 1) Q[at + 1] = e[1];
                if (c > 2) Q[at + 2] = e[2];
                const uint64_t hm = __ballot(has_ms);
                if (hm) near = lp_u32(__shfl(at, __ffsll((unsigned long long)hm) - 1, kWave));
                nq += lp_u32(__shfl(incl, kWave - 1, kWave));
            }
            if (lane == 0) Q[nq] = RpPiece{0, newlen};                    // sentinel
        } else {
            const uint32_t nk = nkept;
            auto span = [&](uint64_t ls, uint64_t le, uint32_t& ja, uint32_t& jb) {
                uint32_t lo = 0, hi = nk;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (K[mid].src_start + K[mid].src_len <= ls) lo = mid + 1; else hi = mid; }
                ja = lo; jb = lo;
                while (jb < nk && K[jb].src_start < le) jb++;
            };
            auto emit = [&](uint32_t i, auto&& f) {
                const uint64_t ls = P[i].lstart, le = P[i + 1].lstart;
                if (le == ls) return;
                uint32_t ja, jb; span(ls, le, ja, jb);
                uint64_t x = ls;
                auto new_pos = [&](uint64_t xx, uint32_t jprev_plus1) -> uint64_t {
                    if (jprev_plus1 == 0) return xx;
                    const RpKept k = K[jprev_plus1 - 1];
                    return k.dst + repl_len + (xx - (k.src_start + k.src_len));
                };
                uint32_t jp = ja;
                for (uint32_t j = ja; j < jb; j++) {
                    const RpKept k = K[j];
                    if (k.src_start > x) f(P[i].src + (x - ls), new_pos(x, jp));
                    if (k.src_start >= ls && repl_len) f(kPieceRepl | pp.repl_off, k.dst);
                    x = k.src_start + k.src_len;
                    jp = j + 1;
                    if (x >= le) break;
                }
                if (x < le) f(P[i].src + (x - ls), new_pos(x, jp));
            };
            for (uint32_t r0 = 0; r0 < np && !overflow; r0 += kWave) {
                LP_STEP(5);
                const uint32_t i = r0 + lane;
                uint32_t c = 0;
                if (i < np) emit(i, [&](uint64_t, uint64_t) { c++; });
                const uint32_t incl = wave_inclusive_sum(c, (uint32_t)lane);
                uint32_t at = nq + incl - c;
                if (i < np) emit(i, [&](uint64_t src, uint64_t pos) { Q[at++] = RpPiece{src, pos}; });
                nq += lp_u32(__shfl(incl, kWave - 1, kWave));
            }
            if (lane == 0) Q[nq] = RpPiece{0, newlen};                    // sentinel
        }
        { RpPiece* t = P; P = Q; Q = t; }
        np = nq;
        lp_sync();
        tick(2);
        if (status == kRpFinished) { curlen = newlen; break; }

        // ---- the next pass's records: old records outside the neighbourhood of the replacements, shifted, + the records of the windows
        // (k_rp_win_meta + k_pt_win_copy + the window scan + k_rp_merge of the piece-table path, one window at a time)
        // A pass with ONE kept match on a list that lives in the haystack's own buffer is done IN PLACE: the records before the match stay where they are,
        // the window's records are collected in the other buffer, the records behind the replaced region move by the difference (and shift with the text),
        // the window's records go into the gap.  (The PMC pass of round 4: rewriting the whole list in every pass was 2 TB/s of traffic.)
        Record* const Rn = R == Rbuf0 ? Rbuf1 : Rbuf0;                   // the other buffer
        const bool inplace = nkept == 1 && (R == Rbuf0 || R == Rbuf1);
        uint64_t cursor = 0, at = 0, e_first = 0;
        auto end_of = [&](uint64_t i) { return R[i].end_pos; };
        auto copy_shifted = [&](uint64_t from, uint64_t to, int64_t shift) {
            for (uint64_t base = from; base < to; base += 2 * kWave) {
                const uint64_t i0 = base + lane, i1 = i0 + kWave;
                Record r0{0, 0, 0}, r1{0, 0, 0};
                if (i0 < to) r0 = R[i0];
                if (i1 < to) r1 = R[i1];
                if (i0 < to) { r0.end_pos = (uint64_t)((int64_t)r0.end_pos + shift); r0.haystack = h; Rn[cursor + (i0 - from)] = r0; }
                if (i1 < to) { r1.end_pos = (uint64_t)((int64_t)r1.end_pos + shift); r1.haystack = h; Rn[cursor + (i1 - from)] = r1; }
            }
        };
        auto count2_le = [&](uint64_t from, uint64_t x1, uint64_t x2, uint64_t& c1, uint64_t& c2) {      // records of R[from .. nr) that end at or before x1 / x2
            lp_count2_le([&](uint64_t i) { return end_of(from + i); }, nr - from, x1, x2, c1, c2, lane, deadline);
        };
        // A window of up to 128 bytes that lies in at most four pieces next to `near` (the usual case: pieces are hundreds of bytes, a window tens): the six
        // entries around `near` in one trip (one per lane, read into scalar registers), every lane's bytes in a second one -- instead of a search through
        // the list and two trips per piece.  false: not such a window, the general gather does it.
        auto gather_small = [&](uint64_t ws, uint32_t wlen) -> bool {
            if (wlen > 2u * kWave) return false;
            const uint32_t first = near > 0 ? near - 1u : 0u;
            RpPiece pc{0, ~0ull};
            if (lane < 6) { const uint32_t idx = first + (uint32_t)lane; pc = P[idx < np ? idx : np]; }      // (P[np]: the sentinel, starts where the text ends)
            uint64_t ls[6], sr[6];
#pragma unroll
            for (int q = 0; q < 6; q++) { ls[q] = uniform_u64(__shfl(pc.lstart, q, kWave)); sr[q] = uniform_u64(__shfl(pc.src, q, kWave)); }
            // the piece the window starts in: entry 1 (= near) if it starts at or before ws, else entry 0; the five entries from there, by name (an
            // array indexed with a run-time value would live in scratch memory)
            const uint64_t wend = ws + wlen;
            const bool from1 = ls[1] <= ws && first + 1u <= np;
            if (!from1 && ls[0] > ws) return false;
            const uint64_t l0 = from1 ? ls[1] : ls[0], l1 = from1 ? ls[2] : ls[1], l2 = from1 ? ls[3] : ls[2], l3 = from1 ? ls[4] : ls[3], l4 = from1 ? ls[5] : ls[4];
            const uint64_t s0 = from1 ? sr[1] : sr[0], s1 = from1 ? sr[2] : sr[1], s2 = from1 ? sr[3] : sr[2], s3 = from1 ? sr[4] : sr[3];
            if (l4 < wend && first + (from1 ? 5u : 4u) < np) return false;          // the window reaches beyond the fourth piece
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const uint32_t x = (uint32_t)lane + (uint32_t)t * kWave;
                if (x < wlen) {
                    const uint64_t pp = ws + x;
                    uint64_t pls = l0, psr = s0;
                    if (pp >= l1) { pls = l1; psr = s1; }
                    if (pp >= l2) { pls = l2; psr = s2; }
                    if (pp >= l3) { pls = l3; psr = s3; }
                    const uint8_t* from = (psr & kPieceRepl) ? a.t.repl + (psr & ~kPieceRepl) : a.text + psr;
                    wt[x] = from[pp - pls];
                }
            }
            return true;
        };
        for (uint32_t j = 0; j < nkept && !overflow; j++) {
            LP_STEP(6);
            RpKept k = K[j];
            k.src_start = uniform_u64(k.src_start); k.src_len = uniform_u64(k.src_len); k.dst = uniform_u64(k.dst);
            // old records that end at or before the replaced region: unchanged context, they move with the text
            uint64_t c_before = 0, c_gone = 0;                           // records of R[at ..) that end at or before the match's start / within reach of its end
            count2_le(at, k.src_start, k.src_start + k.src_len + a.ov, c_before, c_gone);
            const uint64_t e = at + c_before;
            if (inplace) e_first = e;
            else {
                if (cursor + (e - at) > cap_r) { overflow = true; break; }
                copy_shifted(at, e, (int64_t)k.dst - (int64_t)k.src_start);
                cursor += e - at;
            }
            // the window of match j in the new text
            tick(3);
            const uint64_t dst = k.dst;
            uint64_t hi = dst + repl_len + a.ov;
            if (hi > newlen) hi = newlen;
            if (j + 1 < nkept) { const uint64_t nd = uniform_u64(K[j + 1].dst); if (nd < hi) hi = nd; }
            const uint64_t ws = dst > a.ov ? dst - a.ov : 0;
            const uint32_t wlen = hi > dst ? (uint32_t)(hi - ws) : 0u, own_lo = (uint32_t)(dst - ws);
            if (wlen > a.wcap) { overflow = true; break; }
            if (wlen) {
                if (!(nkept == 1 && gather_small(ws, wlen))) lp_gather(P, np, a.text, a.t.repl, ws, wlen, wt, lane, deadline);
                lp_sync();
                tick(4);
                scanned += wlen;
                for (uint32_t base = own_lo; base < wlen && !overflow; base += kWave) {
                    LP_STEP(7);
                    const uint32_t g = base + (uint32_t)lane;
                    bool found = false; uint32_t state = 0, vlen = 0;
                    if (g < wlen) {
                        uint32_t w, w2;
                        load_suffix8(wt, g, w, w2);
                        if (IC) w = fold_dword(w);
                        if (sf_filter_window(a.s.bloom, a.s.bloom_log2_words, a.s.tiers, w)) found = sf_verify<IC>(a.s, wt, g, (uint64_t)g + 1, state, vlen);
                    }
                    const uint64_t fm = __ballot(found);
                    const uint32_t nf = (uint32_t)__popcll(fm);      // (a ballot is scalar already)
                    if (nf) {
                        if (cursor + nf > cap_r) { overflow = true; break; }
                        if (found) Rn[cursor + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull))] = Record{ws + g + 1, h, state};
                        cursor += nf;
                    }
                }
                if (j + 1 < nkept) lp_sync();                            // (the scratch is rewritten by the next window; after the last one the pass's closing wait covers it)
                tick(5);
            }
            // old records that touch the replaced bytes are gone
            at += c_gone;
        }
        if (overflow) break;
        if (inplace) {
            const uint64_t nf = cursor;                                  // the window's records lie in Rn[0 .. nf)
            const int64_t g = (int64_t)(e_first + nf) - (int64_t)at, delta = (int64_t)newlen - (int64_t)curlen;
            if ((int64_t)nr + g > (int64_t)cap_r) { overflow = true; break; }
            Record* const Rw = const_cast<Record*>(R);
            lp_sync();                                                   // (the lanes that found them wrote the window's records)
            if (g != 0 || delta != 0) lp_move(Rw, at, nr, g, [&](Record& r) { r.end_pos = (uint64_t)((int64_t)r.end_pos + delta); }, lane);
            for (uint64_t i = lane; i < nf; i += kWave) Rw[e_first + i] = Rn[i];
            nr = (uint64_t)((int64_t)nr + g);
        } else {
            if (cursor + (nr - at) > cap_r) { overflow = true; break; }
            copy_shifted(at, nr, (int64_t)newlen - (int64_t)curlen);
            cursor += nr - at;
            R = Rn; nr = cursor;
        }
        curlen = newlen; threshold = best;
        lp_sync();
        tick(3);
    }

    if (lane == 0) {
        RpLoopOut o;
        o.len = status == kRpNothing ? 0 : curlen; o.pieces_at = (uint64_t)(P - a.pc_buf); o.n_pieces = np; o.status = status; o.passes = passes; o.pad = 0;
        a.out[h] = o;
        if (overflow) atomicOr(a.ctrl + 0, 1u);
        atomicMax(a.ctrl + 1, passes);
        if (scanned) atomicAdd(reinterpret_cast<unsigned long long*>(a.ctrl + 2), (unsigned long long)scanned);
        if (DBG) {
            t_ph[6] = __builtin_amdgcn_s_memtime() - t_begin;
            for (int i = 0; i < 7; i++) atomicAdd(reinterpret_cast<unsigned long long*>(a.ctrl + 8) + i, (unsigned long long)t_ph[i]);
            atomicAdd(reinterpret_cast<unsigned long long*>(a.ctrl + 8) + 7, (unsigned long long)passes);
        }
    }
}

}  // namespace

// one wavefront (= one workgroup) per haystack: the hardware hands the workgroups out as wavefront slots become free.  (A loop over haystacks
// inside the kernel -- wavefronts drawing haystack numbers from a counter -- was the first version: the compiler merged that loop with the
// pass loop, the haystack number became a loop-carried value of a loop it took for divergent, and the kernel never ended; a haystack number
// that comes from blockIdx is uniform for the compiler too.)
template <bool IC, int W, bool DBG = false>
__global__ void __launch_bounds__(64, W) k_rp_loop(RpLoop a)
{
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t h = a.h_first + blockIdx.x;
    if (h >= a.n_hay) return;
    if (a.redo && a.redo[h] == 0) return;                                 // k_rp_lds (am_rplds.hip) finished this haystack out of LDS
    lp_run_haystack<IC, DBG>(a, h, lane);
}

// region sizes per haystack from its first scan: records 2 x (2 n + 64), pieces 2 x (4 n + 64)  (n = its records; element n_hay: 0)
__global__ void __launch_bounds__(256) k_rp_loop_caps(const uint64_t* __restrict__ rec_first, uint32_t n_hay, uint32_t* __restrict__ cap_r2, uint32_t* __restrict__ cap_p2,
                                                      uint32_t* __restrict__ max_records)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h > n_hay) return;
    if (h == n_hay) { cap_r2[h] = 0; cap_p2[h] = 0; return; }
    const uint64_t n = rec_first[h + 1] - rec_first[h];
    if (n > 64) atomicMax(max_records, (uint32_t)(n > 0xFFFFFFFFull ? 0xFFFFFFFFull : n));      // the longest list of the batch (one wavefront will walk it in every pass)
    cap_r2[h] = (uint32_t)(2 * (2 * n + 64));
    cap_p2[h] = (uint32_t)(2 * (4 * n + 64));
}

hipError_t launch_rp_loop_caps(const uint64_t* rec_first, uint32_t n_hay, uint32_t* cap_r2, uint32_t* cap_p2, uint32_t* max_records, hipStream_t st)
{
    hipLaunchKernelGGL(k_rp_loop_caps, dim3((n_hay + 1 + 255) / 256), dim3(256), 0, st, rec_first, n_hay, cap_r2, cap_p2, max_records);
    return hipGetLastError();
}

hipError_t launch_rp_loop(bool ic, const RpLoop& a, uint32_t n, int waves, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    // wavefronts per SIMD the register budget is cut for (AM_RP_LOOP_WAVES, A/B: the exact phase -- sf_verify -- wants ~125 registers, and cfg5 ran
    // at 68.0 GiB/s with 4 wavefronts per SIMD, 61.9 / 64.4 / 56.4 with budgets cut for 5 / 6 / 8: the spills cost more than the wavefronts bring)
    const dim3 grid(n), block(64);
    if (ic) { hipLaunchKernelGGL((k_rp_loop<true, 4>), grid, block, 0, st, a); return hipGetLastError(); }
    if (a.pad) { hipLaunchKernelGGL((k_rp_loop<false, 4, true>), grid, block, 0, st, a); return hipGetLastError(); }      // per-phase cycle sums (AM_RP_TRACE >= 3)
    switch (waves) {
    case 5: hipLaunchKernelGGL((k_rp_loop<false, 5>), grid, block, 0, st, a); break;
    case 6: hipLaunchKernelGGL((k_rp_loop<false, 6>), grid, block, 0, st, a); break;
    case 8: hipLaunchKernelGGL((k_rp_loop<false, 8>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((k_rp_loop<false, 4>), grid, block, 0, st, a); break;
    }
    return hipGetLastError();
}

}  // namespace dev
}  // namespace am
