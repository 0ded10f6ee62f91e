// am_replace.hip -- device side of one Replacer pass (reference: src/Data/Text/AhoCorasick/Replacer.hs:203-274).
//
// The scan itself is k_sf / k_ac (am_kernels.hip): it leaves one record per (haystack, end position),
// sorted.  The kernels here do what `runWithLimit.go` does with the fold result, for every active
// haystack of the batch at once, without the records ever leaving HBM:
//   k_rp_ranges   record range of every haystack (binary search on the sorted records)
//   k_rp_pass     one wavefront per haystack: prependMatch (:252-260) = best priority below the
//                 haystack's threshold and the matches that carry it; makeMatch (:264-274) = start and
//                 length of each; replacementLength (:183-187) over all of them; removeOverlap
//                 (:191-198) greedily in position order; the new length and what happens next (:228-242)
//   k_rp_route    where every haystack's next text goes (next pass / finished), thresholds for the next pass
//   k_rp_splice   replace (:163-180): copies gaps and replacements into the new text, 16 KiB tiles
// Priorities are distinct (Replacer.hs:100-104 assigns -index; compose :127-131 renumbers), so all matches
// chosen in one pass of one haystack belong to ONE payload: they have the same code-point length, hence
// their start positions are ordered like their end positions and the derived-Ord sort (:159, :241) is the
// identity on the record order.  am_replacer_create rejects payload tables with duplicate priorities.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>

#include "am_device.h"

namespace am {
namespace dev {

namespace {

constexpr int kWave = 64;

__device__ __forceinline__ int64_t wave_max_i64(int64_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int64_t o = __shfl_xor(v, d, kWave); v = o > v ? o : v; }
    return v;
}

__device__ __forceinline__ int64_t wave_sum_i64(int64_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
    return v;
}

__device__ __forceinline__ int64_t wave_inclusive_sum_i64(int64_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) { const int64_t o = __shfl_up(v, d, kWave); if (lane >= d) v += o; }
    return v;
}

// Utf8.hs:256-276 skipCodePointsBackwards, relative to the haystack start; never leaves the haystack.
__device__ __forceinline__ uint64_t skip_code_points_backwards(const uint8_t* hay, uint64_t index, uint64_t n)
{
    int64_t i = (int64_t)index;
    for (;;) {
        while (i > 0 && (hay[i] & 0xC0u) == 0x80u) i--;      // atTrailingByte
        if (n == 0 || i <= 0) return (uint64_t)(i < 0 ? 0 : i);
        i--; n--;
    }
}

}  // namespace

__global__ void __launch_bounds__(256) k_rp_ranges(const Record* __restrict__ recs, uint64_t n_rec, const uint64_t* __restrict__ n_rec_dev, uint64_t* __restrict__ rec_first,
                                                   RpRoute route, uint32_t n_act)
{
    const uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h > n_act) return;
    if (n_rec_dev) n_rec = *n_rec_dev;                 // the count is still on the device (no host round trip between passes)
    if (h == n_act && route.len_next) { route.len_next[h] = 0; route.len_fin[h] = 0; route.tiles[h] = 0; route.act[h] = 0; route.fin[h] = 0; }   // the scans' trailing element
    uint64_t lo = 0, hi = n_rec;                     // first record whose haystack >= h
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (recs[mid].haystack < h) lo = mid + 1; else hi = mid; }
    rec_first[h] = lo;
}

template <bool IC>
__global__ void __launch_bounds__(256) k_rp_pass(RpTables t, const uint8_t* __restrict__ text, const uint64_t* __restrict__ offsets,
                                                 const Record* __restrict__ recs, const uint64_t* __restrict__ rec_first,
                                                 const int64_t* __restrict__ thr, uint64_t max_len, RpKept* __restrict__ kept,
                                                 RpHay* __restrict__ hs, RpRoute route, uint32_t n_act, uint32_t keep_all, RpFused fu)
{
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t h = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    uint64_t r0, r1;
    if (fu.rec_first_w) {
        // fused: the record range of the haystack by two binary searches of its own (k_rp_ranges), the trailing elements of the scans' inputs by
        // the wavefront after the last haystack
        if (h > n_act) return;
        const uint64_t n_rec = fu.n_rec_dev ? *fu.n_rec_dev : fu.n_rec;
        auto first_of = [&](uint64_t hh) { uint64_t lo = 0, hi = n_rec; while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (recs[mid].haystack < hh) lo = mid + 1; else hi = mid; } return lo; };
        r0 = first_of(h);
        if (lane == 0) fu.rec_first_w[h] = r0;
        if (h == n_act) {
            if (lane == 0) {
                route.len_next[h] = 0; route.len_fin[h] = 0; route.tiles[h] = 0; route.act[h] = 0; route.fin[h] = 0;
                if (fu.need) { fu.need[h] = 0; fu.nwin[h] = 0; }
            }
            return;
        }
        r1 = first_of((uint64_t)h + 1);
    } else {
        if (h == n_act && fu.need) {                     // ranges given (the previous pass's merge left them), counts fused: the trailing elements
            if (lane == 0) { route.len_next[h] = 0; route.len_fin[h] = 0; route.tiles[h] = 0; route.act[h] = 0; route.fin[h] = 0; fu.need[h] = 0; fu.nwin[h] = 0; }
            return;
        }
        if (h >= n_act) return;
        r0 = rec_first[h]; r1 = rec_first[h + 1];
    }
    const uint64_t hoff = offsets[h], curlen = offsets[h + 1] - hoff;
    const int64_t threshold = thr[h];

    // ---- prependMatch, first half: the best priority below the threshold (Replacer.hs:255-258)
    int64_t best = INT64_MIN;
    for (uint64_t r = r0 + lane; r < r1; r += kWave) {
        const uint32_t st = recs[r].state;
        for (uint64_t k = t.vals_off[st], ke = t.vals_off[st + 1]; k < ke; k++) {
            const int64_t p = t.payloads[t.vals[k]].priority;
            if (p < threshold && p > best) best = p;
        }
    }
    best = wave_max_i64(best);

    uint32_t status = kRpFinished, nkept = 0, payload = 0;
    uint64_t newlen = curlen;
    if (best != INT64_MIN) {
        // ---- second half: the matches that carry it, makeMatch, removeOverlap
        int64_t delta_all = 0, delta_kept = 0;
        uint64_t last_end = 0;
        for (uint64_t base = r0; base < r1; base += kWave) {
            const uint64_t r = base + lane;
            bool sel = false; uint32_t pl = 0; uint64_t end_pos = 0;
            if (r < r1) {
                const Record rec = recs[r];
                end_pos = rec.end_pos;
                for (uint64_t k = t.vals_off[rec.state], ke = t.vals_off[rec.state + 1]; k < ke; k++) {
                    const uint32_t v = t.vals[k];
                    if (t.payloads[v].priority == best) { sel = true; pl = v; }
                }
            }
            uint64_t start = 0, len = 0; int64_t delta = 0;
            if (sel) {
                const RpPayload pp = t.payloads[pl];
                if (!IC) { len = pp.len_bytes; start = end_pos - len; }                       // Replacer.hs:266-267
                else {                                                                        // :268-274
                    start = pp.len_code_points == 0 ? end_pos : skip_code_points_backwards(text + hoff, end_pos - 1, pp.len_code_points - 1);
                    len = end_pos - start;
                }
                delta = (int64_t)pp.repl_len - (int64_t)len;
            }
            delta_all += delta;
            // removeOverlap (:191-198): in position order keep a match iff it starts at or after the end of the last kept one
            uint64_t pending = keep_all ? 0ull : __ballot(sel);
            bool keep = keep_all && sel;          // am_run_priority: the caller removes overlaps itself
            while (pending) {
                const uint64_t ok = __ballot(sel && start >= last_end) & pending;
                if (!ok) break;
                const int l = __ffsll((unsigned long long)ok) - 1;
                if (lane == l) keep = true;
                last_end = __shfl(start + len, l, kWave);
                pending &= l == 63 ? 0ull : ~((2ull << l) - 1ull);
            }
            const uint64_t keepmask = __ballot(keep);
            if (keepmask) {
                const int64_t kd = keep ? delta : 0;
                const int64_t incl = wave_inclusive_sum_i64(kd, lane);
                if (keep) {
                    const uint32_t rank = __popcll(keepmask & ((1ull << lane) - 1ull));
                    RpKept e; e.src_start = start; e.src_len = len; e.dst = (uint64_t)((int64_t)start + delta_kept + (incl - kd));
                    kept[r0 + nkept + rank] = e;
                    payload = pl;
                }
                delta_kept += __shfl(incl, kWave - 1, kWave);
                nkept += __popcll(keepmask);
            }
        }
        delta_all = wave_sum_i64(delta_all);
        payload = (uint32_t)wave_max_i64((int64_t)payload);     // uniform: every kept match has the same payload
        const int64_t newlen_all = (int64_t)curlen + delta_all;                                // replacementLength over ALL matches (:240)
        if (newlen_all > 0 && (uint64_t)newlen_all > max_len) { status = kRpNothing; newlen = 0; nkept = 0; }
        else {
            newlen = (uint64_t)((int64_t)curlen + delta_kept);
            status = best == t.min_priority ? kRpFinished : kRpActive;                          // :241-242
        }
    }
    if (lane == 0) {
        RpHay o; o.newlen = newlen; o.best = best; o.status = status; o.nkept = nkept; o.payload = payload; o.pad = 0;
        hs[h] = o;
        if (keep_all) { route.tiles[h] = nkept; return; }
        route.len_next[h] = status == kRpActive ? newlen : 0;
        route.len_fin[h] = status == kRpFinished ? newlen : 0;
        route.tiles[h] = status == kRpNothing ? 0u : (uint32_t)((newlen + kRpTile - 1) / kRpTile);
        route.act[h] = status == kRpActive ? 1u : 0u;
        route.fin[h] = status == kRpActive ? 0u : 1u;
        if (fu.need) {                                   // (k_pt_count)
            fu.need[h] = status != kRpNothing ? fu.pc_cnt[h] + 2u * nkept + 1u : 0u;
            fu.nwin[h] = status == kRpActive ? nkept : 0u;
        }
    }
}

// ---- the same fold, parallel over RECORDS instead of one wavefront per haystack (few haystacks with very many matches) ----
// prependMatch: k_rpp_best (atomic max per haystack) and k_rpp_select (which records carry the best priority, makeMatch);
// removeOverlap: selected matches are sorted by start and all have the same code-point length, so a match that does not
// overlap its predecessor is kept whatever happened before it; only runs of consecutively overlapping matches need the
// serial greedy (k_rpp_heads marks the run heads, k_rpp_greedy walks each run); k_rpp_kept / k_rpp_hay write exactly what
// k_rp_pass writes (kept[], hs[], route), so everything after the fold is shared.
__global__ void __launch_bounds__(256) k_rpp_best(RpTables t, const Record* __restrict__ recs, uint64_t n_rec, const int64_t* __restrict__ thr,
                                                  int64_t* __restrict__ best)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t b = INT64_MIN; uint32_t hay = 0xFFFFFFFFu;
    if (r < n_rec) {
        const Record rec = recs[r];
        hay = rec.haystack;
        const int64_t threshold = thr[hay];
        for (uint64_t k = t.vals_off[rec.state], ke = t.vals_off[rec.state + 1]; k < ke; k++) {
            const int64_t p = t.payloads[t.vals[k]].priority;
            if (p < threshold && p > b) b = p;
        }
    }
    const uint32_t h0 = __shfl(hay, 0, kWave);
    if (__ballot(hay != h0 && hay != 0xFFFFFFFFu) == 0) {          // the usual case: the whole wavefront inside one haystack
        b = wave_max_i64(b);
        if ((threadIdx.x & (kWave - 1)) == 0 && b != INT64_MIN && h0 != 0xFFFFFFFFu) atomicMax((long long*)(best + h0), (long long)b);
    } else if (b != INT64_MIN) atomicMax((long long*)(best + hay), (long long)b);
}

template <bool IC>
__global__ void __launch_bounds__(256) k_rpp_select(RpTables t, const uint8_t* __restrict__ text, const uint64_t* __restrict__ offsets,
                                                    const Record* __restrict__ recs, uint64_t n_rec, const int64_t* __restrict__ best,
                                                    uint32_t* __restrict__ selflag, RpSel* __restrict__ cand, int64_t* __restrict__ delta_all,
                                                    uint32_t* __restrict__ payload_of)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rec) return;
    if (r == n_rec) { selflag[r] = 0; return; }                    // the scan's trailing element
    const Record rec = recs[r];
    const int64_t b = best[rec.haystack];
    uint32_t flag = 0;
    if (b != INT64_MIN) {
        for (uint64_t k = t.vals_off[rec.state], ke = t.vals_off[rec.state + 1]; k < ke; k++) {
            const uint32_t v = t.vals[k];
            const RpPayload pp = t.payloads[v];
            if (pp.priority != b) continue;
            uint64_t start, len;
            if (!IC) { len = pp.len_bytes; start = rec.end_pos - len; }
            else {
                start = pp.len_code_points == 0 ? rec.end_pos : skip_code_points_backwards(text + offsets[rec.haystack], rec.end_pos - 1, pp.len_code_points - 1);
                len = rec.end_pos - start;
            }
            RpSel c; c.start = start; c.len = len; c.haystack = rec.haystack; c.pad = 0;
            cand[r] = c;
            payload_of[rec.haystack] = v;
            atomicAdd((unsigned long long*)(delta_all + rec.haystack), (unsigned long long)((int64_t)pp.repl_len - (int64_t)len));
            flag = 1;
        }
    }
    selflag[r] = flag;
}

__global__ void __launch_bounds__(256) k_rpp_compact(const uint32_t* __restrict__ selflag, const uint64_t* __restrict__ sidx, const RpSel* __restrict__ cand,
                                                     uint64_t n_rec, RpSel* __restrict__ sel)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rec && selflag[r]) sel[sidx[r]] = cand[r];
}

__global__ void __launch_bounds__(256) k_rpp_heads(const RpSel* __restrict__ sel, const uint64_t* __restrict__ n_sel_dev, uint32_t* __restrict__ keep)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, n_sel = *n_sel_dev;
    if (i > n_sel) return;
    if (i == n_sel) { keep[i] = 0; return; }
    const RpSel c = sel[i];
    bool head = i == 0;
    if (!head) { const RpSel p = sel[i - 1]; head = p.haystack != c.haystack || c.start >= p.start + p.len; }
    keep[i] = head ? 3u : 0u;                                      // bit 0: kept, bit 1: head of a run
}

// removeOverlap (Replacer.hs:191-198) inside the runs of consecutively overlapping matches.  One WAVEFRONT walks a run, 64 matches
// at a time: which of them survive is a chain (the next kept match is the first one that starts at or after the end of the last
// kept one), followed with ballot + ffs in registers -- ~20 cycles per kept match instead of a dependent global load.  A periodic
// document (1 MB of "a", needle "aa") is ONE run of a million matches; a single thread took a second for it.  Every wavefront owns
// the run heads among its 64 indices and does their runs one after the other.
__global__ void __launch_bounds__(256) k_rpp_greedy(const RpSel* __restrict__ sel, const uint64_t* __restrict__ n_sel_dev, uint32_t* __restrict__ keep)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane, n_sel = *n_sel_dev;
    if (base >= n_sel) return;
    const uint64_t i = base + lane;
    uint64_t heads = __ballot(i < n_sel && (keep[i] & 2u));
    while (heads) {
        const uint32_t h = (uint32_t)__ffsll((unsigned long long)heads) - 1u;
        heads &= heads - 1ull;
        const RpSel head = sel[base + h];
        uint64_t last_end = head.start + head.len;
        for (uint64_t j0 = base + h + 1;; j0 += kWave) {
            const uint64_t j = j0 + lane;
            const bool in = j < n_sel;
            const uint32_t k = in ? keep[j] : 2u;
            const uint64_t stop = __ballot(!in || (k & 2u));                       // the run ends at the next head (or at the end of the list)
            const uint32_t limit = stop ? (uint32_t)__ffsll((unsigned long long)stop) - 1u : (uint32_t)kWave;
            RpSel c{0, 0, 0, 0};
            if (lane < limit) c = sel[j];
            uint32_t cur = 0;
            for (;;) {
                const uint64_t m = __ballot(lane >= cur && lane < limit && c.start >= last_end);
                if (!m) break;
                const uint32_t l = (uint32_t)__ffsll((unsigned long long)m) - 1u;
                if (lane == l) keep[j] = 1u;
                const uint64_t e = c.start + c.len;
                last_end = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(e >> 32), (int)l, kWave) << 32) | (uint32_t)__shfl((int)(uint32_t)e, (int)l, kWave);
                cur = l + 1u;
            }
            if (limit < (uint32_t)kWave) break;
        }
    }
}

__global__ void __launch_bounds__(256) k_rpp_kflags(const RpSel* __restrict__ sel, const uint64_t* __restrict__ n_sel_dev, const uint32_t* __restrict__ keep,
                                                    RpTables t, const uint32_t* __restrict__ payload_of, uint32_t* __restrict__ kflag, uint64_t* __restrict__ kdelta)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, n_sel = *n_sel_dev;
    if (i > n_sel) return;
    const bool k = i < n_sel && (keep[i] & 1u);
    kflag[i] = k ? 1u : 0u;
    kdelta[i] = k ? (uint64_t)((int64_t)t.payloads[payload_of[sel[i].haystack]].repl_len - (int64_t)sel[i].len) : 0ull;
}

__global__ void __launch_bounds__(256) k_rpp_kept(const RpSel* __restrict__ sel, const uint64_t* __restrict__ n_sel_dev, const uint32_t* __restrict__ kflag,
                                                  const uint64_t* __restrict__ kidx, const uint64_t* __restrict__ kdpre, const uint64_t* __restrict__ sidx,
                                                  const uint64_t* __restrict__ rec_first, RpKept* __restrict__ kept)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, n_sel = *n_sel_dev;
    if (i >= n_sel || !kflag[i]) return;
    const RpSel c = sel[i];
    const uint64_t r0 = rec_first[c.haystack], s0 = sidx[r0];      // first record / first selected match of the haystack
    RpKept e; e.src_start = c.start; e.src_len = c.len; e.dst = (uint64_t)((int64_t)c.start + (int64_t)(kdpre[i] - kdpre[s0]));
    kept[r0 + (kidx[i] - kidx[s0])] = e;
}

__global__ void __launch_bounds__(256) k_rpp_hay(RpTables t, const uint64_t* __restrict__ offsets, const uint64_t* __restrict__ rec_first,
                                                 const uint64_t* __restrict__ sidx, const uint64_t* __restrict__ kidx, const uint64_t* __restrict__ kdpre,
                                                 const int64_t* __restrict__ best, const int64_t* __restrict__ delta_all, const uint32_t* __restrict__ payload_of,
                                                 uint64_t max_len, RpHay* __restrict__ hs, RpRoute route, uint32_t n_act)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_act) return;
    const uint64_t curlen = offsets[h + 1] - offsets[h];
    const uint64_t s0 = sidx[rec_first[h]], s1 = sidx[rec_first[h + 1]];
    const int64_t b = best[h];
    uint32_t status = kRpFinished, nkept = 0; uint64_t newlen = curlen;
    if (b != INT64_MIN) {
        const int64_t newlen_all = (int64_t)curlen + delta_all[h];
        if (newlen_all > 0 && (uint64_t)newlen_all > max_len) { status = kRpNothing; newlen = 0; }
        else {
            nkept = (uint32_t)(kidx[s1] - kidx[s0]);
            newlen = (uint64_t)((int64_t)curlen + (int64_t)(kdpre[s1] - kdpre[s0]));
            status = b == t.min_priority ? kRpFinished : kRpActive;
        }
    }
    RpHay o; o.newlen = newlen; o.best = b; o.status = status; o.nkept = nkept; o.payload = b != INT64_MIN ? payload_of[h] : 0u; o.pad = 0;
    hs[h] = o;
    route.len_next[h] = status == kRpActive ? newlen : 0;
    route.len_fin[h] = status == kRpFinished ? newlen : 0;
    route.tiles[h] = status == kRpNothing ? 0u : (uint32_t)((newlen + kRpTile - 1) / kRpTile);
    route.act[h] = status == kRpActive ? 1u : 0u;
    route.fin[h] = status == kRpActive ? 0u : 1u;
}

__global__ void __launch_bounds__(256) k_fill_i64(int64_t* __restrict__ p, uint64_t n, int64_t v)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

hipError_t launch_rpp_best(const RpTables& t, const Record* recs, uint64_t n_rec, const int64_t* thr, int64_t* best, uint32_t n_act, hipStream_t st)
{
    hipLaunchKernelGGL(k_fill_i64, dim3((n_act + 256) / 256), dim3(256), 0, st, best, (uint64_t)n_act + 1, INT64_MIN);
    if (n_rec == 0) return hipGetLastError();
    hipLaunchKernelGGL(k_rpp_best, dim3((uint32_t)((n_rec + 255) / 256)), dim3(256), 0, st, t, recs, n_rec, thr, best);
    return hipGetLastError();
}
hipError_t launch_rpp_select(bool ic, const RpTables& t, const uint8_t* text, const uint64_t* offsets, const Record* recs, uint64_t n_rec, const int64_t* best,
                             uint32_t* selflag, RpSel* cand, int64_t* delta_all, uint32_t* payload_of, hipStream_t st)
{
    const dim3 grid((uint32_t)((n_rec + 1 + 255) / 256)), block(256);
    if (ic) hipLaunchKernelGGL(k_rpp_select<true>, grid, block, 0, st, t, text, offsets, recs, n_rec, best, selflag, cand, delta_all, payload_of);
    else hipLaunchKernelGGL(k_rpp_select<false>, grid, block, 0, st, t, text, offsets, recs, n_rec, best, selflag, cand, delta_all, payload_of);
    return hipGetLastError();
}
hipError_t launch_rpp_compact(const uint32_t* selflag, const uint64_t* sidx, const RpSel* cand, uint64_t n_rec, RpSel* sel, hipStream_t st)
{
    if (n_rec == 0) return hipSuccess;
    hipLaunchKernelGGL(k_rpp_compact, dim3((uint32_t)((n_rec + 255) / 256)), dim3(256), 0, st, selflag, sidx, cand, n_rec, sel);
    return hipGetLastError();
}
// the kernels over selected matches are launched for the upper bound n_rec + 1 threads; the real count is read on the device
hipError_t launch_rpp_overlaps(const RpSel* sel, const uint64_t* n_sel_dev, uint64_t bound, uint32_t* keep, hipStream_t st)
{
    const dim3 grid((uint32_t)((bound + 1 + 255) / 256)), block(256);
    hipLaunchKernelGGL(k_rpp_heads, grid, block, 0, st, sel, n_sel_dev, keep);
    hipLaunchKernelGGL(k_rpp_greedy, grid, block, 0, st, sel, n_sel_dev, keep);
    return hipGetLastError();
}
hipError_t launch_rpp_kflags(const RpSel* sel, const uint64_t* n_sel_dev, uint64_t bound, const uint32_t* keep, const RpTables& t, const uint32_t* payload_of,
                             uint32_t* kflag, uint64_t* kdelta, hipStream_t st)
{
    hipLaunchKernelGGL(k_rpp_kflags, dim3((uint32_t)((bound + 1 + 255) / 256)), dim3(256), 0, st, sel, n_sel_dev, keep, t, payload_of, kflag, kdelta);
    return hipGetLastError();
}
hipError_t launch_rpp_finish(const RpTables& t, const RpSel* sel, const uint64_t* n_sel_dev, uint64_t bound, const uint32_t* kflag, const uint64_t* kidx,
                             const uint64_t* kdpre, const uint64_t* sidx, const uint64_t* offsets, const uint64_t* rec_first, const int64_t* best,
                             const int64_t* delta_all, const uint32_t* payload_of, uint64_t max_len, RpKept* kept, RpHay* hs, const RpRoute& route, uint32_t n_act,
                             hipStream_t st)
{
    hipLaunchKernelGGL(k_rpp_kept, dim3((uint32_t)((bound + 255) / 256 + 1)), dim3(256), 0, st, sel, n_sel_dev, kflag, kidx, kdpre, sidx, rec_first, kept);
    hipLaunchKernelGGL(k_rpp_hay, dim3((n_act + 255) / 256), dim3(256), 0, st, t, offsets, rec_first, sidx, kidx, kdpre, best, delta_all, payload_of, max_len, hs, route, n_act);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_rp_route(const RpHay* __restrict__ hs, RpRouted rt, const uint32_t* __restrict__ orig, uint32_t n_act,
                                                  uint64_t* __restrict__ next_offsets, uint32_t* __restrict__ next_orig, int64_t* __restrict__ next_thr,
                                                  RpFin* __restrict__ fin)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h > n_act) return;
    if (h == n_act) { next_offsets[rt.act_idx[n_act]] = rt.off_next[n_act]; return; }
    const RpHay s = hs[h];
    if (s.status == kRpActive) {
        const uint64_t j = rt.act_idx[h];
        next_offsets[j] = rt.off_next[h]; next_orig[j] = orig[h]; next_thr[j] = s.best;
    } else {
        RpFin f; f.off = rt.off_fin[h]; f.len = s.newlen; f.orig = orig[h]; f.status = s.status;
        fin[rt.fin_idx[h]] = f;
    }
}

__global__ void __launch_bounds__(256) k_rp_splice(RpTables t, const uint8_t* __restrict__ text, const uint64_t* __restrict__ offsets,
                                                   const uint64_t* __restrict__ rec_first, const RpKept* __restrict__ kept,
                                                   const RpHay* __restrict__ hs, RpRouted rt, const uint32_t* __restrict__ tile_hay,
                                                   uint8_t* __restrict__ text_next, uint8_t* __restrict__ text_fin)
{
    typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
    const uint64_t tile = blockIdx.x;
    const uint32_t h = tile_hay[tile];            // k_rp_tilemap: one load instead of a binary search per workgroup
    const RpHay s = hs[h];
    const uint8_t* src = text + offsets[h];
    uint8_t* dst = s.status == kRpActive ? text_next + rt.off_next[h] : text_fin + rt.off_fin[h];
    const RpKept* K = kept + rec_first[h];
    const uint32_t nk = s.nkept;
    const uint64_t newlen = s.newlen;
    const RpPayload pp = t.payloads[s.payload];
    const uint8_t* repl = t.repl + pp.repl_off;
    const uint64_t repl_len = nk ? pp.repl_len : 0;
    const uint64_t tbase = (tile - rt.tile_off[h]) * kRpTile;
    // Most tiles lie inside ONE gap between replacements (a 64-KiB haystack has a few replacements per pass): then the
    // tile is a plain copy from a constant source offset.  Everything here is uniform across the workgroup.
    int64_t kb = -1;                                   // last kept match whose replacement starts at or before the tile
    { int64_t a = 0, b = nk; while (a < b) { const int64_t mid = (a + b) >> 1; if (K[mid].dst <= tbase) a = mid + 1; else b = mid; } kb = a - 1; }
    const uint64_t tend = tbase + kRpTile < newlen ? tbase + kRpTile : newlen;
    uint64_t gap_lo = 0, gap_hi = newlen, gap_src = 0;  // new-text range of the gap after kb and where its first byte comes from
    if (nk) {
        if (kb >= 0) { const RpKept e = K[kb]; gap_lo = e.dst + repl_len; gap_src = e.src_start + e.src_len; }
        if (kb + 1 < (int64_t)nk) gap_hi = K[kb + 1].dst;
    }
    if (tbase >= gap_lo && tend <= gap_hi) {
        const uint8_t* from = src + gap_src - gap_lo;    // from[p] is the source of new-text byte p
#pragma unroll
        for (uint32_t it = 0; it < kRpTile / (256 * 16); it++) {
            const uint64_t o = tbase + ((uint64_t)it * 256 + threadIdx.x) * 16;
            if (o + 16 <= tend) *reinterpret_cast<u32x4_u*>(dst + o) = *reinterpret_cast<const u32x4_u*>(from + o);
            else if (o < tend) for (uint64_t p = o; p < tend; p++) dst[p] = from[p];
        }
        return;
    }
#pragma unroll 1
    for (uint32_t it = 0; it < kRpTile / (256 * 16); it++) {
        const uint64_t o = tbase + ((uint64_t)it * 256 + threadIdx.x) * 16;
        if (o >= newlen) break;
        const uint64_t end = o + 16 < newlen ? o + 16 : newlen;
        // last kept match whose replacement starts at or before o (-1: none)
        int64_t k = -1;
        { int64_t a = 0, b = nk; while (a < b) { const int64_t mid = (a + b) >> 1; if (K[mid].dst <= o) a = mid + 1; else b = mid; } k = a - 1; }
        uint64_t p = o;
        while (p < end) {
            while (k + 1 < (int64_t)nk && K[k + 1].dst <= p) k++;
            bool in_repl = false; uint64_t so = p, seg_end;
            if (k < 0) seg_end = K[0].dst;
            else {
                const RpKept e = K[k];
                const uint64_t re = e.dst + repl_len;
                if (p < re) { in_repl = true; so = p - e.dst; seg_end = re; }
                else { so = e.src_start + e.src_len + (p - re); seg_end = k + 1 < (int64_t)nk ? K[k + 1].dst : newlen; }
            }
            const uint64_t stop = seg_end < end ? seg_end : end;
            const uint8_t* from = in_repl ? repl + so : src + so;
            if (stop - p == 16) *reinterpret_cast<u32x4_u*>(dst + p) = *reinterpret_cast<const u32x4_u*>(from);
            else for (uint64_t q = 0; q < stop - p; q++) dst[p + q] = from[q];
            p = stop;
        }
    }
}

hipError_t launch_rp_ranges(const Record* recs, uint64_t n_rec, uint64_t* rec_first, const RpRoute& route, uint32_t n_act, hipStream_t st)
{
    const uint32_t n = n_act + 1;
    hipLaunchKernelGGL(k_rp_ranges, dim3((n + 255) / 256), dim3(256), 0, st, recs, n_rec, (const uint64_t*)nullptr, rec_first, route, n_act);
    return hipGetLastError();
}

hipError_t launch_rp_ranges_dev(const Record* recs, const uint64_t* n_rec_dev, uint64_t* rec_first, const RpRoute& route, uint32_t n_act, hipStream_t st)
{
    const uint32_t n = n_act + 1;
    hipLaunchKernelGGL(k_rp_ranges, dim3((n + 255) / 256), dim3(256), 0, st, recs, (uint64_t)0, n_rec_dev, rec_first, route, n_act);
    return hipGetLastError();
}

__global__ void k_rp_totals(RpRouted rt, uint32_t n_act, const uint64_t* __restrict__ win_off, const uint64_t* __restrict__ woffs, uint64_t woffs_last,
                            const uint64_t* __restrict__ extra8, const uint64_t* __restrict__ extra9, uint64_t* __restrict__ out10, uint64_t seq)
{
    out10[0] = rt.off_next[n_act]; out10[1] = rt.off_fin[n_act]; out10[2] = rt.tile_off[n_act]; out10[3] = rt.act_idx[n_act]; out10[4] = rt.fin_idx[n_act];
    out10[5] = win_off ? win_off[n_act] : 0;                                                  // windows of the incremental re-scan and their bytes
    out10[6] = woffs ? woffs[woffs_last == ~0ull ? win_off[n_act] : woffs_last] : 0;          // (~0: the scan stopped at the last window)
    out10[7] = 0;
    out10[8] = extra8 ? *extra8 : 0;                                                          // two more scalars of the pass, so that ONE copy brings
    out10[9] = extra9 ? *extra9 : 0;                                                          // everything to the host (each copy is a 16-us blit of its own)
    if (seq) {                                            // out10 is pinned host memory and the host spins on this word: the values above first
        __threadfence_system();
        __hip_atomic_store(&out10[15], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_rp_totals(const RpRouted& rt, uint32_t n_act, const uint64_t* win_off, const uint64_t* woffs, uint64_t woffs_last, uint64_t* out10, hipStream_t st,
                            const uint64_t* extra8, const uint64_t* extra9, uint64_t seq)
{
    hipLaunchKernelGGL(k_rp_totals, dim3(1), dim3(1), 0, st, rt, n_act, win_off, woffs, woffs_last, extra8, extra9, out10, seq);
    return hipGetLastError();
}

hipError_t launch_rp_pass(bool ic, const RpTables& t, const uint8_t* text, const uint64_t* offsets, const Record* recs, const uint64_t* rec_first,
                          const int64_t* thr, uint64_t max_len, RpKept* kept, RpHay* hs, const RpRoute& route, uint32_t n_act, uint32_t keep_all, hipStream_t st,
                          const RpFused* fused)
{
    const RpFused fu = fused ? *fused : RpFused{nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    const dim3 grid((n_act + (fused ? 1u : 0u) + 3) / 4), block(256);      // fused: one more wavefront for the trailing elements
    if (ic) hipLaunchKernelGGL(k_rp_pass<true>, grid, block, 0, st, t, text, offsets, recs, rec_first, thr, max_len, kept, hs, route, n_act, keep_all, fu);
    else hipLaunchKernelGGL(k_rp_pass<false>, grid, block, 0, st, t, text, offsets, recs, rec_first, thr, max_len, kept, hs, route, n_act, keep_all, fu);
    return hipGetLastError();
}

// am_run_priority: the selected matches of every haystack, compacted in haystack order
__global__ void __launch_bounds__(256) k_rp_gather(const RpHay* __restrict__ hs, const uint64_t* __restrict__ rec_first, const RpKept* __restrict__ kept,
                                                   const uint64_t* __restrict__ out_off, RpSelected* __restrict__ out, int64_t* __restrict__ best_out, uint32_t n_act)
{
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t h = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    if (h >= n_act) return;
    const RpHay s = hs[h];
    if (lane == 0) best_out[h] = s.best;
    const RpKept* K = kept + rec_first[h];
    for (uint32_t j = lane; j < s.nkept; j += kWave) {
        RpSelected o; o.start = K[j].src_start; o.len = K[j].src_len; o.haystack = h; o.payload = s.payload;
        out[out_off[h] + j] = o;
    }
}

hipError_t launch_rp_gather(const RpHay* hs, const uint64_t* rec_first, const RpKept* kept, const uint64_t* out_off, RpSelected* out, int64_t* best_out,
                            uint32_t n_act, hipStream_t st)
{
    if (n_act == 0) return hipSuccess;
    hipLaunchKernelGGL(k_rp_gather, dim3((n_act + 3) / 4), dim3(256), 0, st, hs, rec_first, kept, out_off, out, best_out, n_act);
    return hipGetLastError();
}

hipError_t launch_rp_route(const RpHay* hs, const RpRouted& rt, const uint32_t* orig, uint32_t n_act, uint64_t* next_offsets, uint32_t* next_orig,
                           int64_t* next_thr, RpFin* fin, hipStream_t st)
{
    const uint32_t n = n_act + 1;
    hipLaunchKernelGGL(k_rp_route, dim3((n + 255) / 256), dim3(256), 0, st, hs, rt, orig, n_act, next_offsets, next_orig, next_thr, fin);
    return hipGetLastError();
}

// tile -> haystack map of k_rp_splice: one wavefront per haystack writes its own index into its tiles' entries
__global__ void __launch_bounds__(256) k_rp_tilemap(RpRouted rt, uint32_t n_act, uint32_t* __restrict__ tile_hay)
{
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t h = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    if (h >= n_act) return;
    for (uint64_t t = rt.tile_off[h] + lane, e = rt.tile_off[h + 1]; t < e; t += kWave) tile_hay[t] = h;
}

hipError_t launch_rp_splice(const RpTables& t, const uint8_t* text, const uint64_t* offsets, const uint64_t* rec_first, const RpKept* kept,
                            const RpHay* hs, const RpRouted& rt, uint32_t n_act, uint64_t n_tiles, uint32_t* tile_hay, uint8_t* text_next, uint8_t* text_fin,
                            hipStream_t st)
{
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_rp_tilemap, dim3((n_act + 3) / 4), dim3(256), 0, st, rt, n_act, tile_hay);
    hipLaunchKernelGGL(k_rp_splice, dim3((uint32_t)n_tiles), dim3(256), 0, st, t, text, offsets, rec_first, kept, hs, rt, tile_hay, text_next, text_fin);
    return hipGetLastError();
}

// ---- incremental re-scan between Replacer passes -------------------------------------------------------------------------
// The reference re-scans the whole rewritten haystack in every pass (Replacer.hs:223-225).  Whether a needle ends at a
// position depends only on the `ov` bytes before it (ov >= the longest needle in haystack bytes), so after a pass the
// records of the new text are: the old records outside the neighbourhood of the replacements, shifted; plus the
// records of a scan of small windows around the replacements.  For kept match j of a haystack (new-text coordinates,
// dst_j = start of its replacement, rl = replacement length):
//   own range  (dst_j, min(dst_j + rl + ov, dst_{j+1}, newlen)]     end positions re-derived from window j
//   window     [dst_j - ov, upper end of the own range)
//   old records with end in (src_start_j, src_start_j + src_len_j + ov] are dropped, the others shift with the text.
// k_rp_win_meta lays the windows out, k_rp_win_copy gathers their text into a small batch (scanned by k_sf like any
// other batch), k_rp_merge<false/true> counts / writes the next pass's sorted record list.
__global__ void __launch_bounds__(256) k_rp_win_count(const RpHay* __restrict__ hs, uint32_t n_act, uint32_t* __restrict__ nwin)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h > n_act) return;
    nwin[h] = (h < n_act && hs[h].status == kRpActive) ? hs[h].nkept : 0u;
}

__global__ void __launch_bounds__(256) k_rp_win_meta(RpTables t, RpRouted rt,
                                                     const RpHay* __restrict__ hs, const uint64_t* __restrict__ rec_first, const RpKept* __restrict__ kept,
                                                     const uint64_t* __restrict__ win_off, uint32_t ov, RpWin* __restrict__ wins, uint32_t* __restrict__ wlen,
                                                     uint32_t n_act, bool pt)
{
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t h = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    if (h == n_act && lane == 0) wlen[win_off[n_act]] = 0;          // the scan's trailing element
    if (h >= n_act) return;
    const RpHay s = hs[h];
    if (s.status != kRpActive) return;
    const RpKept* K = kept + rec_first[h];
    const uint64_t base = rt.off_next[h], w0 = win_off[h];      // where the haystack will start in the next text
    const uint64_t rl = t.payloads[s.payload].repl_len;
    for (uint32_t j = lane; j < s.nkept; j += kWave) {
        const uint64_t dst = K[j].dst;
        uint64_t hi = dst + rl + ov;
        if (hi > s.newlen) hi = s.newlen;
        if (j + 1 < s.nkept && K[j + 1].dst < hi) hi = K[j + 1].dst;
        // a window may start inside a code point: k_sf compares bytes, and no needle starts with a continuation byte
        const uint64_t ws = dst > ov ? dst - ov : 0;
        RpWin w; w.src_abs = pt ? (rt.act_idx[h] << 40) | ws : base + ws; w.ws = ws;     // piece-table path: (index in the next pass, start) instead of an address
        w.len = hi > dst ? (uint32_t)(hi - ws) : 0u;                // empty own range: nothing to scan
        w.own_lo = (uint32_t)(dst - ws);
        wins[w0 + j] = w;
        wlen[w0 + j] = w.len;
    }
}

__global__ void __launch_bounds__(256) k_rp_win_copy(const RpWin* __restrict__ wins, const uint64_t* __restrict__ woffs, const uint8_t* __restrict__ text_next,
                                                     uint8_t* __restrict__ wtext, uint64_t n_win)
{
    const int lane = threadIdx.x & (kWave - 1);
    const uint64_t wi = (uint64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    if (wi >= n_win) return;
    const RpWin w = wins[wi];
    const uint8_t* src = text_next + w.src_abs;
    uint8_t* dst = wtext + woffs[wi];
    for (uint32_t i = lane; i < w.len; i += kWave) dst[i] = src[i];
}

namespace {
// first index in [lo, hi) whose end_pos is > x
__device__ __forceinline__ uint64_t upper_bound_end(const Record* __restrict__ r, uint64_t lo, uint64_t hi, uint64_t x)
{
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (r[mid].end_pos <= x) lo = mid + 1; else hi = mid; }
    return lo;
}
}  // namespace

template <bool WRITE>
__global__ void __launch_bounds__(256) k_rp_merge(const Record* __restrict__ recs, const uint64_t* __restrict__ rec_first, const RpKept* __restrict__ kept,
                                                  const RpHay* __restrict__ hs, const uint64_t* __restrict__ offsets, RpRouted rt,
                                                  const uint64_t* __restrict__ win_off, const RpWin* __restrict__ wins, const Record* __restrict__ wrecs,
                                                  const uint64_t* __restrict__ wrec_first, uint32_t ov, uint32_t n_act,
                                                  uint32_t* __restrict__ mcount, const uint64_t* __restrict__ moff, Record* __restrict__ out)
{
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t h = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    if (!WRITE && h == 0 && lane == 0) mcount[rt.act_idx[n_act]] = 0;   // the scan's trailing element
    if (h >= n_act) return;
    const RpHay s = hs[h];
    if (s.status != kRpActive) return;
    const uint32_t a = (uint32_t)rt.act_idx[h];                     // index of the haystack in the next pass
    const uint64_t r0 = rec_first[h], r1 = rec_first[h + 1], w0 = win_off[h];
    const RpKept* K = kept + r0;
    const uint64_t curlen = offsets[h + 1] - offsets[h];
    uint64_t cursor = WRITE ? moff[a] : 0, n = 0, at = r0;
    for (uint32_t j = 0; j < s.nkept; j++) {
        const RpKept k = K[j];
        // old records that end at or before the replaced region: unchanged context, they move with the text
        const uint64_t e = upper_bound_end(recs, at, r1, k.src_start);
        if (WRITE) {
            const int64_t shift = (int64_t)k.dst - (int64_t)k.src_start;
            for (uint64_t i = at + lane; i < e; i += kWave) { Record r = recs[i]; r.end_pos = (uint64_t)((int64_t)r.end_pos + shift); r.haystack = a; out[cursor + (i - at)] = r; }
        }
        cursor += e - at; n += e - at;
        // the window's own records
        const RpWin w = wins[w0 + j];
        const uint64_t q1 = wrec_first[w0 + j + 1];
        const uint64_t q0 = upper_bound_end(wrecs, wrec_first[w0 + j], q1, w.own_lo);
        if (WRITE) for (uint64_t i = q0 + lane; i < q1; i += kWave) { Record r = wrecs[i]; r.end_pos += w.ws; r.haystack = a; out[cursor + (i - q0)] = r; }
        cursor += q1 - q0; n += q1 - q0;
        // old records that touch the replaced bytes are gone
        at = upper_bound_end(recs, e, r1, k.src_start + k.src_len + ov);
    }
    if (WRITE) {
        const int64_t shift = (int64_t)s.newlen - (int64_t)curlen;
        for (uint64_t i = at + lane; i < r1; i += kWave) { Record r = recs[i]; r.end_pos = (uint64_t)((int64_t)r.end_pos + shift); r.haystack = a; out[cursor + (i - at)] = r; }
    }
    n += r1 - at;
    if (!WRITE && lane == 0) mcount[a] = (uint32_t)n;
}

hipError_t launch_rp_win_count(const RpHay* hs, uint32_t n_act, uint32_t* nwin, hipStream_t st)
{
    hipLaunchKernelGGL(k_rp_win_count, dim3((n_act + 1 + 255) / 256), dim3(256), 0, st, hs, n_act, nwin);
    return hipGetLastError();
}

hipError_t launch_rp_win_meta(const RpTables& t, const RpRouted& rt, const RpHay* hs, const uint64_t* rec_first,
                              const RpKept* kept, const uint64_t* win_off, uint32_t ov, RpWin* wins, uint32_t* wlen, uint32_t n_act, hipStream_t st, bool pt)
{
    hipLaunchKernelGGL(k_rp_win_meta, dim3((n_act + 1 + 3) / 4), dim3(256), 0, st, t, rt, hs, rec_first, kept, win_off, ov, wins, wlen, n_act, pt);
    return hipGetLastError();
}

hipError_t launch_rp_win_copy(const RpWin* wins, const uint64_t* woffs, const uint8_t* text_next, uint8_t* wtext, uint64_t n_win, hipStream_t st)
{
    if (n_win == 0) return hipSuccess;
    hipLaunchKernelGGL(k_rp_win_copy, dim3((uint32_t)((n_win + 3) / 4)), dim3(256), 0, st, wins, woffs, text_next, wtext, n_win);
    return hipGetLastError();
}

hipError_t launch_rp_merge(bool write, const Record* recs, const uint64_t* rec_first, const RpKept* kept, const RpHay* hs, const uint64_t* offsets, const RpRouted& rt,
                           const uint64_t* win_off, const RpWin* wins, const Record* wrecs, const uint64_t* wrec_first, uint32_t ov, uint32_t n_act,
                           uint32_t* mcount, const uint64_t* moff, Record* out, hipStream_t st)
{
    if (n_act == 0) return hipSuccess;
    const dim3 grid((n_act + 3) / 4), block(256);
    if (write) hipLaunchKernelGGL(k_rp_merge<true>, grid, block, 0, st, recs, rec_first, kept, hs, offsets, rt, win_off, wins, wrecs, wrec_first, ov, n_act, mcount, moff, out);
    else hipLaunchKernelGGL(k_rp_merge<false>, grid, block, 0, st, recs, rec_first, kept, hs, offsets, rt, win_off, wins, wrecs, wrec_first, ov, n_act, mcount, moff, out);
    return hipGetLastError();
}

// ---- piece table: the text of a haystack between Replacer passes without rewriting it ------------------------------------
// `replace` (Replacer.hs:163-180) rebuilds the whole text in every pass; on BASELINE config 5 that is ~160 passes over 64-KiB
// haystacks that each change a dozen bytes per pass: 129 GiB written for 1 GiB of input.  Here the current text of an active
// haystack is a list of pieces -- (source, logical start) pairs pointing into the caller's batch (never modified) or into the
// replacement blob, closed by a sentinel carrying the total length.  A pass turns the list into the next one (k_pt_build: old
// pieces with the kept matches cut out and the replacement spliced in, one thread per haystack: a few hundred 16-byte entries);
// bytes move only twice: into the small windows that are re-scanned (k_pt_win_copy) and, once per haystack, into the result when
// the haystack is finished (k_pt_materialise).  CaseSensitive replacers only: makeMatch of an IgnoreCase replacer walks the
// text backwards (skipCodePointsBackwards), those keep the splicing path.

__global__ void __launch_bounds__(256) k_pt_init(const uint64_t* __restrict__ offsets, uint32_t n_act, RpPiece* __restrict__ pieces, uint64_t* __restrict__ pc_start,
                                                 uint32_t* __restrict__ pc_cnt)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_act) return;
    pieces[2ull * h] = RpPiece{offsets[h], 0};
    pieces[2ull * h + 1] = RpPiece{0, offsets[h + 1] - offsets[h]};          // sentinel: logical end
    pc_start[h] = 2ull * h; pc_cnt[h] = 1;
}

// upper bound of the next list's entries per haystack (the scan's input): old pieces + 2 per kept match + sentinel
// (and the windows of the incremental re-scan per haystack, what k_rp_win_count computes: one launch for both)
__global__ void __launch_bounds__(256) k_pt_count(const RpHay* __restrict__ hs, const uint32_t* __restrict__ pc_cnt, uint32_t n_act, uint32_t* __restrict__ need,
                                                  uint32_t* __restrict__ nwin)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h > n_act) return;
    need[h] = (h < n_act && hs[h].status != kRpNothing) ? pc_cnt[h] + 2u * hs[h].nkept + 1u : 0u;
    nwin[h] = (h < n_act && hs[h].status == kRpActive) ? hs[h].nkept : 0u;
}

// One wavefront per haystack.  Old piece i = old text [ls, le).  The kept matches (sorted, disjoint) cut it into fragments; a
// fragment that starts at old position x lies, in the new text, at x + (sum of the length changes of the matches before it)
// = K[j].dst + repl_len + (x - end of K[j]) for the last match j before x.  The replacement of match j is emitted by the piece
// that contains its first byte.  Two sweeps over the pieces (count, then write) with a wave prefix sum between them.
__global__ void __launch_bounds__(256) k_pt_build(RpTables t, const RpHay* __restrict__ hs, const uint64_t* __restrict__ rec_first, const RpKept* __restrict__ kept,
                                                  const RpPiece* __restrict__ pieces, const uint64_t* __restrict__ pc_start, const uint32_t* __restrict__ pc_cnt,
                                                  const uint64_t* __restrict__ need_off, RpRouted rt, uint32_t n_act,
                                                  RpPiece* __restrict__ out, uint64_t* __restrict__ next_start, uint32_t* __restrict__ next_cnt,
                                                  uint64_t* __restrict__ fin_start, uint32_t* __restrict__ fin_cnt)
{
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t h = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    if (h >= n_act) return;
    const RpHay s = hs[h];
    if (s.status == kRpNothing) { if (lane == 0) { fin_start[rt.fin_idx[h]] = 0; fin_cnt[rt.fin_idx[h]] = 0; } return; }
    const RpPiece* P = pieces + pc_start[h];
    const uint32_t n = pc_cnt[h];                        // P[n] is the sentinel
    const RpKept* K = kept + rec_first[h];
    const uint32_t nk = s.nkept;
    const RpPayload pp = t.payloads[s.payload];
    const uint64_t repl_len = nk ? pp.repl_len : 0;
    RpPiece* Q = out + need_off[h];
    // what piece i contributes: the matches j in [ja, jb) are those that end after its start and start before its end
    auto span = [&](uint64_t ls, uint64_t le, uint32_t& ja, uint32_t& jb) {
        uint32_t a = 0, b = nk;
        while (a < b) { const uint32_t mid = (a + b) >> 1; if (K[mid].src_start + K[mid].src_len <= ls) a = mid + 1; else b = mid; }
        ja = a; jb = a;
        while (jb < nk && K[jb].src_start < le) jb++;
    };
    // walks piece i: f(kind, src, new_pos) for every entry it emits, in order
    auto emit = [&](uint32_t i, auto&& f) {
        const uint64_t ls = P[i].lstart, le = P[i + 1].lstart;
        if (le == ls) return;
        uint32_t ja, jb; span(ls, le, ja, jb);
        uint64_t x = ls;                                  // old position where the next fragment would start
        // new position of old position x when it is not inside a match: after the last match that ends at or before x
        auto new_pos = [&](uint64_t xx, uint32_t jprev_plus1) -> uint64_t {
            if (jprev_plus1 == 0) return xx;
            const RpKept k = K[jprev_plus1 - 1];
            return k.dst + repl_len + (xx - (k.src_start + k.src_len));
        };
        uint32_t jp = ja;                                 // matches [0, jp) end at or before x
        for (uint32_t j = ja; j < jb; j++) {
            const RpKept k = K[j];
            if (k.src_start > x) f(P[i].src + (x - ls), new_pos(x, jp));                          // fragment before the match
            if (k.src_start >= ls && repl_len) f(kPieceRepl | pp.repl_off, k.dst);                  // the match starts in this piece: its replacement
            x = k.src_start + k.src_len;
            jp = j + 1;
            if (x >= le) break;
        }
        if (x < le) f(P[i].src + (x - ls), new_pos(x, jp));
    };
    // 64 pieces per round, lane = piece within the round: count, prefix over the lanes, write -- rounds follow each other in order
    uint32_t base = 0;
    for (uint32_t r0 = 0; r0 < n; r0 += kWave) {
        const uint32_t i = r0 + lane;
        uint32_t c = 0;
        if (i < n) emit(i, [&](uint64_t, uint64_t) { c++; });
        const uint32_t incl = (uint32_t)wave_inclusive_sum_i64((int64_t)c, lane);
        uint32_t at = base + incl - c;
        if (i < n) emit(i, [&](uint64_t src, uint64_t pos) { Q[at++] = RpPiece{src, pos}; });
        base += __shfl(incl, kWave - 1, kWave);
    }
    if (lane == 0) {
        Q[base] = RpPiece{0, s.newlen};                   // sentinel
        if (s.status == kRpActive) { const uint64_t a = rt.act_idx[h]; next_start[a] = need_off[h]; next_cnt[a] = base; }
        else { const uint64_t f = rt.fin_idx[h]; fin_start[f] = need_off[h]; fin_cnt[f] = base; }
    }
}

// bytes [lo, lo + len) of a piece list into dst, one wavefront
__device__ __forceinline__ void pt_gather(const RpPiece* __restrict__ P, uint32_t n, const uint8_t* __restrict__ text, const uint8_t* __restrict__ repl,
                                          uint64_t lo, uint64_t len, uint8_t* __restrict__ dst, int lane)
{
    if (len == 0) return;
    uint32_t a = 0, b = n;                               // last piece whose start is <= lo
    while (b - a > 1) { const uint32_t mid = (a + b) >> 1; if (P[mid].lstart <= lo) a = mid; else b = mid; }
    uint64_t pos = lo;
    const uint64_t end = lo + len;
    for (uint32_t i = a; pos < end; i++) {
        const uint64_t pe = P[i + 1].lstart < end ? P[i + 1].lstart : end;
        const uint64_t s = P[i].src;
        const uint8_t* from = ((s & kPieceRepl) ? repl + (s & ~kPieceRepl) : text + s) + (pos - P[i].lstart);
        uint8_t* to = dst + (pos - lo);
        for (uint64_t x = lane; x < pe - pos; x += kWave) to[x] = from[x];
        pos = pe;
    }
}

// finished haystacks: their text, once.  One workgroup per haystack, a wavefront per piece -- and a piece is a few hundred bytes reached through its list entry, so a
// wavefront that takes piece after piece waits out two dependent trips (entry, bytes) 65 times: 0.98 ms per GiB where a plain copy takes 0.40 (tools/microbench/
// copy_rate.hip).  So the list is read ONCE into LDS by the whole workgroup, and a wavefront has FOUR pieces in flight: their first KiB each is asked for before any of it
// is stored (16-byte copies, the last bytes one by one; what a piece has beyond a KiB follows in a loop of its own).
constexpr uint32_t kMatPieces = 1024;                 // list entries a workgroup keeps in LDS (a longer list is read from memory entry by entry)
constexpr uint32_t kMatFlight = 4;
__global__ void __launch_bounds__(256) k_pt_materialise(const RpPiece* __restrict__ pieces, const uint64_t* __restrict__ fin_start, const uint32_t* __restrict__ fin_cnt,
                                                        const RpFin* __restrict__ fin, const uint8_t* __restrict__ text, const uint8_t* __restrict__ repl,
                                                        uint8_t* __restrict__ text_fin)
{
    typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
    __shared__ RpPiece s_p[kMatPieces + 1];
    const uint32_t f = blockIdx.x;
    const RpFin m = fin[f];
    if (m.status == kRpNothing || m.len == 0) return;
    const RpPiece* P = pieces + fin_start[f];
    const uint32_t n = fin_cnt[f];
    uint8_t* dst = text_fin + m.off;
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const bool staged = n <= kMatPieces;
    if (staged) {
        for (uint32_t i = threadIdx.x; i <= n; i += 256) s_p[i] = P[i];
        __syncthreads();
    }
    for (uint32_t i0 = wave * kMatFlight; i0 < n; i0 += (256 / kWave) * kMatFlight) {
        const uint8_t* from[kMatFlight]; uint8_t* to[kMatFlight]; uint64_t len[kMatFlight];
        u32x4_u v[kMatFlight]; uint32_t tail[kMatFlight];
#pragma unroll
        for (uint32_t u = 0; u < kMatFlight; u++) {
            const uint32_t i = i0 + u;
            len[u] = 0;
            if (i < n) {
                const RpPiece a = staged ? s_p[i] : P[i], b = staged ? s_p[i + 1] : P[i + 1];
                len[u] = b.lstart - a.lstart;
                from[u] = (a.src & kPieceRepl) ? repl + (a.src & ~kPieceRepl) : text + a.src;
                to[u] = dst + a.lstart;
            }
        }
        // first KiB of each: lanes below n16 take 16 bytes, the first `rest` lanes one of the last bytes each as well
#pragma unroll
        for (uint32_t u = 0; u < kMatFlight; u++) {
            const uint64_t head = len[u] < 1024u ? len[u] : 1024u;
            const uint32_t n16 = (uint32_t)(head / 16u), rest = (uint32_t)(head & 15u);
            if (lane < n16) v[u] = *reinterpret_cast<const u32x4_u*>(from[u] + 16u * lane);
            tail[u] = 0;
            if (lane < rest && len[u] <= 1024u) tail[u] = from[u][16u * n16 + lane];
        }
#pragma unroll
        for (uint32_t u = 0; u < kMatFlight; u++) {
            const uint64_t head = len[u] < 1024u ? len[u] : 1024u;
            const uint32_t n16 = (uint32_t)(head / 16u), rest = (uint32_t)(head & 15u);
            if (lane < n16) *reinterpret_cast<u32x4_u*>(to[u] + 16u * lane) = v[u];
            if (lane < rest && len[u] <= 1024u) to[u][16u * n16 + lane] = (uint8_t)tail[u];
        }
        // what a piece has beyond its first KiB (untouched stretches of the text between two replacements)
#pragma unroll
        for (uint32_t u = 0; u < kMatFlight; u++) {
            if (len[u] <= 1024u) continue;
            const uint64_t n16 = len[u] / 16;
            for (uint64_t x = 64u + lane; x < n16; x += kWave) *reinterpret_cast<u32x4_u*>(to[u] + 16 * x) = *reinterpret_cast<const u32x4_u*>(from[u] + 16 * x);
            for (uint64_t x = n16 * 16 + lane; x < len[u]; x += kWave) to[u][x] = from[u][x];
        }
    }
}

// the whole next text of every active haystack (only when the windows of a pass would be larger than the text itself: tiny inputs)
__global__ void __launch_bounds__(256) k_pt_materialise_next(const RpPiece* __restrict__ pieces, const uint64_t* __restrict__ next_start, const uint32_t* __restrict__ next_cnt,
                                                             const uint64_t* __restrict__ next_offsets, uint32_t n_next, const uint8_t* __restrict__ text,
                                                             const uint8_t* __restrict__ repl, uint8_t* __restrict__ out)
{
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t a = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    if (a >= n_next) return;
    pt_gather(pieces + next_start[a], next_cnt[a], text, repl, 0, next_offsets[a + 1] - next_offsets[a], out + next_offsets[a], lane);
}

hipError_t launch_pt_materialise_next(const RpPiece* pieces, const uint64_t* next_start, const uint32_t* next_cnt, const uint64_t* next_offsets, uint32_t n_next,
                                      const uint8_t* text, const uint8_t* repl, uint8_t* out, hipStream_t st)
{
    if (n_next == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pt_materialise_next, dim3((n_next + 3) / 4), dim3(256), 0, st, pieces, next_start, next_cnt, next_offsets, n_next, text, repl, out);
    return hipGetLastError();
}

// the windows of the incremental re-scan, gathered from the NEXT pass's piece lists (RpWin::src_abs = next index << 40 | start)
__global__ void __launch_bounds__(256) k_pt_win_copy(const RpWin* __restrict__ wins, const uint64_t* __restrict__ woffs, const RpPiece* __restrict__ pieces,
                                                     const uint64_t* __restrict__ next_start, const uint32_t* __restrict__ next_cnt,
                                                     const uint8_t* __restrict__ text, const uint8_t* __restrict__ repl, uint8_t* __restrict__ wtext, uint64_t n_win,
                                                     uint64_t total_w, uint64_t padded)
{
    const int lane = threadIdx.x & (kWave - 1);
    const uint64_t wi = (uint64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    if (wi == n_win) { for (uint64_t x = total_w + lane; x < padded; x += kWave) wtext[x] = 0; return; }      // the zero tail the scan kernels expect
    if (wi > n_win) return;
    const RpWin w = wins[wi];
    const uint64_t a = w.src_abs >> 40, lo = w.src_abs & ((1ull << 40) - 1ull);
    pt_gather(pieces + next_start[a], next_cnt[a], text, repl, lo, w.len, wtext + woffs[wi], lane);
}

hipError_t launch_pt_init(const uint64_t* offsets, uint32_t n_act, RpPiece* pieces, uint64_t* pc_start, uint32_t* pc_cnt, hipStream_t st)
{
    if (n_act == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pt_init, dim3((n_act + 255) / 256), dim3(256), 0, st, offsets, n_act, pieces, pc_start, pc_cnt);
    return hipGetLastError();
}
hipError_t launch_pt_count(const RpHay* hs, const uint32_t* pc_cnt, uint32_t n_act, uint32_t* need, uint32_t* nwin, hipStream_t st)
{
    hipLaunchKernelGGL(k_pt_count, dim3((n_act + 1 + 255) / 256), dim3(256), 0, st, hs, pc_cnt, n_act, need, nwin);
    return hipGetLastError();
}
hipError_t launch_pt_build(const RpTables& t, const RpHay* hs, const uint64_t* rec_first, const RpKept* kept, const RpPiece* pieces, const uint64_t* pc_start,
                           const uint32_t* pc_cnt, const uint64_t* need_off, const RpRouted& rt, uint32_t n_act, RpPiece* out, uint64_t* next_start, uint32_t* next_cnt,
                           uint64_t* fin_start, uint32_t* fin_cnt, hipStream_t st)
{
    if (n_act == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pt_build, dim3((n_act + 3) / 4), dim3(256), 0, st, t, hs, rec_first, kept, pieces, pc_start, pc_cnt, need_off, rt, n_act, out, next_start, next_cnt,
                       fin_start, fin_cnt);
    return hipGetLastError();
}
hipError_t launch_pt_materialise(const RpPiece* pieces, const uint64_t* fin_start, const uint32_t* fin_cnt, const RpFin* fin, uint32_t n_fin, const uint8_t* text,
                                 const uint8_t* repl, uint8_t* text_fin, hipStream_t st)
{
    if (n_fin == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pt_materialise, dim3(n_fin), dim3(256), 0, st, pieces, fin_start, fin_cnt, fin, text, repl, text_fin);
    return hipGetLastError();
}
hipError_t launch_pt_win_copy(const RpWin* wins, const uint64_t* woffs, const RpPiece* pieces, const uint64_t* next_start, const uint32_t* next_cnt, const uint8_t* text,
                              const uint8_t* repl, uint8_t* wtext, uint64_t n_win, uint64_t total_w, uint64_t padded, hipStream_t st)
{
    if (n_win == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pt_win_copy, dim3((uint32_t)((n_win + 1 + 3) / 4)), dim3(256), 0, st, wins, woffs, pieces, next_start, next_cnt, text, repl, wtext, n_win, total_w, padded);
    return hipGetLastError();
}

// ---- Searcher.containsAll (Searcher.hs:173-187) on the records: the IntSet of needle ids still missing, as one
// bitmap of `words` 32-bit words per haystack.  k_idset sets the bit of every reported id, k_idset_all tests
// whether all n_needles bits of a haystack are set (IS.null of the final accumulator).
__global__ void __launch_bounds__(256) k_idset(const Record* __restrict__ recs, uint64_t r0, uint64_t r1, const uint64_t* __restrict__ vals_off,
                                               const uint32_t* __restrict__ vals, uint32_t n_needles, uint32_t hay0, uint32_t words, uint32_t* __restrict__ bits)
{
    const uint64_t r = r0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= r1) return;
    const Record rec = recs[r];
    uint32_t* row = bits + (uint64_t)(rec.haystack - hay0) * words;
    for (uint64_t k = vals_off[rec.state], ke = vals_off[rec.state + 1]; k < ke; k++) {
        const uint32_t id = vals[k];
        if (id >= n_needles) continue;                           // IS.delete of an absent key
        const uint32_t m = 1u << (id & 31);
        if (!(row[id >> 5] & m)) atomicOr(row + (id >> 5), m);      // a stale read only costs an extra atomic
    }
}

__global__ void __launch_bounds__(256) k_idset_all(const uint32_t* __restrict__ bits, uint32_t words, uint32_t n_needles, uint32_t n_hay, uint8_t* __restrict__ flags)
{
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t h = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    if (h >= n_hay) return;
    const uint32_t* row = bits + (uint64_t)h * words;
    int64_t c = 0;
    for (uint32_t w = lane; w < words; w += kWave) c += __popc(row[w]);
    c = wave_sum_i64(c);
    if (lane == 0) flags[h] = c == (int64_t)n_needles ? 1 : 0;
}

hipError_t launch_idset(const Record* recs, uint64_t r0, uint64_t r1, const uint64_t* vals_off, const uint32_t* vals, uint32_t n_needles,
                        uint32_t hay0, uint32_t words, uint32_t* bits, hipStream_t st)
{
    if (r1 <= r0) return hipSuccess;
    hipLaunchKernelGGL(k_idset, dim3((uint32_t)((r1 - r0 + 255) / 256)), dim3(256), 0, st, recs, r0, r1, vals_off, vals, n_needles, hay0, words, bits);
    return hipGetLastError();
}

hipError_t launch_idset_all(const uint32_t* bits, uint32_t words, uint32_t n_needles, uint32_t n_hay, uint8_t* flags, hipStream_t st)
{
    if (n_hay == 0) return hipSuccess;
    hipLaunchKernelGGL(k_idset_all, dim3((n_hay + 3) / 4), dim3(256), 0, st, bits, words, n_needles, n_hay, flags);
    return hipGetLastError();
}

// ---- checksum of the fold sequence (SURVEY 8d "parity check at scale"): per haystack, the left fold
//   h' = h * P + mix(matchPos, value)      over every (record, value of machineValues ! record.state) in order
// i.e. exactly what the reference's runWithCase would hand to a fold function, reduced to 64 bits.  One wavefront per
// haystack: every lane folds a contiguous block of the haystack's records, the (hash, count) pairs are combined with
// the associative rule (h1, c1) . (h2, c2) = (h1 * P^c2 + h2, c1 + c2).
constexpr uint64_t kFoldP = 0x100000001B3ull;
__device__ __forceinline__ uint64_t fold_mix(uint64_t pos, uint32_t v)
{
    uint64_t x = (pos * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)v + 0x632BE59BD9B4E019ull);
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32;
    return x;
}
__device__ __forceinline__ uint64_t fold_pow(uint64_t e)
{
    uint64_t r = 1, b = kFoldP;
    while (e) { if (e & 1) r *= b; b *= b; e >>= 1; }
    return r;
}
__global__ void __launch_bounds__(256) k_fold_hash(const Record* __restrict__ recs, const uint64_t* __restrict__ rec_first, const uint64_t* __restrict__ vals_off,
                                                   const uint32_t* __restrict__ vals, uint32_t n_hay, uint64_t* __restrict__ hash_out, uint64_t* __restrict__ count_out)
{
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t h = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    if (h >= n_hay) return;
    const uint64_t r0 = rec_first[h], r1 = rec_first[h + 1];
    const uint64_t per = (r1 - r0 + kWave - 1) / kWave;
    uint64_t a = r0 + per * lane, z = a + per;
    if (a > r1) a = r1;
    if (z > r1) z = r1;
    uint64_t hh = 0, cnt = 0;
    for (uint64_t r = a; r < z; r++) {
        const Record rec = recs[r];
        for (uint64_t k = vals_off[rec.state], ke = vals_off[rec.state + 1]; k < ke; k++) { hh = hh * kFoldP + fold_mix(rec.end_pos, vals[k]); cnt++; }
    }
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {            // lanes whose index is a multiple of 2d absorb their right neighbour's block
        const uint64_t oh = __shfl_down(hh, d, kWave), oc = __shfl_down(cnt, d, kWave);
        hh = hh * fold_pow(oc) + oh; cnt += oc;
    }
    if (lane == 0) { hash_out[h] = hh; if (count_out) count_out[h] = cnt; }
}

hipError_t launch_fold_hash(const Record* recs, const uint64_t* rec_first, const uint64_t* vals_off, const uint32_t* vals, uint32_t n_hay,
                            uint64_t* hash_out, uint64_t* count_out, hipStream_t st)
{
    if (n_hay == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fold_hash, dim3((n_hay + 3) / 4), dim3(256), 0, st, recs, rec_first, vals_off, vals, n_hay, hash_out, count_out);
    return hipGetLastError();
}

// ---- several small exclusive sums in ONE launch (the Replacer's per-pass bookkeeping: a dozen scan launches otherwise).
// One 1024-thread workgroup per job walks its array in tiles of 4096 elements: 4 elements per thread, wave scan with
// shuffles, wave totals through LDS, running carry.  Meant for arrays up to a few hundred thousand elements.
__global__ void __launch_bounds__(1024) k_scan_jobs(ScanJobs jobs)
{
    // exclusive sums by ONE workgroup per job, 4 096 elements per sweep.  What a sweep costs is the latency of its loads (9.3 us for the five
    // sweeps of 16 385 elements, whatever the barriers cost): the loads of FOUR sweeps are issued together (index clamped, value masked: no
    // branch between them), then the four sweeps run.  Per sweep: a wavefront scan, the 16 wavefront totals through LDS (two buffers that take
    // turns: one barrier per sweep) and a second wavefront scan over those 16 in every wavefront, which gives the wavefront's offset and the
    // sweep's total -- the running carry stays in registers.
    constexpr int kBatch = 4;
    __shared__ uint64_t wave_tot[2][16];
    const ScanJob j = jobs.j[blockIdx.x];
    const uint64_t n = j.n_dev ? *j.n_dev + j.n : j.n;
    if (n == 0) return;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    uint64_t carry = 0;
    int turn = 0;
    for (uint64_t base0 = 0; base0 < n; base0 += 4096 * kBatch) {
        uint64_t v[kBatch][4];
        if (j.in64) {
#pragma unroll
            for (int bb = 0; bb < kBatch; bb++)
#pragma unroll
                for (int k = 0; k < 4; k++) { const uint64_t i = base0 + 4096u * bb + (uint64_t)threadIdx.x * 4 + k; const uint64_t x = j.in64[i < n ? i : n - 1]; v[bb][k] = i < n ? x : 0ull; }
        } else {
#pragma unroll
            for (int bb = 0; bb < kBatch; bb++)
#pragma unroll
                for (int k = 0; k < 4; k++) { const uint64_t i = base0 + 4096u * bb + (uint64_t)threadIdx.x * 4 + k; const uint32_t x = j.in32[i < n ? i : n - 1]; v[bb][k] = i < n ? (uint64_t)x : 0ull; }
        }
#pragma unroll
        for (int bb = 0; bb < kBatch; bb++) {
            const uint64_t base = base0 + 4096u * bb;
            if (base >= n) break;                         // (uniform)
            const uint64_t i0 = base + (uint64_t)threadIdx.x * 4;
            const uint64_t mine = v[bb][0] + v[bb][1] + v[bb][2] + v[bb][3];
            const int64_t incl = wave_inclusive_sum_i64((int64_t)mine, lane);
            if (lane == kWave - 1) wave_tot[turn][wave] = (uint64_t)incl;
            __syncthreads();
            const uint64_t wt = lane < 16 ? wave_tot[turn][lane] : 0ull;
            const int64_t wincl = wave_inclusive_sum_i64((int64_t)wt, lane);               // lanes 0..15: totals of the wavefronts up to and including `lane`
            const uint64_t before_waves = (uint64_t)__shfl(wincl, wave, kWave) - (uint64_t)__shfl((int64_t)wt, wave, kWave);
            const uint64_t sweep_total = (uint64_t)__shfl(wincl, 15, kWave);
            uint64_t run = carry + before_waves + (uint64_t)incl - mine;
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint64_t i = i0 + k; if (i < n) j.out[i] = run; run += v[bb][k]; }
            carry += sweep_total;
            turn ^= 1;
        }
    }
}

hipError_t launch_scan_jobs(const ScanJobs& jobs, hipStream_t st)
{
    if (jobs.n_jobs == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scan_jobs, dim3(jobs.n_jobs), dim3(1024), 0, st, jobs);
    return hipGetLastError();
}

}  // namespace dev
}  // namespace am
