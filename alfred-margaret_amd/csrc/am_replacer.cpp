// am_replacer.cpp -- Replacer (Replacer.hs:97-274) behind include/am.h: every pass on the device.  Three loops: all passes of a haystack in one
// kernel (replacer_run_loop, csrc/am_rploop.hip), pass by pass with the texts kept as piece tables (replacer_run_pt) or spliced (replacer_run,
// csrc/am_replace.hip).
#include "am_host.h"

using namespace am;
using namespace am::dev;
using namespace am::host;

// ------------------------------------------------------------------ Replacer (Replacer.hs:97-274), device-resident passes

static_assert(sizeof(am_payload) == sizeof(RpPayload) && offsetof(am_payload, repl_off) == offsetof(RpPayload, repl_off) &&
                  offsetof(am_payload, len_code_points) == offsetof(RpPayload, len_code_points) && offsetof(am_payload, repl_len) == offsetof(RpPayload, repl_len),
              "am_payload must mirror the device payload");

struct am_replacer {
    const am_automaton* a = nullptr;
    int case_mode = 0;
    DevBuf vals_off, vals, payloads, repl, one;
    RpTables t{};
    uint32_t max_repl_len = 0;                        // longest replacement (bounds the re-scan window of the one-kernel loop)
    uint64_t n_repl_bytes = 0;                        // size of the replacement blob
    bool pl_implicit = false;                         // payloads[i].priority == -i for every i (Replacer.hs:100-104): k_rp_lds runs without its payload column (am_rplds.hip, PLI)
    uint32_t max_needle_bytes = 0;                    // longest needle of the AUTOMATON in bytes (depth of its trie in UTF-8 bytes; 0: unknown) = the longest CaseSensitive match
    // the workspace of the last run (device buffers, pinned scratch, copy stream) is kept for the next one: a caller that
    // rewrites one document per call would otherwise pay ~40 hipMalloc/hipFree (4 ms) each time
    mutable std::mutex session_mu;
    mutable std::vector<void*> sessions;              // workspaces of finished runs, kept for the next ones (several: concurrent groups / threads)
    void (*session_delete)(void*) = nullptr;
};

// Finished texts are copied D2H straight into pinned slabs that the result object keeps (no second host
// copy); am_replaced_free hands the slabs back to a small process-wide pool so that repeated calls do
// not pay for pinning again.
namespace {
struct Slab { uint8_t* p = nullptr; size_t cap = 0, used = 0; };
struct SlabPool {
    std::mutex mu;
    std::vector<Slab> free_list;
    bool device = false;           // slabs in the current device's HBM (results that stay on the device) instead of pinned host memory
    static constexpr size_t kSlab = 256ull << 20, kKeep = 8;
    int take(size_t need, Slab* out)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < free_list.size(); i++)
                if (free_list[i].cap >= need) { *out = free_list[i]; out->used = 0; free_list.erase(free_list.begin() + i); return AM_OK; }
        }
        Slab s; s.cap = need > kSlab ? need : kSlab;
        if (device) { if (hipMalloc((void**)&s.p, s.cap) != hipSuccess) return fail(AM_ERR_OOM, "hipMalloc(result slab) failed"); }
        else if (hipHostMalloc((void**)&s.p, s.cap, hipHostMallocPortable) != hipSuccess) return fail(AM_ERR_OOM, "hipHostMalloc(result slab) failed");
        *out = s;
        return AM_OK;
    }
    void give(const Slab& s)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (free_list.size() < kKeep) { free_list.push_back(s); return; }
        }
        if (device) (void)hipFree(s.p); else (void)hipHostFree(s.p);
    }
};
SlabPool g_slabs;
struct DevSlabPools { SlabPool p[kMaxDev]; DevSlabPools() { for (SlabPool& x : p) x.device = true; } } g_dev_slabs;
}  // namespace

struct am_replaced {
    struct Item { const uint8_t* p = nullptr; size_t len = 0; };
    std::vector<Item> text;
    std::vector<uint8_t> just;
    std::vector<Slab> slabs;
    uint64_t passes = 0, scanned = 0, spliced = 0;
    int dev = -1;                  // >= 0: the texts stay in that device's memory (am_replacer_run_batch_device)
    SlabPool& pool() const { return dev >= 0 ? g_dev_slabs.p[dev] : g_slabs; }
    ~am_replaced() { for (const Slab& s : slabs) pool().give(s); }
    // room for n contiguous bytes in the current slab, or a new slab
    int room(size_t n, uint8_t** out)
    {
        if (slabs.empty() || slabs.back().cap - slabs.back().used < n) { Slab s; AM_TRY(pool().take(n, &s)); slabs.push_back(s); }
        *out = slabs.back().p + slabs.back().used;
        slabs.back().used += (n + 63) & ~(size_t)63;
        if (slabs.back().used > slabs.back().cap) slabs.back().used = slabs.back().cap;
        return AM_OK;
    }
};

extern "C" int am_replacer_create(const am_automaton* a, int case_mode, const uint64_t* values_offsets, const uint32_t* values,
                                  const am_payload* payloads, size_t n_payloads, const uint8_t* repl_bytes, size_t n_repl_bytes,
                                  int64_t min_priority, am_replacer** out)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    const Flavor* f = nullptr;
    AM_TRY(prepare(a, case_mode, &f));
    ON_DEVICE(a->dev);
    const uint64_t n_states = f->h.n_states;
    if (!values_offsets || values_offsets[0] != 0) return fail(AM_ERR_INVALID, "values_offsets[0] must be 0");
    uint32_t max_repl = 0, max_needle = 0;
    const uint64_t n_values = values_offsets[n_states];
    if ((n_values && !values) || (n_payloads && !payloads) || (n_repl_bytes && !repl_bytes)) return fail(AM_ERR_INVALID, "null table");
    for (uint64_t s = 0; s < n_states; s++) {
        if (values_offsets[s + 1] < values_offsets[s]) return fail(AM_ERR_INVALID, "values_offsets must be non-decreasing");
        if (a->has_ref && values_offsets[s + 1] - values_offsets[s] != a->values_len[s])
            return fail(AM_ERR_INVALID, "values_offsets disagrees with the values_len given to am_automaton_create");
    }
    for (uint64_t k = 0; k < n_values; k++) if (values[k] >= n_payloads) return fail(AM_ERR_INVALID, "payload index out of range");
    {
        // Replacer.hs:100-104 / :127-131: priorities are 0, -1, -2, ...; the device pass relies on them being distinct
        std::vector<int64_t> pr(n_payloads);
        for (size_t i = 0; i < n_payloads; i++) {
            pr[i] = payloads[i].priority;
            if (payloads[i].repl_len > max_repl) max_repl = payloads[i].repl_len;
            if (pr[i] > 0) return fail(AM_ERR_INVALID, "priorities must be <= 0 (the initial threshold is 1, Replacer.hs:211)");
            if ((uint64_t)payloads[i].repl_off + payloads[i].repl_len > n_repl_bytes) return fail(AM_ERR_INVALID, "replacement slice out of range");
            if (case_mode == AM_IGNORE_CASE && payloads[i].len_code_points == 0)
                return fail(AM_ERR_UNSUPPORTED, "empty needle under IgnoreCase: the reference's skipCodePointsBackwards has no answer (Utf8.hs:259)");
        }
        std::sort(pr.begin(), pr.end());
        for (size_t i = 1; i < n_payloads; i++) if (pr[i] == pr[i - 1]) return fail(AM_ERR_INVALID, "payload priorities must be distinct");
    }
    // How far a CaseSensitive match reaches back = the longest needle IN THE AUTOMATON, in bytes.  The payloads' len_bytes are the lengths of the
    // ORIGINAL needles (Replacer.hs:112) and say nothing about that: a replacer built IgnoreCase holds the lower-cased needles, and lower-casing can
    // add bytes (U+023A, two bytes, becomes U+2C65, three); setCaseSensitivity (Replacer.hs:148-153) then runs those needles CaseSensitive.  So the
    // bound is read off the trie: the deepest state, each goto edge counted with the UTF-8 length of its code point.  Handles attached to a received image
    // carry no arrays: 0 = unknown, and the loop falls back to the bound by code points.
    max_needle = 0;
    if (a->has_ref) {
        const size_t S = a->offsets.size() - 1;
        std::vector<uint32_t> depth(S, 0), queue; queue.reserve(S); queue.push_back(0);
        for (size_t q = 0; q < queue.size(); q++) {
            const uint32_t st = queue[q];
            for (uint64_t i = a->offsets[st]; i < a->transitions.size(); i++) {
                const uint64_t t = a->transitions[i];
                if (t & kWildcard) break;
                const uint32_t cp = (uint32_t)(t & 0x1fffffu), nx = (uint32_t)(t >> 32);
                if (nx >= S || nx == 0) break;                               // (validated at creation; never taken)
                depth[nx] = depth[st] + (cp < 0x80u ? 1u : cp < 0x800u ? 2u : cp < 0x10000u ? 3u : 4u);
                if (depth[nx] > max_needle) max_needle = depth[nx];
                queue.push_back(nx);
            }
        }
    }
    am_replacer* r = new am_replacer();
    r->a = a; r->case_mode = case_mode; r->max_repl_len = max_repl; r->max_needle_bytes = max_needle; r->n_repl_bytes = n_repl_bytes;
    auto up = [&](DevBuf& d, const void* src, size_t bytes) -> int {
        AM_TRY(d.ensure(bytes + 64));
        if (bytes) HIP_TRY(hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice));
        return AM_OK;
    };
    r->pl_implicit = n_payloads > 0 && n_payloads < (1ull << 31);
    for (uint64_t i = 0; i < n_payloads && r->pl_implicit; i++) r->pl_implicit = payloads[i].priority == -(int64_t)i;
    int rc = up(r->vals_off, values_offsets, (n_states + 1) * sizeof(uint64_t));
    if (rc == AM_OK) rc = up(r->vals, values, n_values * sizeof(uint32_t));
    if (rc == AM_OK) rc = up(r->payloads, payloads, n_payloads * sizeof(am_payload));
    if (rc == AM_OK && n_payloads == 0) { hipError_t e = hipMemset(r->payloads.p, 0, sizeof(am_payload)); if (e != hipSuccess) rc = fail(AM_ERR_HIP, hipGetErrorString(e)); }
    if (rc == AM_OK) rc = up(r->repl, repl_bytes, n_repl_bytes);
    if (rc == AM_OK) {
        std::vector<RpStateOne> one(n_states);
        for (uint64_t s = 0; s < n_states; s++) {
            const uint64_t n = values_offsets[s + 1] - values_offsets[s];
            RpStateOne e{0, kRpWalkList};
            if (n == 1) {
                const uint32_t v = values[values_offsets[s]];
                const int64_t pr = payloads[v].priority;
                if (pr >= INT32_MIN) { e.priority = (int32_t)pr; e.payload = v; }
            }
            one[s] = e;
        }
        rc = up(r->one, one.data(), one.size() * sizeof(RpStateOne));
    }
    if (rc != AM_OK) { am_replacer_destroy(r); return rc; }
    r->t = RpTables{(const uint64_t*)r->vals_off.p, (const uint32_t*)r->vals.p, (const RpPayload*)r->payloads.p, (const uint8_t*)r->repl.p, min_priority, (const RpStateOne*)r->one.p};
    *out = r;
    return AM_OK;
}

extern "C" void am_replacer_destroy(am_replacer* r)
{
    if (!r) return;
    if (r->session_delete) for (void* p : r->sessions) r->session_delete(p);
    for (DevBuf* d : {&r->vals_off, &r->vals, &r->payloads, &r->repl, &r->one}) d->release();
    delete r;
}

namespace {

struct RpSession {
    DevBuf text[2], offs[2], orig[2], thr[2];
    DevBuf totals; uint64_t* tot_host = nullptr; uint64_t tot_seq = 0;       // the per-pass totals, read back through pinned memory (tot_host[15]: sequence number of the last pass written)
    hipStream_t copy_stream = nullptr; hipEvent_t ev_spliced = nullptr;     // finished texts travel home next to the window scans
    RpFin* fin_host = nullptr; size_t fin_host_cap = 0;                     // pinned
    int pin_meta(size_t bytes)
    {
        if (bytes <= fin_host_cap) return AM_OK;
        if (fin_host) (void)hipHostFree(fin_host);
        fin_host = nullptr; fin_host_cap = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        if (hipHostMalloc((void**)&fin_host, want, hipHostMallocPortable) != hipSuccess) { fin_host = nullptr; return fail(AM_ERR_OOM, "hipHostMalloc failed"); }
        fin_host_cap = want;
        return AM_OK;
    }
    DevBuf recbuf[2];                    // sorted records of the current pass / of the next one (incremental re-scan)
    DevBuf nwin, win_off, wins, wlen, woffs, wtext, wrec, wrec_first, mcount, moff, tile_hay;
    am_batch ws2;                        // workspace of the window scans
    DevBuf rec_first, rec_first2, kept, hs, len_next, len_fin, tiles, act, fin, off_next, off_fin, tile_off, act_idx, fin_idx, scan_tmp, fin_text, fin_meta;      // (rec_first2: the piece-table loop's second ranges buffer -- a pass's merge writes the next pass's ranges)
    am_batch ws;                         // workspace holder for the scans; never owns its text
    DevBuf first_orig, first_thr;
    DevBuf pt_pieces[2], pt_start[2], pt_cnt[2], pt_need, pt_need_off, pt_fin_start, pt_fin_cnt;      // piece-table path
    DevBuf lp_rec, lp_pc, lp_kept, lp_wtext, lp_out, lp_ctrl, lp_cap_r, lp_cap_p, lp_rec_base, lp_pc_base, lp_fin, lp_fin_start, lp_fin_cnt, lp_redo;      // one-kernel loops (am_rplds.hip, am_rploop.hip)
    DevBuf lp_stage[8];                  // device staging of the haystack groups' finished texts on their way to the host
    void* lp_host = nullptr; size_t lp_host_cap = 0;                        // pinned: the loop's per-haystack results, then the materialise tables
    int pin_loop(size_t bytes)
    {
        if (bytes <= lp_host_cap) return AM_OK;
        if (lp_host) (void)hipHostFree(lp_host);
        lp_host = nullptr; lp_host_cap = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        if (hipHostMalloc(&lp_host, want, hipHostMallocPortable) != hipSuccess) { lp_host = nullptr; return fail(AM_ERR_OOM, "hipHostMalloc failed"); }
        lp_host_cap = want;
        return AM_OK;
    }
    DevBuf pf_best, pf_delta, pf_payload, pf_selflag, pf_sidx, pf_cand, pf_sel, pf_keep, pf_kflag, pf_kdelta, pf_kidx, pf_kdpre, pf_tmp;   // record-parallel fold
    size_t device_bytes() const
    {
        size_t n = 0;
        for (const DevBuf* d : {&text[0], &text[1], &recbuf[0], &recbuf[1], &kept, &wins, &wtext, &wrec, &fin_text, &ws.pool, &ws2.pool, &ws.hidx, &ws2.hidx, &pf_cand, &pf_sel, &pf_sidx,
                                &lp_rec, &lp_pc, &lp_kept, &lp_wtext, &lp_stage[0], &lp_stage[1], &lp_stage[2], &lp_stage[3], &lp_stage[4], &lp_stage[5], &lp_stage[6], &lp_stage[7]}) n += d->cap;
        return n;
    }
    ~RpSession()
    {
        for (DevBuf* d : {&text[0], &text[1], &offs[0], &offs[1], &orig[0], &orig[1], &thr[0], &thr[1], &rec_first, &kept, &hs, &len_next, &len_fin,
                          &recbuf[0], &recbuf[1], &nwin, &win_off, &wins, &wlen, &woffs, &wtext, &wrec, &wrec_first, &mcount, &moff, &tile_hay,
                          &totals, &tiles, &act, &fin, &off_next, &off_fin, &tile_off, &act_idx, &fin_idx, &scan_tmp, &fin_text, &fin_meta, &first_orig, &first_thr,
                          &pf_best, &pf_delta, &pf_payload, &pf_selflag, &pf_sidx, &pf_cand, &pf_sel, &pf_keep, &pf_kflag, &pf_kdelta, &pf_kidx, &pf_kdpre, &pf_tmp,
                          &pt_pieces[0], &pt_pieces[1], &pt_start[0], &pt_start[1], &pt_cnt[0], &pt_cnt[1], &pt_need, &pt_need_off, &pt_fin_start, &pt_fin_cnt,
                          &lp_rec, &lp_pc, &lp_kept, &lp_wtext, &lp_out, &lp_ctrl, &lp_cap_r, &lp_cap_p, &lp_rec_base, &lp_pc_base, &lp_fin, &lp_fin_start, &lp_fin_cnt, &lp_redo, &lp_stage[0], &lp_stage[1], &lp_stage[2], &lp_stage[3], &lp_stage[4], &lp_stage[5], &lp_stage[6], &lp_stage[7]}) d->release();
        if (lp_host) (void)hipHostFree(lp_host);
        if (tot_host) (void)hipHostFree(tot_host);
        if (fin_host) (void)hipHostFree(fin_host);
        if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); }
        if (ev_spliced) (void)hipEventDestroy(ev_spliced);
        for (am_batch* w : {&ws, &ws2})
            for (DevBuf* d : {&w->hidx, &w->unit_counts, &w->unit_offsets, &w->scan_tmp, &w->small, &w->hay_counts, &w->flags, &w->unit_first, &w->pool, &w->block_next}) d->release();
    }
};



// prependMatch + makeMatch + removeOverlap of one pass (Replacer.hs:252-274,191-198): one wavefront per haystack, or -- few
// haystacks with very many matches each -- parallel over the records.  Writes kept[], hs[] and the route arrays.
static int rp_fold(RpSession& s, const am_replacer* r, bool ic, const uint8_t* text, const uint64_t* offs, const Record* recs, uint64_t n_rec, const int64_t* thr,
                   uint64_t max_length, const RpRoute& route, uint32_t n_act, hipStream_t st, const uint64_t* rec_first)
{
    const uint64_t n1 = (uint64_t)n_act + 1;
    // one wavefront per haystack, or -- few haystacks with very many matches each -- parallel over the records
        bool par_fold = n_rec > 2048ull * n_act;
        if (cfg::get(cfg::kRpParallelFold) != cfg::kUnset) par_fold = cfg::get(cfg::kRpParallelFold) != 0;        // tests force either path
        if (!par_fold) {
            Prof pr("rp_pass", st);
            HIP_TRY(launch_rp_pass(ic, r->t, text, offs, recs, rec_first, thr,
                                   max_length, (RpKept*)s.kept.p, (RpHay*)s.hs.p, route, n_act, 0u, st));
        } else {
            Prof pr("rp_pass", st);
            const uint64_t nb = n_rec + 2;
            AM_TRY(s.pf_best.ensure(n1 * 8)); AM_TRY(s.pf_delta.ensure(n1 * 8)); AM_TRY(s.pf_payload.ensure(n1 * 4));
            AM_TRY(s.pf_selflag.ensure(nb * 4)); AM_TRY(s.pf_sidx.ensure(nb * 8)); AM_TRY(s.pf_cand.ensure(nb * sizeof(RpSel))); AM_TRY(s.pf_sel.ensure(nb * sizeof(RpSel)));
            AM_TRY(s.pf_keep.ensure(nb * 4)); AM_TRY(s.pf_kflag.ensure(nb * 4)); AM_TRY(s.pf_kdelta.ensure(nb * 8)); AM_TRY(s.pf_kidx.ensure(nb * 8)); AM_TRY(s.pf_kdpre.ensure(nb * 8));
            size_t t32b = 0, t64b = 0;
            if (scan_temp_bytes(nb, &t32b) != hipSuccess || scan64_temp_bytes(nb, &t64b) != hipSuccess) return fail(AM_ERR_HIP, "scan sizing failed");
            AM_TRY(s.pf_tmp.ensure(std::max(t32b, t64b) + 16));
            const size_t ptmp = s.pf_tmp.cap - 16;
            HIP_TRY(hipMemsetAsync(s.pf_delta.p, 0, n1 * 8, st)); HIP_TRY(hipMemsetAsync(s.pf_payload.p, 0, n1 * 4, st));
            HIP_TRY(hipMemsetAsync(s.pf_kflag.p, 0, nb * 4, st)); HIP_TRY(hipMemsetAsync(s.pf_kdelta.p, 0, nb * 8, st)); HIP_TRY(hipMemsetAsync(s.pf_keep.p, 0, nb * 4, st));
            HIP_TRY(launch_rpp_best(r->t, recs, n_rec, thr, (int64_t*)s.pf_best.p, n_act, st));
            HIP_TRY(launch_rpp_select(ic, r->t, text, offs, recs, n_rec, (const int64_t*)s.pf_best.p,
                                      (uint32_t*)s.pf_selflag.p, (RpSel*)s.pf_cand.p, (int64_t*)s.pf_delta.p, (uint32_t*)s.pf_payload.p, st));
            HIP_TRY(launch_scan(s.pf_tmp.p, ptmp, (const uint32_t*)s.pf_selflag.p, (uint64_t*)s.pf_sidx.p, n_rec + 1, st));
            const uint64_t* n_sel_dev = (const uint64_t*)s.pf_sidx.p + n_rec;
            HIP_TRY(launch_rpp_compact((const uint32_t*)s.pf_selflag.p, (const uint64_t*)s.pf_sidx.p, (const RpSel*)s.pf_cand.p, n_rec, (RpSel*)s.pf_sel.p, st));
            HIP_TRY(launch_rpp_overlaps((const RpSel*)s.pf_sel.p, n_sel_dev, n_rec, (uint32_t*)s.pf_keep.p, st));
            HIP_TRY(launch_rpp_kflags((const RpSel*)s.pf_sel.p, n_sel_dev, n_rec, (const uint32_t*)s.pf_keep.p, r->t, (const uint32_t*)s.pf_payload.p,
                                      (uint32_t*)s.pf_kflag.p, (uint64_t*)s.pf_kdelta.p, st));
            HIP_TRY(launch_scan(s.pf_tmp.p, ptmp, (const uint32_t*)s.pf_kflag.p, (uint64_t*)s.pf_kidx.p, n_rec + 2, st));
            HIP_TRY(launch_scan64(s.pf_tmp.p, ptmp, (const uint64_t*)s.pf_kdelta.p, (uint64_t*)s.pf_kdpre.p, n_rec + 2, st));
            HIP_TRY(launch_rpp_finish(r->t, (const RpSel*)s.pf_sel.p, n_sel_dev, n_rec, (const uint32_t*)s.pf_kflag.p, (const uint64_t*)s.pf_kidx.p,
                                      (const uint64_t*)s.pf_kdpre.p, (const uint64_t*)s.pf_sidx.p, offs, rec_first, (const int64_t*)s.pf_best.p,
                                      (const int64_t*)s.pf_delta.p, (const uint32_t*)s.pf_payload.p, max_length, (RpKept*)s.kept.p, (RpHay*)s.hs.p, route, n_act, st));
        }
    return AM_OK;
}

// The same loop with the text of the active haystacks kept as PIECE TABLES (am_replace.hip): no pass rewrites a text; bytes
// move into the re-scanned windows and, once per haystack, into the result.  CaseSensitive replacers on the suffix-filter
// route (the incremental re-scan is part of the design: after the first pass only windows are scanned).
int replacer_run_pt(const am_replacer* r, const am_batch* in, uint64_t max_length, am_replaced* res, const Flavor* flavor)
{
    const uint32_t n_hay = in->n_hay;
    ON_DEVICE(in->dev);
    hipStream_t st; AM_TRY(get_stream(in->dev, &st));
    RpSession* sp = nullptr;
    { std::lock_guard<std::mutex> lk(r->session_mu); if (!r->sessions.empty()) { sp = static_cast<RpSession*>(r->sessions.back()); r->sessions.pop_back(); } }
    if (!sp) sp = new RpSession();
    struct Return {
        const am_replacer* r; RpSession* sp;
        ~Return()
        {
            if (sp->copy_stream) (void)hipStreamSynchronize(sp->copy_stream);
            if (sp->device_bytes() > (2048ull << 20)) { delete sp; return; }      // keep workspaces of up to 2 GiB between calls
            std::vector<RpSession*> doomed;
            { std::lock_guard<std::mutex> lk(r->session_mu);
              const_cast<am_replacer*>(r)->session_delete = [](void* p) { delete static_cast<RpSession*>(p); };
              r->sessions.push_back(sp);
              // at most 8 cached workspaces and at most 4 GiB of device memory in all of them (each is below 2 GiB): the oldest go first
              for (;;) {
                  size_t held = 0;
                  for (void* q : r->sessions) held += static_cast<RpSession*>(q)->device_bytes();
                  if (r->sessions.size() <= 1 || (r->sessions.size() <= 8 && held <= (4096ull << 20))) break;
                  doomed.push_back(static_cast<RpSession*>(r->sessions.front()));
                  r->sessions.erase(r->sessions.begin());
              } }
            for (RpSession* q : doomed) delete q;
        }
    } give_back{r, sp};
    RpSession& s = *sp;
    AM_TRY(s.totals.ensure(128));
    if (!s.tot_host) {
        if (hipHostMalloc((void**)&s.tot_host, 128, hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess) { s.tot_host = nullptr; return fail(AM_ERR_OOM, "hipHostMalloc failed"); }
        std::memset(s.tot_host, 0, 128);                  // (fine-grained: a device store is visible to the host while the kernel is still running)
    }
    if (!s.copy_stream && (hipStreamCreateWithFlags(&s.copy_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&s.ev_spliced, hipEventDisableTiming) != hipSuccess))
        return fail(AM_ERR_HIP, "could not create the copy stream");
    const uint8_t* base_text = (const uint8_t*)in->d_text;               // never modified: every text piece points into it
    const uint64_t* cur_offs = in->d_offsets;                            // logical offsets of the active haystacks (lengths only after pass 0)
    uint32_t n_act = n_hay;
    int nxt = 0;
    {
        std::vector<uint32_t> o(n_hay); std::vector<int64_t> t(n_hay, 1);      // initialThreshold = 1 (Replacer.hs:211)
        for (uint32_t i = 0; i < n_hay; i++) o[i] = i;
        AM_TRY(s.first_orig.ensure(n_hay * sizeof(uint32_t))); AM_TRY(s.first_thr.ensure(n_hay * sizeof(int64_t)));
        HIP_TRY(hipMemcpyAsync(s.first_orig.p, o.data(), n_hay * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(s.first_thr.p, t.data(), n_hay * sizeof(int64_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    const uint32_t* cur_orig = (const uint32_t*)s.first_orig.p;
    const int64_t* cur_thr = (const int64_t*)s.first_thr.p;
    const uint32_t ov = 4u * (flavor->h.max_needle_cps ? flavor->h.max_needle_cps : 1u) + 4u;
    // piece lists of pass 0: one piece per haystack
    int cur_pt = 0;
    AM_TRY(s.pt_pieces[0].ensure(((size_t)n_hay * 2 + 2) * sizeof(RpPiece)));
    AM_TRY(s.pt_start[0].ensure(((size_t)n_hay + 1) * 8)); AM_TRY(s.pt_cnt[0].ensure(((size_t)n_hay + 1) * 4));
    HIP_TRY(launch_pt_init(in->d_offsets, n_hay, (RpPiece*)s.pt_pieces[0].p, (uint64_t*)s.pt_start[0].p, (uint32_t*)s.pt_cnt[0].p, st));
    // pass 0 scans the caller's batch; afterwards the records come from the window scans + the shifted old records
    uint64_t n_rec = 0;                                   // records of the current pass: exact when n_rec_dev == nullptr, else an upper bound ...
    const uint64_t* n_rec_dev = nullptr;                  // ... and the exact count is still on the device
    int cur_rec = 0;
    {
        res->scanned += in->total;
        s.ws.dev = in->dev; s.ws.d_text = in->d_text; s.ws.d_offsets = in->d_offsets; s.ws.owns = false; s.ws.total = in->total; s.ws.n_hay = n_hay;
        AM_TRY(finish_batch(&s.ws));
        auto sink = [&](uint64_t n, Record** ptr) -> int { AM_TRY(s.recbuf[0].ensure(n * sizeof(Record))); *ptr = (Record*)s.recbuf[0].p; return AM_OK; };
        AM_TRY(run_records(r->a, r->case_mode, &s.ws, sink, &n_rec));
    }
    const bool trace = cfg::on(cfg::kRpTrace);
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_a = 0, t_b = 0, t_c = 0, t_sync = 0;
    // finished haystacks of the previous pass: their bytes are on their way home on the copy stream; the host looks at the list after
    // the next pass's (only) synchronisation
    uint64_t prev_n_fin = 0, prev_total_fin = 0; uint8_t* prev_home = nullptr;
    hipEvent_t ev_copied = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&ev_copied, hipEventDisableTiming));
    struct EvGuard { hipEvent_t e; ~EvGuard() { (void)hipEventDestroy(e); } } ev_guard{ev_copied};
    bool copies_pending = false, ev_copied_used = false;
    auto finished_home = [&]() -> int {
        if (!copies_pending) return AM_OK;
        HIP_TRY(hipStreamSynchronize(s.copy_stream));
        copies_pending = false;
        for (uint64_t i = 0; i < prev_n_fin; i++) {
            const RpFin& f = s.fin_host[i];
            if (f.orig >= n_hay || f.off + f.len > prev_total_fin) return fail(AM_ERR_HIP, "replacer pass produced inconsistent metadata (internal error)");
            if (f.status == kRpNothing) res->just[f.orig] = 0;
            else res->text[f.orig] = am_replaced::Item{prev_home + f.off, (size_t)f.len};
        }
        return AM_OK;
    };

    int cur_rf = 0; bool have_ranges = false;           // (see the ranges buffers below)
    while (n_act > 0) {
        double t0 = now();
        res->passes++;
        DevBuf& records = s.recbuf[cur_rec];
        const uint64_t n1 = (uint64_t)n_act + 1;
        // the record-parallel fold needs the exact count on the host: fetch it when that regime is possible
        if (n_rec_dev && (n_rec > 2048ull * n_act || cfg::get(cfg::kRpParallelFold) != cfg::kUnset)) {
            HIP_TRY(hipMemcpyAsync(&s.tot_host[9], n_rec_dev, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            n_rec = s.tot_host[9]; n_rec_dev = nullptr;
        }
        // record ranges of the haystacks: two buffers that take turns -- the merge at the end of a pass leaves the offsets of the records it
        // writes (per haystack of the next pass) in the other one, which ARE the next pass's ranges: no search then
        DevBuf& rfb = cur_rf ? s.rec_first2 : s.rec_first; DevBuf& rfb_next = cur_rf ? s.rec_first : s.rec_first2;
        AM_TRY(rfb.ensure(n1 * 8)); AM_TRY(s.kept.ensure((n_rec + 1) * sizeof(RpKept))); AM_TRY(s.hs.ensure(n1 * sizeof(RpHay)));
        AM_TRY(s.len_next.ensure(n1 * 8)); AM_TRY(s.len_fin.ensure(n1 * 8)); AM_TRY(s.tiles.ensure(n1 * 4)); AM_TRY(s.act.ensure(n1 * 4)); AM_TRY(s.fin.ensure(n1 * 4));
        AM_TRY(s.off_next.ensure(n1 * 8)); AM_TRY(s.off_fin.ensure(n1 * 8)); AM_TRY(s.tile_off.ensure(n1 * 8)); AM_TRY(s.act_idx.ensure(n1 * 8)); AM_TRY(s.fin_idx.ensure(n1 * 8));
        AM_TRY(s.nwin.ensure(n1 * 4)); AM_TRY(s.win_off.ensure(n1 * 8)); AM_TRY(s.pt_need.ensure(n1 * 4)); AM_TRY(s.pt_need_off.ensure(n1 * 8));
        AM_TRY(s.wins.ensure((n_rec + 1) * sizeof(RpWin))); AM_TRY(s.wlen.ensure((n_rec + 2) * 4)); AM_TRY(s.woffs.ensure((n_rec + 2) * 8));
        size_t t32 = 0, t64 = 0, tw = 0;
        if (scan_temp_bytes(n1, &t32) != hipSuccess || scan64_temp_bytes(n1, &t64) != hipSuccess || scan_temp_bytes(n_rec + 2, &tw) != hipSuccess) return fail(AM_ERR_HIP, "scan sizing failed");
        AM_TRY(s.scan_tmp.ensure(std::max(std::max(t32, t64), tw) + 16));
        const size_t tmp2 = s.scan_tmp.cap - 16;
        AM_TRY(records.ensure(sizeof(Record)));
        RpRoute route{(uint64_t*)s.len_next.p, (uint64_t*)s.len_fin.p, (uint32_t*)s.tiles.p, (uint32_t*)s.act.p, (uint32_t*)s.fin.p};
        RpRouted rt{(const uint64_t*)s.off_next.p, (const uint64_t*)s.off_fin.p, (const uint64_t*)s.tile_off.p, (const uint64_t*)s.act_idx.p, (const uint64_t*)s.fin_idx.p};
        // the per-haystack fold also finds its record range and writes the piece / window counts (one dispatch instead of three in the pass's
        // chain); the record-parallel fold keeps the separate launches
        bool par_fold = (n_rec_dev ? 0 : n_rec) > 2048ull * n_act;
        if (cfg::get(cfg::kRpParallelFold) != cfg::kUnset) par_fold = cfg::get(cfg::kRpParallelFold) != 0;
        const bool no_fuse = cfg::on(cfg::kRpNoFuse);                                             // A/B
        const bool fused = !par_fold && !no_fuse;
        if (fused) {
            Prof pr("rp_pass", st);
            const RpFused fu{have_ranges ? nullptr : (uint64_t*)rfb.p, n_rec_dev ? 0 : n_rec, n_rec_dev, (const uint32_t*)s.pt_cnt[cur_pt].p, (uint32_t*)s.pt_need.p, (uint32_t*)s.nwin.p};
            HIP_TRY(launch_rp_pass(false, r->t, base_text, cur_offs, (const Record*)records.p, (const uint64_t*)rfb.p, cur_thr, max_length, (RpKept*)s.kept.p, (RpHay*)s.hs.p, route, n_act, 0u, st, &fu));
        } else {
            { Prof pr("rp_ranges", st);
              if (n_rec_dev) HIP_TRY(launch_rp_ranges_dev((const Record*)records.p, n_rec_dev, (uint64_t*)rfb.p, route, n_act, st));
              else HIP_TRY(launch_rp_ranges((const Record*)records.p, n_rec, (uint64_t*)rfb.p, route, n_act, st)); }
            AM_TRY(rp_fold(s, r, false, base_text, cur_offs, (const Record*)records.p, n_rec_dev ? 0 : n_rec, cur_thr, max_length, route, n_act, st, (const uint64_t*)rfb.p));
        }
        const bool small = n1 <= (1u << 18);
        { Prof pr("rp_scans", st);
          if (!fused) HIP_TRY(launch_pt_count((const RpHay*)s.hs.p, (const uint32_t*)s.pt_cnt[cur_pt].p, n_act, (uint32_t*)s.pt_need.p, (uint32_t*)s.nwin.p, st));
          if (small) {
              ScanJobs jobs{};
              jobs.j[0] = ScanJob{nullptr, route.len_next, (uint64_t*)s.off_next.p, n1, nullptr};
              jobs.j[1] = ScanJob{nullptr, route.len_fin, (uint64_t*)s.off_fin.p, n1, nullptr};
              jobs.j[2] = ScanJob{(const uint32_t*)s.pt_need.p, nullptr, (uint64_t*)s.pt_need_off.p, n1, nullptr};
              jobs.j[3] = ScanJob{route.act, nullptr, (uint64_t*)s.act_idx.p, n1, nullptr};
              jobs.j[4] = ScanJob{route.fin, nullptr, (uint64_t*)s.fin_idx.p, n1, nullptr};
              jobs.j[5] = ScanJob{(const uint32_t*)s.nwin.p, nullptr, (uint64_t*)s.win_off.p, n1, nullptr};
              jobs.n_jobs = 6;
              HIP_TRY(launch_scan_jobs(jobs, st));
          } else {
              HIP_TRY(launch_scan64(s.scan_tmp.p, tmp2, route.len_next, (uint64_t*)s.off_next.p, n1, st));
              HIP_TRY(launch_scan64(s.scan_tmp.p, tmp2, route.len_fin, (uint64_t*)s.off_fin.p, n1, st));
              HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, (const uint32_t*)s.pt_need.p, (uint64_t*)s.pt_need_off.p, n1, st));
              HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, route.act, (uint64_t*)s.act_idx.p, n1, st));
              HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, route.fin, (uint64_t*)s.fin_idx.p, n1, st));
              HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, (const uint32_t*)s.nwin.p, (uint64_t*)s.win_off.p, n1, st));
          } }
        uint64_t woffs_last = n_rec;
        { Prof pr("rp_windows", st);
          if (!small) HIP_TRY(hipMemsetAsync(s.wlen.p, 0, (n_rec + 2) * 4, st));
          HIP_TRY(launch_rp_win_meta(r->t, rt, (const RpHay*)s.hs.p, (const uint64_t*)rfb.p, (const RpKept*)s.kept.p,
                                     (const uint64_t*)s.win_off.p, ov, (RpWin*)s.wins.p, (uint32_t*)s.wlen.p, n_act, st, true));
          if (small) {
              ScanJobs jobs{};
              jobs.j[0] = ScanJob{(const uint32_t*)s.wlen.p, nullptr, (uint64_t*)s.woffs.p, 1, (const uint64_t*)s.win_off.p + n_act};
              jobs.n_jobs = 1;
              HIP_TRY(launch_scan_jobs(jobs, st));
              woffs_last = ~0ull;
          } else HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, (const uint32_t*)s.wlen.p, (uint64_t*)s.woffs.p, n_rec + 1, st)); }
        // the pass's ONE synchronisation: bytes of next text, bytes of finished text, -, haystacks still active, haystacks finished,
        // windows, window bytes, piece entries, and the exact record count of this pass when it was still on the device
        // (k_rp_totals writes straight into the pinned host block -- 80 bytes of posted PCIe writes -- instead of into device memory that a
        // 16-us copy dispatch would then move)
        // ... and the host waits for the LAST word of that block (a sequence number the kernel stores after a system-scope fence) by spinning on
        // it for a while before it falls back to hipStreamSynchronize: the blocking wait's wake-up cost 20-30 us of every pass's ~250
        const uint64_t seq = ++s.tot_seq;
        HIP_TRY(launch_rp_totals(rt, n_act, (const uint64_t*)s.win_off.p, (const uint64_t*)s.woffs.p, woffs_last, s.tot_host, st,
                                 (const uint64_t*)s.pt_need_off.p + n_act, n_rec_dev, seq));
        const double t_s0 = now();
        {
            const bool no_spin = cfg::on(cfg::kRpNoSpin);                                 // A/B
            bool seen = false;
            if (!no_spin) {
                const double give_up = t_s0 + 2e-3;
                for (uint32_t it = 0; !seen; it++) {
                    seen = __atomic_load_n(&s.tot_host[15], __ATOMIC_ACQUIRE) == seq;
                    if (!seen && (it & 1023u) == 1023u && now() > give_up) break;
                }
            }
            if (!seen) HIP_TRY(hipStreamSynchronize(st));
        }
        if (trace) t_sync += now() - t_s0;
        const uint64_t* tot = s.tot_host;
        const uint64_t total_next = tot[0], total_fin = tot[1], n_next = tot[3], n_fin = tot[4], n_win = tot[5], total_w = tot[6], n_pieces = tot[8];
        if (n_rec_dev) { n_rec = tot[9]; n_rec_dev = nullptr; }
        AM_TRY(finished_home());                              // the previous pass's finished haystacks (their copies have had a whole pass)
        t_a += now() - t0; t0 = now();
        if (n_win >= 0xFFFFFFF0ull) return fail(AM_ERR_UNSUPPORTED, "too many replacements in one pass; split the batch");
        // ---- the next pass's piece lists; finished haystacks are materialised and go home
        AM_TRY(s.offs[nxt].ensure((n_next + 1) * 8)); AM_TRY(s.orig[nxt].ensure((n_next + 1) * 4)); AM_TRY(s.thr[nxt].ensure((n_next + 1) * 8));
        AM_TRY(s.fin_text.ensure(total_fin + 16)); AM_TRY(s.fin_meta.ensure((n_fin + 1) * sizeof(RpFin)));
        AM_TRY(s.pt_pieces[cur_pt ^ 1].ensure((n_pieces + 2) * sizeof(RpPiece)));
        AM_TRY(s.pt_start[cur_pt ^ 1].ensure((n_next + 1) * 8)); AM_TRY(s.pt_cnt[cur_pt ^ 1].ensure((n_next + 1) * 4));
        AM_TRY(s.pt_fin_start.ensure((n_fin + 1) * 8)); AM_TRY(s.pt_fin_cnt.ensure((n_fin + 1) * 4));
        if (ev_copied_used) HIP_TRY(hipStreamWaitEvent(st, ev_copied, 0));      // the previous pass's finished texts are written and their metadata has left fin_meta
        { Prof pr("rp_route", st);
          HIP_TRY(launch_rp_route((const RpHay*)s.hs.p, rt, cur_orig, n_act, (uint64_t*)s.offs[nxt].p, (uint32_t*)s.orig[nxt].p, (int64_t*)s.thr[nxt].p, (RpFin*)s.fin_meta.p, st)); }
        { Prof pr("pt_build", st);
          HIP_TRY(launch_pt_build(r->t, (const RpHay*)s.hs.p, (const uint64_t*)rfb.p, (const RpKept*)s.kept.p, (const RpPiece*)s.pt_pieces[cur_pt].p,
                                  (const uint64_t*)s.pt_start[cur_pt].p, (const uint32_t*)s.pt_cnt[cur_pt].p, (const uint64_t*)s.pt_need_off.p, rt, n_act,
                                  (RpPiece*)s.pt_pieces[cur_pt ^ 1].p, (uint64_t*)s.pt_start[cur_pt ^ 1].p, (uint32_t*)s.pt_cnt[cur_pt ^ 1].p,
                                  (uint64_t*)s.pt_fin_start.p, (uint32_t*)s.pt_fin_cnt.p, st)); }
        res->spliced += total_fin;
        if (n_fin) {
            uint8_t* home = nullptr;
            if (total_fin) AM_TRY(res->room((size_t)total_fin, &home));
            AM_TRY(s.pin_meta((n_fin + 1) * sizeof(RpFin)));
            // the finished texts are written out on the COPY stream (64 us of a 270-us pass that nothing of the next pass waits for): it starts when
            // this pass's piece lists and metadata are complete; what it reads is not touched before the next pass's host-side look at the copy
            // stream (finished_home, after the totals) -- and the next rp_route waits for the event as well
            const bool mat_main = cfg::on(cfg::kRpMatMain);                              // A/B: on the pass's own stream, as before
            hipStream_t mst = mat_main ? st : s.copy_stream;
            if (!mat_main) { HIP_TRY(hipEventRecord(s.ev_spliced, st)); HIP_TRY(hipStreamWaitEvent(s.copy_stream, s.ev_spliced, 0)); }
            { Prof pr("pt_materialise", mst);
              HIP_TRY(launch_pt_materialise((const RpPiece*)s.pt_pieces[cur_pt ^ 1].p, (const uint64_t*)s.pt_fin_start.p, (const uint32_t*)s.pt_fin_cnt.p, (const RpFin*)s.fin_meta.p,
                                            (uint32_t)n_fin, base_text, r->t.repl, res->dev >= 0 && total_fin ? home : (uint8_t*)s.fin_text.p, mst)); }
            if (mat_main) { HIP_TRY(hipEventRecord(s.ev_spliced, st)); HIP_TRY(hipStreamWaitEvent(s.copy_stream, s.ev_spliced, 0)); }
            if (total_fin && res->dev < 0) HIP_TRY(hipMemcpyAsync(home, s.fin_text.p, total_fin, hipMemcpyDeviceToHost, s.copy_stream));
            HIP_TRY(hipMemcpyAsync(s.fin_host, s.fin_meta.p, n_fin * sizeof(RpFin), hipMemcpyDeviceToHost, s.copy_stream));
            HIP_TRY(hipEventRecord(ev_copied, s.copy_stream));
            ev_copied_used = true;
            prev_n_fin = n_fin; prev_total_fin = total_fin; prev_home = home; copies_pending = true;
        }
        t_b += now() - t0; t0 = now();
        // ---- the next pass's records
        uint64_t next_n_rec = 0; const uint64_t* next_n_rec_dev = nullptr; bool next_have_ranges = false;
        if (n_next > 0) {
            DevBuf& next_records = s.recbuf[cur_rec ^ 1];
            if (total_w > total_next) {
                // tiny texts: the windows would be larger than the texts themselves -- materialise the next texts and scan them whole
                AM_TRY(s.text[0].ensure(padded_text(total_next)));
                HIP_TRY(launch_pt_materialise_next((const RpPiece*)s.pt_pieces[cur_pt ^ 1].p, (const uint64_t*)s.pt_start[cur_pt ^ 1].p, (const uint32_t*)s.pt_cnt[cur_pt ^ 1].p,
                                                   (const uint64_t*)s.offs[nxt].p, (uint32_t)n_next, base_text, r->t.repl, (uint8_t*)s.text[0].p, st));
                HIP_TRY(hipMemsetAsync((uint8_t*)s.text[0].p + total_next, 0, padded_text(total_next) - (size_t)total_next, st));
                s.ws.dev = in->dev; s.ws.d_text = s.text[0].p; s.ws.d_offsets = (uint64_t*)s.offs[nxt].p; s.ws.owns = false; s.ws.total = total_next; s.ws.n_hay = (uint32_t)n_next;
                AM_TRY(finish_batch(&s.ws));
                auto sink = [&](uint64_t n, Record** ptr) -> int { AM_TRY(next_records.ensure(n * sizeof(Record))); *ptr = (Record*)next_records.p; return AM_OK; };
                AM_TRY(run_records(r->a, r->case_mode, &s.ws, sink, &next_n_rec));
                res->scanned += total_next;
            } else {
                // windows around the replacements (gathered from the new piece lists) + the shifted old records; no host round trip when the
                // worst-case record pool of the window scan stays small
                const bool lean = total_w <= (64ull << 20);
                uint64_t n_wrec = 0; const uint64_t* n_wrec_dev = nullptr;
                AM_TRY(s.wtext.ensure(padded_text(total_w)));
                AM_TRY(s.wrec.ensure(((lean ? total_w : 0) + 1) * sizeof(Record)));
                if (n_win > 0 && total_w > 0) {
                    { Prof pr("rp_windows", st);
                      HIP_TRY(launch_pt_win_copy((const RpWin*)s.wins.p, (const uint64_t*)s.woffs.p, (const RpPiece*)s.pt_pieces[cur_pt ^ 1].p, (const uint64_t*)s.pt_start[cur_pt ^ 1].p,
                                                 (const uint32_t*)s.pt_cnt[cur_pt ^ 1].p, base_text, r->t.repl, (uint8_t*)s.wtext.p, n_win, total_w, padded_text(total_w), st)); }
                    s.ws2.dev = in->dev; s.ws2.d_text = s.wtext.p; s.ws2.d_offsets = (uint64_t*)s.woffs.p; s.ws2.owns = false; s.ws2.total = total_w; s.ws2.n_hay = (uint32_t)n_win;
                    AM_TRY(finish_batch(&s.ws2));
                    if (lean) AM_TRY(run_records_async(r->a, r->case_mode, &s.ws2, (Record*)s.wrec.p, &n_wrec_dev, st));
                    else {
                        auto wsink = [&](uint64_t n, Record** ptr) -> int { AM_TRY(s.wrec.ensure(n * sizeof(Record))); *ptr = (Record*)s.wrec.p; return AM_OK; };
                        AM_TRY(run_records(r->a, r->case_mode, &s.ws2, wsink, &n_wrec));
                    }
                    res->scanned += total_w;
                }
                const uint64_t wrec_bound = n_wrec_dev ? total_w : n_wrec;
                AM_TRY(s.wrec_first.ensure((n_win + 2) * 8)); AM_TRY(s.mcount.ensure((n_next + 1) * 4)); AM_TRY(rfb_next.ensure((n_next + 1) * 8));
                AM_TRY(next_records.ensure((n_rec + wrec_bound + 1) * sizeof(Record)));
                Prof pr("rp_merge", st);
                if (n_wrec_dev) HIP_TRY(launch_rp_ranges_dev((const Record*)s.wrec.p, n_wrec_dev, (uint64_t*)s.wrec_first.p, RpRoute{nullptr, nullptr, nullptr, nullptr, nullptr}, (uint32_t)n_win, st));
                else HIP_TRY(launch_rp_ranges((const Record*)s.wrec.p, n_wrec, (uint64_t*)s.wrec_first.p, RpRoute{nullptr, nullptr, nullptr, nullptr, nullptr}, (uint32_t)n_win, st));
                HIP_TRY(launch_rp_merge(false, (const Record*)records.p, (const uint64_t*)rfb.p, (const RpKept*)s.kept.p, (const RpHay*)s.hs.p, cur_offs, rt,
                                        (const uint64_t*)s.win_off.p, (const RpWin*)s.wins.p, (const Record*)s.wrec.p, (const uint64_t*)s.wrec_first.p, ov, n_act,
                                        (uint32_t*)s.mcount.p, nullptr, nullptr, st));
                if (n_next + 1 <= (1u << 18)) {
                    ScanJobs jobs{};
                    jobs.j[0] = ScanJob{(const uint32_t*)s.mcount.p, nullptr, (uint64_t*)rfb_next.p, n_next + 1, nullptr};
                    jobs.n_jobs = 1;
                    HIP_TRY(launch_scan_jobs(jobs, st));
                } else HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, (const uint32_t*)s.mcount.p, (uint64_t*)rfb_next.p, n_next + 1, st));
                HIP_TRY(launch_rp_merge(true, (const Record*)records.p, (const uint64_t*)rfb.p, (const RpKept*)s.kept.p, (const RpHay*)s.hs.p, cur_offs, rt,
                                        (const uint64_t*)s.win_off.p, (const RpWin*)s.wins.p, (const Record*)s.wrec.p, (const uint64_t*)s.wrec_first.p, ov, n_act,
                                        (uint32_t*)s.mcount.p, (const uint64_t*)rfb_next.p, (Record*)next_records.p, st));
                next_n_rec = n_rec + wrec_bound;                 // an upper bound; the exact count is read with the next pass's totals
                next_n_rec_dev = (const uint64_t*)rfb_next.p + n_next;
                next_have_ranges = true;
            }
        }
        t_c += now() - t0;
        if (trace && cfg::get(cfg::kRpTrace) == 2)
            std::fprintf(stderr, "[am_replacer pt pass %u] active %u -> %llu, finished %llu, records <= %llu, windows %llu (%llu B), next text %llu B\n", (unsigned)res->passes, n_act,
                         (unsigned long long)n_next, (unsigned long long)n_fin, (unsigned long long)n_rec, (unsigned long long)n_win, (unsigned long long)total_w, (unsigned long long)total_next);
        cur_rec ^= 1; cur_pt ^= 1; n_rec = next_n_rec; n_rec_dev = next_n_rec_dev;
        const bool no_reuse = cfg::on(cfg::kRpNoRangeReuse);                              // A/B
        have_ranges = next_have_ranges && !no_reuse; cur_rf ^= 1;
        cur_offs = (const uint64_t*)s.offs[nxt].p; cur_orig = (const uint32_t*)s.orig[nxt].p; cur_thr = (const int64_t*)s.thr[nxt].p;
        n_act = (uint32_t)n_next; nxt ^= 1;
    }
    HIP_TRY(hipStreamSynchronize(st));
    AM_TRY(finished_home());
    if (trace) std::fprintf(stderr, "[am_replacer pt] fold+scans %.1f ms (of which waiting for the device %.1f), pieces+materialise %.1f ms, windows+merge %.1f ms\n", t_a * 1e3, t_sync * 1e3, t_b * 1e3, t_c * 1e3);
    return AM_OK;
}

// Replacer.hs:203-242 runWithLimit for every haystack of `in`, all passes on the device.
int replacer_run(const am_replacer* r, const am_batch* in, uint64_t max_length, am_replaced* res)
{
    const uint32_t n_hay = in->n_hay;
    res->text.assign(n_hay, am_replaced::Item());
    res->just.assign(n_hay, 1);
    if (n_hay == 0) return AM_OK;
    if (in->dev != r->a->dev) return fail(AM_ERR_INVALID, "replacer and batch live on different devices");
    {
        // CaseSensitive replacers on the suffix-filter route keep the texts as piece tables (AM_RP_SPLICE=1: the splicing loop, for A/B and tests)
        const Flavor* fl = nullptr;
        AM_TRY(prepare(r->a, r->case_mode, &fl));
        // ... when the batch is made of many documents: the piece-table kernels give a haystack to ONE wavefront, the splicing loop
        // cuts every text into 16-KiB tiles.  One 1-MB document with half a million replacements per pass: 472 ms vs 90 ms (measured).
        const bool many_documents = n_hay >= 64 && in->total / n_hay <= (1ull << 20);
        const bool pt = r->case_mode == AM_CASE_SENSITIVE && fl->h.sf_enabled && fl->h.root_vlen == 0 && r->a->kernel_pref != 1 &&
                        !cfg::on(cfg::kRpFullScans) && !cfg::on(cfg::kRpSplice) && (many_documents || cfg::on(cfg::kRpPieces)) &&
                        n_hay < (1u << 24) && in->total < (1ull << 40);        // RpWin::src_abs packs (haystack index << 40 | start): beyond that the splicing loop runs
        if (pt) return replacer_run_pt(r, in, max_length, res, fl);
    }
    ON_DEVICE(in->dev);
    hipStream_t st; AM_TRY(get_stream(in->dev, &st));
    // take the replacer's cached workspace (or make one); it goes back at the end unless it has grown large
    RpSession* sp = nullptr;
    { std::lock_guard<std::mutex> lk(r->session_mu); if (!r->sessions.empty()) { sp = static_cast<RpSession*>(r->sessions.back()); r->sessions.pop_back(); } }
    if (!sp) sp = new RpSession();
    struct Return {
        const am_replacer* r; RpSession* sp;
        ~Return()
        {
            if (sp->copy_stream) (void)hipStreamSynchronize(sp->copy_stream);
            if (sp->device_bytes() > (2048ull << 20)) { delete sp; return; }      // keep workspaces of up to 2 GiB between calls
            std::vector<RpSession*> doomed;
            { std::lock_guard<std::mutex> lk(r->session_mu);
              const_cast<am_replacer*>(r)->session_delete = [](void* p) { delete static_cast<RpSession*>(p); };
              r->sessions.push_back(sp);
              // at most 8 cached workspaces and at most 4 GiB of device memory in all of them (each is below 2 GiB): the oldest go first
              for (;;) {
                  size_t held = 0;
                  for (void* q : r->sessions) held += static_cast<RpSession*>(q)->device_bytes();
                  if (r->sessions.size() <= 1 || (r->sessions.size() <= 8 && held <= (4096ull << 20))) break;
                  doomed.push_back(static_cast<RpSession*>(r->sessions.front()));
                  r->sessions.erase(r->sessions.begin());
              } }
            for (RpSession* q : doomed) delete q;
        }
    } give_back{r, sp};
    RpSession& s = *sp;
    AM_TRY(s.totals.ensure(128));
    if (!s.tot_host) {
        if (hipHostMalloc((void**)&s.tot_host, 128, hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess) { s.tot_host = nullptr; return fail(AM_ERR_OOM, "hipHostMalloc failed"); }
        std::memset(s.tot_host, 0, 128);                  // (fine-grained: a device store is visible to the host while the kernel is still running)
    }
    // pass 0 reads the caller's batch in place; afterwards the text ping-pongs between s.text[0] and s.text[1]
    const uint8_t* cur_text = (const uint8_t*)in->d_text;
    const uint64_t* cur_offs = in->d_offsets;
    uint64_t total = in->total;
    uint32_t n_act = n_hay;
    int nxt = 0;
    DevBuf& first_orig = s.first_orig; DevBuf& first_thr = s.first_thr;
    {
        std::vector<uint32_t> o(n_hay); std::vector<int64_t> t(n_hay, 1);      // initialThreshold = 1 (Replacer.hs:211)
        for (uint32_t i = 0; i < n_hay; i++) o[i] = i;
        AM_TRY(first_orig.ensure(n_hay * sizeof(uint32_t))); AM_TRY(first_thr.ensure(n_hay * sizeof(int64_t)));
        HIP_TRY(hipMemcpy(first_orig.p, o.data(), n_hay * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(first_thr.p, t.data(), n_hay * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    const uint32_t* cur_orig = (const uint32_t*)first_orig.p;
    const int64_t* cur_thr = (const int64_t*)first_thr.p;
    if (!s.copy_stream && (hipStreamCreateWithFlags(&s.copy_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&s.ev_spliced, hipEventDisableTiming) != hipSuccess))
        return fail(AM_ERR_HIP, "could not create the copy stream");
    // Incremental re-scan (am_replace.hip): after the first pass only windows around the replacements are scanned and
    // merged with the shifted records of the previous pass.  Needs the suffix-filter kernel's position-local semantics
    // (automata with the empty needle re-scan everything); AM_RP_FULL_SCANS=1 turns it off (A/B, tests).
    const Flavor* flavor = nullptr;
    AM_TRY(prepare(r->a, r->case_mode, &flavor));
    const uint32_t ov = 4u * (flavor->h.max_needle_cps ? flavor->h.max_needle_cps : 1u) + 4u;
    const bool inc_enabled = flavor->h.sf_enabled && flavor->h.root_vlen == 0 && r->a->kernel_pref != 1 && !cfg::on(cfg::kRpFullScans);
    bool have_inc = false;
    uint64_t inc_n_rec = 0;
    int cur_rec = 0;
    // AM_RP_TRACE=1: wall-clock split of the loop on stderr (development aid)
    const bool trace = cfg::on(cfg::kRpTrace);
    double t_scan = 0, t_fold = 0, t_splice = 0, t_home = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    struct Report { bool on; double &a, &b, &c, &d; ~Report() { if (on) std::fprintf(stderr, "[am_replacer] scan %.1f ms, fold+scans %.1f ms, splice+D2H %.1f ms, scatter %.1f ms\n", a * 1e3, b * 1e3, c * 1e3, d * 1e3); } } report{trace, t_scan, t_fold, t_splice, t_home};

    while (n_act > 0) {
        double t0 = now();
        res->passes++;
        // ---- the scan (Replacer.hs:223-225): everything, unless the previous pass already derived this pass's records
        uint64_t n_rec = 0;
        DevBuf& records = s.recbuf[cur_rec];
        if (have_inc) { n_rec = inc_n_rec; have_inc = false; }
        else {
            res->scanned += total;
            s.ws.dev = in->dev;
            s.ws.d_text = const_cast<uint8_t*>(cur_text); s.ws.d_offsets = const_cast<uint64_t*>(cur_offs); s.ws.owns = false;
            s.ws.total = total; s.ws.n_hay = n_act;
            AM_TRY(finish_batch(&s.ws));
            auto sink = [&](uint64_t n, Record** ptr) -> int { AM_TRY(records.ensure(n * sizeof(Record))); *ptr = (Record*)records.p; return AM_OK; };
            AM_TRY(run_records(r->a, r->case_mode, &s.ws, sink, &n_rec));
        }
        t_scan += now() - t0; t0 = now();
        // ---- per-haystack fold of the records
        const uint64_t n1 = (uint64_t)n_act + 1;
        AM_TRY(s.rec_first.ensure(n1 * 8)); AM_TRY(s.kept.ensure((n_rec + 1) * sizeof(RpKept))); AM_TRY(s.hs.ensure(n1 * sizeof(RpHay)));
        AM_TRY(s.len_next.ensure(n1 * 8)); AM_TRY(s.len_fin.ensure(n1 * 8)); AM_TRY(s.tiles.ensure(n1 * 4)); AM_TRY(s.act.ensure(n1 * 4)); AM_TRY(s.fin.ensure(n1 * 4));
        AM_TRY(s.off_next.ensure(n1 * 8)); AM_TRY(s.off_fin.ensure(n1 * 8)); AM_TRY(s.tile_off.ensure(n1 * 8)); AM_TRY(s.act_idx.ensure(n1 * 8)); AM_TRY(s.fin_idx.ensure(n1 * 8));
        size_t t32 = 0, t64 = 0;
        if (scan_temp_bytes(n1, &t32) != hipSuccess || scan64_temp_bytes(n1, &t64) != hipSuccess) return fail(AM_ERR_HIP, "scan sizing failed");
        const size_t tmp_bytes = t32 > t64 ? t32 : t64;
        AM_TRY(s.scan_tmp.ensure(tmp_bytes + 16));
        AM_TRY(records.ensure(sizeof(Record)));            // a valid pointer even when nothing matched
        RpRoute route{(uint64_t*)s.len_next.p, (uint64_t*)s.len_fin.p, (uint32_t*)s.tiles.p, (uint32_t*)s.act.p, (uint32_t*)s.fin.p};
        { Prof pr("rp_ranges", st); HIP_TRY(launch_rp_ranges((const Record*)records.p, n_rec, (uint64_t*)s.rec_first.p, route, n_act, st)); }
        AM_TRY(rp_fold(s, r, r->case_mode == AM_IGNORE_CASE, cur_text, cur_offs, (const Record*)records.p, n_rec, cur_thr, max_length, route, n_act, st, (const uint64_t*)s.rec_first.p));
        RpRouted rt{(const uint64_t*)s.off_next.p, (const uint64_t*)s.off_fin.p, (const uint64_t*)s.tile_off.p, (const uint64_t*)s.act_idx.p, (const uint64_t*)s.fin_idx.p};
        // windows of the incremental re-scan (their geometry follows from the kept matches alone, the text is copied after the splice)
        const bool try_inc = inc_enabled && n_rec > 0;
        const bool small = n1 <= (1u << 18);          // bookkeeping sums in one launch (k_scan_jobs) instead of a dozen scan launches
        size_t tmp2 = tmp_bytes;
        uint64_t woffs_last = n_rec;
        if (try_inc) {
            AM_TRY(s.nwin.ensure(n1 * 4)); AM_TRY(s.win_off.ensure(n1 * 8));
            AM_TRY(s.wins.ensure((n_rec + 1) * sizeof(RpWin))); AM_TRY(s.wlen.ensure((n_rec + 2) * 4)); AM_TRY(s.woffs.ensure((n_rec + 2) * 8));
            size_t tw = 0;
            if (scan_temp_bytes(n_rec + 1, &tw) != hipSuccess) return fail(AM_ERR_HIP, "scan sizing failed");
            AM_TRY(s.scan_tmp.ensure(std::max(tw, tmp_bytes) + 16));
            tmp2 = s.scan_tmp.cap - 16;
            HIP_TRY(launch_rp_win_count((const RpHay*)s.hs.p, n_act, (uint32_t*)s.nwin.p, st));
        }
        { Prof pr("rp_scans", st);
          if (small) {
              ScanJobs jobs{};
              jobs.j[0] = ScanJob{nullptr, route.len_next, (uint64_t*)s.off_next.p, n1, nullptr};
              jobs.j[1] = ScanJob{nullptr, route.len_fin, (uint64_t*)s.off_fin.p, n1, nullptr};
              jobs.j[2] = ScanJob{route.tiles, nullptr, (uint64_t*)s.tile_off.p, n1, nullptr};
              jobs.j[3] = ScanJob{route.act, nullptr, (uint64_t*)s.act_idx.p, n1, nullptr};
              jobs.j[4] = ScanJob{route.fin, nullptr, (uint64_t*)s.fin_idx.p, n1, nullptr};
              jobs.n_jobs = 5;
              if (try_inc) { jobs.j[5] = ScanJob{(const uint32_t*)s.nwin.p, nullptr, (uint64_t*)s.win_off.p, n1, nullptr}; jobs.n_jobs = 6; }
              HIP_TRY(launch_scan_jobs(jobs, st));
          } else {
              HIP_TRY(launch_scan64(s.scan_tmp.p, tmp2, route.len_next, (uint64_t*)s.off_next.p, n1, st));
              HIP_TRY(launch_scan64(s.scan_tmp.p, tmp2, route.len_fin, (uint64_t*)s.off_fin.p, n1, st));
              HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, route.tiles, (uint64_t*)s.tile_off.p, n1, st));
              HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, route.act, (uint64_t*)s.act_idx.p, n1, st));
              HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, route.fin, (uint64_t*)s.fin_idx.p, n1, st));
              if (try_inc) HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, (const uint32_t*)s.nwin.p, (uint64_t*)s.win_off.p, n1, st));
          } }
        if (try_inc) {
            Prof pr("rp_windows", st);
            if (!small) HIP_TRY(hipMemsetAsync(s.wlen.p, 0, (n_rec + 2) * 4, st));     // at most one window per record; unused entries scan as zeros
            HIP_TRY(launch_rp_win_meta(r->t, rt, (const RpHay*)s.hs.p, (const uint64_t*)s.rec_first.p, (const RpKept*)s.kept.p,
                                       (const uint64_t*)s.win_off.p, ov, (RpWin*)s.wins.p, (uint32_t*)s.wlen.p, n_act, st));
            if (small) {          // exactly n_win + 1 elements: the count is read on the device
                ScanJobs jobs{};
                jobs.j[0] = ScanJob{(const uint32_t*)s.wlen.p, nullptr, (uint64_t*)s.woffs.p, 1, (const uint64_t*)s.win_off.p + n_act};
                jobs.n_jobs = 1;
                HIP_TRY(launch_scan_jobs(jobs, st));
                woffs_last = ~0ull;
            } else HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, (const uint32_t*)s.wlen.p, (uint64_t*)s.woffs.p, n_rec + 1, st));
        }
        // bytes of next text, bytes of finished text, tiles, haystacks still active, haystacks finished, windows, window bytes
        HIP_TRY(launch_rp_totals(rt, n_act, try_inc ? (const uint64_t*)s.win_off.p : nullptr, try_inc ? (const uint64_t*)s.woffs.p : nullptr, woffs_last,
                                 (uint64_t*)s.totals.p, st));
        HIP_TRY(hipMemcpyAsync(s.tot_host, s.totals.p, 56, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const uint64_t* tot = s.tot_host;
        const uint64_t total_next = tot[0], total_fin = tot[1], n_tiles = tot[2], n_next = tot[3], n_fin = tot[4], n_win = tot[5], total_w = tot[6];
        t_fold += now() - t0; t0 = now();
        res->spliced += total_next + total_fin;
        if (n_tiles >= 0x7FFFFFF0ull) return fail(AM_ERR_UNSUPPORTED, "replacement output too large for one launch; split the batch");
        // ---- replace (Replacer.hs:163-180) into the next batch / the finished buffer
        AM_TRY(s.text[nxt].ensure(padded_text(total_next))); AM_TRY(s.offs[nxt].ensure((n_next + 1) * 8));
        AM_TRY(s.orig[nxt].ensure((n_next + 1) * 4)); AM_TRY(s.thr[nxt].ensure((n_next + 1) * 8));
        AM_TRY(s.fin_text.ensure(total_fin + 16)); AM_TRY(s.fin_meta.ensure((n_fin + 1) * sizeof(RpFin))); AM_TRY(s.tile_hay.ensure((n_tiles + 1) * 4));
        { Prof pr("rp_route", st);
          HIP_TRY(launch_rp_route((const RpHay*)s.hs.p, rt, cur_orig, n_act, (uint64_t*)s.offs[nxt].p, (uint32_t*)s.orig[nxt].p, (int64_t*)s.thr[nxt].p, (RpFin*)s.fin_meta.p, st)); }
        { Prof pr("rp_splice", st);
          HIP_TRY(launch_rp_splice(r->t, cur_text, cur_offs, (const uint64_t*)s.rec_first.p, (const RpKept*)s.kept.p, (const RpHay*)s.hs.p, rt, n_act, n_tiles,
                                   (uint32_t*)s.tile_hay.p, (uint8_t*)s.text[nxt].p, (uint8_t*)s.fin_text.p, st)); }
        HIP_TRY(hipMemsetAsync((uint8_t*)s.text[nxt].p + total_next, 0, padded_text(total_next) - (size_t)total_next, st));
        // ---- finished haystacks go home: the copy runs on its own stream, next to the window scans below
        uint8_t* home = nullptr;
        if (total_fin) AM_TRY(res->room((size_t)total_fin, &home));
        AM_TRY(s.pin_meta((n_fin + 1) * sizeof(RpFin)));
        HIP_TRY(hipEventRecord(s.ev_spliced, st));
        HIP_TRY(hipStreamWaitEvent(s.copy_stream, s.ev_spliced, 0));
        if (total_fin) HIP_TRY(hipMemcpyAsync(home, s.fin_text.p, total_fin, hipMemcpyDefault, s.copy_stream));      // the slab is pinned host memory, or device memory for results that stay there
        if (n_fin) HIP_TRY(hipMemcpyAsync(s.fin_host, s.fin_meta.p, n_fin * sizeof(RpFin), hipMemcpyDeviceToHost, s.copy_stream));
        auto finished_home = [&]() -> int {
            HIP_TRY(hipStreamSynchronize(s.copy_stream));
            for (uint64_t i = 0; i < n_fin; i++) {
                const RpFin& f = s.fin_host[i];
                if (f.orig >= n_hay || f.off + f.len > total_fin) return fail(AM_ERR_HIP, "replacer pass produced inconsistent metadata (internal error)");
                if (f.status == kRpNothing) res->just[f.orig] = 0;
                else res->text[f.orig] = am_replaced::Item{home + f.off, (size_t)f.len};
            }
            return AM_OK;
        };
        t_splice += now() - t0; t0 = now();
        t_home += now() - t0; t0 = now();
        // ---- next pass's records without a full scan: windows around the replacements + the shifted old records
        if (try_inc && n_next > 0 && n_win > 0 && n_win < 0xFFFFFFF0ull && total_w <= total_next / 2) {
            const uint8_t* text_next = (const uint8_t*)s.text[nxt].p;
            AM_TRY(s.wtext.ensure(padded_text(total_w)));
            { Prof pr("rp_windows", st);
              HIP_TRY(launch_rp_win_copy((const RpWin*)s.wins.p, (const uint64_t*)s.woffs.p, text_next, (uint8_t*)s.wtext.p, n_win, st));
              HIP_TRY(hipMemsetAsync((uint8_t*)s.wtext.p + total_w, 0, padded_text(total_w) - (size_t)total_w, st)); }
            s.ws2.dev = in->dev;
            s.ws2.d_text = s.wtext.p; s.ws2.d_offsets = (uint64_t*)s.woffs.p; s.ws2.owns = false; s.ws2.total = total_w; s.ws2.n_hay = (uint32_t)n_win;
            AM_TRY(finish_batch(&s.ws2));
            uint64_t n_wrec = 0;
            auto wsink = [&](uint64_t n, Record** ptr) -> int { AM_TRY(s.wrec.ensure(n * sizeof(Record))); *ptr = (Record*)s.wrec.p; return AM_OK; };
            AM_TRY(run_records(r->a, r->case_mode, &s.ws2, wsink, &n_wrec));
            res->scanned += total_w;
            AM_TRY(s.wrec.ensure(sizeof(Record)));
            AM_TRY(s.wrec_first.ensure((n_win + 1) * 8)); AM_TRY(s.mcount.ensure((n_next + 1) * 4)); AM_TRY(s.moff.ensure((n_next + 1) * 8));
            DevBuf& next_records = s.recbuf[cur_rec ^ 1];
            AM_TRY(next_records.ensure((n_rec + n_wrec + 1) * sizeof(Record)));          // upper bound; the exact count arrives with the end-of-pass sync
            Prof pr("rp_merge", st);
            HIP_TRY(launch_rp_ranges((const Record*)s.wrec.p, n_wrec, (uint64_t*)s.wrec_first.p, RpRoute{nullptr, nullptr, nullptr, nullptr, nullptr}, (uint32_t)n_win, st));
            HIP_TRY(hipMemsetAsync((uint32_t*)s.mcount.p + n_next, 0, 4, st));
            HIP_TRY(launch_rp_merge(false, (const Record*)records.p, (const uint64_t*)s.rec_first.p, (const RpKept*)s.kept.p, (const RpHay*)s.hs.p, cur_offs, rt,
                                    (const uint64_t*)s.win_off.p, (const RpWin*)s.wins.p, (const Record*)s.wrec.p, (const uint64_t*)s.wrec_first.p, ov, n_act,
                                    (uint32_t*)s.mcount.p, nullptr, nullptr, st));
            if (n_next + 1 <= (1u << 18)) {
                ScanJobs jobs{};
                jobs.j[0] = ScanJob{(const uint32_t*)s.mcount.p, nullptr, (uint64_t*)s.moff.p, n_next + 1, nullptr};
                jobs.n_jobs = 1;
                HIP_TRY(launch_scan_jobs(jobs, st));
            } else HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, (const uint32_t*)s.mcount.p, (uint64_t*)s.moff.p, n_next + 1, st));
            HIP_TRY(launch_rp_merge(true, (const Record*)records.p, (const uint64_t*)s.rec_first.p, (const RpKept*)s.kept.p, (const RpHay*)s.hs.p, cur_offs, rt,
                                    (const uint64_t*)s.win_off.p, (const RpWin*)s.wins.p, (const Record*)s.wrec.p, (const uint64_t*)s.wrec_first.p, ov, n_act,
                                    (uint32_t*)s.mcount.p, (const uint64_t*)s.moff.p, (Record*)next_records.p, st));
            HIP_TRY(hipMemcpyAsync(&s.tot_host[7], (uint64_t*)s.moff.p + n_next, 8, hipMemcpyDeviceToHost, st));
            have_inc = true;
        }
        HIP_TRY(hipStreamSynchronize(st));            // end of pass: the merged record count (if any) is on the host now
        if (have_inc) inc_n_rec = s.tot_host[7];
        t_scan += now() - t0; t0 = now();
        AM_TRY(finished_home());
        t_home += now() - t0; t0 = now();
        cur_rec ^= 1;
        cur_text = (const uint8_t*)s.text[nxt].p; cur_offs = (const uint64_t*)s.offs[nxt].p;
        cur_orig = (const uint32_t*)s.orig[nxt].p; cur_thr = (const int64_t*)s.thr[nxt].p;
        total = total_next; n_act = (uint32_t)n_next; nxt ^= 1;
        t_scan += now() - t0;
    }
    return AM_OK;
}

}  // namespace

// All passes of every haystack in ONE kernel (am_rploop.hip): a wavefront takes a haystack and runs its loop to the end.  *handled = false:
// the batch is not for this path (or a haystack outgrew its regions) and nothing of `res` was touched: the caller takes the pass-by-pass paths.
// (tests, include/am_debug.h) haystacks the last one-kernel run finished out of LDS (k_rp_lds); the others went through k_rp_loop
static std::atomic<uint32_t> g_last_lds_haystacks{0};
extern "C" uint32_t am_debug_rp_lds_haystacks(void) { return g_last_lds_haystacks.load(std::memory_order_relaxed); }

static int replacer_run_loop(const am_replacer* r, const am_batch* in, uint64_t max_length, am_replaced* res, bool* handled)
{
    *handled = false;
    const uint32_t n_hay = in->n_hay;
    if (n_hay == 0 || in->dev != r->a->dev) return AM_OK;
    const long sw = cfg::get(cfg::kRpLoop);
    if (sw == 0) return AM_OK;
    const Flavor* fl = nullptr;
    AM_TRY(prepare(r->a, r->case_mode, &fl));
    if (!fl->h.sf_enabled || fl->h.root_vlen != 0 || r->a->kernel_pref == 1) return AM_OK;
    if (sw != 1) {
        // unset: batches of many documents, and no switch that asks for one of the other loops
        if (!(n_hay >= 64 && in->total / n_hay <= (1ull << 20))) return AM_OK;
        for (cfg::Key k : {cfg::kRpFullScans, cfg::kRpSplice, cfg::kRpPieces, cfg::kRpParallelFold, cfg::kRpGroups, cfg::kRpNoFuse, cfg::kRpNoRangeReuse, cfg::kRpNoSpin, cfg::kRpMatMain})
            if (cfg::get(k) != cfg::kUnset) return AM_OK;
    }
    // how far a replacement's neighbourhood reaches = the longest needle in haystack bytes: for CaseSensitive replacers the byte depth of the
    // automaton's trie (am_replacer_create; 4 bytes per code point + 4 would make the windows of ASCII needles four times as long), the bound by code
    // points under IgnoreCase (the matched text may be longer than the lower-cased needle) and for automata attached to an image (depth unknown)
    const uint32_t ov_cps = 4u * (fl->h.max_needle_cps ? fl->h.max_needle_cps : 1u) + 4u;
    const uint32_t ov = r->case_mode == AM_CASE_SENSITIVE && r->max_needle_bytes > 0 && r->max_needle_bytes < ov_cps ? r->max_needle_bytes : ov_cps;
    const uint64_t wcap64 = ((2ull * ov + r->max_repl_len + 16ull) + 63ull) & ~63ull;
    if (wcap64 > 4096 || wcap64 * n_hay > (1ull << 30) || in->total >= (1ull << 40)) return AM_OK;
    ON_DEVICE(in->dev);
    hipStream_t st; AM_TRY(get_stream(in->dev, &st));
    RpSession* sp = nullptr;
    { std::lock_guard<std::mutex> lk(r->session_mu); if (!r->sessions.empty()) { sp = static_cast<RpSession*>(r->sessions.back()); r->sessions.pop_back(); } }
    if (!sp) sp = new RpSession();
    struct Return {
        const am_replacer* r; RpSession* sp;
        ~Return()
        {
            if (sp->copy_stream) (void)hipStreamSynchronize(sp->copy_stream);
            if (sp->device_bytes() > (8192ull << 20)) { delete sp; return; }      // (the loop's regions + the groups' staging of a 1-GiB batch are ~3 GiB of the 288)
            std::vector<RpSession*> doomed;
            { std::lock_guard<std::mutex> lk(r->session_mu);
              const_cast<am_replacer*>(r)->session_delete = [](void* p) { delete static_cast<RpSession*>(p); };
              r->sessions.push_back(sp);
              for (;;) {
                  size_t held = 0;
                  for (void* q : r->sessions) held += static_cast<RpSession*>(q)->device_bytes();
                  if (r->sessions.size() <= 1 || (r->sessions.size() <= 8 && held <= (4096ull << 20))) break;
                  doomed.push_back(static_cast<RpSession*>(r->sessions.front()));
                  r->sessions.erase(r->sessions.begin());
              } }
            for (RpSession* q : doomed) delete q;
        }
    } give_back{r, sp};
    RpSession& s = *sp;
    const bool trace = cfg::on(cfg::kRpTrace);
    auto say = [&](const char* what) { if (trace) { (void)hipStreamSynchronize(st); std::fprintf(stderr, "[am_replacer loop] %s\n", what); std::fflush(stderr); } };
    // the first (and only full) scan
    say("first scan");
    uint64_t n_rec = 0;
    s.ws.dev = in->dev; s.ws.d_text = in->d_text; s.ws.d_offsets = in->d_offsets; s.ws.owns = false; s.ws.total = in->total; s.ws.n_hay = n_hay;
    AM_TRY(finish_batch(&s.ws));
    {
        auto sink = [&](uint64_t n, Record** ptr) -> int { AM_TRY(s.recbuf[0].ensure((n + 1) * sizeof(Record))); *ptr = (Record*)s.recbuf[0].p; return AM_OK; };
        AM_TRY(run_records(r->a, r->case_mode, &s.ws, sink, &n_rec));
    }
    if (n_rec >= (1ull << 26)) return AM_OK;                 // (the regions below would not fit: the pass-by-pass loop scans again)
    const uint64_t n1 = (uint64_t)n_hay + 1;
    const uint64_t rec_total = 4 * n_rec + 128ull * n_hay, pc_total = 8 * n_rec + 128ull * n_hay;      // = the sums of k_rp_loop_caps' region sizes
    AM_TRY(s.recbuf[0].ensure(sizeof(Record)));
    AM_TRY(s.rec_first.ensure(n1 * 8));
    AM_TRY(s.lp_cap_r.ensure(n1 * 4)); AM_TRY(s.lp_cap_p.ensure(n1 * 4)); AM_TRY(s.lp_rec_base.ensure(n1 * 8)); AM_TRY(s.lp_pc_base.ensure(n1 * 8));
    AM_TRY(s.lp_rec.ensure((rec_total + 1) * sizeof(Record))); AM_TRY(s.lp_pc.ensure((pc_total + 1) * sizeof(RpPiece)));
    AM_TRY(s.lp_kept.ensure((rec_total / 2 + 1) * sizeof(RpKept)));
    AM_TRY(s.lp_wtext.ensure(wcap64 * n_hay + 64)); AM_TRY(s.lp_out.ensure(n1 * sizeof(RpLoopOut))); AM_TRY(s.lp_ctrl.ensure(128));
    size_t t32 = 0;
    if (scan_temp_bytes(n1, &t32) != hipSuccess) return fail(AM_ERR_HIP, "scan sizing failed");
    AM_TRY(s.scan_tmp.ensure(t32 + 16));
    { Prof pr("rp_ranges", st);
      HIP_TRY(launch_rp_ranges((const Record*)s.recbuf[0].p, n_rec, (uint64_t*)s.rec_first.p, RpRoute{nullptr, nullptr, nullptr, nullptr, nullptr}, n_hay, st)); }
    HIP_TRY(hipMemsetAsync(s.lp_ctrl.p, 0, 128, st));
    { Prof pr("rp_scans", st);
      HIP_TRY(launch_rp_loop_caps((const uint64_t*)s.rec_first.p, n_hay, (uint32_t*)s.lp_cap_r.p, (uint32_t*)s.lp_cap_p.p, (uint32_t*)s.lp_ctrl.p + 6, st));
      HIP_TRY(launch_scan(s.scan_tmp.p, t32, (const uint32_t*)s.lp_cap_r.p, (uint64_t*)s.lp_rec_base.p, n1, st));
      HIP_TRY(launch_scan(s.scan_tmp.p, t32, (const uint32_t*)s.lp_cap_p.p, (uint64_t*)s.lp_pc_base.p, n1, st)); }
    say("ranges + region sizes");
    if (sw != 1) {
        // a wavefront walks its haystack's whole record list in every pass: one document with very many matches would be the tail of the launch
        // (the pass-by-pass loop folds such lists in parallel over the records)
        AM_TRY(s.pin_loop(64));
        uint32_t* c = (uint32_t*)s.lp_host;
        HIP_TRY(hipMemcpyAsync(c, s.lp_ctrl.p, 64, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (c[6] > 4096u) return AM_OK;
    }
    RpLoop a{};
    a.t = r->t; a.s = make_sf_view(fl->d_image, fl->h);
    a.text = (const uint8_t*)in->d_text; a.offsets = in->d_offsets; a.n_hay = n_hay; a.ov = ov;
    a.recs0 = (const Record*)s.recbuf[0].p; a.rec_first0 = (const uint64_t*)s.rec_first.p;
    a.rec_buf = (Record*)s.lp_rec.p; a.rec_base = (const uint64_t*)s.lp_rec_base.p;
    a.pc_buf = (RpPiece*)s.lp_pc.p; a.pc_base = (const uint64_t*)s.lp_pc_base.p;
    a.kept_buf = (RpKept*)s.lp_kept.p; a.wtext = (uint8_t*)s.lp_wtext.p; a.wcap = (uint32_t)wcap64;
    a.max_len = max_length; a.out = (RpLoopOut*)s.lp_out.p; a.ctrl = (uint32_t*)s.lp_ctrl.p;
    a.pad = cfg::get(cfg::kRpTrace) >= 3 ? 1u : 0u;
    // k_rp_lds first: a haystack's lists in LDS for all its passes (am_rplds.hip); what does not fit there raises its redo flag and k_rp_loop, launched
    // right behind, runs exactly those haystacks (lists in global memory).  AM_RP_LDS=0 (A/B, tests), the instrumented instantiation and replacement
    // blobs beyond 2 GiB (piece sources are 31-bit offsets in LDS): k_rp_loop alone.
    const bool use_lds = cfg::get(cfg::kRpLds) != 0 && r->n_repl_bytes < (1ull << 31);
    a.redo = nullptr; a.h_first = 0; a.pl_implicit = r->pl_implicit ? 1u : 0u;
    if (use_lds) {
        AM_TRY(s.lp_redo.ensure((size_t)n_hay * 4 + 64));
        HIP_TRY(hipMemsetAsync(s.lp_redo.p, 0, (size_t)n_hay * 4, st));
        a.redo = (uint32_t*)s.lp_redo.p;
    }
    // Haystack GROUPS.  Results that stay on the device: one group, one launch.  Results that go to the host (Replacer.run :: Text -> Text returns host text;
    // a gibibyte takes 20 ms over PCIe, four times what the passes take): the batch is cut into up to eight groups of >= 2048 haystacks, every group's
    // kernels are queued at once, and while the later groups still run their passes the finished ones are materialised and copied home on a second
    // stream -- the wire is busy from the first group's end to the last byte.  Nothing is shared between haystacks, so a group is just a launch over a
    // range of them (RpLoop::h_first).
    uint32_t n_groups = 1;
    if (res->dev < 0 && n_hay >= 4096 && in->total >= (64ull << 20) && !a.pad && !cfg::on(cfg::kRpMatMain)) { n_groups = n_hay / 2048u; if (n_groups > 8) n_groups = 8; }
    if (n_groups > 1 && !s.copy_stream && (hipStreamCreateWithFlags(&s.copy_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&s.ev_spliced, hipEventDisableTiming) != hipSuccess))
        return fail(AM_ERR_HIP, "could not create the copy stream");
    const size_t out_bytes = (size_t)n_hay * sizeof(RpLoopOut);
    const size_t tab_bytes = (size_t)n_hay * (sizeof(RpFin) + 8 + 4) + 64;
    AM_TRY(s.pin_loop(64 + out_bytes + tab_bytes));
    uint32_t* ctrl_h = (uint32_t*)s.lp_host;
    RpLoopOut* out_h = (RpLoopOut*)((uint8_t*)s.lp_host + 64);
    RpFin* fin_h = (RpFin*)((uint8_t*)s.lp_host + 64 + out_bytes);
    uint64_t* fstart_h = (uint64_t*)(fin_h + n_hay);
    uint32_t* fcnt_h = (uint32_t*)(fstart_h + n_hay);
    AM_TRY(s.lp_fin.ensure((size_t)n_hay * sizeof(RpFin))); AM_TRY(s.lp_fin_start.ensure((size_t)n_hay * 8)); AM_TRY(s.lp_fin_cnt.ensure((size_t)n_hay * 4));
    struct Events {
        std::vector<hipEvent_t> ev;
        ~Events() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); }
    } done;
    done.ev.assign(n_groups, nullptr);
    auto group_lo = [&](uint32_t g) { return (uint32_t)((uint64_t)n_hay * g / n_groups); };
    say("launch");
    for (uint32_t g = 0; g < n_groups; g++) {
        const uint32_t h0 = group_lo(g), h1 = group_lo(g + 1);
        a.h_first = h0;
        if (use_lds) { Prof pr("rp_lds", st); HIP_TRY(launch_rp_lds(r->case_mode == AM_IGNORE_CASE, a, h1 - h0, st)); }
        { Prof pr("rp_loop", st); HIP_TRY(launch_rp_loop(r->case_mode == AM_IGNORE_CASE, a, h1 - h0, (int)cfg::get(cfg::kRpLoopWaves), st)); }
        HIP_TRY(hipMemcpyAsync(out_h + h0, (const RpLoopOut*)s.lp_out.p + h0, (size_t)(h1 - h0) * sizeof(RpLoopOut), hipMemcpyDeviceToHost, st));
        if (g + 1 == n_groups) HIP_TRY(hipMemcpyAsync(ctrl_h, s.lp_ctrl.p, 64, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventCreateWithFlags(&done.ev[g], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(done.ev[g], st));
    }
    say("launched");
    res->text.assign(n_hay, am_replaced::Item());
    res->just.assign(n_hay, 1);
    std::vector<uint8_t*> home_of(n_groups, nullptr);
    uint64_t total_all = 0;
    bool gave_up = false;
    for (uint32_t g = 0; g < n_groups && !gave_up; g++) {
        const uint32_t h0 = group_lo(g), h1 = group_lo(g + 1);
        HIP_TRY(hipEventSynchronize(done.ev[g]));
        // what every haystack of the group ended as; the finished texts: one materialise launch over the final piece lists
        uint64_t total_fin = 0;
        for (uint32_t i = h0; i < h1; i++) {
            const RpLoopOut& o = out_h[i];
            if (o.status > kRpNothing || o.pieces_at + o.n_pieces + 1 > pc_total) { gave_up = true; break; }      // a haystack that was given up (overflow: nothing written) or inconsistent metadata: see below
            fin_h[i] = RpFin{total_fin, o.len, i, o.status};
            fstart_h[i] = o.pieces_at; fcnt_h[i] = o.n_pieces;
            total_fin += o.len;
        }
        if (gave_up) break;
        hipStream_t ms = n_groups > 1 ? s.copy_stream : st;
        if (n_groups > 1) HIP_TRY(hipStreamWaitEvent(ms, done.ev[g], 0));
        HIP_TRY(hipMemcpyAsync((RpFin*)s.lp_fin.p + h0, fin_h + h0, (size_t)(h1 - h0) * sizeof(RpFin), hipMemcpyHostToDevice, ms));
        HIP_TRY(hipMemcpyAsync((uint64_t*)s.lp_fin_start.p + h0, fstart_h + h0, (size_t)(h1 - h0) * 8, hipMemcpyHostToDevice, ms));
        HIP_TRY(hipMemcpyAsync((uint32_t*)s.lp_fin_cnt.p + h0, fcnt_h + h0, (size_t)(h1 - h0) * 4, hipMemcpyHostToDevice, ms));
        uint8_t* home = nullptr;
        if (total_fin) AM_TRY(res->room((size_t)total_fin, &home));
        home_of[g] = home;
        uint8_t* d_fin = home;
        if (res->dev < 0) {
            DevBuf& sb = n_groups > 1 ? s.lp_stage[g] : s.fin_text;      // (the session keeps them: no allocation in the steady state)
            AM_TRY(sb.ensure(total_fin + 16));
            d_fin = (uint8_t*)sb.p;
        }
        { Prof pr("pt_materialise", ms);
          HIP_TRY(launch_pt_materialise((const RpPiece*)s.lp_pc.p, (const uint64_t*)s.lp_fin_start.p + h0, (const uint32_t*)s.lp_fin_cnt.p + h0, (const RpFin*)s.lp_fin.p + h0, h1 - h0,
                                        (const uint8_t*)in->d_text, r->t.repl, d_fin, ms)); }      // (an output-centred variant -- aligned 16-byte chunks, chunk -> piece map in LDS -- was measured in round 5: the same 0.98 ms per GiB)
        if (res->dev < 0 && total_fin) {
            // home in requests of 256 MiB (one huge request keeps the copy engine from overlapping with anything else queued behind it)
            for (uint64_t off = 0; off < total_fin; off += (256ull << 20)) {
                const uint64_t n = std::min<uint64_t>(256ull << 20, total_fin - off);
                HIP_TRY(hipMemcpyAsync(home + off, d_fin + off, n, hipMemcpyDeviceToHost, ms));
            }
        }
        total_all += total_fin;
    }
    say("materialise queued");
    HIP_TRY(hipStreamSynchronize(st));
    if (n_groups > 1) HIP_TRY(hipStreamSynchronize(s.copy_stream));
    if (a.pad && use_lds) {
        uint64_t ph[10];
        HIP_TRY(hipMemcpy(ph, (const uint8_t*)s.lp_ctrl.p + 32, 80, hipMemcpyDeviceToHost));
        static const char* const names[9] = {"records in + fold", "select + payload", "overlap removal", "counts + dead slots", "piece list", "gather", "window scan", "inserts", "whole run"};
        for (int i = 0; i < 9; i++) std::fprintf(stderr, "[am_replacer lds] %-22s %14llu cycles = %5.1f %% of the wavefronts' time, %8.0f per pass\n", names[i], (unsigned long long)ph[i],
                                                 100.0 * (double)ph[i] / (double)(ph[8] ? ph[8] : 1), (double)ph[i] / (double)(ph[9] ? ph[9] : 1));
        std::fprintf(stderr, "[am_replacer lds] passes of all haystacks: %llu\n", (unsigned long long)ph[9]);
    } else if (a.pad) {
        uint64_t ph[8];
        HIP_TRY(hipMemcpy(ph, (const uint8_t*)s.lp_ctrl.p + 32, 64, hipMemcpyDeviceToHost));
        static const char* const names[7] = {"fold 1 (best priority)", "fold 2 (select, overlaps)", "pieces", "record copies + searches", "gather", "window scan", "whole run"};
        for (int i = 0; i < 7; i++) std::fprintf(stderr, "[am_replacer loop] %-26s %14llu cycles = %5.1f %% of the wavefronts' time, %8.0f per pass\n", names[i], (unsigned long long)ph[i],
                                                 100.0 * (double)ph[i] / (double)(ph[6] ? ph[6] : 1), (double)ph[i] / (double)(ph[7] ? ph[7] : 1));
        std::fprintf(stderr, "[am_replacer loop] passes of all haystacks: %llu\n", (unsigned long long)ph[7]);
    }
    if (trace) { std::fprintf(stderr, "[am_replacer loop] kernels done: overflow %u passes %u watchdog %u; %u of %u haystacks out of LDS, %u group(s)\n", ctrl_h[0], ctrl_h[1], ctrl_h[5], ctrl_h[7], n_hay, n_groups); std::fflush(stderr); }
    g_last_lds_haystacks.store(use_lds ? ctrl_h[7] : 0u, std::memory_order_relaxed);
    if (ctrl_h[0] != 0 || gave_up) {
        // a haystack outgrew its regions (its result was never written): the pass-by-pass loop takes the batch; what the groups before it brought home is dropped
        if (ctrl_h[0] == 0) return fail(AM_ERR_HIP, "replacer loop produced inconsistent metadata (internal error)");
        res->text.clear(); res->just.clear();
        for (const Slab& sl : res->slabs) res->pool().give(sl);
        res->slabs.clear();
        return AM_OK;
    }
    say("done");
    for (uint32_t g = 0; g < n_groups; g++) {
        for (uint32_t i = group_lo(g); i < group_lo(g + 1); i++) {
            if (fin_h[i].status == kRpNothing) res->just[i] = 0;
            else res->text[i] = am_replaced::Item{home_of[g] + fin_h[i].off, (size_t)fin_h[i].len};
        }
    }
    res->passes = ctrl_h[1];
    res->scanned += in->total + (((uint64_t)ctrl_h[3] << 32) | ctrl_h[2]);
    res->spliced += total_all;
    *handled = true;
    return AM_OK;
}

// Large batches are cut into a few groups of haystacks that run the pass loop CONCURRENTLY, one host thread and HIP stream per
// group: a pass is a chain of small kernels bound by launch and dependency latency, not by throughput, so the chains of
// different groups overlap on the GPU.  The groups share nothing but the (read-only) batch text and the replacer tables.
static int replacer_run_groups(const am_replacer* r, const am_batch* in, uint64_t max_length, am_replaced* res)
{
    const uint32_t n_hay = in->n_hay;
    { bool handled = false; AM_TRY(replacer_run_loop(r, in, max_length, res, &handled)); if (handled) return AM_OK; }
    uint32_t groups = n_hay / 2048u;
    if (groups > 2) groups = 2;        // measured on config 5: 1 -> 82 ms, 2 -> 54 ms, 4 -> 77 ms, 8 -> 109 ms (the groups' kernels start to queue behind each other)
    { const long v = cfg::get(cfg::kRpGroups); if (v >= 1 && v <= 16) groups = (uint32_t)v; }
    if (groups < 2 || n_hay < groups) return replacer_run(r, in, max_length, res);
    ON_DEVICE(in->dev);
    std::vector<uint64_t> offs((size_t)n_hay + 1);
    HIP_TRY(hipMemcpy(offs.data(), in->d_offsets, offs.size() * 8, hipMemcpyDeviceToHost));
    // group boundaries at haystacks whose text starts 16-byte aligned (the scan kernels load aligned 16-byte groups)
    std::vector<uint32_t> cut(1, 0);
    for (uint32_t g = 1; g < groups; g++) {
        uint32_t h = (uint32_t)((uint64_t)n_hay * g / groups);
        while (h < n_hay && (offs[h] & 15u)) h++;
        if (h > cut.back() && h < n_hay) cut.push_back(h);
    }
    cut.push_back(n_hay);
    const size_t G = cut.size() - 1;
    if (G < 2) return replacer_run(r, in, max_length, res);
    struct Group { am_batch b; am_replaced part; int rc = AM_OK; std::string err; DevBuf offs; };
    const int res_dev = res->dev;
    std::vector<std::unique_ptr<Group>> gs;
    for (size_t g = 0; g < G; g++) {
        auto gp = std::make_unique<Group>();
        gp->part.dev = res_dev;
        const uint32_t h0 = cut[g], h1 = cut[g + 1];
        std::vector<uint64_t> sub(h1 - h0 + 1);
        for (uint32_t i = 0; i <= h1 - h0; i++) sub[i] = offs[h0 + i] - offs[h0];
        AM_TRY(gp->offs.ensure(sub.size() * 8));
        HIP_TRY(hipMemcpy(gp->offs.p, sub.data(), sub.size() * 8, hipMemcpyHostToDevice));
        gp->b.dev = in->dev; gp->b.owns = false; gp->b.d_text = (uint8_t*)in->d_text + offs[h0]; gp->b.d_offsets = (uint64_t*)gp->offs.p;
        gp->b.total = sub.back(); gp->b.n_hay = h1 - h0;
        gs.push_back(std::move(gp));
    }
    std::vector<std::thread> pool;
    // The group threads launch on their own streams.  Work the caller queued on ITS stream before this call (a producer still
    // writing the text of an am_batch_from_device batch) must come first: an event on the caller's stream, waited for by every group stream.
    hipEvent_t caller_done = nullptr;
    {
        hipStream_t caller_st; AM_TRY(get_stream(in->dev, &caller_st));
        HIP_TRY(hipEventCreateWithFlags(&caller_done, hipEventDisableTiming));
        hipError_t e = hipEventRecord(caller_done, caller_st);
        if (e != hipSuccess) { (void)hipEventDestroy(caller_done); return fail(AM_ERR_HIP, std::string("hipEventRecord: ") + hipGetErrorString(e)); }
    }
    auto work = [&](size_t g) {
        Group& x = *gs[g];
        OnDevice od(in->dev);                                   // a fresh thread's current device is 0: the group's buffers and launches belong to the batch's device
        x.rc = od.rc;
        if (x.rc == AM_OK) x.rc = finish_batch(&x.b);
        if (x.rc == AM_OK) {
            hipStream_t st;
            x.rc = get_stream(in->dev, &st);
            if (x.rc == AM_OK && hipStreamWaitEvent(st, caller_done, 0) != hipSuccess) x.rc = fail(AM_ERR_HIP, "hipStreamWaitEvent failed");
        }
        if (x.rc == AM_OK) x.rc = replacer_run(r, &x.b, max_length, &x.part);
        if (x.rc != AM_OK) x.err = am_last_error();
    };
    // every group on a thread of its own (letting the calling thread take one of them serialised the two: 73 ms instead of 38, measured)
    for (size_t g = 0; g < G; g++) {
        try { pool.emplace_back(work, g); }
        catch (const std::exception&) { work(g); }          // no thread to be had: this group runs here (nothing may throw across the C ABI)
    }
    for (auto& t : pool) t.join();
    (void)hipEventDestroy(caller_done);
    res->text.assign(n_hay, am_replaced::Item());
    res->just.assign(n_hay, 1);
    int rc = AM_OK;
    for (size_t g = 0; g < G; g++) {
        Group& x = *gs[g];
        if (x.rc != AM_OK && rc == AM_OK) rc = fail(x.rc, x.err);
        for (uint32_t i = 0; i < x.b.n_hay && i < x.part.text.size(); i++) { res->text[cut[g] + i] = x.part.text[i]; res->just[cut[g] + i] = x.part.just[i]; }
        for (const Slab& sl : x.part.slabs) res->slabs.push_back(sl);      // the result keeps the group's pinned slabs (its texts point into them)
        x.part.slabs.clear();
        res->passes = std::max(res->passes, x.part.passes); res->scanned += x.part.scanned; res->spliced += x.part.spliced;
        for (DevBuf* d : {&x.b.hidx, &x.b.unit_counts, &x.b.unit_offsets, &x.b.scan_tmp, &x.b.small, &x.b.hay_counts, &x.b.flags, &x.b.unit_first, &x.b.pool, &x.b.block_next,
                          &x.b.sparse, &x.b.dense_counts, &x.b.dense_offsets, &x.b.dense_out}) d->release();
        x.offs.release();
    }
    return rc;
}

static int replacer_run_to(const am_replacer* r, const am_batch* b, uint64_t max_length, am_replaced** out, bool on_device)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!r || !b) return fail(AM_ERR_INVALID, "null replacer or batch");
    AM_TRY(ensure_runtime());
    am_replaced* res = new am_replaced();
    if (on_device) res->dev = b->dev;
    const int rc = replacer_run_groups(r, b, max_length, res);
    if (rc != AM_OK) { delete res; return rc; }
    *out = res;
    return AM_OK;
}

extern "C" int am_replacer_run_batch(const am_replacer* r, const am_batch* b, uint64_t max_length, am_replaced** out) { return replacer_run_to(r, b, max_length, out, false); }
extern "C" int am_replacer_run_batch_device(const am_replacer* r, const am_batch* b, uint64_t max_length, am_replaced** out) { return replacer_run_to(r, b, max_length, out, true); }

extern "C" int am_replacer_run(const am_replacer* r, const am_slice* hay, size_t n_hay, uint64_t max_length, am_replaced** out)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!r) return fail(AM_ERR_INVALID, "null replacer");
    ON_DEVICE(r->a->dev);
    am_batch* b = nullptr;
    AM_TRY(am_batch_upload(hay, n_hay, &b));
    const int rc = am_replacer_run_batch(r, b, max_length, out);
    am_batch_destroy(b);
    return rc;
}

// One pass of the fold only (SURVEY 8b am_run_priority): prependMatch + makeMatch (Replacer.hs:252-274) on the device,
// sort / removeOverlap / replace stay with the caller.
static_assert(sizeof(am_prio_match) == sizeof(RpSelected) && offsetof(am_prio_match, haystack) == offsetof(RpSelected, haystack), "am_prio_match layout");

extern "C" int am_run_priority(const am_replacer* r, const am_slice* hay, size_t n_hay, const int64_t* thresholds, int64_t* best_out,
                               am_prio_match** matches_out, size_t* n_matches_out)
{
    if (!matches_out || !n_matches_out) return fail(AM_ERR_INVALID, "out pointers are null");
    *matches_out = nullptr; *n_matches_out = 0;
    if (!r) return fail(AM_ERR_INVALID, "null replacer");
    if (n_hay && (!thresholds || !best_out)) return fail(AM_ERR_INVALID, "thresholds / best_out are null");
    if (n_hay == 0) return AM_OK;
    ON_DEVICE(r->a->dev);
    am_batch* b = nullptr;
    AM_TRY(am_batch_upload(hay, n_hay, &b));
    std::unique_ptr<am_batch, void (*)(am_batch*)> guard(b, am_batch_destroy);
    hipStream_t st; AM_TRY(get_stream(b->dev, &st));
    const uint32_t n = (uint32_t)n_hay;
    const uint64_t n1 = (uint64_t)n + 1;
    DevBuf records, rec_first, kept, hs, nk, off, thr, best, out, tmp;
    struct Release { std::vector<DevBuf*> l; ~Release() { for (DevBuf* d : l) d->release(); } } rel{{&records, &rec_first, &kept, &hs, &nk, &off, &thr, &best, &out, &tmp}};
    uint64_t n_rec = 0;
    auto sink = [&](uint64_t k, Record** ptr) -> int { AM_TRY(records.ensure(k * sizeof(Record))); *ptr = (Record*)records.p; return AM_OK; };
    AM_TRY(run_records(r->a, r->case_mode, b, sink, &n_rec));
    AM_TRY(records.ensure(sizeof(Record)));
    AM_TRY(rec_first.ensure(n1 * 8)); AM_TRY(kept.ensure((n_rec + 1) * sizeof(RpKept))); AM_TRY(hs.ensure(n1 * sizeof(RpHay)));
    AM_TRY(nk.ensure(n1 * 4)); AM_TRY(off.ensure(n1 * 8)); AM_TRY(thr.ensure(n1 * 8)); AM_TRY(best.ensure(n1 * 8));
    size_t tmp_bytes = 0;
    if (scan_temp_bytes(n1, &tmp_bytes) != hipSuccess) return fail(AM_ERR_HIP, "scan sizing failed");
    AM_TRY(tmp.ensure(tmp_bytes + 16));
    HIP_TRY(hipMemcpyAsync(thr.p, thresholds, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync((uint32_t*)nk.p + n, 0, 4, st));
    RpRoute route{nullptr, nullptr, (uint32_t*)nk.p, nullptr, nullptr};
    HIP_TRY(launch_rp_ranges((const Record*)records.p, n_rec, (uint64_t*)rec_first.p, RpRoute{nullptr, nullptr, nullptr, nullptr, nullptr}, n, st));
    HIP_TRY(launch_rp_pass(r->case_mode == AM_IGNORE_CASE, r->t, (const uint8_t*)b->d_text, b->d_offsets, (const Record*)records.p, (const uint64_t*)rec_first.p,
                           (const int64_t*)thr.p, UINT64_MAX, (RpKept*)kept.p, (RpHay*)hs.p, route, n, 1u, st));
    HIP_TRY(launch_scan(tmp.p, tmp_bytes, (const uint32_t*)nk.p, (uint64_t*)off.p, n1, st));
    uint64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, (uint64_t*)off.p + n, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    AM_TRY(out.ensure((total + 1) * sizeof(RpSelected)));
    HIP_TRY(launch_rp_gather((const RpHay*)hs.p, (const uint64_t*)rec_first.p, (const RpKept*)kept.p, (const uint64_t*)off.p, (RpSelected*)out.p, (int64_t*)best.p, n, st));
    am_prio_match* host = (am_prio_match*)std::malloc((total ? total : 1) * sizeof(am_prio_match));
    if (!host) return fail(AM_ERR_OOM, "malloc(matches) failed");
    hipError_t e = hipMemcpyAsync(best_out, best.p, (size_t)n * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && total) e = hipMemcpyAsync(host, out.p, total * sizeof(am_prio_match), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { std::free(host); return fail(AM_ERR_HIP, hipGetErrorString(e)); }
    *matches_out = host; *n_matches_out = (size_t)total;
    return AM_OK;
}

extern "C" void am_prio_matches_free(am_prio_match* m) { std::free(m); }

extern "C" uint64_t am_replaced_size(const am_replaced* r) { return r ? r->text.size() : 0; }
extern "C" uint64_t am_replaced_passes(const am_replaced* r) { return r ? r->passes : 0; }
extern "C" uint64_t am_replaced_scanned_bytes(const am_replaced* r) { return r ? r->scanned : 0; }
extern "C" uint64_t am_replaced_spliced_bytes(const am_replaced* r) { return r ? r->spliced : 0; }

extern "C" int am_replaced_get(const am_replaced* r, size_t i, const uint8_t** ptr, size_t* len)
{
    if (!r || i >= r->text.size() || !ptr || !len) return fail(AM_ERR_INVALID, "bad argument");
    *ptr = r->text[i].p ? r->text[i].p : (const uint8_t*)""; *len = r->text[i].len;
    return r->just[i] ? 1 : 0;
}

extern "C" int am_replaced_device(const am_replaced* r) { return r ? r->dev : -1; }

// copies text i to host memory, wherever the result lives
extern "C" int am_replaced_read(const am_replaced* r, size_t i, uint8_t* dst, size_t cap, size_t* len)
{
    if (!r || i >= r->text.size()) return fail(AM_ERR_INVALID, "index out of range");
    if (len) *len = r->just[i] ? r->text[i].len : 0;
    if (!r->just[i]) return 0;
    const size_t n = r->text[i].len;
    if (n > cap || (n && !dst)) return fail(AM_ERR_INVALID, "destination too small");
    if (n == 0) return 1;
    if (r->dev < 0) { std::memcpy(dst, r->text[i].p, n); return 1; }
    ON_DEVICE(r->dev);
    HIP_TRY(hipMemcpy(dst, r->text[i].p, n, hipMemcpyDeviceToHost));
    return 1;
}

extern "C" void am_replaced_free(am_replaced* r) { delete r; }
