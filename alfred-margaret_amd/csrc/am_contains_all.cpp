// am_contains_all.cpp -- Searcher.containsAll (Searcher.hs:167-187) and the fold checksum of a result, on the device.
#include "am_host.h"

using namespace am;
using namespace am::dev;
using namespace am::host;

// ------------------------------------------------------------------ Searcher.containsAll (Searcher.hs:167-187)

struct am_needle_ids {
    const am_automaton* a = nullptr;
    uint32_t n_needles = 0;
    DevBuf vals_off, vals;
};

extern "C" int am_needle_ids_create(const am_automaton* a, const uint64_t* values_offsets, const uint32_t* values, uint32_t n_needles, am_needle_ids** out)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!a) return fail(AM_ERR_INVALID, "null automaton");
    AM_TRY(ensure_runtime());
    ON_DEVICE(a->dev);
    // a handle attached to a received image (multi-GPU ranks) has no reference arrays: the state count comes from the image
    uint64_t n_states = 0;
    if (a->has_ref) n_states = a->offsets.size() - 1;
    else {
        std::lock_guard<std::mutex> lk(const_cast<am_automaton*>(a)->mu);
        for (const Flavor& f : a->fl) if (f.ready) n_states = f.h.n_states;
        if (!n_states) return fail(AM_ERR_INVALID, "automaton handle has no image");
    }
    if (!values_offsets || values_offsets[0] != 0) return fail(AM_ERR_INVALID, "values_offsets[0] must be 0");
    for (uint64_t s = 0; s < n_states; s++)
        if (values_offsets[s + 1] < values_offsets[s] || (a->has_ref && values_offsets[s + 1] - values_offsets[s] != a->values_len[s]))
            return fail(AM_ERR_INVALID, "values_offsets disagrees with the values_len given to am_automaton_create");
    const uint64_t n_values = values_offsets[n_states];
    if (n_values && !values) return fail(AM_ERR_INVALID, "values is null");
    am_needle_ids* ids = new am_needle_ids();
    ids->a = a; ids->n_needles = n_needles;
    int rc = ids->vals_off.ensure((n_states + 1) * 8);
    if (rc == AM_OK) rc = ids->vals.ensure(n_values * 4 + 4);
    if (rc == AM_OK && hipMemcpy(ids->vals_off.p, values_offsets, (n_states + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) rc = fail(AM_ERR_HIP, "upload failed");
    if (rc == AM_OK && n_values && hipMemcpy(ids->vals.p, values, n_values * 4, hipMemcpyHostToDevice) != hipSuccess) rc = fail(AM_ERR_HIP, "upload failed");
    if (rc != AM_OK) { am_needle_ids_destroy(ids); return rc; }
    *out = ids;
    return AM_OK;
}

extern "C" void am_needle_ids_destroy(am_needle_ids* ids)
{
    if (!ids) return;
    ids->vals_off.release(); ids->vals.release();
    delete ids;
}

extern "C" int am_contains_all_batch(const am_needle_ids* ids, int case_mode, const am_batch* cb, uint8_t* flags_out)
{
    if (!ids || !cb) return fail(AM_ERR_INVALID, "null needle ids or batch");
    am_batch* b = const_cast<am_batch*>(cb);
    const uint32_t n_hay = b->n_hay;
    if (n_hay && !flags_out) return fail(AM_ERR_INVALID, "flags_out is null");
    if (ids->n_needles == 0) { if (n_hay) std::memset(flags_out, 1, n_hay); return AM_OK; }     // IS.null of the empty set (Searcher.hs:176,184)
    if (n_hay == 0) return AM_OK;
    ON_DEVICE(b->dev);
    hipStream_t st; AM_TRY(get_stream(b->dev, &st));
    DevBuf records, rec_first, bits, flags;
    struct Release { DevBuf &a, &b, &c, &d; ~Release() { a.release(); b.release(); c.release(); d.release(); } } rel{records, rec_first, bits, flags};
    const uint32_t words = (ids->n_needles + 31) / 32;
    // The direct route (round 5): the scan itself sets the bit of every needle id it reports -- no record is written -- and a haystack whose set is
    // complete is not looked at any further (`Done`, Searcher.hs:181), like containsAny's first match.  One bitmap row per haystack, up to 8 GiB of them;
    // wider batches, the general kernel and automata with the empty needle fold the records (below).
    if ((uint64_t)n_hay * words * 4 <= (8ull << 30) && !cfg::on(cfg::kNoIdsScan)) {
        AM_TRY(bits.ensure((uint64_t)n_hay * words * 4 + 64));
        AM_TRY(flags.ensure((size_t)n_hay * 4 + 64));                  // (here: the haystacks' missing-id counters)
        bool taken = false;
        AM_TRY(scan_needle_ids(ids->a, case_mode, b, (const uint64_t*)ids->vals_off.p, (const uint32_t*)ids->vals.p, ids->n_needles, (uint32_t*)bits.p, words,
                               (uint32_t*)flags.p, flags_out, &taken));
        if (taken) return AM_OK;
    }
    uint64_t n_rec = 0;
    auto sink = [&](uint64_t n, Record** ptr) -> int { AM_TRY(records.ensure(n * sizeof(Record))); *ptr = (Record*)records.p; return AM_OK; };
    AM_TRY(run_records(ids->a, case_mode, b, sink, &n_rec));
    if (n_rec == 0) { std::memset(flags_out, 0, n_hay); return AM_OK; }
    // one bitmap row per haystack; very wide batches go through in groups of haystacks (records are sorted by haystack)
    const uint64_t budget = 1ull << 30;
    const uint32_t group = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_hay, budget / ((uint64_t)words * 4)));
    AM_TRY(rec_first.ensure(((uint64_t)n_hay + 1) * 8));
    AM_TRY(bits.ensure((uint64_t)group * words * 4));
    AM_TRY(flags.ensure(n_hay));
    HIP_TRY(launch_rp_ranges((const Record*)records.p, n_rec, (uint64_t*)rec_first.p, RpRoute{nullptr, nullptr, nullptr, nullptr, nullptr}, n_hay, st));
    std::vector<uint64_t> first;
    if (group < n_hay) {
        first.resize((size_t)n_hay + 1);
        HIP_TRY(hipMemcpyAsync(first.data(), rec_first.p, first.size() * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    for (uint32_t h0 = 0; h0 < n_hay; h0 += group) {
        const uint32_t h1 = std::min<uint64_t>(n_hay, (uint64_t)h0 + group);
        const uint64_t r0 = first.empty() ? 0 : first[h0], r1 = first.empty() ? n_rec : first[h1];
        HIP_TRY(hipMemsetAsync(bits.p, 0, (uint64_t)(h1 - h0) * words * 4, st));
        { Prof pr("idset", st);
          HIP_TRY(launch_idset((const Record*)records.p, r0, r1, (const uint64_t*)ids->vals_off.p, (const uint32_t*)ids->vals.p, ids->n_needles, h0, words, (uint32_t*)bits.p, st));
          HIP_TRY(launch_idset_all((const uint32_t*)bits.p, words, ids->n_needles, h1 - h0, (uint8_t*)flags.p + h0, st)); }
    }
    HIP_TRY(hipMemcpyAsync(flags_out, flags.p, n_hay, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return AM_OK;
}

extern "C" int am_matches_fold_hash(const am_matches* m, const am_needle_ids* ids, size_t n_hay, uint64_t* hash_out, uint64_t* count_out)
{
    if (!m || !ids) return fail(AM_ERR_INVALID, "null matches or values table");
    if (n_hay && !hash_out) return fail(AM_ERR_INVALID, "hash_out is null");
    if (n_hay >= 0xFFFFFFFFull) return fail(AM_ERR_INVALID, "too many haystacks");
    if (n_hay == 0) return AM_OK;
    if (m->n && !m->d_records) return fail(AM_ERR_UNSUPPORTED, "am_matches_fold_hash: the result was assembled on the host (am_run on a large host batch) and has no records in HBM");
    AM_TRY(ensure_runtime());
    if (m->dev != ids->a->dev) return fail(AM_ERR_INVALID, "result and values table live on different devices");
    ON_DEVICE(m->dev);
    hipStream_t st; AM_TRY(get_stream(m->dev, &st));
    DevBuf rec_first, out, dummy;
    struct Release { DevBuf &a, &b, &c; ~Release() { a.release(); b.release(); c.release(); } } rel{rec_first, out, dummy};
    AM_TRY(rec_first.ensure((n_hay + 1) * 8));
    AM_TRY(out.ensure(n_hay * 16));
    AM_TRY(dummy.ensure(sizeof(Record)));
    const Record* recs = m->n ? m->d_records + m->first : (const Record*)dummy.p;
    HIP_TRY(launch_rp_ranges(recs, m->n, (uint64_t*)rec_first.p, RpRoute{nullptr, nullptr, nullptr, nullptr, nullptr}, (uint32_t)n_hay, st));
    { Prof pr("fold_hash", st);
      HIP_TRY(launch_fold_hash(recs, (const uint64_t*)rec_first.p, (const uint64_t*)ids->vals_off.p, (const uint32_t*)ids->vals.p, (uint32_t)n_hay,
                               (uint64_t*)out.p, (uint64_t*)out.p + n_hay, st)); }
    HIP_TRY(hipMemcpyAsync(hash_out, out.p, n_hay * 8, hipMemcpyDeviceToHost, st));
    if (count_out) HIP_TRY(hipMemcpyAsync(count_out, (uint64_t*)out.p + n_hay, n_hay * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return AM_OK;
}

extern "C" int am_contains_all(const am_needle_ids* ids, int case_mode, const am_slice* hay, size_t n_hay, uint8_t* flags_out)
{
    if (!ids) return fail(AM_ERR_INVALID, "null needle ids");
    ON_DEVICE(ids->a->dev);
    am_batch* b = nullptr;
    AM_TRY(am_batch_upload(hay, n_hay, &b));
    const int rc = am_contains_all_batch(ids, case_mode, b, flags_out);
    am_batch_destroy(b);
    return rc;
}
