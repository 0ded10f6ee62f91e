// facade.cpp -- flat C entry points over the C++ host mirror (automaton.hpp / searcher.hpp /
// replacer.hpp) so that Python (ctypes) tests and bench.py can drive it.  Values are uint32 handles.
#include <cstring>
#include <memory>
#include <string>

#include "replacer.hpp"
#include "splitter.hpp"

using namespace alfred_margaret;

namespace {
thread_local std::string g_err;
struct MachineBox {
    AcMachine<uint32_t> m;
    std::vector<uint64_t> valuesOff; std::vector<uint32_t> valuesFlat;   // flattened machineValues for inspection
};
std::vector<Text> sliceTexts(const am_slice* hay, size_t n)
{
    std::vector<Text> t(n);
    for (size_t i = 0; i < n; i++) t[i] = Text(hay[i].ptr, hay[i].off, hay[i].len);
    return t;
}
template <class Fn> int guarded(Fn fn)
{
    try { fn(); return 0; }
    catch (const AmError& e) { g_err = e.what(); return e.code; }
    catch (const std::exception& e) { g_err = e.what(); return AM_ERR_INVALID; }
}
}  // namespace

extern "C" {

const char* amh_last_error(void) { return g_err.c_str(); }

// Automaton.build; values[i] (or i when values == NULL) is the payload handle of needle i
// (lower_from / lower_to / n_pairs: the caller's lower-case table, am_automaton_create_ex; null = built-in)
int amh_build_ex(const uint8_t* bytes, const uint64_t* offs, size_t n, const uint32_t* values, const uint32_t* lower_from, const uint32_t* lower_to,
                 size_t n_pairs, void** out)
{
    *out = nullptr;
    return guarded([&] {
        std::vector<std::pair<Text, uint32_t>> nv(n);
        for (size_t i = 0; i < n; i++) nv[i] = {Text(bytes, (size_t)offs[i], (size_t)(offs[i + 1] - offs[i])), values ? values[i] : (uint32_t)i};
        std::unique_ptr<utf8::LowerTable> lt;
        if (lower_from && lower_to) lt.reset(new utf8::LowerTable(lower_from, lower_to, n_pairs));
        auto* box = new MachineBox{build(nv, lt.get()), {}, {}};
        box->valuesOff.assign(1, 0);
        for (auto& vs : box->m.machineValues) { box->valuesFlat.insert(box->valuesFlat.end(), vs.begin(), vs.end()); box->valuesOff.push_back(box->valuesFlat.size()); }
        *out = box;
    });
}
int amh_build(const uint8_t* bytes, const uint64_t* offs, size_t n, const uint32_t* values, void** out)
{
    return amh_build_ex(bytes, offs, n, values, nullptr, nullptr, 0, out);
}
void amh_free(void* h) { delete static_cast<MachineBox*>(h); }
size_t amh_num_states(void* h) { return static_cast<MachineBox*>(h)->m.numStates(); }
size_t amh_num_transitions(void* h) { return static_cast<MachineBox*>(h)->m.machineTransitions.size(); }
const uint64_t* amh_transitions(void* h) { return static_cast<MachineBox*>(h)->m.machineTransitions.data(); }
const uint32_t* amh_offsets(void* h) { return static_cast<MachineBox*>(h)->m.machineOffsets.data(); }
const uint64_t* amh_root_ascii(void* h) { return static_cast<MachineBox*>(h)->m.machineRootAsciiTransitions.data(); }
const uint64_t* amh_values_off(void* h) { return static_cast<MachineBox*>(h)->valuesOff.data(); }
const uint32_t* amh_values(void* h) { return static_cast<MachineBox*>(h)->valuesFlat.data(); }
am_automaton* amh_device(void* h) { return static_cast<MachineBox*>(h)->m.device.get(); }

// runWithCase with a list-building fold over a batch: triples (haystack, matchPos, value) in fold
// order.  Call with cap = 0 to get the count.
int amh_run_list(void* h, int case_mode, const am_slice* hay, size_t n_hay, uint32_t* hay_out, uint64_t* pos_out, uint32_t* val_out,
                 uint64_t cap, uint64_t* n_out)
{
    return guarded([&] {
        struct Acc { std::vector<std::pair<uint64_t, uint32_t>> v; };
        auto f = [](Acc a, const Match<uint32_t>& m) { a.v.emplace_back(m.matchPos, m.matchValue); return Next<Acc>::Step(std::move(a)); };
        auto accs = runBatchWithCase((CaseSensitivity)case_mode, Acc{}, f, static_cast<MachineBox*>(h)->m, sliceTexts(hay, n_hay));
        uint64_t k = 0;
        for (size_t i = 0; i < accs.size(); i++)
            for (auto& pv : accs[i].v) { if (k < cap) { hay_out[k] = (uint32_t)i; pos_out[k] = pv.first; val_out[k] = pv.second; } k++; }
        *n_out = k;
    });
}

// countMatches (benchmark/haskell/app/Main.hs:67-76) per haystack
int amh_count(void* h, int case_mode, const am_slice* hay, size_t n_hay, uint64_t* counts_out)
{
    return guarded([&] { amCheck(am_count(static_cast<MachineBox*>(h)->m.device.get(), case_mode, hay, n_hay, counts_out)); });
}

// ---- Searcher
int amh_searcher_build(int case_mode, const uint8_t* bytes, const uint64_t* offs, size_t n, void** out)
{
    *out = nullptr;
    return guarded([&] {
        std::vector<std::string> ns(n);
        for (size_t i = 0; i < n; i++) ns[i].assign((const char*)bytes + offs[i], (size_t)(offs[i + 1] - offs[i]));
        *out = new Searcher<int>(buildNeedleIdSearcher((CaseSensitivity)case_mode, ns));
    });
}
void amh_searcher_free(void* s) { delete static_cast<Searcher<int>*>(s); }
void amh_searcher_set_case(void* s, int case_mode) { static_cast<Searcher<int>*>(s)->setCaseSensitivity((CaseSensitivity)case_mode); }
int amh_searcher_contains_any(void* s, const am_slice* hay, size_t n_hay, uint8_t* out)
{
    return guarded([&] { auto r = containsAnyBatch(*static_cast<Searcher<int>*>(s), sliceTexts(hay, n_hay)); for (size_t i = 0; i < n_hay; i++) out[i] = r[i]; });
}
int amh_searcher_contains_all(void* s, const am_slice* hay, size_t n_hay, uint8_t* out)
{
    return guarded([&] { auto r = containsAllBatch(*static_cast<Searcher<int>*>(s), sliceTexts(hay, n_hay)); for (size_t i = 0; i < n_hay; i++) out[i] = r[i]; });
}

int amh_searcher_contains_all_host_fold(void* s, const am_slice* hay, size_t n_hay, uint8_t* out)
{
    return guarded([&] { auto r = containsAllBatchHostFold(*static_cast<Searcher<int>*>(s), sliceTexts(hay, n_hay)); for (size_t i = 0; i < n_hay; i++) out[i] = r[i]; });
}

// ---- Replacer
int amh_replacer_build_ex(int case_mode, const uint8_t* nbytes, const uint64_t* noffs, const uint8_t* rbytes, const uint64_t* roffs, size_t n,
                          const uint32_t* lower_from, const uint32_t* lower_to, size_t n_pairs, void** out)
{
    *out = nullptr;
    return guarded([&] {
        std::unique_ptr<utf8::LowerTable> lt;
        if (lower_from && lower_to) lt.reset(new utf8::LowerTable(lower_from, lower_to, n_pairs));
        std::vector<std::pair<std::string, std::string>> pairs(n);
        for (size_t i = 0; i < n; i++) {
            pairs[i].first.assign((const char*)nbytes + noffs[i], (size_t)(noffs[i + 1] - noffs[i]));
            pairs[i].second.assign((const char*)rbytes + roffs[i], (size_t)(roffs[i + 1] - roffs[i]));
        }
        *out = new Replacer((CaseSensitivity)case_mode, pairs, lt.get());
    });
}
int amh_replacer_build(int case_mode, const uint8_t* nbytes, const uint64_t* noffs, const uint8_t* rbytes, const uint64_t* roffs, size_t n, void** out)
{
    return amh_replacer_build_ex(case_mode, nbytes, noffs, rbytes, roffs, n, nullptr, nullptr, 0, out);
}
void amh_replacer_free(void* r) { delete static_cast<Replacer*>(r); }
// mapReplacement with the new replacements given as a list (needle i gets rbytes[roffs[i] .. roffs[i+1]))
int amh_replacer_with_replacements(void* r, const uint8_t* rbytes, const uint64_t* roffs, void** out)
{
    *out = nullptr;
    return guarded([&] {
        const Replacer* src = static_cast<Replacer*>(r);
        size_t i = 0;
        std::vector<std::string> fresh(src->searcher().numNeedles());
        for (auto& f : fresh) { f.assign((const char*)rbytes + roffs[i], (size_t)(roffs[i + 1] - roffs[i])); i++; }
        // mapReplacement sees the payloads in needle order for `needles`, and per state for the automaton: map by priority
        *out = new Replacer(src->mapReplacementIndexed([&](size_t needle, const std::string&) { return fresh[needle]; }));
    });
}
// Replacer.compose (Replacer.hs:120-133): *out = null (and AM_OK) when the case sensitivities differ (the reference's Nothing)
int amh_replacer_compose(void* r1, void* r2, void** out)
{
    *out = nullptr;
    return guarded([&] {
        std::optional<Replacer> c = Replacer::compose(*static_cast<Replacer*>(r1), *static_cast<Replacer*>(r2));
        if (c) *out = new Replacer(std::move(*c));
    });
}
int amh_replacer_set_case(void* r, int case_mode, void** out)
{
    *out = nullptr;
    return guarded([&] { *out = new Replacer(static_cast<Replacer*>(r)->setCaseSensitivity((CaseSensitivity)case_mode)); });
}
// Runs a batch; results are returned as one malloc'd blob + offsets (n+1); is_nothing[i] = 1 where the
// reference returns Nothing.  max_len < 0 = maxBound.
static int replacer_run_batch(void* r, bool host_splice, const am_slice* hay, size_t n_hay, long long max_len, uint8_t** blob_out, uint64_t* offs_out, uint8_t* is_nothing)
{
    *blob_out = nullptr;
    return guarded([&] {
        std::vector<std::string> in(n_hay);
        for (size_t i = 0; i < n_hay; i++) in[i].assign((const char*)hay[i].ptr + hay[i].off, hay[i].len);
        const size_t lim = max_len < 0 ? SIZE_MAX : (size_t)max_len;
        auto* rp = static_cast<Replacer*>(r);
        auto res = host_splice ? rp->runBatchWithLimitHostSplice(in, lim) : rp->runBatchWithLimit(in, lim);
        uint64_t total = 0;
        for (size_t i = 0; i < n_hay; i++) { offs_out[i] = total; is_nothing[i] = !res[i].has_value(); if (res[i]) total += res[i]->size(); }
        offs_out[n_hay] = total;
        uint8_t* blob = (uint8_t*)malloc(total ? total : 1);
        for (size_t i = 0; i < n_hay; i++) if (res[i] && !res[i]->empty()) std::memcpy(blob + offs_out[i], res[i]->data(), res[i]->size());
        *blob_out = blob;
    });
}
int amh_replacer_run_batch(void* r, const am_slice* hay, size_t n_hay, long long max_len, uint8_t** blob_out, uint64_t* offs_out, uint8_t* is_nothing)
{
    return replacer_run_batch(r, false, hay, n_hay, max_len, blob_out, offs_out, is_nothing);
}
// same function with only the scans on the GPU (cross-check of the device passes)
int amh_replacer_run_batch_host_splice(void* r, const am_slice* hay, size_t n_hay, long long max_len, uint8_t** blob_out, uint64_t* offs_out, uint8_t* is_nothing)
{
    return replacer_run_batch(r, true, hay, n_hay, max_len, blob_out, offs_out, is_nothing);
}
// the am_replacer* (flattened Searcher Payload in HBM) behind this Replacer, for callers that keep their batches on the device
int amh_replacer_device(void* r, const void** out)
{
    *out = nullptr;
    return guarded([&] { *out = static_cast<Replacer*>(r)->device(); });
}
void amh_replacer_last_stats(void* r, uint64_t* passes, uint64_t* scanned_bytes)
{
    const auto st = static_cast<Replacer*>(r)->lastStats();
    *passes = st.passes; *scanned_bytes = st.scannedBytes;
}
void amh_free_blob(uint8_t* p) { free(p); }

// ---- Splitter
int amh_splitter_build(const uint8_t* sep, size_t len, void** out)
{
    *out = nullptr;
    return guarded([&] { *out = new Splitter(std::string((const char*)sep, len)); });
}
void amh_splitter_free(void* s) { delete static_cast<Splitter*>(s); }
// fragments of all haystacks concatenated: blob + fragment offsets (n_frag + 1) + per-haystack fragment counts
int amh_splitter_split_batch(void* sp, int ignore_case, const am_slice* hay, size_t n_hay, uint8_t** blob_out, uint64_t** offs_out,
                             uint64_t* n_frag_out, uint32_t* frags_per_hay)
{
    *blob_out = nullptr; *offs_out = nullptr;
    return guarded([&] {
        std::vector<std::string> in(n_hay);
        for (size_t i = 0; i < n_hay; i++) in[i].assign((const char*)hay[i].ptr + hay[i].off, hay[i].len);
        auto res = static_cast<Splitter*>(sp)->splitBatch(in, ignore_case != 0);
        uint64_t nf = 0, total = 0;
        for (auto& r : res) { nf += r.size(); for (auto& f : r) total += f.size(); }
        uint8_t* blob = (uint8_t*)malloc(total ? total : 1);
        uint64_t* offs = (uint64_t*)malloc((nf + 1) * sizeof(uint64_t));
        uint64_t k = 0, at = 0;
        for (size_t i = 0; i < n_hay; i++) {
            frags_per_hay[i] = (uint32_t)res[i].size();
            for (auto& f : res[i]) { offs[k++] = at; if (!f.empty()) std::memcpy(blob + at, f.data(), f.size()); at += f.size(); }
        }
        offs[k] = at;
        *blob_out = blob; *offs_out = offs; *n_frag_out = nf;
    });
}
void amh_free_u64(uint64_t* p) { free(p); }

// ---- Utf8 helpers
int64_t amh_skip_code_points_backwards(const uint8_t* d, size_t len, size_t index, size_t n)
{
    try { return (int64_t)utf8::skipCodePointsBackwards(Text(d, 0, len), index, n); } catch (...) { return -1; }
}
size_t amh_lower_utf8(const uint8_t* d, size_t len, uint8_t* out, size_t cap)
{
    std::string s = utf8::lowerUtf8(Text(d, 0, len));
    if (s.size() <= cap) std::memcpy(out, s.data(), s.size());
    return s.size();
}

// Utf8.isCaseInvariant (Utf8.hs:169-171): 1 / 0; -1 on error
int amh_is_case_invariant(const uint8_t* d, size_t len, const uint32_t* lower_from, const uint32_t* lower_to, size_t n_pairs)
{
    try {
        std::unique_ptr<utf8::LowerTable> lt;
        if (lower_from && lower_to) lt.reset(new utf8::LowerTable(lower_from, lower_to, n_pairs));
        return utf8::isCaseInvariant(Text(d, 0, len), lt.get()) ? 1 : 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// Automaton.needleCasings (Automaton.hs:555-566): the casings concatenated + their offsets (n + 1); free with amh_free_blob / amh_free_u64
int amh_needle_casings(const uint8_t* d, size_t len, const uint32_t* lower_from, const uint32_t* lower_to, size_t n_pairs, uint8_t** blob_out, uint64_t** offs_out, uint64_t* n_out)
{
    *blob_out = nullptr; *offs_out = nullptr; *n_out = 0;
    return guarded([&] {
        std::unique_ptr<utf8::LowerTable> lt;
        if (lower_from && lower_to) lt.reset(new utf8::LowerTable(lower_from, lower_to, n_pairs));
        const std::vector<std::string> cs = needleCasings(Text(d, 0, len), lt.get());
        uint64_t total = 0;
        for (auto& c : cs) total += c.size();
        uint8_t* blob = (uint8_t*)malloc(total ? total : 1);
        uint64_t* offs = (uint64_t*)malloc((cs.size() + 1) * sizeof(uint64_t));
        uint64_t at = 0;
        for (size_t i = 0; i < cs.size(); i++) { offs[i] = at; if (!cs[i].empty()) std::memcpy(blob + at, cs[i].data(), cs[i].size()); at += cs[i].size(); }
        offs[cs.size()] = at;
        *blob_out = blob; *offs_out = offs; *n_out = cs.size();
    });
}

}  // extern "C"
