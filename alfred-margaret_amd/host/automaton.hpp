// automaton.hpp -- host mirror of Data.Text.AhoCorasick.Automaton (reference:
// src/Data/Text/AhoCorasick/Automaton.hs).  Same names, same argument meaning:
//   build        :176-200   needles with values -> AcMachine (packed exactly like the reference)
//   runWithCase  :442-534   strict left fold over matches with early exit (Next = Done | Step)
//   runText/runLower :539-553
// The only difference from the reference is WHERE the body of runWithCase executes: the match
// positions come from libam (HIP kernels on the MI355X) through the C ABI in include/am.h, and the
// fold function is applied on the host to the returned records, in the reference's order.
// There is no host matching loop in this file.
#pragma once
#include <algorithm>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "am.h"
#include "utf8.hpp"

namespace alfred_margaret {

using utf8::Text;

enum class CaseSensitivity { CaseSensitive = AM_CASE_SENSITIVE, IgnoreCase = AM_IGNORE_CASE };   // CaseSensitivity.hs:14-22

using CodeUnitIndex = size_t;   // Utf8.hs:106-114

template <class V> struct Match { CodeUnitIndex matchPos; const V& matchValue; };   // Automaton.hs:98-105

template <class A> struct Next {   // Automaton.hs:398
    bool done; A value;
    static Next Done(A a) { return Next{true, std::move(a)}; }
    static Next Step(A a) { return Next{false, std::move(a)}; }
};

struct AmError : std::runtime_error {
    int code;
    AmError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};
inline void amCheck(int rc) { if (rc != AM_OK) throw AmError(rc, std::string("libam: ") + am_last_error()); }

namespace detail {

constexpr uint64_t kWildcard = 0x200000ull;   // Automaton.hs:130-131

// The packed automaton without the values (Automaton.hs:108-123) + which needles end where.
struct Packed {
    std::vector<uint64_t> transitions;       // machineTransitions
    std::vector<uint32_t> offsets;           // machineOffsets (numStates + 1)
    std::vector<uint64_t> rootAscii;         // machineRootAsciiTransitions (128)
    std::vector<std::vector<uint32_t>> valueIdx;   // per state: needle indices in report order
};

// (state << 21 | code point) -> next state: open addressing over two flat arrays (the IntMap-of-IntMaps of the reference, :249-292, is what `build` spends its time in;
// a node-based map made a 100k-needle build 0.65 s where this takes a third)
struct EdgeMap {
    std::vector<uint64_t> keys; std::vector<uint32_t> vals; size_t n = 0, mask = 0;
    static constexpr uint64_t kEmpty = ~0ull;
    explicit EdgeMap(size_t expect) { size_t cap = 1024; while (cap < 2 * expect) cap <<= 1; keys.assign(cap, kEmpty); vals.assign(cap, 0); mask = cap - 1; }
    static size_t slot(uint64_t k) { k *= 0x9E3779B97F4A7C15ull; return (size_t)(k ^ (k >> 29)); }
    // the value of k, or -1
    int64_t find(uint64_t k) const
    {
        for (size_t i = slot(k) & mask;; i = (i + 1) & mask) { if (keys[i] == k) return vals[i]; if (keys[i] == kEmpty) return -1; }
    }
    void insert(uint64_t k, uint32_t v)          // (k is not in the map)
    {
        if (2 * (n + 1) > keys.size()) {
            std::vector<uint64_t> ok; std::vector<uint32_t> ov; ok.swap(keys); ov.swap(vals);
            keys.assign(ok.size() * 2, kEmpty); vals.assign(ok.size() * 2, 0); mask = keys.size() - 1;
            for (size_t i = 0; i < ok.size(); i++) if (ok[i] != kEmpty) { size_t j = slot(ok[i]) & mask; while (keys[j] != kEmpty) j = (j + 1) & mask; keys[j] = ok[i]; vals[j] = ov[i]; }
        }
        size_t i = slot(k) & mask;
        while (keys[i] != kEmpty) i = (i + 1) & mask;
        keys[i] = k; vals[i] = v; n++;
    }
};

// Automaton.hs:176-200 build and its helpers (:249-292, :336-362, :367-380, :166-172, :301-306).
// State ids are allocated in needle order, depth first along each needle, exactly as the reference.
inline Packed buildPacked(const std::vector<Text>& needles)
{
    // flat arrays throughout (a vector per state -- children, own needles -- was most of the time of a 100k-needle build):
    // the goto edges in creation order, the needles that end at a state as a linked list, newest first (insertWith (++), :263)
    struct Edge { uint32_t src, cp, dst; };
    constexpr uint32_t kNoNeedle = 0xFFFFFFFFu;
    size_t total_len = 0;
    for (const Text& t : needles) total_len += t.len;
    EdgeMap edge(total_len + 16);                   // (state << 21 | cp) -> next; at most one edge per needle byte
    std::vector<Edge> edges; edges.reserve(total_len + 16);
    std::vector<uint32_t> own_head(1, kNoNeedle), own_next(needles.size(), kNoNeedle);
    own_head.reserve(total_len + 16);
    for (size_t i = 0; i < needles.size(); i++) {
        const uint8_t* d = needles[i].begin(); const size_t n = needles[i].len;
        uint32_t s = 0;
        for (size_t k = 0; k < n;) {
            size_t u; const uint32_t cp = utf8::decodeAt(d, k, n, u);
            const uint64_t key = ((uint64_t)s << 21) | cp;
            const int64_t hit = edge.find(key);
            if (hit >= 0) s = (uint32_t)hit;
            else {
                const uint32_t nx = (uint32_t)own_head.size();
                own_head.push_back(kNoNeedle);
                edges.push_back(Edge{s, cp, nx});
                edge.insert(key, nx);
                s = nx;
            }
            k += u;
        }
        own_next[i] = own_head[s]; own_head[s] = (uint32_t)i;      // newest first
    }
    const size_t S = own_head.size();
    // a state's children: its run of `kids`, ascending code point
    std::vector<uint32_t> kid_first(S + 1, 0);
    for (const Edge& e : edges) kid_first[e.src + 1]++;
    for (size_t s = 0; s < S; s++) kid_first[s + 1] += kid_first[s];
    std::vector<std::pair<uint32_t, uint32_t>> kids(edges.size());           // (code point, next)
    {
        std::vector<uint32_t> cur(kid_first.begin(), kid_first.end() - 1);
        for (const Edge& e : edges) kids[cur[e.src]++] = {e.cp, e.dst};
    }
    for (size_t s = 0; s < S; s++) if (kid_first[s + 1] - kid_first[s] > 1) std::sort(kids.begin() + kid_first[s], kids.begin() + kid_first[s + 1]);
    auto findKid = [&](uint32_t s, uint32_t cp) -> int64_t { return edge.find(((uint64_t)s << 21) | cp); };
    // level order: fallbacks (:336-362) and values (:367-380) only look at shallower states
    std::vector<uint32_t> order; order.reserve(S); order.push_back(0);
    std::vector<uint32_t> fallback(S, 0);
    Packed p;
    p.valueIdx.resize(S);
    for (uint32_t i = own_head[0]; i != kNoNeedle; i = own_next[i]) p.valueIdx[0].push_back(i);
    for (size_t q = 0; q < order.size(); q++) {
        const uint32_t s = order[q];
        for (uint32_t e = kid_first[s]; e < kid_first[s + 1]; e++) {
            const uint32_t cp = kids[e].first, nx = kids[e].second;
            uint32_t fb = 0;
            for (uint32_t t = s; t != 0;) {               // getFallback (:342-352)
                const uint32_t f = fallback[t];
                const int64_t j = findKid(f, cp);
                if (j >= 0) { fb = (uint32_t)j; break; }
                t = f;
            }
            fallback[nx] = fb;
            if (own_head[nx] != kNoNeedle || !p.valueIdx[fb].empty()) {          // values = own ++ the fallback's (:367-380)
                std::vector<uint32_t>& v = p.valueIdx[nx];
                for (uint32_t i = own_head[nx]; i != kNoNeedle; i = own_next[i]) v.push_back(i);
                v.insert(v.end(), p.valueIdx[fb].begin(), p.valueIdx[fb].end());
            }
            order.push_back(nx);
        }
    }
    // makeTransitions + packTransitions (:190-192, :166-172): descending code point, wildcard last
    p.offsets.resize(S + 1);
    p.transitions.reserve(edges.size() + S);
    for (size_t s = 0; s < S; s++) {
        p.offsets[s] = (uint32_t)p.transitions.size();
        for (uint32_t e = kid_first[s + 1]; e-- > kid_first[s];) p.transitions.push_back(((uint64_t)kids[e].second << 32) | kids[e].first);
        p.transitions.push_back(((uint64_t)fallback[s] << 32) | kWildcard);
    }
    p.offsets[S] = (uint32_t)p.transitions.size();
    p.rootAscii.assign(128, kWildcard);                    // wildcard -> state 0 (:301-306)
    for (uint32_t e = kid_first[0]; e < kid_first[1]; e++) if (kids[e].first < 128) p.rootAscii[kids[e].first] = ((uint64_t)kids[e].second << 32) | kids[e].first;
    return p;
}

struct AutomatonDeleter { void operator()(am_automaton* a) const { am_automaton_destroy(a); } };

}  // namespace detail

// Automaton.hs:108-123 AcMachine v
template <class V> struct AcMachine {
    std::vector<std::vector<V>> machineValues;
    std::vector<uint64_t> machineTransitions;
    std::vector<uint32_t> machineOffsets;
    std::vector<uint64_t> machineRootAsciiTransitions;
    std::shared_ptr<am_automaton> device;     // flattened copy in HBM, owned by libam

    size_t numStates() const { return machineValues.size(); }
};

// Automaton.hs:176  build :: [(Text, v)] -> AcMachine v
// lower (optional): the caller's lower-casing for IgnoreCase runs (am_automaton_create_ex); null = libam's built-in table
template <class V> AcMachine<V> build(const std::vector<std::pair<Text, V>>& needlesWithValues, const utf8::LowerTable* lower = nullptr)
{
    std::vector<Text> needles; needles.reserve(needlesWithValues.size());
    for (auto& nv : needlesWithValues) needles.push_back(nv.first);
    detail::Packed p = detail::buildPacked(needles);
    AcMachine<V> m;
    m.machineValues.resize(p.valueIdx.size());
    std::vector<uint32_t> valuesLen(p.valueIdx.size());
    for (size_t s = 0; s < p.valueIdx.size(); s++) {
        valuesLen[s] = (uint32_t)p.valueIdx[s].size();
        m.machineValues[s].reserve(p.valueIdx[s].size());
        for (uint32_t i : p.valueIdx[s]) m.machineValues[s].push_back(needlesWithValues[i].second);
    }
    m.machineTransitions = std::move(p.transitions);
    m.machineOffsets = std::move(p.offsets);
    m.machineRootAsciiTransitions = std::move(p.rootAscii);
    am_automaton* a = nullptr;
    // an EMPTY caller table is a table (ASCII-only lower-casing): two valid pointers with n = 0, never NULL (= the built-in table)
    static const uint32_t none = 0;
    const uint32_t* lf = !lower ? nullptr : (lower->from.empty() ? &none : lower->from.data());
    const uint32_t* lt = !lower ? nullptr : (lower->to.empty() ? &none : lower->to.data());
    amCheck(am_automaton_create_ex(m.machineTransitions.data(), m.machineTransitions.size(), m.machineOffsets.data(), m.numStates(),
                                   m.machineRootAsciiTransitions.data(), valuesLen.data(), lf, lt, lower ? lower->from.size() : 0, &a));
    m.device.reset(a, detail::AutomatonDeleter());
    return m;
}

// Fold the records of ONE haystack in the reference's order; returns false after a Done.
template <class A, class V, class F>
bool foldRecords(A& acc, F& f, const AcMachine<V>& machine, const am_match* recs, uint64_t n)
{
    for (uint64_t i = 0; i < n; i++) {
        for (const V& v : machine.machineValues[recs[i].state]) {           // collectMatches (:522-534)
            Next<A> nx = f(std::move(acc), Match<V>{(CodeUnitIndex)recs[i].end_pos, v});
            acc = std::move(nx.value);
            if (nx.done) return false;
        }
    }
    return true;
}

// Automaton.hs:442-534 runWithCase, for a batch of independent haystacks (one fold per haystack).
template <class A, class V, class F>
std::vector<A> runBatchWithCase(CaseSensitivity cs, const A& seed, F f, const AcMachine<V>& machine, const std::vector<Text>& texts)
{
    std::vector<am_slice> slices(texts.size());
    for (size_t i = 0; i < texts.size(); i++) slices[i] = am_slice{texts[i].data, texts[i].off, texts[i].len};
    am_matches* ms = nullptr;
    amCheck(am_run(machine.device.get(), (int)cs, slices.data(), slices.size(), &ms));
    std::unique_ptr<am_matches, void (*)(am_matches*)> guard(ms, am_matches_free);
    const uint64_t n = am_matches_size(ms);
    const am_match* recs = am_matches_data(ms);
    if (n && !recs) throw AmError(AM_ERR_HIP, am_last_error());
    std::vector<A> out(texts.size(), seed);
    for (uint64_t i = 0; i < n;) {
        uint64_t j = i;
        while (j < n && recs[j].haystack == recs[i].haystack) j++;
        foldRecords(out[recs[i].haystack], f, machine, recs + i, j - i);
        i = j;
    }
    return out;
}

// Automaton.hs:443 runWithCase :: CaseSensitivity -> a -> (a -> Match v -> Next a) -> AcMachine v -> Text -> a
template <class A, class V, class F>
A runWithCase(CaseSensitivity cs, A seed, F f, const AcMachine<V>& machine, const Text& text)
{
    return std::move(runBatchWithCase(cs, seed, f, machine, std::vector<Text>{text})[0]);
}

template <class A, class V, class F> A runText(A seed, F f, const AcMachine<V>& m, const Text& t) { return runWithCase(CaseSensitivity::CaseSensitive, std::move(seed), f, m, t); }   // :539-541
template <class A, class V, class F> A runLower(A seed, F f, const AcMachine<V>& m, const Text& t) { return runWithCase(CaseSensitivity::IgnoreCase, std::move(seed), f, m, t); }     // :551-553


// Automaton.hs:555-566 needleCasings: every text that lower-cases to the given (lower case) text: the product of unlowerCodePoint over its code
// points, first code point slowest, each in the reference's order ("abc" -> abc abC aBc aBC Abc AbC ABc ABC; "ABC" -> nothing).
inline std::vector<std::string> needleCasings(const Text& needle, const utf8::LowerTable* lt = nullptr)
{
    std::vector<std::vector<uint32_t>> sets;
    const uint8_t* d = needle.begin();
    for (size_t i = 0; i < needle.len;) { size_t u; const uint32_t cp = utf8::decodeAt(d, i, needle.len, u); i += u; sets.push_back(utf8::unlowerCodePoint(cp, lt)); }
    std::vector<std::string> out{std::string()};
    for (size_t k = sets.size(); k-- > 0;) {                  // loop (c:cs) = (:) <$> unlowerCodePoint c <*> loop cs
        std::vector<std::string> next;
        next.reserve(out.size() * sets[k].size());
        for (uint32_t c : sets[k]) { std::string head; utf8::encode(c, head); for (const std::string& tail : out) next.push_back(head + tail); }
        out.swap(next);
    }
    return out;
}

}  // namespace alfred_margaret
