// searcher.hpp -- host mirror of Data.Text.AhoCorasick.Searcher (reference:
// src/Data/Text/AhoCorasick/Searcher.hs): needles + case mode + automaton.
//   build / buildWithValues :110-118, mapSearcher :121-125, containsAny :156-164, buildNeedleIdSearcher :167-169,
//   containsAll :173-187, setCaseSensitivity :142-145.
// containsAny dispatches to libam's flag kernel; containsAll to libam's needle-id bitmap fold
// (am_contains_all); containsAllBatchHostFold is the same function folding the records on the host.
#pragma once
#include <mutex>

#include "automaton.hpp"

namespace alfred_margaret {

template <class V> class Searcher {
public:
    Searcher(CaseSensitivity cs, std::vector<std::pair<std::string, V>> needlesWithValues, const utf8::LowerTable* lower = nullptr)
        : case_(cs), needles_(std::move(needlesWithValues))
    {
        std::vector<std::pair<Text, V>> nv; nv.reserve(needles_.size());
        for (auto& p : needles_) nv.emplace_back(Text(p.first), p.second);
        automaton_ = alfred_margaret::build(nv, lower);           // Searcher.hs:118  Aho.build ns
    }
    CaseSensitivity caseSensitivity() const { return case_; }
    void setCaseSensitivity(CaseSensitivity cs) { case_ = cs; }   // Searcher.hs:142-145: needles untouched
    const std::vector<std::pair<std::string, V>>& needles() const { return needles_; }
    size_t numNeedles() const { return needles_.size(); }
    const AcMachine<V>& automaton() const { return automaton_; }
    // Searcher.hs:121-125 mapSearcher: new values, same needles and transitions (the device automaton is shared, it never sees the values)
    template <class F> auto mapSearcher(F f) const -> Searcher<decltype(f(std::declval<const V&>()))>
    {
        using B = decltype(f(std::declval<const V&>()));
        Searcher<B> out;
        out.case_ = case_;
        out.needles_.reserve(needles_.size());
        for (const auto& p : needles_) out.needles_.emplace_back(p.first, f(p.second));
        out.automaton_.machineTransitions = automaton_.machineTransitions;
        out.automaton_.machineOffsets = automaton_.machineOffsets;
        out.automaton_.machineRootAsciiTransitions = automaton_.machineRootAsciiTransitions;
        out.automaton_.device = automaton_.device;
        out.automaton_.machineValues.reserve(automaton_.machineValues.size());
        for (const auto& vs : automaton_.machineValues) { std::vector<B> bs; bs.reserve(vs.size()); for (const V& v : vs) bs.push_back(f(v)); out.automaton_.machineValues.push_back(std::move(bs)); }
        return out;
    }
    // per-searcher device-side extras (flattened machineValues), created on first use by the functions below
    std::shared_ptr<void>& deviceExtra() const { return extra_; }
    std::mutex& deviceExtraMutex() const { return *extraMu_; }

private:
    template <class U> friend class Searcher;
    Searcher() = default;
    mutable std::shared_ptr<void> extra_;
    std::shared_ptr<std::mutex> extraMu_ = std::make_shared<std::mutex>();
    CaseSensitivity case_;
    std::vector<std::pair<std::string, V>> needles_;
    AcMachine<V> automaton_;
};

struct Unit {};

// Searcher.hs:110-111 build :: CaseSensitivity -> [Text] -> Searcher ()
inline Searcher<Unit> buildSearcher(CaseSensitivity cs, const std::vector<std::string>& needles)
{
    std::vector<std::pair<std::string, Unit>> nv; nv.reserve(needles.size());
    for (auto& n : needles) nv.emplace_back(n, Unit{});
    return Searcher<Unit>(cs, std::move(nv));
}

// Searcher.hs:167-169 buildNeedleIdSearcher
inline Searcher<int> buildNeedleIdSearcher(CaseSensitivity cs, const std::vector<std::string>& needles, const utf8::LowerTable* lower = nullptr)
{
    std::vector<std::pair<std::string, int>> nv; nv.reserve(needles.size());
    for (size_t i = 0; i < needles.size(); i++) nv.emplace_back(needles[i], (int)i);
    return Searcher<int>(cs, std::move(nv), lower);
}

// Searcher.hs:156-164 containsAny, for a batch of haystacks (one Bool each)
template <class V>
std::vector<bool> containsAnyBatch(const Searcher<V>& s, const std::vector<Text>& texts)
{
    std::vector<am_slice> slices(texts.size());
    for (size_t i = 0; i < texts.size(); i++) slices[i] = am_slice{texts[i].data, texts[i].off, texts[i].len};
    std::vector<uint8_t> flags(texts.size() ? texts.size() : 1, 0);
    amCheck(am_contains_any(s.automaton().device.get(), (int)s.caseSensitivity(), slices.data(), slices.size(), flags.data()));
    return std::vector<bool>(flags.begin(), flags.begin() + texts.size());
}

template <class V> bool containsAny(const Searcher<V>& s, const Text& text) { return containsAnyBatch(s, std::vector<Text>{text})[0]; }

// Searcher.hs:173-187 containsAll on the device: the IntSet is a bitmap row per haystack in HBM
inline std::vector<bool> containsAllBatch(const Searcher<int>& s, const std::vector<Text>& texts)
{
    am_needle_ids* ids = nullptr;
    {
        std::lock_guard<std::mutex> lk(s.deviceExtraMutex());
        if (!s.deviceExtra()) {
            const auto& m = s.automaton();
            std::vector<uint64_t> voff(m.machineValues.size() + 1, 0);
            std::vector<uint32_t> vals;
            for (size_t st = 0; st < m.machineValues.size(); st++) {
                for (int id : m.machineValues[st]) vals.push_back(id < 0 ? UINT32_MAX : (uint32_t)id);
                voff[st + 1] = vals.size();
            }
            am_needle_ids* raw = nullptr;
            amCheck(am_needle_ids_create(m.device.get(), voff.data(), vals.data(), (uint32_t)s.numNeedles(), &raw));
            s.deviceExtra() = std::shared_ptr<void>(raw, [](void* p) { am_needle_ids_destroy(static_cast<am_needle_ids*>(p)); });
        }
        ids = static_cast<am_needle_ids*>(s.deviceExtra().get());
    }
    std::vector<am_slice> slices(texts.size());
    for (size_t k = 0; k < texts.size(); k++) slices[k] = am_slice{texts[k].data, texts[k].off, texts[k].len};
    std::vector<uint8_t> flags(texts.size() ? texts.size() : 1, 0);
    amCheck(am_contains_all(ids, (int)s.caseSensitivity(), slices.data(), slices.size(), flags.data()));
    return std::vector<bool>(flags.begin(), flags.begin() + texts.size());
}

// the same function with only the scan on the GPU: delete each reported needle id from the set, Done when empty
inline std::vector<bool> containsAllBatchHostFold(const Searcher<int>& s, const std::vector<Text>& texts)
{
    struct Acc { std::vector<uint8_t> present; size_t remaining; };
    Acc seed{std::vector<uint8_t>(s.numNeedles(), 1), s.numNeedles()};
    auto f = [](Acc acc, const Match<int>& m) {
        if (acc.present[m.matchValue]) { acc.present[m.matchValue] = 0; acc.remaining--; }
        return acc.remaining == 0 ? Next<Acc>::Done(std::move(acc)) : Next<Acc>::Step(std::move(acc));
    };
    std::vector<Acc> accs = runBatchWithCase(s.caseSensitivity(), seed, f, s.automaton(), texts);
    std::vector<bool> out(texts.size());
    for (size_t i = 0; i < texts.size(); i++) out[i] = accs[i].remaining == 0;
    return out;
}

inline bool containsAll(const Searcher<int>& s, const Text& text) { return containsAllBatch(s, std::vector<Text>{text})[0]; }

}  // namespace alfred_margaret
