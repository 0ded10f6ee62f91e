// replacer.hpp -- host mirror of Data.Text.AhoCorasick.Replacer (reference:
// src/Data/Text/AhoCorasick/Replacer.hs): sequential multi-needle replace with priorities.
//   Payload :59-70, build :97-116, compose :120-133, mapReplacement :135-141, setCaseSensitivity :148-153,
//   run :200-201, runWithLimit :203-274,
//   removeOverlap :191-198, replace :163-180, replacementLength :183-187.
// runBatchWithLimit hands the whole multi-pass loop to libam (am_replacer_*): scan, priority fold,
// overlap removal and splice all run in HBM, only finished haystacks come back.
// runBatchWithLimitHostSplice is the same function with only the scans on the GPU (am_run on the
// batch of still-active haystacks per pass) and the fold/sort/splice on the host; it is kept as an
// independent cross-check of the device passes.
#pragma once
#include <climits>
#include <mutex>
#include <optional>

#include "searcher.hpp"

namespace alfred_margaret {

struct Payload {                        // Replacer.hs:59-70
    long long needlePriority;
    size_t needleLengthBytes;
    size_t needleLengthCodePoints;
    std::string needleReplacement;
};

class Replacer {
public:
    // Replacer.hs:97-116 build: needle i has priority -i; IgnoreCase lower-cases the needle, the
    // payload lengths are those of the ORIGINAL needle.
    // lower (optional): the caller's Data.Char.toLower as data, used for the needles here (:105-107) and by the device for the haystacks
    Replacer(CaseSensitivity cs, const std::vector<std::pair<std::string, std::string>>& replaces, const utf8::LowerTable* lower = nullptr)
        : searcher_(cs, mapNeedles(cs, replaces, lower), lower) {}

    CaseSensitivity caseSensitivity() const { return searcher_.caseSensitivity(); }
    const Searcher<Payload>& searcher() const { return searcher_; }

    // Replacer.hs:120-133 compose (same case sensitivity required)
    static std::optional<Replacer> compose(const Replacer& a, const Replacer& b)
    {
        if (a.caseSensitivity() != b.caseSensitivity()) return std::nullopt;
        std::vector<std::pair<std::string, Payload>> ns = a.searcher_.needles();
        ns.insert(ns.end(), b.searcher_.needles().begin(), b.searcher_.needles().end());
        for (size_t i = 0; i < ns.size(); i++) ns[i].second.needlePriority = -(long long)i;
        return Replacer(Searcher<Payload>(a.caseSensitivity(), std::move(ns)));
    }

    // Replacer.hs:135-141 mapReplacement: new replacements, needles untouched
    template <class F> Replacer mapReplacement(F f) const
    {
        return Replacer(searcher_.mapSearcher([&](const Payload& p) { Payload q = p; q.needleReplacement = f(p.needleReplacement); return q; }));
    }
    // the same with the needle's index (= -priority) handed to f
    template <class F> Replacer mapReplacementIndexed(F f) const
    {
        return Replacer(searcher_.mapSearcher([&](const Payload& p) { Payload q = p; q.needleReplacement = f((size_t)(-p.needlePriority), p.needleReplacement); return q; }));
    }
    // Replacer.hs:148-153 setCaseSensitivity: flips the mode without touching the needles (the caller makes sure they are lower case for IgnoreCase)
    Replacer setCaseSensitivity(CaseSensitivity cs) const
    {
        Searcher<Payload> s = searcher_.mapSearcher([](const Payload& p) { return p; });
        s.setCaseSensitivity(cs);
        return Replacer(std::move(s));
    }

    // statistics of the last run THE CALLING THREAD made on this replacer (runs are const and may be made from several threads
    // on one shared handle: the numbers live in thread-local storage, not in the object)
    struct RunStats { uint64_t passes = 0, scannedBytes = 0; };
    RunStats lastStats() const { const auto& t = threadStats(); return t.first == this ? t.second : RunStats{}; }

    // Replacer.hs:203-274 runWithLimit on a batch; nullopt where the reference returns Nothing.
    std::vector<std::optional<std::string>> runBatchWithLimit(const std::vector<std::string>& inputs, size_t maxLength) const
    {
        std::vector<am_slice> slices(inputs.size());
        for (size_t k = 0; k < inputs.size(); k++) slices[k] = am_slice{(const uint8_t*)inputs[k].data(), 0, inputs[k].size()};
        am_replaced* raw = nullptr;
        amCheck(am_replacer_run(device(), slices.data(), slices.size(), maxLength == SIZE_MAX ? UINT64_MAX : (uint64_t)maxLength, &raw));
        std::unique_ptr<am_replaced, void (*)(am_replaced*)> res(raw, am_replaced_free);
        RunStats& stats_ = statsOfThisThread();
        stats_ = RunStats{am_replaced_passes(raw), am_replaced_scanned_bytes(raw)};
        std::vector<std::optional<std::string>> out(inputs.size());
        for (size_t k = 0; k < inputs.size(); k++) {
            const uint8_t* p = nullptr; size_t n = 0;
            const int just = am_replaced_get(raw, k, &p, &n);
            if (just < 0) throw AmError(just, am_last_error());
            if (just) out[k] = std::string((const char*)p, n);
        }
        return out;
    }

    // The flattened Searcher Payload (machineValues in CSR form + payload table) in HBM, made on first use.
    const am_replacer* device() const
    {
        std::lock_guard<std::mutex> lk(*mu_);
        if (!device_) {
            const auto& m = searcher_.automaton();
            const auto& ns = searcher_.needles();
            std::vector<am_payload> payloads(ns.size());
            std::string repl;
            for (size_t i = 0; i < ns.size(); i++) {
                const Payload& p = ns[i].second;
                if (p.needlePriority != -(long long)i) throw AmError(AM_ERR_INVALID, "Replacer payload priorities must be -index (Replacer.hs:100-104)");
                payloads[i] = am_payload{p.needlePriority, (uint32_t)p.needleLengthBytes, (uint32_t)p.needleLengthCodePoints, repl.size(), (uint32_t)p.needleReplacement.size(), 0};
                repl += p.needleReplacement;
            }
            std::vector<uint64_t> voff(m.machineValues.size() + 1, 0);
            std::vector<uint32_t> vals;
            for (size_t s = 0; s < m.machineValues.size(); s++) {
                for (const Payload& p : m.machineValues[s]) vals.push_back((uint32_t)(-p.needlePriority));
                voff[s + 1] = vals.size();
            }
            am_replacer* r = nullptr;
            amCheck(am_replacer_create(m.device.get(), (int)caseSensitivity(), voff.data(), vals.data(), payloads.data(), payloads.size(),
                                       (const uint8_t*)repl.data(), repl.size(), 1 - (long long)ns.size(), &r));
            device_.reset(r, am_replacer_destroy);
        }
        return device_.get();
    }

    std::vector<std::optional<std::string>> runBatchWithLimitHostSplice(const std::vector<std::string>& inputs, size_t maxLength) const
    {
        struct RMatch { size_t pos, len; const std::string* repl; };                       // Replacer.hs:159
        const bool ic = caseSensitivity() == CaseSensitivity::IgnoreCase;
        const long long minPriority = 1 - (long long)searcher_.numNeedles();               // :217
        std::vector<std::optional<std::string>> cur(inputs.begin(), inputs.end());
        std::vector<long long> threshold(inputs.size(), 1);                                // :211
        std::vector<size_t> active(inputs.size());
        for (size_t i = 0; i < active.size(); i++) active[i] = i;
        struct Acc { long long pBest; std::vector<RMatch> matches; const std::string* hay; long long threshold; };
        RunStats& stats_ = statsOfThisThread();
        stats_ = RunStats{};
        while (!active.empty()) {
            stats_.passes++;
            for (size_t i : active) stats_.scannedBytes += cur[i]->size();
            std::vector<Text> texts; texts.reserve(active.size());
            for (size_t i : active) texts.emplace_back(*cur[i]);
            // one GPU scan of every active haystack; the fold below is prependMatch (:252-260)
            std::vector<am_slice> slices(texts.size());
            for (size_t k = 0; k < texts.size(); k++) slices[k] = am_slice{texts[k].data, texts[k].off, texts[k].len};
            am_matches* ms = nullptr;
            amCheck(am_run(searcher_.automaton().device.get(), (int)caseSensitivity(), slices.data(), slices.size(), &ms));
            std::unique_ptr<am_matches, void (*)(am_matches*)> guard(ms, am_matches_free);
            const uint64_t n = am_matches_size(ms);
            const am_match* recs = am_matches_data(ms);
            if (n && !recs) throw AmError(AM_ERR_HIP, am_last_error());
            std::vector<Acc> accs(active.size());
            for (size_t k = 0; k < active.size(); k++) accs[k] = Acc{LLONG_MIN, {}, &*cur[active[k]], threshold[active[k]]};
            auto f = [ic](Acc acc, const Match<Payload>& m) {
                const Payload& p = m.matchValue;
                if (p.needlePriority < acc.threshold && p.needlePriority >= acc.pBest) {
                    if (p.needlePriority > acc.pBest) { acc.pBest = p.needlePriority; acc.matches.clear(); }
                    RMatch rm;                                                              // makeMatch (:264-274)
                    if (!ic) { rm.pos = m.matchPos - p.needleLengthBytes; rm.len = p.needleLengthBytes; }
                    else {
                        const size_t start = utf8::skipCodePointsBackwards(Text(*acc.hay), m.matchPos - 1, p.needleLengthCodePoints - 1);
                        rm.pos = start; rm.len = m.matchPos - start;
                    }
                    rm.repl = &p.needleReplacement;
                    acc.matches.push_back(rm);
                }
                return Next<Acc>::Step(std::move(acc));
            };
            for (uint64_t i = 0; i < n;) {
                uint64_t j = i;
                while (j < n && recs[j].haystack == recs[i].haystack) j++;
                foldRecords(accs[recs[i].haystack], f, searcher_.automaton(), recs + i, j - i);
                i = j;
            }
            std::vector<size_t> next;
            for (size_t k = 0; k < active.size(); k++) {
                const size_t idx = active[k];
                Acc& acc = accs[k];
                if (acc.matches.empty()) continue;                                          // (_, []) -> Just haystack (:230)
                const std::string& hay = *cur[idx];
                long long newLen = (long long)hay.size();                                   // replacementLength (:183-187)
                for (auto& m : acc.matches) newLen += (long long)m.repl->size() - (long long)m.len;
                if (newLen > (long long)maxLength && maxLength != SIZE_MAX) { cur[idx] = std::nullopt; continue; }   // :240
                std::sort(acc.matches.begin(), acc.matches.end(), [](const RMatch& a, const RMatch& b) {   // derived Ord (:159)
                    if (a.pos != b.pos) return a.pos < b.pos;
                    if (a.len != b.len) return a.len < b.len;
                    return *a.repl < *b.repl;
                });
                std::string out; out.reserve((size_t)std::max<long long>(newLen, 0));
                size_t at = 0, lastEnd = 0; bool any = false;
                for (auto& m : acc.matches) {                                               // removeOverlap (:191-198) + replace (:163-180)
                    if (any && m.pos < lastEnd) continue;
                    out.append(hay, at, m.pos - at);
                    out.append(*m.repl);
                    at = m.pos + m.len; lastEnd = at; any = true;
                }
                out.append(hay, at, std::string::npos);
                cur[idx] = std::move(out);
                if (acc.pBest != minPriority) { threshold[idx] = acc.pBest; next.push_back(idx); }   // :241-242
            }
            active.swap(next);
        }
        return cur;
    }

    // Replacer.hs:200-201 run
    std::string run(const std::string& text) const { return *runBatchWithLimit({text}, SIZE_MAX)[0]; }
    std::optional<std::string> runWithLimit(size_t maxLength, const std::string& text) const { return runBatchWithLimit({text}, maxLength)[0]; }

private:
    explicit Replacer(Searcher<Payload> s) : searcher_(std::move(s)) {}
    static std::vector<std::pair<std::string, Payload>> mapNeedles(CaseSensitivity cs, const std::vector<std::pair<std::string, std::string>>& replaces,
                                                                   const utf8::LowerTable* lower)
    {
        std::vector<std::pair<std::string, Payload>> out; out.reserve(replaces.size());
        for (size_t i = 0; i < replaces.size(); i++) {
            const std::string& needle = replaces[i].first;
            Payload p{-(long long)i, needle.size(), utf8::lengthCodePoints(Text(needle)), replaces[i].second};
            out.emplace_back(cs == CaseSensitivity::IgnoreCase ? utf8::lowerUtf8(Text(needle), lower) : needle, std::move(p));
        }
        return out;
    }
    Searcher<Payload> searcher_;
    std::shared_ptr<std::mutex> mu_ = std::make_shared<std::mutex>();
    mutable std::shared_ptr<am_replacer> device_;
    static std::pair<const Replacer*, RunStats>& threadStats() { static thread_local std::pair<const Replacer*, RunStats> t{nullptr, RunStats{}}; return t; }
    RunStats& statsOfThisThread() const { auto& t = threadStats(); t.first = this; return t.second; }
};

}  // namespace alfred_margaret
