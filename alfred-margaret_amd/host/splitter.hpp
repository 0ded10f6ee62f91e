// splitter.hpp -- host mirror of Data.Text.AhoCorasick.Splitter (reference:
// src/Data/Text/AhoCorasick/Splitter.hs): split on a single separator with a 1-needle automaton.
//   build :64-67, split :84-85, splitIgnoreCase :96-97, splitReverse :100-107,
//   splitReverseIgnoreCase :112-121, Accum / stepAccum / finalizeAccum :128-170.
// The match positions come from libam (GPU); the fold below is the reference's stepAccum.
#pragma once
#include <algorithm>

#include "automaton.hpp"

namespace alfred_margaret {

class Splitter {
public:
    struct Unit0 {};
    explicit Splitter(std::string sep) : separator_(std::move(sep))
    {
        std::vector<std::pair<Text, Unit0>> nv{{Text(separator_), Unit0{}}};
        automaton_ = alfred_margaret::build(nv);                     // Splitter.hs:66  Aho.build [(sep, ())]
    }
    const std::string& separator() const { return separator_; }

    // Splitter.hs:84-85 / :96-97, for a batch: fragments of every haystack in order
    std::vector<std::vector<std::string>> splitBatch(const std::vector<std::string>& texts, bool ignoreCase) const
    {
        struct Accum { std::vector<std::string> result; size_t fragmentStart; const std::string* hay; };   // Splitter.hs:128-136
        // case sensitive: separator length in bytes (:105); ignore case: in code points (:118)
        const size_t sepBytes = separator_.size(), sepCps = utf8::lengthCodePoints(Text(separator_));
        auto step = [&](Accum acc, const Match<Unit0>& m) {                                                   // stepAccum :158-170
            const size_t newFragmentStart = m.matchPos;
            const size_t sepStart = ignoreCase ? utf8::skipCodePointsBackwards(Text(*acc.hay), newFragmentStart - 1, sepCps - 1)
                                               : newFragmentStart - sepBytes;
            if (sepStart >= acc.fragmentStart) {
                acc.result.emplace_back(*acc.hay, acc.fragmentStart, sepStart - acc.fragmentStart);
                acc.fragmentStart = newFragmentStart;
            }
            return Next<Accum>::Step(std::move(acc));
        };
        std::vector<Text> ts; ts.reserve(texts.size());
        for (auto& t : texts) ts.emplace_back(t);
        // one accumulator per haystack: the seed cannot carry the haystack pointer, so fold per record group
        std::vector<am_slice> slices(ts.size());
        for (size_t i = 0; i < ts.size(); i++) slices[i] = am_slice{ts[i].data, ts[i].off, ts[i].len};
        am_matches* ms = nullptr;
        amCheck(am_run(automaton_.device.get(), ignoreCase ? AM_IGNORE_CASE : AM_CASE_SENSITIVE, slices.data(), slices.size(), &ms));
        std::unique_ptr<am_matches, void (*)(am_matches*)> guard(ms, am_matches_free);
        const uint64_t n = am_matches_size(ms);
        const am_match* recs = am_matches_data(ms);
        if (n && !recs) throw AmError(AM_ERR_HIP, am_last_error());
        std::vector<Accum> accs(texts.size());
        for (size_t i = 0; i < texts.size(); i++) accs[i] = Accum{{}, 0, &texts[i]};                         // zeroAccum :150-151
        for (uint64_t i = 0; i < n;) {
            uint64_t j = i;
            while (j < n && recs[j].haystack == recs[i].haystack) j++;
            foldRecords(accs[recs[i].haystack], step, automaton_, recs + i, j - i);
            i = j;
        }
        std::vector<std::vector<std::string>> out(texts.size());
        for (size_t i = 0; i < texts.size(); i++) {                                                          // finalizeAccum :141-146
            accs[i].result.emplace_back(texts[i], accs[i].fragmentStart, std::string::npos);
            out[i] = std::move(accs[i].result);
        }
        return out;
    }
    std::vector<std::string> split(const std::string& text) const { return splitBatch({text}, false)[0]; }
    std::vector<std::string> splitIgnoreCase(const std::string& text) const { return splitBatch({text}, true)[0]; }
    // Splitter.hs:100-107 / :112-121: the fragments last-first (what the fold accumulates before `split` reverses it)
    std::vector<std::string> splitReverse(const std::string& text) const { auto v = split(text); std::reverse(v.begin(), v.end()); return v; }
    std::vector<std::string> splitReverseIgnoreCase(const std::string& text) const { auto v = splitIgnoreCase(text); std::reverse(v.begin(), v.end()); return v; }
    const AcMachine<Unit0>& automaton() const { return automaton_; }       // Splitter.hs:71-72

private:
    std::string separator_;
    AcMachine<Unit0> automaton_;
};

}  // namespace alfred_margaret
