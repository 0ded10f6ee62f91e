// utf8.hpp -- host mirror of the parts of Data.Text.Utf8 that sit on the Aho-Corasick path
// (reference: src/Data/Text/Utf8.hs).  Header-only; lower-casing defers to libam's table.
#pragma once
#include <cstddef>
#include <cstdint>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "am.h"

namespace alfred_margaret {
namespace utf8 {

// Data.Text.Internal.Text: array + offset + length in code units (bytes)
struct Text {
    const uint8_t* data = nullptr;
    size_t off = 0;
    size_t len = 0;
    Text() = default;
    Text(const uint8_t* d, size_t o, size_t l) : data(d), off(o), len(l) {}
    Text(const std::string& s) : data(reinterpret_cast<const uint8_t*>(s.data())), off(0), len(s.size()) {}
    const uint8_t* begin() const { return data + off; }
};

// Utf8.hs:337-350 unsafeIndexCodePoint' / decodeN (reads guarded by `end`)
inline uint32_t decodeAt(const uint8_t* d, size_t idx, size_t end, size_t& units)
{
    const uint32_t cu0 = d[idx];
    const uint32_t cu1 = idx + 1 < end ? d[idx + 1] : 0, cu2 = idx + 2 < end ? d[idx + 2] : 0, cu3 = idx + 3 < end ? d[idx + 3] : 0;
    if (cu0 < 0xc0) { units = 1; return cu0; }
    if (cu0 < 0xe0) { units = 2; return ((cu0 & 0x1f) << 6) | (cu1 & 0x3f); }
    if (cu0 < 0xf0) { units = 3; return ((cu0 & 0xf) << 12) | ((cu1 & 0x3f) << 6) | (cu2 & 0x3f); }
    units = 4;
    return ((cu0 & 0x7) << 18) | ((cu1 & 0x3f) << 12) | ((cu2 & 0x3f) << 6) | (cu3 & 0x3f);
}

// Utf8.hs:154-160 unicode2utf8
inline void encode(uint32_t c, std::string& out)
{
    if (c < 0x80) out.push_back((char)c);
    else if (c < 0x800) { out.push_back((char)(0xc0 | (c >> 6))); out.push_back((char)(0x80 | (c & 0x3f))); }
    else if (c < 0x10000) { out.push_back((char)(0xe0 | (c >> 12))); out.push_back((char)(0x80 | ((c >> 6) & 0x3f))); out.push_back((char)(0x80 | (c & 0x3f))); }
    else { out.push_back((char)(0xf0 | (c >> 18))); out.push_back((char)(0x80 | ((c >> 12) & 0x3f))); out.push_back((char)(0x80 | ((c >> 6) & 0x3f))); out.push_back((char)(0x80 | (c & 0x3f))); }
}

// The lower-casing of the host language as data: what Data.Char.toLower of the caller's GHC does (Utf8.hs:151), as the (c, toLower c) pairs
// with toLower c /= c.  Handed to am_automaton_create_ex by build(); null everywhere = libam's built-in Unicode 14.0 table.
// Same rules as libam's own table (am_automaton_create_ex): pairs below 128 and identity pairs are ignored, duplicates collapse, one code
// point with two different images (or a code point beyond U+10FFFF) is refused -- so the host-side lower-casing of the needles and the
// device-side table can never disagree about a table the device would have rejected.
struct LowerTable {
    std::vector<uint32_t> from, to;                            // sorted by from
    std::vector<std::pair<uint32_t, uint32_t>> inverse;        // (to, from), sorted: unlowerCodePoint
    LowerTable(const uint32_t* f, const uint32_t* t, size_t n)
    {
        std::vector<std::pair<uint32_t, uint32_t>> v;
        for (size_t i = 0; i < n; i++) {
            if (f[i] > 0x10FFFFu || t[i] > 0x10FFFFu) throw std::invalid_argument("lower-case pair beyond U+10FFFF");
            if (f[i] < 128u || f[i] == t[i]) continue;
            v.emplace_back(f[i], t[i]);
        }
        std::sort(v.begin(), v.end());
        for (size_t i = 1; i < v.size(); i++)
            if (v[i].first == v[i - 1].first && v[i].second != v[i - 1].second) throw std::invalid_argument("lower-case table maps one code point to two different ones");
        v.erase(std::unique(v.begin(), v.end()), v.end());
        for (auto& p : v) { from.push_back(p.first); to.push_back(p.second); inverse.emplace_back(p.second, p.first); }
        for (uint32_t c = 'A'; c <= 'Z'; c++) inverse.emplace_back(c + 0x20u, c);
        std::sort(inverse.begin(), inverse.end());
    }
    uint32_t lower(uint32_t cp) const
    {
        if (cp < 128) return cp - 0x41u < 26u ? cp + 0x20u : cp;         // toLowerAscii (Utf8.hs:131-135)
        const auto it = std::lower_bound(from.begin(), from.end(), cp);
        return it != from.end() && *it == cp ? to[(size_t)(it - from.begin())] : cp;
    }
};

// Utf8.hs:145-151 lowerCodePoint
inline uint32_t lowerCodePoint(uint32_t cp, const LowerTable* lt = nullptr) { return lt ? lt->lower(cp) : am_lower_code_point(cp); }

// Utf8.hs:138-140 lowerUtf8
inline std::string lowerUtf8(const Text& t, const LowerTable* lt = nullptr)
{
    std::string out;
    out.reserve(t.len);
    const uint8_t* d = t.begin();
    for (size_t i = 0; i < t.len;) { size_t u; uint32_t cp = decodeAt(d, i, t.len, u); encode(lowerCodePoint(cp, lt), out); i += u; }
    return out;
}

// Utf8/Unlower.hs:26-40 unlowerCodePoint: every code point whose lower case is `cp`, in the reference's order -- its table is filled by
// `insertWith (++) (toLower c) [c]` over ascending c, so a list runs from the HIGHEST code point down: unlowerCodePoint 'i' == "\304iI",
// 'a' == "aA", 'A' == "" (Utf8Spec.hs:55-62).
inline std::vector<uint32_t> unlowerCodePoint(uint32_t cp, const LowerTable* lt = nullptr)
{
    std::vector<uint32_t> out;
    if (!lt) {
        uint32_t buf[16];
        size_t n = am_unlower_code_point(cp, buf, 16);                   // ascending set (built-in table)
        if (n > 16) { std::vector<uint32_t> big(n); n = am_unlower_code_point(cp, big.data(), n); out.assign(big.begin(), big.begin() + n); }
        else out.assign(buf, buf + n);
    } else {
        if (lt->lower(cp) == cp) out.push_back(cp);
        for (auto it = std::lower_bound(lt->inverse.begin(), lt->inverse.end(), std::make_pair(cp, 0u)); it != lt->inverse.end() && it->first == cp; ++it) out.push_back(it->second);
        std::sort(out.begin(), out.end());
    }
    std::reverse(out.begin(), out.end());
    return out;
}

// Utf8.hs:169-171 isCaseInvariant = Text.all (\c -> unlowerCodePoint (lowerCodePoint c) == [c])
inline bool isCaseInvariant(const Text& t, const LowerTable* lt = nullptr)
{
    const uint8_t* d = t.begin();
    for (size_t i = 0; i < t.len;) {
        size_t u; const uint32_t cp = decodeAt(d, i, t.len, u); i += u;
        const std::vector<uint32_t> back = unlowerCodePoint(lowerCodePoint(cp, lt), lt);
        if (back.size() != 1 || back[0] != cp) return false;
    }
    return true;
}

// Text.length: code points
inline size_t lengthCodePoints(const Text& t)
{
    size_t n = 0;
    const uint8_t* d = t.begin();
    for (size_t i = 0; i < t.len;) { size_t u; (void)decodeAt(d, i, t.len, u); i += u; n++; }
    return n;
}

// Utf8.hs:256-276 skipCodePointsBackwards; throws where the reference calls `error`
inline size_t skipCodePointsBackwards(const Text& t, size_t index0, size_t n0)
{
    if (index0 >= t.len) throw std::out_of_range("Invalid use of skipCodePointsBackwards");
    const uint8_t* d = t.begin();
    long long index = (long long)index0; size_t n = n0;
    for (;;) {
        if (index >= 0 && (d[index] & 0xC0) == 0x80) { index--; continue; }
        if (index < 0) throw std::out_of_range("Invalid use of skipCodePointsBackwards");
        if (n == 0) return (size_t)index;
        index--; n--;
    }
}

}  // namespace utf8
}  // namespace alfred_margaret
