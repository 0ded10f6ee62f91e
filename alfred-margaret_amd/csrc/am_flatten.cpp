// am_flatten.cpp -- host side of `Automaton.build`'s new second half: take the reference's packed
// automaton (Automaton.hs:108-123 AcMachine: Word64 transitions, Word32 offsets, root ASCII table,
// per-state value-list lengths) and flatten it into the device image described in am_image.h.
// The reference's `build` (Automaton.hs:176-200) stays authoritative for state numbering and
// value order; nothing here changes what a match means.
#include "am_flatten.h"

#include "am_config.h"

#include <algorithm>
#include <chrono>
#include <future>
#include <system_error>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace am {

// ------------------------------------------------------------------ simple lowercase
//
// The reference lowers with GHC base's Data.Char.toLower (Utf8.hs:145-151), whose table follows the compiler's Unicode version.  So the
// table is DATA: the caller of am_automaton_create_ex may hand over its own (from, to) pairs; without them the built-in Unicode 14.0
// table (unicode_lower_tbl.inc) is used.  ASCII is never taken from the table: lowerCodePoint handles it itself (toLowerAscii, :131-135).

namespace {
struct LowerPair { uint32_t from, to; };
const LowerPair kLower[] = {
#include "unicode_lower_tbl.inc"
};
constexpr size_t kNLower = sizeof(kLower) / sizeof(kLower[0]);
}  // namespace

int LowerTable::make(const uint32_t* from, const uint32_t* to, size_t n, LowerTable& out, std::string& err)
{
    std::vector<std::pair<uint32_t, uint32_t>> v;
    v.reserve(n + 26);
    for (uint32_t c = 'A'; c <= 'Z'; c++) v.emplace_back(c, c + 0x20u);            // toLowerAscii
    for (size_t i = 0; i < n; i++) {
        if (from[i] > 0x10FFFFu || to[i] > 0x10FFFFu) { err = "lower-case pair beyond U+10FFFF"; return -1; }
        if (from[i] < 128u) continue;                                               // ASCII: not the table's business
        if (from[i] == to[i]) continue;                                             // identity pairs carry no information
        v.emplace_back(from[i], to[i]);
    }
    std::sort(v.begin(), v.end());
    for (size_t i = 1; i < v.size(); i++) {
        if (v[i].first != v[i - 1].first) continue;
        if (v[i].second != v[i - 1].second) { err = "lower-case table maps one code point to two different ones"; return -1; }
    }
    v.erase(std::unique(v.begin(), v.end()), v.end());
    out.from.clear(); out.to.clear(); out.inverse.clear();
    uint64_t h = 0xCBF29CE484222325ull;
    for (auto& p : v) {
        out.from.push_back(p.first); out.to.push_back(p.second);
        out.inverse.emplace(p.second, p.first);
        for (uint32_t x : {p.first, p.second}) for (int b = 0; b < 4; b++) { h ^= (x >> (8 * b)) & 0xFFu; h *= 0x100000001B3ull; }
    }
    out.hash = (uint32_t)(h ^ (h >> 32));
    if (out.hash == 0) out.hash = 1;
    return 0;
}

uint32_t LowerTable::lower(uint32_t cp) const
{
    if (cp < 128u) return fold_byte(cp);
    const auto it = std::lower_bound(from.begin(), from.end(), cp);
    return (it != from.end() && *it == cp) ? to[(size_t)(it - from.begin())] : cp;
}

void LowerTable::unlower(uint32_t cp, std::vector<uint32_t>& out) const
{
    out.clear();
    if (lower(cp) == cp) out.push_back(cp);
    auto range = inverse.equal_range(cp);
    for (auto it = range.first; it != range.second; ++it) out.push_back(it->second);
    std::sort(out.begin(), out.end());
}

const LowerTable& builtin_lower_table()
{
    static const LowerTable* t = [] {
        std::vector<uint32_t> f(kNLower), g(kNLower);
        for (size_t i = 0; i < kNLower; i++) { f[i] = kLower[i].from; g[i] = kLower[i].to; }
        LowerTable* lt = new LowerTable(); std::string err;
        (void)LowerTable::make(f.data(), g.data(), kNLower, *lt, err);
        return lt;
    }();
    return *t;
}

// Utf8.hs:145-151 lowerCodePoint with the built-in table (non-ASCII = Data.Char.toLower's simple mapping, Unicode 14.0: kUnicodeLowerVersion)
uint32_t simple_lower(uint32_t cp) { return builtin_lower_table().lower(cp); }
// Utf8/Unlower.hs:26-40 unlowerCodePoint: every code point whose lowercase is `cp` (as a set), built-in table
void unlower(uint32_t cp, std::vector<uint32_t>& out) { builtin_lower_table().unlower(cp, out); }

static void append_utf8(uint32_t c, std::string& out)   // Utf8.hs:154-160 unicode2utf8
{
    if (c < 0x80) { out.push_back((char)c); }
    else if (c < 0x800) { out.push_back((char)(0xc0 | (c >> 6))); out.push_back((char)(0x80 | (c & 0x3f))); }
    else if (c < 0x10000) { out.push_back((char)(0xe0 | (c >> 12))); out.push_back((char)(0x80 | ((c >> 6) & 0x3f))); out.push_back((char)(0x80 | (c & 0x3f))); }
    else { out.push_back((char)(0xf0 | (c >> 18))); out.push_back((char)(0x80 | ((c >> 12) & 0x3f))); out.push_back((char)(0x80 | ((c >> 6) & 0x3f))); out.push_back((char)(0x80 | (c & 0x3f))); }
}

// Byte strings (forward order) a haystack may contain where the lowered haystack has needle code
// point `c`.  IgnoreCase: UTF-8 of every x with lower(x) == c, ASCII bytes folded to lower case
// (the kernels fold haystack bytes the same way).  CaseSensitive: just UTF-8 of c.
static void variants_of(const LowerTable& lt, uint32_t c, bool ignore_case, std::vector<std::string>& out)
{
    out.clear();
    if (!ignore_case) { std::string s; append_utf8(c, s); out.push_back(s); return; }
    std::vector<uint32_t> xs;
    lt.unlower(c, xs);
    for (uint32_t x : xs) {
        std::string s;
        append_utf8(x < 128 ? fold_byte(x) : x, s);
        if (std::find(out.begin(), out.end(), s) == out.end()) out.push_back(s);
    }
}

// ------------------------------------------------------------------ image assembly

namespace {

struct Blob {
    std::vector<uint8_t> bytes;
    uint64_t reserve_section(size_t nbytes)
    {
        size_t off = (bytes.size() + 255) & ~(size_t)255;
        bytes.resize(off + nbytes, 0);
        return off;
    }
    template <class T> uint64_t put(const std::vector<T>& v)
    {
        uint64_t off = reserve_section(v.size() * sizeof(T));
        if (!v.empty()) std::memcpy(bytes.data() + off, v.data(), v.size() * sizeof(T));
        return off;
    }
};

uint32_t log2_ceil(uint64_t n) { uint32_t l = 0; while ((1ull << l) < n) l++; return l; }

struct TierEntry { uint32_t key, node; };
constexpr size_t kDfaSmallStates = 32768;     // automata up to this many states get a DFA section unasked (am_flatten.cpp, DFA section)

}  // namespace

// AM_FLATTEN_TRACE=1: the phases of a flatten with their wall time on stderr (measurements; bench.py's build split)
struct FlattenTrace {
    bool on; std::chrono::steady_clock::time_point t0, last;
    std::chrono::steady_clock::time_point sub_last = std::chrono::steady_clock::now();
    FlattenTrace() : on(cfg::on(cfg::kFlattenTrace)), t0(std::chrono::steady_clock::now()), last(t0) {}
    void mark(const char* what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[flatten] %-28s %8.1f ms (at %8.1f)\n", what, std::chrono::duration<double, std::milli>(now - last).count(), std::chrono::duration<double, std::milli>(now - t0).count());
        last = now; sub_last = now;
    }
    // a step inside a phase (AM_FLATTEN_TRACE=2): its own time; the phase's mark() still reports the whole phase
    void sub(const char* what)
    {
        if (!on || cfg::get(cfg::kFlattenTrace) < 2) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[flatten]     %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - sub_last).count());
        sub_last = now;
    }
};

int flatten(const RefArrays& ref, int case_mode, std::vector<uint8_t>& image, std::string& err, const LowerTable* lower_table)
{
    FlattenTrace tr;
    struct JoinOnExit { std::future<void>& f; ~JoinOnExit() { if (f.valid()) f.wait(); } };      // a task reads this function's locals: no return leaves it running
    const LowerTable& lt = lower_table ? *lower_table : builtin_lower_table();
    const size_t S = ref.n_states;
    const bool ic = case_mode == 1;
    if (case_mode != 0 && case_mode != 1) { err = "case_mode must be 0 (CaseSensitive) or 1 (IgnoreCase)"; return -1; }
    if (S == 0 || !ref.transitions || !ref.offsets || !ref.root_ascii || !ref.values_len) { err = "null or empty automaton arrays"; return -1; }
    if (S >= 0xFFFFFFFEull) { err = "too many states"; return -1; }

    // ---- read the trie back out of the packed transitions (Automaton.hs:166-172,190-192)
    std::vector<uint32_t> parent(S, kNone), cp_in(S, 0), fallback(S, 0), depth(S, 0);
    std::vector<uint32_t> edge_begin(S), edge_count(S);
    for (size_t s = 0; s < S; s++) {
        uint64_t i = ref.offsets[s];
        edge_begin[s] = (uint32_t)i;
        for (;; i++) {
            if (i >= ref.n_transitions) { err = "transition list of a state is not wildcard-terminated"; return -1; }
            const uint64_t t = ref.transitions[i];
            const uint32_t next = (uint32_t)(t >> 32);
            if (next >= S) { err = "transition to a state out of range"; return -1; }
            if (t & kWildcard) { fallback[s] = next; break; }
            if (next == 0 || parent[next] != kNone) { err = "goto edges do not form a trie"; return -1; }
            parent[next] = (uint32_t)s;
            cp_in[next] = (uint32_t)(t & 0x1fffffu);
        }
        edge_count[s] = (uint32_t)(i - edge_begin[s]);
        // machineOffsets is a scanl over the per-state list lengths (Automaton.hs:170): every state owns its own run of
        // entries, wildcard last.  Arrays that share entries between states would pass the checks above and then break the
        // sizing of the tables below.
        if (s + 1 < S && ref.offsets[s + 1] != i + 1) { err = "offsets do not partition the transition array"; return -1; }
        if (s + 1 == S && i + 1 != ref.n_transitions) { err = "transition array has entries beyond the last state's list"; return -1; }
    }
    std::vector<uint32_t> bfs; bfs.reserve(S); bfs.push_back(0);
    for (size_t q = 0; q < bfs.size(); q++) {
        const uint32_t s = bfs[q];
        for (uint32_t k = 0; k < edge_count[s]; k++) {
            const uint32_t nx = (uint32_t)(ref.transitions[edge_begin[s] + k] >> 32);
            depth[nx] = depth[s] + 1;
            bfs.push_back(nx);
        }
    }
    if (bfs.size() != S) { err = "unreachable states in the automaton"; return -1; }

    // own values / canonical output state (Automaton.hs:367-380: values[s] = own ++ values[fallback s])
    std::vector<uint32_t> canon(S, 0), vlen(ref.values_len, ref.values_len + S);
    std::vector<uint8_t> owns(S, 0);
    uint32_t max_needle_cps = 0;
    for (uint32_t s : bfs) {
        if (s == 0) { owns[0] = vlen[0] > 0; continue; }
        if (depth[fallback[s]] >= depth[s]) { err = "fallback edge does not point to a shallower state"; return -1; }
        if (vlen[s] < vlen[fallback[s]]) { err = "values_len is not consistent with the fallback chain"; return -1; }
        owns[s] = vlen[s] > vlen[fallback[s]];
        canon[s] = owns[s] ? s : canon[fallback[s]];
        if (owns[s]) max_needle_cps = std::max(max_needle_cps, depth[s]);
    }

    ImageHeader h;
    std::memset(&h, 0, sizeof(h));
    h.magic = kImageMagic; h.version = kImageVersion; h.case_mode = (uint32_t)case_mode; h.flags = lt.hash;
    h.n_states = (uint32_t)S; h.max_needle_cps = max_needle_cps; h.root_vlen = vlen[0];
    {
        const uint64_t warm = 4ull * (max_needle_cps ? max_needle_cps : 1) + 4;
        uint64_t chunk = 256;
        while (chunk < 4 * warm && chunk < (1u << 20)) chunk <<= 1;
        h.ac_chunk = (uint32_t)chunk;
    }

    Blob blob;
    // (address space for the whole image up front -- about 190 bytes per state at 100k needles, 225 for a dictionary with its DFA section: the sections are appended one
    // after the other, and a vector that grows by doubling copies what it holds, and faults its pages in, again and again)
    blob.bytes.reserve(std::min<size_t>((size_t)1 << 31, S * 256 + ref.n_transitions * 8 + ((size_t)16 << 20)));
    blob.reserve_section(sizeof(ImageHeader));

    // ---- AC section: the reference's arrays verbatim
    h.n_transitions = ref.n_transitions;
    h.off_transitions = blob.reserve_section(ref.n_transitions * 8);
    std::memcpy(blob.bytes.data() + h.off_transitions, ref.transitions, ref.n_transitions * 8);
    h.off_offsets = blob.reserve_section((S + 1) * 4);
    std::memcpy(blob.bytes.data() + h.off_offsets, ref.offsets, (S + 1) * 4);
    h.off_root_ascii = blob.reserve_section(128 * 8);
    std::memcpy(blob.bytes.data() + h.off_root_ascii, ref.root_ascii, 128 * 8);
    h.off_canon = blob.put(canon);
    h.off_vlen = blob.put(vlen);
    // goto hash + fallback array: what the general kernel walks instead of the reference's per-state edge lists.  Their place in the image is reserved here; the
    // table is filled on a thread of its own while this one builds the suffix structure, and copied in before the checksum.
    std::vector<u32x4> goto_tab; std::vector<uint32_t> goto_fail;
    std::future<void> goto_task;
    JoinOnExit goto_join{goto_task};
    {
        const uint64_t n_edges = S - 1;                       // one goto edge per non-root state (checked above: the edges form a trie)
        uint32_t lc = 4;
        while ((1ull << lc) < 2 * n_edges + 8) lc++;
        if (lc > 31) { err = "automaton too large for the goto table"; return -1; }
        h.ac_goto_log2_cap = lc;
        h.off_goto = blob.reserve_section(((size_t)1 << lc) * sizeof(u32x4));
        h.off_fail = blob.reserve_section(S * sizeof(uint32_t));
        auto fill = [&ref, &goto_tab, &goto_fail, S, lc] {
            goto_tab.assign((size_t)1 << lc, u32x4{0, 0, 0, 0});
            goto_fail.assign(S, 0);
            const uint32_t mask = (1u << lc) - 1u;
            for (uint32_t st = 0; st < (uint32_t)S; st++) {
                for (uint64_t i = ref.offsets[st];; i++) {
                    const uint64_t t = ref.transitions[i];
                    if (t & kWildcard) { goto_fail[st] = (uint32_t)(t >> 32); break; }
                    uint32_t slot = ac_goto_slot(st, (uint32_t)(t & 0x1fffffu), lc);
                    while (goto_tab[slot].w) slot = (slot + 1u) & mask;
                    goto_tab[slot] = u32x4{st, (uint32_t)(t & 0x1fffffu), (uint32_t)(t >> 32), 1u};
                }
            }
        };
        if (cfg::on(cfg::kFlattenSerial)) fill();
        else try { goto_task = std::async(std::launch::async, fill); } catch (const std::system_error&) { fill(); }
    }
    if (ic) {
        const uint32_t n_lower = (lt.from.back() + 256u) & ~255u;                 // (never empty: the ASCII pairs are always there)
        std::vector<int32_t> delta(n_lower, 0);
        for (size_t i = 0; i < lt.from.size(); i++) delta[lt.from[i]] = (int32_t)lt.to[i] - (int32_t)lt.from[i];
        h.n_lower = n_lower;
        h.off_lower = blob.put(delta);
    } else {
        h.n_lower = 0;
        h.off_lower = blob.reserve_section(16);
    }

    tr.mark("trie + AC section");
    // ---- SF section.  The suffix filter answers "which is the deepest needle that ENDS at this position".  With the
    // empty needle among the needles the root owns values, every state's list contains them (Automaton.hs:373-376), and the
    // reference folds them wherever the automaton is not at the root after a code point (collectMatches runs after every
    // successful goto, :502-503,519) -- i.e. wherever some needle PREFIX ends.  Position-parallel form of that rule:
    //   * a prefix of one code point ends here  <=>  the (lowered) code point is the first of some needle: a per-position
    //     test on the text alone (ends_first_code_point below), done by the dense pass of the ABI layer (k_dense_*);
    //   * longer prefixes whose LAST code point starts no needle (the blank in "new york" when nothing starts with a
    //     blank) become extra terminals of the reversed trie, reporting canon[s] -- the list the reference folds there.
    // No automaton is refused (round 2 kept the general kernel when the extra terminals outnumbered the real ones several times over: that
    // kernel is 50-100 x slower, so the suffix structure simply grows -- its tables are sized by the number of terminals either way, and
    // such an automaton reports at almost every position, i.e. it is bound by its output, not by the tables).
    h.sf_enabled = 1u;
    std::vector<uint32_t> terminals;
    for (size_t s = 1; s < S; s++) if (owns[s]) terminals.push_back((uint32_t)s);
    if (vlen[0] > 0) {
        std::vector<uint32_t> first;
        for (size_t s = 1; s < S; s++) if (parent[s] == 0) first.push_back(cp_in[s]);
        std::sort(first.begin(), first.end());
        std::vector<uint32_t> extra;
        for (size_t s = 1; s < S; s++)
            if (!owns[s] && depth[s] >= 2 && !std::binary_search(first.begin(), first.end(), cp_in[s])) extra.push_back((uint32_t)s);
        terminals.insert(terminals.end(), extra.begin(), extra.end());
    }

    std::vector<SfNode> nodes;
    std::vector<SfEdge> edges_out;
    std::vector<uint32_t> many_nodes;                 // nodes with more than 4 children (they get a row of edge lines)
    uint32_t row_first = 0;
    std::vector<TierEntry> tier_entries[4];

    if (h.sf_enabled && !terminals.empty()) {
        // reversed needles as code point strings: walking parent links yields c_n, c_(n-1), ..., c_1
        std::vector<uint32_t> pool; std::vector<uint64_t> str_off(terminals.size() + 1, 0);
        for (size_t k = 0; k < terminals.size(); k++) {
            for (uint32_t s = terminals[k]; s != 0; s = parent[s]) pool.push_back(cp_in[s]);
            str_off[k + 1] = pool.size();
        }
        std::vector<uint32_t> order(terminals.size());
        for (size_t k = 0; k < order.size(); k++) order[k] = (uint32_t)k;
        // (lexicographic order of the code point strings; the first three code points of a string, 21 bits each, decide most comparisons as one integer -- a code point
        // + 1 each, so that "no third code point" sorts before every code point, as a shorter string does)
        std::vector<uint64_t> head(terminals.size());
        for (size_t k = 0; k < head.size(); k++) {
            const size_t len = (size_t)(str_off[k + 1] - str_off[k]);
            uint64_t key = 0;
            for (size_t j = 0; j < 3; j++) key = (key << 21) | (j < len ? (uint64_t)pool[str_off[k] + j] + 1u : 0u);
            head[k] = key;
        }
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            if (head[a] != head[b]) return head[a] < head[b];
            return std::lexicographical_compare(pool.begin() + str_off[a], pool.begin() + str_off[a + 1],
                                                pool.begin() + str_off[b], pool.begin() + str_off[b + 1]);
        });

        tr.sub("sf: reversed needles sorted");
        // code-point trie of the reversed needles, nodes created in preorder
        struct CpEdge { uint32_t src, cp, dst; };
        std::vector<CpEdge> cp_edges;
        std::vector<uint32_t> term_state(1, kNone);     // per cp-node: reference state or kNone
        std::vector<uint32_t> path(1, 0);               // node ids along the previous string
        const uint32_t* prev = nullptr; size_t prev_len = 0;
        for (uint32_t k : order) {
            const uint32_t* str = pool.data() + str_off[k]; const size_t len = (size_t)(str_off[k + 1] - str_off[k]);
            size_t lcp = 0;
            while (lcp < len && lcp < prev_len && str[lcp] == prev[lcp]) lcp++;
            path.resize(lcp + 1);
            for (size_t j = lcp; j < len; j++) {
                const uint32_t id = (uint32_t)term_state.size();
                term_state.push_back(kNone);
                cp_edges.push_back({path.back(), str[j], id});
                path.push_back(id);
            }
            term_state[path.back()] = terminals[k];
            prev = str; prev_len = len;
        }
        const size_t n_cp_nodes = term_state.size();

        // group cp edges by source (stable: children stay in ascending code point order)
        std::vector<uint32_t> cp_first(n_cp_nodes + 1, 0);
        for (const CpEdge& e : cp_edges) cp_first[e.src + 1]++;
        for (size_t i = 0; i < n_cp_nodes; i++) cp_first[i + 1] += cp_first[i];
        std::vector<CpEdge> cp_sorted(cp_edges.size());
        { std::vector<uint32_t> cur(cp_first.begin(), cp_first.end() - 1);
          for (const CpEdge& e : cp_edges) cp_sorted[cur[e.src]++] = e; }

        tr.sub("sf: code-point trie");
        // expand every code point edge into its (reversed, folded) UTF-8 variants
        struct ByteEdge { uint32_t src, byte, dst; };
        std::vector<ByteEdge> bedges; bedges.reserve(cp_edges.size() + cp_edges.size() / 4);
        uint32_t n_byte_nodes = (uint32_t)n_cp_nodes;
        std::unordered_map<uint32_t, std::vector<std::string>> variant_cache;
        std::vector<ByteEdge> local;
        for (size_t u = 0; u < n_cp_nodes; u++) {
            local.clear();
            for (uint32_t e = cp_first[u]; e < cp_first[u + 1]; e++) {
                const uint32_t c = cp_sorted[e].cp, v = cp_sorted[e].dst;
                auto it = variant_cache.find(c);
                if (it == variant_cache.end()) { std::vector<std::string> vs; variants_of(lt, c, ic, vs); it = variant_cache.emplace(c, std::move(vs)).first; }
                for (const std::string& var : it->second) {
                    uint32_t cur = (uint32_t)u;
                    for (size_t j = var.size(); j-- > 1;) {            // last byte first; all but the lead byte
                        const uint32_t b = (uint8_t)var[j];
                        uint32_t next = kNone;
                        for (const ByteEdge& le : local) if (le.src == cur && le.byte == b) { next = le.dst; break; }
                        if (next == kNone) { next = n_byte_nodes++; local.push_back({cur, b, next}); bedges.push_back({cur, b, next}); }
                        cur = next;
                    }
                    bedges.push_back({cur, (uint32_t)(uint8_t)var[0], v});
                }
            }
        }
        if (n_byte_nodes >= 0xFFFFFFF0u) { err = "automaton too large for 32-bit node ids"; return -1; }

        tr.sub("sf: byte variants");
        // adjacency of the byte graph, children sorted by byte
        std::vector<uint32_t> b_first(n_byte_nodes + 1, 0);
        for (const ByteEdge& e : bedges) b_first[e.src + 1]++;
        for (size_t i = 0; i < n_byte_nodes; i++) b_first[i + 1] += b_first[i];
        std::vector<ByteEdge> b_sorted(bedges.size());
        { std::vector<uint32_t> cur(b_first.begin(), b_first.end() - 1);
          for (const ByteEdge& e : bedges) b_sorted[cur[e.src]++] = e; }
        for (size_t i = 0; i < n_byte_nodes; i++)
            if (b_first[i + 1] - b_first[i] > 1)
                std::sort(b_sorted.begin() + b_first[i], b_sorted.begin() + b_first[i + 1],
                          [](const ByteEdge& a, const ByteEdge& b) { return a.byte < b.byte; });

        tr.sub("sf: byte adjacency");
        // ---- path compression.  A node is absorbed into its incoming edge when it has exactly one
        // parent, exactly one child, no needle ends at it, and no path reaches it with <= 4 bytes
        // (every node within 4 bytes of the root stays explicit: the suffix tables point at them).
        std::vector<uint32_t> indeg(n_byte_nodes, 0), min_depth(n_byte_nodes, 0xFFFFFFFFu);
        for (const ByteEdge& e : b_sorted) indeg[e.dst]++;
        {
            std::vector<uint32_t> bq; bq.push_back(0); min_depth[0] = 0;
            for (size_t qi = 0; qi < bq.size(); qi++) {
                const uint32_t x = bq[qi];
                for (uint32_t e = b_first[x]; e < b_first[x + 1]; e++) {
                    const uint32_t y = b_sorted[e].dst;
                    if (min_depth[y] == 0xFFFFFFFFu) { min_depth[y] = min_depth[x] + 1; bq.push_back(y); }
                }
            }
        }
        auto is_terminal = [&](uint32_t x) { return x < n_cp_nodes && term_state[x] != kNone; };
        auto mergeable = [&](uint32_t x) {
            return indeg[x] == 1 && b_first[x + 1] - b_first[x] == 1 && !is_terminal(x) && min_depth[x] != 0xFFFFFFFFu && min_depth[x] > 4;
        };
        struct CEdge { uint32_t src, byte, dst; uint8_t skip[kMaxSkip]; uint32_t n_skip; };    // skip bytes in walk order (in the edge itself: a vector each was a million allocations)
        std::vector<CEdge> cedges;
        std::vector<uint8_t> explicit_node(n_byte_nodes, 0);
        {
            std::vector<uint32_t> work; work.push_back(0); explicit_node[0] = 1;
            while (!work.empty()) {
                const uint32_t u = work.back(); work.pop_back();
                for (uint32_t e = b_first[u]; e < b_first[u + 1]; e++) {
                    CEdge ce{u, b_sorted[e].byte, b_sorted[e].dst, {0}, 0u};
                    while (mergeable(ce.dst) && ce.n_skip < kMaxSkip) {
                        const ByteEdge& nx = b_sorted[b_first[ce.dst]];
                        ce.skip[ce.n_skip++] = (uint8_t)nx.byte;
                        ce.dst = nx.dst;
                    }
                    if (!explicit_node[ce.dst]) { explicit_node[ce.dst] = 1; work.push_back(ce.dst); }
                    cedges.push_back(ce);
                }
            }
        }
        tr.sub("sf: path compression");
        // adjacency of the compressed graph (edges of one source stay in selector-byte order)
        std::vector<uint32_t> c_first(n_byte_nodes + 1, 0);
        for (const CEdge& e : cedges) c_first[e.src + 1]++;
        for (size_t i = 0; i < n_byte_nodes; i++) c_first[i + 1] += c_first[i];
        std::vector<uint32_t> c_order(cedges.size());
        { std::vector<uint32_t> cur(c_first.begin(), c_first.end() - 1);
          for (uint32_t i = 0; i < cedges.size(); i++) c_order[cur[cedges[i].src]++] = i; }
        for (size_t i = 0; i < n_byte_nodes; i++)
            if (c_first[i + 1] - c_first[i] > 1)
                std::sort(c_order.begin() + c_first[i], c_order.begin() + c_first[i + 1], [&](uint32_t a, uint32_t b) { return cedges[a].byte < cedges[b].byte; });

        tr.sub("sf: compressed adjacency");
        // renumber explicit nodes in DFS preorder (a needle's tail is a run of consecutive records)
        std::vector<uint32_t> new_id(n_byte_nodes, kNone), stack;
        uint32_t next_id = 0;
        stack.push_back(0);
        while (!stack.empty()) {
            const uint32_t x = stack.back(); stack.pop_back();
            if (new_id[x] != kNone) continue;
            new_id[x] = next_id++;
            for (uint32_t e = c_first[x + 1]; e-- > c_first[x];) { const uint32_t y = cedges[c_order[e]].dst; if (new_id[y] == kNone) stack.push_back(y); }
        }
        auto pack_label = [](const CEdge& ce, uint32_t (&label)[4]) {
            // walk order = backwards in the haystack; store in text order, right-aligned in 16 bytes
            uint8_t bytes[kMaxSkip] = {0};
            const uint8_t* skip = ce.skip;
            const size_t n = ce.n_skip;
            for (size_t j = 0; j < n; j++) bytes[kMaxSkip - 1 - j] = skip[j];
            for (int i = 0; i < 4; i++) label[i] = (uint32_t)bytes[4 * i] | ((uint32_t)bytes[4 * i + 1] << 8) | ((uint32_t)bytes[4 * i + 2] << 16) | ((uint32_t)bytes[4 * i + 3] << 24);
        };
        nodes.assign(next_id, SfNode{0, 0, 0, 0, {0, 0, 0, 0}});
        for (uint32_t x = 0; x < n_byte_nodes; x++) {
            if (new_id[x] == kNone) continue;
            SfNode& rec = nodes[new_id[x]];
            if (is_terminal(x)) { rec.x = canon[term_state[x]] + 1; rec.y = vlen[term_state[x]]; }      // canon[s] == s for needle ends; the prefix terminals of an empty-needle automaton report the deepest owner on their fallback chain
            const uint32_t n = c_first[x + 1] - c_first[x];
            if (n > 0xFFFF) { err = "node fan-out exceeds 65535"; return -1; }
            if (n == 1) {
                const CEdge& ce = cedges[c_order[c_first[x]]];
                rec.z = new_id[ce.dst];
                rec.w = 1u | (ce.byte << 16) | (ce.n_skip << 24);
                pack_label(ce, rec.label);
            } else if (n > 1) {
                rec.z = (uint32_t)edges_out.size();
                rec.w = n;
                if (n <= 4) for (uint32_t i = 0; i < n; i++) rec.label[0] |= cedges[c_order[c_first[x] + i]].byte << (8u * i);   // inline selectors
                else {
                    // more children: the node gets a ROW of edge lines below (row displacement); here only the order of the edges is checked
                    uint32_t prev = 0;
                    for (uint32_t i = 0; i < n; i++) {
                        const uint32_t b = cedges[c_order[c_first[x] + i]].byte & 0xFFu;
                        if (i && b <= prev) { err = "edges of a node are not sorted by selector byte (internal error)"; return -1; }
                        prev = b;
                        const uint32_t f = b % 96u;
                        rec.label[1 + f / 32u] |= 1u << (f & 31u);
                    }
                    many_nodes.push_back(new_id[x]);
                }
                for (uint32_t e = c_first[x]; e < c_first[x + 1]; e++) {
                    const CEdge& ce = cedges[c_order[e]];
                    SfEdge ed{ce.byte, new_id[ce.dst], ce.n_skip, 0, {0, 0, 0, 0}, SfNode{0, 0, 0, 0, {0, 0, 0, 0}}};
                    pack_label(ce, ed.label);
                    edges_out.push_back(ed);
                }
            }
        }
        tr.sub("sf: node and edge records");
        // Row displacement for the nodes with more than 4 children: node -> row offset r such that the lines r + b of all its selector bytes b are free
        // (first fit, largest nodes first); the edge of byte b is then ONE load away, edges[label[0] + b], and that line names its owner (SfEdge::pad).
        row_first = (uint32_t)edges_out.size();
        std::vector<uint32_t> row_of(many_nodes.size(), 0);
        // (which lines are taken: a bit each, and the number of lines the table has so far -- a line beyond it is free)
        std::vector<uint64_t> occ_bits(64, 0);
        size_t occ_n = 0;
        auto occ_reserve = [&](size_t lines) { const size_t w = (lines >> 6) + 8; if (occ_bits.size() < w) occ_bits.resize(2 * w, 0); };
        {
            std::vector<uint32_t> order(many_nodes.size());
            for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return (nodes[many_nodes[a]].w & 0xFFFFu) > (nodes[many_nodes[b]].w & 0xFFFFu); });
            size_t first_free = 0;
            for (uint32_t oi : order) {
                const SfNode& nd = nodes[many_nodes[oi]];
                const uint32_t n = nd.w & 0xFFFFu, b_min = edges_out[nd.z].byte, b_max = edges_out[nd.z + n - 1].byte;
                while (first_free < occ_n && ((occ_bits[first_free >> 6] >> (first_free & 63u)) & 1u)) first_free++;
                size_t r = first_free > b_min ? first_free - b_min : 0;
                // first fit within 8192 rows, 64 rows at a time: row base + k fits iff the line base + k + b is free for every selector byte b, i.e. bit k of the AND over b of
                // ~(the bitset shifted to base + b).  The same row as testing one row after the other (what this replaces: 270 of the phase's 560 ms at 100k needles).
                {
                    const size_t r_first = r;
                    bool found = false;
                    for (size_t base = r_first; base < r_first + 8192 && !found; base += 64) {
                        occ_reserve(base + 64 + 256 + 128);
                        uint64_t fits = ~0ull;
                        for (uint32_t i = 0; i < n && fits; i++) {
                            const size_t x = base + (edges_out[nd.z + i].byte & 0xFFu);
                            const uint32_t sh = (uint32_t)(x & 63u);
                            const uint64_t taken = sh ? (occ_bits[x >> 6] >> sh) | (occ_bits[(x >> 6) + 1] << (64u - sh)) : occ_bits[x >> 6];      // (no bit is set beyond occ_n: free)
                            fits &= ~taken;
                        }
                        const size_t left = r_first + 8192 - base;                   // rows of this block that are still among the 8192
                        if (left < 64) fits &= (1ull << left) - 1ull;
                        if (fits) { r = base + (size_t)__builtin_ctzll(fits); found = true; }
                    }
                    if (!found) r = occ_n;                                            // (a crowded table: open a fresh stretch instead of searching on)
                }
                if (occ_n < r + b_max + 1) occ_n = r + b_max + 1;
                occ_reserve(occ_n + 320);
                for (uint32_t i = 0; i < n; i++) { const size_t at = r + edges_out[nd.z + i].byte; occ_bits[at >> 6] |= 1ull << (at & 63u); }
                row_of[oi] = (uint32_t)r;
            }
            if (!many_nodes.empty()) occ_n += 256;                                // label[0] + b stays inside the array for every byte b
            if ((uint64_t)row_first + occ_n >= 0xFFFFFFF0ull) { err = "too many edge lines"; return -1; }
            for (uint32_t i = 0; i < many_nodes.size(); i++) nodes[many_nodes[i]].label[0] = row_first + row_of[i];
        }
        tr.sub("sf: row displacement (first fit)");
        for (SfEdge& ed : edges_out) ed.to = nodes[ed.child];          // every edge's line carries its child's (now final) record
        tr.sub("sf: children's records into the edges");
        if (!many_nodes.empty()) {
            edges_out.resize((size_t)row_first + occ_n, SfEdge{0, 0, 0, kNone, {0, 0, 0, 0}, SfNode{0, 0, 0, 0, {0, 0, 0, 0}}});
            for (uint32_t i = 0; i < many_nodes.size(); i++) {
                const SfNode& nd = nodes[many_nodes[i]];
                for (uint32_t e = 0; e < (nd.w & 0xFFFFu); e++) {
                    SfEdge line = edges_out[nd.z + e];
                    line.pad = nd.z + 1u;
                    edges_out[(size_t)nd.label[0] + line.byte] = line;
                }
            }
        }

        tr.sub("sf: displaced rows written");
        // suffix tables: every byte path of length <= 4 from the root (never inside a compressed edge)
        struct Frame { uint32_t node, depth, key; };
        std::vector<Frame> fs; fs.push_back({0, 0, 0});
        while (!fs.empty()) {
            const Frame f = fs.back(); fs.pop_back();
            for (uint32_t e = c_first[f.node]; e < c_first[f.node + 1]; e++) {
                const CEdge& ce = cedges[c_order[e]];
                if (ce.n_skip) { err = "compressed edge within 4 bytes of the root (internal error)"; return -1; }
                const uint32_t d = f.depth + 1, child = ce.dst;
                const uint32_t key = f.key | (ce.byte << (32u - 8u * d));
                if (d == 4) { tier_entries[3].push_back({key, new_id[child]}); continue; }
                if (is_terminal(child)) tier_entries[d - 1].push_back({key >> (8u * (4u - d)), new_id[child]});
                fs.push_back({child, d, key});
            }
        }
    }

    tr.mark("suffix trie");
    struct DfaOut {
        bool made = false;
        std::vector<uint32_t> next2, hot2, fb2; std::vector<u32x2> chain, out2; std::vector<uint8_t> cls; std::vector<u32x4> rare_tab, chain2; uint32_t n_single = 0;
        uint32_t hot_lc = 0, rare_lc = 0, n_rows = 0, n_states = 0, lc = 0, warm = 0, chunk = 0;
    };
    // (reads what the trie phase left -- bfs, edge_begin / edge_count, vlen, canon, the reference's arrays -- and nothing the suffix structure makes: it runs on a thread
    // of its own next to the filter and suffix tables, started as soon as the automaton is known to be a dictionary)
    auto make_dfa = [&](DfaOut& o) {
        FlattenTrace dtr;
        if (vlen[0] == 0 && S > 1) {
            struct BEdge { uint32_t src, byte, dst; };
            std::vector<BEdge> be; be.reserve(S + S / 8);
            uint32_t n_nodes = (uint32_t)S;
            std::vector<uint32_t> bdepth(S, 0);                                  // longest spelling of the state's string, in bytes
            std::unordered_map<uint32_t, std::vector<std::string>> vcache;
            std::vector<BEdge> local;
            bool present[256] = {false};
            for (uint32_t u : bfs) {
                local.clear();
                for (uint32_t k = 0; k < edge_count[u]; k++) {
                    const uint64_t t = ref.transitions[edge_begin[u] + k];
                    const uint32_t c = (uint32_t)(t & 0x1fffffu), v = (uint32_t)(t >> 32);
                    auto it = vcache.find(c);
                    if (it == vcache.end()) { std::vector<std::string> vs; variants_of(lt, c, ic, vs); it = vcache.emplace(c, std::move(vs)).first; }
                    for (const std::string& var : it->second) {
                        uint32_t cur = u;
                        for (size_t j = 0; j + 1 < var.size(); j++) {
                            const uint32_t b = (uint8_t)var[j];
                            uint32_t nx = kNone;
                            for (const BEdge& le : local) if (le.src == cur && le.byte == b) { nx = le.dst; break; }
                            if (nx == kNone) { nx = n_nodes++; local.push_back({cur, b, nx}); be.push_back({cur, b, nx}); present[b] = true; }
                            cur = nx;
                        }
                        be.push_back({cur, (uint32_t)(uint8_t)var.back(), v});
                        present[(uint8_t)var.back()] = true;
                        bdepth[v] = std::max(bdepth[v], bdepth[u] + (uint32_t)var.size());
                    }
                }
            }
            dtr.mark("dfa: byte edges");
            // Classes: a row has a column for the COMMON bytes only -- the fewest (a power of two, with class 0) that label all but a thousandth of the edges.  In a
            // dictionary 31 bytes do (the letters, the blank, the lead bytes of the accented ones); the bytes of a few Cyrillic words and of the upper-case spellings of
            // non-ASCII letters would double the row twice for nothing.  A RARE byte (class kDfaRare) takes the textbook route instead: the state's own edge on it (a
            // small hash of the rare edges) or the same question at the state's fallback (dfa_rare_step in am_image.h).
            uint64_t edge_cnt[256] = {0};
            for (const BEdge& e : be) edge_cnt[e.byte]++;
            std::vector<uint32_t> by_cnt;
            for (uint32_t b = 0; b < 256; b++) if (present[b]) by_cnt.push_back(b);
            std::sort(by_cnt.begin(), by_cnt.end(), [&](uint32_t a, uint32_t b2) { return edge_cnt[a] != edge_cnt[b2] ? edge_cnt[a] > edge_cnt[b2] : a < b2; });
            const uint64_t rare_permille = cfg::get(cfg::kDfaRarePermille) >= 0 ? (uint64_t)cfg::get(cfg::kDfaRarePermille) : 1ull;
            uint32_t lc = 3;
            for (;; lc++) {
                uint64_t left_out = 0;
                for (size_t k = (1u << lc) - 1u; k < by_cnt.size(); k++) left_out += edge_cnt[by_cnt[k]];
                if (left_out * 1000ull <= be.size() * rare_permille || lc == 8) break;
            }
            uint32_t n_cls = 1;
            std::vector<uint8_t> cls(256, 0);
            for (size_t k = 0; k < by_cnt.size(); k++) cls[by_cnt[k]] = k + 1u < (1u << lc) ? (uint8_t)n_cls++ : (uint8_t)kDfaRare;
            if (ic) for (uint32_t b = 'A'; b <= 'Z'; b++) cls[b] = cls[b + 0x20u];       // the kernels fold ASCII; the variants hold the folded byte only
            const uint64_t table_bytes = ((uint64_t)n_nodes << lc) * 4ull;
            if (n_nodes < kDfaStateMask && table_bytes <= (1ull << 30)) {
                dtr.mark("dfa: classes");
                // adjacency by source
                std::vector<uint32_t> first(n_nodes + 1, 0);
                for (const BEdge& e : be) first[e.src + 1]++;
                for (uint32_t i = 0; i < n_nodes; i++) first[i + 1] += first[i];
                std::vector<BEdge> adj(be.size());
                { std::vector<uint32_t> cur(first.begin(), first.end() - 1); for (const BEdge& e : be) adj[cur[e.src]++] = e; }
                // breadth-first numbering; a node's fallback is set when it is first reached: delta(fallback(parent), byte), the root's children fall back to the root
                std::vector<uint32_t> id(n_nodes, kNone), order; order.reserve(n_nodes);
                std::vector<uint32_t> fb(n_nodes, 0);                            // fallback, as a breadth-first number
                std::vector<uint32_t> tree_parent(n_nodes, 0);                   // the state that discovered it ...
                std::vector<uint8_t> tree_byte(n_nodes, 0);                      // ... and the byte of that edge
                std::vector<uint32_t> n_goto(n_nodes, 0), child_of(n_nodes, 0);  // edges of a state; the child of its last common-byte edge ...
                std::vector<uint8_t> child_cls(n_nodes, 0), has_rare(n_nodes, 0); // ... and that edge's class; does it have an edge on a rare byte
                std::vector<uint32_t> next((size_t)n_nodes << lc, 0);            // every state's dense row for now (what the rows of the image are cut from)
                const uint32_t C = 1u << lc;
                std::unordered_map<uint64_t, uint32_t> rare_goto;                // (state << 8 | byte) -> child, the edges on rare bytes
                auto delta_rare = [&](uint32_t st, uint32_t byte) {              // delta(st, rare byte) by the fallback chain
                    for (;;) {
                        const auto it = rare_goto.find(((uint64_t)st << 8) | byte);
                        if (it != rare_goto.end()) return it->second;
                        if (st == 0) return 0u;
                        st = fb[st];
                    }
                };
                id[0] = 0; order.push_back(0);
                for (size_t qi = 0; qi < order.size(); qi++) {
                    const uint32_t x = order[qi], xi = (uint32_t)qi;
                    uint32_t* row = next.data() + ((size_t)xi << lc);
                    if (xi != 0) std::memcpy(row, next.data() + ((size_t)fb[xi] << lc), (size_t)C * 4);      // (the root's row starts as all-root = zeros)
                    for (uint32_t e = first[x]; e < first[x + 1]; e++) {
                        const uint32_t y = adj[e].dst, byte = adj[e].byte, cb = cls[byte];
                        if (id[y] == kNone) {
                            id[y] = (uint32_t)order.size(); order.push_back(y);
                            fb[id[y]] = xi == 0 ? 0u : cb == kDfaRare ? delta_rare(fb[xi], byte) : next[((size_t)fb[xi] << lc) + cb];
                            tree_parent[id[y]] = xi; tree_byte[id[y]] = (uint8_t)byte;
                        }
                        n_goto[xi]++;
                        if (cb == kDfaRare) { rare_goto[((uint64_t)xi << 8) | byte] = id[y]; has_rare[xi] = 1; }
                        else { row[cb] = id[y]; child_of[xi] = id[y]; child_cls[xi] = (uint8_t)cb; }
                    }
                }
                dtr.mark("dfa: rows of all states (breadth-first)");
                {
                    // (nodes no byte string reaches -- an upper-case needle letter under IgnoreCase has no spelling -- get no state: the reference never reaches them either)
                    const uint32_t n_reached = (uint32_t)order.size();
                    std::vector<u32x2> out(n_reached, u32x2{0, 0});
                    for (uint32_t x = 1; x < (uint32_t)S; x++) if (vlen[x] > 0 && id[x] != kNone) out[id[x]] = u32x2{canon[x] + 1u, vlen[x]};
                    // ROW states and RECORD states.  A dense row is 256 bytes of which a visitor reads four, and deep in a dictionary a state's row differs from the row of a
                    // state it falls back to in an entry or two (its own child; a child one of the states on the way there has).  Such a state keeps a RECORD instead: the row
                    // state R it leans on -- the nearest one on its chain of fallbacks -- and the one or two classes on which it differs from R's row, with where they lead
                    // (8 or 16 bytes); any other byte is answered by R's row: delta(x, c) = delta(R, c).  A step is a record and, at most, one row entry.
                    // (Image version 17.  Until then a chain state had one child and leaned on the state it falls back to, which for that was given a row: 125k of the
                    // dictionary's 143k rows were rows of states with at most two children -- 88 % of the rows for 19 % of the steps, and most of the walk's L2 misses.
                    // Records that lean on records, followed hop by hop, have the fewest rows of all and were measured first: the hops are dependent trips, 10.1 ms against
                    // 8.1 per 2 GiB -- LABNOTES R6.8.)  Rows: the root, the states that differ from their R in three entries or more, the states with an edge on a rare byte.
                    // Single-child records are numbered along their paths (a word's tail shares cache lines).
                    struct Rec { uint32_t n = 0, cls[2] = {0, 0}, to[2] = {0, 0}; };
                    std::vector<uint8_t> is_row(n_reached, 0);
                    std::vector<uint32_t> near_row(n_reached, 0);                     // R(x): the nearest row state on the chain of fallbacks (x itself excluded)
                    std::vector<Rec> rec(n_reached);
                    is_row[0] = 1;
                    const bool no_chains = cfg::get(cfg::kDfaNoChains) > 0;           // A/B: dense rows for every state (round 5's first layout)
                    for (uint32_t i = 1; i < n_reached; i++) {                        // breadth-first order: fb[i] < i
                        const uint32_t r = is_row[fb[i]] ? fb[i] : near_row[fb[i]];
                        near_row[i] = r;
                        if (has_rare[i] || no_chains) { is_row[i] = 1; continue; }
                        const uint32_t *mine = next.data() + ((size_t)i << lc), *theirs = next.data() + ((size_t)r << lc);
                        Rec q;
                        for (uint32_t c = 1; c < C && q.n <= 2; c++)
                            if (mine[c] != theirs[c]) { if (q.n < 2) { q.cls[q.n] = c; q.to[q.n] = mine[c]; } q.n++; }
                        if (q.n > 2) is_row[i] = 1; else rec[i] = q;
                    }
                    dtr.mark("dfa: row and record states");
                    // numbers: the root, then the row states by weight (k_dfa keeps the first rows in LDS; breadth-first order among equals); then the chain states, path by path
                    // How often will text visit a state?  The dictionary is the one sample of its language the flattener has: the needles, one after the other with a
                    // blank between them, are walked through the automaton and the visits counted (weight[state] = steps that START there; col_use[class] = bytes).
                    // That sees what the number of needles below a state does not: the states "word + blank" that every dictionary word of the text leads to when
                    // some phrase starts with it, and the states a needle's tail falls back to.  (Measured on the natural-text workload, tools/experiments/dfa_visits.py:
                    // the first 1 024 / 16 384 rows by this weight take 44.8 / 73.8 % of the steps, by needles below 41.2 / 65.7 %, by the text's own counts 46.7 / 76.2 %.)
                    std::vector<uint32_t> weight(n_reached, 0);
                    std::vector<uint64_t> col_use(C, 0);
                    {
                        // The walk is one chain of dependent look-ups in a table of 250 bytes per state (95 ms of a dictionary's 270-ms DFA section), so it is cut into
                        // stretches of needles that threads walk side by side.  The state an automaton is in depends on the last `deepest` bytes it has read and on nothing
                        // before them, so a stretch first walks -- without counting -- the needles in front of it that make up that many bytes, and is then in the state the
                        // one walk would be in: the counts add up to the same numbers, whatever the number of threads.
                        std::vector<uint32_t> items;
                        for (uint32_t i = 1; i < n_reached; i++)
                            if (out[i].x && out[i].y > out[fb[i]].y) items.push_back(i);          // a needle of its own ends here (values = own ++ the fallback's, Automaton.hs:367-380)
                        uint32_t deepest = 1;
                        for (uint32_t x = 0; x < (uint32_t)S; x++) deepest = std::max(deepest, bdepth[x]);
                        auto walk_stretch = [&](size_t a, size_t z, uint32_t* w, uint64_t* cu) {
                            std::vector<uint8_t> spell;
                            size_t from = a, have = 0;
                            while (from > 0 && have < deepest) {
                                from--;
                                for (uint32_t y = items[from]; y != 0; y = tree_parent[y]) have++;
                                have++;                                                       // (the blank behind it)
                            }
                            uint32_t st = 0;
                            for (size_t k = from; k < z; k++) {
                                const bool counted = k >= a;
                                spell.clear();
                                spell.push_back(0x20u);
                                for (uint32_t y = items[k]; y != 0; y = tree_parent[y]) spell.push_back(tree_byte[y]);
                                for (size_t j = spell.size(); j-- > 0;) {
                                    const uint32_t byte = spell[j], cb = cls[byte];
                                    if (counted) { w[st]++; if (cb != kDfaRare) cu[cb]++; }
                                    st = cb == kDfaRare ? delta_rare(st, byte) : next[((size_t)st << lc) + cb];
                                }
                            }
                        };
                        const unsigned hw = std::thread::hardware_concurrency();
                        const size_t n_thr = (cfg::on(cfg::kFlattenSerial) || items.size() < 8192) ? 1 : std::max<size_t>(1, std::min<size_t>(8, hw / 2));
                        if (n_thr == 1) walk_stretch(0, items.size(), weight.data(), col_use.data());
                        else {
                            std::vector<std::vector<uint32_t>> ws(n_thr - 1, std::vector<uint32_t>(n_reached, 0));
                            std::vector<std::vector<uint64_t>> cs(n_thr - 1, std::vector<uint64_t>(C, 0));
                            std::vector<std::thread> pool;
                            const size_t per = (items.size() + n_thr - 1) / n_thr;
                            bool threads_ok = true;
                            for (size_t t = 1; t < n_thr && threads_ok; t++) {
                                const size_t a = std::min(items.size(), t * per), z = std::min(items.size(), a + per);
                                try { pool.emplace_back(walk_stretch, a, z, ws[t - 1].data(), cs[t - 1].data()); } catch (const std::system_error&) { threads_ok = false; }
                            }
                            // (a thread that could not be started: its stretch and the ones behind it are walked here)
                            const size_t started = pool.size() + 1;
                            walk_stretch(0, std::min(items.size(), per), weight.data(), col_use.data());
                            if (!threads_ok) walk_stretch(std::min(items.size(), started * per), items.size(), weight.data(), col_use.data());
                            for (auto& th : pool) th.join();
                            for (size_t t = 0; t + 1 < started; t++) {
                                for (uint32_t i = 0; i < n_reached; i++) weight[i] += ws[t][i];
                                for (uint32_t c = 0; c < C; c++) col_use[c] += cs[t][c];
                            }
                        }
                    }
                    dtr.mark("dfa: weights (the needles walked)");
                    std::vector<uint32_t> rows;
                    for (uint32_t i = 1; i < n_reached; i++) if (is_row[i]) rows.push_back(i);
                    auto heavier = [&](uint32_t a, uint32_t b2) { return weight[a] != weight[b2] ? weight[a] > weight[b2] : a < b2; };
                    std::sort(rows.begin(), rows.end(), heavier);          // ALL of them (image version 16): neighbours in the table are about equally hot, and two rows share a line of the hot table
                    std::vector<uint32_t> renum(n_reached, kNone);
                    renum[0] = 0;
                    uint32_t nxt = 1;
                    for (uint32_t i : rows) renum[i] = nxt++;
                    const uint32_t n_rows = nxt;
                    // Columns by how often the dictionary itself uses them (image version 16): a class's weight = its bytes in the walk above.
                    // Until here the classes were numbered by the number of EDGES (what decides which bytes get a column at all); the blank of a dictionary with phrases
                    // labels few edges and is every sixth byte of the text.  The first 2^dfa_hot_log2 columns after column 0 form the hot table (below).
                    std::vector<uint32_t> col_new(C);
                    {
                        const std::vector<uint64_t>& col_weight = col_use;
                        std::vector<uint32_t> by_w;
                        for (uint32_t c = 1; c < C; c++) by_w.push_back(c);
                        std::stable_sort(by_w.begin(), by_w.end(), [&](uint32_t a, uint32_t b2) { return col_weight[a] > col_weight[b2]; });
                        col_new[0] = 0;
                        for (uint32_t k = 0; k < by_w.size(); k++) col_new[by_w[k]] = k + 1u;
                        for (uint32_t b = 0; b < 256; b++) if (cls[b] != kDfaRare) cls[b] = (uint8_t)col_new[cls[b]];
                    }
                    // the single-child records (and the childless ones) path by path, the paths by the weight of their heads: the tails of the words text is made of share lines
                    // with each other, not with the tails of words that never come; then the two-children records by weight
                    auto is_single = [&](uint32_t y) { return !is_row[y] && rec[y].n <= 1; };
                    {
                        std::vector<uint32_t> heads;
                        for (uint32_t i = 1; i < n_reached; i++) if (is_single(i) && !is_single(tree_parent[i])) heads.push_back(i);
                        std::stable_sort(heads.begin(), heads.end(), [&](uint32_t a, uint32_t b2) { return weight[a] > weight[b2]; });
                        for (uint32_t i : heads)
                            for (uint32_t y = i; is_single(y) && renum[y] == kNone;) {
                                renum[y] = nxt++;
                                if (rec[y].n != 1) break;
                                y = rec[y].to[0];
                            }
                    }
                    for (uint32_t i = 1; i < n_reached; i++) {
                        if (renum[i] != kNone || !is_single(i)) continue;                 // (a single not yet on a path -- an IgnoreCase DAG reaches a state by several parents: breadth-first order)
                        for (uint32_t y = i; is_single(y) && renum[y] == kNone;) {
                            renum[y] = nxt++;
                            if (rec[y].n != 1) break;
                            y = rec[y].to[0];
                        }
                    }
                    const uint32_t n_single = nxt - n_rows;
                    {
                        std::vector<uint32_t> doubles;
                        for (uint32_t i = 1; i < n_reached; i++) if (!is_row[i] && rec[i].n == 2) doubles.push_back(i);
                        std::sort(doubles.begin(), doubles.end(), heavier);
                        for (uint32_t i : doubles) renum[i] = nxt++;
                    }
                    if (n_rows >= (1u << 24)) { /* a record holds the row state it leans on in 24 bits: no DFA section for this automaton */ }
                    else {
                    dtr.mark("dfa: numbering");
                    std::vector<uint32_t> next2((size_t)n_rows << lc);
                    std::vector<u32x2> chain(n_single + 1u, u32x2{0, 0}), out2(n_reached);       // (+ 1: the device reads 16 bytes at a record)
                    std::vector<u32x4> chain2(n_reached - n_rows - n_single);
                    std::vector<uint32_t> fb2(n_reached);
                    for (uint32_t i = 0; i < n_reached; i++) {
                        out2[renum[i]] = out[i];
                        fb2[renum[i]] = renum[fb[i]];
                    }
                    for (uint32_t i = 0; i < n_reached; i++) {
                        if (is_row[i]) {
                            const uint32_t* from = next.data() + ((size_t)i << lc);
                            uint32_t* to = next2.data() + ((size_t)renum[i] << lc);
                            for (uint32_t c = 0; c < C; c++) { const uint32_t t = renum[from[c]]; to[col_new[c]] = t | dfa_end_bits(out2[t]); }
                        } else if (rec[i].n <= 1) {
                            const uint32_t ch = rec[i].n == 1 ? renum[rec[i].to[0]] : 0u;
                            chain[renum[i] - n_rows] = u32x2{rec[i].n == 1 ? (ch | dfa_end_bits(out2[ch])) : 0u,
                                                             ((rec[i].n == 1 ? col_new[rec[i].cls[0]] : kDfaNoChild) << 24) | renum[near_row[i]]};
                        } else {
                            // two entries: {target a, class a << 24 | R, target b, class b << 24} -- a single-child record and a second entry
                            const uint32_t ta = renum[rec[i].to[0]], tb = renum[rec[i].to[1]];
                            chain2[renum[i] - n_rows - n_single] = u32x4{ta | dfa_end_bits(out2[ta]), (col_new[rec[i].cls[0]] << 24) | renum[near_row[i]], tb | dfa_end_bits(out2[tb]), col_new[rec[i].cls[1]] << 24};
                        }
                    }
                    dtr.mark("dfa: tables out");
                    // the rare edges: open addressing, (state, byte) -> child | its end bits (dfa_rare_slot in am_image.h)
                    uint32_t rare_lc = 4;
                    while ((1ull << rare_lc) < 2ull * rare_goto.size() + 8ull) rare_lc++;
                    std::vector<u32x4> rare_tab((size_t)1 << rare_lc, u32x4{0, 0, 0, 0});
                    for (const auto& kv : rare_goto) {
                        const uint32_t st = renum[(uint32_t)(kv.first >> 8)], byte = (uint32_t)(kv.first & 0xFFu), to = renum[kv.second];
                        uint32_t slot = dfa_rare_slot(st, byte, rare_lc);
                        while (rare_tab[slot].w) slot = (slot + 1u) & ((1u << rare_lc) - 1u);
                        rare_tab[slot] = u32x4{st, byte, to | dfa_end_bits(out2[to]), 1u};
                    }
                    uint32_t warm = 1;
                    for (uint32_t x = 0; x < (uint32_t)S; x++) warm = std::max(warm, bdepth[x]);
                    long chunk = cfg::get(cfg::kDfaChunk);
                    if (chunk < 64 || chunk > (1 << 20)) chunk = 2048;          // (512: 10 % of the steps are warm-up; measured 131 / 136 / 139 / 139 GiB/s counting at 512 / 1024 / 2048 / 4096)
                    chunk = (chunk + 15) & ~15L;
                    while ((uint64_t)chunk < 4ull * warm && chunk < (1 << 20)) chunk *= 2;          // the warm-up stays a fraction of the lane's own bytes
                    // The HOT table (image version 16): columns 1 .. 2^hot_log2 of every row once more, dense -- hot[(row << hot_log2) + class - 1].  A row of the full table is
                    // 256 bytes of which text touches the first half; here two rows (hot_log2 = 4) share a 128-byte line, and rows of about the same weight are neighbours,
                    // so the L2 of an XCD holds twice the rows per MiB for the classes that are 85 % of natural text.  (Column 0 -- bytes no needle contains -- leads to the
                    // root from everywhere and is in neither LDS nor the hot table: the walk answers it without a load.)
                    long hot_cfg = cfg::get(cfg::kDfaHotLog2);
                    uint32_t hot_lc = hot_cfg >= 1 && hot_cfg <= 8 ? (uint32_t)hot_cfg : 4u;
                    while (hot_lc > 0 && (1u << hot_lc) > C - 1u) hot_lc--;
                    std::vector<uint32_t> hot2((size_t)n_rows << hot_lc);
                    for (uint32_t r = 0; r < n_rows; r++)
                        for (uint32_t c = 0; c < (1u << hot_lc); c++) hot2[((size_t)r << hot_lc) + c] = next2[((size_t)r << lc) + c + 1u];
                    o.next2.swap(next2); o.hot2.swap(hot2); o.chain.swap(chain); o.chain2.swap(chain2); o.n_single = n_single; o.out2.swap(out2); o.cls.swap(cls); o.fb2.swap(fb2); o.rare_tab.swap(rare_tab);
                    o.hot_lc = hot_lc; o.rare_lc = rare_lc; o.n_rows = n_rows; o.n_states = n_reached; o.lc = lc; o.warm = warm - 1u > 0 ? warm - 1u : 1u; o.chunk = (uint32_t)chunk;
                    o.made = true;
                    }
                }
            }
        }
    };
    DfaOut dfa_early;
    std::future<void> dfa_task;
    JoinOnExit dfa_join{dfa_task};                          // (an early return must not leave the task with dangling references)
    {
        // will the automaton get a DFA section?  AM_DFA decides, or -- unset -- whether the suffix tables below give heavy depth-4 nodes their children (sf_t4_children > 0):
        // the same test the tables make, made here so that the section's work starts now and not behind them
        const long dfa_cfg = cfg::get(cfg::kDfa);
        bool likely = dfa_cfg != cfg::kUnset ? dfa_cfg != 0 : S <= kDfaSmallStates;
        if (dfa_cfg == cfg::kUnset && !cfg::on(cfg::kSfNoChildren)) {
            size_t keys = 0, potential = 0;
            for (int t = 0; t < 4; t++) keys += tier_entries[t].size();
            uint32_t max_lw = 15;
            { const long v = cfg::get(cfg::kSfMaxBloomLog2Words); if (v >= 8 && v <= 15) max_lw = (uint32_t)v; }
            if (std::max(8u, std::min(max_lw, log2_ceil((keys * 16 + 31) / 32))) < 15) {
                for (const TierEntry& e : tier_entries[3]) { const SfNode& nd = nodes[e.node]; const uint32_t c = nd.w & 0xFFFFu; if (!nd.x && c >= 2) potential += c; }
                likely = potential >= tier_entries[3].size() && potential > 0;
            }
        }
        if (likely && !cfg::on(cfg::kFlattenSerial)) {
            try { dfa_task = std::async(std::launch::async, [&] { make_dfa(dfa_early); }); } catch (const std::system_error&) { /* no thread: made in place below */ }
        }
    }
    h.sf_n_nodes = (uint32_t)nodes.size();
    h.n_edges = edges_out.size();
    h.n_edge_maps = 0; h.sf_row_first = row_first;
    h.sf_tiers = 0;
    size_t total_keys = 0;
    for (int t = 0; t < 4; t++) { if (!tier_entries[t].empty()) h.sf_tiers |= 1u << t; total_keys += tier_entries[t].size(); }
    {
        uint32_t lw = log2_ceil((total_keys * 16 + 31) / 32);
        uint32_t max_lw = 15;                                  // 128 KiB of the CU's 160 KiB LDS
        { const long v = cfg::get(cfg::kSfMaxBloomLog2Words); if (v >= 8 && v <= 15) max_lw = (uint32_t)v; }
        lw = std::max(8u, std::min(max_lw, lw));
        h.sf_bloom_log2_words = lw;
        std::vector<uint32_t> bloom((size_t)1 << lw, 0);
        for (int t = 0; t < 4; t++)
            for (const TierEntry& e : tier_entries[t]) { const uint32_t hh = bloom_hash(e.key, (uint32_t)t + 1); bloom[bloom_word(hh, lw)] |= bloom_mask(hh); }
        h.off_bloom = blob.put(bloom);
    }
    for (int t = 0; t < 3; t++) {       // 1..3-byte needles: plain open addressing (rare)
        const uint32_t lc = std::max(4u, log2_ceil(tier_entries[t].size() * 2 + 1));
        if (lc > 28) { err = "suffix table too large"; return -1; }
        h.tier_log2_cap[t] = lc;
        const uint32_t cap_mask = (1u << lc) - 1;
        std::vector<u32x2> tab((size_t)1 << lc, u32x2{0, kNone});
        for (const TierEntry& e : tier_entries[t]) {
            uint32_t i = tier_slot(e.key, lc);
            while (tab[i].y != kNone) {
                if (tab[i].x == e.key) { err = "duplicate suffix key (internal error)"; return -1; }
                i = (i + 1) & cap_mask;
            }
            tab[i] = u32x2{e.key, e.node};
        }
        h.off_tier[t] = blob.put(tab);
    }
    {
        // 4-byte suffixes: (2,2) cuckoo hashing -- 2 candidate buckets of 2 slots per key, ~75 % full.
        // HOT side: 4 bytes per slot = 8 B per bucket; for 100k needles that is 1 MiB, resident in
        // every XCD's L2 next to the streamed haystack.  COLD side: full keys + node ids, 16 B per bucket.
        // Entries: one per key first (`edge` = 0: the node as a whole).  A depth-4 node that branches (2-4 children, no
        // needle end) is upgraded afterwards to one entry PER CHILD (`edge` = i + 1), each fixing that child's selector
        // byte, wherever its four candidate slots have room: under IgnoreCase every needle letter with several encodings
        // before the suffix (k/K/KELVIN SIGN, i/I/İ, any non-ASCII letter) makes such a node, and left as "always look
        // closer" entries they are half of the candidates that reach phase 2.  The copies share key and node; the probe
        // ORs over the four candidate slots anyway.
        // HEAVY nodes (round 5): a branching depth-4 node without a needle end that could not be split in place -- more than four children, or no room
        // among its four candidate slots -- gets kT4Heavy in its hot word and one entry per child under the FIVE-byte key t4_key5(key, child byte),
        // placed like any key (am_image.h, kT4Heavy).  Only automata with a small LDS filter (few distinct 4-byte suffixes) whose suffixes branch as a rule
        // take them: that is the signature of a dictionary whose words share their endings (natural language: 100k words, 12k suffixes), and it is the
        // k_sf instantiations with LW = 0 that look for them; the benchmark automata (random needles) have next to no such nodes and keep their probe as it was.
        struct HotEntry { uint32_t key, node, edge; bool remote; };      // edge: 0 = the node as a whole, i + 1 = its child i; remote: placed under the five-byte key
        auto place_key = [&](const HotEntry& e) -> uint32_t {
            if (!e.remote) return e.key;
            return t4_key5(e.key, edges_out[nodes[e.node].z + e.edge - 1].byte & 0xFFu);
        };
        bool want_children = h.sf_bloom_log2_words < 15 && !cfg::on(cfg::kSfNoChildren);
        size_t potential_children = 0;
        if (want_children) {
            for (const TierEntry& e : tier_entries[3]) {
                const SfNode& nd = nodes[e.node]; const uint32_t c = nd.w & 0xFFFFu;
                if (!nd.x && c >= 2) potential_children += c;
            }
            // worth it when branching is the rule (a dictionary: several children per suffix), not the odd IgnoreCase variant of a random needle set
            if (potential_children < tier_entries[3].size()) { want_children = false; potential_children = 0; }
        }
        std::vector<HotEntry> ents;
        std::vector<uint32_t> owner;          // slot -> entry index
        std::vector<uint8_t> heavy;           // per base entry: its children have entries of their own
        size_t n_base = 0;
        uint32_t lb = 2;
        while (((uint64_t)2 << lb) * 75 < (uint64_t)(tier_entries[3].size() + potential_children) * 100) lb++;
        uint32_t n_children = 0;
        for (;; lb++) {
            if (lb > 27) { err = "suffix table too large"; return -1; }
            ents.clear();
            ents.reserve(tier_entries[3].size() + tier_entries[3].size() / 4 + potential_children);
            for (const TierEntry& e : tier_entries[3]) ents.push_back(HotEntry{e.key, e.node, 0, false});
            n_base = ents.size();
            heavy.assign(n_base, 0);
            n_children = 0;
            owner.assign((size_t)2 << lb, kNone);
            uint32_t rng = 0x12345u;
            // one entry into the table, evicting along the way (cuckoo); false: no place found, the table grows
            auto insert = [&](uint32_t first) -> int {
                uint32_t cur = first;
                uint32_t bucket = t4_bucket(t4_hash_a(place_key(ents[cur])), lb);
                for (int kicks = 0;; kicks++) {
                    const uint32_t pk = place_key(ents[cur]);
                    const uint32_t ba = t4_bucket(t4_hash_a(pk), lb), bb = t4_bucket(t4_hash_b(pk), lb);
                    for (uint32_t bsel : {ba, bb}) {
                        for (uint32_t j = 0; j < 2; j++) {
                            uint32_t& o = owner[2u * bsel + j];
                            if (o != kNone && ents[o].key == ents[cur].key && ents[o].edge == ents[cur].edge && ents[o].remote == ents[cur].remote) return -1;
                            if (o == kNone) { o = cur; return 1; }
                        }
                    }
                    if (kicks > 2000) return 0;
                    // evict a pseudo-random resident of the bucket we did not come from
                    bucket = (bucket == ba) ? bb : ba;
                    rng = rng * 1664525u + 1013904223u;
                    std::swap(cur, owner[2u * bucket + (rng >> 31)]);
                }
            };
            bool ok = true;
            for (uint32_t k = 0; k < n_base && ok; k++) {
                const int r = insert(k);
                if (r < 0) { err = "duplicate suffix key (internal error)"; return -1; }
                ok = r > 0;
            }
            if (!ok) continue;
            // upgrade branching nodes IN PLACE: free slots among the key's four candidates, plus slots freed by moving a neighbour
            // to a free slot of ITS other bucket (one step, no chains).  Not for a dictionary (want_children): in-place copies and their siblings fill the
            // two buckets of their key and cannot move, and a table a third of whose buckets are such blocks has no room for the walk that places the five-byte
            // entries (measured: 26k entries needed 2^18 slots); there every branching node takes the five-byte route.
            if (!want_children) {
                std::vector<uint32_t> where(n_base, kNone);
                for (size_t sl = 0; sl < owner.size(); sl++) if (owner[sl] != kNone) where[owner[sl]] = (uint32_t)sl;
                for (uint32_t k = 0; k < n_base; k++) {
                    const SfNode& nd = nodes[ents[k].node];
                    const uint32_t c = nd.w & 0xFFFFu;
                    if (nd.x || c < 2 || c > 4) continue;
                    const uint32_t ba = t4_bucket(t4_hash_a(ents[k].key), lb), bb = t4_bucket(t4_hash_b(ents[k].key), lb);
                    uint32_t cand[4] = {2u * ba, 2u * ba + 1u, 2u * bb, 2u * bb + 1u};
                    const uint32_t n_cand = ba == bb ? 2u : 4u;
                    std::vector<uint32_t> room;
                    for (uint32_t i = 0; i < n_cand; i++) if (owner[cand[i]] == kNone) room.push_back(cand[i]);
                    for (uint32_t i = 0; i < n_cand && room.size() + 1 < c; i++) {
                        const uint32_t o = owner[cand[i]];
                        if (o == kNone || o == k || ents[o].key == ents[k].key) continue;
                        const uint32_t oa = t4_bucket(t4_hash_a(ents[o].key), lb), ob = t4_bucket(t4_hash_b(ents[o].key), lb);
                        const uint32_t other = (cand[i] >> 1) == oa ? ob : oa;
                        if (other == ba || other == bb) continue;
                        for (uint32_t j = 0; j < 2; j++) {
                            if (owner[2u * other + j] != kNone) continue;
                            owner[2u * other + j] = o;
                            if (o < n_base) where[o] = 2u * other + j;
                            owner[cand[i]] = kNone;
                            room.push_back(cand[i]);
                            break;
                        }
                    }
                    if (room.size() + 1 < c) continue;                         // no room: the node keeps its single entry
                    ents[k].edge = 1;
                    for (uint32_t i = 1; i < c; i++) { owner[room[i - 1]] = (uint32_t)ents.size(); ents.push_back(HotEntry{ents[k].key, ents[k].node, i + 1, false}); }
                }
            }
            // ... and the rest of them -- still one entry for a branching node without a needle end -- through five-byte entries anywhere in the table
            if (want_children) {
                for (uint32_t k = 0; k < n_base && ok; k++) {
                    const SfNode& nd = nodes[ents[k].node];
                    const uint32_t c = nd.w & 0xFFFFu;
                    if (nd.x || c < 2 || ents[k].edge != 0) continue;
                    heavy[k] = 1;
                    for (uint32_t i = 0; i < c && ok; i++) {
                        ents.push_back(HotEntry{ents[k].key, ents[k].node, i + 1, true});
                        const int r = insert((uint32_t)ents.size() - 1u);
                        if (r < 0) { err = "duplicate child key (internal error)"; return -1; }
                        ok = r > 0;
                        n_children++;
                    }
                }
            }
            if (ok) break;
        }
        h.sf_t4_children = n_children;
        h.tier_log2_cap[3] = lb;
        std::vector<u32x2> hot((size_t)1 << lb, u32x2{0, 0});
        std::vector<SfSlot> slots((size_t)2 << lb, SfSlot{0, 0, 0, 0, 0, 0, 0, 0, {0, 0, 0, 0}, 0, 0, 0, 0});
        for (size_t sl = 0; sl < owner.size(); sl++) {
            if (owner[sl] == kNone) continue;
            const HotEntry& e = ents[owner[sl]];
            const SfNode& nd = nodes[e.node];
            const uint32_t n_edges = nd.w & 0xFFFFu;
            // bytes before the 4-byte suffix that every needle through this node fixes (0 when a needle ends here or the
            // node branches): sel1 = the selector of the single edge; sel2 = the first skip byte of that edge, or (no
            // skip) the selector of the child's single edge provided no needle ends at the child
            uint32_t fixed = 0, sel1 = 0, sel2 = 0;
            if (!nd.x && n_edges == 1) {
                fixed = 1; sel1 = (nd.w >> 16) & 0xFFu;
                const uint32_t skip = nd.w >> 24;
                if (skip >= 1) { fixed = 2; sel2 = nd.label[3] >> 24; }       // walk order byte 0 = last label byte in text order
                else {
                    const SfNode& ch = nodes[nd.z];
                    if (!ch.x && (ch.w & 0xFFFFu) == 1u) { fixed = 2; sel2 = (ch.w >> 16) & 0xFFu; }
                }
            }
            if (e.edge && !e.remote) {                                         // one child of a branching node, in one of the key's own slots
                const SfEdge& ed = edges_out[nd.z + e.edge - 1];
                fixed = 1; sel1 = ed.byte & 0xFFu;
                if (ed.skip >= 1) { fixed = 2; sel2 = ed.label[3] >> 24; }
                else {
                    const SfNode& ch = nodes[ed.child];
                    if (!ch.x && (ch.w & 0xFFFFu) == 1u) { fixed = 2; sel2 = (ch.w >> 16) & 0xFFu; }
                }
            }
            if (e.remote) {
                // one child of a HEAVY node under the five-byte key: the child byte is part of the key, the selectors are the bytes BEHIND it (bytes six
                // and seven of the context, in walk order) as far as the trie fixes them without a needle ending on the way
                const SfEdge& ed = edges_out[nd.z + e.edge - 1];
                fixed = 0;
                if (ed.skip >= 2) { fixed = 2; sel1 = ed.label[3] >> 24; sel2 = (ed.label[3] >> 16) & 0xFFu; }
                else {
                    const SfNode& ch = nodes[ed.child];
                    const bool chain = !ch.x && (ch.w & 0xFFFFu) == 1u;       // behind the edge: no needle end, one way on
                    if (ed.skip == 1) {
                        fixed = 1; sel1 = ed.label[3] >> 24;
                        if (chain) { fixed = 2; sel2 = (ch.w >> 16) & 0xFFu; }
                    } else if (chain) {
                        fixed = 1; sel1 = (ch.w >> 16) & 0xFFu;
                        if ((ch.w >> 24) >= 1u) { fixed = 2; sel2 = ch.label[3] >> 24; }
                    }
                }
            }
            uint32_t word = t4_slot_word(t4_fingerprint(t4_hash_a(place_key(e)), lb), fixed, sel1, sel2);
            if (!e.edge && owner[sl] < n_base && heavy[owner[sl]]) word |= kT4Heavy;      // (fixed = 0 here: the bits above 15 are free)
            (&hot[sl >> 1].x)[sl & 1] = word;
            // the slot's line for phase 2: the depth-4 node, its single edge (or this copy's child edge) and that edge's child
            SfSlot& so = slots[sl];
            so.key = e.key; so.flags = kSlotOccupied | (e.edge ? kSlotChildCopy : 0u);      // (a remote child's line is never asked for -- phase 2 reads the node's own slot -- but says the right thing)
            so.x = nd.x; so.y = nd.y;
            const SfNode* ch = nullptr;
            if (e.edge) {
                const SfEdge& ed = edges_out[nd.z + e.edge - 1];
                so.w = 1u | ((ed.byte & 0xFFu) << 16) | (ed.skip << 24);
                so.z = ed.child;
                for (int i = 0; i < 4; i++) so.label[i] = ed.label[i];
                ch = &nodes[ed.child];
            } else if (n_edges == 1) {
                so.w = nd.w; so.z = nd.z;
                for (int i = 0; i < 4; i++) so.label[i] = nd.label[i];
                ch = &nodes[nd.z];
            } else if (n_edges == 0) {
                so.w = 0; so.z = 0;
            } else {
                so.w = n_edges; so.z = e.node; so.ez = nd.z; so.el0 = nd.label[0]; so.label[1] = nd.label[1]; so.label[2] = nd.label[2]; so.label[3] = nd.label[3];      // branching and not split: phase 2 starts its walk from this line
            }
            if (ch) { so.cx = ch->x; so.cy = ch->y; so.cw = ch->w; }
        }
        h.off_tier[3] = blob.put(hot);
        h.off_t4_slots = blob.put(slots);
    }
    h.off_nodes = blob.put(nodes);
    h.off_edges = blob.put(edges_out);
    h.off_edge_maps = 0;
    tr.mark("filter + suffix tables");

    // ---- DFA section.  A dictionary of natural-language words meets text in which a needle ends every few bytes: there the suffix filter filters nothing
    // (four positions in ten pass it) and k_sf is bound by the divergent loads and the instructions of its resolve phase (LABNOTES R5.7).  For such an automaton
    // the image also carries the classic alternative: the byte-level Aho-Corasick automaton with EVERY transition resolved (delta(state, byte) = one table entry,
    // the fallback chain folded in), so that a lane walks its stretch of the haystack with one dependent load per byte and nothing else (k_dfa).
    //   states   = the reference's states + the nodes inside multi-byte code points, numbered breadth-first (the states natural text dwells in come first);
    //              a code point edge becomes the UTF-8 of every haystack code point that lowers to it (variants_of, as in the suffix trie), so an IgnoreCase
    //              automaton is a DAG: both spellings of a letter lead to the same state.  Its fallback is that of the reference (a property of the folded string);
    //   classes  = the bytes that occur on some edge (IgnoreCase: upper-case ASCII shares the class of its fold); every other byte leads to the root from anywhere;
    //   needle ends: only at the reference's states (a needle ends with a whole code point): vlen[s] > 0, reported as canon[s] like everywhere else.
    // the DFA section: made by the task started below the suffix trie (or here, when it was not started), kept only if the automaton turned out to be a dictionary
    {
        const long dfa_cfg = cfg::get(cfg::kDfa);
        // unset: dictionaries (the suffix tables gave heavy depth-4 nodes their children), and every SMALL automaton -- its table costs nothing (32k states x 64 classes = 8 MiB
        // at most) and whether a batch takes it is the sample walk's decision (am_abi.cpp make_plan): three needles that end at every position of the text (needles a, aa, aaa
        // over a...a: 43 GiB/s on the filter, output-bound) are the table walk's case as much as a dictionary over its language
        const bool want = dfa_cfg == cfg::kUnset ? (h.sf_t4_children > 0 || S <= kDfaSmallStates) : dfa_cfg != 0;
        DfaOut o;
        if (dfa_task.valid()) { dfa_task.get(); if (want) o = std::move(dfa_early); }
        else if (want) make_dfa(o);
        if (o.made) {
            h.off_dfa_next = blob.put(o.next2);
            h.off_dfa_hot = blob.put(o.hot2);
            h.dfa_hot_log2 = o.hot_lc;
            h.off_dfa_chain = blob.put(o.chain);
            h.off_dfa_chain2 = blob.put(o.chain2);
            h.dfa_n_single = o.n_single;
            h.off_dfa_out = blob.put(o.out2);
            h.off_dfa_cls = blob.put(o.cls);
            h.off_dfa_fail = blob.put(o.fb2);
            h.off_dfa_rare = blob.put(o.rare_tab);
            h.dfa_rare_log2_cap = o.rare_lc;
            h.dfa_n_rows = o.n_rows;
            h.dfa_n_states = o.n_states; h.dfa_log2_classes = o.lc; h.dfa_warm = o.warm; h.dfa_chunk = o.chunk;
        }
    }
    tr.mark("DFA section");
    if (goto_task.valid()) goto_task.get();
    std::memcpy(blob.bytes.data() + h.off_goto, goto_tab.data(), goto_tab.size() * sizeof(u32x4));
    std::memcpy(blob.bytes.data() + h.off_fail, goto_fail.data(), goto_fail.size() * sizeof(uint32_t));
    blob.reserve_section(16);      // tail padding
    h.total_bytes = blob.bytes.size();
    h.checksum = image_checksum(blob.bytes.data() + sizeof(h), blob.bytes.size() - sizeof(h));
    std::memcpy(blob.bytes.data(), &h, sizeof(h));
    image.swap(blob.bytes);
    tr.mark("checksum");
    return 0;
}

uint64_t image_checksum(const uint8_t* p, size_t n)
{
    // FNV-1a over 8-byte words (the tail bytes one by one): integrity of a stored image, not security
    uint64_t x = 0xcbf29ce484222325ull;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; std::memcpy(&w, p + i, 8); x = (x ^ w) * 0x100000001b3ull; }
    for (; i < n; i++) x = (x ^ p[i]) * 0x100000001b3ull;
    return x;
}

// One host pass over the body of an image that comes from outside (a file): every index the kernels follow must stay inside its
// table, so that a stale or damaged image is refused instead of making k_ac / k_sf read out of bounds.
bool image_body_valid(const uint8_t* img, const ImageHeader& h, std::string& err)
{
    const uint32_t S = h.n_states;
    const uint32_t* offsets = (const uint32_t*)(img + h.off_offsets);
    const uint64_t* tr = (const uint64_t*)(img + h.off_transitions);
    const uint32_t* canon = (const uint32_t*)(img + h.off_canon);
    const uint32_t* fail = (const uint32_t*)(img + h.off_fail);
    for (uint32_t s = 0; s + 1 <= S; s++) {
        if (offsets[s] > offsets[s + 1] || offsets[s + 1] > h.n_transitions) { err = "image: offsets out of range"; return false; }
        if (canon[s] >= S || fail[s] >= S) { err = "image: canon/fail state out of range"; return false; }
    }
    for (uint64_t i = 0; i < h.n_transitions; i++) if ((uint32_t)(tr[i] >> 32) >= S) { err = "image: transition target out of range"; return false; }
    if (h.case_mode == 1) {
        // the lower-case table the general kernel reads must be the one the header names (ImageHeader::flags = LowerTable::hash)
        const int32_t* delta = (const int32_t*)(img + h.off_lower);
        uint64_t hh = 0xCBF29CE484222325ull;
        for (uint32_t cp = 0; cp < h.n_lower; cp++) {
            if (!delta[cp]) continue;
            const int64_t to = (int64_t)cp + delta[cp];
            if (to < 0 || to > 0x10FFFF) { err = "image: lower-case table leaves the code point range"; return false; }
            for (uint32_t x : {cp, (uint32_t)to}) for (int b = 0; b < 4; b++) { hh ^= (x >> (8 * b)) & 0xFFu; hh *= 0x100000001B3ull; }
        }
        uint32_t h32 = (uint32_t)(hh ^ (hh >> 32));
        if (h32 == 0) h32 = 1;
        if (h32 != h.flags) { err = "image: lower-case table does not match the header"; return false; }
    }
    {
        // the 128 root entries k_ac and ends_first_code_point follow (am_image.h ac_step): a goto target or the wildcard
        const uint64_t* root = (const uint64_t*)(img + h.off_root_ascii);
        for (uint32_t c = 0; c < 128; c++) if (!(root[c] & kWildcard) && (uint32_t)(root[c] >> 32) >= S) { err = "image: root table target out of range"; return false; }
        // every fallback chain must reach the root, or ac_step never returns: colour the states along each chain (0 = unseen, 1 = on
        // the chain being followed, 2 = known to end at the root)
        if (S && fail[0] != 0) { err = "image: the root's fallback is not the root"; return false; }
        std::vector<uint8_t> colour(S, 0);
        std::vector<uint32_t> chain;
        if (S) colour[0] = 2;
        for (uint32_t s0 = 1; s0 < S; s0++) {
            chain.clear();
            uint32_t s = s0;
            while (colour[s] == 0) { colour[s] = 1; chain.push_back(s); s = fail[s]; }
            if (colour[s] == 1) { err = "image: fallback chain does not end at the root"; return false; }
            for (uint32_t c : chain) colour[c] = 2;
        }
    }
    const u32x4* go = (const u32x4*)(img + h.off_goto);
    bool goto_has_empty = false;
    for (uint64_t i = 0; i < (1ull << h.ac_goto_log2_cap); i++) {
        if (!go[i].w) goto_has_empty = true;
        else if (go[i].x >= S || go[i].z >= S) { err = "image: goto table entry out of range"; return false; }
    }
    if (!goto_has_empty) { err = "image: goto table without an empty slot"; return false; }      // the open-addressing probe must terminate
    if (!h.sf_enabled) return true;
    const SfNode* nodes = (const SfNode*)(img + h.off_nodes);
    const SfEdge* edges = (const SfEdge*)(img + h.off_edges);
    for (uint32_t i = 0; i < h.sf_n_nodes; i++) {
        const SfNode& n = nodes[i];
        const uint32_t ne = n.w & 0xFFFFu;
        if (n.x > S) { err = "image: node state out of range"; return false; }
        if (ne == 1 && (n.z >= h.sf_n_nodes || (n.w >> 24) > kMaxSkip)) { err = "image: node child out of range"; return false; }
        if (ne > 1 && ((uint64_t)n.z + ne > h.n_edges)) { err = "image: node edge range out of range"; return false; }
    }
    if (h.sf_row_first > h.n_edges) { err = "image: row region out of range"; return false; }
    for (uint64_t i = 0; i < h.n_edges; i++) {
        if (i >= h.sf_row_first && edges[i].pad == kNone) continue;                                        // an empty line of the row region
        if ((i < h.sf_row_first) != (edges[i].pad == 0)) { err = "image: edge line with a wrong owner mark"; return false; }
        if (edges[i].child >= h.sf_n_nodes || edges[i].skip > kMaxSkip || edges[i].byte > 0xFFu) { err = "image: edge out of range"; return false; }
        if (std::memcmp(&edges[i].to, &nodes[edges[i].child], sizeof(SfNode)) != 0) { err = "image: an edge's copy of its child differs from the child"; return false; }
    }
    for (uint32_t i = 0; i < h.sf_n_nodes; i++) {
        const uint32_t ne = nodes[i].w & 0xFFFFu;
        if (ne <= 4) continue;
        // its row: inside the array for every selector byte, and each of its edges is there under its byte, marked as this node's
        if (nodes[i].label[0] < h.sf_row_first || (uint64_t)nodes[i].label[0] + 256u > h.n_edges) { err = "image: a node's row is out of range"; return false; }
        for (uint32_t e = 0; e < ne; e++) {
            const SfEdge& own = edges[nodes[i].z + e];
            const SfEdge& line = edges[(uint64_t)nodes[i].label[0] + own.byte];
            const uint32_t f = own.byte % 96u;
            if (line.pad != nodes[i].z + 1u || line.byte != own.byte || line.child != own.child || line.skip != own.skip || std::memcmp(line.label, own.label, 16) != 0 ||
                !((nodes[i].label[1 + f / 32u] >> (f & 31u)) & 1u)) { err = "image: a node's row does not hold its edges"; return false; }
        }
    }
    for (int t = 0; t < 3; t++) {
        if (!(h.sf_tiers & (1u << t))) continue;
        const u32x2* tab = (const u32x2*)(img + h.off_tier[t]);
        bool has_empty = false;
        for (uint64_t i = 0; i < (1ull << h.tier_log2_cap[t]); i++) { if (tab[i].y == kNone) has_empty = true; else if (tab[i].y >= h.sf_n_nodes) { err = "image: suffix table node out of range"; return false; } }
        if (!has_empty) { err = "image: suffix table without an empty slot"; return false; }      // the linear probe must terminate
    }
    if (h.sf_tiers & 8u) {
        const SfSlot* slots = (const SfSlot*)(img + h.off_t4_slots);
        for (uint64_t i = 0; i < (2ull << h.tier_log2_cap[3]); i++) {
            if (!(slots[i].flags & kSlotOccupied)) continue;
            const uint32_t kind = slots[i].w & 0xFFFFu;
            if ((kind != 0 && slots[i].z >= h.sf_n_nodes) || (kind == 1 && (slots[i].w >> 24) > kMaxSkip)) { err = "image: suffix slot out of range"; return false; }
        }
    }
    if (h.dfa_n_states) {
        const uint32_t* next = (const uint32_t*)(img + h.off_dfa_next);
        const u32x2* out = (const u32x2*)(img + h.off_dfa_out);
        const uint8_t* cls = img + h.off_dfa_cls;
        const uint32_t* vl = (const uint32_t*)(img + h.off_vlen);
        for (uint32_t b = 0; b < 256; b++) if (cls[b] >= (1u << h.dfa_log2_classes) && cls[b] != kDfaRare) { err = "image: DFA byte class out of range"; return false; }
        {
            // the rare-byte walk: fallbacks lead towards the root (a smaller breadth-first depth is not recorded in the image: a walk of n_states steps that has not
            // reached the root is a cycle), the hash has an empty slot, its entries stay inside the table
            const uint32_t* fl = (const uint32_t*)(img + h.off_dfa_fail);
            const u32x4* rt = (const u32x4*)(img + h.off_dfa_rare);
            if (fl[0] != 0) { err = "image: DFA root fallback"; return false; }
            for (uint32_t i = 0; i < h.dfa_n_states; i++) if (fl[i] >= h.dfa_n_states) { err = "image: DFA fallback out of range"; return false; }
            std::vector<uint8_t> ok_root(h.dfa_n_states, 0);       // 1: the chain from here reaches the root
            ok_root[0] = 1;
            std::vector<uint32_t> path;
            for (uint32_t i = 0; i < h.dfa_n_states; i++) {
                path.clear();
                uint32_t x = i;
                while (!ok_root[x]) { path.push_back(x); if (path.size() > h.dfa_n_states) { err = "image: DFA fallbacks form a cycle"; return false; } x = fl[x]; }
                for (uint32_t y : path) ok_root[y] = 1;
            }
            bool has_empty = false;
            for (uint64_t i = 0; i < (1ull << h.dfa_rare_log2_cap); i++) {
                if (!rt[i].w) { has_empty = true; continue; }
                const uint32_t to = rt[i].z & kDfaStateMask;
                if (rt[i].x >= h.dfa_n_states || rt[i].y > 0xFFu || to >= h.dfa_n_states || (rt[i].z & ~kDfaStateMask) != dfa_end_bits(out[to])) { err = "image: DFA rare edge out of range"; return false; }
            }
            if (!has_empty) { err = "image: DFA rare-edge table without an empty slot"; return false; }
        }
        for (uint32_t i = 0; i < h.dfa_n_states; i++)
            if (out[i].x > S || (out[i].x != 0 && (canon[out[i].x - 1u] != out[i].x - 1u || vl[out[i].x - 1u] == 0 || out[i].y == 0))) { err = "image: DFA needle end out of range"; return false; }
        const uint64_t n = (uint64_t)h.dfa_n_rows << h.dfa_log2_classes;
        for (uint64_t i = 0; i < n; i++) {
            const uint32_t to = next[i] & kDfaStateMask;
            if (to >= h.dfa_n_states || (next[i] & ~kDfaStateMask) != dfa_end_bits(out[to])) { err = "image: DFA transition out of range"; return false; }
        }
        {
            // column 0 (bytes no needle contains) leads to the root from every row (the walk answers it without looking); the hot table repeats columns 1 .. 2^hot_log2
            const uint32_t* hot = (const uint32_t*)(img + h.off_dfa_hot);
            for (uint32_t r = 0; r < h.dfa_n_rows; r++) {
                if (next[(uint64_t)r << h.dfa_log2_classes] != 0u) { err = "image: DFA column 0 does not lead to the root"; return false; }
                for (uint32_t c = 0; c < (1u << h.dfa_hot_log2); c++)
                    if (hot[((uint64_t)r << h.dfa_hot_log2) + c] != next[((uint64_t)r << h.dfa_log2_classes) + c + 1u]) { err = "image: DFA hot table differs from the rows"; return false; }
            }
        }
        // records: the entries' classes and end bits, and the state a record leans on: a ROW state on the record state's chain of fallbacks (fail[] was checked above to
        // lead to the root without a cycle)
        const uint32_t* fl = (const uint32_t*)(img + h.off_dfa_fail);
        auto leans_on = [&](uint32_t state, uint32_t r) {
            if (r >= h.dfa_n_rows) return false;
            for (uint32_t s = state; s != 0u;) { s = fl[s]; if (s == r) return true; }
            return false;
        };
        const u32x2* chain = (const u32x2*)(img + h.off_dfa_chain);
        for (uint32_t i = 0; i < h.dfa_n_single; i++) {
            const uint32_t to = chain[i].x & kDfaStateMask, cl = chain[i].y >> 24, fbs = chain[i].y & 0xFFFFFFu;
            if (!leans_on(h.dfa_n_rows + i, fbs) || (cl >= (1u << h.dfa_log2_classes) && cl != kDfaNoChild) || to >= h.dfa_n_states || (chain[i].x & ~kDfaStateMask) != (cl != kDfaNoChild ? dfa_end_bits(out[to]) : 0u) ||
                (cl == kDfaNoChild && chain[i].x != 0) || cl == 0u) { err = "image: DFA chain record out of range"; return false; }
        }
        const u32x4* chain2 = (const u32x4*)(img + h.off_dfa_chain2);
        for (uint32_t i = 0; i < h.dfa_n_states - h.dfa_n_rows - h.dfa_n_single; i++) {
            const u32x4& q = chain2[i];
            const uint32_t ta = q.x & kDfaStateMask, tb = q.z & kDfaStateMask, ca = q.y >> 24, cb = q.w >> 24;
            if (!leans_on(h.dfa_n_rows + h.dfa_n_single + i, q.y & 0xFFFFFFu) || (q.w & 0xFFFFFFu) != 0u || ca == 0u || cb == 0u || ca == cb || ca >= (1u << h.dfa_log2_classes) || cb >= (1u << h.dfa_log2_classes) ||
                ta >= h.dfa_n_states || tb >= h.dfa_n_states || (q.x & ~kDfaStateMask) != dfa_end_bits(out[ta]) || (q.z & ~kDfaStateMask) != dfa_end_bits(out[tb])) {
                err = "image: DFA two-children record out of range"; return false;
            }
        }
    }
    return true;
}

bool image_sections_in_bounds(const ImageHeader& h)
{
    const uint64_t T = h.total_bytes;
    auto ok = [T](uint64_t off, uint64_t count, uint64_t elem) { return off <= T && count <= (T - off) / elem; };
    if (h.magic != kImageMagic || h.version != kImageVersion || h.case_mode > 1 || T < sizeof(ImageHeader)) return false;
    bool good = ok(h.off_transitions, h.n_transitions, 8) && ok(h.off_offsets, (uint64_t)h.n_states + 1, 4) && ok(h.off_root_ascii, 128, 8) &&
                ok(h.off_canon, h.n_states, 4) && ok(h.off_vlen, h.n_states, 4) && ok(h.off_lower, h.n_lower, 4) &&
                h.ac_goto_log2_cap >= 4 && h.ac_goto_log2_cap <= 31 && ok(h.off_goto, 1ull << h.ac_goto_log2_cap, 16) && ok(h.off_fail, h.n_states, 4);
    if (h.sf_enabled) {
        if (h.sf_bloom_log2_words > 20) return false;
        good = good && ok(h.off_bloom, 1ull << h.sf_bloom_log2_words, 4) && ok(h.off_nodes, h.sf_n_nodes, 32) && ok(h.off_edges, h.n_edges, 64) && h.n_edge_maps == 0;
        for (int t = 0; t < 4; t++) {
            if (!(h.sf_tiers & (1u << t))) continue;
            if (h.tier_log2_cap[t] > 30) return false;
            good = good && ok(h.off_tier[t], 1ull << h.tier_log2_cap[t], 8);
            if (t == 3) good = good && ok(h.off_t4_slots, 2ull << h.tier_log2_cap[t], 64) && (h.off_t4_slots & 63u) == 0;
        }
    }
    if (h.dfa_n_states) {
        if (h.dfa_log2_classes < 3 || h.dfa_log2_classes > 8 || h.dfa_n_states >= kDfaStateMask || h.dfa_chunk < 64 || (h.dfa_chunk & 15u) || h.dfa_warm == 0 || h.root_vlen != 0) return false;
        if (h.dfa_n_rows == 0 || h.dfa_n_rows > h.dfa_n_states || h.dfa_n_rows >= (1u << 24) || h.dfa_n_single > h.dfa_n_states - h.dfa_n_rows) return false;
        good = good && ok(h.off_dfa_chain2, h.dfa_n_states - h.dfa_n_rows - h.dfa_n_single, 16) && (h.off_dfa_chain2 & 15u) == 0;
        if ((1u << h.dfa_hot_log2) > (1u << h.dfa_log2_classes) - 1u || h.dfa_hot_log2 > 8) return false;
        good = good && ok(h.off_dfa_hot, (uint64_t)h.dfa_n_rows << h.dfa_hot_log2, 4) && (h.off_dfa_hot & 15u) == 0;
        good = good && ok(h.off_dfa_next, (uint64_t)h.dfa_n_rows << h.dfa_log2_classes, 4) && ok(h.off_dfa_chain, (uint64_t)h.dfa_n_single + 1u, 8) && (h.off_dfa_chain & 7u) == 0 &&
               ok(h.off_dfa_out, h.dfa_n_states, 8) && ok(h.off_dfa_cls, 256, 1) &&
               (h.off_dfa_next & 15u) == 0 && (h.off_dfa_out & 7u) == 0 && ok(h.off_dfa_fail, h.dfa_n_states, 4) && h.dfa_rare_log2_cap >= 4 && h.dfa_rare_log2_cap <= 30 &&
               ok(h.off_dfa_rare, 1ull << h.dfa_rare_log2_cap, 16) && (h.off_dfa_rare & 15u) == 0;
    }
    return good;
}

}  // namespace am
