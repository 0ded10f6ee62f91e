// am_image.h -- layout of the flattened automaton image in HBM and the per-position walk
// logic shared by the HIP kernels (am_kernels.hip) and the host-side image checker
// (am_imgcheck.cpp, test-only).  Everything here is position-independent: the image is one
// contiguous blob of byte offsets, so it can be broadcast between GPUs over RCCL as-is.
//
// Two structures live in one image (one image per case mode):
//
//  * "AC" section  -- the reference's own packed arrays (Automaton.hs:108-123: Word64
//    transitions, Word32 offsets, 128-entry root table) plus canon[]/vlen[] and the simple
//    lowercase table.  Walked by the general kernel (one lane per chunk, warm-up overlap).
//
//  * "SF" section  -- suffix-filter structures for the failureless fast path: a trie of the
//    REVERSED needles over (case-folded) UTF-8 bytes, so one lane per END position walks
//    backwards; a Bloom filter over the last 1..4 bytes that is staged into LDS; exact
//    hash tables for those suffixes.  No failure links are needed (position parallelism
//    replaces them), and case-insensitivity is baked into the byte edges, so the haystack is
//    never re-encoded and match positions are original byte offsets by construction.
//
// Record semantics (both kernels): at most ONE record per end position -- (haystack,
// end_pos, state) where `state` is the canonical reference state whose machineValues list is
// exactly what the reference's collectMatches (Automaton.hs:522-534) would fold at that
// position.  canon[s] = deepest state on s's fallback chain (s included) that owns values.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define AM_HD __host__ __device__ __forceinline__
#else
#define AM_HD inline
#endif

namespace am {

constexpr uint32_t kImageMagic = 0x31474D41u;   // "AMG1"
constexpr uint32_t kImageVersion = 17;
constexpr uint32_t kUnicodeLowerVersion = 0x0E00;   // Unicode 14.0 (major << 8 | minor): the simple-lowercase table baked into IgnoreCase images (ImageHeader::flags bits 0-15)
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint64_t kWildcard = 0x200000ull;     // Automaton.hs:130-131
constexpr uint32_t kHidxShift = 10;             // haystack-index table granularity: 1 KiB
constexpr uint32_t kSfChunk = 1024;             // bytes per wavefront step in the SF kernel (64 lanes x 16 B)

struct ImageHeader {
    uint32_t magic, version, case_mode, flags;
    uint64_t total_bytes;
    uint32_t n_states, max_needle_cps, root_vlen, ac_chunk;
    // AC section (byte offsets from the image base)
    uint64_t off_transitions, n_transitions;
    uint64_t off_offsets;       // u32[n_states + 1]
    uint64_t off_root_ascii;    // u64[128]
    uint64_t off_canon;         // u32[n_states]
    uint64_t off_vlen;          // u32[n_states] = length machineValues[s]
    uint64_t off_lower;         // i32[n_lower]: lower(cp) = cp + delta[cp] for cp < n_lower
    uint32_t n_lower, sf_enabled;
    // SF section
    uint32_t sf_tiers;          // bit t-1 set: some needle variant is exactly t bytes (t=1..3); bit 3: >= 4 bytes
    uint32_t sf_bloom_log2_words;
    uint32_t sf_n_nodes;
    uint32_t ac_goto_log2_cap;  // AC goto hash: 1 << cap slots
    uint64_t off_bloom;         // u32[1 << sf_bloom_log2_words]
    uint64_t off_tier[4];       // tiers 1-3: u32x2{key, node}[1 << cap]; tier 4: hot fingerprint buckets u32x2[1 << cap] (2 slots each)
    uint32_t tier_log2_cap[4];
    uint64_t off_nodes;         // SfNode[sf_n_nodes]  (32 B: record + inline label of the single outgoing edge)
    uint64_t off_edges;         // SfEdge[n_edges]     (64 B: out-edges of nodes with more than one child, each with a copy of its child's record)
    uint64_t n_edges;
    uint64_t off_edge_maps;     // (images before version 12: selector maps of the nodes with more than 4 children; now 0, see sf_row_first)
    uint64_t n_edge_maps;
    uint64_t off_t4_slots;      // cold side of tier 4: one 64-byte SfSlot per cuckoo slot (2 per bucket): full key + the depth-4 node, its single edge and that edge's child
    uint64_t checksum;          // of everything after the header (checked when an image comes from the host)
    uint64_t off_goto;          // AC: u32x4{state, cp, next, used}[1 << ac_goto_log2_cap], open addressing: (state, cp) -> goto target
    uint64_t off_fail;          // AC: u32[n_states] fallback state (the target of each state's wildcard entry)
    uint32_t sf_t4_children;    // hot entries keyed by FIVE bytes (a heavy depth-4 node's children, am_flatten.cpp); 0: none, the probe never looks for them
    uint32_t sf_row_first;      // edges[sf_row_first .. n_edges): the ROWS of the nodes with more than 4 children (SfNode::label, SfEdge::pad)
    // DFA section (version 13; dfa_n_states == 0: none): the byte-level automaton with every transition resolved, for dictionaries that meet text in which
    // a needle ends every few bytes (am_flatten.cpp decides; k_dfa in am_dfa.hip)
    uint64_t off_dfa_next;      // u32[dfa_n_rows << dfa_log2_classes]: next state (bits 0-27) | bits 28-31: the length of what ends there (kDfaEndShift)
    uint64_t off_dfa_out;       // u32x2[dfa_n_states] {canonical reference state + 1 (0: no needle ends), vlen}
    uint64_t off_dfa_cls;       // u8[256]: byte -> class (IgnoreCase: the ASCII fold is part of the map); class 0 = bytes no needle contains; kDfaRare = a byte
                                // that few edges carry: no column, dfa_rare_step
    uint64_t off_dfa_fail;      // u32[dfa_n_states]: fallback state (the rare-byte walk)
    uint64_t off_dfa_rare;      // u32x4{state, byte, child | its end bits, used}[1 << dfa_rare_log2_cap]: the edges on rare bytes, open addressing
    uint32_t dfa_rare_log2_cap;
    uint32_t dfa_n_rows;        // states [0, dfa_n_rows) have a dense row in `next`; the others a chain record (off_dfa_chain)
    uint64_t off_dfa_chain;     // u32x2[dfa_n_single + 1]: {target | its end bits, its class << 24 | the row state R this one leans on: the nearest row state on its chain of fallbacks; any other class is answered by R's row}
    uint32_t dfa_n_states, dfa_log2_classes;
    uint32_t dfa_warm;          // bytes of history that determine the state: longest needle (variant) in bytes - 1
    uint32_t dfa_chunk;         // bytes of the batch one lane owns (multiple of 16)
    // (version 16) rows are numbered by weight, columns by the dictionary's own use of them, and the first columns of every row exist a second time, dense:
    uint64_t off_dfa_hot;       // u32[dfa_n_rows << dfa_hot_log2]: hot[(row << dfa_hot_log2) + class - 1] = next[(row << dfa_log2_classes) + class] for 1 <= class <= 2^dfa_hot_log2
    uint32_t dfa_hot_log2;
    // (version 17) states [dfa_n_rows, dfa_n_rows + dfa_n_single) have a single-child record (off_dfa_chain), the rest a two-children record:
    uint32_t dfa_n_single;
    uint64_t off_dfa_chain2;    // u32x4[dfa_n_states - dfa_n_rows - dfa_n_single]: {target a | its end bits, class a << 24 | the row state R this one leans on, target b | its end bits, class b << 24}
};

// Resolved pointers, passed to kernels by value (SGPRs).
struct AcView {
    const uint64_t* transitions;
    const uint32_t* offsets;
    const uint64_t* root_ascii;
    const uint32_t* canon;
    const uint32_t* vlen;
    const int32_t* lower;
    uint32_t n_lower, max_needle_cps, chunk, root_vlen;
    const struct u32x4* goto_tab;   // (state, cp) -> next, see ImageHeader::off_goto
    const uint32_t* fail;
    uint32_t goto_log2_cap;
};

struct alignas(8) u32x2 { uint32_t x, y; };
struct alignas(16) u32x4 { uint32_t x, y, z, w; };

// Path-compressed ("Patricia") trie of the reversed needles.  An edge = one selector byte (the next
// haystack byte going backwards) + up to 16 further bytes that must match (`skip`), stored in TEXT
// order, right-aligned in a 16-byte slot, so one unaligned 16 B haystack load + one 16 B label load
// verify the whole edge.  Long unary chains (the tails of the needles) are 1-2 edges instead of one
// dependent load per byte.
struct alignas(32) SfNode {
    uint32_t x;          // canonical reference state + 1 (0: no needle ends here)
    uint32_t y;          // vlen = length machineValues[state]
    uint32_t z;          // n_edges == 1: child node; n_edges > 1: first SfEdge index
    uint32_t w;          // n_edges (bits 0-15) | selector byte of the single edge (16-23) | its skip length (24-31)
    uint32_t label[4];   // n_edges == 1: skip bytes of the single edge; 2..4 edges: label[0] = their selector bytes (edge i in byte i);
                         // more: label[0] = the node's ROW: the edge of selector byte b is the line edges[label[0] + b], if that line is this
                         // node's (SfEdge::pad); label[1..3] = which selector bytes exist, folded to 96 bits (bit b % 96): a clear bit saves the load
};
// An out-edge of a branching node, one 64-byte line: the edge AND a copy of the child's record, so that one step of the walk
// (choose the edge, compare its label, arrive at the child) is one dependent load.
struct alignas(64) SfEdge {
    uint32_t byte, child, skip;
    uint32_t pad;        // 0 in a node's contiguous run of edges [z, z + n); in the row region (>= sf_row_first): owner's SfNode::z + 1, kNone = empty line
    uint32_t label[4];
    SfNode to;           // = nodes[child]
};
// A node with more than 4 children finds the edge of selector byte b WITHOUT a lookup structure of its own: row displacement.  The flattener gives the
// node a row offset such that the lines  row + b  of all its selector bytes are free, and puts copies of its edges there (am_flatten.cpp); a line
// says whose it is.  One dependent load per step for every kind of node (a selector map per node, as before version 12, cost such a step a second trip).
constexpr uint32_t kMaxSkip = 16;

// Cold side of the 4-byte-suffix table: one 64-byte line per cuckoo slot, so that the position of a hot slot that matched IS the
// address of everything the resolve needs for a typical needle: the full key, the depth-4 node (own needle end), its single edge
// (selector, skip, label) and the child's needle end.  A needle of <= 4 + 1 + 16 bytes resolves with this one line; longer ones
// and branching nodes continue in `nodes` / `edges`.  A branching depth-4 node that was split into one hot entry per child
// (am_flatten.cpp) has one SfSlot per child too: w/label/z/c* describe that child's edge (kSlotChildCopy).
struct alignas(64) SfSlot {
    uint32_t key, flags;   // flags: kSlotOccupied, kSlotChildCopy
    uint32_t x, y;         // needle end AT the depth-4 node: canonical state + 1 (0: none), vlen
    uint32_t w;            // bits 0-15: 0 = leaf, 1 = the single (or this copy's) edge is described here, >= 2 = branching: continue at nodes[z]
                           // bits 16-23 selector byte, 24-31 skip length of that edge
    uint32_t z;            // w & 0xFFFF == 1: child node id; >= 2: the depth-4 node's own id
    uint32_t cx, cy;       // needle end at the child (state + 1, vlen)
    uint32_t label[4];     // skip bytes of the edge (text order, right-aligned); branching node: label[1..3] = its SfNode::label[1..3]
    uint32_t cw;           // the child's SfNode::w (its edge count decides whether the walk goes on)
    uint32_t ez, el0;      // branching node (w & 0xFFFF >= 2): its SfNode::z (first edge) and label[0] (inline selectors / row), so
                           // that the walk starts from this line without loading the node's record
    uint32_t pad;
};
constexpr uint32_t kSlotOccupied = 1u, kSlotChildCopy = 2u;

struct SfView {
    const uint32_t* bloom;
    const u32x2* tier[3];    // exact tables for needles (variants) of exactly 1, 2, 3 bytes
    // 4-byte suffixes: (2,2) cuckoo table.  A key lives in one of the 2 slots of bucket_a(key) or
    // bucket_b(key).  HOT side (what the probe reads, 8 B per bucket, ~1 MiB for 100k needles, so
    // it stays in each XCD's L2): per slot an 11-bit fingerprint, flags, and the next TWO bytes the
    // trie requires after the 4-byte suffix when it is a plain chain there.  COLD side (read only
    // when a needle may really end at the position): the full keys and the node ids.
    const u32x2* t4_hot;
    const SfSlot* t4_slots;  // 2 per bucket, same index as the hot slot
    const SfNode* nodes;
    const SfEdge* edges;
    uint32_t bloom_log2_words, tiers;
    uint32_t tier_log2_cap[4];
    uint32_t n_nodes;
    uint32_t t4_children;    // the table holds five-byte child entries of heavy depth-4 nodes (sf_probe_children)
};

struct DfaView {
    const uint32_t* next;
    const u32x2* out;
    const uint8_t* cls;
    const uint32_t* fail;
    const u32x4* rare;
    const u32x2* chain;
    const uint32_t* hot;     // columns 1 .. 2^hot_log2 of every row, dense (ImageHeader::off_dfa_hot)
    uint32_t hot_log2;
    const u32x4* chain2;     // two-children records (ImageHeader::off_dfa_chain2)
    uint32_t n_single;       // states [n_rows, n_rows + n_single): single-child records
    uint32_t n_states, n_rows, log2_classes, warm, chunk, rare_log2_cap;
    uint32_t ic;             // IgnoreCase image: haystack bytes A-Z count as a-z (the class map already says so; the rare-byte walk has to be told)
};
// a transition entry = next state (bits 0-27) | what ends there (bits 28-31): 0 nothing, 1..14 = vlen of the needle-end list (a count needs nothing else), 15 = longer, see out[]
constexpr uint32_t kDfaStateMask = 0x0FFFFFFFu, kDfaEndShift = 28u, kDfaEndLookUp = 15u;
AM_HD uint32_t dfa_end_bits(const u32x2& out_entry) { return out_entry.x ? (out_entry.y < kDfaEndLookUp ? out_entry.y : kDfaEndLookUp) << kDfaEndShift : 0u; }
constexpr uint32_t kDfaRare = 0xFFu;
constexpr uint32_t kDfaNoChild = 0xFEu;         // chain record of a state without a child: no class equals it

struct BatchView {
    const uint8_t* text;       // concatenated haystack bytes, 16-B aligned, readable up to round_up(total, 16)
    const uint64_t* offsets;   // n_hay + 1
    const uint32_t* hidx;      // (total >> 10) + 2 entries: haystack containing byte min(k << 10, total - 1)
    uint64_t total;
    uint32_t n_hay, pad;
};

inline AcView make_ac_view(const void* base, const ImageHeader& h)
{
    const uint8_t* b = (const uint8_t*)base;
    AcView v;
    v.transitions = (const uint64_t*)(b + h.off_transitions);
    v.offsets = (const uint32_t*)(b + h.off_offsets);
    v.root_ascii = (const uint64_t*)(b + h.off_root_ascii);
    v.canon = (const uint32_t*)(b + h.off_canon);
    v.vlen = (const uint32_t*)(b + h.off_vlen);
    v.lower = (const int32_t*)(b + h.off_lower);
    v.n_lower = h.n_lower; v.max_needle_cps = h.max_needle_cps; v.chunk = h.ac_chunk; v.root_vlen = h.root_vlen;
    v.goto_tab = (const u32x4*)(b + h.off_goto); v.fail = (const uint32_t*)(b + h.off_fail); v.goto_log2_cap = h.ac_goto_log2_cap;
    return v;
}

inline DfaView make_dfa_view(const void* base, const ImageHeader& h)
{
    const uint8_t* b = (const uint8_t*)base;
    DfaView v;
    v.next = (const uint32_t*)(b + h.off_dfa_next); v.out = (const u32x2*)(b + h.off_dfa_out); v.cls = b + h.off_dfa_cls;
    v.fail = (const uint32_t*)(b + h.off_dfa_fail); v.rare = (const u32x4*)(b + h.off_dfa_rare); v.chain = (const u32x2*)(b + h.off_dfa_chain);
    v.n_states = h.dfa_n_states; v.n_rows = h.dfa_n_rows; v.log2_classes = h.dfa_log2_classes; v.warm = h.dfa_warm; v.chunk = h.dfa_chunk; v.rare_log2_cap = h.dfa_rare_log2_cap; v.ic = h.case_mode;
    v.hot = (const uint32_t*)(b + h.off_dfa_hot); v.hot_log2 = h.dfa_hot_log2;
    v.chain2 = (const u32x4*)(b + h.off_dfa_chain2); v.n_single = h.dfa_n_single;
    return v;
}

inline SfView make_sf_view(const void* base, const ImageHeader& h)
{
    const uint8_t* b = (const uint8_t*)base;
    SfView v;
    v.bloom = (const uint32_t*)(b + h.off_bloom);
    for (int t = 0; t < 3; t++) v.tier[t] = (const u32x2*)(b + h.off_tier[t]);
    v.t4_hot = (const u32x2*)(b + h.off_tier[3]);
    v.t4_slots = (const SfSlot*)(b + h.off_t4_slots);
    for (int t = 0; t < 4; t++) v.tier_log2_cap[t] = h.tier_log2_cap[t];
    v.nodes = (const SfNode*)(b + h.off_nodes);
    v.edges = (const SfEdge*)(b + h.off_edges);
    v.bloom_log2_words = h.sf_bloom_log2_words; v.tiers = h.sf_tiers; v.n_nodes = h.sf_n_nodes;
    v.t4_children = h.sf_t4_children;
    return v;
}

// ------------------------------------------------------------------ small helpers

// one 16-byte load (global_load_dwordx4 on the device: a single request per lane instead of up to four)
AM_HD u32x4 load16(const u32x4* p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint4 v = *reinterpret_cast<const uint4*>(p);
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));   // keep it ONE dwordx4: stop load narrowing/splitting
    return u32x4{v.x, v.y, v.z, v.w};
#else
    return *p;
#endif
}

AM_HD uint32_t mulhi32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

// ASCII A-Z -> a-z on one byte.  Safe on raw UTF-8: bytes 0x41..0x5A only occur as ASCII.
AM_HD uint32_t fold_byte(uint32_t b) { return (b - 0x41u < 26u) ? b + 0x20u : b; }

// The same on four packed bytes (SWAR).
AM_HD uint32_t fold_dword(uint32_t x)
{
    uint32_t hept = x & 0x7f7f7f7fu;
    uint32_t ge_a = hept + 0x3f3f3f3fu;      // bit 7 set iff heptet >= 0x41
    uint32_t gt_z = hept + 0x25252525u;      // bit 7 set iff heptet >  0x5A
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t up = __builtin_amdgcn_bitop3_b32(ge_a, gt_z, x, 0x10);      // ge_a & ~gt_z & ~x in one v_bitop3 (six VALU per dword in all)
#else
    const uint32_t up = ge_a & ~gt_z & ~x;
#endif
    return x | ((up >> 2) & 0x20202020u);
}

// Bloom filter over needle suffixes: one 32-bit word per key, FOUR bits in it.  Word and bits both come from ONE 32-bit
// multiply: the word index is the top bits of the product, and the four bit positions are one of 512 precomputed masks,
// selected by product bits 2..10.  In the kernel the mask table sits in LDS next to the filter, so a position is tested
// with two LDS reads, one AND and one compare -- `(word & mask) == mask` -- instead of extracting and shifting three bit
// fields (the VALU is the bottleneck of k_sf, the LDS pipe has room).  Measured on the 100k-needle workload (128 KiB
// filter, 3.8 keys per word): 4.3 % false positives, the same as three independent bit fields gave; on the sparse
// filters of small automata four bits beat three and two.  Tier 4 (the hot one) needs no salt.
constexpr uint32_t kBloomMul = 0x9E3779B1u;
constexpr uint32_t kBloomMaskLog2 = 9;                          // 512 masks = 2 KiB of LDS
constexpr uint32_t kBloomMasks = 1u << kBloomMaskLog2;
AM_HD uint32_t bloom_hash(uint32_t key, uint32_t tier) { return (key + (4u - tier) * 0x7F4A7C15u) * kBloomMul; }
AM_HD uint32_t bloom_word(uint32_t h, uint32_t log2_words) { return h >> (32u - log2_words); }
AM_HD uint32_t bloom_mask_index(uint32_t h) { return (h >> 2) & (kBloomMasks - 1u); }
// mask i of the table: four distinct bits chosen by a mix of i (the kernel fills its LDS copy with this at start-up)
AM_HD uint32_t bloom_mask_entry(uint32_t i)
{
    uint32_t x = (i + 1u) * 0x9E3779B1u;
    x ^= x >> 15; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    uint32_t m = 0;
    for (uint32_t k = 0; k < 4u; k++) {
        uint32_t b = (x >> (5u * k)) & 31u;
        while (m & (1u << b)) b = (b + 1u) & 31u;
        m |= 1u << b;
    }
    return m;
}
// mask of hash h; `tab` = the 512-entry table when the caller has one (the kernel's LDS copy), else null
AM_HD uint32_t bloom_mask(uint32_t h, const uint32_t* tab = nullptr) { return tab ? tab[bloom_mask_index(h)] : bloom_mask_entry(bloom_mask_index(h)); }
// true iff all mask bits of h are set in the filter word v
AM_HD bool bloom_hit(uint32_t v, uint32_t h, const uint32_t* tab = nullptr) { const uint32_t m = bloom_mask(h, tab); return (v & m) == m; }

AM_HD uint32_t tier_slot(uint32_t key, uint32_t log2_cap) { return (key * 0x85EBCA6Bu) >> (32u - log2_cap); }

// Exact suffix table lookup: open addressing, linear probing, empty = {*, kNone}.
AM_HD uint32_t tier_lookup(const u32x2* tab, uint32_t log2_cap, uint32_t key)
{
    const uint32_t cap_mask = (1u << log2_cap) - 1u;
    uint32_t i = tier_slot(key, log2_cap);
    for (;;) {
        u32x2 e = tab[i];
        if (e.y == kNone) return kNone;
        if (e.x == key) return e.y;
        i = (i + 1u) & cap_mask;
    }
}

// Largest h with offsets[h] <= pos (pos < total), bracketed by the 1-KiB haystack index.
AM_HD uint32_t find_haystack(const BatchView& b, uint64_t pos)
{
    const uint64_t k = pos >> kHidxShift;
    uint32_t lo = b.hidx[k], hi = b.hidx[k + 1];
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo + 1u) / 2u;
        if (b.offsets[mid] <= pos) lo = mid; else hi = mid - 1u;
    }
    return lo;
}

// ------------------------------------------------------------------ SF: verify one end position
//
// `gpos` = global index of the LAST byte of the candidate match, `avail` = bytes of its haystack up
// to and including gpos.  Finds the deepest terminal of the reversed-needle trie along
// text[gpos], text[gpos-1], ...  Returns true and (state, vlen) if any needle ends here.
// Memory traffic per candidate is what bounds the kernel (address-divergent loads), so the common
// case is two loads: the last 8 haystack bytes (one unaligned 8 B load) and one 16 B table entry
// that already carries the depth-4 node's terminal flag and its single outgoing edge byte.

// last 8 bytes ending at global index gpos: w = bytes gpos-3..gpos, w2 = bytes gpos-7..gpos-4, newest
// byte on top.  Bytes that lie before the haystack's start (in the previous haystack, or before the
// buffer: zeros) are never interpreted: every use is guarded by `avail`.
AM_HD void load_suffix8(const uint8_t* text, uint64_t gpos, uint32_t& w, uint32_t& w2)
{
    // branch-free: always one 8-byte read; for the first 7 bytes of the batch read text[0..7] and shift
    const uint64_t base = gpos >= 7 ? gpos - 7 : 0;
    const uint32_t drop = gpos >= 7 ? 0u : (uint32_t)(7 - gpos);     // missing leading bytes
    uint64_t v;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint64_t __attribute__((aligned(1), may_alias)) u64_unaligned;
    v = *reinterpret_cast<const u64_unaligned*>(text + base);
#else
    v = 0;
    for (uint32_t j = 0; j < 8; j++) v |= (uint64_t)text[base + j] << (8u * j);
#endif
    v = drop ? (v << (8u * drop)) : v;          // byte gpos ends up in the top byte, bytes before the buffer are zero
    w2 = (uint32_t)v; w = (uint32_t)(v >> 32);
}

// 16 haystack bytes ending just before global index `end` (text order, dword 3 = the newest four)
AM_HD void load_text16(const uint8_t* text, uint64_t end, uint32_t (&t)[4])
{
    if (end >= 16) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
        const u32x4_u v = *reinterpret_cast<const u32x4_u*>(text + end - 16);
        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
#else
        for (int i = 0; i < 4; i++) { t[i] = 0; for (int j = 0; j < 4; j++) t[i] |= (uint32_t)text[end - 16 + 4 * i + j] << (8 * j); }
#endif
    } else {
        for (int i = 0; i < 4; i++) {
            t[i] = 0;
            for (int j = 0; j < 4; j++) { const uint64_t off = 4u * i + j; if (end + off >= 16) t[i] |= (uint32_t)text[end + off - 16] << (8 * j); }
        }
    }
}

// do the last `skip` (1..16) bytes of t equal the right-aligned label?
AM_HD bool label_match(const uint32_t (&t)[4], const uint32_t (&l)[4], uint32_t skip)
{
    const uint32_t r = kMaxSkip - skip;      // leading bytes to ignore
    uint32_t diff = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t g = r > 4u * i ? (r - 4u * i < 4u ? r - 4u * i : 4u) : 0u;
        const uint32_t mask = g >= 4u ? 0u : 0xFFFFFFFFu << (8u * g);
        diff |= (t[i] ^ l[i]) & mask;
    }
    return diff == 0;
}

AM_HD void load_node(const SfNode* p, SfNode& n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint4 a = reinterpret_cast<const uint4*>(p)[0], b = reinterpret_cast<const uint4*>(p)[1];   // same 32-byte sector
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
    n.x = a.x; n.y = a.y; n.z = a.z; n.w = a.w; n.label[0] = b.x; n.label[1] = b.y; n.label[2] = b.z; n.label[3] = b.w;
#else
    n = *p;
#endif
}

// Tier-4 hashing.  One multiply gives bucket A (top bits) and the fingerprint (11 bits below them); a
// second, differently mixed multiply gives bucket B.
AM_HD uint32_t t4_hash_a(uint32_t key) { return key * 0x85EBCA6Bu; }
AM_HD uint32_t t4_hash_b(uint32_t key) { return (key ^ (key >> 15)) * 0xC2B2AE35u; }
AM_HD uint32_t t4_bucket(uint32_t h, uint32_t log2_buckets) { return h >> (32u - log2_buckets); }
AM_HD uint32_t t4_fingerprint(uint32_t ha, uint32_t log2_buckets) { return (ha >> (log2_buckets < 22u ? 22u - log2_buckets : 0u)) & 0x3FFu; }
// Hot slot word, laid out so that the probe decides a slot with TWO instructions (v_lshl_or + v_bitop3, no compares,
// no selects -- the VALU is the bottleneck of k_sf):
//   bits 0-4   k8 = 8 x (number of bytes before the 4-byte suffix that are fixed): 0 when a needle (variant) ends at the
//              depth-4 node or the node has several outgoing edges; 8 when it has exactly one (sel1 = its selector byte);
//              16 when after sel1 the trie is still a plain chain without a needle end (the next byte must be sel2)
//   bits 5-14  fingerprint, bit 15 occupied, bits 16-23 sel1, bits 24-31 sel2 (0 where not fixed)
// A slot may belong to a needle ending at a position with fingerprint fp and preceding bytes nb (nearest), nb2 iff
// t4_slot_diff(slot, t4_expect(fp, nb | nb2 << 8)) == 0: the shift by k8 (the hardware reads the low 5 bits of the
// slot itself as the amount) blanks the selector fields that are not fixed.  Whether those bytes lie inside the
// haystack is not checked here; phase 2 is exact, a candidate at the very start of a haystack is merely deferred.
constexpr uint32_t kT4Occupied = 1u << 15;
// HEAVY depth-4 nodes (round 5).  A node that branches, with no needle ending at it, fixes no byte before the suffix (k8 = 0): every position
// with its 4-byte suffix is deferred.  Dictionaries of natural-language words are full of them ("tion", "ing ", "ness" have dozens of different
// bytes before them): 329 deferred positions per KiB of natural text where 203 share SIX bytes with some needle.  Such a node's hot word carries
// kT4Heavy (bits 16-31 of a k8 = 0 word are ignored by t4_slot_diff) and each of its children has a hot entry of its own under the FIVE-byte key
// t4_key5(key, child byte), placed by the same cuckoo scheme anywhere in the table, with the child's own fixed bytes as selectors (bytes six
// and seven of the context).  A position whose agreeing slot is heavy is deferred only if one of its child entries agrees as well
// (sf_probe_children: two more buckets, requested when the first two have been looked at).
constexpr uint32_t kT4Heavy = 1u << 16;
AM_HD uint32_t t4_key5(uint32_t key, uint32_t b) { return (key ^ ((b + 1u) * 0x9E3779B1u)) + 0x7F4A7C15u; }
AM_HD uint32_t t4_slot_word(uint32_t fp, uint32_t fixed_bytes, uint32_t sel1, uint32_t sel2)
{
    return (8u * fixed_bytes) | (fp << 5) | kT4Occupied | (fixed_bytes >= 1 ? sel1 << 16 : 0u) | (fixed_bytes >= 2 ? sel2 << 24 : 0u);
}
AM_HD uint32_t t4_expect(uint32_t fp, uint32_t nbs) { return kT4Occupied | (fp << 5) | (nbs << 16); }
AM_HD uint32_t t4_slot_diff(uint32_t slot, uint32_t expect)
{
    const uint32_t ignore = (0xFFFF0000u << (slot & 31u)) | 0x1Fu;
    return (slot ^ expect) & ~ignore;
}

// Phase 1, N candidates per lane at once, in two halves so that the kernel can leave the loads in flight across other work:
// sf_probe_issue requests the two hot buckets of every candidate (inputs straight from the filter stage: w = the 4 bytes ending at
// the position, nbs = the two bytes before them, nearest in bits 0-7, the other in bits 8-15; all case-folded) and returns the raw
// buckets + the word a matching slot must equal; sf_probe_decide looks at them: defer[k] = true: a needle may end here, phase 2
// must look (exactly); hint[k] = which of the four candidate slots agreed.  No data-dependent loop, the only memory traffic is the
// two 8-byte buckets.
template <int N>
AM_HD void sf_probe_issue(const SfView& s, const uint32_t (&w)[N], const uint32_t (&nbs)[N], const uint64_t (&avail)[N], const bool (&valid)[N],
                          u32x2 (&ba)[N], u32x2 (&bb)[N], uint32_t (&expect)[N], const bool one_bucket = false /* timing experiment (AM_SF_ABLATE=12): the second load asks for the first bucket again; wrong results */)
{
    const uint32_t lb = s.tier_log2_cap[3];
#pragma unroll
    for (int k = 0; k < N; k++) {
        const bool probe = valid[k] && (s.tiers & 8u) && avail[k] >= 4;
        const uint32_t ha = t4_hash_a(w[k]), hb = t4_hash_b(w[k]);
        expect[k] = t4_expect(t4_fingerprint(ha, lb), nbs[k] & 0xFFFFu);
        ba[k] = u32x2{0, 0}; bb[k] = ba[k];                      // empty buckets for the lanes that do not probe
        if (probe) {
#if defined(__HIP_DEVICE_COMPILE__)
            const uint2 ra = *reinterpret_cast<const uint2*>(s.t4_hot + t4_bucket(ha, lb)), rb = *reinterpret_cast<const uint2*>(s.t4_hot + t4_bucket(one_bucket ? ha : hb, lb));
            ba[k] = u32x2{ra.x, ra.y}; bb[k] = u32x2{rb.x, rb.y};
#else
            ba[k] = s.t4_hot[t4_bucket(ha, lb)]; bb[k] = s.t4_hot[t4_bucket(hb, lb)];
#endif
        }
    }
}

template <int N>
AM_HD void sf_probe_decide(const SfView& s, const u32x2 (&ba)[N], const u32x2 (&bb)[N], const uint32_t (&expect)[N], const bool (&valid)[N],
                           bool (&defer)[N], uint32_t (&hint)[N])
{
#pragma unroll
    for (int k = 0; k < N; k++) {
        const uint32_t e = expect[k];
        const uint32_t za = t4_slot_diff(ba[k].x, e), zb = t4_slot_diff(ba[k].y, e), zc = t4_slot_diff(bb[k].x, e), zd = t4_slot_diff(bb[k].y, e);
        const uint32_t za_b = za < zb ? za : zb, zc_d = zc < zd ? zc : zd;
        uint32_t hit = (uint32_t)((za_b < zc_d ? za_b : zc_d) == 0u);   // some slot agrees (empty buckets were substituted for non-probes)
        hit |= (uint32_t)((s.tiers & 7u) != 0u);                 // 1..3-byte needles: always consult their tables
        defer[k] = valid[k] & (hit != 0u);
        // which of the four candidate slots agreed (the first one; 3 also when none did): phase 2 reads exactly that slot's line
        hint[k] = za_b == 0u ? (uint32_t)(za != 0u) : 2u + (uint32_t)(zc != 0u);
    }
}

// ... and whether EVERY slot that agreed is a heavy node's (kT4Heavy): then -- and only then: a plain slot of another key that agrees by its
// fingerprint keeps the position deferred, as before -- the position is deferred only if a child entry agrees too (sf_probe_children)
template <int N>
AM_HD void sf_probe_heavy(const SfView& s, const u32x2 (&ba)[N], const u32x2 (&bb)[N], const uint32_t (&expect)[N], const bool (&defer)[N], bool (&heavy)[N])
{
#pragma unroll
    for (int k = 0; k < N; k++) {
        const uint32_t e = expect[k], words[4] = {ba[k].x, ba[k].y, bb[k].x, bb[k].y};
        bool plain = false, hv = false;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const bool agree = t4_slot_diff(words[c], e) == 0u;
            const bool is_heavy = (words[c] & (kT4Heavy | 0x1Fu)) == kT4Heavy;      // (k8 = 0: bits 16-31 are not selector bytes)
            plain = plain || (agree && !is_heavy);
            hv = hv || (agree && is_heavy);
        }
        // (with 1..3-byte needles every position is deferred for their tables' sake)
        heavy[k] = defer[k] && (s.tiers & 7u) == 0u && hv && !plain;
    }
}

// the child entries of heavy candidates: key5 = t4_key5(w, nearest byte before the window), expect5 = t4_expect(fingerprint of key5, the two bytes
// before that); defer stays true only where one of the four slots of key5's two buckets agrees
// hint[k] of a position a child entry speaks for becomes 4 | (which of key5's four slots): phase 2 then reads THAT slot's line -- the child's edge
// and the node behind it, one or two steps further down than the heavy node's own line
template <int N>
AM_HD void sf_probe_children(const SfView& s, const uint32_t (&key5)[N], const uint32_t (&expect5)[N], const bool (&heavy)[N], bool (&defer)[N], uint32_t (&hint)[N])
{
    const uint32_t lb = s.tier_log2_cap[3];
    u32x2 ca[N], cb[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        ca[k] = u32x2{0, 0}; cb[k] = ca[k];
        if (heavy[k]) {
#if defined(__HIP_DEVICE_COMPILE__)
            const uint2 ra = *reinterpret_cast<const uint2*>(s.t4_hot + t4_bucket(t4_hash_a(key5[k]), lb)), rb = *reinterpret_cast<const uint2*>(s.t4_hot + t4_bucket(t4_hash_b(key5[k]), lb));
            ca[k] = u32x2{ra.x, ra.y}; cb[k] = u32x2{rb.x, rb.y};
#else
            ca[k] = s.t4_hot[t4_bucket(t4_hash_a(key5[k]), lb)]; cb[k] = s.t4_hot[t4_bucket(t4_hash_b(key5[k]), lb)];
#endif
        }
    }
#pragma unroll
    for (int k = 0; k < N; k++) {
        if (!heavy[k]) continue;
        const uint32_t e = expect5[k];
        const uint32_t za = t4_slot_diff(ca[k].x, e), zb = t4_slot_diff(ca[k].y, e), zc = t4_slot_diff(cb[k].x, e), zd = t4_slot_diff(cb[k].y, e);
        defer[k] = za == 0u || zb == 0u || zc == 0u || zd == 0u;
        if (defer[k]) hint[k] = 4u | (za == 0u ? 0u : zb == 0u ? 1u : zc == 0u ? 2u : 3u);
    }
}
// the two inputs of sf_probe_children from the window and the THREE bytes before it (nbs: nearest in bits 0-7)
AM_HD void t4_child_inputs(const SfView& s, uint32_t w, uint32_t nbs3, uint32_t& key5, uint32_t& expect5)
{
    key5 = t4_key5(w, nbs3 & 0xFFu);
    expect5 = t4_expect(t4_fingerprint(t4_hash_a(key5), s.tier_log2_cap[3]), (nbs3 >> 8) & 0xFFFFu);
}

// nbs: the THREE bytes before the window (nearest in bits 0-7); the third only matters to heavy nodes' child entries
template <int N>
AM_HD void sf_probe_n(const SfView& s, const uint32_t (&w)[N], const uint32_t (&nbs)[N], const uint64_t (&avail)[N],
                      const bool (&valid)[N], bool (&defer)[N], uint32_t (&hint)[N])
{
    u32x2 ba[N], bb[N];
    uint32_t expect[N];
    sf_probe_issue<N>(s, w, nbs, avail, valid, ba, bb, expect);
    sf_probe_decide<N>(s, ba, bb, expect, valid, defer, hint);
    if (s.t4_children) {
        bool heavy[N]; uint32_t key5[N], expect5[N];
        sf_probe_heavy<N>(s, ba, bb, expect, defer, heavy);
#pragma unroll
        for (int k = 0; k < N; k++) t4_child_inputs(s, w[k], nbs[k], key5[k], expect5[k]);
        sf_probe_children<N>(s, key5, expect5, heavy, defer, hint);
    }
}

// ---------------------------------------------------------------------------------------------------
// Phase 2 ("resolve"): exact answer for the few positions that survived filter + probe, N positions
// per lane in lock step.  Every step issues the loads of all N items before any of them is consumed,
// so the N dependent-load chains (haystack bytes -> cold bucket -> trie node -> [edge -> child ...])
// overlap instead of adding up.  Data-dependent loops live only here.

// is the flag set in any lane of the wavefront?  (uniform: lets the whole wavefront skip a rarely needed block of loads)
AM_HD bool wave_any(bool x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __ballot(x) != 0ull;
#else
    return x;
#endif
}

AM_HD void node_from_raw(const u32x4& a, const u32x4& b, SfNode& n)
{
    n.x = a.x; n.y = a.y; n.z = a.z; n.w = a.w; n.label[0] = b.x; n.label[1] = b.y; n.label[2] = b.z; n.label[3] = b.w;
}

// SHORT = false promises that the automaton has no needle (variant) shorter than 4 bytes (s.tiers & 7 == 0): step 5 and the
// three small tables drop out of the kernel.  Depths are 32-bit: a needle is far shorter than 4 GiB, and the bytes
// available in the haystack only matter up to that.
// `hint` = which of the four candidate slots the probe saw agree (sf_probe_n), or 4 | which of the four slots of the position's five-byte key (a child
// entry of a heavy node); any value is correct, the right one saves loads.
// `between` runs after the slot-line loads have been issued and before they are used: the kernel computes avail64 there
// (haystack index -> offsets: two dependent loads of its own), so that chain overlaps with haystack bytes -> slot line
// instead of preceding it.  avail64 is not read before that.
struct SfNoHook { AM_HD void operator()() const {} };

// Phase 2 in three pieces (k_resolve runs them as separate stages: the head for every item, the walk only for the items that need
// it, 64 of those at a time; the host checker and sf_resolve_n run them back to back):
//   sf_resolve_head   steps 1-3: haystack bytes, the slot line, what the slot line settles
//   sf_resolve_walk   step 4: the compressed trie, backwards along the haystack, to the deepest needle end
//   sf_resolve_short  step 5: needles of 1..3 bytes (only if nothing longer ends at the position)
// State handed from the head to the walk, per item: go (the walk has something to do), node (where it continues), have_rec (rec already
// holds that node's record: a branching depth-4 node described by its slot line), depth, best_state (+1; 0: none yet), best_vlen,
// avail (bytes of the haystack up to and including gpos, clamped to 32 bits), w2 (the four bytes before the 4-byte suffix, folded).
template <bool IC, int N, class Between = SfNoHook>
AM_HD void sf_resolve_head(const SfView& s, const uint8_t* text, const uint64_t (&gpos)[N], const uint64_t (&avail64)[N], const bool (&valid)[N],
                           const uint32_t (&hint)[N], Between between, uint32_t (&w)[N], uint32_t (&w2)[N], uint32_t (&avail)[N],
                           uint32_t (&best_state)[N], uint32_t (&best_vlen)[N], uint32_t (&depth)[N], bool (&go)[N], uint32_t (&node)[N],
                           SfNode (&rec)[N], bool (&have_rec)[N], uint32_t (&t16)[N][4], uint64_t* dbg_iters = nullptr)
{
    // ---- step 1: the last 8 haystack bytes and the 16 before the 4-byte suffix (what the first edge label is compared with)
#pragma unroll
    for (int k = 0; k < N; k++) {
        w[k] = 0; w2[k] = 0; t16[k][0] = t16[k][1] = t16[k][2] = t16[k][3] = 0;
        if (valid[k]) { load_suffix8(text, gpos[k], w[k], w2[k]); if (gpos[k] >= 4) load_text16(text, gpos[k] - 4, t16[k]); }
    }
#pragma unroll
    for (int k = 0; k < N; k++) if (IC) {
        w[k] = fold_dword(w[k]); w2[k] = fold_dword(w2[k]);
        t16[k][0] = fold_dword(t16[k][0]); t16[k][1] = fold_dword(t16[k][1]); t16[k][2] = fold_dword(t16[k][2]); t16[k][3] = fold_dword(t16[k][3]);
    }
    // ---- step 2: the slot the probe pointed at: ONE 64-byte line with the full key, the depth-4 node, its edge and the edge's child
    SfSlot sl[N];
    uint32_t slot_of[4][N];
    bool look[N];
    {
        const u32x4* slots16 = reinterpret_cast<const u32x4*>(s.t4_slots);
        const uint32_t lb = s.tier_log2_cap[3];
        u32x4 q0[N], q1[N], q2[N], q3[N];
#pragma unroll
        for (int k = 0; k < N; k++) {
            look[k] = valid[k] && (s.tiers & 8u);
            const uint32_t ba = look[k] ? t4_bucket(t4_hash_a(w[k]), lb) : 0u, bb = look[k] ? t4_bucket(t4_hash_b(w[k]), lb) : 0u;
            slot_of[0][k] = 2u * ba; slot_of[1][k] = 2u * ba + 1u; slot_of[2][k] = 2u * bb; slot_of[3][k] = 2u * bb + 1u;
            const uint32_t h = hint[k] & 3u;
            uint32_t idx = h == 0 ? slot_of[0][k] : h == 1 ? slot_of[1][k] : h == 2 ? slot_of[2][k] : slot_of[3][k];
            if (look[k] && (hint[k] & 4u)) {                     // a heavy node's child spoke for the position: its line sits under the five-byte key
                const uint32_t k5 = t4_key5(w[k], w2[k] >> 24);
                idx = 2u * ((h & 2u) ? t4_bucket(t4_hash_b(k5), lb) : t4_bucket(t4_hash_a(k5), lb)) + (h & 1u);
            }
            q0[k] = slots16[4u * idx]; q1[k] = slots16[4u * idx + 1u]; q2[k] = slots16[4u * idx + 2u]; q3[k] = slots16[4u * idx + 3u];
        }
#if defined(__HIP_DEVICE_COMPILE__)
        if (dbg_iters) { const uint64_t now = __builtin_amdgcn_s_memtime(); dbg_iters[2] += now - dbg_iters[7]; dbg_iters[7] = now; }      // debug build: where a batch's cycles go
#endif
        between();
#if defined(__HIP_DEVICE_COMPILE__)
        if (dbg_iters) { const uint64_t now = __builtin_amdgcn_s_memtime(); dbg_iters[3] += now - dbg_iters[7]; dbg_iters[7] = now; }
#endif
#pragma unroll
        for (int k = 0; k < N; k++) avail[k] = avail64[k] > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)avail64[k];
#pragma unroll
        for (int k = 0; k < N; k++) {
            sl[k].key = q0[k].x; sl[k].flags = q0[k].y; sl[k].x = q0[k].z; sl[k].y = q0[k].w;
            sl[k].w = q1[k].x; sl[k].z = q1[k].y; sl[k].cx = q1[k].z; sl[k].cy = q1[k].w;
            sl[k].label[0] = q2[k].x; sl[k].label[1] = q2[k].y; sl[k].label[2] = q2[k].z; sl[k].label[3] = q2[k].w;
            sl[k].cw = q3[k].x; sl[k].ez = q3[k].y; sl[k].el0 = q3[k].z;
        }
        // the slot is the right one if it holds this key -- and, for one child's copy of a branching node, this child's selector byte.
        // Otherwise (a fingerprint collision, or a position deferred without any hot slot agreeing: automata with 1..3-byte needles,
        // the host checker's verify-everything mode) look at the keys of all four candidate slots and fetch the right line.
        bool miss[N], any_miss = false;
#pragma unroll
        for (int k = 0; k < N; k++) {
            const uint32_t b4 = w2[k] >> 24;                 // the byte before the 4-byte suffix
            const bool ok = (sl[k].flags & kSlotOccupied) && sl[k].key == w[k] && !((sl[k].flags & kSlotChildCopy) && ((sl[k].w >> 16) & 0xFFu) != b4);
            miss[k] = look[k] && !ok;
            if (!look[k] || !ok) sl[k].flags = 0;
            any_miss = any_miss || miss[k];
        }
        if (wave_any(any_miss)) {
#pragma unroll
            for (int k = 0; k < N; k++) {
                if (!miss[k]) continue;
                const uint32_t b4 = w2[k] >> 24;
                uint32_t found_idx = kNone;
                for (int c = 0; c < 4; c++) {
                    const u32x4 a0 = slots16[4u * slot_of[c][k]], a1 = slots16[4u * slot_of[c][k] + 1u];
                    const bool ok = (a0.y & kSlotOccupied) && a0.x == w[k] && !((a0.y & kSlotChildCopy) && ((a1.x >> 16) & 0xFFu) != b4);
                    if (ok && found_idx == kNone) found_idx = slot_of[c][k];
                }
                if (found_idx != kNone) {
                    const u32x4 a0 = slots16[4u * found_idx], a1 = slots16[4u * found_idx + 1u], a2 = slots16[4u * found_idx + 2u], a3 = slots16[4u * found_idx + 3u];
                    sl[k].key = a0.x; sl[k].flags = a0.y; sl[k].x = a0.z; sl[k].y = a0.w;
                    sl[k].w = a1.x; sl[k].z = a1.y; sl[k].cx = a1.z; sl[k].cy = a1.w;
                    sl[k].label[0] = a2.x; sl[k].label[1] = a2.y; sl[k].label[2] = a2.z; sl[k].label[3] = a2.w;
                    sl[k].cw = a3.x; sl[k].ez = a3.y; sl[k].el0 = a3.z;
                }
            }
        }
    }
    // ---- step 3: what the slot line settles: the needle ending at the depth-4 node, the single edge and the needle ending at its
    // child.  Only needles longer than that, and branching nodes, go on to the node records.
#pragma unroll
    for (int k = 0; k < N; k++) {
        best_state[k] = 0; best_vlen[k] = 0; depth[k] = 4; go[k] = false; node[k] = kNone;
        rec[k] = SfNode{0, 0, 0, 0, {0, 0, 0, 0}}; have_rec[k] = false;
        if (!(sl[k].flags & kSlotOccupied) || avail[k] < 4) continue;       // the 4-byte suffix does not fit into the haystack
        if (sl[k].x) { best_state[k] = sl[k].x; best_vlen[k] = sl[k].y; }
        const uint32_t kind = sl[k].w & 0xFFFFu;
        if (kind == 0 || avail[k] <= 4) continue;
        if (kind == 1) {
            const uint32_t skip = sl[k].w >> 24;
            if (((sl[k].w >> 16) & 0xFFu) != (w2[k] >> 24)) continue;
            if (5u + skip > avail[k]) continue;
            if (skip && !label_match(t16[k], sl[k].label, skip)) continue;
            depth[k] = 5u + skip;
            if (sl[k].cx) { best_state[k] = sl[k].cx; best_vlen[k] = sl[k].cy; }
            if ((sl[k].cw & 0xFFFFu) != 0 && depth[k] < avail[k]) { go[k] = true; node[k] = sl[k].z; }     // the walk continues at the child
        } else {                                                                                           // branching: the walk starts at the depth-4 node itself
            go[k] = true; node[k] = sl[k].z; have_rec[k] = true;
            rec[k].x = sl[k].x; rec[k].y = sl[k].y; rec[k].z = sl[k].ez; rec[k].w = kind; rec[k].label[0] = sl[k].el0;
            rec[k].label[1] = sl[k].label[1]; rec[k].label[2] = sl[k].label[2]; rec[k].label[3] = sl[k].label[3];      // (more than 4 children: which selector bytes exist)
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if (dbg_iters) { const uint64_t now = __builtin_amdgcn_s_memtime(); dbg_iters[4] += now - dbg_iters[7]; dbg_iters[7] = now; }
#endif
}

constexpr uint32_t kSelUnknown = 0x100u;
// byte `idx` (0..15, text order) of 16 haystack bytes held as four little-endian dwords
AM_HD uint32_t byte_of_16(const uint32_t (&t)[4], uint32_t idx)
{
    const uint32_t t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];      // (unconditional loads with constant indices: the array stays in registers)
    const uint32_t d = idx < 8u ? (idx < 4u ? t0 : t1) : (idx < 12u ? t2 : t3);
    return (d >> (8u * (idx & 3u))) & 0xFFu;
}

// ---- step 4: walk the compressed trie backwards along the haystack to the deepest needle end
template <bool IC, int N>
AM_HD void sf_resolve_walk(const SfView& s, const uint8_t* text, const uint64_t (&gpos)[N], const uint32_t (&avail)[N], const uint32_t (&w2)[N],
                           bool (&go)[N], const uint32_t (&node)[N], SfNode (&rec)[N], const bool (&have_rec)[N], uint32_t (&depth)[N],
                           uint32_t (&best_state)[N], uint32_t (&best_vlen)[N], uint64_t* dbg_iters = nullptr, uint32_t max_iters = 0xFFFFFFFFu,
                           const uint32_t (*t16)[4] = nullptr,      // t16 (optional): the head's 16 folded bytes before the 4-byte suffix = what a step at depth 4 compares with
                           uint32_t* sel_io = nullptr)              // (optional, N values, in and out) the next selector byte where a previous step already saw it, else kSelUnknown
{
    const u32x4* nodes16 = reinterpret_cast<const u32x4*>(s.nodes);      // 2 x 16 B per node
    const u32x4* edges16 = reinterpret_cast<const u32x4*>(s.edges);      // 4 x 16 B per edge
    // The selector byte of a step (the haystack byte at gpos - depth) comes from registers whenever something already loaded it: the 8 bytes of the
    // suffix word (depth < 8), the head's 16 bytes (depth 5..20), or the 16 bytes the PREVIOUS step compared its label with (they end right before
    // that step's selector, and the label is at most 16 long: only a 16-byte label leaves the next selector outside).  A load of its own would be a
    // dependent trip in front of the edge line's.
    uint32_t sel[N];
#pragma unroll
    for (int k = 0; k < N; k++) sel[k] = sel_io ? sel_io[k] : kSelUnknown;
    bool any_load = false;
#pragma unroll
    for (int k = 0; k < N; k++) any_load = any_load || (go[k] && !have_rec[k]);
    if (wave_any(any_load)) {                        // walks that go on at the single edge's child need its record
        u32x4 r0[N], r1[N];
#pragma unroll
        for (int k = 0; k < N; k++) { const uint32_t id = go[k] && !have_rec[k] ? node[k] : 0u; r0[k] = nodes16[2u * id]; r1[k] = nodes16[2u * id + 1u]; }
#pragma unroll
        for (int k = 0; k < N; k++) if (go[k] && !have_rec[k]) node_from_raw(r0[k], r1[k], rec[k]);
    }
    // max_iters: the caller takes the walk over again after that many steps (go[] still set, rec / depth / best_* as they stand)
    for (uint32_t iter = 0; iter < max_iters; iter++) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < N; k++) any = any || go[k];
        if (!wave_any(any)) break;
        if (dbg_iters) {
            dbg_iters[0]++;
#if defined(__HIP_DEVICE_COMPILE__)
            dbg_iters[1] += (uint64_t)__popcll(__ballot(go[0]));
#endif
        }
        // One step of the walk = ONE round of loads for all lanes: a single-edge node has its edge inline (selector, skip, label) and
        // needs its child's record; a branching node needs the chosen edge's 64-byte line, which carries the child's record; the 16
        // haystack bytes the label is compared with depend only on the depth.  A node with more than 4 children reads the line of its
        // row that the selector byte names, and sees there whether that line is its own.
        uint32_t which[N], next[N], skip[N], label[N][4], owner[N];
#pragma unroll
        for (int k = 0; k < N; k++) {
            which[k] = kNone; next[k] = kNone; skip[k] = 0; owner[k] = 0;
            label[k][0] = rec[k].label[0]; label[k][1] = rec[k].label[1]; label[k][2] = rec[k].label[2]; label[k][3] = rec[k].label[3];
            if (!go[k]) continue;
            const uint32_t n_edges = rec[k].w & 0xFFFFu;
            uint32_t b;
            if (sel[k] != kSelUnknown) b = sel[k];
            else if (depth[k] < 8) b = (w2[k] >> (8u * (7u - depth[k]))) & 0xFFu;
            else if (t16 && depth[k] <= 20) b = byte_of_16(t16[k], 20u - depth[k]);
            else { b = text[gpos[k] - depth[k]]; if (IC) b = fold_byte(b); }
            if (n_edges == 1) {
                if (((rec[k].w >> 16) & 0xFFu) == b) { next[k] = rec[k].z; skip[k] = rec[k].w >> 24; }
            } else if (n_edges <= 4) {
                for (uint32_t i = 0; i < n_edges; i++) if (((rec[k].label[0] >> (8u * i)) & 0xFFu) == b) which[k] = rec[k].z + i;
            } else {
                const uint32_t f = b >= 192u ? b - 192u : b >= 96u ? b - 96u : b;                                  // b % 96
                const uint32_t fw = f < 32u ? rec[k].label[1] : f < 64u ? rec[k].label[2] : rec[k].label[3];
                if ((fw >> (f & 31u)) & 1u) { which[k] = rec[k].label[0] + b; owner[k] = rec[k].z + 1u; }
            }
        }
        SfNode child[N];
        uint32_t t[N][4];
        {
            u32x4 e0[N], e1[N], c0[N], c1[N];
#pragma unroll
            for (int k = 0; k < N; k++) {
                e0[k] = u32x4{0, 0, 0, 0}; e1[k] = e0[k]; c0[k] = e0[k]; c1[k] = e0[k];
                t[k][0] = t[k][1] = t[k][2] = t[k][3] = 0;
                if (!go[k]) continue;
                if (which[k] != kNone) {                                   // the edge's line: edge + the child's record
                    e0[k] = edges16[4u * which[k]]; e1[k] = edges16[4u * which[k] + 1u]; c0[k] = edges16[4u * which[k] + 2u]; c1[k] = edges16[4u * which[k] + 3u];
                } else if (next[k] != kNone) { c0[k] = nodes16[2u * next[k]]; c1[k] = nodes16[2u * next[k] + 1u]; }      // single edge: the child's record
                else continue;
                if (t16 && depth[k] == 4) { t[k][0] = t16[k][0]; t[k][1] = t16[k][1]; t[k][2] = t16[k][2]; t[k][3] = t16[k][3]; }
                else {
                    load_text16(text, gpos[k] - depth[k], t[k]);       // the 16 bytes before the selector byte
                    if (IC) { t[k][0] = fold_dword(t[k][0]); t[k][1] = fold_dword(t[k][1]); t[k][2] = fold_dword(t[k][2]); t[k][3] = fold_dword(t[k][3]); }
                }
            }
#pragma unroll
            for (int k = 0; k < N; k++) {
                if (which[k] != kNone && owner[k] && e0[k].w != owner[k]) which[k] = kNone;      // a row line of another node (or an empty one): no such edge
                if (which[k] != kNone) {
                    next[k] = e0[k].y; skip[k] = e0[k].z;                       // SfEdge {byte, child, skip, pad, label[4], to}
                    label[k][0] = e1[k].x; label[k][1] = e1[k].y; label[k][2] = e1[k].z; label[k][3] = e1[k].w;
                }
                node_from_raw(c0[k], c1[k], child[k]);
                if (next[k] == kNone || (uint64_t)depth[k] + 1u + skip[k] > avail[k]) { go[k] = false; next[k] = kNone; }
            }
        }
        // compare the label, advance
#pragma unroll
        for (int k = 0; k < N; k++) {
            if (!go[k]) continue;
            if (skip[k] && !label_match(t[k], label[k], skip[k])) { go[k] = false; continue; }
            sel[k] = skip[k] < 16u ? byte_of_16(t[k], 15u - skip[k]) : kSelUnknown;      // t = the 16 bytes before this step's selector
            depth[k] += 1u + skip[k];
            rec[k] = child[k];
            if (rec[k].x) { best_state[k] = rec[k].x; best_vlen[k] = rec[k].y; }
            go[k] = depth[k] < avail[k] && (rec[k].w & 0xFFFFu) != 0;
        }
    }
    if (sel_io) {
#pragma unroll
        for (int k = 0; k < N; k++) sel_io[k] = sel[k];
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if (dbg_iters) { const uint64_t now = __builtin_amdgcn_s_memtime(); dbg_iters[5] += now - dbg_iters[7]; dbg_iters[7] = now; }
#endif
}

// ---- step 5: needles of 1..3 bytes (only if nothing longer ends here)
template <int N>
AM_HD void sf_resolve_short(const SfView& s, const bool (&valid)[N], const uint32_t (&avail)[N], const uint32_t (&w)[N], uint32_t (&best_state)[N], uint32_t (&best_vlen)[N])
{
#pragma unroll
    for (int k = 0; k < N; k++) {
        if (valid[k] && !best_state[k] && (s.tiers & 7u)) {
            uint32_t short_node = kNone;
            for (uint32_t t = 3; t >= 1; t--) {
                if ((s.tiers & (1u << (t - 1))) && avail[k] >= t) {
                    short_node = tier_lookup(s.tier[t - 1], s.tier_log2_cap[t - 1], w[k] >> (8u * (4u - t)));
                    if (short_node != kNone) break;
                }
            }
            if (short_node != kNone) { best_state[k] = s.nodes[short_node].x; best_vlen[k] = s.nodes[short_node].y; }
        }
    }
}

// SHORT = false promises that the automaton has no needle (variant) shorter than 4 bytes (s.tiers & 7 == 0).
// `hint` = which of the four candidate slots the probe saw agree (sf_probe_n); any value is correct, the right one saves loads.
// `between` runs after the slot-line loads have been issued and before they are used (the caller computes avail64 there: haystack
// index -> offsets, two dependent loads of its own, overlapping haystack bytes -> slot line).  avail64 is not read before that.
template <bool IC, int N, bool SHORT = true, class Between = SfNoHook>
AM_HD void sf_resolve_n(const SfView& s, const uint8_t* text, const uint64_t (&gpos)[N], const uint64_t (&avail64)[N], const bool (&valid)[N],
                        const uint32_t (&hint)[N], bool (&found)[N], uint32_t (&state)[N], uint32_t (&vlen)[N], Between between = Between(),
                        uint64_t* dbg_iters = nullptr, uint32_t dbg_ablate = 0)
{
    uint32_t w[N], w2[N], avail[N], best_state[N], best_vlen[N], depth[N], node[N], t16[N][4];
    bool go[N], have_rec[N];
    SfNode rec[N];
    sf_resolve_head<IC, N>(s, text, gpos, avail64, valid, hint, between, w, w2, avail, best_state, best_vlen, depth, go, node, rec, have_rec, t16, dbg_iters);
    if (dbg_ablate != 11) sf_resolve_walk<IC, N>(s, text, gpos, avail, w2, go, node, rec, have_rec, depth, best_state, best_vlen, dbg_iters, 0xFFFFFFFFu, t16);
    if (SHORT) sf_resolve_short<N>(s, valid, avail, w, best_state, best_vlen);
#pragma unroll
    for (int k = 0; k < N; k++) {
        found[k] = valid[k] && best_state[k] != 0;
        state[k] = best_state[k] - 1u; vlen[k] = best_vlen[k];
    }
}

template <bool IC>
AM_HD bool sf_resolve(const SfView& s, const uint8_t* text, uint64_t gpos, uint64_t avail, uint32_t& state, uint32_t& vlen, uint32_t hint = 0)
{
    const uint64_t g[1] = {gpos}, a[1] = {avail};
    const bool v[1] = {true};
    bool f[1]; uint32_t st[1], vl[1];
    const uint32_t hi[1] = {hint};
    sf_resolve_n<IC, 1>(s, text, g, a, v, hi, f, st, vl);
    state = st[0]; vlen = vl[0];
    return f[0];
}

// probe + resolve for one position (host checker, and the reference for what the kernel computes)
template <bool IC>
AM_HD bool sf_verify(const SfView& s, const uint8_t* text, uint64_t gpos, uint64_t avail, uint32_t& state, uint32_t& vlen)
{
    uint32_t w, w2;
    load_suffix8(text, gpos, w, w2);
    if (IC) { w = fold_dword(w); w2 = fold_dword(w2); }
    const uint32_t wa[1] = {w}, nba[1] = {(w2 >> 24) | (((w2 >> 16) & 0xFFu) << 8) | (((w2 >> 8) & 0xFFu) << 16)};
    const uint64_t a[1] = {avail};
    const bool v[1] = {true};
    bool defer[1]; uint32_t hint[1];
    sf_probe_n<1>(s, wa, nba, a, v, defer, hint);
    if (!defer[0]) return false;
    return sf_resolve<IC>(s, text, gpos, avail, state, vlen, hint[0]);
}

// Bloom test of one window for every active tier; returns true if any tier may match.
// `bloom` may point to LDS (device) or to the image (host checker); `masks` = the kernel's LDS copy of the mask table or
// null.  The kernel's hot loop uses a batched form of the tier-4 test (all LDS reads of a lane in flight together) and
// calls sf_filter_short only for automata that contain needles shorter than 4 bytes.
AM_HD bool sf_filter_short(const uint32_t* bloom, uint32_t log2_words, uint32_t tiers, uint32_t w, const uint32_t* masks = nullptr)
{
    bool hit = false;
    for (uint32_t t = 1; t <= 3; t++) {
        if (tiers & (1u << (t - 1))) {
            const uint32_t h = bloom_hash(w >> (8u * (4u - t)), t);
            hit = hit || bloom_hit(bloom[bloom_word(h, log2_words)], h, masks);
        }
    }
    return hit;
}
AM_HD bool sf_filter_window(const uint32_t* bloom, uint32_t log2_words, uint32_t tiers, uint32_t w)
{
    const uint32_t h = bloom_hash(w, 4);
    bool hit = (tiers & 8u) && bloom_hit(bloom[bloom_word(h, log2_words)], h);
    if (tiers & 7u) hit = hit || sf_filter_short(bloom, log2_words, tiers, w);
    return hit;
}

// ------------------------------------------------------------------ AC: the reference's state machine

AM_HD uint32_t lower_cp(const AcView& a, uint32_t cp)
{
    // Utf8.hs:145-151 lowerCodePoint (ASCII fast path, else the simple-lowercase table)
    if (cp < 128u) return fold_byte(cp);
    return cp < a.n_lower ? (uint32_t)((int32_t)cp + a.lower[cp]) : cp;
}

// slot of (state, cp) in the AC goto hash (flattener and kernels must agree)
AM_HD uint32_t ac_goto_slot(uint32_t state, uint32_t cp, uint32_t log2_cap)
{
    uint32_t h = state * 0x9E3779B1u ^ cp * 0x85EBCA6Bu;
    h ^= h >> 15;
    return (h * 0x2C1B3C6Du) >> (32u - log2_cap);
}

// Automaton.hs:482-520 followCodePoint / lookupTransition / lookupRootAsciiTransition.
// Returns true iff a goto edge was taken (then collectMatches applies to the new state).
// The reference scans the state's edge list linearly (:489-510); the image answers the same question
// "(state, cp) -> goto target or none" with one hash probe, and the wildcard entry's target is fail[state].
AM_HD bool ac_step(const AcView& a, uint32_t& state, uint32_t cp)
{
    const uint32_t mask = (1u << a.goto_log2_cap) - 1u;
    for (;;) {
        if (state == 0 && cp < 128u) {
            const uint64_t t = a.root_ascii[cp];
            if (t & kWildcard) return false;
            state = (uint32_t)(t >> 32);
            return true;
        }
        uint32_t i = ac_goto_slot(state, cp, a.goto_log2_cap);
        for (;;) {
            const u32x4 e = load16(a.goto_tab + i);
            if (e.w == 0u) break;                                       // empty slot: no such edge
            if (e.x == state && e.y == cp) { state = e.z; return true; }
            i = (i + 1u) & mask;
        }
        if (state == 0) return false;                                   // wildcard at the root: next input (:496-497)
        state = a.fail[state];
    }
}

// Automata with the empty needle, dense part (see am_flatten.cpp, SF section): is byte g (hs <= g < he, the bounds of its
// haystack) the LAST byte of a code point whose lowered value is the first code point of some needle?  There the reference
// is not at the root after the code point and folds at least the root's values.
AM_HD bool ends_first_code_point(const AcView& a, bool ic, const uint8_t* text, uint64_t hs, uint64_t he, uint64_t g)
{
    const uint32_t b0 = text[g];
    if (b0 < 0x80u) return !(a.root_ascii[ic ? fold_byte(b0) : b0] & kWildcard);
    if ((b0 & 0xC0u) != 0x80u) return false;                                  // a lead byte ends nothing
    if (g + 1 < he && (text[g + 1] & 0xC0u) == 0x80u) return false;         // the code point goes on
    uint32_t k = 1;
    while (k <= 3 && g >= hs + k && (text[g - k] & 0xC0u) == 0x80u) k++;
    if (k > 3 || g < hs + k) return false;                                   // no lead byte inside the haystack
    const uint32_t lead = text[g - k];
    const uint32_t units = lead < 0xc0u ? 1u : lead < 0xe0u ? 2u : lead < 0xf0u ? 3u : 4u;
    if (units != k + 1u) return false;
    const uint32_t c1 = text[g - k + 1], c2 = units > 2 ? text[g - k + 2] : 0u, c3 = units > 3 ? text[g - k + 3] : 0u;
    uint32_t cp = units == 2 ? ((lead & 0x1fu) << 6) | (c1 & 0x3fu)
                : units == 3 ? ((lead & 0xfu) << 12) | ((c1 & 0x3fu) << 6) | (c2 & 0x3fu)
                             : ((lead & 0x7u) << 18) | ((c1 & 0x3fu) << 12) | ((c2 & 0x3fu) << 6) | (c3 & 0x3fu);
    if (ic) cp = lower_cp(a, cp);
    if (cp < 128u) return !(a.root_ascii[cp] & kWildcard);                   // a non-ASCII code point may lower to ASCII (U+212A -> k)
    const uint32_t mask = (1u << a.goto_log2_cap) - 1u;
    for (uint32_t i = ac_goto_slot(0u, cp, a.goto_log2_cap);; i = (i + 1u) & mask) {
        const u32x4 e = load16(a.goto_tab + i);
        if (e.w == 0u) return false;
        if (e.x == 0u && e.y == cp) return true;
    }
}

// One lane's unit of the general kernel: bytes [unit*chunk, (unit+1)*chunk) of the batch.  The lane
// owns every match whose LAST byte lies in its chunk; it warms the state up from root over
// >= max_needle_cps code points before the chunk (the AC state depends only on that much history),
// clipped to the haystack start.  emit(haystack, end_pos_in_haystack, canon_state, vlen).
template <bool IC, class Emit>
AM_HD void ac_scan_unit(const AcView& a, const BatchView& b, uint64_t unit, Emit& emit)
{
    const uint64_t cs = unit * a.chunk;
    if (cs >= b.total) return;
    const uint64_t ce = (cs + a.chunk < b.total) ? cs + a.chunk : b.total;
    uint32_t h = find_haystack(b, cs);
    uint64_t hs = b.offsets[h], he = b.offsets[h + 1];
    const uint64_t warm = 4ull * (a.max_needle_cps ? a.max_needle_cps : 1u) + 4ull;
    uint64_t offset = (cs - hs > warm) ? cs - warm : hs;
    if (offset > hs) while (offset < cs && (b.text[offset] & 0xC0u) == 0x80u) offset++;   // snap to a code point start
    uint32_t state = 0;
    while (offset < ce) {
        if (offset >= he) {                       // next non-empty haystack starts here: fresh run
            do { h++; hs = he; he = b.offsets[h + 1]; } while (he == hs);
            state = 0;
        }
        // Utf8.hs:337-350 decodeN, reads guarded by the haystack end
        const uint32_t cu0 = b.text[offset];
        const uint32_t units = cu0 < 0xc0u ? 1u : cu0 < 0xe0u ? 2u : cu0 < 0xf0u ? 3u : 4u;
        uint64_t nend = offset + units;
        if (nend > he) nend = he;
        if (nend > ce) break;                     // its last byte belongs to the next unit
        uint32_t cp = cu0;
        if (units > 1) {
            const uint32_t cu1 = offset + 1 < he ? b.text[offset + 1] : 0u;
            const uint32_t cu2 = (units > 2 && offset + 2 < he) ? b.text[offset + 2] : 0u;
            const uint32_t cu3 = (units > 3 && offset + 3 < he) ? b.text[offset + 3] : 0u;
            cp = units == 2 ? ((cu0 & 0x1fu) << 6) | (cu1 & 0x3fu)
               : units == 3 ? ((cu0 & 0xfu) << 12) | ((cu1 & 0x3fu) << 6) | (cu2 & 0x3fu)
                            : ((cu0 & 0x7u) << 18) | ((cu1 & 0x3fu) << 12) | ((cu2 & 0x3fu) << 6) | (cu3 & 0x3fu);
        }
        if (IC) cp = lower_cp(a, cp);
        const bool collected = ac_step(a, state, cp);
        offset = nend;
        if (collected && nend > cs) {
            const uint32_t v = a.vlen[state];
            if (v) emit((uint32_t)h, nend - hs, a.canon[state], v);
        }
    }
}

// slot of (state, byte) in the hash of the edges on rare bytes (flattener and kernels must agree)
AM_HD uint32_t dfa_rare_slot(uint32_t state, uint32_t byte, uint32_t log2_cap)
{
    uint32_t h = state * 0x9E3779B1u ^ byte * 0x85EBCA6Bu;
    h ^= h >> 15;
    return (h * 0x2C1B3C6Du) >> (32u - log2_cap);
}
// delta(state, byte) for a byte without a column: the state's own edge on it, else the same question at its fallback, the root answering "root"
// (Automaton.hs:489-510 as it stands; the dense rows are this loop precomputed for the common bytes).  Returns the transition entry (next state | end bits).
AM_HD uint32_t dfa_rare_step(const DfaView& d, uint32_t state, uint32_t byte)
{
    const uint32_t mask = (1u << d.rare_log2_cap) - 1u;
    for (;;) {
        for (uint32_t i = dfa_rare_slot(state, byte, d.rare_log2_cap);; i = (i + 1u) & mask) {
            const u32x4 e = load16(d.rare + i);
            if (e.w == 0u) break;
            if (e.x == state && e.y == byte) return e.z;
        }
        if (state == 0u) return 0u;
        state = d.fail[state];
    }
}

// delta(state, class) for a byte with a column.  A ROW state has its dense row.  A RECORD state -- one whose row would differ in at most two entries from the row of
// R, the nearest row state on its chain of fallbacks; the single-entry ones numbered along their paths so that the records of a word's tail share cache lines -- has 8
// or 16 bytes: where its one or two classes lead, and R for every other class (delta(x, c) = delta(R, c) there).
AM_HD uint32_t dfa_common_step(const DfaView& d, uint32_t state, uint32_t cl)
{
    if (cl == 0u) return 0u;                                                        // a byte no needle contains: the root, nothing ends (the image check holds every row to it)
    if (state >= d.n_rows) {                                                        // a record state: one of its entries answers, else the row state it leans on
        if (state < d.n_rows + d.n_single) {
            const u32x2 r = d.chain[state - d.n_rows];
            if ((r.y >> 24) == cl) return r.x;
            state = r.y & 0xFFFFFFu;
        } else {
            const u32x4 q = load16(d.chain2 + (state - d.n_rows - d.n_single));
            if ((q.y >> 24) == cl) return q.x;
            if ((q.w >> 24) == cl) return q.z;
            state = q.y & 0xFFFFFFu;
        }
    }
    if (cl <= (1u << d.hot_log2)) return d.hot[((uint64_t)state << d.hot_log2) + cl - 1u];
    return d.next[((uint64_t)state << d.log2_classes) + cl];
}

// One lane's unit of the DFA kernel, the plain form (the host image interpreter runs this; k_dfa in am_dfa.hip is the same walk with wide loads): bytes
// [unit*chunk, (unit+1)*chunk) of the batch; the lane owns every match whose LAST byte lies there and warms the state up from the root over `warm` bytes
// (any byte offset will do: a needle starts with no continuation byte, so a walk that starts inside a code point stays at the root until the next one).
// emit(haystack, end_pos_in_haystack, canon_state, vlen) -- Automaton.hs:482-520 with every fallback step folded into the table.
template <class Emit>
AM_HD void dfa_scan_unit(const DfaView& d, const BatchView& b, uint64_t unit, Emit& emit)
{
    const uint64_t cs = unit * d.chunk;
    if (cs >= b.total) return;
    const uint64_t ce = (cs + d.chunk < b.total) ? cs + d.chunk : b.total;
    uint32_t h = find_haystack(b, cs);
    uint64_t hs = b.offsets[h], he = b.offsets[h + 1];
    uint64_t offset = (cs - hs > d.warm) ? cs - d.warm : hs;
    uint32_t state = 0;
    while (offset < ce) {
        if (offset >= he) { do { h++; hs = he; he = b.offsets[h + 1]; } while (he == hs); state = 0; }
        uint32_t byte = b.text[offset];
        const uint32_t cl = d.cls[byte];
        if (cl == kDfaRare && d.ic && byte - 0x41u < 26u) byte += 0x20u;    // (the edges of an IgnoreCase automaton carry the folded letter)
        const uint32_t e = cl == kDfaRare ? dfa_rare_step(d, state, byte) : dfa_common_step(d, state, cl);
        state = e & kDfaStateMask;
        offset++;
        if ((e >> kDfaEndShift) && offset > cs) { const u32x2 o = d.out[state]; emit((uint32_t)h, offset - hs, o.x - 1u, o.y); }
    }
}

}  // namespace am
