// am_dfa.hip -- k_dfa: the byte-level Aho-Corasick automaton with every transition resolved (ImageHeader::off_dfa_next, built by am_flatten.cpp for
// dictionaries that meet match-dense text), walked one lane per stretch of the batch.  Reference semantics: Automaton.hs:482-520 (followCodePoint /
// collectMatches) -- the fallback loop is folded into the table, so a step is ONE dependent load: next = table[state << log2_classes | class(byte)] for the states that
// have a dense row; the others -- states whose row would differ in one or two entries from the row of R, the nearest row state on their chain of fallbacks -- keep an 8- or
// 16-byte RECORD {where those classes lead, R}, and a byte the record does not name is answered by R's row: at most two trips (image layout: am_image.h, DfaView; version 17).
//
// Why a second scan kernel: k_sf is a filter.  On natural-language text against a 100k-word dictionary four positions in ten pass its LDS filter and a needle
// ends every 6.6 bytes; its resolve phase then costs ~19 divergent 16-byte loads and ~22 VALU instructions per deferred position and the kernel runs at the
// issue limit of the CUs' address units and vector ALUs (profiles/r05_pmc_natural_units.md).  A table walk costs one 4-byte load and a few instructions per BYTE
// whatever the text is -- slower than k_sf where matches are rare (it cannot skip anything), several times faster where they are dense.
//
// What bounds it (round 6, profiles/r06_pmc_dfa.md, LABNOTES R6.1 / R6.8): the wavefronts wait 85-90 % of their time, and not for latency -- half the wavefronts per CU give 86 % of
// the rate.  Every lane-load is an L2 request (2 048 lanes per CU touch 2 048 different lines between two visits of one lane: the 32-KiB L1 holds nothing), and the L2s take about
// 300 G requests/s from full wavefronts of dependent random loads (16 channels x 8 XCDs x a line per clock; tools/microbench/l2_curve.hip), 190 G/s from this kernel's loads with a
// third of their lanes active; whether a request hits or misses its XCD's 4-MiB L2 matters little next to that (a fit over two layouts: 1 / 200 G/s against 1 / 135 G/s).  So the
// kernel is written to the currency "L2 requests per byte": the byte class, the hottest rows and the hottest records in LDS, a record in ONE 16-byte load (which brings the next
// record of its path along: a lane that follows the path asks for nothing), a byte no needle contains answered without any load (it leads to the root from everywhere), text asked
// for 64 bytes at a time, and the first 16 columns of every row a second time in a table of their own where two rows share a line (am_flatten.cpp).
//
// Work split: unit u = bytes [u * chunk, (u + 1) * chunk) of the concatenated batch, one lane each; the lane owns the matches whose LAST byte lies in its unit and
// warms its state up from the root over the `warm` bytes before it (clipped to the haystack start; a haystack boundary inside the unit resets the state).
// A wavefront takes GROUPS of 64 consecutive units; inside a group every position is a 32-bit offset from the group's start minus the warm-up (one 64-bit base in scalar
// registers: the text loads are saddr + 32-bit lane offset, and so are the table loads -- the DFA section is < 4 GiB).
//   count / any  one launch; unit_counts[u] = the unit's records (what the exclusive scan turns into the records' final places).
//   records      ONE walk as well (kModeTokens): a lane knows the running number `seq` of each of its matches, so it drops an 8-byte TOKEN
//                {DFA state | group ordinal, offset in the unit | seq | lane} into its wavefront's current superblock of the pool (slot = one LDS atomic; a superblock =
//                4096 tokens, one device atomic each; order inside does not matter), and after the scan k_dfa_place puts the token of (unit, seq) at unit_offsets[unit] +
//                seq as the record it stands for: position order without a sort and without walking the text a second time.  A pool that turns out too small only costs
//                the tokens (the counts stay exact): the host repeats the call with the size the kernel reports, as for k_sf's record blocks.
//                (kModeEmit is the second pass of the plain count -> scan -> emit protocol, kept for units beyond 8192 bytes and for batches whose tokens the device
//                could not hold.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>

#include "am_bounds.h"
#include "am_config.h"
#include "am_device.h"
#include "am_wave.h"

AM_BOUNDS_TU("am_dfa.hip")

namespace am {
namespace dev {

namespace {

constexpr uint32_t kWave = 64;
constexpr int kModeTokens = 16;               // template value only: records in one walk, tokens into the pool
constexpr uint32_t kDfaSuper = 4096;          // tokens per superblock (32 KiB)
constexpr uint32_t kDfaSuperReserve = 1024;   // free slots a wavefront makes sure of before 64 lanes take (at most) 16 steps
// a token: w0 = DFA state (28 bits) | the group's ordinal in its superblock << 28 (a superblock holds tokens of at most 16 groups of its wavefront),
//          w1 = offset of the match's last byte in its unit (13 bits) | seq << 13 (13 bits) | lane << 26 -- hence units of at most 8192 bytes on this route
constexpr uint32_t kTokPosBits = 13, kTokOrdShift = 28, kTokMaxOrd = 16, kTokMaxChunk = 1u << kTokPosBits;
constexpr uint32_t kLdsLog2Cols = 5;          // LDS holds columns 1 .. 32 of the first rows (the classes are numbered by the dictionary's use of them: 31 are 98 % of natural text)

typedef uint32_t u32x2_v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(8)));      // a record read: 16 bytes from an 8-byte boundary
typedef __attribute__((address_space(3))) uint8_t lds_u8_t;
__device__ __forceinline__ uint32_t lds_read_u8(uint32_t byte_addr) { return *reinterpret_cast<const lds_u8_t*>((uintptr_t)byte_addr); }

// hot table and chain records as byte offsets from `next` (the flattener puts them behind the rows; launch_dfa_tw checks that the section spans < 4 GiB)
// lds_rec1 / lds_rec2: LDS addresses of the first lds_n1 single-entry and lds_n2 two-entry records (the flattener numbers both kinds by weight: the first are the hottest)
struct DfaDev { const uint8_t* base; uint32_t off_hot, off_chain, off_chain2, lds_rec1, lds_rec2, lds_n1, lds_n2; };
typedef __attribute__((address_space(3))) u32x4_u lds_u32x4_u_t;

template <int MODE>
struct DfaLane {
    const DfaView& d; const BatchView& b; const ScanOut& o;
    uint32_t* wv;                             // token mode: this wavefront's words in LDS: [0] tokens in its superblock, [1] the superblock's id, [2] pool exhausted,
                                              // [3] the ordinal (k) of the superblock's first group, [4] the current group, [5] its ordinal
    uint32_t nrec = 0;
    // count mode: the unit's values, and those of the haystack the lane is in (added with one atomic when it leaves it), in 32 bits: lists of fewer than 15 values (the end
    // bits of the entry) add up here -- at most 14 per byte of a unit --, a longer list goes to the 64-bit sums in memory at once
    uint32_t nval = 0, run_hay = kNone, run_val = 0;
    Record* out = nullptr;                    // emit mode: where this unit's next record goes
    uint32_t ord = 0, sb = kNone;             // token mode: what the block's tokens carry / go to (read from LDS after the reserve at the top of a block)
    __device__ __forceinline__ void flush()
    {
        if (MODE == kModeCount && o.hay_counts && run_hay != kNone && run_val) atomicAdd(reinterpret_cast<unsigned long long*>(o.hay_counts + run_hay), (unsigned long long)run_val);
        run_val = 0;
    }
    // in_unit = offset of the match's last byte in the unit; end = the entry's end bits: the length of the needle-end list (1..14), kDfaEndLookUp = longer (out[] knows)
    __device__ __forceinline__ void found(uint32_t hay, uint32_t in_unit, uint64_t end_pos, uint32_t state, uint32_t end)
    {
        if (MODE == kModeAny) { o.flags[hay] = 1; return; }
        if (MODE == kModeTokens) {
            if (sb != kNone) {
                const uint32_t slot = atomicAdd(&wv[0], 1u);          // (LDS; < kDfaSuper by the reserve made at the top of the block.  A rank from the ballot of the finders + one
                                                                      // write by the first of them instead of the atomic: measured, no faster -- LABNOTES R6.8)
                AM_BOUNDS(slot < kDfaSuper && sb < o.n_blocks && ord < kTokMaxOrd && in_unit < kTokMaxChunk && nrec < kTokMaxChunk && state < d.n_states);
                u32x2_v t; t.x = state | (ord << kTokOrdShift); t.y = in_unit | (nrec << kTokPosBits) | (lane_id() << 26);
                reinterpret_cast<u32x2_v*>(o.pool)[(uint64_t)sb * kDfaSuper + slot] = t;      // (the DFA state: k_dfa_place looks the reference state up, off the walk's dependent chain)
            }
            nrec++;
            return;
        }
        if (MODE == kModeCount) {
            nrec++;
            if (end < kDfaEndLookUp) {                                 // (a count needs no load of its own unless a position reports 15 values or more)
                nval += end;
                if (o.hay_counts) { if (hay != run_hay) { flush(); run_hay = hay; } run_val += end; }
            } else {
                const unsigned long long vl = d.out[state].y;
                atomicAdd(reinterpret_cast<unsigned long long*>(o.total_values), vl);
                if (o.hay_counts) atomicAdd(reinterpret_cast<unsigned long long*>(o.hay_counts + hay), vl);
            }
        } else {
            const u32x2 e = d.out[state];
            AM_BOUNDS(e.x != 0u && hay < b.n_hay && end_pos != 0u && end_pos <= b.offsets[hay + 1] - b.offsets[hay]);
            out[nrec++] = Record{end_pos, hay, e.x - 1u};
        }
    }
};

// delta(state, class) for a byte with a column (dfa_common_step in am_image.h is the plain form): a record state answers with a child or hands the question to the state
// it falls back to -- ONE 8- or 16-byte load per record; a byte no needle contains (class 0) leads to the root; the first rows' first columns sit in LDS; the first columns of every row in
// the hot table; one global load serves the hot and the cold case (both tables lie behind `next`).
// (pf: the 16 bytes read at a single-entry record hold the next record of its path as well; a lane that follows the path finds it here and asks for nothing)
struct DfaAhead { uint32_t state = kNone, x = 0, y = 0; };
__device__ __forceinline__ uint32_t dfa_step(const DfaView& d, const DfaDev& v, uint32_t lds_rows_addr, uint32_t hot_rows, uint32_t state, uint32_t cl, DfaAhead& pf)
{
    if (cl == 0u) return 0u;
    AM_BOUNDS(state < d.n_states && cl < (1u << d.log2_classes));
    if (state >= d.n_rows) {                                // a record state (at most two entries): an entry answers, else the row state it leans on
        // ONE 16-byte load for both kinds (a divergent if / else would be two dependent trips per turn): a single-child record is the first half of what is read
        // (the flattener pads the table by a record), a two-children record all of it
        const bool two = state >= d.n_rows + d.n_single;
        const uint32_t idx = two ? state - d.n_rows - d.n_single : state - d.n_rows;
        const uint32_t at = two ? v.off_chain2 + (idx << 4) : v.off_chain + (idx << 3);
        u32x4_u q;
        if (idx < (two ? v.lds_n2 : v.lds_n1)) q = *reinterpret_cast<const lds_u32x4_u_t*>((uintptr_t)(two ? v.lds_rec2 + (idx << 4) : v.lds_rec1 + (idx << 3)));      // the hottest records: LDS
        else if (!two && state == pf.state) { q.x = pf.x; q.y = pf.y; q.z = 0u; q.w = 0u; pf.state = kNone; }
        else {
            q = *reinterpret_cast<const u32x4_u*>(v.base + at);
            if (!two) { pf.state = state + 1u; pf.x = q.z; pf.y = q.w; }      // (state + 1 may be the first two-entry record: what was read there is the table's pad, and `two` keeps it from being used)
        }
        if ((q.y >> 24) == cl) return q.x;
        if (two && (q.w >> 24) == cl) return q.z;
        state = q.y & 0xFFFFFFu;
        AM_BOUNDS(state < d.n_rows);
    }
    const bool in_lds = state < hot_rows && cl <= (1u << kLdsLog2Cols);
    uint32_t e;
    if (in_lds) e = lds_read_u32(lds_rows_addr + (((state << kLdsLog2Cols) + cl - 1u) << 2));
    else {
        const uint32_t off = cl <= (1u << d.hot_log2) ? v.off_hot + (((state << d.hot_log2) + cl - 1u) << 2) : (((state << d.log2_classes) + cl) << 2);
        e = *reinterpret_cast<const uint32_t*>(v.base + off);
    }
    return e;
}

// One lane's unit.  All positions are 32-bit offsets from tp = text + (first byte of the group) - W, W = the warm-up rounded up to 16 (so offsets keep the
// alignment of the text).  TW = bytes of text a lane asks for at a time (16 or 64): a lane's unit lies `chunk` bytes from its neighbour's, so a 16-byte load touches 64
// different lines and uses an eighth of each; the other pieces come 16, 32, ... steps (tens of microseconds) later, by when the line has long left the L2 (an XCD's L2
// turns over every ~5 us here): eight L2 misses per line of text.  With TW = 64 a lane asks for half a line at once (four loads issued back to back, held in registers:
// buf[0] = the next block, the blocks move down one place per 16 steps): two.
template <int MODE, int TW, int VAR>
__device__ __forceinline__ void dfa_walk_unit(DfaLane<MODE>& L, const DfaDev& v, const uint8_t* tp, uint64_t tp_off, uint32_t lds_cls_addr, uint32_t lds_rows_addr,
                                              uint32_t hot_rows, uint32_t W, uint32_t end_r)
{
    constexpr int NV = TW / 16;
    const DfaView& d = L.d; const BatchView& b = L.b; const ScanOut& o = L.o;
    // where the lane's unit starts and ends: recomputed from the lane number where they are needed (two registers less to hold; W and end_r -- the end of the batch -- are uniform)
#define cs_r (lane_id() * d.chunk + W)
#define ce_r (cs_r + d.chunk < end_r ? cs_r + d.chunk : end_r)
    // tp_off = the batch offset tp stands for (for the first group it is "negative", i.e. wraps: nothing is read there, and the sums below wrap back)
    const uint64_t cs = tp_off + cs_r;
    uint32_t h = find_haystack(b, cs);
    uint64_t hs = b.offsets[h];
    const uint64_t he64 = b.offsets[h + 1];
    uint32_t pos = (cs - hs > d.warm) ? ((cs_r - d.warm) & ~15u) : (uint32_t)(hs - tp_off);      // a longer warm-up is as exact; 16-byte blocks from the start
    if (cs - hs > d.warm && tp_off + pos < hs) pos = (uint32_t)(hs - tp_off);                    // (never before the haystack: found by the index assertions of round 6 -- harmless, the
                                                                                                 // state at cs depends on the `warm` bytes before it only, but the plain form does not either)
    uint32_t he_r = he64 - tp_off < 0xFFFFFFFFull ? (uint32_t)(he64 - tp_off) : 0xFFFFFFFFu;
    uint32_t state = 0;
    DfaAhead pf;
    u32x4_n buf[NV];
    uint32_t avail = 0;                                     // blocks in buf (only blocks that lie wholly inside the haystack and the unit are ever asked for)
    bool rare_next = false;                                 // the byte at pos has no column: the byte-by-byte path takes it (and the bytes up to the next block)
    while (pos < ce_r) {
        if (MODE == kModeTokens) {
            // the lanes that are still walking agree (they read the same LDS words) on whether their wavefront's superblock can take what the next
            // block may produce (64 lanes x 16 bytes); if not, the first of them seals it and draws the next one
            volatile uint32_t* wv = L.wv;
            const uint32_t fill = wv[0], sb = wv[1], exhausted = wv[2];
            if (!exhausted && (sb == kNone || fill + kDfaSuperReserve > kDfaSuper)) {
                const uint64_t act = __ballot(1);
                if (lane_id() == (uint32_t)__ffsll((unsigned long long)act) - 1u) {
                    if (sb != kNone) o.block_next[sb] = fill;
                    const uint32_t id = atomicAdd(o.pool_ctrl, 1u);
                    if (id >= o.n_blocks) { o.pool_ctrl[1] = 1u; wv[1] = kNone; wv[2] = 1u; }      // the counts stay exact; the host repeats the call with a pool sized by them
                    else { wv[1] = id; o.block_next[o.n_blocks + id] = wv[4]; wv[3] = wv[5]; }     // (its first group, for k_dfa_place; the ordinal its tokens count from)
                    wv[0] = 0u;
                }
                wave_lds_fence();
            }
            L.sb = (uint32_t)__builtin_amdgcn_readfirstlane((int)wv[1]); L.ord = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wv[5] - wv[3]));      // (uniform: scalar registers)
        }
        if (pos >= he_r) {                                  // the next non-empty haystack starts here
            uint64_t e64 = tp_off + he_r, s64;
            do { h++; s64 = e64; e64 = b.offsets[h + 1]; } while (e64 == s64);
            hs = s64;
            he_r = e64 - tp_off < 0xFFFFFFFFull ? (uint32_t)(e64 - tp_off) : 0xFFFFFFFFu;
            state = 0;
        }
        const uint32_t lim = he_r < ce_r ? he_r : ce_r;
        if ((pos & 15u) == 0 && pos + 16u <= lim && !rare_next) {
            const bool mine = pos >= cs_r;                  // (cs_r is a multiple of 16: a block lies on one side of it)
            if (MODE == kModeAny && mine && o.flags[h]) { pos = lim; avail = 0; continue; }      // the haystack is already flagged: on to the next one
            if (avail == 0) {                               // a whole TW-byte piece when it lies inside the haystack and the unit, one block otherwise (the ends of a unit)
                AM_BOUNDS(tp_off + pos + 16u <= ((b.total + 15u) & ~15ull) && tp_off + pos >= hs && pos < ce_r);
                if (NV > 1 && (pos & (uint32_t)(TW - 1)) == 0 && pos + (uint32_t)TW <= lim) {
#pragma unroll
                    for (int j = 0; j < NV; j++) buf[j] = *reinterpret_cast<const u32x4_n*>(tp + pos + 16u * j);
                    avail = NV;
                } else {
                    buf[0] = *reinterpret_cast<const u32x4_n*>(tp + pos);
                    avail = 1;
                }
            }
            // A byte without a column (a byte in a thousand, by the flattener's choice of the columns) ends the lane's block: the byte-by-byte path below walks it, and the
            // fall-back walk's registers stay out of this loop.
            uint32_t done = 16u;
            if (VAR == 1) {
                // The lanes of a wavefront do not keep step inside a block: a lane whose byte needs a second trip (a record that hands the question on) takes it in the
                // next turn while its neighbours go on to their next byte, and the wavefront waits for ONE load per turn -- the same instruction for a record and for an
                // entry of a row (16 aligned bytes; what was asked for is picked out of them).  A block costs the turns of its slowest lane, not 16 x the slowest step.
                uint32_t cur = 0;
                while (cur < 16u) {
                    const uint32_t word = cur < 8u ? (cur < 4u ? buf[0].x : buf[0].y) : (cur < 12u ? buf[0].z : buf[0].w);
                    const uint32_t cl = lds_read_u8(lds_cls_addr + ((word >> ((cur & 3u) << 3)) & 0xFFu));
                    if (cl == kDfaRare) { done = cur; break; }
                    uint32_t e = 0u;
                    bool adv = true;
                    if (cl != 0u) {
                        AM_BOUNDS(state < d.n_states && cl < (1u << d.log2_classes));
                        const bool rec = state >= d.n_rows, two = state >= d.n_rows + d.n_single;
                        const bool in_lds = state < hot_rows && cl <= (1u << kLdsLog2Cols);
                        const uint32_t off = rec ? (two ? v.off_chain2 + ((state - d.n_rows - d.n_single) << 4) : v.off_chain + ((state - d.n_rows) << 3))
                                                 : (cl <= (1u << d.hot_log2) ? v.off_hot + (((state << d.hot_log2) + cl - 1u) << 2) : (((state << d.log2_classes) + cl) << 2));
                        u32x4_n q = {0u, 0u, 0u, 0u};
                        if (!in_lds) q = *reinterpret_cast<const u32x4_n*>(v.base + (off & ~15u));
                        if (in_lds) e = lds_read_u32(lds_rows_addr + (((state << kLdsLog2Cols) + cl - 1u) << 2));
                        else {
                            const uint32_t lo = (off & 8u) ? q.z : q.x, hi = (off & 8u) ? q.w : q.y;      // the 8 bytes the offset names
                            if (!rec) e = (off & 4u) ? hi : lo;
                            else if ((hi >> 24) == cl) e = lo;
                            else if (two && (q.w >> 24) == cl) e = q.z;
                            else { state = hi & 0xFFFFFFu; adv = false; AM_BOUNDS(state < d.n_rows); }
                        }
                    }
                    if (adv) {
                        state = e & kDfaStateMask;
                        if ((e >> kDfaEndShift) && mine) L.found(h, pos + cur - cs_r, tp_off + pos + (uint64_t)cur + 1u - hs, state, e >> kDfaEndShift);
                        cur++;
                    }
                }
            } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t cur = q == 0 ? buf[0].x : q == 1 ? buf[0].y : q == 2 ? buf[0].z : buf[0].w;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t byte = (cur >> (8 * i)) & 0xFFu;
                    const uint32_t cl = lds_read_u8(lds_cls_addr + byte);
                    if (done == 16u) {
                        if (cl == kDfaRare) done = (uint32_t)(4 * q + i);
                        else {
                            const uint32_t e = dfa_step(d, v, lds_rows_addr, hot_rows, state, cl, pf);
                            state = e & kDfaStateMask;
                            if ((e >> kDfaEndShift) && mine) L.found(h, pos + (uint32_t)(4 * q + i) - cs_r, tp_off + pos + (uint64_t)(4 * q + i) + 1u - hs, state, e >> kDfaEndShift);
                        }
                    }
                }
                asm volatile("" ::: "memory");              // (keeps the class look-ups of later words behind this word's steps: they would each hold a register)
            }
            }
#pragma unroll
            for (int j = 0; j + 1 < NV; j++) buf[j] = buf[j + 1];
            avail--;
            pos += done;
            if (done != 16u) { avail = 0; rare_next = true; }
        } else {
            rare_next = false;
            const uint32_t byte = tp[pos], cl = lds_read_u8(lds_cls_addr + byte);
            const uint32_t e = cl == kDfaRare ? dfa_rare_step(d, state, (d.ic && byte - 0x41u < 26u) ? byte + 0x20u : byte) : dfa_step(d, v, lds_rows_addr, hot_rows, state, cl, pf);
            state = e & kDfaStateMask;
            pos++;
            if ((e >> kDfaEndShift) && pos > cs_r) L.found(h, pos - 1u - cs_r, tp_off + pos - hs, state, e >> kDfaEndShift);
        }
    }
    L.flush();
#undef cs_r
#undef ce_r
}

}  // namespace

// Persistent wavefronts: a workgroup of 16 copies the byte -> class map and the first `hot_rows` rows of the table into LDS (the flattener numbers the states so
// that these are the root, the first letters and the heaviest prefixes: a third or more of the steps on natural text never leave the CU), then its wavefronts take
// groups of 64 units until none is left.  Token mode: a wavefront's superblock serves the groups it takes until it is full or 16 groups old.
template <int MODE, int TW, int VAR>
__global__ __launch_bounds__(1024, 8) void k_dfa(DfaView d, BatchView b, ScanOut o, uint64_t n_units, uint32_t hot_rows, uint32_t lds_n1, uint32_t lds_n2)
{
    extern __shared__ uint32_t s_dyn[];
    uint32_t* s_rows = s_dyn;                                                     // hot_rows << 5 entries: columns 1 .. 32 of the first rows
    uint32_t* s_wave = s_dyn + ((size_t)hot_rows << kLdsLog2Cols);                // 16 x 8: per wavefront, token mode (DfaLane::wv)
    uint8_t* s_cls = reinterpret_cast<uint8_t*>(s_wave + 128);
    uint32_t* s_rec1 = reinterpret_cast<uint32_t*>(s_cls + 256);                  // lds_n1 single-entry records (+ one of padding: a record is read as 16 bytes), then lds_n2 two-entry records
    uint32_t* s_rec2 = s_rec1 + 2u * (lds_n1 + 2u);
    const uint32_t n_cls = 1u << d.log2_classes;
    for (uint32_t i = threadIdx.x; i < (hot_rows << kLdsLog2Cols); i += 1024u) {
        const uint32_t c = (i & ((1u << kLdsLog2Cols) - 1u)) + 1u;
        s_rows[i] = c < n_cls ? d.next[((uint64_t)(i >> kLdsLog2Cols) << d.log2_classes) + c] : 0u;
    }
    if (threadIdx.x < 256u) s_cls[threadIdx.x] = d.cls[threadIdx.x];
    for (uint32_t i = threadIdx.x; i < 2u * (lds_n1 + 2u); i += 1024u) s_rec1[i] = i < 2u * (lds_n1 + 1u) ? reinterpret_cast<const uint32_t*>(d.chain)[i] : 0u;      // (chain[] is padded by one record)
    for (uint32_t i = threadIdx.x; i < 4u * lds_n2; i += 1024u) s_rec2[i] = reinterpret_cast<const uint32_t*>(d.chain2)[i];
    // (the wavefront's number goes through readfirstlane: everything derived from it -- the group, the text base, the wavefront's LDS words -- is then uniform for
    // the compiler as well and lives in scalar registers)
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave)), lane = threadIdx.x % kWave;
    uint32_t* wv = &s_wave[8u * w];
    if (lane < 8u) wv[lane] = lane == 1u ? kNone : 0u;
    __syncthreads();
    DfaDev v;
    v.base = reinterpret_cast<const uint8_t*>(d.next);
    v.off_hot = (uint32_t)(reinterpret_cast<const uint8_t*>(d.hot) - v.base);
    v.off_chain = (uint32_t)(reinterpret_cast<const uint8_t*>(d.chain) - v.base);
    v.off_chain2 = (uint32_t)(reinterpret_cast<const uint8_t*>(d.chain2) - v.base);
    v.lds_rec1 = (uint32_t)(uintptr_t)s_rec1; v.lds_rec2 = (uint32_t)(uintptr_t)s_rec2; v.lds_n1 = lds_n1; v.lds_n2 = lds_n2;
    const uint32_t lds_rows_addr = (uint32_t)(uintptr_t)s_rows, lds_cls_addr = (uint32_t)(uintptr_t)s_cls;       // (the kernel has no static LDS: the dynamic block starts at 0)
    const uint64_t n_groups = (n_units + kWave - 1) / kWave, n_waves = (uint64_t)gridDim.x * 16u;
    const uint32_t W = (d.warm + 15u) & ~15u;
    uint32_t k = 0;                                                               // this wavefront's k-th group
    for (uint64_t g = (uint64_t)blockIdx.x * 16u + w; g < n_groups; g += n_waves, k++) {
        const uint64_t u = g * kWave + lane;
        uint32_t group_values = 0;
        if (MODE == kModeTokens) {
            // a superblock older than 16 groups is sealed (a token names its group by a 4-bit ordinal); the next block's reserve draws a new one
            if (lane == 0) {
                if (wv[1] != kNone && k - wv[3] >= kTokMaxOrd) { o.block_next[wv[1]] = wv[0]; wv[1] = kNone; wv[0] = 0u; }
                wv[4] = (uint32_t)g; wv[5] = k;
            }
            wave_lds_fence();
        }
        if (u < n_units) {
            const uint64_t tp_off = g * kWave * d.chunk - W;                      // (wraps below zero for g = 0: an offset, never an address that is read)
            const uint8_t* tp = b.text + tp_off;
            const uint32_t end_r = b.total - tp_off < 0xFFFFFFFFull ? (uint32_t)(b.total - tp_off) : 0xFFFFFFFFu;      // the end of the batch, as an offset like the others
            DfaLane<MODE> L{d, b, o, wv};
            if (MODE == kModeEmit) L.out = o.records + o.unit_offsets[u];
            dfa_walk_unit<MODE, TW, VAR>(L, v, tp, tp_off, lds_cls_addr, lds_rows_addr, hot_rows, W, end_r);
            if (MODE == kModeCount || MODE == kModeTokens) o.unit_counts[u] = L.nrec;
            group_values = L.nval;
        }
        if (MODE == kModeCount) {                           // the group's values: one atomic per wavefront and group (at most 14 x 64 x chunk: 32 bits hold it)
            const uint32_t sum = wave_inclusive_sum(group_values, lane);
            if (lane == kWave - 1u && sum) atomicAdd(reinterpret_cast<unsigned long long*>(o.total_values), (unsigned long long)sum);
        }
        if (MODE == kModeTokens) wave_lds_fence();          // (the token slots of this group are taken before the next group's first reserve looks at the fill)
    }
    if (MODE == kModeTokens && lane == 0 && wv[1] != kNone) o.block_next[wv[1]] = wv[0];      // the last superblock's fill
}

// How dense are needle ends in this batch?  n_samples lanes, spread evenly over the text, each walk `len` bytes from the root (no warm-up, haystack boundaries
// ignored: an estimate) and count the steps that land on a needle end.  What the ABI layer chooses the route of a dictionary by (am_abi.cpp make_plan).
__global__ __launch_bounds__(256) void k_dfa_sample(DfaView d, const uint8_t* __restrict__ text, uint64_t total, uint32_t n_samples, uint32_t len, uint32_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t ends = 0;
    if (i < n_samples) {
        const uint64_t at = ((total / n_samples) * i) & ~15ull;
        const uint64_t stop = at + len < total ? at + len : total;
        uint32_t state = 0;
        for (uint64_t p = at; p < stop; p++) {
            const uint32_t byte = text[p], cl = d.cls[byte];
            const uint32_t e = cl == kDfaRare ? dfa_rare_step(d, state, (d.ic && byte - 0x41u < 26u) ? byte + 0x20u : byte) : dfa_common_step(d, state, cl);
            state = e & kDfaStateMask;
            ends += (e >> kDfaEndShift) ? 1u : 0u;
        }
    }
    const uint64_t sum = wave_sum_u64(ends);
    if (lane_id() == 0 && sum) atomicAdd(out, (uint32_t)sum);
}
hipError_t launch_dfa_sample(const DfaView& d, const uint8_t* text, uint64_t total, uint32_t n_samples, uint32_t len, uint32_t* out, hipStream_t st)
{
    if (n_samples == 0 || total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_dfa_sample, dim3((n_samples + 255u) / 256u), dim3(256), 0, st, d, text, total, n_samples, len, out);
    return hipGetLastError();
}

// token -> the record it stands for, at unit_offsets[unit] + seq.  A workgroup takes superblock after superblock (a persistent grid: what it learns about the states stays).  A superblock's tokens come from ONE wavefront, in the order it
// made them (step by step over the 64 units of a group, then the next group it took): written out token by token that is a 16-byte write scattered over 64 stretches
// of the result, one L2 request each.  So the workgroup first sorts the superblock in LDS by (group ordinal, lane, seq) -- a counting sort: a unit's tokens in one
// superblock have consecutive seq, so slot = bucket start + seq - the bucket's smallest seq -- and then neighbouring lanes write neighbouring records.  The haystack of a
// token is looked up from its position (the per-KiB haystack index of the batch; neighbouring records ask for neighbouring entries), the reference state from the DFA state:
// a random 8-byte read per record -- most of this kernel's L2 requests (2.6 of 2.8 x 10^8 per 2 GiB of natural text) -- which a direct-mapped table in LDS answers for the
// states it has seen (text repeats its words: the workgroup keeps the table from superblock to superblock).
constexpr uint32_t kPlaceBuckets = kTokMaxOrd * kWave;
constexpr uint32_t kPlaceCacheLog2 = 12;         // 4 096 entries of 8 bytes -- or 8 192 of 4 where a state's tag and its reference state share a word (ref_bits != 0)
__global__ __launch_bounds__(1024) void k_dfa_place(const u32x2_v* __restrict__ pool, const uint32_t* __restrict__ fill, const uint32_t* __restrict__ first_group, uint32_t n_super,
                                                    const uint64_t* __restrict__ unit_offsets, BatchView b, const u32x2* __restrict__ dfa_out, uint32_t n_states, uint32_t chunk, uint32_t n_waves,
                                                    uint32_t ref_bits, Record* __restrict__ out)
{
    __shared__ u32x2_v s_tok[kDfaSuper];
    __shared__ uint32_t s_cnt[kPlaceBuckets], s_min[kPlaceBuckets], s_base[kPlaceBuckets];
    __shared__ u32x2_v s_seen[1u << kPlaceCacheLog2];              // {DFA state, its reference state + 1}: one 8-byte word, written and read whole
    static_assert(kPlaceBuckets == 1024, "one bucket per thread of the workgroup");
    for (uint32_t i = threadIdx.x; i < (1u << kPlaceCacheLog2); i += 1024u) { u32x2_v e; e.x = ref_bits ? 0u : kNone; e.y = 0u; s_seen[i] = e; }      // (4-byte entries: 0 = empty, a reference state + 1 is never 0)
    for (uint32_t sb = blockIdx.x; sb < n_super; sb += gridDim.x) {
    const uint32_t n = fill[sb];
    if (n == 0) continue;                                        // (uniform: the whole workgroup goes on)
    const u32x2_v* tok = pool + (uint64_t)sb * kDfaSuper;
    __syncthreads();                                             // (the previous superblock's records are out of s_tok)
    s_cnt[threadIdx.x] = 0u; s_min[threadIdx.x] = 0xFFFFFFFFu;
    __syncthreads();
    u32x2_v t[kDfaSuper / 1024];
#pragma unroll
    for (uint32_t k = 0; k < kDfaSuper / 1024; k++) {
        const uint32_t i = threadIdx.x + k * 1024u;
        if (i < n) {
            t[k] = __builtin_nontemporal_load(tok + i);
            const uint32_t bucket = ((t[k].x >> kTokOrdShift) << 6) | (t[k].y >> 26);
            atomicAdd(&s_cnt[bucket], 1u);
            atomicMin(&s_min[bucket], (t[k].y >> kTokPosBits) & (kTokMaxChunk - 1u));
        }
    }
    __syncthreads();
    if (threadIdx.x < kWave) {                                   // exclusive sums of 1024 counts by one wavefront: 16 per lane
        uint32_t c[16], sum = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) { c[j] = s_cnt[threadIdx.x * 16u + j]; sum += c[j]; }
        uint32_t run = wave_inclusive_sum(sum, threadIdx.x) - sum;
#pragma unroll
        for (int j = 0; j < 16; j++) { s_base[threadIdx.x * 16u + j] = run; run += c[j]; }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < kDfaSuper / 1024; k++) {
        const uint32_t i = threadIdx.x + k * 1024u;
        if (i < n) {
            const uint32_t bucket = ((t[k].x >> kTokOrdShift) << 6) | (t[k].y >> 26);
            AM_BOUNDS(s_base[bucket] + ((t[k].y >> kTokPosBits) & (kTokMaxChunk - 1u)) - s_min[bucket] < n);
            s_tok[s_base[bucket] + ((t[k].y >> kTokPosBits) & (kTokMaxChunk - 1u)) - s_min[bucket]] = t[k];
        }
    }
    __syncthreads();
    const uint64_t g_first = first_group[sb];
    for (uint32_t i = threadIdx.x; i < n; i += 1024u) {
        const u32x2_v q = s_tok[i];
        const uint64_t u = (g_first + (uint64_t)(q.x >> kTokOrdShift) * n_waves) * kWave + (q.y >> 26);
        const uint64_t g = u * chunk + (q.y & (kTokMaxChunk - 1u));
        const uint32_t h = find_haystack(b, g);
        AM_BOUNDS(g < b.total && h < b.n_hay && b.offsets[h] <= g && g < b.offsets[h + 1] && (q.x & kDfaStateMask) < n_states && dfa_out[q.x & kDfaStateMask].x != 0u &&
                  unit_offsets[u] + ((q.y >> kTokPosBits) & (kTokMaxChunk - 1u)) < unit_offsets[u + 1]);
        const uint32_t st = q.x & kDfaStateMask;
        uint32_t ref1;
        if (ref_bits) {
            // 4-byte entries, twice as many, found by the state's low bits: the states are numbered by weight, so the hottest of a kind do not collide with each other
            // (a simulation over the text's own counts: 47 % of the look-ups answered where the hashed 4 096 answer 34 %)
            uint32_t* seen4 = reinterpret_cast<uint32_t*>(s_seen);
            const uint32_t slot = st & ((2u << kPlaceCacheLog2) - 1u), tag = st >> (kPlaceCacheLog2 + 1u), e = seen4[slot];
            ref1 = e & ((1u << ref_bits) - 1u);
            if (e == 0u || (e >> ref_bits) != tag) { ref1 = dfa_out[st].x; seen4[slot] = (tag << ref_bits) | ref1; }
        } else {
            const uint32_t slot = (st * 0x9E3779B1u) >> (32u - kPlaceCacheLog2);
            const u32x2_v seen = s_seen[slot];
            ref1 = seen.y;
            if (seen.x != st) { ref1 = dfa_out[st].x; u32x2_v e; e.x = st; e.y = ref1; s_seen[slot] = e; }      // (lanes that race for a slot each write a whole, valid pair)
        }
        u32x4_n r;
        const uint64_t end_pos = g + 1u - b.offsets[h];
        r.x = (uint32_t)end_pos; r.y = (uint32_t)(end_pos >> 32); r.z = h; r.w = ref1 - 1u;
        __builtin_nontemporal_store(r, reinterpret_cast<u32x4_n*>(out + unit_offsets[u] + ((q.y >> kTokPosBits) & (kTokMaxChunk - 1u))));
    }
    }
}

uint64_t dfa_units(const DfaView& d, const BatchView& b) { return d.chunk ? (b.total + d.chunk - 1) / d.chunk : 0; }

// How many wavefronts does a CU of this device run at a time?  32 by the architecture -- two workgroups of 16 share a CU and its 160 KiB of LDS --, but boxes were met on which the second
// 16 wait for the first (LABNOTES R6.13: k_dfa 7.4 -> 10.9 ms per 2 GiB there).  On such a device ONE workgroup per CU is launched and given all of the LDS: twice the rows, three times the
// records -- fewer L2 requests for the wavefronts that do run.  The probe (am_debug_resident_waves' spinning kernel, 16 against 32 wavefronts per CU, twice; ~2 ms, once per device and
// process) must say "twice the time" both times.
static uint32_t dfa_resident_sets(int dev)
{
    static std::atomic<int> state[64];                       // 0: not asked, 1: one set of 16 wavefronts at a time, 2: two
    if (dev < 0 || dev >= 64) return 2;
    int s = state[dev].load(std::memory_order_acquire);
    if (s == 0) {
        s = 2;
        int n_cu = 0;
        uint32_t* d_out = nullptr;
        hipEvent_t a = nullptr, b = nullptr;
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n_cu > 0 && hipMalloc((void**)&d_out, 64) == hipSuccess &&
            hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) {
            int slow = 0;
            for (int round = 0; round < 2; round++) {
                float ms[2] = {0, 0};
                bool good = true;
                for (int k = 0; k < 2 && good; k++) {
                    const uint32_t wgs = k == 0 ? (uint32_t)n_cu : 8u * (uint32_t)n_cu, threads = k == 0 ? 1024u : 256u;
                    good = launch_spin(wgs, threads, 1000, d_out, nullptr) == hipSuccess && hipEventRecord(a, nullptr) == hipSuccess && launch_spin(wgs, threads, 600000, d_out, nullptr) == hipSuccess &&
                           hipEventRecord(b, nullptr) == hipSuccess && hipEventSynchronize(b) == hipSuccess && hipEventElapsedTime(&ms[k], a, b) == hipSuccess;
                }
                if (good && ms[1] > 1.7f * ms[0]) slow++;
            }
            if (slow == 2) s = 1;
        }
        (void)hipGetLastError();
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
        if (d_out) (void)hipFree(d_out);
        state[dev].store(s, std::memory_order_release);
    }
    return (uint32_t)s;
}
static uint32_t dfa_tune();
static uint32_t dfa_per_cu()
{
    const uint32_t asked = (dfa_tune() >> 4) & 15u;
    if (asked) return asked;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 2;
    return dfa_resident_sets(dev);
}
// what a workgroup keeps in LDS.  Two workgroups per CU (80 KiB each): the first 512 rows (64 KiB) and the 640 hottest records of either kind (measured on the natural-text workload:
// the first 640 single-entry records take 1.9 % of the steps, the first 640 two-entry ones 1.9 %; a row more takes 0.01 %).  One workgroup per CU: 1 008 rows, 2 048 + 1 024 records.
struct DfaLds { uint32_t rows, n1, n2; };
static DfaLds dfa_lds(const DfaView& d, uint32_t per_cu)
{
    const bool all = per_cu == 1u;
    DfaLds l;
    l.rows = std::min<uint32_t>(d.n_rows, all ? 1008u : 512u);
    l.n1 = std::min<uint32_t>(d.n_single, all ? 2048u : 640u);
    l.n2 = std::min<uint32_t>(d.n_states - d.n_rows - d.n_single, all ? 1024u : 640u);
    return l;
}
// AM_DFA_TUNE (measurements only; no value changes a result): bits 0-3 = the walk (1: 16 bytes of text per request, lanes in step; 2: 64 bytes, lanes in step; 3: 64 bytes, lanes out of step; 0: the default),
// bits 4-7 = workgroups per CU (0: two, or one with all of the LDS where the device runs 16 wavefronts per CU at a time), bits 8-23 = rows kept in LDS + 1 (0: what fits), bit 24 = no records in LDS, bit 25 = k_dfa_place's table of seen states with 8-byte entries
static uint32_t dfa_tune() { const long v = cfg::get(cfg::kDfaTune); return v > 0 ? (uint32_t)v : 0u; }
static uint32_t dfa_workgroups(const DfaView& d, const BatchView& b, int n_cu)
{
    const uint64_t n_groups = (dfa_units(d, b) + kWave - 1) / kWave;
    const uint32_t per_cu = dfa_per_cu();
    return (uint32_t)std::min<uint64_t>((uint64_t)n_cu * per_cu, (n_groups + 15) / 16);
}
// Can this device walk this section at all?  The walk addresses hot table and chain records as 32-bit offsets from the rows and a group's text as 32-bit offsets
// from its start, and a workgroup wants more than 64 KiB of dynamic LDS (an attribute per instantiation and device, raised here once).  make_plan asks before it
// chooses the route: a section that cannot be walked is not an error, the suffix filter takes the batch.
template <int MODE, int TW, int VAR>
static bool dfa_raise_lds(int dev)
{
    static std::atomic<int> state[64];                       // 0: not tried, 1: raised, 2: refused
    int s = state[dev].load(std::memory_order_acquire);
    if (s == 0) {
        s = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dfa<MODE, TW, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess ? 1 : 2;
        if (s == 2) (void)hipGetLastError();
        state[dev].store(s, std::memory_order_release);
    }
    return s == 1;
}
template <int MODE>
static bool dfa_raise_lds_mode(int dev) { return dfa_raise_lds<MODE, 16, 0>(dev) && dfa_raise_lds<MODE, 64, 0>(dev) && dfa_raise_lds<MODE, 64, 1>(dev); }
bool dfa_usable(const DfaView& d)
{
    const uint8_t *p_next = reinterpret_cast<const uint8_t*>(d.next), *p_hot = reinterpret_cast<const uint8_t*>(d.hot), *p_chain = reinterpret_cast<const uint8_t*>(d.chain),
                  *p_chain2 = reinterpret_cast<const uint8_t*>(d.chain2);
    if (d.n_single > d.n_states - d.n_rows) return false;
    if (p_hot < p_next || p_chain < p_next || p_chain2 < p_next || (uint64_t)(p_chain - p_next) + ((uint64_t)d.n_single + 1u) * 8u >= (1ull << 32) ||
        (uint64_t)(p_chain2 - p_next) + (uint64_t)(d.n_states - d.n_rows - d.n_single) * 16u >= (1ull << 32) ||
        (uint64_t)(p_hot - p_next) + ((uint64_t)d.n_rows << (d.hot_log2 + 2u)) >= (1ull << 32) || (uint64_t)d.chunk * kWave + d.warm + 16u >= (1ull << 31)) return false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    return dfa_raise_lds_mode<kModeCount>(dev) && dfa_raise_lds_mode<kModeEmit>(dev) && dfa_raise_lds_mode<kModeAny>(dev) && dfa_raise_lds_mode<kModeTokens>(dev);
}
template <int MODE, int TW, int VAR>
static hipError_t launch_dfa_tw(const DfaView& d, const BatchView& b, const ScanOut& o, int n_cu, hipStream_t st)
{
    const uint64_t n_units = dfa_units(d, b);
    if (n_units == 0) return hipSuccess;
    if (!dfa_usable(d)) return hipErrorInvalidValue;         // (make_plan does not come here with such a section)
    const DfaLds l = dfa_lds(d, dfa_per_cu());
    uint32_t hot = l.rows;
    if ((dfa_tune() >> 8) & 0xFFFFu) hot = std::min<uint32_t>(hot, ((dfa_tune() >> 8) & 0xFFFFu) - 1u);
    const size_t lds_fixed = ((size_t)l.rows << (kLdsLog2Cols + 2u)) + 128 * 4 + 256;
    uint32_t n1 = l.n1, n2 = l.n2;
    if ((dfa_tune() >> 24) & 1u) n1 = n2 = 0u;
    const size_t lds = lds_fixed + 8u * (n1 + 2u) + 16u * n2;
    hipLaunchKernelGGL((k_dfa<MODE, TW, VAR>), dim3(dfa_workgroups(d, b, n_cu)), dim3(1024), lds, st, d, b, o, n_units, hot, n1, n2);
    return hipGetLastError();
}
template <int MODE>
static hipError_t launch_dfa_t(const DfaView& d, const BatchView& b, const ScanOut& o, int n_cu, hipStream_t st)
{
    switch (dfa_tune() & 15u) {
    case 1: return launch_dfa_tw<MODE, 16, 0>(d, b, o, n_cu, st);
    case 3: return launch_dfa_tw<MODE, 64, 1>(d, b, o, n_cu, st);
    default: return launch_dfa_tw<MODE, 64, 0>(d, b, o, n_cu, st);
    }
}
hipError_t launch_dfa(int mode, const DfaView& d, const BatchView& b, const ScanOut& o, int n_cu, hipStream_t st)
{
    if (mode == kModeCount) return launch_dfa_t<kModeCount>(d, b, o, n_cu, st);
    if (mode == kModeEmit) return launch_dfa_t<kModeEmit>(d, b, o, n_cu, st);
    if (mode == kModeAny) return launch_dfa_t<kModeAny>(d, b, o, n_cu, st);
    return hipErrorInvalidValue;
}

// records in one walk: is the unit small enough for a token's 13-bit fields, and how many wavefronts walk (= superblocks that may end up partly filled)
bool dfa_tokens_ok(const DfaView& d) { return d.chunk != 0 && d.chunk <= kTokMaxChunk; }
uint32_t dfa_token_waves(const DfaView& d, const BatchView& b, int n_cu) { return dfa_workgroups(d, b, n_cu) * 16u; }
// superblocks that hold `records` tokens whatever the split between the wavefronts: a sealed superblock holds at least kDfaSuper - kDfaSuperReserve tokens unless it was
// sealed for its age -- after 16 groups of its wavefront, so at most one such per 16 groups
uint64_t dfa_token_superblocks(uint64_t records, uint32_t n_waves, uint64_t n_units) { return records / (kDfaSuper - kDfaSuperReserve) + n_waves + 16 + n_units / (kWave * kTokMaxOrd); }
uint64_t dfa_superblock_bytes() { return (uint64_t)kDfaSuper * sizeof(u32x2_v); }
hipError_t launch_dfa_tokens(const DfaView& d, const BatchView& b, const ScanOut& o, int n_cu, hipStream_t st) { return launch_dfa_t<kModeTokens>(d, b, o, n_cu, st); }
// o.block_next = [n_blocks fill counts | n_blocks first groups]
// n_waves = dfa_token_waves() as it was when the walk was launched: a token names its group by an ordinal counted in steps of it
hipError_t launch_dfa_place(const DfaView& d, const BatchView& b, const ScanOut& o, uint32_t n_super, const uint64_t* unit_offsets, int n_cu, uint32_t n_waves, uint32_t n_ref_states, Record* out, hipStream_t st)
{
    if (n_super == 0) return hipSuccess;
    if (n_waves != dfa_token_waves(d, b, n_cu)) return hipErrorInvalidValue;       // (the launch parameters changed between the walk and the placement)
    // a table entry of 4 bytes = the state's bits above the slot index | its reference state + 1, where both fit (645k states, 640k reference states: 7 + 20 bits)
    uint32_t rb = 1; while ((1ull << rb) <= (uint64_t)n_ref_states + 1u) rb++;
    uint32_t sb = 1; while ((1ull << sb) < (uint64_t)d.n_states) sb++;
    const uint32_t tag_bits = sb > kPlaceCacheLog2 + 1u ? sb - (kPlaceCacheLog2 + 1u) : 0u;
    const uint32_t ref_bits = (rb < 32u && rb + tag_bits <= 32u && !((dfa_tune() >> 25) & 1u)) ? rb : 0u;
    hipLaunchKernelGGL(k_dfa_place, dim3(std::min<uint32_t>(n_super, (uint32_t)n_cu * 2u)), dim3(1024), 0, st, reinterpret_cast<const u32x2_v*>(o.pool), o.block_next, o.block_next + o.n_blocks, n_super, unit_offsets, b, d.out, d.n_states, d.chunk, n_waves, ref_bits, out);
    return hipGetLastError();
}

}  // namespace dev
}  // namespace am
