// am_dfa.hip -- k_dfa: the byte-level Aho-Corasick automaton with every transition resolved (ImageHeader::off_dfa_next, built by am_flatten.cpp for
// dictionaries that meet match-dense text), walked one lane per stretch of the batch.  Reference semantics: Automaton.hs:482-520 (followCodePoint /
// collectMatches) -- the fallback loop is folded into the table, so a step is ONE dependent load: next = table[state << log2_classes | class(byte)] for the states that
// have a dense row, an 8-byte chain record {child, its class, fallback row} for the single-child states of a word's tail (image layout: am_image.h, DfaView).
//
// Why a second scan kernel: k_sf is a filter.  On natural-language text against a 100k-word dictionary four positions in ten pass its LDS filter and a needle
// ends every 6.6 bytes; its resolve phase then costs ~19 divergent 16-byte loads and ~22 VALU instructions per deferred position and the kernel runs at the
// issue limit of the CUs' address units and vector ALUs (profiles/r05_pmc_natural_units.md).  A table walk costs one 4-byte load and ~8 instructions per BYTE
// whatever the text is -- slower than k_sf where matches are rare (it cannot skip anything), several times faster where they are dense.
//
// Work split: unit u = bytes [u * chunk, (u + 1) * chunk) of the concatenated batch, one lane each; the lane owns the matches whose LAST byte lies in its unit and
// warms its state up from the root over the `warm` bytes before it (clipped to the haystack start; a haystack boundary inside the unit resets the state).
// Bytes are read 16 at a time (aligned, nontemporal: the text is a stream, the table is what the caches are for); the byte -> class map sits in LDS.
//   count / any  one launch; unit_counts[u] = the unit's records (what the exclusive scan turns into the records' final places).
//   records      ONE walk as well (kModeEmit with ScanOut::pool set): a lane knows the running number `seq` of each of its matches, so it drops a 16-byte TOKEN
//                {unit, seq, offset in the unit, haystack, DFA state} into its wavefront's current superblock of the pool (slot = one LDS atomic; a superblock = 4096
//                tokens, one device atomic each; order inside does not matter), and after the scan k_dfa_place puts token (u, seq) at unit_offsets[u] + seq as the
//                record it stands for: position order without a sort and without walking the text a second time.  A pool that turns out too small only costs the
//                tokens (the counts stay exact): the host repeats the call with the size the kernel reports, as for k_sf's record blocks.
//                (kModeEmit with ScanOut::records set is the second pass of the plain count -> scan -> emit protocol, kept for chunks beyond 65536 bytes and for
//                batches whose tokens the device could not hold.)
#include <hip/hip_runtime.h>

#include <algorithm>

#include "am_device.h"
#include "am_wave.h"

namespace am {
namespace dev {

namespace {

constexpr uint32_t kWave = 64;
constexpr int kModeTokens = 16;               // template value only: kModeEmit with a token pool
constexpr uint32_t kDfaSuper = 4096;          // tokens per superblock (64 KiB)
constexpr uint32_t kDfaSuperReserve = 1024;   // free slots a wavefront makes sure of before 64 lanes take (at most) 16 steps

template <int MODE>
struct DfaLane {
    const DfaView& d; const BatchView& b; const ScanOut& o;
    Record* out;                              // emit mode: where this unit's next record goes
    uint32_t* wv;                             // token mode: this wavefront's {tokens in its superblock, the superblock's id, pool exhausted} in LDS
    uint64_t unit;
    uint32_t nrec = 0; uint64_t nval = 0;
    uint32_t run_hay = kNone; uint64_t run_val = 0;      // count mode: values of the haystack the lane is in, added with one atomic when it leaves it
    __device__ __forceinline__ void flush()
    {
        if (MODE == kModeCount && o.hay_counts && run_hay != kNone && run_val) atomicAdd(reinterpret_cast<unsigned long long*>(o.hay_counts + run_hay), (unsigned long long)run_val);
        run_val = 0;
    }
    // g = index of the match's last byte in the batch, hs = where its haystack starts
    // end = the entry's end bits: the length of the needle-end list (1..14), kDfaEndLookUp = longer (out[] knows)
    __device__ __forceinline__ void found(uint32_t hay, uint64_t g, uint64_t hs, uint32_t state, uint32_t end)
    {
        const uint64_t end_pos = g + 1u - hs;
        if (MODE == kModeAny) { o.flags[hay] = 1; return; }
        if (MODE == kModeTokens) {
            const uint32_t sb = wv[1];
            if (sb != kNone) {
                const uint32_t slot = atomicAdd(&wv[0], 1u);          // (LDS; < kDfaSuper by the reserve made at the top of the step loop)
                const uint64_t pos = g - unit * d.chunk;                   // offset of the match's last byte in the unit
                o.pool[(uint64_t)sb * kDfaSuper + slot] = Record{(unit << 32) | ((uint64_t)nrec << 16) | pos, hay, state};      // (the DFA state: k_dfa_place looks the reference state up, off the walk's dependent chain)
            }
            nrec++;
            return;
        }
        if (MODE == kModeCount) {
            const uint32_t vl = end < kDfaEndLookUp ? end : d.out[state].y;      // (a count needs no load of its own unless a position reports 15 values or more)
            nrec++; nval += vl;
            if (o.hay_counts) { if (hay != run_hay) { flush(); run_hay = hay; } run_val += vl; }
        } else {
            const u32x2 e = d.out[state];
            out[nrec++] = Record{end_pos, hay, e.x - 1u};
        }
    }
};

}  // namespace

// dfa_common_step (am_image.h) with the first rows of the table in LDS: a chain state answers with its child or hands the question to its fallback's row
// (LDS holds the first min(32, classes) columns of the first hot_rows rows: the classes are numbered by frequency, the 31 most frequent bytes are 98 % of natural text)
constexpr uint32_t kDfaHotLog2Classes = 5;
__device__ __forceinline__ uint32_t dfa_step_lds(const DfaView& d, const uint32_t* s_rows, uint32_t hot_rows, uint32_t state, uint32_t cl)
{
    if (state >= d.n_rows) {
        const u32x2 r = d.chain[state - d.n_rows];
        if ((r.y >> 24) == cl) return r.x;
        state = r.y & 0xFFFFFFu;
    }
    const uint32_t hlc = d.log2_classes < kDfaHotLog2Classes ? d.log2_classes : kDfaHotLog2Classes;
    return (state < hot_rows && cl < (1u << hlc)) ? s_rows[(state << hlc) + cl] : d.next[((uint64_t)state << d.log2_classes) + cl];
}

// one lane's unit (see the head of the file); s_cls = the byte -> class map in LDS
template <int MODE>
__device__ __forceinline__ void dfa_walk_unit(DfaLane<MODE>& L, const uint8_t* s_cls, const uint32_t* s_rows, uint32_t hot_rows, uint64_t u)
{
    const DfaView& d = L.d; const BatchView& b = L.b; const ScanOut& o = L.o;
    const uint64_t cs = u * d.chunk;
    const uint64_t ce = (cs + d.chunk < b.total) ? cs + d.chunk : b.total;
    uint32_t h = find_haystack(b, cs);
    uint64_t hs = b.offsets[h], he = b.offsets[h + 1];
    uint64_t offset = hs;
    if (cs - hs > d.warm) { offset = (cs - d.warm) & ~15ull; if (offset < hs) offset = hs; }      // a longer warm-up is as exact; 16-byte blocks from the start
    uint32_t state = 0;
    typedef uint32_t u32x4_n __attribute__((ext_vector_type(4)));
    while (offset < ce) {
        if (MODE == kModeTokens) {
            // the lanes that are still walking agree (they read the same LDS words) on whether their wavefront's superblock can take what the next
            // step may produce (64 lanes x 16 bytes); if not, the first of them seals it and draws the next one
            volatile uint32_t* wv = L.wv;
            const uint32_t fill = wv[0], sb = wv[1], exhausted = wv[2];
            if (!exhausted && (sb == kNone || fill + kDfaSuperReserve > kDfaSuper)) {
                const uint64_t act = __ballot(1);
                if (lane_id() == (uint32_t)__ffsll((unsigned long long)act) - 1u) {
                    if (sb != kNone) o.block_next[sb] = fill;
                    const uint32_t id = atomicAdd(o.pool_ctrl, 1u);
                    if (id >= o.n_blocks) { o.pool_ctrl[1] = 1u; wv[1] = kNone; wv[2] = 1u; }      // the counts stay exact; the host repeats the call with a pool sized by them
                    else wv[1] = id;
                    wv[0] = 0u;
                }
                wave_lds_fence();
            }
        }
        if (offset >= he) {                                 // the next non-empty haystack starts here
            do { h++; hs = he; he = b.offsets[h + 1]; } while (he == hs);
            state = 0;
        }
        const uint64_t lim = he < ce ? he : ce;
        if ((offset & 15u) == 0 && offset + 16 <= lim) {
            const u32x4_n t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_n*>(b.text + offset));
            const uint32_t w[4] = {t.x, t.y, t.z, t.w};
            const bool mine = offset >= cs;                 // (cs is a multiple of 16: a block lies on one side of it)
            if (MODE == kModeAny && mine && o.flags[h]) { offset = he; continue; }      // the haystack is already flagged: on to the next one
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const uint32_t byte = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                const uint32_t cl = s_cls[byte];
                const uint32_t e = cl == kDfaRare ? dfa_rare_step(d, state, (d.ic && byte - 0x41u < 26u) ? byte + 0x20u : byte) : dfa_step_lds(d, s_rows, hot_rows, state, cl);
                state = e & kDfaStateMask;
                if ((e >> kDfaEndShift) && mine) L.found(h, offset + (uint64_t)i, hs, state, e >> kDfaEndShift);
            }
            offset += 16;
        } else {
            const uint32_t byte = b.text[offset], cl = s_cls[byte];
            const uint32_t e = cl == kDfaRare ? dfa_rare_step(d, state, (d.ic && byte - 0x41u < 26u) ? byte + 0x20u : byte) : dfa_step_lds(d, s_rows, hot_rows, state, cl);
            state = e & kDfaStateMask;
            offset++;
            if ((e >> kDfaEndShift) && offset > cs) L.found(h, offset - 1u, hs, state, e >> kDfaEndShift);
        }
    }
    L.flush();
}

// Persistent wavefronts: a workgroup of 16 copies the byte -> class map and the first `hot_rows` rows of the table into LDS (the flattener numbers the states so
// that these are the root, the first letters and the heaviest prefixes: a third or more of the steps on natural text never leave the CU), then its wavefronts take
// groups of 64 units until none is left.  Token mode: a wavefront's superblock serves all the groups it takes.
template <int MODE>
__global__ __launch_bounds__(1024, 8) void k_dfa(DfaView d, BatchView b, ScanOut o, uint64_t n_units, uint32_t hot_rows)
{
    extern __shared__ uint32_t s_dyn[];
    const uint32_t hlc = d.log2_classes < kDfaHotLog2Classes ? d.log2_classes : kDfaHotLog2Classes;
    uint32_t* s_rows = s_dyn;                                                     // hot_rows << hlc entries: the first columns of the first rows
    uint32_t* s_wave = s_dyn + ((size_t)hot_rows << hlc);                         // 16 x 4: per wavefront (token mode) tokens in its superblock, its id, pool exhausted
    uint8_t* s_cls = reinterpret_cast<uint8_t*>(s_wave + 64);
    for (uint32_t i = threadIdx.x; i < (hot_rows << hlc); i += 1024u) s_rows[i] = d.next[((uint64_t)(i >> hlc) << d.log2_classes) + (i & ((1u << hlc) - 1u))];
    if (threadIdx.x < 256u) s_cls[threadIdx.x] = d.cls[threadIdx.x];
    const uint32_t w = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    if (lane == 0) { s_wave[4u * w] = 0u; s_wave[4u * w + 1u] = kNone; s_wave[4u * w + 2u] = 0u; }
    __syncthreads();
    const uint64_t n_groups = (n_units + kWave - 1) / kWave, n_waves = (uint64_t)gridDim.x * 16u;
    uint64_t nval = 0;
    for (uint64_t g = (uint64_t)blockIdx.x * 16u + w; g < n_groups; g += n_waves) {
        const uint64_t u = g * kWave + lane;
        if (u < n_units) {
            DfaLane<MODE> L{d, b, o, nullptr, &s_wave[4u * w], u};
            if (MODE == kModeEmit) L.out = o.records + o.unit_offsets[u];
            dfa_walk_unit<MODE>(L, s_cls, s_rows, hot_rows, u);
            if (MODE == kModeCount || MODE == kModeTokens) o.unit_counts[u] = L.nrec;
            nval += L.nval;
        }
        if (MODE == kModeTokens) wave_lds_fence();          // (the token slots of this group are taken before the next group's first reserve looks at the fill)
    }
    if (MODE == kModeTokens && lane == 0 && s_wave[4u * w + 1u] != kNone) o.block_next[s_wave[4u * w + 1u]] = s_wave[4u * w];      // the last superblock's fill
    if (MODE == kModeCount) {
        nval = wave_sum_u64(nval);
        if (lane == 0 && nval) atomicAdd(reinterpret_cast<unsigned long long*>(o.total_values), (unsigned long long)nval);
    }
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

// token (unit, seq) -> the record it stands for, at unit_offsets[unit] + seq.  One workgroup per superblock.  A superblock's tokens come from ONE wavefront, in the
// order it made them (step by step over the 64 units of a group, then the next group it took): written out token by token that is a 16-byte write scattered over 64
// stretches of the result, and the kernel is bound by the address units like everything else here.  So the workgroup first sorts the superblock in LDS by (unit, seq) --
// a counting sort: a unit's tokens in one superblock have consecutive seq, so slot = bucket start + seq - the bucket's smallest seq -- and then neighbouring lanes write
// neighbouring records.  Buckets: 64 units x the first 4 groups the superblock holds tokens of; tokens of later groups (text with few matches: a superblock spans
// hundreds of groups) are placed directly.
constexpr uint32_t kPlaceBuckets = 256;
__device__ __forceinline__ void dfa_place_one(const Record& t, const uint64_t* __restrict__ unit_offsets, const uint64_t* __restrict__ hay_offsets,
                                              const u32x2* __restrict__ dfa_out, uint32_t chunk, Record* __restrict__ out)
{
    const uint64_t u = t.end_pos >> 32, seq = (t.end_pos >> 16) & 0xFFFFu, pos = t.end_pos & 0xFFFFu;
    out[unit_offsets[u] + seq] = Record{u * chunk + pos + 1u - hay_offsets[t.haystack], t.haystack, dfa_out[t.state].x - 1u};
}
__global__ __launch_bounds__(1024) void k_dfa_place(const Record* __restrict__ pool, const uint32_t* __restrict__ fill, uint32_t n_super, const uint64_t* __restrict__ unit_offsets,
                                                    const uint64_t* __restrict__ hay_offsets, const u32x2* __restrict__ dfa_out, uint32_t chunk, uint32_t n_waves,
                                                    Record* __restrict__ out)
{
    __shared__ Record s_tok[kDfaSuper];
    __shared__ uint32_t s_cnt[kPlaceBuckets], s_min[kPlaceBuckets], s_base[kPlaceBuckets + 1];
    const uint32_t sb = blockIdx.x;
    if (sb >= n_super) return;
    const uint32_t n = fill[sb];
    if (n == 0) return;
    const Record* tok = pool + (uint64_t)sb * kDfaSuper;
    if (threadIdx.x < kPlaceBuckets) { s_cnt[threadIdx.x] = 0u; s_min[threadIdx.x] = 0xFFFFFFFFu; }
    __syncthreads();
    const uint64_t g_first = tok[0].end_pos >> 38;              // the first group's number (unit >> 6); the wavefront's later groups follow at a stride of n_waves
    Record t[kDfaSuper / 1024];
    uint32_t bucket[kDfaSuper / 1024];
#pragma unroll
    for (uint32_t k = 0; k < kDfaSuper / 1024; k++) {
        const uint32_t i = threadIdx.x + k * 1024u;
        bucket[k] = kNone;
        if (i < n) {
            t[k] = tok[i];
            const uint64_t ord = ((t[k].end_pos >> 38) - g_first) / n_waves;
            if (ord < kPlaceBuckets / 64u) {
                bucket[k] = (uint32_t)ord * 64u + (uint32_t)((t[k].end_pos >> 32) & 63u);
                atomicAdd(&s_cnt[bucket[k]], 1u);
                atomicMin(&s_min[bucket[k]], (uint32_t)((t[k].end_pos >> 16) & 0xFFFFu));
            } else dfa_place_one(t[k], unit_offsets, hay_offsets, dfa_out, chunk, out);
        }
    }
    __syncthreads();
    if (threadIdx.x < kWave) {                                   // exclusive sums of 256 counts by one wavefront: 4 per lane
        uint32_t c[4], sum = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { c[j] = s_cnt[threadIdx.x * 4u + j]; sum += c[j]; }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(incl, off, 64); if ((int)threadIdx.x >= off) incl += v; }
        uint32_t run = incl - sum;
#pragma unroll
        for (int j = 0; j < 4; j++) { s_base[threadIdx.x * 4u + j] = run; run += c[j]; }
        if (threadIdx.x == 63u) s_base[kPlaceBuckets] = run;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < kDfaSuper / 1024; k++)
        if (bucket[k] != kNone) s_tok[s_base[bucket[k]] + (uint32_t)((t[k].end_pos >> 16) & 0xFFFFu) - s_min[bucket[k]]] = t[k];
    __syncthreads();
    const uint32_t n_sorted = s_base[kPlaceBuckets];
    for (uint32_t i = threadIdx.x; i < n_sorted; i += 1024u) dfa_place_one(s_tok[i], unit_offsets, hay_offsets, dfa_out, chunk, out);
}

uint64_t dfa_units(const DfaView& d, const BatchView& b) { return d.chunk ? (b.total + d.chunk - 1) / d.chunk : 0; }

// rows of the table a workgroup keeps in LDS: what fits into 64 KiB (two workgroups of 16 wavefronts share a CU's 160 KiB)
static uint32_t dfa_hot_log2_classes(const DfaView& d) { return std::min<uint32_t>(d.log2_classes, kDfaHotLog2Classes); }
static uint32_t dfa_hot_rows(const DfaView& d) { return std::min<uint32_t>(d.n_rows, (64u * 1024u) >> (dfa_hot_log2_classes(d) + 2u)); }
static uint32_t dfa_workgroups(const DfaView& d, const BatchView& b, int n_cu)
{
    const uint64_t n_groups = (dfa_units(d, b) + kWave - 1) / kWave;
    return (uint32_t)std::min<uint64_t>((uint64_t)n_cu * 2u, (n_groups + 15) / 16);
}
template <int MODE>
static hipError_t launch_dfa_t(const DfaView& d, const BatchView& b, const ScanOut& o, int n_cu, hipStream_t st)
{
    const uint64_t n_units = dfa_units(d, b);
    if (n_units == 0) return hipSuccess;
    const uint32_t hot = dfa_hot_rows(d);
    const size_t lds = ((size_t)hot << (dfa_hot_log2_classes(d) + 2u)) + 64 * 4 + 256;
    static bool raised[64] = {false};                        // (more than 64 KiB of dynamic LDS needs the attribute, once per instantiation and device)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!raised[dev]) { const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dfa<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); if (e != hipSuccess) return e; raised[dev] = true; }
    hipLaunchKernelGGL((k_dfa<MODE>), dim3(dfa_workgroups(d, b, n_cu)), dim3(1024), lds, st, d, b, o, n_units, hot);
    return hipGetLastError();
}
hipError_t launch_dfa(int mode, const DfaView& d, const BatchView& b, const ScanOut& o, int n_cu, hipStream_t st)
{
    if (mode == kModeCount) return launch_dfa_t<kModeCount>(d, b, o, n_cu, st);
    if (mode == kModeEmit) return launch_dfa_t<kModeEmit>(d, b, o, n_cu, st);
    if (mode == kModeAny) return launch_dfa_t<kModeAny>(d, b, o, n_cu, st);
    return hipErrorInvalidValue;
}

// records in one walk: is the unit small enough for a token's 16-bit fields, and how many wavefronts walk (= superblocks that may end up partly filled)
bool dfa_tokens_ok(const DfaView& d) { return d.chunk != 0 && d.chunk <= 65536u; }
uint32_t dfa_token_waves(const DfaView& d, const BatchView& b, int n_cu) { return dfa_workgroups(d, b, n_cu) * 16u; }
// superblocks that hold `records` tokens whatever the split between the wavefronts: a sealed superblock holds at least kDfaSuper - kDfaSuperReserve
uint64_t dfa_token_superblocks(uint64_t records, uint32_t n_waves) { return records / (kDfaSuper - kDfaSuperReserve) + n_waves + 16; }
uint64_t dfa_superblock_bytes() { return (uint64_t)kDfaSuper * sizeof(Record); }
hipError_t launch_dfa_tokens(const DfaView& d, const BatchView& b, const ScanOut& o, int n_cu, hipStream_t st) { return launch_dfa_t<kModeTokens>(d, b, o, n_cu, st); }
hipError_t launch_dfa_place(const DfaView& d, const BatchView& b, const ScanOut& o, uint32_t n_super, const uint64_t* unit_offsets, int n_cu, Record* out, hipStream_t st)
{
    if (n_super == 0) return hipSuccess;
    hipLaunchKernelGGL(k_dfa_place, dim3(n_super), dim3(1024), 0, st, o.pool, o.block_next, n_super, unit_offsets, b.offsets, d.out, d.chunk, dfa_token_waves(d, b, n_cu), out);
    return hipGetLastError();
}

}  // namespace dev
}  // namespace am
