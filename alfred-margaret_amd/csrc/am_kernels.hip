// am_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X / CDNA4), wave64.
//
//  k_sf   "suffix filter" fast path (failureless Aho-Corasick, end-position parallel):
//         each wavefront streams 1 KiB of haystack per step with one coalesced 16 B load per lane,
//         builds the 16 four-byte suffix windows per lane in registers (v_alignbyte), probes a Bloom
//         filter of needle suffixes that lives in LDS (up to 128 KiB of the CU's 160 KiB), compacts
//         the surviving positions into a per-wave LDS queue with a wave prefix sum, probes them against
//         an L2-resident fingerprint table (requested in one chunk's iteration, looked at in the next)
//         and verifies the few survivors 64 at a time against one 64-byte table line each (+ the
//         reversed-needle trie for long needles).  Match records are compacted with ballot + popcount
//         into the unit's chain of pool blocks.
//  k_ac   general path: the reference's own packed automaton walked by one lane per chunk with a
//         warm-up overlap (the independent second algorithm of the parity gate; automata the suffix
//         structure cannot take).
//  k_hidx 1-KiB haystack index (position -> haystack id bracket).
//
// No MFMA anywhere: this is integer pointer chasing; the roofline is HBM bandwidth
// (1 B read per haystack byte + 16 B per record).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "am_bounds.h"
#include "am_config.h"
#include "am_device.h"
#include "am_wave.h"

AM_BOUNDS_TU("am_kernels.hip")

namespace am {
namespace dev {

// ------------------------------------------------------------------ haystack index

// (also clears up to two result / counter arrays of the call that follows, so that a one-document call needs no memset launches of its own)
__global__ void k_hidx(const uint64_t* __restrict__ offsets, uint32_t n_hay, uint64_t total, uint32_t* __restrict__ hidx, uint64_t n_entries,
                       uint32_t* __restrict__ z0, uint64_t n0, uint32_t* __restrict__ z1, uint64_t n1)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = k; i < n0; i += stride) z0[i] = 0u;
    for (uint64_t i = k; i < n1; i += stride) z1[i] = 0u;
    if (k >= n_entries) return;
    uint64_t p = k << kHidxShift;
    if (p > total - 1) p = total - 1;
    uint32_t lo = 0, hi = n_hay - 1;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo + 1u) / 2u;
        if (offsets[mid] <= p) lo = mid; else hi = mid - 1u;
    }
    hidx[k] = lo;
}

// ------------------------------------------------------------------ SF kernel

constexpr int kSfThreads = 1024;                 // 16 waves: with a 128 KiB filter one workgroup owns the CU
constexpr int kSfWaves = kSfThreads / 64;
// The LIGHT configuration for small batches (one document per call): workgroups of 4 wavefronts.  A 1024-thread workgroup needs 16 free
// wavefront slots on ONE CU at the same moment; next to another stream's kernel that keeps every CU busy with short workgroups it can
// wait for that kernel's grid to drain (seen in round 2: tests/test_multi.py), a 4-wavefront workgroup gets in like any of the others.
constexpr int kSfLightThreads = 256;
constexpr uint64_t kSfLightChunks = 16;          // batches of up to 16 KiB take the light configuration (every workgroup copies the filter into its LDS:
                                                 // four-wavefront workgroups on anything larger would copy it four times as often -- the Replacer's
                                                 // window scans of 50-250 KiB ran 15 % slower with a 256-KiB limit)
constexpr int kSfQ1 = 128;                       // per-wave queue of candidate positions (u16, offset in the chunk); more take several sub-passes
constexpr int kSfQ2 = 256;                       // per-wave ring of deferred positions (u16: chunk-in-unit << 12 | agreeing slot << 10 | offset): candidates that need the exact look
constexpr uint32_t kSfMaxUnitChunks = 64;
constexpr uint32_t kSfEpochChunks = 16;          // the ring is drained every 16 chunks, so that the chunk index fits the 4 bits an entry has for it
constexpr int kSfStage = 1056;                   // per-wave copy of the current chunk (folded): 8 bytes before it at offset 8, the chunk at 16, padding
constexpr uint32_t kSfMaskBytes = kBloomMasks * 4u;      // the Bloom mask table: first thing in LDS, the filter words follow

// ILP: 2 = a probe round always looks at two candidates per lane (up to 128 per round: automata with many needles, ~83 candidates per KiB on the
// benchmark text); 1 = rounds of at most 64 candidates look at one per lane -- half the probe's instructions -- chosen for automata with a small
// 4-byte-suffix table, which leave a handful of candidates per chunk (cfg2: +4.7 %; the same branches cost cfg3 1.5 %, hence two instantiations).
//
// Work unit = `unit_chunks` consecutive 1-KiB chunks.  A wavefront starts with unit (workgroup, wave) and draws every further
// one from a global counter: with a fixed stride the slowest wavefront (they differ by +-20 %: contents, the memory channels
// its units hit, its neighbours on the SIMD) finished 1.2x after the average one, and the launch lasts as long as the slowest.
// Structure of one wavefront's loop (everything between two filter steps is wave-synchronous):
//   filter   16 positions per lane against the LDS Bloom filter                      (LDS + VALU only)
//   compact  the survivors' offsets into the wave's queue (DPP prefix sum)
//   look at  the buckets requested for the PREVIOUS chunk's survivors; park the few that agree in a ring
//   request  both cuckoo buckets of this chunk's survivors, up to 128 at a time (window bytes from the
//            chunk staged in LDS); no data-dependent loop                             (phase 1)
//   resolve  when 64 are parked, and at the end of every 16-chunk epoch: one slot line per item, the
//            trie walk for the rare long ones                                         (phase 2)
// Phase 2 is FIFO, so the records of a unit come out in position order: ballot + popcount rank them,
// and they are appended to the unit's chain of 64-record pool blocks (blocks are taken 32 per atomic).
// LW: log2 of the filter size in words when it is the usual 128 KiB (15), so that the word address is a
// constant shift + constant mask (VOP2 with immediates issues at almost twice the rate of anything that
// reads an SGPR or needs the VOP3 encoding on gfx950, tools/microbench/valu_rates*.hip); 0 = any size
// DBG: the timing / ablation experiments (AM_SF_ABLATE) live in their own instantiations, the production kernel
// carries none of their code, registers or branches.
template <bool IC, int MODE, int ILP, int LW, bool SHORT, bool DBG, int NT = kSfThreads, bool CHILDREN = false>
__global__ __launch_bounds__(NT) void k_sf(SfView s, BatchView b, ScanOut o, uint64_t n_chunks)
{
    constexpr int kSfThreads = NT, kSfWaves = NT / 64;      // (shadow the namespace constants: the body is written against these names)
    constexpr bool kFlagMode = MODE == kModeAny || MODE == kModeIds;      // a flagged haystack is not looked at any further (containsAny: matched; containsAll: every needle seen)
    // CHILDREN: this instantiation looks for the five-byte child entries of heavy depth-4 nodes (am_image.h kT4Heavy).  It exists for <ILP 2, LW 0, no short
    // needles> only and is launched for images that have such entries: many needles behind a small LDS filter, i.e. few distinct 4-byte suffixes -- a
    // dictionary whose words share their endings (natural language: 100k words, 12k suffixes).  Every other instantiation carries none of it (this kernel's
    // hot loop pays for every instruction and register it does not need: compiled in behind a run-time flag the same code cost cfg2 20 %); a kernel that
    // ignores the entries defers every position of a heavy node, as before.
    constexpr bool CH = CHILDREN, children = CHILDREN;
    // chunks per ring epoch (the ring is drained at its end; an entry names its chunk within the epoch): flag mode drains every 4 chunks -- the first match is
    // what everybody waits for --, with child entries every 8 (bit 15 of an entry says "the slot of a child entry")
    constexpr uint32_t kEpoch = kFlagMode ? 4u : CH ? 8u : kSfEpochChunks;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t words = 1u << s.bloom_log2_words;
    uint32_t* masks = lds;                                            // LDS bytes [0, kSfMaskBytes)
    uint32_t* bloom = lds + kBloomMasks;                              // LDS bytes [kSfMaskBytes, kSfMaskBytes + 4 * words)
    uint8_t* stage_all = reinterpret_cast<uint8_t*>(bloom + words);
    uint16_t* q1_all = reinterpret_cast<uint16_t*>(stage_all + kSfWaves * kSfStage);
    uint16_t* q2_all = q1_all + kSfWaves * kSfQ1;

    for (uint32_t i = threadIdx.x; i < kBloomMasks; i += kSfThreads) masks[i] = bloom_mask_entry(i);
    for (uint32_t i = threadIdx.x; i < words; i += kSfThreads) bloom[i] = s.bloom[i];
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: unit / chunk arithmetic stays scalar
    // this wave's staging area and queues, as absolute LDS byte addresses (32-bit, uniform): no 64-bit pointer per structure
    const uint32_t stage = kSfMaskBytes + 4u * words + wave * (uint32_t)kSfStage;
    const uint32_t q1 = kSfMaskBytes + 4u * words + (uint32_t)kSfWaves * kSfStage + wave * (uint32_t)(kSfQ1 * 2);
    const uint32_t q2 = kSfMaskBytes + 4u * words + (uint32_t)kSfWaves * (kSfStage + kSfQ1 * 2) + wave * (uint32_t)(kSfQ2 * 2);
    // walker queue (only when the filter leaves room, i.e. small automata -- the ones that see match-dense text): wq_cap entries of 16 dwords
    const uint32_t wq_cap = o.wq_cap;
    const uint32_t wq = kSfMaskBytes + 4u * words + (uint32_t)kSfWaves * (kSfStage + kSfQ1 * 2 + kSfQ2 * 2) + wave * (wq_cap * 64u);
    uint32_t wq_n = 0;                                     // parked walkers (uniform)
    (void)stage_all; (void)q1_all; (void)q2_all;
    const uint64_t n_waves = (uint64_t)gridDim.x * kSfWaves;
    const uint32_t log2_words = s.bloom_log2_words, tiers = s.tiers;
    const uint32_t UC = o.unit_chunks;
    const uint64_t n_units = (n_chunks + UC - 1) / UC;
    uint64_t nval = 0;
    uint32_t q2_head = 0, q2_tail = 0;                   // monotonic; slot = index % kSfQ2
    // emit mode: state of the unit being written
    uint64_t unit_base_chunk = 0, epoch_base_chunk = 0;
    uint32_t unit_count = 0, unit_slots = 0, cur_block = kNone, first_block = kNone, grant_next = 0, grant_left = 0;   // unit_slots: record slots taken (found + parked walkers); unit_count: records
    bool pool_ok = true;

    // optional phase timing (AM_SF_ABLATE>=8): s_memtime deltas per wavefront, summed into o.dbg
    const bool timing = DBG && o.dbg != nullptr;
    const uint32_t ablate = DBG ? o.ablate : 0u;
    uint64_t t_filter = 0, t_compact = 0, t_probe = 0, t_resolve = 0, t_probe_pre = 0, t_mark = 0;
    uint64_t dbg_iters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t t_r0 = 0, t_r1 = 0, t_r2 = 0, t_r3 = 0, n_batches = 0, n_cand = 0, n_probes = 0, n_defer = 0, n_found = 0;
    auto tick = [&](uint64_t& acc) { if (timing) { const uint64_t now = __builtin_amdgcn_s_memtime(); acc += now - t_mark; t_mark = now; } };

    // ---- phase 2: resolve the oldest `nb` (<= 64 * RN) deferred items, RN per lane in lock step (all
    // belong to the current unit).  Item j of the batch sits in lane j % 64, slot j / 64, so ranks by
    // (slot, lane) reproduce the FIFO = position order.
    uint32_t cnt_hay = kNone; uint64_t cnt_val = 0;          // count mode: running per-haystack sum of this wave
    auto flush_count = [&]() {
        if (cnt_hay != kNone && cnt_val && lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(o.hay_counts + cnt_hay), (unsigned long long)cnt_val);
        cnt_val = 0;
    };
    // per-haystack counts (count mode): the found lanes of a batch almost always lie inside one haystack, then the wave adds its sum to a
    // running (haystack, count) pair that is flushed with ONE atomic when the haystack changes
    auto add_counts = [&](bool found, uint32_t hay, uint32_t vlen) {
        if (found) nval += vlen;
        const uint64_t fm = __ballot(found);
        if (!o.hay_counts || !fm) return;
        const uint32_t h0 = __shfl(hay, __ffsll((unsigned long long)fm) - 1, 64);
        if (__ballot(found && hay != h0) == 0) {
            const uint64_t sum = wave_sum_u64(found ? (uint64_t)vlen : 0ull);
            if (h0 != cnt_hay) { flush_count(); cnt_hay = h0; }
            cnt_val += sum;
        } else if (found) atomicAdd(reinterpret_cast<unsigned long long*>(o.hay_counts + hay), (unsigned long long)vlen);
    };
    // ---- the parked walkers: the newest <= 64 of them, walked to the end in lock step (all lanes busy; see resolve_batch)
    auto walk_parked = [&]() {
        const uint32_t n = wq_n < 64u ? wq_n : 64u, base = wq_n - n;
        wq_n = base;
        const bool valid = lane < n;
        const uint32_t e = wq + (base + (valid ? lane : 0u)) * 64u;
        const u32x4_n a0 = *reinterpret_cast<const lds_u32x4_t*>((uintptr_t)e), a1 = *reinterpret_cast<const lds_u32x4_t*>((uintptr_t)(e + 16u)),
                      a2 = *reinterpret_cast<const lds_u32x4_t*>((uintptr_t)(e + 32u)), a3 = *reinterpret_cast<const lds_u32x4_t*>((uintptr_t)(e + 48u));
        uint64_t gpos[1] = {((uint64_t)a0.y << 32) | a0.x};
        uint32_t avail[1] = {a0.z}, depth[1] = {a1.x}, best_state[1] = {a1.y}, best_vlen[1] = {a1.z}, w2[1] = {a1.w}, node[1] = {kNone};
        const uint32_t slot = a0.w, hay = a3.z, prior = a1.y;
        SfNode rec[1] = {SfNode{0, 0, a2.x, a2.y, {a2.z, a2.w, a3.x, a3.y}}};
        bool go[1] = {valid}, have_rec[1] = {true};
        wave_lds_fence();                                   // the entries are in registers before anything overwrites them
        uint32_t sel[1] = {a3.w};
        sf_resolve_walk<IC, 1>(s, b.text, gpos, avail, w2, go, node, rec, have_rec, depth, best_state, best_vlen, nullptr, 0xFFFFFFFFu, nullptr, sel);
        if (SHORT) {
            const bool vv[1] = {valid && !best_state[0]};
            uint32_t w[1] = {0}, dummy = 0;
            if (vv[0]) { load_suffix8(b.text, gpos[0], w[0], dummy); if (IC) w[0] = fold_dword(w[0]); }
            sf_resolve_short<1>(s, vv, avail, w, best_state, best_vlen);
        }
        const bool found = valid && best_state[0] != 0;
        if (timing) n_found += (uint32_t)__popcll(__ballot(found && prior == 0u));
        if (MODE == kModeCount) add_counts(found, hay, best_vlen[0]);
        else if (MODE == kModeEmit) {
            if (found && best_state[0] != prior && pool_ok) { AM_BOUNDS(slot < (uint64_t)o.n_blocks * kPoolBlock); o.pool[slot].state = best_state[0] - 1u; }      // the batch wrote end_pos and haystack into the walker's slot
            unit_count += (uint32_t)__popcll(__ballot(found && prior == 0u));
        } else if (MODE == kModeAny && found) atomicOr(reinterpret_cast<uint32_t*>(o.flags) + (hay >> 2), 1u << (8u * (hay & 3u)));      // (the flag modes have no walker queue: not reached)
    };

    // ---- phase 2: resolve the oldest `nb` (<= 64) deferred items in lock step (all belong to the current unit); item j of the batch
    // sits in lane j, so ranks by lane reproduce the FIFO = position order.
    //   head   haystack bytes + haystack index -> slot line + haystack start -> what the slot line settles (sf_resolve_head)
    //   walk   the items that go on into the trie.  A walk lasts as long as its deepest lane.  With a walker queue (wq_cap > 0: small
    //          filters, the automata that meet match-dense text) the batch only takes two steps; whoever is still walking then is parked
    //          in LDS with 16 dwords of state and walked later, 64 at a time.  (Natural text: 97 % of the items walk, 44 % finish after one
    //          step, 34 % after two, the deepest of 64 after six or seven: the batch used to wait for that one with 60 lanes idle.)
    // Emit mode: a record SLOT of the unit's block chain for every item that has a record or is parked; a parked walker that finds
    // nothing leaves state = kNone in its slot and k_permute drops it.  unit_slots counts slots, unit_count records.
    // RN items per lane.  The code below is written for any RN; 2 was measured for the small filters (LW == 0) in round 3: twice the loads in
    // flight per dependent step, half the batches -- but 128 VGPRs do not hold two items' slot lines next to the chunk loop's state (16
    // dwords spilled to scratch, some of them in the chunk loop; 43 KB of code instead of 27): cfg2 994 -> 726 GiB/s, natural text 63.5 -> 56.4.
    constexpr int RN = 1;
    // nb == 0 && drain: only walk what is parked (end of a unit).  walk_parked has ONE call site here (it contains a whole trie walk).
    auto resolve_batch = [&](uint32_t nb, bool drain) {
        __builtin_amdgcn_s_setprio(3);                       // (wave priorities: see the filter below)
        if (timing) { const uint64_t now = __builtin_amdgcn_s_memtime(); t_r0 += now - t_mark; t_mark = now; n_batches++; }
        uint64_t gpos[RN];
        uint32_t w2[RN], avail[RN], best_state[RN], best_vlen[RN], depth[RN], hay[RN], slot[RN], sel[RN];
        SfNode rec[RN];
        bool parked[RN];
        uint64_t pm[RN];
#pragma unroll
        for (int k = 0; k < RN; k++) {
            gpos[k] = 0; w2[k] = avail[k] = best_state[k] = best_vlen[k] = depth[k] = hay[k] = slot[k] = 0;
            rec[k] = SfNode{0, 0, 0, 0, {0, 0, 0, 0}}; parked[k] = false; pm[k] = 0; sel[k] = kSelUnknown;
        }
        if (nb) {
            bool valid[RN];
            uint32_t hint[RN], hlo[RN], hhi[RN];
            uint64_t end_pos[RN];
#pragma unroll
            for (int k = 0; k < RN; k++) {
                valid[k] = 64u * k + lane < nb;
                const uint32_t item = valid[k] ? lds_read_u16(q2 + 2u * ((q2_head + 64u * k + lane) % kSfQ2)) : 0u;
                gpos[k] = (epoch_base_chunk + ((item >> 12) & (kEpoch - 1u))) * kSfChunk + (item & 1023u);
                hint[k] = ((item >> 10) & 3u) | ((CH && children) ? (item >> 13) & 4u : 0u);
                hlo[k] = hhi[k] = 0; end_pos[k] = 0;
                if (valid[k]) { hlo[k] = b.hidx[gpos[k] >> kHidxShift]; hhi[k] = b.hidx[(gpos[k] >> kHidxShift) + 1]; }
            }
            // haystack index -> start offset: two dependent loads, made while the head's first two (haystack bytes -> slot line) are in flight
            auto locate = [&]() {
#pragma unroll
                for (int k = 0; k < RN; k++) {
                    hay[k] = hlo[k];
                    uint64_t start = valid[k] ? b.offsets[hlo[k]] : 0;
                    if (valid[k] && hlo[k] != hhi[k]) { hay[k] = find_haystack(b, gpos[k]); start = b.offsets[hay[k]]; }
                    end_pos[k] = valid[k] ? gpos[k] - start + 1 : 0;
                }
            };
            uint32_t w[RN], node[RN], t16[RN][4];
            bool go[RN], have_rec[RN], found[RN];
            sf_resolve_head<IC, RN>(s, b.text, gpos, end_pos, valid, hint, locate, w, w2, avail, best_state, best_vlen, depth, go, node, rec, have_rec, t16, timing ? dbg_iters : nullptr);
            if (ablate != 11) sf_resolve_walk<IC, RN>(s, b.text, gpos, avail, w2, go, node, rec, have_rec, depth, best_state, best_vlen, timing ? dbg_iters : nullptr, wq_cap ? o.wq_iters : 0xFFFFFFFFu, t16, sel);
#pragma unroll
            for (int k = 0; k < RN; k++) parked[k] = go[k] && valid[k];      // still walking after two steps (only with a walker queue)
            if (SHORT) {
                bool vv[RN];
#pragma unroll
                for (int k = 0; k < RN; k++) vv[k] = valid[k] && !parked[k];
                sf_resolve_short<RN>(s, vv, avail, w, best_state, best_vlen);
            }
#pragma unroll
            for (int k = 0; k < RN; k++) {
                found[k] = valid[k] && best_state[k] != 0;
                if (timing) n_found += (uint32_t)__popcll(__ballot(found[k]));
            }
            if (MODE == kModeCount) {
#pragma unroll
                for (int k = 0; k < RN; k++) add_counts(found[k] && !parked[k], hay[k], best_vlen[k]);      // a parked walker is counted when its walk is over
            } else if (MODE == kModeEmit && RN == 1) {
                const uint64_t take_mask = __ballot(found[0] || parked[0]);
                const uint32_t F = (uint32_t)__popcll(take_mask);
                if (F) {
                    const uint32_t r = unit_slots & (kPoolBlock - 1u);      // fill of the current block
                    const bool need_new = r == 0u || r + F > kPoolBlock;
                    uint32_t new_block = kNone;
                    if (need_new) {
                        // blocks are drawn from the pool kSfBlockGrant at a time: one atomic on the (single, device-wide) counter
                        // costs ~10 ns of serialised time, and match-dense text needs more than two blocks per KiB per wavefront
                        if (grant_left == 0) {
                            uint32_t g = 0;
                            if (lane == 0) g = atomicAdd(o.pool_ctrl, kSfBlockGrant);
                            grant_next = __builtin_amdgcn_readfirstlane(g);
                            grant_left = kSfBlockGrant;
                        }
                        const uint32_t id = grant_next++;
                        grant_left--;
                        if (id >= o.n_blocks) { pool_ok = false; if (lane == 0) o.pool_ctrl[1] = 1u; }   // keep counting, host retries with a larger pool
                        else {
                            new_block = id;
                            if (lane == 0) { o.block_next[id] = kNone; if (cur_block != kNone) o.block_next[cur_block] = id; }
                            if (first_block == kNone) first_block = id;
                        }
                    }
                    if ((found[0] || parked[0]) && pool_ok) {
                        const uint32_t p = r + (uint32_t)__popcll(take_mask & ((1ull << lane) - 1ull));
                        slot[0] = (r != 0u && p < kPoolBlock) ? cur_block * kPoolBlock + p : new_block * kPoolBlock + (r != 0u ? p - kPoolBlock : p);
                        AM_BOUNDS(slot[0] < (uint64_t)o.n_blocks * kPoolBlock && hay[0] < b.n_hay);
                        o.pool[slot[0]] = Record{end_pos[0], hay[0], found[0] ? best_state[0] - 1u : kNone};
                    }
                    if (need_new && pool_ok) cur_block = new_block;
                    unit_slots += F;
                    unit_count += (uint32_t)__popcll(__ballot(found[0]));
                }
            } else if (MODE == kModeEmit) {
                // RN items per lane: the slots of item 0 of all lanes, then those of item 1 (= position order); up to RN + 1 blocks join the chain
                uint64_t take[RN]; uint32_t before[RN], F = 0, n_found_recs = 0;
#pragma unroll
                for (int k = 0; k < RN; k++) {
                    take[k] = __ballot(found[k] || parked[k]); before[k] = F; F += (uint32_t)__popcll(take[k]);
                    n_found_recs += (uint32_t)__popcll(__ballot(found[k]));
                }
                if (F) {
                    const uint32_t s0 = unit_slots, have = (s0 + kPoolBlock - 1u) / kPoolBlock;      // blocks the chain has
                    const uint32_t n_new = (s0 + F + kPoolBlock - 1u) / kPoolBlock - have;         // <= RN + 1
                    const uint32_t prev_block = cur_block;          // takes the slots below `have` * 64 (when its last block is not full)
                    uint32_t ids[RN + 1];
#pragma unroll
                    for (int t = 0; t < RN + 1; t++) {
                        ids[t] = kNone;
                        if ((uint32_t)t < n_new) {
                            if (grant_left == 0) {
                                uint32_t g = 0;
                                if (lane == 0) g = atomicAdd(o.pool_ctrl, kSfBlockGrant);
                                grant_next = __builtin_amdgcn_readfirstlane(g);
                                grant_left = kSfBlockGrant;
                            }
                            const uint32_t id = grant_next++;
                            grant_left--;
                            if (id >= o.n_blocks) { pool_ok = false; if (lane == 0) o.pool_ctrl[1] = 1u; }   // keep counting, host retries with a larger pool
                            else if (pool_ok) {
                                ids[t] = id;
                                if (lane == 0) { o.block_next[id] = kNone; if (cur_block != kNone) o.block_next[cur_block] = id; }
                                if (first_block == kNone) first_block = id;
                                cur_block = id;
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < RN; k++) {
                        if ((found[k] || parked[k]) && pool_ok) {
                            const uint32_t si = s0 + before[k] + (uint32_t)__popcll(take[k] & ((1ull << lane) - 1ull));
                            const uint32_t q = si / kPoolBlock;
                            uint32_t blk = prev_block;
#pragma unroll
                            for (int t = 0; t < RN + 1; t++) if (q == have + (uint32_t)t) blk = ids[t];
                            slot[k] = blk * kPoolBlock + (si & (kPoolBlock - 1u));
                            AM_BOUNDS(slot[k] < (uint64_t)o.n_blocks * kPoolBlock && hay[k] < b.n_hay);
                            o.pool[slot[k]] = Record{end_pos[k], hay[k], found[k] ? best_state[k] - 1u : kNone};
                        }
                    }
                    unit_slots += F;
                    unit_count += n_found_recs;
                }
            } else if (MODE == kModeAny) {
#pragma unroll
                for (int k = 0; k < RN; k++) {
                    // ONE atomic per batch and haystack, and none for a haystack that is flagged already: on match-dense text every lane of every batch has found
                    // something, and 64 atomics on one word are served one after the other (~10 ns each: natural text, 2 048 haystacks: 103 ms for containsAny where
                    // counting every match took 24)
                    const uint64_t fm = __ballot(found[k]);
                    if (fm) {
                        const uint32_t h0 = __shfl(hay[k], __ffsll((unsigned long long)fm) - 1, 64);
                        const bool same = __ballot(found[k] && hay[k] != h0) == 0;
                        const bool mine = found[k] && (same ? lane == (uint32_t)__ffsll((unsigned long long)fm) - 1u : true);
                        if (mine) {
                            uint32_t* word = reinterpret_cast<uint32_t*>(o.flags) + (hay[k] >> 2);
                            const uint32_t bit = 1u << (8u * (hay[k] & 3u));
                            if (!(__atomic_load_n(word, __ATOMIC_RELAXED) & bit)) atomicOr(word, bit);      // (an atomic: the other XCDs' wavefronts look at it while the kernel runs)
                        }
                    }
                }
            } else {
                // containsAll (Searcher.hs:173-187): every needle id the state reports leaves the haystack's set -- here: its bit enters the haystack's row;
                // the bit that empties the set (`Done`, :181) raises the haystack's flag, and flagged haystacks are skipped like containsAny's
#pragma unroll
                for (int k = 0; k < RN; k++) {
                    if (found[k]) {
                        const uint32_t st = best_state[k] - 1u;
                        uint32_t* row = o.ids_bits + (uint64_t)hay[k] * o.ids_words;
                        for (uint64_t v = o.ids_vals_off[st], ve = o.ids_vals_off[st + 1]; v < ve; v++) {
                            const uint32_t id = o.ids_vals[v];
                            if (id >= o.ids_n) continue;                  // IS.delete of an absent key
                            const uint32_t bit = 1u << (id & 31u);
                            if (row[id >> 5] & bit) continue;             // (a stale read only costs an atomic)
                            if (!(atomicOr(row + (id >> 5), bit) & bit) && atomicSub(o.ids_missing + hay[k], 1u) == 1u)
                                atomicOr(reinterpret_cast<uint32_t*>(o.flags) + (hay[k] >> 2), 1u << (8u * (hay[k] & 3u)));
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < RN; k++) pm[k] = __ballot(parked[k]);
        }
        // park the walkers that are not done (rec = the record of the node they stand at), item 0's first; walk the queue when it cannot take
        // the next lot, when it holds a full batch, and -- drain -- to the end
        uint64_t pm_any = 0;
#pragma unroll
        for (int k = 0; k < RN; k++) pm_any |= pm[k];
        if (pm_any || (drain && wq_n)) {
            int next_k = 0;                                      // the item whose walkers are pushed next (RN: all pushed)
            for (;;) {
#pragma unroll
                for (int k = 0; k < RN; k++) {
                    if (next_k == k) {
                        const uint32_t n_park = (uint32_t)__popcll(pm[k]);
                        if (wq_n + n_park <= wq_cap) {           // (wq_cap >= 64 whenever a queue exists: after a walk there is room)
                            if (parked[k]) {
                                const uint32_t e = wq + (wq_n + (uint32_t)__popcll(pm[k] & ((1ull << lane) - 1ull))) * 64u;
                                AM_BOUNDS(e + 64u <= wq + wq_cap * 64u);
                                lds_write_u32x4(e, make_uint4((uint32_t)gpos[k], (uint32_t)(gpos[k] >> 32), avail[k], slot[k]));
                                lds_write_u32x4(e + 16u, make_uint4(depth[k], best_state[k], best_vlen[k], w2[k]));
                                lds_write_u32x4(e + 32u, make_uint4(rec[k].z, rec[k].w, rec[k].label[0], rec[k].label[1]));
                                lds_write_u32x4(e + 48u, make_uint4(rec[k].label[2], rec[k].label[3], hay[k], sel[k]));
                            }
                            wq_n += n_park;
                            wave_lds_fence();
                            next_k = k + 1;
                        }
                    }
                }
                if (next_k == RN && (drain ? wq_n == 0u : wq_n < 64u)) break;
                walk_parked();
                wave_lds_fence();
            }
        }
        if (timing) { const uint64_t now = __builtin_amdgcn_s_memtime(); t_r3 += now - t_mark; t_mark = now; }
        q2_head += nb;
        __builtin_amdgcn_s_setprio(2);
    };

    // ---- phase 1, second half: look at the buckets requested by the last probe round, park the survivors in the ring
    u32x2 p_a[2], p_b[2];                                  // the raw buckets (loads possibly still in flight)
    uint32_t p_e[2] = {0, 0}, p_pos[2] = {0, 0};           // the word a matching slot equals; offset in the chunk | 0x8000 (0: no candidate)
    uint32_t p_k5[2] = {0, 0}, p_e5[2] = {0, 0};           // CH: the candidates' five-byte key and the word a matching child entry equals
    uint32_t p_ci = 0;                                     // chunk (within its epoch) the round belongs to
    bool pending = false, p_two = false;                   // p_two: the round has more than 64 candidates (two per lane; else only item 0 is live)
    p_a[0] = p_a[1] = p_b[0] = p_b[1] = u32x2{0, 0};
    auto consume_round = [&]() {
        auto park = [&](bool defer, uint32_t hint, uint32_t pos) {
            const uint64_t m = __ballot(defer);
            // (CH: epochs of 8 chunks leave bit 15 for "the slot of a child entry", hint bit 2)
            AM_BOUNDS(q2_tail - q2_head + (uint32_t)__popcll(m) <= (uint32_t)kSfQ2);       // the ring of deferred positions is never overrun
            if (defer) lds_write_u16(q2 + 2u * ((q2_tail + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) % kSfQ2), (p_ci << 12) | ((hint & 3u) << 10) | (pos & 1023u) | (CH ? (hint & 4u) << 13 : 0u));      // (hint bit 2 is only ever set with child entries)
            q2_tail += (uint32_t)__popcll(m);
            if (timing) n_defer += (uint32_t)__popcll(m);
        };
        if (ILP == 2 || p_two) {
            bool valid[2], defer[2];
            uint32_t hint[2];
#pragma unroll
            for (int k = 0; k < 2; k++) valid[k] = (p_pos[k] & 0x8000u) != 0;
            sf_probe_decide<2>(s, p_a, p_b, p_e, valid, defer, hint);
            if (CH && children) {
                // candidates whose only agreeing slot is a heavy node's: their child entries decide -- two more buckets, requested now and waited for
                // (one more trip through L2 for the round; what it spares is a 64-byte slot line and a walk per rejected position)
                bool heavy[2];
                sf_probe_heavy<2>(s, p_a, p_b, p_e, defer, heavy);
                if (wave_any(heavy[0] || heavy[1])) sf_probe_children<2>(s, p_k5, p_e5, heavy, defer, hint);
            }
            if (ablate == 4) { defer[0] = false; defer[1] = false; }      // timing experiment only: no resolve
            park(defer[0], hint[0], p_pos[0]);
            park(defer[1], hint[1], p_pos[1]);
        } else {
            // at most 64 candidates (automata with few needles leave a handful per chunk): one per lane, half the work
            const u32x2 a1[1] = {p_a[0]}, b1[1] = {p_b[0]};
            const uint32_t e1[1] = {p_e[0]};
            const bool valid[1] = {(p_pos[0] & 0x8000u) != 0};
            bool defer[1]; uint32_t hint[1];
            sf_probe_decide<1>(s, a1, b1, e1, valid, defer, hint);
            if (CH && children) {
                bool heavy[1];
                sf_probe_heavy<1>(s, a1, b1, e1, defer, heavy);
                if (wave_any(heavy[0])) { const uint32_t k5[1] = {p_k5[0]}, e5[1] = {p_e5[0]}; sf_probe_children<1>(s, k5, e5, heavy, defer, hint); }
            }
            if (ablate == 4) defer[0] = false;
            park(defer[0], hint[0], p_pos[0]);
        }
        pending = false;
    };

    // software pipeline: the next chunk's 16 B per lane are requested before the current chunk is filtered and probed,
    // so HBM latency hides behind that work.  The bytes BEFORE a lane's 16 (its windows reach 3 bytes back, the probe
    // 5) come from the lane below with one DPP move; lane 0 takes them from `carry` = the last 8 (folded) bytes of the
    // chunk this wave processed just before, or -- at the start of a unit -- 8 bytes fetched from memory.
    auto fetch = [&](uint64_t cc, uint4& v) {
        const uint64_t p = cc * kSfChunk + lane * 16u;
        v = make_uint4(0, 0, 0, 0);
        if (cc < n_chunks && p < b.total) {
            typedef uint32_t u32x4_native __attribute__((ext_vector_type(4)));
            const u32x4_native* src = reinterpret_cast<const u32x4_native*>(b.text + p);
            const u32x4_native t = *src;                                            // global_load_dwordx4 (a non-temporal load changes neither the time nor the L2 miss count: measured)
            v = make_uint4(t.x, t.y, t.z, t.w);
        }
    };
    auto fetch_before = [&](uint64_t cc, uint32_t& c3, uint32_t& c4) {          // uniform: one request for the wave
        uint2 t = make_uint2(0, 0);
        if (cc < n_chunks && cc > 0) {
            t = *reinterpret_cast<const uint2*>(b.text + cc * kSfChunk - 8);
            t.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.x); t.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.y);   // waited for here, not at the join
        }
        c3 = IC ? fold_dword(t.x) : t.x; c4 = IC ? fold_dword(t.y) : t.y;
    };
    const uint64_t t_begin = timing ? __builtin_amdgcn_s_memtime() : 0;
    if (timing) t_mark = t_begin;
    uint64_t u = (uint64_t)blockIdx.x * kSfWaves + wave;
    uint64_t hs0 = 1, he0 = 0; uint32_t hay0 = 0;       // cached haystack bracket [hs0, he0) of haystack hay0: empty until the first lookup
    uint4 cur_v; uint32_t carry3, carry4;
    fetch(u * UC, cur_v);
    fetch_before(u * UC, carry3, carry4);
    // the first chunk's data is waited for HERE: if it were still pending at the loop header, the compiler's (static) wait at
    // the top of the loop body would also cover the prefetch of the next chunk that every iteration issues first
    asm volatile("" : "+v"(cur_v.x), "+v"(cur_v.y), "+v"(cur_v.z), "+v"(cur_v.w));

    // containsAny stops at the first match (Searcher.hs:156-164: `Done True`; Automaton.hs:528-532): in flag mode a wavefront looks at the flag
    // of the haystack its chunk lies in -- as the whole device has left it, one chunk ago: the load has a chunk's time -- and skips the chunk
    // if it is set.  A 1-GiB document that matches in its first KiB costs a few chunks per wavefront, not the scan.
    uint32_t any_word = 0, any_hay = kNone, flagged_hay = kNone;      // the flag word requested last, whose it is; the haystack known to be flagged

    uint64_t u_next = u;
    for (; u < n_units; u = u_next) {
        // the unit after this one (its first chunk is prefetched while this unit's last chunk is processed)
        u_next = u + n_waves;
        if (o.next_unit && !kFlagMode) {                // (flag mode draws at the unit's last chunk, when it knows whether it goes on at all)
            uint32_t ticket = 0;
            if (lane == 0) ticket = atomicAdd(o.next_unit, 1u);
            u_next = n_waves + (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
        }
        unit_base_chunk = u * UC;
        unit_count = 0; unit_slots = 0; cur_block = kNone; first_block = kNone;
        const uint32_t n_in_unit = (uint32_t)(unit_base_chunk + UC <= n_chunks ? UC : n_chunks - unit_base_chunk);
        if (kFlagMode && flagged_hay != kNone && hay0 == flagged_hay) {
            // flag mode: a unit that lies inside a haystack known to be flagged is not looked at at all (only the next unit's first bytes are
            // fetched); and when that haystack is the batch's last, the wavefront is done: units are handed out in ascending order, whatever it
            // would get from here on lies behind this one (a draw from the unit counter costs more than skipping a unit does)
            const uint64_t ub = unit_base_chunk * kSfChunk, ue0 = ub + (uint64_t)n_in_unit * kSfChunk, ue = ue0 < b.total ? ue0 : b.total;
            if (ub >= hs0 && he0 >= b.total) break;
            if (ub >= hs0 && ue <= he0) {
                fetch(u_next * UC, cur_v);
                fetch_before(u_next * UC, carry3, carry4);
                asm volatile("" : "+v"(cur_v.x), "+v"(cur_v.y), "+v"(cur_v.z), "+v"(cur_v.w));
                continue;
            }
        }
        bool jumped = false;                                  // flag mode: the loop went from a skipped chunk to the unit's last one
        for (uint32_t ci = 0; ci < n_in_unit; ci++) {
            const uint64_t c = unit_base_chunk + ci;
            if ((ci & (kEpoch - 1u)) == 0 && !jumped) epoch_base_chunk = c;      // (after a jump the ring may still hold items of the epoch that was left)
            uint4 next_v = make_uint4(0, 0, 0, 0);
            uint32_t next_c3 = 0, next_c4 = 0;
            const bool last_of_unit = ci + 1 >= n_in_unit;
            // flag mode: the flag word requested one chunk ago is looked at BEFORE the next prefetch goes out (loads return in order: a wait for
            // it after the prefetch would cover the prefetch).  A chunk inside a haystack that is flagged is skipped -- and so will the next one be
            // if it lies in the same haystack (known from the bracket of the chunk before): its bytes are not even requested.
            if (kFlagMode && any_hay != kNone) {
                const uint32_t seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)any_word);
                if (((seen >> (8u * (any_hay & 3u))) & 0xFFu) != 0u) flagged_hay = any_hay;
                any_hay = kNone;
            }
            if (kFlagMode && last_of_unit) {
                // the next unit: none when everything behind this one lies in a flagged haystack (the 1-GiB document that matches in its first KiB:
                // 4096 draws from the unit counter would be 45 us, the rest of the launch is 20)
                if (flagged_hay != kNone && flagged_hay == hay0 && he0 >= b.total && unit_base_chunk * kSfChunk >= hs0) u_next = n_units;
                else if (o.next_unit) {
                    uint32_t ticket = 0;
                    if (lane == 0) ticket = atomicAdd(o.next_unit, 1u);
                    u_next = n_waves + (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
                }
            }
            const bool next_skipped = kFlagMode && !last_of_unit && flagged_hay != kNone && flagged_hay == hay0 &&
                                      (c + 1) * kSfChunk >= hs0 && (c + 2) * kSfChunk <= he0;
            if (!next_skipped)
            fetch(!last_of_unit ? c + 1 : u_next * UC, next_v);
            if (last_of_unit) fetch_before(u_next * UC, next_c3, next_c4);

            const uint64_t c0 = c * kSfChunk;
            const uint64_t p0 = c0 + lane * 16u;
            // the haystack that contains the chunk's first byte is looked up only when the chunk leaves
            // the one found last time (same address in every lane: one request per load); almost every
            // chunk lies inside one haystack, then no candidate needs its own lookup either.
            // (The bracket goes through readfirstlane so that its loads are waited for INSIDE this rarely taken branch: a load
            // left pending at the join would make the compiler wait with vmcnt(0) on every chunk -- and vmcnt counts in order,
            // so that wait would also cover the prefetch issued just above and expose a full HBM latency per chunk.)
            if (c0 >= he0 || c0 < hs0) {
                hay0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b.hidx[c0 >> kHidxShift]);      // c0 is a multiple of 1 KiB: the index names its haystack (one load instead of a binary search)
                hs0 = uniform_u64(b.offsets[hay0]); he0 = uniform_u64(b.offsets[hay0 + 1]);
            }
            const bool single = (c0 + kSfChunk < b.total ? c0 + kSfChunk : b.total) <= he0;
            bool skip = false;
            if (kFlagMode) {
                skip = single && flagged_hay == hay0;
                // ... and when the rest of the unit lies in that haystack too, the loop goes straight to the unit's last chunk (which is skipped
                // like this one, drains what is pending and fetches the next unit's first chunk): a skipped chunk still costs half a filtered one
                if (skip && ci + 2 < n_in_unit) {
                    const uint64_t ue0 = (unit_base_chunk + n_in_unit) * kSfChunk;
                    if ((ue0 < b.total ? ue0 : b.total) <= he0) { ci = n_in_unit - 2u; jumped = true; }
                }
                // the flag of this chunk's haystack as the whole device has left it: asked for every 8th chunk (4096 wavefronts polling one word
                // on every chunk cost more than the scan: requests for one address are served one at a time), looked at one chunk later
                if (!skip && (ci & 7u) == 0u) {
                    any_word = __hip_atomic_load(reinterpret_cast<const uint32_t*>(o.flags) + (hay0 >> 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    any_hay = hay0;
                }
            }

            uint32_t d1 = cur_v.x, d2 = cur_v.y, d3 = cur_v.z, d4 = cur_v.w;
            if (IC) { d1 = fold_dword(d1); d2 = fold_dword(d2); d3 = fold_dword(d3); d4 = fold_dword(d4); }
            // the 4 bytes before the lane's 16: the lane below's last dword (wave_shr:1; lane 0 keeps `old` = the carry)
            const uint32_t d0 = (uint32_t)__builtin_amdgcn_update_dpp((int)carry4, (int)d4, 0x138, 0xf, 0xf, false);
            const uint32_t d[5] = {d0, d1, d2, d3, d4};
            lds_write_u32x4(stage + 16u + lane * 16u, make_uint4(d1, d2, d3, d4));
            if (lane == 0) lds_write_u32x2(stage + 8u, make_uint2(carry3, carry4));
            if (!last_of_unit) {                                  // the next chunk follows this one: its carry is this chunk's tail
                next_c3 = (uint32_t)__builtin_amdgcn_readlane((int)d3, 63);
                next_c4 = (uint32_t)__builtin_amdgcn_readlane((int)d4, 63);
            }
            // Wave priority: LOW only while a wavefront filters -- 16 positions of pure arithmetic and LDS reads -- and raised for everything
            // that issues or waits for memory requests (compaction, probe, prefetch: 2; the resolve's dependent trips: 3), so that the SIMD's
            // arbiter lets a wavefront that is about to put a load in flight go before one that only computes.  Measured in a same-box A/B:
            // k_sf 10.39 -> 10.26 ms per 10 GiB (cfg3 937 -> 949 GiB/s), cfg2 +1.7 %, cfg4 +0.9 %, natural text +0.2 %.
            __builtin_amdgcn_s_setprio(0);
            uint32_t cand = 0;
            if (!kFlagMode || !skip) {
                // tier 4 (needles of >= 4 bytes): straight-line code, the lane's 32 LDS reads (filter word + mask per
                // position) are all in flight before the first one is tested
                uint32_t h[16], v[16], m[16];
                const uint32_t sh_word = 32u - log2_words;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int j = k >> 2, sh = k & 3;      // window = bytes k-3..k of the lane's 16, newest byte on top
                    const uint32_t w = sh == 3 ? d[j + 1] : __builtin_amdgcn_alignbyte(d[j + 1], d[j], sh + 1);
                    h[k] = w * kBloomMul;
                }
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if (LW) v[k] = lds_read_u32(kSfMaskBytes + ((h[k] >> (30 - LW)) & (((1u << LW) - 1u) << 2)));
                    else v[k] = lds_read_u32(kSfMaskBytes + ((h[k] >> sh_word) << 2));
                    m[k] = lds_read_u32(h[k] & ((kBloomMasks - 1u) << 2));
                }
#pragma unroll
                for (int k = 15; k >= 0; k--) cand = (cand << 1) | (uint32_t)((v[k] & m[k]) == m[k]);      // bit k of cand = position k
            }
            if (SHORT && (!kFlagMode || !skip)) {    // automata with 1..3-byte needles: extra probes per position
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int j = k >> 2, sh = k & 3;
                    const uint32_t w = sh == 3 ? d[j + 1] : __builtin_amdgcn_alignbyte(d[j + 1], d[j], sh + 1);
                    if (sf_filter_short(bloom, log2_words, tiers, w, masks)) cand |= 1u << k;
                }
            }
            __builtin_amdgcn_s_setprio(2);
            if (p0 + 16 > b.total) cand &= p0 < b.total ? (1u << (uint32_t)(b.total - p0)) - 1u : 0u;
            if (ablate == 1) cand = 0;               // timing experiment only
            if (timing) { asm volatile("" :: "v"(cand)); tick(t_filter); }

            // Probe pipeline: the two hot buckets of a round's candidates are REQUESTED at the end of a chunk's iteration and LOOKED AT
            // after the next chunk has been filtered and compacted, so the ~3k cycles of an L2 round trip pass under that work instead
            // of stopping the wavefront (four in-order wavefronts per SIMD cannot hide them otherwise).
            for (bool first = true;; first = false) {
                // compact (up to kSfQ1) candidate positions into the wave's LDS queue, in position order
                const uint32_t n = __popc(cand);
                const uint32_t incl = wave_inclusive_sum(n, lane);
                const uint32_t total = __shfl(incl, 63, 64);
                if (total == 0 && !(first && pending)) break;
                uint32_t idx = incl - n;
                // the loop runs as long as the busiest lane has candidates, so it only queues positions; the probe
                // (dense, one candidate per lane) picks the window and the two bytes before it out of the staged chunk
                while (cand && idx < (uint32_t)kSfQ1) {
                    const uint32_t k = __builtin_ctz(cand);
                    cand &= cand - 1u;
                    AM_BOUNDS(idx < (uint32_t)kSfQ1);
                    lds_write_u16(q1 + 2u * idx++, lane * 16u + k);
                }
                const uint32_t n_q1 = total < (uint32_t)kSfQ1 ? total : (uint32_t)kSfQ1;
                if (timing) n_cand += n_q1;
                wave_lds_fence();
                tick(t_compact);
                // the round requested before this one (the previous chunk's last, or this chunk's previous 128 candidates)
                if (pending) {
                    consume_round();
                    wave_lds_fence();
                    tick(t_probe);
                    while (q2_tail - q2_head >= 64u * RN) { resolve_batch(64u * RN, false); wave_lds_fence(); }   // keeps room for the next round
                    tick(t_resolve);
                }
                // request this round: up to 128 survivors, two per lane
                if (n_q1 && ablate != 5) {
                    uint64_t avail[2] = {0, 0};
                    uint32_t w[2] = {0, 0}, nb[2] = {0, 0};
                    bool valid[2] = {false, false};
                    const bool two = ILP == 2 || n_q1 > 64u || o.probe_two;      // (uniform; ILP == 2: a constant, the branches below are not there)
                    p_pos[1] = 0u;
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        if (k == 1 && !two) break;
                        const uint32_t e = 64u * k + lane;
                        valid[k] = e < n_q1;
                        const uint32_t pos = valid[k] ? lds_read_u16(q1 + 2u * e) : 0u;
                        // bytes pos-5 .. pos of the staged chunk (stage offset 11 + pos): window = the last four (newest on
                        // top), nb = the two before it, nearest in bits 0-7
                        if (CH && children) {
                            // bytes pos-6 .. pos (stage offset 10 + pos): the THIRD byte before the window too -- a child entry may fix it
                            const uint32_t a = 10u + pos, sh = a & 3u;
                            const uint32_t sp = stage + (a & ~3u);
                            const uint32_t x0 = lds_read_u32(sp), x1 = lds_read_u32(sp + 4u), x2 = lds_read_u32(sp + 8u);
                            const uint32_t three = __builtin_amdgcn_alignbyte(x1, x0, sh) & 0xFFFFFFu;      // bytes pos-6, pos-5, pos-4
                            const uint32_t nbs3 = (three >> 16) | (three & 0xFF00u) | ((three & 0xFFu) << 16);      // nearest in bits 0-7
                            nb[k] = nbs3 & 0xFFFFu;
                            w[k] = sh < 1u ? __builtin_amdgcn_alignbyte(x1, x0, 3u) : __builtin_amdgcn_alignbyte(x2, x1, sh - 1u);
                            t4_child_inputs(s, w[k], nbs3, p_k5[k], p_e5[k]);
                        } else {
                        const uint32_t a = 11u + pos, sh = a & 3u;
                        const uint32_t sp = stage + (a & ~3u);
                        const uint32_t x0 = lds_read_u32(sp), x1 = lds_read_u32(sp + 4u), x2 = lds_read_u32(sp + 8u);
                        const uint32_t two = __builtin_amdgcn_alignbyte(x1, x0, sh) & 0xFFFFu;
                        nb[k] = (two >> 8) | ((two & 0xFFu) << 8);
                        w[k] = sh < 2u ? __builtin_amdgcn_alignbyte(x1, x0, sh + 2u) : __builtin_amdgcn_alignbyte(x2, x1, sh - 2u);
                        }
                        const uint64_t gpos = c0 + pos;
                        avail[k] = gpos - hs0 + 1;
                        if (valid[k] && !single) avail[k] = gpos - b.offsets[find_haystack(b, gpos)] + 1;
                        p_pos[k] = valid[k] ? (pos | 0x8000u) : 0u;
                    }
                    if (two) sf_probe_issue<2>(s, w, nb, avail, valid, p_a, p_b, p_e, ablate == 12);
                    else {
                        const uint32_t w1[1] = {w[0]}, nb1[1] = {nb[0]};
                        const uint64_t av1[1] = {avail[0]};
                        const bool v1[1] = {valid[0]};
                        u32x2 a1[1], b1[1]; uint32_t e1[1];
                        sf_probe_issue<1>(s, w1, nb1, av1, v1, a1, b1, e1, ablate == 12);
                        p_a[0] = a1[0]; p_b[0] = b1[0]; p_e[0] = e1[0];
                    }
                    p_two = two;
                    p_ci = ci & (kEpoch - 1u);
                    pending = true;
                    if (timing) { n_probes++; tick(t_probe_pre); }
                }
                if (total <= (uint32_t)kSfQ1) break;
                wave_lds_fence();
            }
            cur_v = next_v; carry3 = next_c3; carry4 = next_c4;
            // end of an epoch (and of the unit): drain the ring, so that every item of a batch belongs to one epoch of one unit
            if ((ci & (kEpoch - 1u)) == kEpoch - 1u || last_of_unit) {
                if (pending) consume_round();                 // not overlapped: once per 16 chunks
                wave_lds_fence();
                tick(t_compact);
                // (at the end of a unit the last call also walks every parked walker to the end: they all belong to this unit)
                // (flag mode walks them at the end of every epoch: a parked walker may be the match everybody is waiting for)
                const bool walk_all = last_of_unit || kFlagMode;
                while (q2_tail != q2_head || (walk_all && wq_n)) {
                    const uint32_t left = q2_tail - q2_head, nb = left < 64u * RN ? left : 64u * RN;
                    resolve_batch(nb, walk_all && left == nb);
                }
                tick(t_resolve);
            }
        }
        if (MODE == kModeEmit && lane == 0) { o.unit_counts[u] = unit_count; o.unit_first[u] = first_block; o.unit_slots[u] = unit_slots; }
    }
    if (timing && lane == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(o.dbg + 0), (unsigned long long)t_filter);
        atomicAdd(reinterpret_cast<unsigned long long*>(o.dbg + 1), (unsigned long long)t_compact);
        atomicAdd(reinterpret_cast<unsigned long long*>(o.dbg + 2), (unsigned long long)t_probe);
        atomicAdd(reinterpret_cast<unsigned long long*>(o.dbg + 3), (unsigned long long)t_resolve);
        atomicAdd(reinterpret_cast<unsigned long long*>(o.dbg + 4), 1ull);
        atomicAdd(reinterpret_cast<unsigned long long*>(o.dbg + 5), (unsigned long long)t_probe_pre);
        atomicAdd(reinterpret_cast<unsigned long long*>(o.dbg + 6), (unsigned long long)n_batches);
        atomicAdd(reinterpret_cast<unsigned long long*>(o.dbg + 7), (unsigned long long)t_r0);
        atomicAdd(reinterpret_cast<unsigned long long*>(o.dbg + 8), (unsigned long long)t_r1);
        atomicAdd(reinterpret_cast<unsigned long long*>(o.dbg + 9), (unsigned long long)t_r2);
        atomicAdd(reinterpret_cast<unsigned long long*>(o.dbg + 10), (unsigned long long)t_r3);
        for (int i = 0; i < 7; i++) atomicAdd(reinterpret_cast<unsigned long long*>(o.dbg + 16 + 2 * 8192 + i), (unsigned long long)dbg_iters[i]);
        atomicAdd(reinterpret_cast<unsigned long long*>(o.dbg + 11), (unsigned long long)n_cand);
        atomicAdd(reinterpret_cast<unsigned long long*>(o.dbg + 12), (unsigned long long)n_probes);
        atomicAdd(reinterpret_cast<unsigned long long*>(o.dbg + 13), (unsigned long long)n_defer);
        atomicAdd(reinterpret_cast<unsigned long long*>(o.dbg + 14), (unsigned long long)n_found);
        atomicMax(reinterpret_cast<unsigned long long*>(o.dbg + 15), (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin));
        const uint64_t gw = (uint64_t)blockIdx.x * kSfWaves + wave;
        if (gw < 8192) {                                   // per wavefront: duration and where it ran
            uint32_t hw_id, xcc_id;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
            o.dbg[16 + 2 * gw] = __builtin_amdgcn_s_memtime() - t_begin;
            o.dbg[17 + 2 * gw] = ((uint64_t)xcc_id << 32) | hw_id;
        }
    }

    if (MODE == kModeCount) {
        if (o.hay_counts) flush_count();
        nval = wave_sum_u64(nval);
        if (lane == 0 && nval) atomicAdd(reinterpret_cast<unsigned long long*>(o.total_values), (unsigned long long)nval);
    }
}

// copy every unit's chain of pool blocks to its final place (one wavefront per unit), dropping the slots that hold no record
// (state == kNone: a parked walker that found no needle end)
__global__ __launch_bounds__(256) void k_permute(ScanOut o, const uint64_t* __restrict__ unit_offsets, Record* __restrict__ out, uint64_t n_units)
{
    const uint64_t u = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (u >= n_units) return;
    const uint32_t n = o.unit_slots[u];
    uint64_t at = unit_offsets[u];
    uint32_t blk = o.unit_first[u];
    // (records move as one 16-byte vector each -- {end_pos lo, end_pos hi, haystack, state} -- and never as a struct temporary: the
    // compiler put a `Record` local into LDS, which made every launch read its dispatch packet from host memory: 6 -> 22 us per launch)
    const uint4* pool4 = reinterpret_cast<const uint4*>(o.pool);
    uint4* out4 = reinterpret_cast<uint4*>(out);
    for (uint32_t done = 0; done < n; done += kPoolBlock) {
        if (blk == kNone) break;                           // (a chain shorter than its slot count: only after a pool overflow, whose result the host discards)
        uint4 r = make_uint4(0u, 0u, 0u, kNone);
        if (done + lane < n) r = pool4[(uint64_t)blk * kPoolBlock + lane];
        const uint64_t m = __ballot(r.w != kNone);
        if (r.w != kNone) out4[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = r;
        at += (uint32_t)__popcll(m);
        blk = o.block_next[blk];
    }
}

// ------------------------------------------------------------------ launchers

hipError_t launch_hidx(const BatchView& b, uint32_t* hidx, uint64_t n_entries, hipStream_t st, uint32_t* z0, uint64_t n0, uint32_t* z1, uint64_t n1)
{
    const uint32_t blocks = (uint32_t)((n_entries + 255) / 256);      // >= 1: n_entries >= 2; the zero jobs are grid-stride loops
    hipLaunchKernelGGL(k_hidx, dim3(blocks), dim3(256), 0, st, b.offsets, b.n_hay, b.total, hidx, n_entries, z0, n0, z1, n1);
    return hipGetLastError();
}

uint64_t sf_chunks(const BatchView& b) { return (b.total + kSfChunk - 1) / kSfChunk; }

// chunks per work unit.  Big batches: 64-KiB units, four or more per wavefront, drawn from the unit counter (balance: see k_sf).  A draw is
// a device-scope atomic on ONE word and the memory side serves those one at a time, ~11 ns each (measured in round 4: 16384 one-chunk units
// of a 16-MiB batch took 0.196 ms, the same batch as 4096 four-chunk units without the counter takes a tenth of that; tools/any_exit.py).
// So a draw has to be paid for by >= 32 chunks of work: up to 64 chunks per wavefront (256 MiB on 256 CUs) every wavefront gets ONE unit and
// nothing is drawn; up to 256 the batch is cut into two or three units per wavefront of 33-64 chunks each.
uint32_t sf_unit_chunks(const BatchView& b, int n_cu)
{
    const uint64_t n_chunks = sf_chunks(b), waves = (uint64_t)(n_cu > 0 ? n_cu : 1) * kSfWaves;
    const uint64_t per_wave = (n_chunks + waves - 1) / waves;                     // chunks per wavefront
    const uint64_t k = (per_wave + kSfMaxUnitChunks - 1) / kSfMaxUnitChunks;      // units per wavefront
    if (k >= 4) return kSfMaxUnitChunks;      // (8 units per wavefront: -2 %, 16: -10 %: the unit counter's atomics)
    const uint64_t uc = k <= 1 ? per_wave : (n_chunks + waves * k - 1) / (waves * k);
    return (uint32_t)(uc < 1 ? 1 : uc);
}

hipError_t launch_permute(const ScanOut& o, const uint64_t* unit_offsets, Record* out, uint64_t n_units, hipStream_t st)
{
    if (n_units == 0) return hipSuccess;
    const uint64_t blocks = (n_units * 64 + 255) / 256;
    hipLaunchKernelGGL(k_permute, dim3((uint32_t)blocks), dim3(256), 0, st, o, unit_offsets, out, n_units);
    return hipGetLastError();
}
uint64_t ac_units(const AcView& a, const BatchView& b) { return (b.total + a.chunk - 1) / a.chunk; }

static size_t sf_lds_bytes_w(const SfView& s, int waves) { return kSfMaskBytes + ((size_t)4 << s.bloom_log2_words) + (size_t)waves * (kSfStage + kSfQ1 * sizeof(uint16_t) + kSfQ2 * sizeof(uint16_t)); }
// walker-queue entries per wavefront that fit next to the rest (0 when fewer than 64 would: the queue must take a whole batch)
static uint32_t sf_wq_cap(const SfView& s, int waves)
{
    const long forced = cfg::get(cfg::kSfWq);      // A/B: 0 = no queue
    const size_t base = sf_lds_bytes_w(s, waves), limit = 160 * 1024;
    if (base >= limit) return 0;
    size_t cap = (limit - base) / ((size_t)waves * 64);
    if (cap > 128) cap = 128;
    if (forced >= 0 && (size_t)forced < cap) cap = (size_t)forced;
    return cap >= 64 ? (uint32_t)cap : 0u;
}
size_t sf_lds_bytes(const SfView& s) { return sf_lds_bytes_w(s, kSfWaves); }

template <bool IC, int MODE, int ILP, int LW, bool SHORT, bool DBG = false, int NT = kSfThreads, bool CHILDREN = false>
static hipError_t launch_sf_v(const SfView& s, const BatchView& b, const ScanOut& o, int n_cu, hipStream_t st)
{
    constexpr int waves_per_wg = NT / 64;
    constexpr bool kFlagMode = MODE == kModeAny || MODE == kModeIds;
    const uint32_t wq_cap = kFlagMode ? 0u : sf_wq_cap(s, waves_per_wg);
    const size_t lds = sf_lds_bytes_w(s, waves_per_wg) + (size_t)waves_per_wg * wq_cap * 64;
    static bool attr_set = false;     // per instantiation
    if (!attr_set) {
        // allow the full 160 KiB of a CU's LDS as dynamic shared memory (not fatal if the runtime objects)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sf<IC, MODE, ILP, LW, SHORT, DBG, NT, CHILDREN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            (void)hipGetLastError();
        attr_set = true;
    }
    const uint64_t n_chunks = sf_chunks(b);
    const int per_cu = NT == kSfThreads ? (lds <= 80 * 1024 ? 2 : 1) : 8;          // light: up to 32 wavefronts per CU, like two full workgroups
    uint64_t blocks = (uint64_t)n_cu * per_cu;
    const uint64_t n_units = (n_chunks + o.unit_chunks - 1) / o.unit_chunks;
    const uint64_t need = (n_units + waves_per_wg - 1) / waves_per_wg;
    if (blocks > need) blocks = need;
    if (blocks == 0) return hipSuccess;
    ScanOut oo = o;
    oo.wq_cap = wq_cap;
    const long wqi = cfg::get(cfg::kSfWqIters);
    const uint32_t wq_iters = wqi >= 1 && wqi <= 16 ? (uint32_t)wqi : 2u;   // A/B: steps a batch walks before it parks
    oo.wq_iters = wq_iters;
    if (n_units <= blocks * waves_per_wg) oo.next_unit = nullptr;          // one unit per wavefront at most: nothing to draw
    // (else: *next_unit is zero -- it lives in the batch's 64-byte counter block, which every caller clears before the launch together with
    // its other counters; a memset of its own here was one more dispatch in every call)
    hipLaunchKernelGGL((k_sf<IC, MODE, ILP, LW, SHORT, DBG, NT, CHILDREN>), dim3((uint32_t)blocks), dim3(NT), lds, st, s, b, oo, n_chunks);
    return hipGetLastError();
}

static uint64_t* g_sf_dbg = nullptr;
// debug: per-phase s_memtime sums of k_sf launches run with AM_SF_ABLATE=9 (filter, compact, probe, resolve, waves)
hipError_t read_sf_wave_records(uint64_t* out, size_t n_waves)
{
    if (!g_sf_dbg) return hipErrorInvalidValue;
    if (n_waves == 0) { hipError_t e = hipMemcpy(out, g_sf_dbg + 16 + 2 * 8192, 64, hipMemcpyDeviceToHost); if (e == hipSuccess) e = hipMemset(g_sf_dbg + 16 + 2 * 8192, 0, 64); return e; }
    return hipMemcpy(out, g_sf_dbg + 16, 16 * (n_waves < 8192 ? n_waves : 8192), hipMemcpyDeviceToHost);
}

hipError_t read_sf_phase_cycles(uint64_t* out5)
{
    for (int i = 0; i < 16; i++) out5[i] = 0;
    if (!g_sf_dbg) return hipSuccess;
    hipError_t e = hipMemcpy(out5, g_sf_dbg, 128, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemset(g_sf_dbg, 0, 128);
    return e;
}

template <bool IC, int MODE>
static hipError_t launch_sf_t(const SfView& s, const BatchView& b, const ScanOut& o, int n_cu, hipStream_t st)
{
    const bool lw15 = s.bloom_log2_words == 15;
    constexpr bool kFlagMode = MODE == kModeAny || MODE == kModeIds;
    if constexpr (MODE == kModeIds) {
        // containsAll: two instantiations per case mode and workgroup size (any filter size, two candidates per lane) -- its time goes into the value lists
        // and the bitmap atomics of the matches, not into the filter the other variants are tuned for
        if (sf_chunks(b) <= kSfLightChunks)
            return (s.tiers & 7u) ? launch_sf_v<IC, MODE, 2, 0, true, false, kSfLightThreads>(s, b, o, n_cu, st) : launch_sf_v<IC, MODE, 2, 0, false, false, kSfLightThreads>(s, b, o, n_cu, st);
        return (s.tiers & 7u) ? launch_sf_v<IC, MODE, 2, 0, true>(s, b, o, n_cu, st) : launch_sf_v<IC, MODE, 2, 0, false>(s, b, o, n_cu, st);
    } else {
    if ((o.ablate || o.dbg) && !kFlagMode) {                                          // experiments (AM_SF_ABLATE)
        if (!lw15) return launch_sf_v<IC, MODE, 2, 0, true, true>(s, b, o, n_cu, st);
        return (s.tiers & 7u) ? launch_sf_v<IC, MODE, 2, 15, true, true>(s, b, o, n_cu, st) : launch_sf_v<IC, MODE, 2, 15, false, true>(s, b, o, n_cu, st);
    }
    if (sf_chunks(b) <= kSfLightChunks)                                                     // one small document: the light configuration
        return (s.tiers & 7u) ? launch_sf_v<IC, MODE, 2, 0, true, false, kSfLightThreads>(s, b, o, n_cu, st) : launch_sf_v<IC, MODE, 2, 0, false, false, kSfLightThreads>(s, b, o, n_cu, st);
    // few 4-byte-suffix keys (a hot table of <= 2^15 buckets: up to ~30k needles) = a handful of candidates per chunk: the one-per-lane probe round
    // a dictionary with heavy suffix nodes (the image has five-byte child entries): its own instantiation, two candidates per lane whatever the table's size
    // (such text leaves hundreds of candidates per chunk)
    if (!lw15 && s.t4_children && !(s.tiers & 7u) && !kFlagMode) return launch_sf_v<IC, MODE, 2, 0, false, false, kSfThreads, true>(s, b, o, n_cu, st);
    const bool few = s.tier_log2_cap[3] <= 15u && !o.probe_two;
    if (s.tiers & 7u) {                                                                     // needles shorter than 4 bytes present
        if (few) return lw15 ? launch_sf_v<IC, MODE, 1, 15, true>(s, b, o, n_cu, st) : launch_sf_v<IC, MODE, 1, 0, true>(s, b, o, n_cu, st);
        return lw15 ? launch_sf_v<IC, MODE, 2, 15, true>(s, b, o, n_cu, st) : launch_sf_v<IC, MODE, 2, 0, true>(s, b, o, n_cu, st);
    }
    if (few) return lw15 ? launch_sf_v<IC, MODE, 1, 15, false>(s, b, o, n_cu, st) : launch_sf_v<IC, MODE, 1, 0, false>(s, b, o, n_cu, st);
    return lw15 ? launch_sf_v<IC, MODE, 2, 15, false>(s, b, o, n_cu, st) : launch_sf_v<IC, MODE, 2, 0, false>(s, b, o, n_cu, st);
    }
}

hipError_t launch_sf(bool ic, int mode, const SfView& s, const BatchView& b, const ScanOut& o_in, int n_cu, hipStream_t st)
{
    ScanOut o = o_in;
    const uint32_t ablate = cfg::get(cfg::kSfAblate) > 0 ? (uint32_t)cfg::get(cfg::kSfAblate) : 0u;
    o.ablate = ablate;
    o.probe_two = cfg::on(cfg::kSfProbeTwo) ? 1u : 0u;
    static uint64_t* dbg = nullptr;
    if (ablate >= 8 && ablate != 12) {                     // (12: the one-bucket timing experiment, no phase sums)
        if (!dbg) { if (hipMalloc((void**)&dbg, 256 + 16 * 8192) != hipSuccess) dbg = nullptr; else (void)hipMemset(dbg, 0, 256 + 16 * 8192); }
        o.dbg = dbg;
        g_sf_dbg = dbg;
    }
    if (ic) {
        if (mode == kModeCount) return launch_sf_t<true, kModeCount>(s, b, o, n_cu, st);
        if (mode == kModeEmit) return launch_sf_t<true, kModeEmit>(s, b, o, n_cu, st);
        if (mode == kModeIds) return launch_sf_t<true, kModeIds>(s, b, o, n_cu, st);
        return launch_sf_t<true, kModeAny>(s, b, o, n_cu, st);
    }
    if (mode == kModeCount) return launch_sf_t<false, kModeCount>(s, b, o, n_cu, st);
    if (mode == kModeEmit) return launch_sf_t<false, kModeEmit>(s, b, o, n_cu, st);
    if (mode == kModeIds) return launch_sf_t<false, kModeIds>(s, b, o, n_cu, st);
    return launch_sf_t<false, kModeAny>(s, b, o, n_cu, st);
}

}  // namespace dev
}  // namespace am
