// am_sfx.hip -- k_sfx: the suffix-filter scan with ROLE-SPECIALISED wavefronts (gfx950, wave64).
//
// k_sf (am_kernels.hip) runs filter, compaction, probe and resolve in every wavefront.  Only the filter needs the 128-KiB LDS table
// that pins the occupancy at one 1024-thread workgroup per CU = four in-order wavefronts per SIMD, and those four cannot hide the
// probe's L2 round trips.  k_sfx keeps the SAME LDS filter, tables, record format and unit bookkeeping but
// splits the workgroup's 16 wavefronts into two roles:
//
//   F  12 wavefronts   stream the haystack (one coalesced 16-B load per lane and KiB), fold, filter against the LDS Bloom filter, compact
//                      the surviving positions and push (4-byte window, the two bytes before it, offset in the unit) entries into a ring
//                      in LDS -- no table access, no wait but for the prefetched stream.  The positions their P sends back ("a needle
//                      may end here") they resolve themselves, 64 at a time, exactly as k_sf does (sf_resolve_head / _walk of am_image.h:
//                      haystack bytes -> slot line -> trie), and write the records of their own units (block chains, unit_counts /
//                      unit_first / unit_slots as k_sf writes them).
//   P   4 wavefronts   one per SIMD; each serves three F rings: pop <= 64 entries, hash, request both hot cuckoo buckets, and look at
//                      them kProbeDepth passes later -- the requests of kProbeDepth rounds stay in flight (registers rotate by name, the
//                      compiler counts vmcnt statically: vmcnt(2 * kProbeDepth - 1)).  Survivors go back to their F's ring of deferred
//                      positions; a unit-end marker follows a unit's last survivor.
//
// (First version, measured: 12 F + 3 P + ONE resolver wavefront that owned all output -- bit-exact, and 2.3 x slower than k_sf: a
// resolve batch is a chain of dependent trips, ~35k cycles for 104 positions, and one wavefront per CU cannot take 40 positions per
// chunk time; the three Ps waited for it half of their time.  profiles/history/r04_sfx_v1_roles.log.)
//
// Hand-over is single-producer / single-consumer (F -> its P: candidate ring; P -> that F: deferred ring).  An F that waits for ring
// room serves its deferred ring meanwhile, so the two directions cannot block each other.  All waits are bounded (watchdog): a
// hand-over that does not move sets pool_ctrl[2] and every wavefront leaves.
//
// Used for the 128-KiB filter (large automata), needles of >= 4 bytes only, count and emit mode, batches large enough to fill the chip;
// everything else stays with k_sf.  Semantics: Automaton.hs:442-534 as for k_sf (one record per end position, position order per unit).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "am_device.h"
#include "am_wave.h"

namespace am {
namespace dev {

namespace {

constexpr int kXThreads = 1024;
constexpr int kXF = 12, kXP = 4, kXFperP = 3;         // wavefronts 0..11 filter (+ resolve), 12..15 probe
static_assert(kXF == kXP * kXFperP && kXF + kXP == kXThreads / 64, "role split");
constexpr int kProbeDepth = 4;                         // probe rounds whose bucket requests are in flight per P wavefront

constexpr uint32_t kXMaskBytes = kBloomMasks * 4u;
constexpr uint32_t kXBloomBytes = 4u << 15;            // the 128-KiB filter only
constexpr uint32_t kXStage = 1056;                     // per F: copy of the current chunk (folded): 8 bytes before it at offset 8, the chunk at 16, padding
constexpr uint32_t kXRing = 128;                       // per F: entries {window, nb << 16 | offset in the unit} (8 B), F -> P
constexpr uint32_t kXQ2 = 96;                          // per F: deferred positions / unit-end markers (u32), P -> F
constexpr uint32_t kXUq = 4;                           // units an F may have open or unfinished at a time
// per-F block: stage | ring | q2 | control
constexpr uint32_t kXOffRing = kXStage, kXOffQ2 = kXOffRing + kXRing * 8, kXOffCtrl = kXOffQ2 + kXQ2 * 4;
// control: +0 W0 = {tail, units closed | done << 31} (8 B, ONE store, by F)   +8 W1 = {ring head, q2 tail} (8 B, both by P)   +16 q2 head (by F)
//          +32 end_T[kXUq] (by F)   +48 unit ids of the unfinished units [kXUq] (F's own)
constexpr uint32_t kXFBytes = kXOffCtrl + 64;
constexpr uint32_t kXBase = kXMaskBytes + kXBloomBytes;
constexpr uint32_t kXAbort = kXBase + kXF * kXFBytes;  // one word: somebody's wait timed out
constexpr uint32_t kXLdsBytes = kXAbort + 16;
static_assert(kXLdsBytes <= 160 * 1024, "k_sfx LDS budget");
static_assert((kXOffCtrl & 7u) == 0 && (kXFBytes & 15u) == 0, "alignment of the control words");

constexpr uint32_t kDoneBit = 1u << 31;
constexpr uint32_t kQ2Mark = 1u << 31;                 // q2 entry: end of the F's oldest unfinished unit; else hint << 16 | offset in the unit
constexpr uint32_t kServe = 64;                        // an F resolves its deferred positions when this many wait (or a wait forces it)
constexpr uint32_t kSpinLimit = 1u << 21;
__device__ __forceinline__ uint32_t q2_slot(uint32_t i) { return i % kXQ2; }

// Ordering between the wavefronts of the workgroup concerns LDS only, and the LDS executes one wavefront's DS operations in the order they were
// issued.  So "publish after the data" is: the data stores, then the control store, in program order -- no wait in between -- and "read the data
// after the control word" is two loads in program order (the second usually needs the first one's value anyway).  What has to be stopped is the
// COMPILER moving them: a fence at wavefront scope emits no instruction.  (A workgroup-scope fence would wait for lgkmcnt(0) AND vmcnt(0) -- every
// global load in flight -- and the P wavefronts live on keeping four rounds of bucket requests in flight across these hand-overs.)
__device__ __forceinline__ void lds_order() { wave_lds_fence(); }
__device__ __forceinline__ uint32_t lds_ld(uint32_t byte_addr)
{
    return __hip_atomic_load(reinterpret_cast<lds_u32_t*>((uintptr_t)byte_addr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t lds_ld_acq(uint32_t byte_addr)      // uniform address -> scalar value (the readfirstlane waits for it); later LDS reads come after it
{
    const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_ld(byte_addr));
    lds_order();
    return r;
}
typedef __attribute__((address_space(3))) uint64_t lds_u64_t;
__device__ __forceinline__ uint64_t lds_ld64(uint32_t byte_addr)
{
    return __hip_atomic_load(reinterpret_cast<lds_u64_t*>((uintptr_t)byte_addr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st(uint32_t byte_addr, uint32_t v) { __hip_atomic_store(reinterpret_cast<lds_u32_t*>((uintptr_t)byte_addr), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// lane 0 publishes a control word after everything this wavefront wrote to LDS before
__device__ __forceinline__ void lds_st_rel(uint32_t byte_addr, uint32_t v, uint32_t lane)
{
    lds_order();
    if (lane == 0) lds_st(byte_addr, v);
}
__device__ __forceinline__ void lds_st64_rel(uint32_t byte_addr, uint32_t lo, uint32_t hi, uint32_t lane)
{
    lds_order();
    if (lane == 0) __hip_atomic_store(reinterpret_cast<lds_u64_t*>((uintptr_t)byte_addr), ((uint64_t)hi << 32) | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The kernel's arguments as they lie in the kernarg segment: the P role (a function of its own) reads them from there (scalar loads).
struct XKArgs { SfView s; BatchView b; ScanOut o; uint64_t n_chunks; };
// What an F wavefront's resolve step reads and updates
struct XServe {
    uint32_t unit_count, unit_slots, cur_block, first_block, grant_next, grant_left, Hq2, n_fin;
    bool pool_ok;
    uint64_t nval, d_batches, d_items, d_found, d_resolve;
};

// ---- an F wavefront's deferred ring: resolve up to 64 positions of the oldest unfinished unit in lock step (item j in lane j = position order),
// write their records; a marker ends the unit.  min_batch: nothing is done for fewer waiting entries; force: batch after batch until fewer
// are left (else one batch).  Inlined at its ONE call site (the resolve is ~1500 instructions; the instruction cache is shared by the CU's wavefronts).
template <bool IC, int MODE, bool DBG>
__device__ __forceinline__ void sfx_serve(const SfView& s, const BatchView& b, const ScanOut& o, uint32_t fblock, uint32_t lane, uint32_t min_batch, bool force, XServe& st)
{
    const uint32_t q2 = fblock + kXOffQ2, ctrl = fblock + kXOffCtrl;
    const uint64_t unit_bytes = (uint64_t)o.unit_chunks * kSfChunk;
    for (;;) {
        const uint32_t n_wait = lds_ld_acq(ctrl + 12u) - st.Hq2;
        if (n_wait == 0 || n_wait < min_batch) break;
        uint32_t m = n_wait < 64u ? n_wait : 64u;
        const uint32_t e = lane < m ? lds_ld(q2 + q2_slot(st.Hq2 + lane) * 4u) : 0u;
        const uint64_t marks = __ballot(lane < m && (e & kQ2Mark) != 0u);
        bool unit_ends = false;
        if (marks) { m = (uint32_t)__builtin_ctzll(marks) + 1u; unit_ends = true; }      // the entries behind a marker belong to the next unit
        const uint32_t n_items = unit_ends ? m - 1u : m;
        const uint32_t ru = lds_ld_acq(ctrl + 48u + (st.n_fin & (kXUq - 1u)) * 4u);        // the unit these positions lie in
        if (n_items) {
            __builtin_amdgcn_s_setprio(3);
            const uint64_t t0 = DBG ? __builtin_amdgcn_s_memtime() : 0;
            bool valid[1] = {lane < n_items};
            uint64_t gpos[1] = {(uint64_t)ru * unit_bytes + (e & 0xFFFFu)};
            const uint32_t hint[1] = {(e >> 16) & 3u};
            uint32_t hlo = 0, hhi = 0, hay = 0;
            uint64_t end_pos[1] = {0};
            if (valid[0]) { hlo = b.hidx[gpos[0] >> kHidxShift]; hhi = b.hidx[(gpos[0] >> kHidxShift) + 1]; }
            auto locate = [&]() {      // haystack index -> start offset, while the head's first two trips (haystack bytes -> slot line) are in flight
                hay = hlo;
                uint64_t start = valid[0] ? b.offsets[hlo] : 0;
                if (valid[0] && hlo != hhi) { hay = find_haystack(b, gpos[0]); start = b.offsets[hay]; }
                end_pos[0] = valid[0] ? gpos[0] - start + 1 : 0;
            };
            uint32_t w[1], w2[1], avail[1], best_state[1], best_vlen[1], depth[1], node[1], t16[1][4];
            bool go[1], have_rec[1];
            SfNode rec[1];
            sf_resolve_head<IC, 1>(s, b.text, gpos, end_pos, valid, hint, locate, w, w2, avail, best_state, best_vlen, depth, go, node, rec, have_rec, t16);
            sf_resolve_walk<IC, 1>(s, b.text, gpos, avail, w2, go, node, rec, have_rec, depth, best_state, best_vlen, nullptr, 0xFFFFFFFFu, t16);
            const bool found = valid[0] && best_state[0] != 0;
            if (DBG) { st.d_resolve += __builtin_amdgcn_s_memtime() - t0; st.d_batches++; st.d_items += n_items; st.d_found += (uint32_t)__popcll(__ballot(found)); }
            if (MODE == kModeCount) {
                if (found) {
                    st.nval += best_vlen[0];
                    if (o.hay_counts) atomicAdd(reinterpret_cast<unsigned long long*>(o.hay_counts + hay), (unsigned long long)best_vlen[0]);
                }
            } else {
                const uint64_t take_mask = __ballot(found);
                const uint32_t Fn = (uint32_t)__popcll(take_mask);
                if (Fn) {
                    const uint32_t r = st.unit_slots & (kPoolBlock - 1u);      // fill of the current block
                    const bool need_new = r == 0u || r + Fn > kPoolBlock;
                    uint32_t new_block = kNone;
                    if (need_new) {
                        // blocks are drawn from the pool kSfBlockGrant at a time: one atomic on the (single, device-wide) counter costs ~10 ns
                        if (st.grant_left == 0) {
                            uint32_t g = 0;
                            if (lane == 0) g = atomicAdd(o.pool_ctrl, kSfBlockGrant);
                            st.grant_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
                            st.grant_left = kSfBlockGrant;
                        }
                        const uint32_t id = st.grant_next++;
                        st.grant_left--;
                        if (id >= o.n_blocks) { st.pool_ok = false; if (lane == 0) o.pool_ctrl[1] = 1u; }   // keep counting, the host retries with a larger pool
                        else {
                            new_block = id;
                            if (lane == 0) { o.block_next[id] = kNone; if (st.cur_block != kNone) o.block_next[st.cur_block] = id; }
                            if (st.first_block == kNone) st.first_block = id;
                        }
                    }
                    if (found && st.pool_ok) {
                        const uint32_t p = r + (uint32_t)__popcll(take_mask & ((1ull << lane) - 1ull));
                        const uint32_t slot = (r != 0u && p < kPoolBlock) ? st.cur_block * kPoolBlock + p : new_block * kPoolBlock + (r != 0u ? p - kPoolBlock : p);
                        reinterpret_cast<uint4*>(o.pool)[slot] = make_uint4((uint32_t)end_pos[0], (uint32_t)(end_pos[0] >> 32), hay, best_state[0] - 1u);
                    }
                    if (need_new && st.pool_ok) st.cur_block = new_block;
                    st.unit_slots += Fn; st.unit_count += Fn;
                }
            }
            // nothing of the resolve stays in flight where the paths join again: a load whose last dword is never looked at would otherwise stay
            // "pending" for the compiler, and the next use of its register -- in the filter loop -- would be preceded by a wait for every request
            // in flight, the chunk prefetch included
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_setprio(0);
        }
        if (unit_ends) {
            if (MODE == kModeEmit && lane == 0) { o.unit_counts[ru] = st.unit_count; o.unit_first[ru] = st.first_block; o.unit_slots[ru] = st.unit_slots; }
            st.unit_count = 0; st.unit_slots = 0; st.cur_block = kNone; st.first_block = kNone;
            st.n_fin++;
        }
        st.Hq2 += m;
        lds_st_rel(ctrl + 16u, st.Hq2, lane);              // the P may overwrite these entries
        if (!force) break;
    }
}

// ================================================================================= F: filter, and resolve what comes back
// (inlined into the kernel; the P role and the resolve are FUNCTIONS: each gets its own register allocation -- with all three in one body
// their uniform values competed for the 100-odd SGPRs and the filter loop was full of v_readlane / v_writelane spill traffic.  The filter
// role itself must not be a function: what is live across its calls of sfx_serve would then be spilled to scratch inside the chunk loop.)
template <bool IC, int MODE, bool DBG>
__device__ __forceinline__ void sfx_role_filter(const SfView& s, const BatchView& b, const ScanOut& o, uint64_t n_chunks, uint32_t f, uint32_t lane, uint64_t t_begin)
{
    const uint32_t UC = o.unit_chunks;
    const uint64_t n_units = (n_chunks + UC - 1) / UC;
    const uint32_t stage = kXBase + f * kXFBytes, ring = stage + kXOffRing, ctrl = stage + kXOffCtrl;
    const uint64_t n_f = (uint64_t)gridDim.x * kXF;
    uint32_t T = 0, n_closed = 0;                          // entries pushed so far; units closed so far
    uint32_t n_open = 0;                                   // units started
    uint32_t head = 0;                                     // the ring's head as last seen (re-read only when it leaves no room)
    bool ok = true;
    XServe st{0u, 0u, kNone, kNone, 0u, 0u, 0u, 0u, true, 0ull, 0ull, 0ull, 0ull, 0ull};
    uint64_t d_wait_ring = 0, d_wait_uq = 0, d_chunks = 0, d_cands = 0;
    __builtin_amdgcn_s_setprio(0);
    auto aborted = [&]() -> bool { return lds_ld_acq(kXAbort) != 0u; };
    auto give_up = [&]() { if (lane == 0) lds_st(kXAbort, 1u); };
    auto fetch = [&](uint64_t cc, u32x4_n& v) {
        const uint64_t p = cc * kSfChunk + lane * 16u;
        v = u32x4_n{0, 0, 0, 0};
        if (cc < n_chunks && p < b.total) v = *reinterpret_cast<const u32x4_n*>(b.text + p);      // global_load_dwordx4
    };
    auto fetch_before = [&](uint64_t cc, uint32_t& c3, uint32_t& c4) {          // the 8 (folded) bytes before chunk cc; uniform
        uint2 t = make_uint2(0, 0);
        if (cc < n_chunks && cc > 0) {
            t = *reinterpret_cast<const uint2*>(b.text + cc * kSfChunk - 8);
            t.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.x); t.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.y);
        }
        c3 = IC ? fold_dword(t.x) : t.x; c4 = IC ? fold_dword(t.y) : t.y;
    };
    // The loop over units runs ONE extra, empty trip at the end (`drain`): its only chunk has no candidates and its hand-over step waits until
    // the last unit's marker has come back.  Every wait of this wavefront -- for ring room, for a free unit slot, for the last markers -- is
    // that one hand-over step, and sfx_serve has exactly ONE call site in it.
    uint64_t u = (uint64_t)blockIdx.x * kXF + f;
    u32x4_n cur_v; uint32_t carry3, carry4;
    fetch(u * UC, cur_v);
    fetch_before(u * UC, carry3, carry4);
    asm volatile("" : "+v"(cur_v));
    uint64_t u_next = u;
    for (bool drain = false; ok && !drain; u = u_next) {
        drain = u >= n_units;
        if (!drain) {
            u_next = u + n_f;
            if (o.next_unit) {
                uint32_t ticket = 0;
                if (lane == 0) ticket = atomicAdd(o.next_unit, 1u);
                u_next = n_f + (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
            }
        } else lds_st64_rel(ctrl, T, n_closed | kDoneBit, lane);      // nothing more will be pushed or closed
        const uint64_t unit_base_chunk = u * UC;
        const uint32_t n_in_unit = drain ? 1u : (uint32_t)(unit_base_chunk + UC <= n_chunks ? UC : n_chunks - unit_base_chunk);
        bool unit_started = false;
        for (uint32_t ci = 0; ok && ci < n_in_unit; ci++) {
            const uint64_t c = unit_base_chunk + ci;
            u32x4_n next_v = u32x4_n{0, 0, 0, 0};
            uint32_t next_c3 = 0, next_c4 = 0;
            const bool last_of_unit = ci + 1 >= n_in_unit;
            uint32_t cand = 0;
            if (!drain) {
                // everything requested so far has to be here now anyway (this chunk's bytes were requested one trip ago); saying so BEFORE the
                // next request goes out keeps the compiler from waiting for "all requests" -- the prefetch included -- further down
                __builtin_amdgcn_s_waitcnt(0x0F70);
                fetch(!last_of_unit ? c + 1 : u_next * UC, next_v);
                if (last_of_unit) fetch_before(u_next * UC, next_c3, next_c4);
                const uint64_t p0 = c * kSfChunk + lane * 16u;
                uint32_t d1 = cur_v.x, d2 = cur_v.y, d3 = cur_v.z, d4 = cur_v.w;
                if (IC) { d1 = fold_dword(d1); d2 = fold_dword(d2); d3 = fold_dword(d3); d4 = fold_dword(d4); }
                const uint32_t d0 = (uint32_t)__builtin_amdgcn_update_dpp((int)carry4, (int)d4, 0x138, 0xf, 0xf, false);      // the lane below's last dword (lane 0: the carry)
                const uint32_t d[5] = {d0, d1, d2, d3, d4};
                lds_write_u32x4(stage + 16u + lane * 16u, make_uint4(d1, d2, d3, d4));
                if (lane == 0) lds_write_u32x2(stage + 8u, make_uint2(carry3, carry4));
                if (!last_of_unit) {
                    next_c3 = (uint32_t)__builtin_amdgcn_readlane((int)d3, 63);
                    next_c4 = (uint32_t)__builtin_amdgcn_readlane((int)d4, 63);
                }
                uint32_t h[16], v[16], m[16];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int j = k >> 2, sh = k & 3;      // window = bytes k-3..k of the lane's 16, newest byte on top
                    const uint32_t w = sh == 3 ? d[j + 1] : __builtin_amdgcn_alignbyte(d[j + 1], d[j], sh + 1);
                    h[k] = w * kBloomMul;
                }
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    v[k] = lds_read_u32(kXMaskBytes + ((h[k] >> 15) & (((1u << 15) - 1u) << 2)));
                    m[k] = lds_read_u32(h[k] & ((kBloomMasks - 1u) << 2));
                }
#pragma unroll
                for (int k = 15; k >= 0; k--) cand = (cand << 1) | (uint32_t)((v[k] & m[k]) == m[k]);
                if (p0 + 16 > b.total) cand &= p0 < b.total ? (1u << (uint32_t)(b.total - p0)) - 1u : 0u;
            }
            // ---- compaction + hand-over, up to 128 candidates per pass.  The ring slots [T, T + n) are reserved first; the compaction loop writes
            // each candidate's offset into its slot, the dense step (two slots per lane) replaces it by the entry: no queue in between.  A unit's
            // first chunk always goes through the hand-over step (with nothing to push, if it comes to that): that is where the unit takes its slot
            for (;;) {
                const uint32_t n = __popc(cand);
                const uint32_t incl = wave_inclusive_sum(n, lane);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (total == 0 && unit_started) break;
                const uint32_t n_p = total < 128u ? total : 128u;
                if (DBG) d_cands += n_p;
                bool pushed = n_p == 0u;
                uint32_t spins = 0;
                for (;;) {
                    if (!unit_started && !drain && n_open - st.n_fin < kXUq) {
                        if (lane == 0) lds_st(ctrl + 48u + (n_open & (kXUq - 1u)) * 4u, (uint32_t)u);
                        n_open++;
                        unit_started = true;
                    }
                    if (unit_started && !pushed && T + n_p - head > kXRing) head = lds_ld_acq(ctrl + 8u);      // no room by the head last seen: look again
                    if (unit_started && !pushed && T + n_p - head <= kXRing) {
                        uint32_t idx = incl - n;
                        while (cand && idx < 128u) {
                            const uint32_t k = __builtin_ctz(cand);
                            cand &= cand - 1u;
                            lds_write_u16(ring + ((T + idx++) & (kXRing - 1u)) * 8u + 4u, lane * 16u + k);
                        }
                        lds_order();
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                            const uint32_t e = 64u * k + lane;
                            if (e < n_p) {
                                const uint32_t slot = ring + ((T + e) & (kXRing - 1u)) * 8u;
                                const uint32_t pos = lds_read_u16(slot + 4u);
                                // bytes pos-5 .. pos of the staged chunk (stage offset 11 + pos): window = the last four (newest on top), nb = the two before it
                                const uint32_t a = 11u + pos, sh = a & 3u;
                                const uint32_t sp = stage + (a & ~3u);
                                const uint32_t x0 = lds_read_u32(sp), x1 = lds_read_u32(sp + 4u), x2 = lds_read_u32(sp + 8u);
                                const uint32_t two = __builtin_amdgcn_alignbyte(x1, x0, sh) & 0xFFFFu;
                                const uint32_t nb = (two >> 8) | ((two & 0xFFu) << 8);
                                const uint32_t ew = sh < 2u ? __builtin_amdgcn_alignbyte(x1, x0, sh + 2u) : __builtin_amdgcn_alignbyte(x2, x1, sh - 2u);
                                lds_write_u32x2(slot, make_uint2(ew, (nb << 16) | ((ci << 10) + pos)));
                            }
                        }
                        T += n_p;
                        lds_st64_rel(ctrl, T, n_closed, lane);
                        pushed = true;
                    }
                    const bool complete = drain ? st.n_fin == n_open : (unit_started && pushed);
                    // the deferred ring.  Every fourth chunk: a batch worth its dependent trips (kServe).  Stuck for ring room: the P is behind --
                    // half a batch, and in any case before the deferred ring is full (the P may be waiting for room THERE: the two directions
                    // must not wait for each other).  Stuck for a unit slot, or draining: whatever is there -- the marker is what is waited for.
                    if (!complete || (ci & 3u) == 3u) {
                        const bool for_marker = !complete && (drain || !unit_started);
                        sfx_serve<IC, MODE, DBG>(s, b, o, stage, lane, complete ? kServe : (for_marker ? 1u : kServe / 2u), for_marker, st);      // (the one call site)
                    }
                    if (complete) break;
                    // stuck: a short sleep; the watchdog
                    const uint64_t t0 = DBG ? __builtin_amdgcn_s_memtime() : 0;
                    __builtin_amdgcn_s_sleep(1);
                    if (DBG) { if (unit_started) d_wait_ring += __builtin_amdgcn_s_memtime() - t0; else d_wait_uq += __builtin_amdgcn_s_memtime() - t0; }
                    if (++spins > kSpinLimit) { give_up(); ok = false; break; }
                    if ((spins & 63u) == 0 && aborted()) { ok = false; break; }
                }
                if (!ok || drain || total <= 128u) break;
                lds_order();
            }
            // the prefetched bytes are taken over HERE, after the hand-over (left to itself the scheduler moves these copies into the middle of
            // the filter, with a wait for the request issued a few hundred instructions earlier)
            asm volatile("" : "+v"(next_v));
            cur_v = next_v; carry3 = next_c3; carry4 = next_c4;
            if (DBG && !drain) d_chunks++;
        }
        if (!drain && ok) {
            // close the unit: everything up to T belongs to it
            if (lane == 0) lds_st(ctrl + 32u + (n_closed & (kXUq - 1u)) * 4u, T);
            n_closed++;
            lds_st64_rel(ctrl, T, n_closed, lane);
        }
    }
    if (lds_ld_acq(kXAbort) != 0u && lane == 0 && o.pool_ctrl) o.pool_ctrl[2] = 1u;
    if (MODE == kModeCount) {
        const uint64_t sum = wave_sum_u64(st.nval);
        if (lane == 0 && sum) atomicAdd(reinterpret_cast<unsigned long long*>(o.total_values), (unsigned long long)sum);
    }
    if (DBG && o.dbg && lane == 0) {
        unsigned long long* q = reinterpret_cast<unsigned long long*>(o.dbg + 32);
        atomicAdd(q + 0, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin)); atomicAdd(q + 1, (unsigned long long)d_wait_ring);
        atomicAdd(q + 2, (unsigned long long)d_wait_uq); atomicAdd(q + 3, (unsigned long long)d_chunks); atomicAdd(q + 4, (unsigned long long)d_cands);
        atomicAdd(q + 18, (unsigned long long)st.d_batches); atomicAdd(q + 19, (unsigned long long)st.d_items); atomicAdd(q + 20, (unsigned long long)st.d_found);
        atomicAdd(q + 21, (unsigned long long)st.d_resolve);
    }
}

// ================================================================================= P: probe
// (its arguments by value: the few words it needs.  Handed a pointer to the kernel's arguments it read them -- and the buckets -- with FLAT
// loads, which count under vmcnt AND lgkmcnt: the static request pipeline below fell apart.  The bucket table is addressed through a
// global-address-space pointer for the same reason.)
typedef __attribute__((address_space(1))) const u32x2_n g_uint2_t;
typedef __attribute__((address_space(1))) uint32_t g_u32_t;
typedef __attribute__((address_space(1))) unsigned long long g_u64_t;
template <bool DBG>
__device__ __attribute__((noinline)) void sfx_role_probe(uint64_t hot_addr, uint32_t lb_hot, uint32_t tiers, uint64_t pool_ctrl_addr, uint64_t dbg_addr, uint32_t pi, uint64_t t_begin)
{
    SfView s{};                                            // (sf_probe_decide looks at s.tiers only)
    s.tiers = tiers;
    g_uint2_t* hot = reinterpret_cast<g_uint2_t*>((uintptr_t)hot_addr);
    const uint32_t lane = lane_id();
    auto aborted = [&]() -> bool { return lds_ld_acq(kXAbort) != 0u; };
    auto give_up = [&]() { if (lane == 0) lds_st(kXAbort, 1u); };
    __builtin_amdgcn_s_setprio(2);
    // Per-F state lives in VGPR LANES: lane L holds the state of this P's F number L & 3 (3: none).  Indexed arrays would end up in scratch
    // memory -- whose loads count under vmcnt and would make every pass wait for all the bucket requests in flight.
    const uint32_t jf = lane & 3u;
    const bool mine = jf < (uint32_t)kXFperP;
    const uint32_t fb_l = kXBase + (pi * kXFperP + (mine ? jf : 0u)) * kXFBytes;      // this lane's F block
    const uint32_t fc_l = fb_l + kXOffCtrl;
    uint32_t vH = 0, vUqi = 0, vInfl = 0, vTq = 0, vQh = 0;  // entries popped; units whose marker went out; rounds in flight; q2 entries pushed; q2 head as last seen
    bool vFin = !mine;
    uint64_t d_wait_q2 = 0, d_passes = 0, d_rounds = 0, d_cands = 0, d_defer = 0;
    // rounds in flight (slot J of the rotation): raw buckets, the word a matching slot equals, offset | valid, and whose they are
    u32x2 r_a[kProbeDepth], r_b[kProbeDepth];
    uint32_t r_e[kProbeDepth], r_pos[kProbeDepth], r_f[kProbeDepth];
#pragma unroll
    for (int J = 0; J < kProbeDepth; J++) { r_a[J] = u32x2{0, 0}; r_b[J] = u32x2{0, 0}; r_e[J] = 0; r_pos[J] = 0; r_f[J] = kNone; }
    bool ok = true, finished = false;
    uint32_t idle_streak = 0;
    // W1 of F number j = {ring head, deferred-ring tail}: both are this wavefront's words, written together
    auto publish_w1 = [&](uint32_t j) {
        const uint32_t fb = kXBase + (pi * kXFperP + j) * kXFBytes;
        lds_st64_rel(fb + kXOffCtrl + 8u, (uint32_t)__builtin_amdgcn_readlane((int)vH, (int)j), (uint32_t)__builtin_amdgcn_readlane((int)vTq, (int)j), lane);
    };
    // push one entry per flagged lane (in lane order) to the deferred ring of F number j; waits for room (the head is re-read only when
    // the last seen one leaves none)
    auto q2_push = [&](uint32_t j, bool flag, uint32_t value) -> bool {
        const uint64_t m = __ballot(flag);
        const uint32_t n = (uint32_t)__popcll(m);
        if (!n) return true;
        const uint32_t fb = kXBase + (pi * kXFperP + j) * kXFBytes;
        const uint32_t tq = (uint32_t)__builtin_amdgcn_readlane((int)vTq, (int)j);
        uint32_t qh = (uint32_t)__builtin_amdgcn_readlane((int)vQh, (int)j);
        if (tq + n - qh > kXQ2) {
            uint32_t spins = 0;
            const uint64_t t0 = DBG ? __builtin_amdgcn_s_memtime() : 0;
            for (;;) {
                qh = lds_ld_acq(fb + kXOffCtrl + 16u);
                if (tq + n - qh <= kXQ2) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit) { give_up(); return false; }
                if ((spins & 63u) == 0 && aborted()) return false;
            }
            if (DBG) d_wait_q2 += __builtin_amdgcn_s_memtime() - t0;
            if (jf == j) vQh = qh;
        }
        if (flag) lds_st(fb + kXOffQ2 + q2_slot(tq + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) * 4u, value);
        if (jf == j) vTq += n;
        publish_w1(j);
        return true;
    };
    auto pass = [&](auto slot_c) __attribute__((always_inline)) {
        constexpr int J = decltype(slot_c)::value;
        if (DBG) d_passes++;
        // (1) look at the round requested kProbeDepth passes ago.  Unconditionally: a slot that holds no round has no valid lane -- were the
        // look skipped, the buckets requested into its registers would be dead values, the registers would be reused at once, and the
        // hardware hazard (write after an outstanding load) would make the compiler wait for every request in flight
        {
            const bool valid[1] = {(r_pos[J] & 0x10000u) != 0};
            bool defer[1]; uint32_t hint[1];
            const u32x2 a1[1] = {r_a[J]}, b1[1] = {r_b[J]};
            const uint32_t e1[1] = {r_e[J]};
            sf_probe_decide<1>(s, a1, b1, e1, valid, defer, hint);
            if (DBG) d_defer += (uint32_t)__popcll(__ballot(defer[0]));
            if (!q2_push(r_f[J] & 3u, defer[0], (hint[0] << 16) | (r_pos[J] & 0xFFFFu))) ok = false;
            if (jf == r_f[J]) vInfl--;
            r_f[J] = kNone;
        }
        // (2) the state of the three rings: ONE 8-byte read per F gives {tail, units closed} as they were together -- a tail seen with
        // "not closed yet" never reaches into the next unit; the end of a closed unit is fetched only when there is one
        const uint64_t w0 = lds_ld64(fc_l);
        lds_order();
        const uint32_t tl = (uint32_t)w0, ncl = (uint32_t)(w0 >> 32) & ~kDoneBit;
        const bool f_done = ((uint32_t)(w0 >> 32) & kDoneBit) != 0u;
        const bool closed = !vFin && (int32_t)(ncl - vUqi) > 0;
        uint32_t uend = 0;
        if (__ballot(closed)) { uend = lds_ld(fc_l + 32u + (vUqi & (kXUq - 1u)) * 4u); lds_order(); }
        const uint32_t avail = vFin ? 0u : (closed ? uend : tl) - vH;
        if (!vFin && f_done && ncl == vUqi && vInfl == 0) vFin = true;
        // a unit that is complete, popped and looked at: its end marker follows its last deferred position
        const bool unit_over = closed && avail == 0 && vInfl == 0;
        {
            uint32_t mm = (uint32_t)(__ballot(unit_over) & 0x7ull);
            while (mm) {
                const uint32_t j = (uint32_t)__builtin_ctz(mm);
                mm &= mm - 1u;
                if (!q2_push(j, lane == 0, kQ2Mark)) ok = false;
            }
            if (unit_over) vUqi++;
        }
        // choose: a full round first (of a complete unit before an open one), else the most entries
        const uint32_t score = (unit_over || vFin) ? 0u : (avail >= 64u ? 64u + (closed ? 1u : 0u) : avail);
        uint32_t best = 0, best_j = kNone;
#pragma unroll
        for (int j = 0; j < kXFperP; j++) {
            const uint32_t sc = (uint32_t)__builtin_amdgcn_readlane((int)score, j);
            if (sc > best) { best = sc; best_j = (uint32_t)j; }
        }
        const bool best_closed = best_j != kNone && ((__ballot(closed) >> best_j) & 1ull) != 0;
        // a round of fewer than 32 entries is only worth a pass when its unit is complete or nothing else has come for a while
        const bool take = best_j != kNone && (best >= 32u || best_closed || idle_streak >= 2u);
        uint32_t w = 0, meta = 0; bool valid = false;
        if (take) {
            const uint32_t Hj = (uint32_t)__builtin_amdgcn_readlane((int)vH, (int)best_j);
            const uint32_t av = (uint32_t)__builtin_amdgcn_readlane((int)avail, (int)best_j);
            const uint32_t m = av < 64u ? av : 64u;
            const uint32_t fb = kXBase + (pi * kXFperP + best_j) * kXFBytes;
            valid = lane < m;
            if (valid) {
                const u32x2_n e = *reinterpret_cast<const lds_u32x2_t*>((uintptr_t)(fb + kXOffRing + ((Hj + lane) & (kXRing - 1u)) * 8u));
                w = e.x; meta = e.y;
            }
            asm volatile("" : "+v"(w), "+v"(meta));
            if (jf == best_j) { vH += m; vInfl++; }
            publish_w1(best_j);                             // the entries are read (their loads were issued before this store): the F may overwrite them
            r_f[J] = best_j;
            idle_streak = 0;
            if (DBG) { d_rounds++; d_cands += m; }
        } else idle_streak++;      // (DBG: idle passes = passes - rounds)
        // (3) request the round's buckets.  EVERY pass issues exactly these two loads (lanes without an entry read bucket 0): the compiler
        // counts the loads in flight along straight code only, one load behind a branch and every wait becomes "all of them"
        {
            const uint32_t ha = t4_hash_a(w), hb = t4_hash_b(w);
            r_e[J] = t4_expect(t4_fingerprint(ha, lb_hot), meta >> 16);
            const u32x2_n ra = hot[valid ? t4_bucket(ha, lb_hot) : 0u];
            const u32x2_n rb = hot[valid ? t4_bucket(hb, lb_hot) : 0u];
            r_a[J] = u32x2{ra.x, ra.y}; r_b[J] = u32x2{rb.x, rb.y};
            r_pos[J] = valid ? ((meta & 0xFFFFu) | 0x10000u) : 0u;
        }
        if (!take) {
            const bool all = (__ballot(vFin) & 0xFull) == 0xFull;
            bool none = true;
#pragma unroll
            for (int K = 0; K < kProbeDepth; K++) none = none && r_f[K] == kNone;
            if (all && none) finished = true;
            else if (none) __builtin_amdgcn_s_sleep(8);        // nothing to pop, nothing in flight: an idle pass costs the SIMD's issue slots, a sleep does not
            else __builtin_amdgcn_s_sleep(1);
        }
    };
    // (NO exit between the passes of one trip: with a branch out of the loop after each pass the compiler's wait-count pass fell back to
    // vmcnt(0) at the loop header; with straight passes it waits for a round's buckets with vmcnt(2 * kProbeDepth - 1), as designed.
    // A pass after `finished` or a failure is harmless: nothing to pop, nothing in flight.)
    uint32_t guard = 0;
    while (ok && !finished) {
        pass(std::integral_constant<int, 0>{});
        if (kProbeDepth > 1) pass(std::integral_constant<int, 1 % kProbeDepth>{});
        if (kProbeDepth > 2) pass(std::integral_constant<int, 2 % kProbeDepth>{});
        if (kProbeDepth > 3) pass(std::integral_constant<int, 3 % kProbeDepth>{});
        if (kProbeDepth > 4) pass(std::integral_constant<int, 4 % kProbeDepth>{});
        if (kProbeDepth > 5) pass(std::integral_constant<int, 5 % kProbeDepth>{});
        if ((++guard & 63u) == 0 && aborted()) ok = false;
        if (idle_streak > kSpinLimit) { give_up(); ok = false; }
    }
    if (lds_ld_acq(kXAbort) != 0u && lane == 0 && pool_ctrl_addr) reinterpret_cast<g_u32_t*>((uintptr_t)pool_ctrl_addr)[2] = 1u;
    if (DBG && dbg_addr && lane == 0) {
        unsigned long long* q = reinterpret_cast<unsigned long long*>((uintptr_t)dbg_addr) + 32;
        atomicAdd(q + 8, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin)); atomicAdd(q + 9, (unsigned long long)d_wait_q2);
        atomicAdd(q + 10, (unsigned long long)d_passes); atomicAdd(q + 11, (unsigned long long)(d_passes - d_rounds)); atomicAdd(q + 12, (unsigned long long)d_rounds);
        atomicAdd(q + 13, (unsigned long long)d_cands); atomicAdd(q + 14, (unsigned long long)d_defer);
    }
}

}  // namespace

template <bool IC, int MODE, bool DBG>
__global__ __launch_bounds__(kXThreads) void k_sfx(SfView s, BatchView b, ScanOut o, uint64_t n_chunks)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < kBloomMasks; i += kXThreads) lds[i] = bloom_mask_entry(i);
    for (uint32_t i = threadIdx.x; i < (1u << 15); i += kXThreads) lds[kBloomMasks + i] = s.bloom[i];
    for (uint32_t i = threadIdx.x; i < kXF * 16u; i += kXThreads) lds[((kXBase + (i >> 4) * kXFBytes + kXOffCtrl) >> 2) + (i & 15u)] = 0;      // rings empty, nothing closed
    if (threadIdx.x == 0) lds[kXAbort >> 2] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t t_begin = DBG ? __builtin_amdgcn_s_memtime() : 0;
    if (wave < (uint32_t)kXF) sfx_role_filter<IC, MODE, DBG>(s, b, o, n_chunks, wave, lane, t_begin);
    else sfx_role_probe<DBG>((uint64_t)(uintptr_t)s.t4_hot, s.tier_log2_cap[3], s.tiers, (uint64_t)(uintptr_t)o.pool_ctrl, (uint64_t)(uintptr_t)o.dbg, wave - (uint32_t)kXF, t_begin);      // (a function: its own registers)
}

// Is k_sfx the kernel for this scan?  (128-KiB filter, needles of >= 4 bytes only, count / emit, enough units for every F wavefront of every CU)
bool sfx_eligible(const SfView& s, const BatchView& b, const ScanOut& o, int mode, int n_cu, bool any_size)
{
    if (mode != kModeCount && mode != kModeEmit) return false;
    if (s.bloom_log2_words != 15 || (s.tiers & 7u) != 0 || !(s.tiers & 8u)) return false;
    if (o.unit_chunks == 0 || o.unit_chunks > 64) return false;
    const uint64_t n_units = (sf_chunks(b) + o.unit_chunks - 1) / o.unit_chunks;
    return any_size ? n_units >= 1 : n_units >= (uint64_t)n_cu * kXF * 2u;
}

template <bool IC, int MODE, bool DBG>
static hipError_t launch_sfx_t(const SfView& s, const BatchView& b, const ScanOut& o, int n_cu, hipStream_t st)
{
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sfx<IC, MODE, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError();
        attr_set = true;
    }
    const uint64_t n_chunks = sf_chunks(b);
    const uint64_t n_units = (n_chunks + o.unit_chunks - 1) / o.unit_chunks;
    uint64_t blocks = (uint64_t)n_cu;
    const uint64_t need = (n_units + kXF - 1) / kXF;
    if (blocks > need) blocks = need;
    if (blocks == 0) return hipSuccess;
    ScanOut oo = o;
    if (n_units <= blocks * kXF) oo.next_unit = nullptr;
    hipLaunchKernelGGL((k_sfx<IC, MODE, DBG>), dim3((uint32_t)blocks), dim3(kXThreads), kXLdsBytes, st, s, b, oo, n_chunks);
    return hipGetLastError();
}

hipError_t launch_sfx(bool ic, int mode, const SfView& s, const BatchView& b, const ScanOut& o, int n_cu, hipStream_t st)
{
    const bool dbg = o.dbg != nullptr;
    if (ic) {
        if (mode == kModeCount) return dbg ? launch_sfx_t<true, kModeCount, true>(s, b, o, n_cu, st) : launch_sfx_t<true, kModeCount, false>(s, b, o, n_cu, st);
        return dbg ? launch_sfx_t<true, kModeEmit, true>(s, b, o, n_cu, st) : launch_sfx_t<true, kModeEmit, false>(s, b, o, n_cu, st);
    }
    if (mode == kModeCount) return dbg ? launch_sfx_t<false, kModeCount, true>(s, b, o, n_cu, st) : launch_sfx_t<false, kModeCount, false>(s, b, o, n_cu, st);
    return dbg ? launch_sfx_t<false, kModeEmit, true>(s, b, o, n_cu, st) : launch_sfx_t<false, kModeEmit, false>(s, b, o, n_cu, st);
}

}  // namespace dev
}  // namespace am
