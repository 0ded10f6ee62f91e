// am_sfx.hip -- k_sfx: the suffix-filter scan with ROLE-SPECIALISED wavefronts (gfx950, wave64).
//
// k_sf (am_kernels.hip) runs filter, compaction, probe and resolve in every wavefront.  Only the filter needs the 128-KiB LDS table
// that pins the occupancy at one 1024-thread workgroup per CU = four in-order wavefronts per SIMD, and those four cannot hide the
// probe's L2 round trips and the resolve's dependent HBM trips.  k_sfx keeps the SAME LDS filter, tables, record
// format and unit bookkeeping but gives each of the workgroup's 16 wavefronts one job:
//
//   F  12 wavefronts   stream the haystack (one coalesced 16-B load per lane and KiB), fold, filter against the LDS Bloom filter, compact
//                      the surviving positions and push (4-byte window, the two bytes before it, offset in the unit) entries into a ring
//                      in LDS.  No table access, no memory wait but the prefetched stream.
//   P   3 wavefronts   each serves four F rings: pop <= 64 entries, hash, request both hot cuckoo buckets, and look at them kProbeDepth
//                      passes later (the requests of kProbeDepth rounds stay in flight: registers rotate by name, the compiler counts
//                      vmcnt statically); survivors go to the P's ring of deferred positions.
//   R   1 wavefront    pops deferred positions of all three P rings, <= 128 at a time (two per lane in lock step), resolves them exactly
//                      (sf_resolve_head / _walk of am_image.h: haystack bytes -> slot line -> trie) and owns ALL record output: per F it
//                      keeps the current unit's chain of pool blocks; a unit-end marker that travels behind the unit's last deferred
//                      position closes the unit (unit_counts / unit_first / unit_slots exactly as k_sf writes them).
//
// Every SIMD then holds three wavefronts that never wait for memory next to one that always does.  Hand-over is single-producer /
// single-consumer everywhere (F -> its P, P -> R, R -> F for unit-slot recycling), so nobody waits in a cycle: R waits for nobody.
// All waits are bounded (watchdog): a hand-over that does not move sets pool_ctrl[2] and every wavefront leaves.
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
constexpr int kXF = 12, kXP = 3, kXFperP = 4;         // wavefronts 0..11 filter, 12..14 probe, 15 resolves
static_assert(kXF == kXP * kXFperP && kXF + kXP + 1 == kXThreads / 64, "role split");
constexpr int kProbeDepth = 4;                         // probe rounds whose bucket requests are in flight per P wavefront
constexpr int kRN = 2;                                 // deferred positions per lane in one resolve batch

constexpr uint32_t kXMaskBytes = kBloomMasks * 4u;
constexpr uint32_t kXBloomBytes = 4u << 15;            // the 128-KiB filter only
constexpr uint32_t kXStage = 1056;                     // per F: copy of the current chunk (folded): 8 bytes before it at offset 8, the chunk at 16, padding
constexpr uint32_t kXQ1 = 128;                         // per F: candidate offsets of one sub-pass (u16)
constexpr uint32_t kXRing = 128;                       // per F: entries {window, nb << 16 | offset in the unit} (8 B)
constexpr uint32_t kXUq = 4;                           // per F: unit slots {unit id, end_T}
constexpr uint32_t kXFCtrl = 16 + kXUq * 8;            // tail, head (by P), uq_r (units retired, by R), uq_w (units announced), slots
constexpr uint32_t kXFBytes = kXStage + kXQ1 * 2 + kXRing * 8 + kXFCtrl;
constexpr uint32_t kXQ2 = 128;                         // per P: deferred positions / markers (u32)
constexpr uint32_t kXPBytes = kXQ2 * 4 + 16;           // ring + {tail, head (by R)}
constexpr uint32_t kXRState = 16;                      // per F, owned by R: unit_slots, unit_count, cur_block, first_block
constexpr uint32_t kXBase = kXMaskBytes + kXBloomBytes;
constexpr uint32_t kXPBase = kXBase + kXF * kXFBytes;
constexpr uint32_t kXRBase = kXPBase + kXP * kXPBytes;
constexpr uint32_t kXAbort = kXRBase + kXF * kXRState; // one word: somebody's wait timed out
constexpr uint32_t kXLdsBytes = kXAbort + 16;
static_assert(kXLdsBytes <= 160 * 1024, "k_sfx LDS budget");

constexpr uint32_t kUnitOpen = 0xFFFFFFFFu;            // end_T of a unit that is still being filtered
constexpr uint32_t kUnitEmpty = 0xFFFFFFFEu;           // unit id of a free slot
constexpr uint32_t kUnitDone = 0xFFFFFFFDu;            // unit id: this F has no more units
constexpr uint32_t kQ2Mark = 1u << 31, kQ2PDone = 1u << 30;      // q2 entry: unit-end marker of F (bits 20-23) / this P is done; else F << 20 | hint << 16 | offset
constexpr uint32_t kSpinLimit = 1u << 21;

// Ordering between the wavefronts of the workgroup concerns LDS only: a wavefront's DS operations execute in issue order, so "everything
// before is done" = s_waitcnt lgkmcnt(0).  (A workgroup-scope fence would also wait for vmcnt(0) -- every global load in flight -- and
// the P wavefronts live on keeping four rounds of bucket requests in flight across these hand-overs.)
__device__ __forceinline__ void lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (compiler ordering only: no instruction at wavefront scope)
    __builtin_amdgcn_s_waitcnt(0xC07F);                         // lgkmcnt(0), vmcnt and expcnt untouched
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t lds_ld_acq(uint32_t byte_addr)
{
    const uint32_t v = __hip_atomic_load(reinterpret_cast<lds_u32_t*>((uintptr_t)byte_addr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    lds_fence();
    return r;
}
__device__ __forceinline__ uint32_t lds_ld(uint32_t byte_addr)
{
    return __hip_atomic_load(reinterpret_cast<lds_u32_t*>((uintptr_t)byte_addr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// lane 0 publishes a control word after everything this wavefront wrote to LDS before
__device__ __forceinline__ void lds_st_rel(uint32_t byte_addr, uint32_t v, uint32_t lane)
{
    lds_fence();
    if (lane == 0) __hip_atomic_store(reinterpret_cast<lds_u32_t*>((uintptr_t)byte_addr), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st(uint32_t byte_addr, uint32_t v) { __hip_atomic_store(reinterpret_cast<lds_u32_t*>((uintptr_t)byte_addr), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

}  // namespace

template <bool IC, int MODE, bool DBG>
__global__ __launch_bounds__(kXThreads) void k_sfx(SfView s, BatchView b, ScanOut o, uint64_t n_chunks)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < kBloomMasks; i += kXThreads) lds[i] = bloom_mask_entry(i);
    for (uint32_t i = threadIdx.x; i < (1u << 15); i += kXThreads) lds[kBloomMasks + i] = s.bloom[i];
    // control blocks: rings empty, every unit slot free, R's per-F unit state empty
    for (uint32_t i = threadIdx.x; i < kXF; i += kXThreads) {
        const uint32_t c = kXBase + i * kXFBytes + kXStage + kXQ1 * 2 + kXRing * 8;
        lds[(c >> 2) + 0] = 0; lds[(c >> 2) + 1] = 0; lds[(c >> 2) + 2] = 0; lds[(c >> 2) + 3] = 0;
        for (uint32_t k = 0; k < kXUq; k++) { lds[(c >> 2) + 4 + 2 * k] = kUnitEmpty; lds[(c >> 2) + 5 + 2 * k] = kUnitOpen; }
        const uint32_t r = kXRBase + i * kXRState;
        lds[(r >> 2) + 0] = 0; lds[(r >> 2) + 1] = 0; lds[(r >> 2) + 2] = kNone; lds[(r >> 2) + 3] = kNone;
    }
    if (threadIdx.x < kXP) { const uint32_t c = kXPBase + threadIdx.x * kXPBytes + kXQ2 * 4; lds[(c >> 2)] = 0; lds[(c >> 2) + 1] = 0; }
    if (threadIdx.x == 0) lds[kXAbort >> 2] = 0;
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t UC = o.unit_chunks;
    const uint64_t n_units = (n_chunks + UC - 1) / UC;
    const uint64_t unit_bytes = (uint64_t)UC * kSfChunk;
    const uint64_t t_begin = DBG ? __builtin_amdgcn_s_memtime() : 0;
    auto f_ctrl = [](uint32_t f) -> uint32_t { return kXBase + f * kXFBytes + kXStage + kXQ1 * 2 + kXRing * 8; };      // tail +0, head +4, uq_r +8, uq_w +12, slots +16
    auto aborted = [&]() -> bool { return lds_ld_acq(kXAbort) != 0u; };
    // a wait that does not end: tell the other wavefronts (LDS only -- a global store inside the spin loops would make the compiler drain every
    // load in flight before each of them); whoever leaves with the flag set reports it to the host (report_abort)
    auto give_up = [&]() { if (lane == 0) lds_st(kXAbort, 1u); };
    auto report_abort = [&]() { if (lds_ld_acq(kXAbort) != 0u && lane == 0 && o.pool_ctrl) o.pool_ctrl[2] = 1u; };

    if (wave < (uint32_t)kXF) {
        // =========================================================================== F: filter
        const uint32_t f = wave;
        const uint32_t stage = kXBase + f * kXFBytes, q1 = stage + kXStage, ring = q1 + kXQ1 * 2, ctrl = ring + kXRing * 8;
        const uint64_t n_f = (uint64_t)gridDim.x * kXF;
        uint32_t T = 0, uq_w = 0;                              // entries pushed so far; units announced so far
        uint64_t d_wait_ring = 0, d_wait_uq = 0, d_chunks = 0, d_cands = 0;
        __builtin_amdgcn_s_setprio(0);
        auto fetch = [&](uint64_t cc, uint4& v) {
            const uint64_t p = cc * kSfChunk + lane * 16u;
            v = make_uint4(0, 0, 0, 0);
            if (cc < n_chunks && p < b.total) {
                typedef uint32_t u32x4_native __attribute__((ext_vector_type(4)));
                const u32x4_native t = *reinterpret_cast<const u32x4_native*>(b.text + p);
                v = make_uint4(t.x, t.y, t.z, t.w);
            }
        };
        auto fetch_before = [&](uint64_t cc, uint32_t& c3, uint32_t& c4) {          // the 8 (folded) bytes before chunk cc; uniform
            uint2 t = make_uint2(0, 0);
            if (cc < n_chunks && cc > 0) {
                t = *reinterpret_cast<const uint2*>(b.text + cc * kSfChunk - 8);
                t.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.x); t.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.y);
            }
            c3 = IC ? fold_dword(t.x) : t.x; c4 = IC ? fold_dword(t.y) : t.y;
        };
        // waits until unit slot `uq_w` is free (R has retired the unit that used it), writes {unit, end_t} there and publishes the new count of
        // announced units (kUnitDone: the last word of this F)
        auto announce = [&](uint32_t unit, uint32_t end_t) -> bool {
            uint32_t spins = 0;
            const uint64_t t0 = DBG ? __builtin_amdgcn_s_memtime() : 0;
            while (uq_w - lds_ld_acq(ctrl + 8u) >= kXUq) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > kSpinLimit) { give_up(); return false; }
                if ((spins & 63u) == 0 && aborted()) return false;
            }
            if (DBG) d_wait_uq += __builtin_amdgcn_s_memtime() - t0;
            const uint32_t slot = ctrl + 16u + (uq_w & (kXUq - 1u)) * 8u;
            if (lane == 0) { lds_st(slot, unit); lds_st(slot + 4u, end_t); }
            uq_w++;
            lds_st_rel(ctrl + 12u, uq_w, lane);
            return true;
        };
        uint64_t u = (uint64_t)blockIdx.x * kXF + f;
        uint4 cur_v; uint32_t carry3, carry4;
        fetch(u * UC, cur_v);
        fetch_before(u * UC, carry3, carry4);
        asm volatile("" : "+v"(cur_v.x), "+v"(cur_v.y), "+v"(cur_v.z), "+v"(cur_v.w));
        uint64_t u_next = u;
        bool ok = true;
        for (; ok && u < n_units; u = u_next) {
            u_next = u + n_f;
            if (o.next_unit) {
                uint32_t ticket = 0;
                if (lane == 0) ticket = atomicAdd(o.next_unit, 1u);
                u_next = n_f + (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
            }
            if (!announce((uint32_t)u, kUnitOpen)) { ok = false; break; }
            const uint32_t my_slot = ctrl + 16u + ((uq_w - 1u) & (kXUq - 1u)) * 8u;
            const uint64_t unit_base_chunk = u * UC;
            const uint32_t n_in_unit = (uint32_t)(unit_base_chunk + UC <= n_chunks ? UC : n_chunks - unit_base_chunk);
            for (uint32_t ci = 0; ok && ci < n_in_unit; ci++) {
                const uint64_t c = unit_base_chunk + ci;
                uint4 next_v = make_uint4(0, 0, 0, 0);
                uint32_t next_c3 = 0, next_c4 = 0;
                const bool last_of_unit = ci + 1 >= n_in_unit;
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
                uint32_t cand = 0;
                {
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
                }
                if (p0 + 16 > b.total) cand &= p0 < b.total ? (1u << (uint32_t)(b.total - p0)) - 1u : 0u;
                // ---- compaction + hand-over, up to 128 candidates per sub-pass
                for (;;) {
                    const uint32_t n = __popc(cand);
                    const uint32_t incl = wave_inclusive_sum(n, lane);
                    const uint32_t total = __shfl(incl, 63, 64);
                    if (total == 0) break;
                    uint32_t idx = incl - n;
                    while (cand && idx < kXQ1) {
                        const uint32_t k = __builtin_ctz(cand);
                        cand &= cand - 1u;
                        lds_write_u16(q1 + 2u * idx++, lane * 16u + k);
                    }
                    const uint32_t n_q1 = total < kXQ1 ? total : kXQ1;
                    if (DBG) d_cands += n_q1;
                    wave_lds_fence();
                    uint32_t ew[2], em[2];
                    bool ev[2];
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const uint32_t e = 64u * k + lane;
                        ev[k] = e < n_q1;
                        const uint32_t pos = ev[k] ? lds_read_u16(q1 + 2u * e) : 0u;
                        // bytes pos-5 .. pos of the staged chunk (stage offset 11 + pos): window = the last four (newest on top), nb = the two before it
                        const uint32_t a = 11u + pos, sh = a & 3u;
                        const uint32_t sp = stage + (a & ~3u);
                        const uint32_t x0 = lds_read_u32(sp), x1 = lds_read_u32(sp + 4u), x2 = lds_read_u32(sp + 8u);
                        const uint32_t two = __builtin_amdgcn_alignbyte(x1, x0, sh) & 0xFFFFu;
                        const uint32_t nb = (two >> 8) | ((two & 0xFFu) << 8);
                        ew[k] = sh < 2u ? __builtin_amdgcn_alignbyte(x1, x0, sh + 2u) : __builtin_amdgcn_alignbyte(x2, x1, sh - 2u);
                        em[k] = (nb << 16) | ((ci << 10) + pos);
                    }
                    // push: entry g may be written once g - head < kXRing
                    const uint32_t T0 = T, Tend = T0 + n_q1;
                    uint32_t done = T0, spins = 0;
                    for (;;) {
                        const uint32_t head = lds_ld_acq(ctrl + 4u);
                        const uint32_t room_end = head + kXRing;
                        const uint32_t upto = (int32_t)(Tend - room_end) <= 0 ? Tend : room_end;
                        if (upto != done) {
#pragma unroll
                            for (int k = 0; k < 2; k++) {
                                const uint32_t g = T0 + 64u * k + lane;
                                if (ev[k] && (int32_t)(g - done) >= 0 && (int32_t)(g - upto) < 0) lds_write_u32x2(ring + (g & (kXRing - 1u)) * 8u, make_uint2(ew[k], em[k]));
                            }
                            done = upto;
                            lds_st_rel(ctrl, done, lane);
                        }
                        if (done == Tend) break;
                        const uint64_t t0 = DBG ? __builtin_amdgcn_s_memtime() : 0;
                        __builtin_amdgcn_s_sleep(1);
                        if (DBG) d_wait_ring += __builtin_amdgcn_s_memtime() - t0;
                        if (++spins > kSpinLimit) { give_up(); ok = false; break; }
                        if ((spins & 63u) == 0 && aborted()) { ok = false; break; }
                    }
                    T = Tend;
                    if (!ok || total <= kXQ1) break;
                    wave_lds_fence();
                }
                cur_v = next_v; carry3 = next_c3; carry4 = next_c4;
                if (DBG) d_chunks++;
            }
            // close the unit: everything up to T belongs to it
            lds_st_rel(my_slot + 4u, T, lane);
        }
        if (ok) (void)announce(kUnitDone, T);
        report_abort();
        if (DBG && o.dbg && lane == 0) {
            unsigned long long* q = reinterpret_cast<unsigned long long*>(o.dbg + 32);
            atomicAdd(q + 0, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin)); atomicAdd(q + 1, (unsigned long long)d_wait_ring);
            atomicAdd(q + 2, (unsigned long long)d_wait_uq); atomicAdd(q + 3, (unsigned long long)d_chunks); atomicAdd(q + 4, (unsigned long long)d_cands);
        }
        return;
    }

    if (wave < (uint32_t)(kXF + kXP)) {
        // =========================================================================== P: probe
        const uint32_t pi = wave - kXF;
        const uint32_t q2 = kXPBase + pi * kXPBytes, q2c = q2 + kXQ2 * 4;       // ring, {tail, head}
        __builtin_amdgcn_s_setprio(2);
        // Per-F state lives in VGPR LANES: lane L holds the state of this P's F number L & 3 (replicated 16 times).  Indexed arrays would end up
        // in scratch memory -- whose loads count under vmcnt and would make every pass wait for all the bucket requests in flight.
        const uint32_t jf = lane & 3u;
        const uint32_t fc_l = f_ctrl(pi * kXFperP + jf);         // this lane's F: control block, ring
        uint32_t vH = 0, vUqi = 0, vInfl = 0;                    // entries popped; units finished with; rounds in flight
        bool vFin = false;
        uint32_t Tq = 0;                                         // q2 entries pushed
        uint64_t d_wait_q2 = 0, d_passes = 0, d_rounds = 0, d_cands = 0, d_defer = 0;
        // rounds in flight (slot J of the rotation): raw buckets, the word a matching slot equals, offset | valid, and whose they are
        u32x2 r_a[kProbeDepth], r_b[kProbeDepth];
        uint32_t r_e[kProbeDepth], r_pos[kProbeDepth], r_f[kProbeDepth];
#pragma unroll
        for (int J = 0; J < kProbeDepth; J++) { r_a[J] = u32x2{0, 0}; r_b[J] = u32x2{0, 0}; r_e[J] = 0; r_pos[J] = 0; r_f[J] = kNone; }
        const uint32_t lb_hot = s.tier_log2_cap[3];
        bool ok = true, finished = false;
        uint32_t idle_streak = 0;
        // push one entry per flagged lane (in lane order) to q2; waits for room
        auto q2_push = [&](bool flag, uint32_t value) -> bool {
            const uint64_t m = __ballot(flag);
            const uint32_t n = (uint32_t)__popcll(m);
            if (!n) return true;
            uint32_t spins = 0;
            const uint64_t t0 = DBG ? __builtin_amdgcn_s_memtime() : 0;
            while (Tq + n - lds_ld_acq(q2c + 4u) > kXQ2) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit) { give_up(); return false; }
                if ((spins & 63u) == 0 && aborted()) return false;
            }
            if (DBG) d_wait_q2 += __builtin_amdgcn_s_memtime() - t0;
            if (flag) lds_st(q2 + ((Tq + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) & (kXQ2 - 1u)) * 4u, value);
            Tq += n;
            lds_st_rel(q2c, Tq, lane);
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
                const uint32_t fg = pi * kXFperP + (r_f[J] & 3u);
                if (DBG) d_defer += (uint32_t)__popcll(__ballot(defer[0]));
                if (!q2_push(defer[0], (fg << 20) | (hint[0] << 16) | (r_pos[J] & 0xFFFFu))) ok = false;
                if (jf == r_f[J]) vInfl--;
                r_f[J] = kNone;
            }
            // (2) the state of the four rings: tail and announced-unit count FIRST, then the current unit's slot (a tail read before its slot never
            // reaches into the next unit; a slot is only looked at once its unit has been announced -- before that it still holds an older unit)
            const uint32_t tl = lds_ld(fc_l), aw = lds_ld(fc_l + 12u);
            const uint32_t sa = fc_l + 16u + (vUqi & (kXUq - 1u)) * 8u;
            lds_fence();
            const uint32_t uid = lds_ld(sa), uend = lds_ld(sa + 4u);
            lds_fence();
            const bool have_unit = !vFin && (int32_t)(aw - vUqi) > 0;
            const bool done_word = have_unit && uid == kUnitDone;
            const bool closed = have_unit && !done_word && uend != kUnitOpen;
            const uint32_t avail = (have_unit && !done_word) ? (closed ? uend : tl) - vH : 0u;
            if (done_word && vInfl == 0) vFin = true;
            // a unit that is complete, popped and looked at: its end marker follows its last deferred position
            const bool unit_over = closed && avail == 0 && vInfl == 0;
            {
                uint32_t mm = (uint32_t)(__ballot(unit_over) & 0xFull);
                while (mm) {
                    const uint32_t j = (uint32_t)__builtin_ctz(mm);
                    mm &= mm - 1u;
                    if (!q2_push(lane == 0, kQ2Mark | ((pi * kXFperP + j) << 20))) ok = false;
                }
                if (unit_over) vUqi++;
            }
            // choose: a full round first (of a complete unit before an open one), else the most entries
            const uint32_t score = unit_over ? 0u : (avail >= 64u ? 64u + (closed ? 1u : 0u) : avail);
            uint32_t best = 0, best_j = kNone;
#pragma unroll
            for (int j = 0; j < kXFperP; j++) {
                const uint32_t sc = (uint32_t)__builtin_amdgcn_readlane((int)score, j);
                if (sc > best) { best = sc; best_j = (uint32_t)j; }
            }
            const bool best_closed = best_j != kNone && ((__ballot(closed) >> best_j) & 1ull) != 0;
            // a round of fewer than 32 entries is only worth a pass when its unit is complete or nothing else has come for a while
            const bool take = best_j != kNone && (best >= 32u || best_closed || idle_streak >= 4u);
            uint32_t w = 0, meta = 0; bool valid = false;
            if (take) {
                const uint32_t Hj = (uint32_t)__builtin_amdgcn_readlane((int)vH, (int)best_j);
                const uint32_t av = (uint32_t)__builtin_amdgcn_readlane((int)avail, (int)best_j);
                const uint32_t m = av < 64u ? av : 64u;
                const uint32_t fc = f_ctrl(pi * kXFperP + best_j);
                const uint32_t rg = fc - kXRing * 8u;
                valid = lane < m;
                if (valid) {
                    const u32x2_n e = *reinterpret_cast<const lds_u32x2_t*>((uintptr_t)(rg + ((Hj + lane) & (kXRing - 1u)) * 8u));
                    w = e.x; meta = e.y;
                }
                asm volatile("" : "+v"(w), "+v"(meta));
                lds_st_rel(fc + 4u, Hj + m, lane);                 // the entries are in registers: the F may overwrite them
                if (jf == best_j) { vH += m; vInfl++; }
                r_f[J] = best_j;
                idle_streak = 0;
                if (DBG) { d_rounds++; d_cands += m; }
            } else idle_streak++;      // (DBG: idle passes = passes - rounds)
            // (3) request the round's buckets.  EVERY pass issues exactly these two loads (lanes without an entry read bucket 0): the compiler
            // counts the loads in flight along straight code only, one load behind a branch and every wait becomes "all of them"
            {
                const uint32_t ha = t4_hash_a(w), hb = t4_hash_b(w);
                r_e[J] = t4_expect(t4_fingerprint(ha, lb_hot), meta >> 16);
                const uint2 ra = *reinterpret_cast<const uint2*>(s.t4_hot + (valid ? t4_bucket(ha, lb_hot) : 0u));
                const uint2 rb = *reinterpret_cast<const uint2*>(s.t4_hot + (valid ? t4_bucket(hb, lb_hot) : 0u));
                r_a[J] = u32x2{ra.x, ra.y}; r_b[J] = u32x2{rb.x, rb.y};
                r_pos[J] = valid ? ((meta & 0xFFFFu) | 0x10000u) : 0u;
            }
            if (!take) {
                const bool all = (__ballot(vFin) & 0xFull) == 0xFull;
                bool none = true;
#pragma unroll
                for (int K = 0; K < kProbeDepth; K++) none = none && r_f[K] == kNone;
                if (all && none) finished = true;
                else if (idle_streak < 8u) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(4);
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
        if (ok) (void)q2_push(lane == 0, kQ2PDone);
        report_abort();
        if (DBG && o.dbg && lane == 0) {
            unsigned long long* q = reinterpret_cast<unsigned long long*>(o.dbg + 32);
            atomicAdd(q + 8, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin)); atomicAdd(q + 9, (unsigned long long)d_wait_q2);
            atomicAdd(q + 10, (unsigned long long)d_passes); atomicAdd(q + 11, (unsigned long long)(d_passes - d_rounds)); atomicAdd(q + 12, (unsigned long long)d_rounds);
            atomicAdd(q + 13, (unsigned long long)d_cands); atomicAdd(q + 14, (unsigned long long)d_defer);
        }
        return;
    }

    // =============================================================================== R: resolve + all output
    {
        __builtin_amdgcn_s_setprio(3);
        uint32_t Hq[kXP]; bool pdone[kXP];
#pragma unroll
        for (int p = 0; p < kXP; p++) { Hq[p] = 0; pdone[p] = false; }
        uint64_t nval = 0;
        uint32_t grant_next = 0, grant_left = 0;
        bool pool_ok = true;
        uint64_t d_wait = 0, d_batches = 0, d_items = 0, d_found = 0, d_resolve = 0;
        uint32_t idle = 0;
        // a pool block for this wavefront's next records (blocks are drawn kSfBlockGrant at a time: one atomic on the device-wide counter costs ~10 ns)
        auto new_block = [&]() -> uint32_t {
            if (grant_left == 0) {
                uint32_t g = 0;
                if (lane == 0) g = atomicAdd(o.pool_ctrl, kSfBlockGrant);
                grant_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
                grant_left = kSfBlockGrant;
            }
            const uint32_t id = grant_next++;
            grant_left--;
            if (id >= o.n_blocks) { pool_ok = false; if (lane == 0) o.pool_ctrl[1] = 1u; return kNone; }       // keep counting, the host retries with a larger pool
            return id;
        };
        for (;;) {
            // ---- gather up to 64 * kRN entries from the three rings (tails first, then the entries)
            uint32_t cnt[kXP], total = 0;
#pragma unroll
            for (int p = 0; p < kXP; p++) {
                const uint32_t t = lds_ld_acq(kXPBase + (uint32_t)p * kXPBytes + kXQ2 * 4);
                uint32_t n = t - Hq[p];
                const uint32_t cap = 64u * kRN - total;
                if (n > cap) n = cap;
                cnt[p] = n; total += n;
            }
            if (total == 0) {
                bool all = true;
#pragma unroll
                for (int p = 0; p < kXP; p++) all = all && pdone[p];
                if (all) break;
                const uint64_t t0 = DBG ? __builtin_amdgcn_s_memtime() : 0;
                __builtin_amdgcn_s_sleep(2);
                if (DBG) d_wait += __builtin_amdgcn_s_memtime() - t0;
                if (++idle > kSpinLimit) { give_up(); break; }
                if ((idle & 63u) == 0 && aborted()) break;
                continue;
            }
            // a small batch waits a little for company (a resolve costs the same ~9k cycles for 10 items as for 128) unless a marker may be waiting in it
            if (total < 48u && idle < 24u) { idle++; __builtin_amdgcn_s_sleep(8); continue; }
            idle = 0;
            uint32_t ent[kRN]; bool have[kRN];
            for (int round = 0; round < 2; round++) {
                // item i (< total) sits in lane i % 64, slot i / 64; segment p covers [base_p, base_p + cnt_p)
                uint32_t first_mark[kXP];
#pragma unroll
                for (int p = 0; p < kXP; p++) first_mark[p] = kNone;
#pragma unroll
                for (int k = 0; k < kRN; k++) {
                    const uint32_t i = 64u * k + lane;
                    have[k] = i < total;
                    // segment p and the ring index i + (Hq[p] - base_p), by sums of steps (a select chain over the three rings' values would be
                    // turned into an indexed array in scratch memory)
                    const uint32_t ge1 = i >= cnt[0] ? 1u : 0u, ge2 = i >= cnt[0] + cnt[1] ? 1u : 0u;
                    const uint32_t p = ge1 + ge2;
                    const uint32_t off0 = Hq[0], off1 = Hq[1] - cnt[0], off2 = Hq[2] - cnt[0] - cnt[1];
                    const uint32_t ri = i + off0 + ge1 * (off1 - off0) + ge2 * (off2 - off1);
                    ent[k] = have[k] ? lds_ld(kXPBase + p * kXPBytes + (ri & (kXQ2 - 1u)) * 4u) : 0u;
                    const bool mk = have[k] && (ent[k] & (kQ2Mark | kQ2PDone)) != 0u;
#pragma unroll
                    for (int q = 0; q < kXP; q++) {
                        const uint64_t mm = __ballot(mk && p == (uint32_t)q);
                        if (mm && first_mark[q] == kNone) first_mark[q] = 64u * k + (uint32_t)__builtin_ctzll(mm);
                    }
                }
                // a segment ends with its first marker (the entries behind it belong to the F's next unit / come after a P's last word)
                bool cut = false;
                uint32_t base = 0;
#pragma unroll
                for (int p = 0; p < kXP; p++) {
                    const uint32_t old = cnt[p];
                    if (first_mark[p] != kNone && first_mark[p] - base + 1u < cnt[p]) { cnt[p] = first_mark[p] - base + 1u; cut = true; }
                    base += old;
                }
                if (!cut) break;
                total = cnt[0] + cnt[1] + cnt[2];
            }
            lds_fence();
            // ---- the items: F, hint, position; their unit through the F's current unit slot (the slot R retires next)
            bool valid[kRN]; uint32_t fg[kRN], hint[kRN]; uint64_t gpos[kRN];
            bool any_item = false;
#pragma unroll
            for (int k = 0; k < kRN; k++) {
                valid[k] = have[k] && (ent[k] & (kQ2Mark | kQ2PDone)) == 0u;
                fg[k] = (ent[k] >> 20) & 15u; hint[k] = (ent[k] >> 16) & 3u;
                gpos[k] = 0;
                if (valid[k]) {
                    const uint32_t fc = f_ctrl(fg[k]);
                    const uint32_t r = lds_ld(fc + 8u);
                    const uint32_t unit = lds_ld(fc + 16u + (r & (kXUq - 1u)) * 8u);
                    gpos[k] = (uint64_t)unit * unit_bytes + (ent[k] & 0xFFFFu);
                }
                any_item = any_item || valid[k];
            }
            if (__ballot(any_item)) {
                const uint64_t t0 = DBG ? __builtin_amdgcn_s_memtime() : 0;
                uint32_t hlo[kRN], hhi[kRN], hay[kRN];
                uint64_t end_pos[kRN];
#pragma unroll
                for (int k = 0; k < kRN; k++) {
                    hlo[k] = hhi[k] = 0; hay[k] = 0; end_pos[k] = 0;
                    if (valid[k]) { hlo[k] = b.hidx[gpos[k] >> kHidxShift]; hhi[k] = b.hidx[(gpos[k] >> kHidxShift) + 1]; }
                }
                auto locate = [&]() {
#pragma unroll
                    for (int k = 0; k < kRN; k++) {
                        hay[k] = hlo[k];
                        uint64_t start = valid[k] ? b.offsets[hlo[k]] : 0;
                        if (valid[k] && hlo[k] != hhi[k]) { hay[k] = find_haystack(b, gpos[k]); start = b.offsets[hay[k]]; }
                        end_pos[k] = valid[k] ? gpos[k] - start + 1 : 0;
                    }
                };
                uint32_t w[kRN], w2[kRN], avail[kRN], best_state[kRN], best_vlen[kRN], depth[kRN], node[kRN], t16[kRN][4];
                bool go[kRN], have_rec[kRN], found[kRN];
                SfNode rec[kRN];
                sf_resolve_head<IC, kRN>(s, b.text, gpos, end_pos, valid, hint, locate, w, w2, avail, best_state, best_vlen, depth, go, node, rec, have_rec, t16);
                sf_resolve_walk<IC, kRN>(s, b.text, gpos, avail, w2, go, node, rec, have_rec, depth, best_state, best_vlen, nullptr, 0xFFFFFFFFu, t16);
#pragma unroll
                for (int k = 0; k < kRN; k++) found[k] = valid[k] && best_state[k] != 0;
                if (DBG) {
                    d_resolve += __builtin_amdgcn_s_memtime() - t0; d_batches++;
#pragma unroll
                    for (int k = 0; k < kRN; k++) { d_items += (uint32_t)__popcll(__ballot(valid[k])); d_found += (uint32_t)__popcll(__ballot(found[k])); }
                }
                if (MODE == kModeCount) {
#pragma unroll
                    for (int k = 0; k < kRN; k++) if (found[k]) {
                        nval += best_vlen[k];
                        if (o.hay_counts) atomicAdd(reinterpret_cast<unsigned long long*>(o.hay_counts + hay[k]), (unsigned long long)best_vlen[k]);
                    }
                } else {
                    // ---- records: per F (its current unit's chain), slot 0 items of all lanes before slot 1 items = position order
                    bool rem[kRN];
#pragma unroll
                    for (int k = 0; k < kRN; k++) rem[k] = found[k];
                    for (;;) {
                        uint64_t rm[kRN]; uint64_t any = 0;
#pragma unroll
                        for (int k = 0; k < kRN; k++) { rm[k] = __ballot(rem[k]); any |= rm[k]; }
                        if (!any) break;
                        uint32_t f0 = 0;
                        {
                            bool got = false;
#pragma unroll
                            for (int k = 0; k < kRN; k++) if (!got && rm[k]) { f0 = (uint32_t)__builtin_amdgcn_readlane((int)fg[k], (int)__builtin_ctzll(rm[k])); got = true; }
                        }
                        bool sel[kRN]; uint64_t take[kRN]; uint32_t before[kRN], Fn = 0;
#pragma unroll
                        for (int k = 0; k < kRN; k++) { sel[k] = rem[k] && fg[k] == f0; take[k] = __ballot(sel[k]); before[k] = Fn; Fn += (uint32_t)__popcll(take[k]); rem[k] = rem[k] && !sel[k]; }
                        const uint32_t st = kXRBase + f0 * kXRState;
                        const u32x4_n sv = *reinterpret_cast<const lds_u32x4_t*>((uintptr_t)st);
                        uint32_t unit_slots = (uint32_t)__builtin_amdgcn_readfirstlane((int)sv.x), unit_count = (uint32_t)__builtin_amdgcn_readfirstlane((int)sv.y);
                        uint32_t cur_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)sv.z), first_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)sv.w);
                        const uint32_t s0 = unit_slots, have_blocks = (s0 + kPoolBlock - 1u) / kPoolBlock;
                        const uint32_t n_new = (s0 + Fn + kPoolBlock - 1u) / kPoolBlock - have_blocks;           // <= kRN + 1
                        const uint32_t prev_block = cur_block;
                        uint32_t ids[kRN + 1];
#pragma unroll
                        for (int t = 0; t < kRN + 1; t++) {
                            ids[t] = kNone;
                            if ((uint32_t)t < n_new) {
                                const uint32_t id = new_block();
                                if (id != kNone && pool_ok) {
                                    ids[t] = id;
                                    if (lane == 0) { o.block_next[id] = kNone; if (cur_block != kNone) o.block_next[cur_block] = id; }
                                    if (first_block == kNone) first_block = id;
                                    cur_block = id;
                                }
                            }
                        }
                        // (the new blocks' ids sit in the lanes 0 .. kRN of one register and are picked with a lane shuffle: a select chain over an
                        // array gets turned into an indexed array in scratch memory)
                        uint32_t idv = ids[0];
#pragma unroll
                        for (int t = 1; t < kRN + 1; t++) { const uint32_t us = (uint32_t)__builtin_amdgcn_readfirstlane((int)ids[t]); asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(idv) : "s"(us), "n"(t)); }
#pragma unroll
                        for (int k = 0; k < kRN; k++) {
                            const bool wr = sel[k] && pool_ok;
                            const uint32_t si = s0 + before[k] + (uint32_t)__popcll(take[k] & ((1ull << lane) - 1ull));
                            const uint32_t q = si / kPoolBlock;
                            const uint32_t from_new = (uint32_t)__shfl((int)idv, (int)((q - have_blocks) & 63u), 64);      // (all lanes take part in the shuffle)
                            const uint32_t blk = q < have_blocks ? prev_block : from_new;
                            if (wr) {
                                reinterpret_cast<uint4*>(o.pool)[(uint64_t)blk * kPoolBlock + (si & (kPoolBlock - 1u))] =
                                    make_uint4((uint32_t)end_pos[k], (uint32_t)(end_pos[k] >> 32), hay[k], best_state[k] - 1u);
                            }
                        }
                        unit_slots += Fn; unit_count += Fn;
                        if (lane == 0) lds_write_u32x4(st, make_uint4(unit_slots, unit_count, cur_block, first_block));
                        wave_lds_fence();
                    }
                }
            }
            // ---- markers: a segment's last entry may be one
            {
                uint32_t base = 0;
#pragma unroll
                for (int p = 0; p < kXP; p++) {
                    if (cnt[p]) {
                        const uint32_t i = base + cnt[p] - 1u;
                        uint32_t e = 0;
#pragma unroll
                        for (int k = 0; k < kRN; k++) if ((i >> 6) == (uint32_t)k) e = (uint32_t)__builtin_amdgcn_readlane((int)ent[k], (int)(i & 63u));
                        if (e & kQ2PDone) pdone[p] = true;
                        else if (e & kQ2Mark) {
                            const uint32_t f0 = (e >> 20) & 15u;
                            const uint32_t fc = f_ctrl(f0);
                            const uint32_t r = lds_ld_acq(fc + 8u);
                            const uint32_t slot = fc + 16u + (r & (kXUq - 1u)) * 8u;
                            const uint32_t unit = lds_ld_acq(slot);
                            if (MODE == kModeEmit) {
                                const uint32_t st = kXRBase + f0 * kXRState;
                                const u32x4_n sv = *reinterpret_cast<const lds_u32x4_t*>((uintptr_t)st);
                                if (lane == 0) {
                                    o.unit_slots[unit] = sv.x; o.unit_counts[unit] = sv.y; o.unit_first[unit] = sv.w;
                                    lds_write_u32x4(st, make_uint4(0u, 0u, kNone, kNone));
                                }
                            }
                            lds_st_rel(fc + 8u, r + 1u, lane);         // the F may announce into this slot again
                        }
                    }
                    base += cnt[p];
                }
            }
            // ---- the entries are done with: the Ps may overwrite them
#pragma unroll
            for (int p = 0; p < kXP; p++) if (cnt[p]) { Hq[p] += cnt[p]; lds_st_rel(kXPBase + (uint32_t)p * kXPBytes + kXQ2 * 4 + 4u, Hq[p], lane); }
        }
        report_abort();
        if (MODE == kModeCount) {
            nval = wave_sum_u64(nval);
            if (lane == 0 && nval) atomicAdd(reinterpret_cast<unsigned long long*>(o.total_values), (unsigned long long)nval);
        }
        if (DBG && o.dbg && lane == 0) {
            unsigned long long* q = reinterpret_cast<unsigned long long*>(o.dbg + 32);
            atomicAdd(q + 16, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin)); atomicAdd(q + 17, (unsigned long long)d_wait);
            atomicAdd(q + 18, (unsigned long long)d_batches); atomicAdd(q + 19, (unsigned long long)d_items); atomicAdd(q + 20, (unsigned long long)d_found);
            atomicAdd(q + 21, (unsigned long long)d_resolve);
        }
    }
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
