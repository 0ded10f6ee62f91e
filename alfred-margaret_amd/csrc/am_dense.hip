// am_dense.hip -- automata that contain the empty needle, on top of the suffix-filter kernel.
//
// With the empty needle among the needles the root owns values, and the reference folds them after EVERY successful goto
// (Automaton.hs:373-376,502-503,519): wherever some needle prefix ends, i.e. at almost every position of the text.  The
// suffix filter (k_sf) still finds the sparse part -- needle ends, and the prefix terminals the flattener adds -- and
// these kernels add the dense part: every position where a FIRST code point of some needle ends and k_sf reported
// nothing gets a record with state 0 (machineValues ! 0 = the root's list).  One workgroup per k_sf work unit (<= 64 KiB
// of text): the unit's sparse records are marked in an LDS bitmap, the text is classified position by position
// (ends_first_code_point, am_image.h; ASCII through a 128-bit mask in LDS), and the union is written in position order.
// Output volume is what bounds this path: 16 bytes of records per byte of text.
#include <hip/hip_runtime.h>

#include "am_device.h"

namespace am {
namespace dev {

namespace {

constexpr int kDenseThreads = 256;
constexpr uint32_t kDenseWords = 2048;            // 64 KiB of text = 65536 positions = 2048 bitmap words

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* scratch /* kDenseThreads + 1 */, uint32_t* total)
{
    // Hillis-Steele over the 256 threads in LDS (this path is bound by its output volume, not by this scan)
    const uint32_t t = threadIdx.x;
    scratch[t] = v;
    __syncthreads();
    for (uint32_t d = 1; d < kDenseThreads; d <<= 1) {
        const uint32_t add = t >= d ? scratch[t - d] : 0u;
        __syncthreads();
        scratch[t] += add;
        __syncthreads();
    }
    const uint32_t incl = scratch[t];
    if (total) *total = scratch[kDenseThreads - 1];
    __syncthreads();
    return incl - v;
}

}  // namespace

// WRITE = false: unit_totals[u] = records of unit u (sparse and dense together).  WRITE = true: writes them at out_offsets[u].
template <bool IC, bool WRITE>
__global__ void __launch_bounds__(kDenseThreads) k_dense(AcView a, BatchView b, const Record* __restrict__ sparse, const uint64_t* __restrict__ sparse_offsets,
                                                        uint32_t unit_chunks, uint64_t n_units, uint32_t* __restrict__ unit_totals,
                                                        const uint64_t* __restrict__ out_offsets, Record* __restrict__ out)
{
    __shared__ uint32_t sp_bits[kDenseWords], un_bits[kDenseWords];
    __shared__ uint32_t un_pre[WRITE ? kDenseWords : 1], sp_pre[WRITE ? kDenseWords : 1];      // write pass: records (all / sparse) of the unit before word w
    __shared__ uint32_t first_ascii[4];
    __shared__ uint32_t scratch[kDenseThreads + 1];
    const uint64_t u = blockIdx.x;
    if (u >= n_units) return;
    const uint32_t t = threadIdx.x;
    const uint64_t unit_start = u * unit_chunks * (uint64_t)kSfChunk;
    const uint64_t unit_end = unit_start + (uint64_t)unit_chunks * kSfChunk < b.total ? unit_start + (uint64_t)unit_chunks * kSfChunk : b.total;
    const uint32_t n_words = (uint32_t)((unit_end - unit_start + 31) >> 5);
    for (uint32_t w = t; w < kDenseWords; w += kDenseThreads) { sp_bits[w] = 0; un_bits[w] = 0; }
    if (t < 4) first_ascii[t] = 0;
    __syncthreads();
    if (t < 128 && !(a.root_ascii[t] & kWildcard)) atomicOr(&first_ascii[t >> 5], 1u << (t & 31u));
    const uint64_t s0 = sparse_offsets[u], s1 = sparse_offsets[u + 1];
    for (uint64_t r = s0 + t; r < s1; r += kDenseThreads) {
        const Record rec = sparse[r];
        const uint32_t bit = (uint32_t)(b.offsets[rec.haystack] + rec.end_pos - 1 - unit_start);
        atomicOr(&sp_bits[bit >> 5], 1u << (bit & 31u));
    }
    __syncthreads();
    // classify: word w covers bytes unit_start + 32 w ..  Two steps per word.  ASCII bytes are answered from the 128-bit mask; a byte that ENDS a longer code point (a
    // continuation byte with no continuation byte behind it) is only marked.  Then the marked positions -- one or two per word in cfg3's text -- take the long way
    // (decode, lower-case, the root's goto probe: half a dozen dependent loads).  Until round 6 the long way sat inside the byte loop: with 64 lanes per wavefront some lane
    // took it at nearly every one of the 32 turns and all waited -- 2.6 ms per 256 MiB and pass, most of this path's time (LABNOTES R6.10).
    uint32_t mine = 0, mine_sp = 0;
    for (uint32_t w = t; w < n_words; w += kDenseThreads) {
        const uint64_t g0 = unit_start + 32ull * w;
        uint32_t hay = find_haystack(b, g0);
        uint64_t he = b.offsets[hay + 1];
        const uint32_t n_in = unit_end - g0 < 32u ? (uint32_t)(unit_end - g0) : 32u;
        // the word's 32 bytes + the one behind them (the text is padded by 16 zero bytes and g0 is a multiple of 32)
        // (a borrowed batch is readable up to round_up(total, 16): the second half of the last word may lie beyond that)
        const uint64_t readable = (b.total + 15u) & ~15ull;
        const u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};
        const u32x4 lo = load16(reinterpret_cast<const u32x4*>(b.text + g0)), hi = g0 + 32u <= readable ? load16(reinterpret_cast<const u32x4*>(b.text + g0 + 16)) : zero4;
        const uint32_t behind = g0 + 32u < b.total ? b.text[g0 + 32u] : 0u;
        const uint32_t wd[9] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w, behind};
        uint32_t bits = 0, cand = 0;
#pragma unroll
        for (uint32_t j = 0; j < 32; j++) {
            const uint64_t g = g0 + j;
            const uint32_t b0 = (wd[j >> 2] >> (8u * (j & 3u))) & 0xFFu, b1 = (wd[(j + 1u) >> 2] >> (8u * ((j + 1u) & 3u))) & 0xFFu;
            if (j < n_in) {
                while (g >= he) { hay++; he = b.offsets[hay + 1]; }
                if (b0 < 0x80u) { const uint32_t c = IC ? fold_byte(b0) : b0; bits |= ((first_ascii[c >> 5] >> (c & 31u)) & 1u) << j; }
                else if ((b0 & 0xC0u) == 0x80u && !(g + 1 < he && (b1 & 0xC0u) == 0x80u)) cand |= 1u << j;
            }
        }
        while (cand) {
            const uint32_t j = (uint32_t)__builtin_ctz(cand);
            cand &= cand - 1u;
            const uint64_t g = g0 + j;
            const uint32_t h2 = find_haystack(b, g);
            bits |= (uint32_t)ends_first_code_point(a, IC, b.text, b.offsets[h2], b.offsets[h2 + 1], g) << j;
        }
        const uint32_t un = bits | sp_bits[w];
        un_bits[w] = un;
        mine += __popc(un); mine_sp += __popc(sp_bits[w]);
    }
    __syncthreads();
    if (!WRITE) {
        uint32_t total = 0;
        (void)block_exclusive_scan(mine, scratch, &total);
        if (t == 0) unit_totals[u] = total;
        return;
    }
    // The write pass.  Where every record goes: word w's records start at un_pre[w] (sparse ones among them: sp_pre[w]) -- exclusive sums over the words in position
    // order (thread t sums the contiguous words [t * per, (t + 1) * per), a scan over the threads, the words' own sums on top).  Then the workgroup walks the POSITIONS
    // 256 at a time: neighbouring lanes hold neighbouring positions, so (nearly) neighbouring records -- until round 6 each thread wrote its own 8 words' records one after
    // the other, 64 lanes of a store 3.7 KB apart: 16 bytes per L2 request, 0.56 TB/s of records, the one robustness row below 60 GiB/s.
    const uint32_t per = (n_words + kDenseThreads - 1) / kDenseThreads;
    const uint32_t w0 = t * per < n_words ? t * per : n_words, w1 = (t + 1) * per < n_words ? (t + 1) * per : n_words;
    uint32_t cu = 0, cs = 0;
    for (uint32_t w = w0; w < w1; w++) { cu += __popc(un_bits[w]); cs += __popc(sp_bits[w]); }
    uint32_t run_un = block_exclusive_scan(cu, scratch, nullptr);
    uint32_t run_sp = block_exclusive_scan(cs, scratch, nullptr);
    for (uint32_t w = w0; w < w1; w++) { un_pre[w] = run_un; sp_pre[w] = run_sp; run_un += __popc(un_bits[w]); run_sp += __popc(sp_bits[w]); }
    __syncthreads();
    const uint64_t base = out_offsets[u];
    const uint32_t n_pos = (uint32_t)(unit_end - unit_start);
    for (uint32_t p = t; p < n_pos; p += kDenseThreads) {
        const uint32_t w = p >> 5, j = p & 31u, un = un_bits[w];
        if (!((un >> j) & 1u)) continue;
        const uint32_t below = (1u << j) - 1u;
        const uint64_t at = base + un_pre[w] + __popc(un & below);
        const uint32_t sp = sp_bits[w];
        if ((sp >> j) & 1u) { out[at] = sparse[s0 + sp_pre[w] + __popc(sp & below)]; continue; }
        const uint64_t g = unit_start + p;
        const uint32_t hay = find_haystack(b, g);
        out[at] = Record{g - b.offsets[hay] + 1, hay, 0u};
    }
}

hipError_t launch_dense(bool ic, bool write, const AcView& a, const BatchView& b, const Record* sparse, const uint64_t* sparse_offsets, uint32_t unit_chunks,
                        uint64_t n_units, uint32_t* unit_totals, const uint64_t* out_offsets, Record* out, hipStream_t st)
{
    if (n_units == 0) return hipSuccess;
    if (unit_chunks * (kSfChunk / 32u) > kDenseWords) return hipErrorInvalidValue;
    const dim3 grid((uint32_t)n_units), block(kDenseThreads);
    if (ic) {
        if (write) hipLaunchKernelGGL((k_dense<true, true>), grid, block, 0, st, a, b, sparse, sparse_offsets, unit_chunks, n_units, unit_totals, out_offsets, out);
        else hipLaunchKernelGGL((k_dense<true, false>), grid, block, 0, st, a, b, sparse, sparse_offsets, unit_chunks, n_units, unit_totals, out_offsets, out);
    } else {
        if (write) hipLaunchKernelGGL((k_dense<false, true>), grid, block, 0, st, a, b, sparse, sparse_offsets, unit_chunks, n_units, unit_totals, out_offsets, out);
        else hipLaunchKernelGGL((k_dense<false, false>), grid, block, 0, st, a, b, sparse, sparse_offsets, unit_chunks, n_units, unit_totals, out_offsets, out);
    }
    return hipGetLastError();
}

// records -> per-haystack value counts / total / any-flags (count and containsAny entry points of automata whose records
// come out of the dense pass)
__global__ void __launch_bounds__(256) k_records_reduce(const Record* __restrict__ recs, uint64_t n, const uint32_t* __restrict__ vlen, uint64_t* __restrict__ hay_counts,
                                                        uint64_t* __restrict__ total, uint8_t* __restrict__ flags)
{
    __shared__ uint64_t wave_sum[4];
    uint64_t v = 0;
    // grid-stride: a few thousand workgroups, ONE atomic on the total per workgroup (a record at almost every position means
    // hundreds of millions of records; an atomic per wave on one address would serialise for tens of milliseconds)
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
        const Record rec = recs[r];
        const uint64_t x = vlen[rec.state];
        v += x;
        if (hay_counts && x) atomicAdd(reinterpret_cast<unsigned long long*>(hay_counts + rec.haystack), (unsigned long long)x);
        if (flags) flags[rec.haystack] = 1;
    }
    if (total) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
        if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) { const uint64_t s4 = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3]; if (s4) atomicAdd(reinterpret_cast<unsigned long long*>(total), (unsigned long long)s4); }
    }
}

hipError_t launch_records_reduce(const Record* recs, uint64_t n, const uint32_t* vlen, uint64_t* hay_counts, uint64_t* total, uint8_t* flags, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_records_reduce, dim3((uint32_t)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, recs, n, vlen, hay_counts, total, flags);
    return hipGetLastError();
}

// ---- one haystack scanned in ranges (am_run_range): the records of ONE haystack are sorted by end position
// out[0] = records with end_pos <= x0, out[1] = records with end_pos <= x1 (two binary searches)
__global__ void k_range_bounds(const Record* __restrict__ recs, uint64_t n, uint64_t x0, uint64_t x1, uint64_t* __restrict__ out)
{
    if (threadIdx.x > 1 || blockIdx.x) return;
    const uint64_t x = threadIdx.x ? x1 : x0;
    uint64_t lo = 0, hi = n;
    while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if (recs[mid].end_pos <= x) lo = mid + 1; else hi = mid; }
    out[threadIdx.x] = lo;
}
__global__ void k_range_rebase(Record* __restrict__ recs, uint64_t n, uint64_t add)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) recs[i].end_pos += add;
}
// a segment of a host batch scanned on its own (am_run on 1 GiB of slices and more): its haystack numbers count from the segment's first haystack
__global__ void k_hay_rebase(Record* __restrict__ recs, uint64_t n, uint32_t add)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) recs[i].haystack += add;
}
hipError_t launch_hay_rebase(Record* recs, uint64_t n, uint32_t add, hipStream_t st)
{
    if (n == 0 || add == 0) return hipSuccess;
    hipLaunchKernelGGL(k_hay_rebase, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, recs, n, add);
    return hipGetLastError();
}
// am_debug_resident_waves: every wavefront spins for `cycles` of the shader clock
__global__ void __launch_bounds__(1024) k_spin(uint64_t cycles, uint32_t* __restrict__ out)
{
    __shared__ uint32_t s[1024];
    s[threadIdx.x] = threadIdx.x;
    const uint64_t t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) { }
    if (s[(threadIdx.x + 1u) % blockDim.x] == 0xFFFFFFFFu) out[0] = 1u;
}
hipError_t launch_spin(uint32_t workgroups, uint32_t threads, uint64_t cycles, uint32_t* out, hipStream_t st)
{
    hipLaunchKernelGGL(k_spin, dim3(workgroups), dim3(threads), 0, st, cycles, out);
    return hipGetLastError();
}
hipError_t launch_range_bounds(const Record* recs, uint64_t n, uint64_t x0, uint64_t x1, uint64_t* out2, hipStream_t st)
{
    hipLaunchKernelGGL(k_range_bounds, dim3(1), dim3(64), 0, st, recs, n, x0, x1, out2);
    return hipGetLastError();
}
hipError_t launch_range_rebase(Record* recs, uint64_t n, uint64_t add, hipStream_t st)
{
    if (n == 0 || add == 0) return hipSuccess;
    hipLaunchKernelGGL(k_range_rebase, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, recs, n, add);
    return hipGetLastError();
}

}  // namespace dev
}  // namespace am
