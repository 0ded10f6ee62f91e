// am_scan.hip -- exclusive prefix sums of the launch paths (unit counts -> record offsets, the Replacer's bookkeeping arrays), the library's
// own: three small launches for any n (tile sums, ONE workgroup over the tile sums, tiles again with their base), no temporary but one u64
// per 4096-element tile.  (Until round 4 this was hipcub::DeviceScan: a header library written for another vendor's execution model behind a
// compatibility layer, in the timed step.)  Memory-bound and tiny next to the scans they follow: n = 163 841 for the 10-GiB benchmark step.
#include <hip/hip_runtime.h>

#include "am_device.h"

namespace am {
namespace dev {

namespace {
constexpr int kScanThreads = 256, kScanPer = 16;
constexpr uint64_t kScanTile = (uint64_t)kScanThreads * kScanPer;      // 4096 elements per workgroup

__device__ __forceinline__ uint64_t wave_incl_u64(uint64_t x, uint32_t lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint64_t y = __shfl_up(x, d, 64); if (lane >= (uint32_t)d) x += y; }
    return x;
}

// exclusive sum over the workgroup's 256 values; returns this thread's base, *total = the workgroup's sum
__device__ __forceinline__ uint64_t block_exclusive(uint64_t v, uint64_t* total)
{
    __shared__ uint64_t wsum[kScanThreads / 64];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint64_t incl = wave_incl_u64(v, lane);
    __syncthreads();                                         // (wsum may still be read by the previous call's last phase)
    if (lane == 63u) wsum[w] = incl;
    __syncthreads();
    uint64_t base = 0, all = 0;
#pragma unroll
    for (int k = 0; k < kScanThreads / 64; k++) { if ((uint32_t)k < w) base += wsum[k]; all += wsum[k]; }
    *total = all;
    return base + incl - v;
}

template <class T>
__global__ __launch_bounds__(kScanThreads) void k_scan_tile_sums(const T* __restrict__ in, uint64_t n, uint64_t* __restrict__ tile_sums)
{
    const uint64_t i0 = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanPer;
    uint64_t v = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; k++) if (i0 + k < n) v += (uint64_t)in[i0 + k];
    uint64_t total;
    (void)block_exclusive(v, &total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// one workgroup: tile_sums[0 .. nt) -> their exclusive sums, in place
__global__ __launch_bounds__(kScanThreads) void k_scan_tiles(uint64_t* __restrict__ tile_sums, uint64_t nt)
{
    uint64_t carry = 0;
    for (uint64_t i0 = 0; i0 < nt; i0 += kScanThreads) {
        const uint64_t i = i0 + threadIdx.x;
        const uint64_t v = i < nt ? tile_sums[i] : 0;
        uint64_t total;
        const uint64_t ex = block_exclusive(v, &total);
        if (i < nt) tile_sums[i] = carry + ex;
        carry += total;
    }
}

template <class T>
__global__ __launch_bounds__(kScanThreads) void k_scan_apply(const T* __restrict__ in, uint64_t n, const uint64_t* __restrict__ tile_base, uint64_t* __restrict__ out)
{
    const uint64_t i0 = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanPer;
    uint64_t x[kScanPer], v = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; k++) { x[k] = i0 + k < n ? (uint64_t)in[i0 + k] : 0; v += x[k]; }
    uint64_t total;
    uint64_t run = tile_base[blockIdx.x] + block_exclusive(v, &total);
#pragma unroll
    for (int k = 0; k < kScanPer; k++) { if (i0 + k < n) out[i0 + k] = run; run += x[k]; }
}

template <class T>
hipError_t scan_any(void* temp, size_t temp_bytes, const T* in, uint64_t* out, uint64_t n, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const uint64_t nt = (n + kScanTile - 1) / kScanTile;
    if (temp_bytes < nt * sizeof(uint64_t) || nt > 0x7FFFFFFFull) return hipErrorInvalidValue;
    uint64_t* tiles = (uint64_t*)temp;
    hipLaunchKernelGGL((k_scan_tile_sums<T>), dim3((uint32_t)nt), dim3(kScanThreads), 0, st, in, n, tiles);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(kScanThreads), 0, st, tiles, nt);
    hipLaunchKernelGGL((k_scan_apply<T>), dim3((uint32_t)nt), dim3(kScanThreads), 0, st, in, n, (const uint64_t*)tiles, out);
    return hipGetLastError();
}
}  // namespace

// bytes of temporary storage for a scan of n elements (either element type)
hipError_t scan_temp_bytes(uint64_t n, size_t* bytes) { *bytes = ((n + kScanTile - 1) / kScanTile + 1) * sizeof(uint64_t); return hipSuccess; }
hipError_t scan64_temp_bytes(uint64_t n, size_t* bytes) { return scan_temp_bytes(n, bytes); }

// exclusive prefix sum of n u32 counts into n u64 offsets (n includes the trailing zero pad, so offsets[n-1] is the total)
hipError_t launch_scan(void* temp, size_t temp_bytes, const uint32_t* counts, uint64_t* offsets, uint64_t n, hipStream_t st) { return scan_any<uint32_t>(temp, temp_bytes, counts, offsets, n, st); }
hipError_t launch_scan64(void* temp, size_t temp_bytes, const uint64_t* in, uint64_t* out, uint64_t n, hipStream_t st) { return scan_any<uint64_t>(temp, temp_bytes, in, out, n, st); }

}  // namespace dev
}  // namespace am
