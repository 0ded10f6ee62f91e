// Round 6: what a plain device-to-device copy of 1 GiB takes on this box (16 bytes per lane, neighbouring lanes neighbouring bytes) -- the yardstick for k_pt_materialise
// (cfg5: 1 GiB read + 1 GiB written in 0.98 ms = 2.2 TB/s moved).  hipcc --offload-arch=gfx950 -O3 copy_rate.hip -o copy_rate.bin && ./copy_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u32x4 v = NT ? __builtin_nontemporal_load(src + i) : src[i];
        if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
}
// one workgroup per 64-KiB haystack, as k_pt_materialise launches
__global__ __launch_bounds__(256) void k_copy_per_hay(const u32x4* __restrict__ src, u32x4* __restrict__ dst)
{
    const size_t base = (size_t)blockIdx.x * 4096;
    for (unsigned i = threadIdx.x; i < 4096; i += 256) dst[base + i] = src[base + i];
}
// the same copy with the source, the destination or both shifted by `sh` bytes off their 16-byte grid (what a piece of a rewritten text is: replacements move the bytes behind them by
// any number of bytes)
typedef unsigned int u32x4_b __attribute__((ext_vector_type(4), aligned(1)));
__global__ __launch_bounds__(256) void k_copy_shift(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, size_t n, unsigned s_sh, unsigned d_sh)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 1 < n; i += (size_t)gridDim.x * 256)
        *reinterpret_cast<u32x4_b*>(dst + 16 * i + d_sh) = *reinterpret_cast<const u32x4_b*>(src + 16 * i + s_sh);
}
int main()
{
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    u32x4 *a, *b; (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMemset(a, 1, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto time = [&](const char* what, auto launch) {
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); for (int k = 0; k < 10; k++) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-44s %.3f ms per GiB = %.2f TB/s moved\n", what, ms, 2.0 * bytes / ms / 1e9);
    };
    for (int g : {1024, 2048, 4096, 16384}) {
        char nm[64]; snprintf(nm, sizeof nm, "grid-stride copy, %d workgroups", g);
        time(nm, [&] { k_copy<false><<<g, 256>>>(a, b, n); });
    }
    time("grid-stride copy, nontemporal, 4096 workgroups", [&] { k_copy<true><<<4096, 256>>>(a, b, n); });
    time("one workgroup per 64 KiB (16384)", [&] { k_copy_per_hay<<<16384, 256>>>(a, b); });
    for (unsigned sh : {1u, 4u, 8u}) {
        char nm[96];
        snprintf(nm, sizeof nm, "source shifted by %u bytes (4096 workgroups)", sh); time(nm, [&] { k_copy_shift<<<4096, 256>>>((const unsigned char*)a, (unsigned char*)b, n, sh, 0); });
        snprintf(nm, sizeof nm, "destination shifted by %u bytes", sh); time(nm, [&] { k_copy_shift<<<4096, 256>>>((const unsigned char*)a, (unsigned char*)b, n, 0, sh); });
        snprintf(nm, sizeof nm, "both shifted by %u bytes", sh); time(nm, [&] { k_copy_shift<<<4096, 256>>>((const unsigned char*)a, (unsigned char*)b, n, sh, sh); });
    }
    time("hipMemcpyDtoD", [&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
