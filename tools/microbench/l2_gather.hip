// Round 6: what does the memory system give 2 048 lanes per CU that each chase dependent random 4-byte loads (k_dfa's access pattern) -- requests per second by table size
// (all L2 hits ... mostly L2 misses that the 256-MiB Infinity Cache serves) and by the load's cache policy.  hipcc --offload-arch=gfx950 -O3 l2_gather.hip -o l2_gather.bin && ./l2_gather.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int POL>
__device__ __forceinline__ uint32_t ld(const uint32_t* p)
{
    if (POL == 1) return __builtin_nontemporal_load(p);
    if (POL == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (POL == 3) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (POL == 4) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return *p;
}
template <int POL>
__global__ __launch_bounds__(1024, 8) void k_chase(const uint32_t* __restrict__ tab, uint32_t mask, uint32_t steps, uint32_t* out)
{
    uint32_t x = (blockIdx.x * 1024u + threadIdx.x) * 0x9E3779B1u;
    for (uint32_t i = 0; i < steps; i++) x = ld<POL>(tab + ((x ^ (x >> 15)) & mask)) + i * 0x85EBCA6Bu;
    if (x == 0x12345678u) out[0] = x;
}
template <int POL>
static double run(const uint32_t* d, uint32_t mask, uint32_t* out, int n_cu, uint32_t steps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_chase<POL><<<n_cu * 2, 1024>>>(d, mask, 64, out);
    hipEventRecord(a);
    k_chase<POL><<<n_cu * 2, 1024>>>(d, mask, steps, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return (double)n_cu * 2 * 1024 * steps / ms / 1e6;
}
int main()
{
    const int n_cu = 256;
    uint32_t* out; hipMalloc(&out, 64);
    printf("G dependent random 4-byte loads per second, 524288 lanes in flight; policy: plain / nontemporal / agent-scope atomic (sc1) / system-scope atomic (sc0 sc1) / workgroup-scope atomic (sc0)\n");
    for (size_t mb : {1, 4, 8, 16, 32, 64, 256}) {
        const size_t n = (mb << 20) / 4;
        std::vector<uint32_t> h(n);
        uint64_t s = 88172645463325252ull;
        for (size_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)s; }
        uint32_t* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        const uint32_t steps = 1024, m = (uint32_t)n - 1;
        printf("table %4zu MiB: %7.1f %7.1f %7.1f %7.1f %7.1f\n", mb, run<0>(d, m, out, n_cu, steps), run<1>(d, m, out, n_cu, steps), run<2>(d, m, out, n_cu, steps), run<3>(d, m, out, n_cu, steps), run<4>(d, m, out, n_cu, steps));
        hipFree(d);
    }
    return 0;
}
