// Do two workgroups of 1024 threads share a CU?  512 workgroups (two per CU) spin for ~2 ms each; with both resident all of them start at once and the launch takes one spin,
// otherwise two.  By dynamic LDS per workgroup and by scratch use.  hipcc --offload-arch=gfx950 -O3 coresident.hip -o coresident.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <bool SCRATCH>
__global__ __launch_bounds__(1024, 8) void k_spin(unsigned long long cycles, unsigned* out, unsigned idx)
{
    extern __shared__ unsigned s[];
    volatile unsigned priv[SCRATCH ? 64 : 1];
    if (SCRATCH) for (unsigned i = 0; i < 64; i++) priv[(i * 7 + idx) & 63] = i;
    s[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) { }
    if (s[(threadIdx.x + 1) & 1023] == 0xFFFFFFFFu || (SCRATCH && priv[idx & 63] == 0xFFFFFFFFu)) out[0] = 1;
}
template <bool SCRATCH>
static void run(const char* what, size_t lds, unsigned* out)
{
    hipFuncSetAttribute((const void*)k_spin<SCRATCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {256, 512}) {
        k_spin<SCRATCH><<<blocks, 1024, lds>>>(100000, out, 3); hipDeviceSynchronize();
        hipEventRecord(a); k_spin<SCRATCH><<<blocks, 1024, lds>>>(200000000ull / 100, out, 3); hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        printf("%-28s dynamic LDS %6zu B, %3d workgroups: %.3f ms\n", what, lds, blocks, ms);
    }
}
__global__ __launch_bounds__(1024) void k_spin_any(unsigned long long cycles, unsigned* out)
{
    extern __shared__ unsigned s[];
    s[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) { }
    if (s[(threadIdx.x + 1) % blockDim.x] == 0xFFFFFFFFu) out[0] = 1;
}
int main()
{
    unsigned* out; hipMalloc(&out, 64);
    {   // 32 wavefronts per CU asked for in workgroups of 256 / 512 / 1024 threads (and 16 per CU for comparison)
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int bs : {256, 512, 1024}) for (int waves_per_cu : {16, 32}) {
            const int blocks = 256 * waves_per_cu * 64 / bs;
            k_spin_any<<<blocks, bs, 4096>>>(100000, out); hipDeviceSynchronize();
            hipEventRecord(a); k_spin_any<<<blocks, bs, 4096>>>(2000000ull, out); hipEventRecord(b); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            printf("workgroups of %4d threads, %2d wavefronts per CU asked for: %.3f ms\n", bs, waves_per_cu, ms);
        }
    }
    for (size_t lds : {4096ul, 66304ul, 81680ul}) { run<false>("no scratch", lds, out); run<true>("scratch (256 B per lane)", lds, out); }
    return 0;
}
