// Round 6: is the rate of dependent random 4-byte loads (tools/microbench/l2_gather.hip: 320 G/s on L2 hits, 70 G/s on misses, 2 048 lanes per CU) a bound of the memory
// system or of the latency x the lanes in flight?  Lanes per CU x independent chains per lane, for a table that hits L2 and one that misses it.
// hipcc --offload-arch=gfx950 -O3 l2_curve.hip -o l2_curve.bin && ./l2_curve.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int ILP>
__global__ __launch_bounds__(1024) void k_chase(const uint32_t* __restrict__ tab, uint32_t mask, uint32_t steps, uint32_t* out)
{
    uint32_t x[ILP];
#pragma unroll
    for (int k = 0; k < ILP; k++) x[k] = ((blockIdx.x * 1024u + threadIdx.x) * ILP + k) * 0x9E3779B1u;
    for (uint32_t i = 0; i < steps; i++) {
#pragma unroll
        for (int k = 0; k < ILP; k++) x[k] = tab[(x[k] ^ (x[k] >> 15)) & mask] + i * 0x85EBCA6Bu;
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < ILP; k++) s ^= x[k];
    if (s == 0x12345678u) out[0] = s;
}
// a third of the lanes, chosen anew every turn, load; the others compute: what does a load instruction with few lanes cost?  (counts[0] += loads made)
__global__ __launch_bounds__(1024) void k_chase_partial(const uint32_t* __restrict__ tab, uint32_t mask, uint32_t steps, uint32_t third, unsigned long long* counts)
{
    uint32_t x = (blockIdx.x * 1024u + threadIdx.x) * 0x9E3779B1u, n = 0;
    for (uint32_t i = 0; i < steps; i++) {
        if (((x >> 9) & 1023u) < third) { x = tab[(x ^ (x >> 15)) & mask] + i * 0x85EBCA6Bu; n++; }
        else x = x * 1664525u + 1013904223u + i;
    }
    atomicAdd(counts, (unsigned long long)n);
    if (x == 0x12345678u) counts[1] = x;
}
static double run_partial(const uint32_t* d, uint32_t mask, unsigned long long* counts, int blocks, int threads, uint32_t steps, uint32_t third)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k_chase_partial<<<blocks, threads>>>(d, mask, 64, third, counts);
    (void)hipMemset(counts, 0, 16);
    (void)hipEventRecord(a);
    k_chase_partial<<<blocks, threads>>>(d, mask, steps, third, counts);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long n = 0; (void)hipMemcpy(&n, counts, 8, hipMemcpyDeviceToHost);
    return (double)n / ms / 1e6;
}
template <int ILP>
static double run(const uint32_t* d, uint32_t mask, uint32_t* out, int blocks, int threads, uint32_t steps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_chase<ILP><<<blocks, threads>>>(d, mask, 64, out);
    hipEventRecord(a);
    k_chase<ILP><<<blocks, threads>>>(d, mask, steps, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return (double)blocks * threads * ILP * steps / ms / 1e6;
}
int main()
{
    const int n_cu = 256;
    uint32_t* out; hipMalloc(&out, 64);
    printf("G dependent random 4-byte loads per second; columns: 1 / 2 / 4 independent chains per lane; then the latency one chain sees (lanes x chains / rate), ns\n");
    for (size_t mb : {2, 64}) {
        const size_t n = (mb << 20) / 4;
        std::vector<uint32_t> h(n);
        uint64_t s = 88172645463325252ull;
        for (size_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)s; }
        uint32_t* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        const uint32_t steps = 1024, m = (uint32_t)n - 1;
        for (int lanes : {64, 256, 512, 1024, 2048}) {
            const int threads = lanes >= 1024 ? 1024 : lanes, blocks = n_cu * (lanes / threads);
            const double r1 = run<1>(d, m, out, blocks, threads, steps), r2 = run<2>(d, m, out, blocks, threads, steps), r4 = run<4>(d, m, out, blocks, threads, steps);
            const double tot = (double)n_cu * lanes;
            printf("table %3zu MiB, %4d lanes per CU: %7.1f %7.1f %7.1f   latency %6.0f %6.0f %6.0f\n", mb, lanes, r1, r2, r4, tot / r1, tot * 2 / r2, tot * 4 / r4);
        }
        unsigned long long* counts; (void)hipMalloc(&counts, 16);
        for (int lanes : {1024, 2048}) {
            const int threads = 1024, blocks = n_cu * (lanes / threads);
            printf("table %3zu MiB, %4d lanes per CU, loads by 100 / 66 / 33 / 10 %% of the lanes per turn: %7.1f %7.1f %7.1f %7.1f G loads/s\n", mb, lanes,
                   run_partial(d, m, counts, blocks, threads, steps, 1024), run_partial(d, m, counts, blocks, threads, steps, 676), run_partial(d, m, counts, blocks, threads, steps, 338),
                   run_partial(d, m, counts, blocks, threads, steps, 102));
        }
        (void)hipFree(counts);
        hipFree(d);
    }
    return 0;
}
