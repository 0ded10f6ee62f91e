// VALU issue-rate microbenchmark for gfx950: which integer ops are full rate?  (tools/microbench, not part of the product)
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed, int iters)
{
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
    uint32_t c = seed | 1;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1\n" : "+v"(a0), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 1) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %1\n v_mul_lo_u32 %3, %3, %1\n v_mul_lo_u32 %4, %4, %1\n" : "+v"(a0), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 2) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %2, %2, %1\n v_mul_u32_u24 %3, %3, %1\n v_mul_u32_u24 %4, %4, %1\n" : "+v"(a0), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 3) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %2, %2, %1, %3\n v_mad_u32_u24 %3, %3, %1, %4\n v_mad_u32_u24 %4, %4, %1, %0\n" : "+v"(a0), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 4) { REP16(asm volatile("v_lshrrev_b32 %0, %1, %0\n v_lshrrev_b32 %2, %1, %2\n v_lshrrev_b32 %3, %1, %3\n v_lshrrev_b32 %4, %1, %4\n" : "+v"(a0), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 5) { REP16(asm volatile("v_bfe_u32 %0, %0, %1, 1\n v_bfe_u32 %2, %2, %1, 1\n v_bfe_u32 %3, %3, %1, 1\n v_bfe_u32 %4, %4, %1, 1\n" : "+v"(a0), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 6) { REP16(asm volatile("v_alignbyte_b32 %0, %0, %1, 1\n v_alignbyte_b32 %2, %2, %1, 2\n v_alignbyte_b32 %3, %3, %1, 3\n v_alignbyte_b32 %4, %4, %1, 1\n" : "+v"(a0), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 7) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %2, %2, %1\n v_mul_hi_u32 %3, %3, %1\n v_mul_hi_u32 %4, %4, %1\n" : "+v"(a0), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 8) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 1, %1\n v_lshl_or_b32 %2, %2, 1, %1\n v_lshl_or_b32 %3, %3, 1, %1\n v_lshl_or_b32 %4, %4, 1, %1\n" : "+v"(a0), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 9) { REP16(asm volatile("v_dot4_u32_u8 %0, %0, %1, %2\n v_dot4_u32_u8 %2, %2, %1, %3\n v_dot4_u32_u8 %3, %3, %1, %4\n v_dot4_u32_u8 %4, %4, %1, %0\n" : "+v"(a0), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 10) { REP16(asm volatile("v_pk_mul_lo_u16 %0, %0, %1\n v_pk_mul_lo_u16 %2, %2, %1\n v_pk_mul_lo_u16 %3, %3, %1\n v_pk_mul_lo_u16 %4, %4, %1\n" : "+v"(a0), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 11) { REP16(asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_cndmask_b32 %0, %2, %3, vcc\n v_cmp_eq_u32 vcc, %2, %1\n v_cndmask_b32 %2, %3, %4, vcc\n" : "+v"(a0), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP> void run(const char* name, uint32_t* d)
{
    const int blocks = 256 * 8, iters = 2000;       // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double insts = (double)blocks * 4 /*waves*/ * iters * 64.0;      // wave-level instructions
    // per SIMD: 1024 SIMDs; cycles at 2.4 GHz
    printf("%-18s %8.3f ms  %.2f wave-instr/ns  -> %.2f cycles per wave-instr per SIMD (at 2.4 GHz, 1024 SIMDs)\n", name, ms, insts / (ms * 1e6),
           (ms * 1e-3 * 2.4e9) / (insts / 1024.0));
}

int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", d); run<1>("v_mul_lo_u32", d); run<2>("v_mul_u32_u24", d); run<3>("v_mad_u32_u24", d); run<4>("v_lshrrev_b32", d);
    run<5>("v_bfe_u32", d); run<6>("v_alignbyte_b32", d); run<7>("v_mul_hi_u32", d); run<8>("v_lshl_or_b32", d); run<9>("v_dot4_u32_u8", d);
    run<10>("v_pk_mul_lo_u16", d); run<11>("v_cmp+v_cndmask", d);
    hipFree(d);
    return 0;
}
