// tools/microbench/valu_rates4.hip -- issue cost of the instructions k_sf's hot loop is made of, measured in SHADER
// CYCLES (s_memtime inside the kernel, so the DVFS clock does not matter), at the occupancy k_sf runs at: one
// 1024-thread workgroup per CU = 4 waves per SIMD.  Each test issues 64 instructions per loop iteration on four
// independent registers (dependency distance 4).  Reported: cycles per wave-instruction per SIMD
// = elapsed cycles of a wave / (instructions per wave * 4 waves sharing the SIMD).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/vr4 tools/microbench/valu_rates4.hip && /tmp/vr4
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
#define T4(fmt) asm volatile(fmt : "+v"(a0), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc", "s4", "s6", "s7", "s8", "s9", "s10", "s11");

template <int OP> __global__ void __launch_bounds__(1024) k(uint32_t* out, uint64_t* cyc, uint32_t seed, int iters)
{
    __shared__ uint32_t sh[8192];
    for (int i = threadIdx.x; i < 8192; i += 1024) sh[i] = i * seed;
    __syncthreads();
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, c = (seed | 1) & 31;
    uint32_t r0 = (a0 * 2654435761u) & 0x7FFC, r1 = (a1 * 2654435761u) & 0x7FFC, r2 = (a2 * 2654435761u) & 0x7FFC, r3 = (a3 * 2654435761u) & 0x7FFC;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP16(T4("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1\n")) }
        if (OP == 1) { REP16(T4("v_and_b32 %0, 0x1fffc, %0\n v_and_b32 %2, 0x1fffc, %2\n v_and_b32 %3, 0x1fffc, %3\n v_and_b32 %4, 0x1fffc, %4\n")) }
        if (OP == 2) { REP16(T4("v_lshrrev_b32 %0, 15, %0\n v_lshrrev_b32 %2, 15, %2\n v_lshrrev_b32 %3, 15, %3\n v_lshrrev_b32 %4, 15, %4\n")) }
        if (OP == 3) { REP16(T4("v_lshrrev_b32 %0, %1, %0\n v_lshrrev_b32 %2, %1, %2\n v_lshrrev_b32 %3, %1, %3\n v_lshrrev_b32 %4, %1, %4\n")) }
        if (OP == 4) { REP16(T4("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %1\n v_mul_lo_u32 %3, %3, %1\n v_mul_lo_u32 %4, %4, %1\n")) }
        if (OP == 5) { REP16(T4("v_mul_lo_u32 %0, %0, s4\n v_mul_lo_u32 %2, %2, s4\n v_mul_lo_u32 %3, %3, s4\n v_mul_lo_u32 %4, %4, s4\n")) }
        if (OP == 6) { REP16(T4("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %2, %2, %1\n v_mul_hi_u32 %3, %3, %1\n v_mul_hi_u32 %4, %4, %1\n")) }
        if (OP == 7) { REP16(T4("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %2, %2, %1\n v_mul_u32_u24 %3, %3, %1\n v_mul_u32_u24 %4, %4, %1\n")) }
        if (OP == 8) { REP16(T4("v_mad_u32_u24 %0, %0, %1, %0\n v_mad_u32_u24 %2, %2, %1, %2\n v_mad_u32_u24 %3, %3, %1, %3\n v_mad_u32_u24 %4, %4, %1, %4\n")) }
        if (OP == 9) { REP16(T4("v_alignbyte_b32 %0, %0, %1, 1\n v_alignbyte_b32 %2, %2, %1, 1\n v_alignbyte_b32 %3, %3, %1, 1\n v_alignbyte_b32 %4, %4, %1, 1\n")) }
        if (OP == 10) { REP16(T4("v_perm_b32 %0, %0, %1, %1\n v_perm_b32 %2, %2, %1, %1\n v_perm_b32 %3, %3, %1, %1\n v_perm_b32 %4, %4, %1, %1\n")) }
        if (OP == 11) { REP16(T4("v_cmp_eq_u32 vcc, %0, %1\n v_cmp_eq_u32 vcc, %2, %1\n v_cmp_eq_u32 vcc, %3, %1\n v_cmp_eq_u32 vcc, %4, %1\n")) }
        if (OP == 12) { REP16(T4("v_cmp_eq_u32_e64 s[8:9], %0, %1\n v_cmp_eq_u32_e64 s[8:9], %2, %1\n v_cmp_eq_u32_e64 s[8:9], %3, %1\n v_cmp_eq_u32_e64 s[8:9], %4, %1\n")) }
        if (OP == 13) { REP16(T4("v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n v_addc_co_u32 %4, vcc, %4, %4, vcc\n")) }
        if (OP == 14) { REP16(T4("v_cmp_eq_u32 vcc, %0, %1\n s_nop 1\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_cmp_eq_u32 vcc, %3, %1\n s_nop 1\n v_addc_co_u32 %4, vcc, %4, %4, vcc\n")) }
        if (OP == 15) { REP16(T4("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %1, vcc\n v_cndmask_b32 %3, %3, %1, vcc\n v_cndmask_b32 %4, %4, %1, vcc\n")) }
        if (OP == 16) { REP16(T4("v_cndmask_b32_e64 %0, 0, 2, vcc\n v_cndmask_b32_e64 %2, 0, 2, vcc\n v_cndmask_b32_e64 %3, 0, 2, vcc\n v_cndmask_b32_e64 %4, 0, 2, vcc\n")) }
        if (OP == 17) { REP16(T4("v_lshl_or_b32 %0, %0, 1, %1\n v_lshl_or_b32 %2, %2, 1, %1\n v_lshl_or_b32 %3, %3, 1, %1\n v_lshl_or_b32 %4, %4, 1, %1\n")) }
        if (OP == 18) { REP16(T4("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x10\n v_bitop3_b32 %2, %2, %1, %1 bitop3:0x10\n v_bitop3_b32 %3, %3, %1, %1 bitop3:0x10\n v_bitop3_b32 %4, %4, %1, %1 bitop3:0x10\n")) }
        if (OP == 19) { REP16(T4("v_and_or_b32 %0, %0, %1, %1\n v_and_or_b32 %2, %2, %1, %1\n v_and_or_b32 %3, %3, %1, %1\n v_and_or_b32 %4, %4, %1, %1\n")) }
        if (OP == 20) { REP16(T4("v_bfe_u32 %0, %0, %1, 1\n v_bfe_u32 %2, %2, %1, 1\n v_bfe_u32 %3, %3, %1, 1\n v_bfe_u32 %4, %4, %1, 1\n")) }
        if (OP == 21) { REP16(T4("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n")) }
        if (OP == 22) { REP16(T4("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %1, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %4, %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n")) }
        if (OP == 23) { REP16(T4("v_readlane_b32 s10, %0, 63\n v_readlane_b32 s10, %2, 63\n v_readlane_b32 s10, %3, 63\n v_readlane_b32 s10, %4, 63\n")) }
        if (OP == 24) { REP16(T4("v_readfirstlane_b32 s10, %0\n v_readfirstlane_b32 s10, %2\n v_readfirstlane_b32 s10, %3\n v_readfirstlane_b32 s10, %4\n")) }
        if (OP == 25) { REP16(T4("v_add_u32 %0, s4, %0\n v_add_u32 %2, s4, %2\n v_add_u32 %3, s4, %3\n v_add_u32 %4, s4, %4\n")) }
        if (OP == 27) { REP16(T4("v_ffbl_b32 %0, %0\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3\n v_ffbl_b32 %4, %4\n")) }
        if (OP == 28) { REP16(T4("v_bcnt_u32_b32 %0, %0, %1\n v_bcnt_u32_b32 %2, %2, %1\n v_bcnt_u32_b32 %3, %3, %1\n v_bcnt_u32_b32 %4, %4, %1\n")) }
        if (OP == 29) { REP16(T4("v_mbcnt_lo_u32_b32 %0, s6, %0\n v_mbcnt_lo_u32_b32 %2, s6, %2\n v_mbcnt_lo_u32_b32 %3, s6, %3\n v_mbcnt_lo_u32_b32 %4, s6, %4\n")) }
        if (OP == 30) { REP16(T4("v_dot4_u32_u8 %0, %0, %1, %0\n v_dot4_u32_u8 %2, %2, %1, %2\n v_dot4_u32_u8 %3, %3, %1, %3\n v_dot4_u32_u8 %4, %4, %1, %4\n")) }
        if (OP == 31) { REP16(T4("v_pk_mul_lo_u16 %0, %0, %1\n v_pk_mul_lo_u16 %2, %2, %1\n v_pk_mul_lo_u16 %3, %3, %1\n v_pk_mul_lo_u16 %4, %4, %1\n")) }
        if (OP == 32) { REP16(T4("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD src0_sel:WORD_1 src1_sel:DWORD\n v_and_b32_sdwa %2, %2, %1 dst_sel:DWORD src0_sel:WORD_1 src1_sel:DWORD\n v_and_b32_sdwa %3, %3, %1 dst_sel:DWORD src0_sel:WORD_1 src1_sel:DWORD\n v_and_b32_sdwa %4, %4, %1 dst_sel:DWORD src0_sel:WORD_1 src1_sel:DWORD\n")) }
        // LDS: 4 reads per group, random dword addresses over 32 KiB (bank conflicts as in the Bloom filter) / conflict-free
        if (OP == 40) { REP16(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %5\n ds_read_b32 %2, %6\n ds_read_b32 %3, %7\n s_waitcnt lgkmcnt(0)\n" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(r0), "v"(r1), "v"(r2), "v"(r3));) r0 = (r0 + a0) & 0x7FFC; }
        if (OP == 41) { const uint32_t l = (threadIdx.x & 63) * 4; REP16(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)\n" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(l));) }
        if (OP == 42) { REP16(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %5\n ds_read_b32 %2, %6\n ds_read_b32 %3, %7\n" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(r0), "v"(r1), "v"(r2), "v"(r3));) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (OP == 43) { REP16(asm volatile("ds_read_u8 %0, %4\n ds_read_u8 %1, %5\n ds_read_u8 %2, %6\n ds_read_u8 %3, %7\n" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(r0 | 1), "v"(r1 | 2), "v"(r2 | 3), "v"(r3));) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        // mixes: what one filter position costs in the old / new formulation (VALU only, operands in registers)
        if (OP == 50) { REP16(T4("v_mul_lo_u32 %0, %0, s4\n v_lshrrev_b32 %2, 15, %0\n v_and_b32 %2, 0x1fffc, %2\n v_and_b32 %3, 0x7fc, %0\n v_and_b32 %4, %2, %3\n v_cmp_eq_u32 vcc, %4, %3\n s_nop 1\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n")) }
        if (OP == 51) { REP16(T4("v_mul_lo_u32 %0, %0, s4\n v_lshrrev_b32 %2, 15, %0\n v_and_b32 %2, 0x1fffc, %2\n v_lshrrev_b32 %3, 12, %0\n v_lshrrev_b32 %4, %0, %2\n v_lshrrev_b32 %2, 7, %0\n v_bfe_u32 %3, %2, %3, 1\n v_lshrrev_b32 %2, %2, %4\n v_bitop3_b32 %3, %3, %2, %4 bitop3:0x80\n v_lshl_or_b32 %1, %1, 1, %3\n")) }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ c ^ sh[threadIdx.x];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP> void run(const char* name, int per_iter, uint32_t* d, uint64_t* dc)
{
    const int blocks = 256, iters = 400;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, d, dc, 12345u, 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, d, dc, 12345u, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    std::vector<uint64_t> h(blocks * 16);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (uint64_t x : h) sum += (double)x;
    const double avg = sum / h.size(), insts = (double)per_iter * iters;
    printf("%-34s %7.2f cycles/instr/SIMD   (wave: %6.2f cycles/instr, clock %.2f GHz)\n", name, avg / (insts * 4.0), avg / insts, avg / (ms * 1e-3) / 1e9);
}

int main()
{
    uint32_t* d; uint64_t* dc;
    (void)hipMalloc(&d, 256 * 1024 * 4); (void)hipMalloc(&dc, 256 * 16 * 8);
    run<0>("v_add_u32 vv", 64, d, dc);
    run<1>("v_and_b32 literal", 64, d, dc);
    run<2>("v_lshrrev_b32 const", 64, d, dc);
    run<3>("v_lshrrev_b32 var", 64, d, dc);
    run<4>("v_mul_lo_u32 vv", 64, d, dc);
    run<5>("v_mul_lo_u32 sgpr", 64, d, dc);
    run<6>("v_mul_hi_u32", 64, d, dc);
    run<7>("v_mul_u32_u24", 64, d, dc);
    run<8>("v_mad_u32_u24", 64, d, dc);
    run<9>("v_alignbyte_b32", 64, d, dc);
    run<10>("v_perm_b32", 64, d, dc);
    run<11>("v_cmp_eq_u32 vcc (e32)", 64, d, dc);
    run<12>("v_cmp_eq_u32_e64 sgpr", 64, d, dc);
    run<13>("v_addc_co_u32 (e32)", 64, d, dc);
    run<14>("cmp + s_nop 1 + addc (pair=2)", 64, d, dc);
    run<15>("v_cndmask_b32 vcc (e32)", 64, d, dc);
    run<16>("v_cndmask_b32_e64 0,2,vcc", 64, d, dc);
    run<17>("v_lshl_or_b32", 64, d, dc);
    run<18>("v_bitop3_b32", 64, d, dc);
    run<19>("v_and_or_b32", 64, d, dc);
    run<20>("v_bfe_u32", 64, d, dc);
    run<21>("v_mov_b32_dpp wave_shr:1", 64, d, dc);
    run<22>("v_add_u32_dpp row_shr:1", 64, d, dc);
    run<23>("v_readlane_b32", 64, d, dc);
    run<24>("v_readfirstlane_b32", 64, d, dc);
    run<25>("v_add_u32 sgpr operand", 64, d, dc);
    run<27>("v_ffbl_b32", 64, d, dc);
    run<28>("v_bcnt_u32_b32", 64, d, dc);
    run<29>("v_mbcnt_lo sgpr", 64, d, dc);
    run<30>("v_dot4_u32_u8", 64, d, dc);
    run<31>("v_pk_mul_lo_u16", 64, d, dc);
    run<32>("v_and_b32_sdwa WORD_1", 64, d, dc);
    run<40>("ds_read_b32 random x4 + wait", 64, d, dc);
    run<41>("ds_read_b32 linear x4 + wait", 64, d, dc);
    run<42>("ds_read_b32 random x64, one wait", 64, d, dc);
    run<43>("ds_read_u8 random x64, one wait", 64, d, dc);
    run<50>("filter position, new (7 VALU)", 16 * 7, d, dc);
    run<51>("filter position, old (10 VALU)", 16 * 10, d, dc);
    return 0;
}
