// am_wave.h -- wave64 helpers shared by the HIP kernels of libam (gfx950): lane id, DPP prefix sum, uniform values, LDS by absolute address
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace am {
namespace dev {

// ------------------------------------------------------------------ wave helpers (wave64)

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// inclusive prefix sum over the 64 lanes with DPP adds (no LDS round trips): Kogge-Stone inside each
// 16-lane row (row_shr 1,2,4,8), then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x, uint32_t /*lane*/)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);   // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return x;
}

// a value that is the same in every lane, moved to scalar registers (the two v_readfirstlane also force any load that
// produces it to be waited for right here)
__device__ __forceinline__ uint64_t uniform_u64(uint64_t x)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t x)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d, 64);
    return x;   // valid in lane 0
}

// order LDS traffic between lanes of one wavefront (DS ops of a wave execute in issue order;
// this only stops the compiler from moving them)
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS by absolute byte address.  k_sf declares no static LDS, so its dynamic LDS starts at address 0; reading through an
// address-space-3 pointer made from an integer lets the compiler put constant parts into the instruction's offset field
// instead of adding the (relocatable, always zero) base of the extern array to every address (16 v_add per chunk).
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) uint16_t lds_u16_t;
typedef uint32_t u32x2_n __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_n __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x2_n lds_u32x2_t;
typedef __attribute__((address_space(3))) u32x4_n lds_u32x4_t;
__device__ __forceinline__ uint32_t lds_read_u32(uint32_t byte_addr) { return *reinterpret_cast<const lds_u32_t*>((uintptr_t)byte_addr); }
__device__ __forceinline__ uint32_t lds_read_u16(uint32_t byte_addr) { return *reinterpret_cast<const lds_u16_t*>((uintptr_t)byte_addr); }
__device__ __forceinline__ void lds_write_u16(uint32_t byte_addr, uint32_t v) { *reinterpret_cast<lds_u16_t*>((uintptr_t)byte_addr) = (uint16_t)v; }
__device__ __forceinline__ void lds_write_u32x2(uint32_t byte_addr, uint2 v) { u32x2_n t; t.x = v.x; t.y = v.y; *reinterpret_cast<lds_u32x2_t*>((uintptr_t)byte_addr) = t; }
__device__ __forceinline__ void lds_write_u32x4(uint32_t byte_addr, uint4 v) { u32x4_n t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *reinterpret_cast<lds_u32x4_t*>((uintptr_t)byte_addr) = t; }


}  // namespace dev
}  // namespace am
