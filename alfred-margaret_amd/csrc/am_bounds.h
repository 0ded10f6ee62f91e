// am_bounds.h -- AM_BOUNDS(cond): index assertions of the kernels, compiled in only by a -DAM_BOUNDS_CHECK build (tools/bounds_check.sh: GPU sanitizers are not
// available on this pool, so the kernels' LDS queue indices, pool slots and image offsets are checked by the kernels themselves, once per round, on the parity tests).
// A failing assertion counts itself in a device-side word of its translation unit and records the first line; am_debug_bounds_report (include/am_debug.h) sums the
// translation units.  In the product build the macro is empty and the header declares nothing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef AM_BOUNDS_CHECK
namespace am {
namespace dev {
// per translation unit (no relocatable device code): [0] failed assertions, [1] line of the first one
static __device__ uint32_t g_bounds_words[2];
__device__ __forceinline__ void bounds_fail(uint32_t line)
{
    if (atomicAdd(&g_bounds_words[0], 1u) == 0u) g_bounds_words[1] = line;
}
// the ABI layer's list of readers: every translation unit that includes this header adds its own at load time
void bounds_register(const char* file, hipError_t (*read)(uint32_t* out2));
namespace {
hipError_t bounds_read_this_tu(uint32_t* out2) { return hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_bounds_words), sizeof(uint32_t) * 2); }
struct BoundsRegistrar { BoundsRegistrar(const char* f) { bounds_register(f, &bounds_read_this_tu); } };
}  // namespace
}  // namespace dev
}  // namespace am
#define AM_BOUNDS_TU(file) namespace { am::dev::BoundsRegistrar am_bounds_registrar_(file); }
#define AM_BOUNDS(cond) do { if (!(cond)) am::dev::bounds_fail(__LINE__); } while (0)
#else
#define AM_BOUNDS_TU(file)
#define AM_BOUNDS(cond) ((void)0)
#endif
