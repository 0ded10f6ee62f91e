// am_ac.hip -- TEST INFRASTRUCTURE, not part of libam.so: k_ac, the reference's state machine walked one lane per chunk (Automaton.hs:482-520;
// the walk itself is ac_scan_unit in csrc/am_image.h, shared with the host image interpreter), built into libam_check.so.  It is the independent
// second algorithm of the parity gate (bench.py) and of the GPU tests: 60 x slower than k_sf, and no automaton is routed to it by the product.
// Loading libam_check.so registers launch_ac with libam (am_debug_set_general_kernel, include/am_debug.h); without it
// am_automaton_set_kernel(a, 1) makes every scan of `a` fail with AM_ERR_UNSUPPORTED.
#include <hip/hip_runtime.h>

#include "am_debug.h"
#include "am_device.h"
#include "am_wave.h"

namespace am {
namespace dev {

// ------------------------------------------------------------------ AC kernel

struct EmitCount {
    uint32_t nrec; uint64_t nval; uint64_t* hay_counts;
    __device__ __forceinline__ void operator()(uint32_t hay, uint64_t, uint32_t, uint32_t vlen)
    {
        nrec++; nval += vlen;
        if (hay_counts) atomicAdd(reinterpret_cast<unsigned long long*>(hay_counts + hay), (unsigned long long)vlen);
    }
};
struct EmitWrite {
    Record* out;
    __device__ __forceinline__ void operator()(uint32_t hay, uint64_t end_pos, uint32_t state, uint32_t) { *out++ = Record{end_pos, hay, state}; }
};
struct EmitFlag {
    uint8_t* flags;
    __device__ __forceinline__ void operator()(uint32_t hay, uint64_t, uint32_t, uint32_t) { flags[hay] = 1; }
};

template <bool IC, int MODE>
__global__ __launch_bounds__(256) void k_ac(AcView a, BatchView b, ScanOut o, uint64_t n_units)
{
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t nval = 0;
    if (u < n_units) {
        if (MODE == kModeCount) {
            EmitCount e{0, 0, o.hay_counts};
            ac_scan_unit<IC>(a, b, u, e);
            o.unit_counts[u] = e.nrec;
            nval = e.nval;
        } else if (MODE == kModeEmit) {
            EmitWrite e{o.records + o.unit_offsets[u]};
            ac_scan_unit<IC>(a, b, u, e);
        } else {
            EmitFlag e{o.flags};
            ac_scan_unit<IC>(a, b, u, e);
        }
    }
    if (MODE == kModeCount) {
        nval = wave_sum_u64(nval);
        if (lane_id() == 0 && nval) atomicAdd(reinterpret_cast<unsigned long long*>(o.total_values), (unsigned long long)nval);
    }
}

template <bool IC, int MODE>
static hipError_t launch_ac_t(const AcView& a, const BatchView& b, const ScanOut& o, hipStream_t st)
{
    const uint64_t n_units = (b.total + a.chunk - 1) / a.chunk;      // = ac_units (csrc/am_kernels.hip): what libam sized the unit arrays for
    if (n_units == 0) return hipSuccess;
    const uint64_t blocks = (n_units + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_ac<IC, MODE>), dim3((uint32_t)blocks), dim3(256), 0, st, a, b, o, n_units);
    return hipGetLastError();
}

hipError_t launch_ac(bool ic, int mode, const AcView& a, const BatchView& b, const ScanOut& o, hipStream_t st)
{
    if (ic) {
        if (mode == kModeCount) return launch_ac_t<true, kModeCount>(a, b, o, st);
        if (mode == kModeEmit) return launch_ac_t<true, kModeEmit>(a, b, o, st);
        return launch_ac_t<true, kModeAny>(a, b, o, st);
    }
    if (mode == kModeCount) return launch_ac_t<false, kModeCount>(a, b, o, st);
    if (mode == kModeEmit) return launch_ac_t<false, kModeEmit>(a, b, o, st);
    return launch_ac_t<false, kModeAny>(a, b, o, st);
}

}  // namespace dev
}  // namespace am

// the registration runs when the library is loaded (ctypes.CDLL / dlopen); am_check_registered() tells a caller whether libam took it
static int g_registered = 0;
__attribute__((constructor)) static void am_check_register()
{
    g_registered = am_debug_set_general_kernel((void*)&am::dev::launch_ac, am::kImageVersion) == 0 ? 1 : 0;
}
extern "C" __attribute__((visibility("default"))) int am_check_registered(void) { return g_registered; }
