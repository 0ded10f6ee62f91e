/*
 * am_debug.h -- the entry points libam.so exports for TESTS AND MEASUREMENTS, next to the product ABI of include/am.h.  Nothing here is
 * needed by (or meant for) a caller of the library: a Haskell / C host binds am.h only.  None of these functions changes a result.
 */
#ifndef AM_DEBUG_H
#define AM_DEBUG_H

#include "am.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A test / measurement switch of csrc/am_config.h by the name of its environment variable ("AM_RP_LOOP", "AM_SF_POOL_BLOCKS", ...);
 * value -1 = unset.  AM_ERR_INVALID: no such switch. */
AM_API int am_debug_set(const char* name, long value);
/* Page-locked staging memory of all threads, living or parked (the leak test). */
AM_API uint64_t am_debug_pinned_bytes(void);
/* Cycle sums per k_sf phase / per-wavefront record counts of launches made under AM_SF_ABLATE=9 (tools/phase_timing.py). */
AM_API int am_debug_sf_phase_cycles(uint64_t* out5);
AM_API int am_debug_sf_wave_records(uint64_t* out, size_t n_waves);
/* Haystacks the calling process's last one-kernel Replacer run finished with their lists in LDS (k_rp_lds); the rest took k_rp_loop. */
AM_API uint32_t am_debug_rp_lds_haystacks(void);
/* The general AC-walk kernel (k_ac) is test infrastructure and lives in libam_check.so (tests/native/am_ac.hip).  Loading that library
 * hands its launcher to libam through this call; `launcher` is am::dev::launch_ac of a build with the same csrc/am_device.h
 * (image_version must equal am_image_version()), NULL takes it away again.  Without a launcher am_automaton_set_kernel(a, 1) makes every
 * scan of `a` fail with AM_ERR_UNSUPPORTED. */
AM_API int am_debug_set_general_kernel(void* launcher, uint32_t image_version);
/* A -DAM_BOUNDS_CHECK build of the library (tools/bounds_check.sh) compiles index assertions into its kernels (csrc/am_bounds.h: LDS queue indices, pool slots,
 * image offsets).  *failed_out = assertions that failed on the current device since the library was loaded, *first_line_out = source line of the first one (0: none),
 * *checked_units_out = translation units that carry assertions: 0 in the product build, where the call reports nothing and costs nothing. */
AM_API int am_debug_bounds_report(uint64_t* failed_out, uint32_t* first_line_out, uint32_t* checked_units_out);
/* How many wavefronts does a CU of the current device really run at the same time?  A spinning kernel is launched with 16 and with 32 wavefronts per CU (workgroups of 1024 and of
 * 256 threads, 4 KiB of LDS, a handful of registers): *one_ms_out / *two_ms_out = the launch times.  Equal times: 32 are resident, as the architecture says; twice the time: the
 * second half waited for the first (seen on the GPU pool during round 6: k_dfa, k_rp_lds and the small-filter k_sf, which count on two workgroups per CU, lose 10-45 % there).
 * bench.py prints the answer as `machine.resident_waves_per_cu`. */
AM_API int am_debug_resident_waves(float* one_ms_out, float* two_ms_out);

#ifdef __cplusplus
}
#endif
#endif /* AM_DEBUG_H */
