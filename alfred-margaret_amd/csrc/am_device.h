// am_device.h -- interface between the C-ABI layer (am_abi.cpp) and the kernels (am_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "am_image.h"

namespace am {
namespace dev {

constexpr int kModeCount = 0;   // unit_counts[u] = records of unit u; optional per-haystack value counts; total values
constexpr int kModeEmit = 1;    // write records at unit_offsets[u]
constexpr int kModeAny = 2;     // flags[haystack] = 1 if anything matches

// 16-byte match record in HBM.  Same layout as am_match in include/am.h.
struct alignas(16) Record { uint64_t end_pos; uint32_t haystack; uint32_t state; };

struct ScanOut {
    // general (AC) kernel, two passes: count pass fills unit_counts[u], emit pass writes at unit_offsets[u]
    uint32_t* unit_counts;
    const uint64_t* unit_offsets;
    Record* records;
    // suffix-filter kernel, single pass: records go to 64-record blocks taken from a pool with one
    // atomic per block; the blocks of a unit form a linked list (block_next); k_permute then copies
    // every unit's list to its final place (unit_offsets = exclusive scan of unit_counts)
    Record* pool;
    uint32_t* block_next;
    uint32_t* pool_ctrl;          // [0] next free block (keeps counting past n_blocks = blocks needed), [1] overflow flag
    uint32_t* unit_first;         // first block of each unit (kNone: none)
    uint32_t n_blocks;
    uint32_t unit_chunks;         // 1-KiB chunks per unit
    // all kernels
    uint64_t* hay_counts;         // count mode, may be null
    uint64_t* total_values;       // count mode
    uint8_t* flags;               // any mode
    uint32_t ablate;              // timing experiments only (AM_SF_ABLATE); 0 in production
    uint64_t* dbg;                // timing experiments only: per-phase cycle sums
};

constexpr uint32_t kPoolBlock = 64;   // records per pool block (1 KiB)

hipError_t launch_hidx(const BatchView& b, uint32_t* hidx, uint64_t n_entries, hipStream_t st);
uint64_t sf_chunks(const BatchView& b);
uint32_t sf_unit_chunks(const BatchView& b, int n_cu);
hipError_t launch_permute(const ScanOut& o, const uint64_t* unit_offsets, Record* out, uint64_t n_units, hipStream_t st);
uint64_t ac_units(const AcView& a, const BatchView& b);
size_t sf_lds_bytes(const SfView& s);
hipError_t launch_sf(bool ic, int mode, const SfView& s, const BatchView& b, const ScanOut& o, int n_cu, hipStream_t st);
hipError_t launch_ac(bool ic, int mode, const AcView& a, const BatchView& b, const ScanOut& o, hipStream_t st);
hipError_t read_sf_phase_cycles(uint64_t* out5);
hipError_t scan_temp_bytes(uint64_t n, size_t* bytes);
hipError_t launch_scan(void* temp, size_t temp_bytes, const uint32_t* counts, uint64_t* offsets, uint64_t n, hipStream_t st);

}  // namespace dev
}  // namespace am
