// am_synth.hip -- libam_synth.so: synthetic haystack generation on the device (bench input born in
// HBM) and, from the same source, on the host (CPU baseline / tests).  Benchmark support only.
#include <hip/hip_runtime.h>

#include "am_synth.h"

using namespace amsynth;

__global__ void k_synth(Params p, const uint8_t* __restrict__ needle_bytes, const uint64_t* __restrict__ needle_offs,
                        uint64_t first_cell, uint64_t n_cells, uint8_t* __restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cells) return;
    generate_cell(p, needle_bytes, needle_offs, first_cell + i, out + i * p.cell_bytes);
}

extern "C" {

// d_out: n_cells * cell_bytes bytes in HBM; d_needle_*: needle table in HBM.  Returns hipError_t.
int amsynth_generate_device(uint64_t seed, uint32_t mode, uint32_t cell_bytes, const void* d_needle_bytes, const void* d_needle_offs,
                            uint32_t n_needles, uint64_t first_cell, uint64_t n_cells, void* d_out, void* stream)
{
    Params p{seed, cell_bytes, mode, n_needles, 1, 0, 0, nullptr, nullptr, nullptr};
    const uint64_t blocks = (n_cells + 63) / 64;
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(k_synth, dim3((uint32_t)blocks), dim3(64), 0, (hipStream_t)stream, p, (const uint8_t*)d_needle_bytes,
                       (const uint64_t*)d_needle_offs, first_cell, n_cells, (uint8_t*)d_out);
    return (int)hipGetLastError();
}

// the robustness-sweep variants: `plants` needles per cell, or natural text from a vocabulary (kind 1; all tables in HBM)
int amsynth_generate_device_ex(uint64_t seed, uint32_t mode, uint32_t cell_bytes, const void* d_needle_bytes, const void* d_needle_offs,
                               uint32_t n_needles, uint64_t first_cell, uint64_t n_cells, void* d_out, void* stream,
                               uint32_t plants, uint32_t kind, const void* d_vocab_bytes, const void* d_vocab_offs, const void* d_quantile, uint32_t n_quantile)
{
    Params p{seed, cell_bytes, mode, n_needles, plants, kind, n_quantile, (const uint8_t*)d_vocab_bytes, (const uint64_t*)d_vocab_offs, (const uint32_t*)d_quantile};
    const uint64_t blocks = (n_cells + 63) / 64;
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(k_synth, dim3((uint32_t)blocks), dim3(64), 0, (hipStream_t)stream, p, (const uint8_t*)d_needle_bytes,
                       (const uint64_t*)d_needle_offs, first_cell, n_cells, (uint8_t*)d_out);
    return (int)hipGetLastError();
}

void amsynth_generate_host_ex(uint64_t seed, uint32_t mode, uint32_t cell_bytes, const uint8_t* needle_bytes, const uint64_t* needle_offs,
                              uint32_t n_needles, uint64_t first_cell, uint64_t n_cells, uint8_t* out,
                              uint32_t plants, uint32_t kind, const uint8_t* vocab_bytes, const uint64_t* vocab_offs, const uint32_t* quantile, uint32_t n_quantile)
{
    Params p{seed, cell_bytes, mode, n_needles, plants, kind, n_quantile, vocab_bytes, vocab_offs, quantile};
    for (uint64_t i = 0; i < n_cells; i++) generate_cell(p, needle_bytes, needle_offs, first_cell + i, out + i * cell_bytes);
}

void amsynth_generate_host(uint64_t seed, uint32_t mode, uint32_t cell_bytes, const uint8_t* needle_bytes, const uint64_t* needle_offs,
                           uint32_t n_needles, uint64_t first_cell, uint64_t n_cells, uint8_t* out)
{
    Params p{seed, cell_bytes, mode, n_needles, 1, 0, 0, nullptr, nullptr, nullptr};
    for (uint64_t i = 0; i < n_cells; i++) generate_cell(p, needle_bytes, needle_offs, first_cell + i, out + i * cell_bytes);
}

}  // extern "C"
