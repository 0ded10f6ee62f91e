// am_flatten.h -- reference packed automaton -> device image (see am_image.h, am_flatten.cpp)
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "am_image.h"

namespace am {

// The reference's AcMachine fields (Automaton.hs:108-123), borrowed.
struct RefArrays {
    const uint64_t* transitions; size_t n_transitions;   // machineTransitions
    const uint32_t* offsets;                             // machineOffsets (n_states + 1 entries, :170)
    size_t n_states;
    const uint64_t* root_ascii;                          // machineRootAsciiTransitions (128 entries)
    const uint32_t* values_len;                          // length (machineValues ! s)
};

int flatten(const RefArrays& ref, int case_mode, std::vector<uint8_t>& image, std::string& err);

// ImageHeader::checksum: of everything after the header (checked when an image comes from the host)
uint64_t image_checksum(const uint8_t* p, size_t n);
// magic / version / every section inside total_bytes
bool image_sections_in_bounds(const ImageHeader& h);
// every index inside the body stays inside its table (images that come from a file)
bool image_body_valid(const uint8_t* image, const ImageHeader& h, std::string& err);

uint32_t simple_lower(uint32_t cp);
void unlower(uint32_t cp, std::vector<uint32_t>& out);

}  // namespace am
