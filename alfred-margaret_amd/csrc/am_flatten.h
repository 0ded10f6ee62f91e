// am_flatten.h -- reference packed automaton -> device image (see am_image.h, am_flatten.cpp)
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <unordered_map>
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

// Simple lower-casing as data (Utf8.hs:145-151 lowerCodePoint / Utf8/Unlower.hs:26-40 unlowerCodePoint): the reference's table is whatever
// Data.Char.toLower of the GHC that built it says, so a caller may supply its own pairs (am_automaton_create_ex); ASCII is A-Z -> a-z always.
struct LowerTable {
    std::vector<uint32_t> from, to;                          // sorted by `from`; the 26 ASCII pairs + the non-ASCII pairs given
    std::unordered_multimap<uint32_t, uint32_t> inverse;     // to -> from
    uint32_t hash = 0;                                       // of the pairs: recorded in ImageHeader::flags (never 0)
    static int make(const uint32_t* from, const uint32_t* to, size_t n, LowerTable& out, std::string& err);
    uint32_t lower(uint32_t cp) const;
    void unlower(uint32_t cp, std::vector<uint32_t>& out) const;
};
const LowerTable& builtin_lower_table();                     // Unicode 14.0 (unicode_lower_tbl.inc)

// lower_table == nullptr: the built-in table
int flatten(const RefArrays& ref, int case_mode, std::vector<uint8_t>& image, std::string& err, const LowerTable* lower_table = nullptr);

// ImageHeader::checksum: of everything after the header (checked when an image comes from the host)
uint64_t image_checksum(const uint8_t* p, size_t n);
// magic / version / every section inside total_bytes
bool image_sections_in_bounds(const ImageHeader& h);
// every index inside the body stays inside its table (images that come from a file)
bool image_body_valid(const uint8_t* image, const ImageHeader& h, std::string& err);

uint32_t simple_lower(uint32_t cp);
void unlower(uint32_t cp, std::vector<uint32_t>& out);

}  // namespace am
