// am_device.h -- interface between the C-ABI layer (am_abi.cpp) and the kernels (am_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "am_image.h"

namespace am {
namespace dev {

constexpr int kModeCount = 0;   // unit_counts[u] = records of unit u; optional per-haystack value counts; total values
constexpr int kModeEmit = 1;    // write records at unit_offsets[u]
constexpr int kModeAny = 2;     // flags[haystack] = 1 if anything matches
constexpr int kModeIds = 3;     // Searcher.containsAll: the needle ids of everything that matches into the haystack's bitmap row; flags[haystack] = 1 when the row is full
                                // (k_sf only: a flag mode like kModeAny -- a flagged haystack is not looked at any further)

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
    uint32_t* unit_slots;         // record SLOTS in each unit's chain: its records (unit_counts) + the slots of parked walkers that found nothing
                                  // (state == kNone; k_permute drops them)
    uint32_t wq_iters;            // set by launch_sf: trie steps a resolve batch takes before it parks the walkers that are not done (2)
    uint32_t wq_cap;              // set by launch_sf: walker-queue entries per wavefront in LDS (0: the filter leaves no room)
    uint32_t n_blocks;
    uint32_t unit_chunks;         // 1-KiB chunks per unit
    uint32_t* next_unit;          // zeroed before the launch: wavefronts draw their second and later units from it (null: fixed stride)
    // all kernels
    uint64_t* hay_counts;         // count mode, may be null
    uint64_t* total_values;       // count mode
    uint8_t* flags;               // any mode; ids mode: the haystack's row is full
    // ids mode (containsAll): machineValues in flat form = the needle ids a state reports; one row of ids_words 32-bit words per haystack;
    // ids_missing[h] = ids of haystack h not seen yet (starts at the number of needles: the bit that brings it to 0 raises flags[h])
    const uint64_t* ids_vals_off; const uint32_t* ids_vals; uint32_t* ids_bits; uint32_t* ids_missing; uint32_t ids_words, ids_n;
    uint32_t probe_two;           // A/B (AM_SF_PROBE_TWO=1): the two-candidates-per-lane instantiation also for automata with few 4-byte-suffix keys
    uint32_t ablate;              // timing experiments only (AM_SF_ABLATE); 0 in production
    uint64_t* dbg;                // timing experiments only: per-phase cycle sums
};

constexpr uint32_t kPoolBlock = 64;   // records per pool block (1 KiB)
constexpr uint32_t kSfBlockGrant = 32; // pool blocks a k_sf wavefront takes per atomic
// blocks that may stay unused in the wavefronts' grants: one grant per wavefront that can be resident (<= 32 per CU), never more than units
// (wavefronts launched = 16 per workgroup, one workgroup per CU -- two when the filter is small -- and never more workgroups than 16-unit groups)
inline uint64_t pool_grant_slack(int n_cu, uint64_t n_units, bool two_per_cu)
{
    uint64_t w = (uint64_t)n_cu * (two_per_cu ? 32u : 16u);
    const uint64_t need = (n_units + 15u) / 16u * 16u;
    if (w > need) w = need;
    return w * kSfBlockGrant;
}

// z0 / z1 (nullable): up to two arrays of n0 / n1 dwords that the same launch clears
hipError_t launch_hidx(const BatchView& b, uint32_t* hidx, uint64_t n_entries, hipStream_t st, uint32_t* z0 = nullptr, uint64_t n0 = 0, uint32_t* z1 = nullptr, uint64_t n1 = 0);
uint64_t sf_chunks(const BatchView& b);
uint32_t sf_unit_chunks(const BatchView& b, int n_cu);
hipError_t launch_permute(const ScanOut& o, const uint64_t* unit_offsets, Record* out, uint64_t n_units, hipStream_t st);
uint64_t ac_units(const AcView& a, const BatchView& b);
size_t sf_lds_bytes(const SfView& s);
hipError_t launch_sf(bool ic, int mode, const SfView& s, const BatchView& b, const ScanOut& o, int n_cu, hipStream_t st);
hipError_t launch_ac(bool ic, int mode, const AcView& a, const BatchView& b, const ScanOut& o, hipStream_t st);
// table-walk kernel (am_dfa.hip): same two-pass protocol as the general kernel (count -> scan -> emit), unit = one lane's DfaView::chunk bytes
uint64_t dfa_units(const DfaView& d, const BatchView& b);
bool dfa_usable(const DfaView& d);      // on the current device: offsets fit, the LDS attribute could be raised (checked once per device)
hipError_t launch_dfa(int mode, const DfaView& d, const BatchView& b, const ScanOut& o, int n_cu, hipStream_t st);
// records in ONE walk: 8-byte tokens into ScanOut::pool (superblocks; ScanOut::block_next = 2 x n_blocks words: their fill counts, zeroed before the launch, then their
// first groups; pool_ctrl[0] superblocks drawn, [1] pool exhausted; n_blocks = superblocks in the pool), unit_counts as in count mode; then scan(unit_counts) and launch_dfa_place
// an estimate of the needle ends per byte of a batch: n_samples lanes spread over the text walk len bytes each and add their count to *out (zeroed by the caller)
hipError_t launch_dfa_sample(const DfaView& d, const uint8_t* text, uint64_t total, uint32_t n_samples, uint32_t len, uint32_t* out, hipStream_t st);
bool dfa_tokens_ok(const DfaView& d);
uint32_t dfa_token_waves(const DfaView& d, const BatchView& b, int n_cu);
uint64_t dfa_token_superblocks(uint64_t records, uint32_t n_waves, uint64_t n_units);
uint64_t dfa_superblock_bytes();
hipError_t launch_dfa_tokens(const DfaView& d, const BatchView& b, const ScanOut& o, int n_cu, hipStream_t st);
hipError_t launch_dfa_place(const DfaView& d, const BatchView& b, const ScanOut& o, uint32_t n_super, const uint64_t* unit_offsets, int n_cu, uint32_t n_waves, uint32_t n_ref_states, Record* out, hipStream_t st);
hipError_t read_sf_phase_cycles(uint64_t* out5);
hipError_t read_sf_wave_records(uint64_t* out, size_t n_waves);
hipError_t scan_temp_bytes(uint64_t n, size_t* bytes);
hipError_t launch_scan(void* temp, size_t temp_bytes, const uint32_t* counts, uint64_t* offsets, uint64_t n, hipStream_t st);


// ---- automata with the empty needle: dense pass over the suffix-filter kernel's records (am_dense.hip)
hipError_t launch_dense(bool ic, bool write, const AcView& a, const BatchView& b, const Record* sparse, const uint64_t* sparse_offsets, uint32_t unit_chunks,
                        uint64_t n_units, uint32_t* unit_totals, const uint64_t* out_offsets, Record* out, hipStream_t st);
hipError_t launch_records_reduce(const Record* recs, uint64_t n, const uint32_t* vlen, uint64_t* hay_counts, uint64_t* total, uint8_t* flags, hipStream_t st);
// one haystack scanned in ranges (am_run_range): out2 = {records with end_pos <= x0, records with end_pos <= x1}; end_pos += add
hipError_t launch_range_bounds(const Record* recs, uint64_t n, uint64_t x0, uint64_t x1, uint64_t* out2, hipStream_t st);
hipError_t launch_range_rebase(Record* recs, uint64_t n, uint64_t add, hipStream_t st);
hipError_t launch_hay_rebase(Record* recs, uint64_t n, uint32_t add, hipStream_t st);
hipError_t launch_spin(uint32_t workgroups, uint32_t threads, uint64_t cycles, uint32_t* out, hipStream_t st);      // am_debug_resident_waves

// ---- Replacer pass (am_replace.hip) ----------------------------------------------------------
// same layout as am_payload in include/am.h (Replacer.hs:59-70 Payload, replacement text as a slice of one blob)
struct RpPayload { int64_t priority; uint32_t len_bytes; uint32_t len_code_points; uint64_t repl_off; uint32_t repl_len; uint32_t reserved; };
// machineValues of the Replacer's automaton in CSR form: the list of state s is payloads[vals[vals_off[s] .. vals_off[s+1])]
// per state, 8 bytes: for the states that carry exactly ONE value (almost all of them) that value's priority and payload index, so that the fold
// reads one small entry -- the table of a 50k-needle automaton stays in L2 -- instead of walking state -> value list -> payload (three dependent
// loads); payload == kRpWalkList: several values (or a priority beyond 32 bits): walk the list
struct RpStateOne { int32_t priority; uint32_t payload; };
constexpr uint32_t kRpWalkList = 0xFFFFFFFFu;
struct RpTables { const uint64_t* vals_off; const uint32_t* vals; const RpPayload* payloads; const uint8_t* repl; int64_t min_priority; const RpStateOne* one; };
struct RpKept { uint64_t src_start, src_len, dst; };     // a match that survives removeOverlap; dst = where its replacement starts in the new text
constexpr uint32_t kRpActive = 0, kRpFinished = 1, kRpNothing = 2;
struct RpHay { uint64_t newlen; int64_t best; uint32_t status, nkept, payload, pad; };
struct RpFin { uint64_t off, len; uint32_t orig, status; };
struct RpRoute { uint64_t* len_next; uint64_t* len_fin; uint32_t* tiles; uint32_t* act; uint32_t* fin; };                       // per haystack, written by k_rp_pass
struct RpRouted { const uint64_t* off_next; const uint64_t* off_fin; const uint64_t* tile_off; const uint64_t* act_idx; const uint64_t* fin_idx; };   // their exclusive sums
constexpr uint64_t kRpTile = 16384;                      // bytes of new text per k_rp_splice workgroup

hipError_t launch_rp_ranges(const Record* recs, uint64_t n_rec, uint64_t* rec_first, const RpRoute& route, uint32_t n_act, hipStream_t st);
hipError_t launch_rp_ranges_dev(const Record* recs, const uint64_t* n_rec_dev, uint64_t* rec_first, const RpRoute& route, uint32_t n_act, hipStream_t st);
hipError_t launch_rp_totals(const RpRouted& rt, uint32_t n_act, const uint64_t* win_off, const uint64_t* woffs, uint64_t woffs_last, uint64_t* out10, hipStream_t st,
                            const uint64_t* extra8 = nullptr, const uint64_t* extra9 = nullptr, uint64_t seq = 0);      // out10[8], out10[9] = *extra8, *extra9 (0 when null); seq != 0: out10 is pinned host memory, out10[15] = seq last
// what k_rp_ranges (before the fold) and k_pt_count (after it) would write, done by k_rp_pass itself: one dispatch instead of three in the
// chain of a Replacer pass.  rec_first_w == nullptr: not fused (rec_first is read).
struct RpFused { uint64_t* rec_first_w; uint64_t n_rec; const uint64_t* n_rec_dev; const uint32_t* pc_cnt; uint32_t* need; uint32_t* nwin; };
hipError_t launch_rp_pass(bool ic, const RpTables& t, const uint8_t* text, const uint64_t* offsets, const Record* recs, const uint64_t* rec_first,
                          const int64_t* thr, uint64_t max_len, RpKept* kept, RpHay* hs, const RpRoute& route, uint32_t n_act, uint32_t keep_all, hipStream_t st,
                          const RpFused* fused = nullptr);
struct RpSelected { uint64_t start, len; uint32_t haystack, payload; };     // = am_prio_match in include/am.h
hipError_t launch_rp_gather(const RpHay* hs, const uint64_t* rec_first, const RpKept* kept, const uint64_t* out_off, RpSelected* out, int64_t* best_out,
                            uint32_t n_act, hipStream_t st);
hipError_t launch_rp_route(const RpHay* hs, const RpRouted& rt, const uint32_t* orig, uint32_t n_act, uint64_t* next_offsets, uint32_t* next_orig,
                           int64_t* next_thr, RpFin* fin, hipStream_t st);
hipError_t launch_rp_splice(const RpTables& t, const uint8_t* text, const uint64_t* offsets, const uint64_t* rec_first, const RpKept* kept,
                            const RpHay* hs, const RpRouted& rt, uint32_t n_act, uint64_t n_tiles, uint32_t* tile_hay /* n_tiles entries, scratch */,
                            uint8_t* text_next, uint8_t* text_fin, hipStream_t st);
hipError_t launch_idset(const Record* recs, uint64_t r0, uint64_t r1, const uint64_t* vals_off, const uint32_t* vals, uint32_t n_needles,
                        uint32_t hay0, uint32_t words, uint32_t* bits, hipStream_t st);
hipError_t launch_fold_hash(const Record* recs, const uint64_t* rec_first, const uint64_t* vals_off, const uint32_t* vals, uint32_t n_hay,
                            uint64_t* hash_out, uint64_t* count_out, hipStream_t st);
hipError_t launch_idset_all(const uint32_t* bits, uint32_t words, uint32_t n_needles, uint32_t n_hay, uint8_t* flags, hipStream_t st);
// incremental re-scan between Replacer passes (am_replace.hip)
struct RpWin { uint64_t src_abs; uint64_t ws; uint32_t len; uint32_t own_lo; };   // window: bytes src_abs.. of the next text; ws = its start inside the haystack; records with end > own_lo are its own
hipError_t launch_rp_win_count(const RpHay* hs, uint32_t n_act, uint32_t* nwin, hipStream_t st);
hipError_t launch_rp_win_meta(const RpTables& t, const RpRouted& rt, const RpHay* hs, const uint64_t* rec_first,
                              const RpKept* kept, const uint64_t* win_off, uint32_t ov, RpWin* wins, uint32_t* wlen, uint32_t n_act, hipStream_t st, bool pt = false);
// piece table (am_replace.hip): the text of a haystack between passes as (source, logical start) pairs + a sentinel
struct RpPiece { uint64_t src; uint64_t lstart; };
constexpr uint64_t kPieceRepl = 1ull << 63;             // RpPiece::src: offset into the replacement blob instead of the batch text
// all passes of a haystack in one kernel (am_rploop.hip): every haystack owns two record lists, two piece lists, a kept list and a window scratch
struct RpLoopOut { uint64_t len; uint64_t pieces_at; uint32_t n_pieces, status, passes, pad; };      // pieces_at: index of the final list in pc_buf
struct RpLoop {
    RpTables t; SfView s;
    const uint8_t* text; const uint64_t* offsets; uint32_t n_hay, ov;
    const Record* recs0; const uint64_t* rec_first0;        // the first scan's sorted records and their range per haystack
    Record* rec_buf; const uint64_t* rec_base;              // haystack h: records [rec_base[h], rec_base[h + 1]), two halves
    RpPiece* pc_buf; const uint64_t* pc_base;               // the same for its piece lists
    RpKept* kept_buf;                                       // haystack h: from rec_base[h] / 2, half a record region long
    uint8_t* wtext; uint32_t wcap, pad;                     // haystack h: wcap bytes of window scratch; pad != 0: the instrumented instantiation
    uint64_t max_len;
    RpLoopOut* out;
    uint32_t* ctrl;                                         // [0] overflow, [1] passes (max), [2..3] window bytes scanned, [5] watchdog: the loop that ran out of time (100 +: in k_rp_lds), [6] the longest record list of the batch (k_rp_loop_caps), [7] haystacks k_rp_lds finished, [8..23] eight 64-bit phase sums of the instrumented instantiation
    uint32_t* redo;                                         // per haystack: 1 = k_rp_lds (LDS-resident lists, am_rplds.hip) gave it up, k_rp_loop runs it; null: k_rp_loop runs every haystack
    uint32_t h_first, pl_implicit;                          // the launch covers haystacks h_first + blockIdx.x (groups of a batch, one launch each); pl_implicit: every payload's
                                                            // priority is minus its index (Replacer.hs:100-104): k_rp_lds needs no payload column (am_rplds.hip, PLI)
};
hipError_t launch_rp_loop_caps(const uint64_t* rec_first, uint32_t n_hay, uint32_t* cap_r2, uint32_t* cap_p2, uint32_t* max_records /* atomic max, cleared by the caller */, hipStream_t st);
hipError_t launch_rp_loop(bool ic, const RpLoop& a, uint32_t n /* haystacks from a.h_first */, int waves_per_simd, hipStream_t st);
hipError_t launch_rp_lds(bool ic, const RpLoop& a, uint32_t n /* haystacks from a.h_first */, hipStream_t st);
hipError_t launch_pt_init(const uint64_t* offsets, uint32_t n_act, RpPiece* pieces, uint64_t* pc_start, uint32_t* pc_cnt, hipStream_t st);
hipError_t launch_pt_count(const RpHay* hs, const uint32_t* pc_cnt, uint32_t n_act, uint32_t* need, uint32_t* nwin, hipStream_t st);
hipError_t launch_pt_build(const RpTables& t, const RpHay* hs, const uint64_t* rec_first, const RpKept* kept, const RpPiece* pieces, const uint64_t* pc_start,
                           const uint32_t* pc_cnt, const uint64_t* need_off, const RpRouted& rt, uint32_t n_act, RpPiece* out, uint64_t* next_start, uint32_t* next_cnt,
                           uint64_t* fin_start, uint32_t* fin_cnt, hipStream_t st);
hipError_t launch_pt_materialise(const RpPiece* pieces, const uint64_t* fin_start, const uint32_t* fin_cnt, const RpFin* fin, uint32_t n_fin, const uint8_t* text,
                                 const uint8_t* repl, uint8_t* text_fin, hipStream_t st);
hipError_t launch_pt_materialise_next(const RpPiece* pieces, const uint64_t* next_start, const uint32_t* next_cnt, const uint64_t* next_offsets, uint32_t n_next,
                                      const uint8_t* text, const uint8_t* repl, uint8_t* out, hipStream_t st);
hipError_t launch_pt_win_copy(const RpWin* wins, const uint64_t* woffs, const RpPiece* pieces, const uint64_t* next_start, const uint32_t* next_cnt, const uint8_t* text,
                              const uint8_t* repl, uint8_t* wtext, uint64_t n_win, uint64_t total_w, uint64_t padded, hipStream_t st);
hipError_t launch_rp_win_copy(const RpWin* wins, const uint64_t* woffs, const uint8_t* text_next, uint8_t* wtext, uint64_t n_win, hipStream_t st);
hipError_t launch_rp_merge(bool write, const Record* recs, const uint64_t* rec_first, const RpKept* kept, const RpHay* hs, const uint64_t* offsets, const RpRouted& rt,
                           const uint64_t* win_off, const RpWin* wins, const Record* wrecs, const uint64_t* wrec_first, uint32_t ov, uint32_t n_act,
                           uint32_t* mcount, const uint64_t* moff, Record* out, hipStream_t st);
// the record-parallel variant of the fold (few haystacks with very many matches), same outputs as launch_rp_pass
struct RpSel { uint64_t start, len; uint32_t haystack, pad; };
hipError_t launch_rpp_best(const RpTables& t, const Record* recs, uint64_t n_rec, const int64_t* thr, int64_t* best /* n_act + 1 */, uint32_t n_act, hipStream_t st);
hipError_t launch_rpp_select(bool ic, const RpTables& t, const uint8_t* text, const uint64_t* offsets, const Record* recs, uint64_t n_rec, const int64_t* best,
                             uint32_t* selflag, RpSel* cand, int64_t* delta_all, uint32_t* payload_of, hipStream_t st);
hipError_t launch_rpp_compact(const uint32_t* selflag, const uint64_t* sidx, const RpSel* cand, uint64_t n_rec, RpSel* sel, hipStream_t st);
hipError_t launch_rpp_overlaps(const RpSel* sel, const uint64_t* n_sel_dev, uint64_t bound, uint32_t* keep, hipStream_t st);
hipError_t launch_rpp_kflags(const RpSel* sel, const uint64_t* n_sel_dev, uint64_t bound, const uint32_t* keep, const RpTables& t, const uint32_t* payload_of,
                             uint32_t* kflag, uint64_t* kdelta, hipStream_t st);
hipError_t launch_rpp_finish(const RpTables& t, const RpSel* sel, const uint64_t* n_sel_dev, uint64_t bound, const uint32_t* kflag, const uint64_t* kidx,
                             const uint64_t* kdpre, const uint64_t* sidx, const uint64_t* offsets, const uint64_t* rec_first, const int64_t* best,
                             const int64_t* delta_all, const uint32_t* payload_of, uint64_t max_len, RpKept* kept, RpHay* hs, const RpRoute& route, uint32_t n_act,
                             hipStream_t st);
// several small exclusive sums in one launch: out[i] = sum of in[0..i), for i < n (+ *n_dev when given)
struct ScanJob { const uint32_t* in32; const uint64_t* in64; uint64_t* out; uint64_t n; const uint64_t* n_dev; };
struct ScanJobs { ScanJob j[6]; uint32_t n_jobs; };
hipError_t launch_scan_jobs(const ScanJobs& jobs, hipStream_t st);
hipError_t scan64_temp_bytes(uint64_t n, size_t* bytes);
hipError_t launch_scan64(void* temp, size_t temp_bytes, const uint64_t* in, uint64_t* out, uint64_t n, hipStream_t st);

}  // namespace dev
}  // namespace am
