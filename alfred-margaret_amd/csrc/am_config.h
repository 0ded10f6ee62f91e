// am_config.h -- the library's test / measurement switches in ONE place.  None of them changes a result; they select between
// equivalent code paths (A/B measurements, tests that force a rarely taken path).  Every switch is read from its environment
// variable ONCE, when the library first looks at any of them, and can be set afterwards with am_debug_set(name, value) (exported for
// the tests, not part of include/am.h).  A read is one relaxed atomic load: no getenv on any call path.
#pragma once
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace am {
namespace cfg {

enum Key {
    kSfAblate,            // AM_SF_ABLATE: debug instantiation of k_sf (1, 4, 5, 11: parts switched off; 9: per-phase cycle sums)
    kSfPoolBlocks,        // AM_SF_POOL_BLOCKS: size of the record-block pool of the first attempt (tests force overflow + retry)
    kSfWq,                // AM_SF_WQ: walker-queue entries per wavefront (0: none)
    kSfWqIters,           // AM_SF_WQ_ITERS: trie steps a resolve batch takes before it parks
    kSfMaxBloomLog2Words, // AM_SF_MAX_BLOOM_LOG2_WORDS: cap on the LDS filter size (tests: dense filters)
    kSfNoChildren,        // AM_SF_NO_CHILDREN: the flattener gives heavy depth-4 nodes no five-byte child entries (A/B; read when an image is flattened)
    kSfProbeTwo,          // AM_SF_PROBE_TWO: A/B -- automata with few 4-byte-suffix keys also take the instantiation whose probe rounds always look at two candidates per lane
    kNoSmallRun,          // AM_NO_SMALL_RUN: am_run on small batches takes the general path
    kDfa,                 // AM_DFA: 0 = no automaton gets a DFA section / none is used; 1 = every automaton whose table fits gets one (read when an image is flattened); unset: dictionaries with heavy suffix nodes
    kDfaChunk,            // AM_DFA_CHUNK: bytes of the batch one lane of k_dfa owns (read when an image is flattened; default 2048)
    kDfaRarePermille,     // AM_DFA_RARE_PERMILLE: share of the edges (in thousandths; default 1) whose bytes may go without a column of the DFA table (tests: 300 makes most bytes rare)
    kDfaMinKiB,           // AM_DFA_MIN_KIB: batch size from which a dictionary's scans take the table walk by themselves (default 32 768: below, a unit's walk costs more than the filter's whole scan)
    kDfaTune,             // AM_DFA_TUNE: launch parameters of k_dfa for A/B measurements (am_dfa.hip dfa_tune)
    kDfaHotLog2,          // AM_DFA_HOT_LOG2: columns of the DFA section's hot table, as a power of two (read when an image is flattened; default 4 = 16 columns, two rows per 128-byte line)
    kDfaNoChains,         // AM_DFA_NO_CHAINS: every state of the DFA section gets a dense row (A/B against the chain records; read when an image is flattened)
    kFlattenTrace,        // AM_FLATTEN_TRACE: the flattener prints its phases with their wall time on stderr
    kFlattenSerial,       // AM_FLATTEN_SERIAL: the flattener starts no task (tests: the images are the same byte for byte; am_automaton_create flattens IgnoreCase on first use)
    kNoIdsScan,           // AM_NO_IDS_SCAN: containsAll folds the records of a full scan (k_idset) instead of setting the id bits inside k_sf
    kRpFullScans, kRpSplice, kRpPieces, kRpParallelFold, kRpGroups, kRpNoFuse, kRpNoSpin, kRpMatMain, kRpNoRangeReuse, kRpTrace,
    kRpLoopWaves,         // AM_RP_LOOP_WAVES: wavefronts per SIMD k_rp_loop's register budget is cut for (4, 5, 6, 8)
    kRpLds,               // AM_RP_LDS: 0 = the one-kernel loop never keeps a haystack's lists in LDS (k_rp_loop alone, round 4's kernel); unset / 1: k_rp_lds first, k_rp_loop for what it gives up
    kRpNoPli,             // AM_RP_NO_PLI: k_rp_lds keeps its payload column also for replacers whose priorities are minus the payload index (A/B: 4 instead of 5 wavefronts per SIMD)
    kRpLoop,              // AM_RP_LOOP: all passes of a haystack in one kernel (am_rploop.hip): 0 = never, 1 = whenever the replacer allows it; unset: many documents
    kRunSegments,         // AM_RUN_SEGMENTS: am_run on host slices in segments whose records come back while the next segment goes up: 0 = never (the call in one piece), k > 0 = always, in segments of k KiB (tests); unset: from 1 GiB on, 256-MiB segments
    kCount
};

struct Table {
    std::atomic<long> v[kCount];
    std::once_flag once;
};
inline Table& table() { static Table t; return t; }
inline const char* name_of(int k)
{
    static const char* const names[kCount] = {"AM_SF_ABLATE", "AM_SF_POOL_BLOCKS", "AM_SF_WQ", "AM_SF_WQ_ITERS", "AM_SF_MAX_BLOOM_LOG2_WORDS", "AM_SF_NO_CHILDREN", "AM_SF_PROBE_TWO", "AM_NO_SMALL_RUN", "AM_DFA", "AM_DFA_CHUNK", "AM_DFA_RARE_PERMILLE", "AM_DFA_MIN_KIB", "AM_DFA_TUNE", "AM_DFA_HOT_LOG2", "AM_DFA_NO_CHAINS", "AM_FLATTEN_TRACE", "AM_FLATTEN_SERIAL", "AM_NO_IDS_SCAN",
                                              "AM_RP_FULL_SCANS", "AM_RP_SPLICE", "AM_RP_PIECES", "AM_RP_PARALLEL_FOLD", "AM_RP_GROUPS", "AM_RP_NO_FUSE", "AM_RP_NO_SPIN",
                                              "AM_RP_MAT_MAIN", "AM_RP_NO_RANGE_REUSE", "AM_RP_TRACE", "AM_RP_LOOP_WAVES", "AM_RP_LDS", "AM_RP_NO_PLI", "AM_RP_LOOP", "AM_RUN_SEGMENTS"};
    return names[k];
}
constexpr long kUnset = -1;
inline void init()
{
    Table& t = table();
    std::call_once(t.once, [&t] {
        for (int k = 0; k < kCount; k++) {
            const char* e = std::getenv(name_of(k));
            t.v[k].store(e ? (*e ? std::atol(e) : 1L) : kUnset, std::memory_order_relaxed);
        }
    });
}
// the switch's value, or kUnset (-1)
inline long get(Key k) { init(); return table().v[k].load(std::memory_order_relaxed); }
inline bool on(Key k) { const long v = get(k); return v != kUnset && v != 0; }
inline bool set(const char* name, long value)
{
    init();
    for (int k = 0; k < kCount; k++) if (std::strcmp(name, name_of(k)) == 0) { table().v[k].store(value, std::memory_order_relaxed); return true; }
    return false;
}

}  // namespace cfg
}  // namespace am
