/*
 * am.h -- C ABI of libam.so, the MI355X-native drop-in for the hot path of
 * channable/alfred-margaret's Data.Text.AhoCorasick.Automaton.
 *
 * The reference is a pure Haskell library with no FFI layer of its own on this path; its only
 * precedent for a C boundary is benchmark/rust-ffi (app/Main.hs:28-52: `foreign import ccall
 * unsafe "perform_ac"`, slices passed as {ptr, off, len}, pinned byte arrays, callee borrows).
 * This header keeps that slice convention.  Each entry point names the reference function whose
 * body it replaces (paths relative to the reference repository root); INTEGRATION.md shows the
 * `foreign import ccall` stubs a maintainer would add.
 *
 * Division of labour (unchanged semantics):
 *   Haskell keeps   build (Automaton.hs:176-200)  -> state numbering, value lists, value order
 *                   the fold function f, Done/Step (Automaton.hs:398,522-534)
 *   libam replaces  the body of runWithCase (Automaton.hs:442-534): decode, lower-case,
 *                   transition lookup and the "is there anything to report here" test,
 *                   for a whole batch of haystacks at once on the GPU.
 *
 * Result format: one am_match per (haystack, end position) at which the reference would call
 * the fold function at least once.  `state` is a reference state id such that
 * `machineValues ! state` (Automaton.hs:109) is exactly the list of values the reference folds
 * at that position, in the reference's order.  Records are sorted by (haystack, end_pos), which
 * is the order of the reference's left fold.  So
 *     runWithCase cs seed f machine text
 *  == foldl over records r, then over (machineValues ! r.state), of f acc (Match r.end_pos v),
 *     stopping at the first Done.
 *
 * Conventions: every function returns AM_OK (0) or a negative AM_ERR_* code and never throws or
 * aborts; am_last_error() gives a thread-local message.  Inputs are borrowed for the duration of
 * the call only.  Handles are immutable after creation and may be shared between threads; every
 * calling thread launches on its own HIP stream (am_set_stream replaces it for that thread), so
 * calls from different threads run concurrently; two calls on the SAME am_batch serialise on its
 * workspaces.  There is no CPU execution path: without a usable MI355X every run entry point
 * returns AM_ERR_NO_DEVICE.
 */
#ifndef AM_H
#define AM_H

#include <stddef.h>
#include <stdint.h>

/* libam.so is built with -fvisibility=hidden: the functions below (and include/am_debug.h's, for tests and measurements) are
 * everything it exports. */
#ifndef AM_API
#define AM_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* Data.Text.CaseSensitivity (src/Data/Text/CaseSensitivity.hs:14-22) */
#define AM_CASE_SENSITIVE 0
#define AM_IGNORE_CASE 1

#define AM_OK 0
#define AM_ERR_INVALID (-1)      /* malformed arguments / automaton arrays */
#define AM_ERR_NO_DEVICE (-2)    /* no usable HIP device */
#define AM_ERR_HIP (-3)          /* HIP runtime error, see am_last_error() */
#define AM_ERR_OOM (-4)
#define AM_ERR_UNSUPPORTED (-5)

typedef struct am_automaton am_automaton;
typedef struct am_replacer am_replacer;
typedef struct am_needle_ids am_needle_ids;
typedef struct am_replaced am_replaced;
typedef struct am_batch am_batch;
typedef struct am_matches am_matches;

/* A Text slice: bytes ptr[off .. off+len).  Same shape as U8Slice in
 * benchmark/rust-ffi/app/Main.hs:34-45 (= Data.Text.Internal.Text array/offset/length). */
typedef struct am_slice {
    const uint8_t* ptr;
    size_t off;
    size_t len;
} am_slice;

/* One reportable position.  end_pos = matchPos (Automaton.hs:99-102): code unit index one past the
 * match, relative to the START OF THE SLICE (Automaton.hs:452,530), not to the array. */
typedef struct am_match {
    uint64_t end_pos;
    uint32_t haystack;   /* index into the batch */
    uint32_t state;      /* fold (machineValues ! state) here */
} am_match;

AM_API const char* am_last_error(void);

/* ---- automaton ------------------------------------------------------------------------------
 * am_automaton_create: second half of `build` (Automaton.hs:176-200).  Takes the AcMachine fields
 * exactly as the reference packs them (Automaton.hs:108-123, bit layout :75-94):
 *   transitions   machineTransitions, n_transitions entries
 *   offsets       machineOffsets, n_states + 1 entries (scanl, :170)
 *   root_ascii    machineRootAsciiTransitions, 128 entries
 *   values_len    length (machineValues ! s) for every state (the values themselves stay in Haskell)
 * and flattens them into the device image (LDS filter, cuckoo fingerprint table and Patricia trie of
 * the reversed needles, goto hash of the general path).  The image for each case mode
 * is built and uploaded on first use. */
AM_API int am_automaton_create(const uint64_t* transitions, size_t n_transitions,
                        const uint32_t* offsets, size_t n_states,
                        const uint64_t* root_ascii,
                        const uint32_t* values_len,
                        am_automaton** out);
/* The same with the caller's lower-casing.  The reference lowers non-ASCII code points with GHC base's Data.Char.toLower
 * (src/Data/Text/Utf8.hs:145-151 lowerCodePoint, :138-140 lowerUtf8; inverse src/Data/Text/Utf8/Unlower.hs:26-40), whose table
 * follows the Unicode version of the compiler that builds it (alfred-margaret.cabal:69 `base >= 4.7 && < 5`; 9.6-9.10 = Unicode 15,
 * 9.12+ = 16).  So that IgnoreCase is bit-exact against ANY such build, the table crosses the boundary as data:
 *   lower_from[i] -> lower_to[i], n_lower_pairs pairs = [(c, toLower c) | c <- [minBound ..], toLower c /= c]
 * in any order (pairs with lower_from < 128 are ignored: ASCII is toLowerAscii, :131-135; identity pairs are ignored; one code point
 * with two different images is AM_ERR_INVALID).  NULL, NULL, 0 = the built-in Unicode 14.0 table (am_unicode_version()).
 * The IgnoreCase image bakes the table in (byte edges of every x with toLower x == c; the general kernel's delta table) and
 * records a hash of it: am_automaton_lower_hash(a) == am_lower_table_hash(pairs). */
AM_API int am_automaton_create_ex(const uint64_t* transitions, size_t n_transitions,
                           const uint32_t* offsets, size_t n_states,
                           const uint64_t* root_ascii,
                           const uint32_t* values_len,
                           const uint32_t* lower_from, const uint32_t* lower_to, size_t n_lower_pairs,
                           am_automaton** out);
AM_API uint32_t am_automaton_lower_hash(const am_automaton* a);
AM_API uint32_t am_lower_table_hash(const uint32_t* lower_from, const uint32_t* lower_to, size_t n_pairs);   /* NULL: the built-in table's; 0 on error */
AM_API void am_automaton_destroy(am_automaton* a);
/* Route k: 0 = automatic (the suffix-filter kernel; the table-walk kernel k_dfa for automata whose image carries a DFA section -- dictionaries with
 * heavy suffix nodes -- on batches of 32 MiB and more, from 64 MiB on only where a sample walk finds a needle end every few bytes), 1 = force the general AC kernel (test infrastructure: AM_ERR_UNSUPPORTED unless libam_check.so
 * is loaded), 2 = force the suffix-filter kernel, 3 = force the table-walk kernel (AM_ERR_UNSUPPORTED at run time when the image has no DFA section).
 * All routes report the same matches. */
AM_API int am_automaton_set_kernel(am_automaton* a, int k);

/* ---- one-shot entry points on host slices (what the Haskell shim binds) ----------------------
 * am_count:        runWithCase cs 0 (\n _ -> Step (n+1))  per haystack
 *                  (benchmark/haskell/app/Main.hs:67-76 countMatches; counts VALUES, i.e. fold calls)
 * am_contains_any: Searcher.containsAny (Searcher.hs:156-164) per haystack
 * am_run:          runWithCase / runText / runLower (Automaton.hs:442-553): all records.  A batch of 1 GiB and more against an automaton that may meet
 *                  match-dense text (a dictionary, a small automaton) goes up in segments of whole haystacks, and a segment's records travel to the host
 *                  while the next one is uploaded and scanned; such a result lives on the host only (am_matches_data is immediate,
 *                  am_matches_device_data is NULL, am_matches_fold_hash is not available).  Use am_batch_upload + am_run_batch to keep records in HBM. */
AM_API int am_count(const am_automaton* a, int case_mode, const am_slice* hay, size_t n_hay, uint64_t* counts_out);
AM_API int am_contains_any(const am_automaton* a, int case_mode, const am_slice* hay, size_t n_hay, uint8_t* flags_out);
AM_API int am_run(const am_automaton* a, int case_mode, const am_slice* hay, size_t n_hay, am_matches** out);

/* ---- ONE haystack in ranges (the reference folds one Text of any size, Automaton.hs:468-480) --
 * am_run_range:   the records of runWithCase on `hay` whose end position lies in (lo, hi], 0 <= lo <= hi <= hay->len; end_pos relative to the
 *                 whole slice, haystack = 0.  Only a window of the text is uploaded and scanned: the range plus, before it, 4 bytes per code
 *                 point of the longest needle (whether a needle ends at a position depends only on one maximal match before it), both ends on
 *                 code point boundaries.  Ranges that partition (0, len] give, concatenated, exactly the records of am_run on the whole
 *                 haystack: that is how am_multi_run_single spreads one document over several GPUs, and how a caller scans a document
 *                 larger than device memory.
 * am_count_range: countMatches (benchmark/haskell/app/Main.hs:67-76) over the same positions. */
AM_API int am_run_range(const am_automaton* a, int case_mode, const am_slice* hay, uint64_t lo, uint64_t hi, am_matches** out);
AM_API int am_count_range(const am_automaton* a, int case_mode, const am_slice* hay, uint64_t lo, uint64_t hi, uint64_t* count_out);

/* ---- device-resident batches (bulk callers; what bench.py times) ----------------------------- */
AM_API int am_batch_upload(const am_slice* hay, size_t n_hay, am_batch** out);
/* Borrow a batch that already lives in HBM: d_bytes = concatenated haystacks (16-byte aligned,
 * readable up to round_up(total, 16) bytes), d_offsets = n_hay + 1 ascending uint64 byte offsets
 * with d_offsets[0] == 0 and d_offsets[n_hay] == total_bytes.  The offsets must not change while the batch
 * exists (the text may). */
AM_API int am_batch_from_device(const void* d_bytes, const void* d_offsets, size_t n_hay, uint64_t total_bytes, am_batch** out);
AM_API void am_batch_destroy(am_batch* b);
AM_API uint64_t am_batch_total_bytes(const am_batch* b);

AM_API int am_count_batch(const am_automaton* a, int case_mode, const am_batch* b, uint64_t* counts_out /* n_hay, nullable */, uint64_t* total_out /* nullable */);
AM_API int am_contains_any_batch(const am_automaton* a, int case_mode, const am_batch* b, uint8_t* flags_out);
AM_API int am_run_batch(const am_automaton* a, int case_mode, const am_batch* b, am_matches** out);

/* ---- results ----------------------------------------------------------------------------------
 * Records are produced in HBM; am_matches_data copies them to the host on first use. */
AM_API uint64_t am_matches_size(const am_matches* m);
AM_API const am_match* am_matches_data(am_matches* m);           /* host pointer, owned by m; NULL on error */
AM_API const void* am_matches_device_data(const am_matches* m);  /* am_match[size] in HBM, owned by m; NULL for a result assembled on the host (am_run on a large host batch) */
/* One haystack's records of a large result, without copying all of it: the records are sorted by (haystack, end_pos), so they are the contiguous run
 * [*first_out, *first_out + *count_out) (count 0: no match in that haystack); am_matches_copy brings records [first, first + count) to caller memory.
 * What a lazy fold over one document of a big batch consumes (Automaton.hs:522-534 folds a haystack's matches in order). */
AM_API int am_matches_haystack_range(const am_matches* m, uint32_t haystack, uint64_t* first_out, uint64_t* count_out);
AM_API int am_matches_copy(const am_matches* m, uint64_t first, uint64_t count, am_match* out);
AM_API void am_matches_free(am_matches* m);

/* ---- Searcher.containsAll (src/Data/Text/AhoCorasick/Searcher.hs:167-187) ------------------------
 * For a `Searcher Int` made by buildNeedleIdSearcher (:167-169): machineValues in flat form (the list
 * of state s is values[values_offsets[s] .. values_offsets[s+1]), needle ids 0 .. n_needles-1).  The
 * IntSet fold (:175-183) becomes one bitmap row per haystack in HBM; flags_out[i] = 1 iff every id was
 * reported in haystack i (IS.null of the final set; all ones when n_needles == 0).  `a` must outlive ids. */
AM_API int am_needle_ids_create(const am_automaton* a, const uint64_t* values_offsets, const uint32_t* values, uint32_t n_needles, am_needle_ids** out);
AM_API void am_needle_ids_destroy(am_needle_ids* ids);
AM_API int am_contains_all(const am_needle_ids* ids, int case_mode, const am_slice* hay, size_t n_hay, uint8_t* flags_out);
AM_API int am_contains_all_batch(const am_needle_ids* ids, int case_mode, const am_batch* b, uint8_t* flags_out);

/* Checksum of the fold sequence of a result (harness aid; SURVEY 8d "parity check at scale",
 * benchmark/benchmark.py:65-69 asserts count identity on every run).  For every haystack i < n_hay:
 *   hash_out[i]  = foldl (\h (pos, v) -> h * 0x100000001B3 + mix pos v) 0  over the matches the reference's
 *                  runWithCase would hand to its fold function for haystack i, in its order, where
 *                  mix pos v = let x0 = (pos * 0x9E3779B97F4A7C15) xor (v + 0x632BE59BD9B4E019); x1 = x0 xor (x0 >> 32);
 *                                  x2 = x1 * 0xD6E8FEB86659FD93 in x2 xor (x2 >> 32)          (all mod 2^64)
 *   count_out[i] = the number of those matches (= countMatches, benchmark/haskell/app/Main.hs:67-76); nullable.
 * `values` carries machineValues in flat form (am_needle_ids_create; any uint32 payload handles, n_needles unused).
 * Computed on the device from the records in HBM; two runs agree on hash and count iff they fold the same
 * (matchPos, value) sequences (up to 64-bit collisions). */
AM_API int am_matches_fold_hash(const am_matches* m, const am_needle_ids* values, size_t n_hay, uint64_t* hash_out, uint64_t* count_out);

/* ---- Replacer: all passes of Replacer.run on the device -----------------------------------------
 * Replaces the loop `runWithLimit.go` (src/Data/Text/AhoCorasick/Replacer.hs:219-242) for a batch of
 * haystacks: per pass one scan (:223-225), prependMatch/makeMatch (:252-274), the replacementLength
 * check (:183-187, :240), sort + removeOverlap (:191-198, :241) and replace (:163-180) all run in HBM;
 * only finished haystacks travel back.  The caller hands over machineValues of the Replacer's
 * automaton (`Searcher Payload`, :72-77) in flat form:
 *   values_offsets  n_states + 1 entries; the list of state s (in the reference's order) is
 *                   payloads[values[values_offsets[s] .. values_offsets[s+1])]
 *   payloads        Payload (:59-70) per needle: priority, lengths of the ORIGINAL needle in bytes
 *                   and code points (:112-113), replacement = repl_bytes[repl_off .. +repl_len)
 *   min_priority    1 - numNeedles (:217)
 * Priorities must be distinct and <= 0, as build (:100-104) and compose (:127-131) make them.
 * `a` must outlive the replacer.  case_mode is the Replacer's replacerCaseSensitivity.
 * Batches of many documents (>= 64, <= 1 MiB on average, no document with more than 4096 matches) run ALL passes of a
 * haystack inside one kernel (one wavefront per haystack, csrc/am_rploop.hip); everything else goes pass by pass
 * (csrc/am_replace.hip).  The results are the same texts either way; am_replaced_passes reports the passes of the
 * haystack that needed the most. */
typedef struct am_payload {
    int64_t priority;
    uint32_t len_bytes;
    uint32_t len_code_points;
    uint64_t repl_off;
    uint32_t repl_len;
    uint32_t reserved;
} am_payload;
AM_API int am_replacer_create(const am_automaton* a, int case_mode,
                       const uint64_t* values_offsets, const uint32_t* values,
                       const am_payload* payloads, size_t n_payloads,
                       const uint8_t* repl_bytes, size_t n_repl_bytes,
                       int64_t min_priority, am_replacer** out);
AM_API void am_replacer_destroy(am_replacer* r);
/* max_length: runWithLimit's maxLength (:203); UINT64_MAX = `run` (:200-201, maxBound). */
AM_API int am_replacer_run(const am_replacer* r, const am_slice* hay, size_t n_hay, uint64_t max_length, am_replaced** out);
AM_API int am_replacer_run_batch(const am_replacer* r, const am_batch* b, uint64_t max_length, am_replaced** out);   /* b is not modified */
/* The same, but the rewritten texts STAY IN DEVICE MEMORY (the batch's device), for callers that feed them to the next device
   stage: am_replaced_get then returns device pointers, am_replaced_read copies one text to the host.  This is what a
   device-resident pipeline measures; am_replacer_run_batch additionally moves every result over PCIe into pinned host memory. */
AM_API int am_replacer_run_batch_device(const am_replacer* r, const am_batch* b, uint64_t max_length, am_replaced** out);
/* One pass only, for callers that keep sort / removeOverlap / replace (Replacer.hs:159-198) on their side: the fold
 * `prependMatch` (:252-260) with seed (minBound, []) and the given threshold per haystack.  best_out[i] = the best
 * priority below thresholds[i] among the matches of haystack i (INT64_MIN: none); *matches_out = every match that
 * carries it, as makeMatch (:264-274) builds them (start and length in code units of the haystack), in
 * (haystack, start) order = the order `sort` (:241) gives them.  Free with am_prio_matches_free. */
typedef struct am_prio_match {
    uint64_t start;
    uint64_t len;
    uint32_t haystack;
    uint32_t payload;    /* index into the payload table given to am_replacer_create */
} am_prio_match;
AM_API int am_run_priority(const am_replacer* r, const am_slice* hay, size_t n_hay, const int64_t* thresholds,
                    int64_t* best_out, am_prio_match** matches_out, size_t* n_matches_out);
AM_API void am_prio_matches_free(am_prio_match* m);
AM_API uint64_t am_replaced_size(const am_replaced* r);
/* Returns 1 and the text for `Just`, 0 for `Nothing` (longer than max_length), < 0 on error.  *ptr is owned by r. */
AM_API int am_replaced_get(const am_replaced* r, size_t i, const uint8_t** ptr, size_t* len);
AM_API int am_replaced_device(const am_replaced* r);               /* -1: the texts are in host memory; otherwise the device that holds them */
/* Copies text i into dst (host memory, cap bytes) wherever the result lives; *len = its length.  Returns like am_replaced_get. */
AM_API int am_replaced_read(const am_replaced* r, size_t i, uint8_t* dst, size_t cap, size_t* len);
AM_API uint64_t am_replaced_passes(const am_replaced* r);          /* scans that were needed (max over the batch) */
AM_API uint64_t am_replaced_scanned_bytes(const am_replaced* r);   /* haystack bytes scanned over all passes (after the first pass only windows around the replacements) */
AM_API uint64_t am_replaced_spliced_bytes(const am_replaced* r);   /* bytes of rewritten text produced over all passes */
AM_API void am_replaced_free(am_replaced* r);

/* ---- several GPUs ---------------------------------------------------------------------
 * Every handle lives on one device: an automaton / batch on the device that was current when it was made (or that its
 * memory belongs to: am_batch_from_device, am_automaton_from_image), results and replacers with their automaton.
 * Entry points make that device current for the calling thread while they run, so one process can drive all GPUs.
 *
 * am_multi spans the devices with RCCL.  The path shards trivially (independent haystacks, read-only automaton), so
 * the only traffic between devices is one broadcast of the flattened automaton over xGMI and an all-reduce of match
 * counts; match lists are concatenated on the host in haystack order.
 *   am_multi_create        one process drives devices 0..n-1 (ncclCommInitAll); n_devices = 0: all visible devices
 *   am_multi_create_rank   one process per GPU (ncclCommInitRank on the current device); the id comes from
 *                          am_multi_unique_id on one rank and reaches the others through the launcher
 * Reference shape of a foreign binding: benchmark/rust-ffi/app/Main.hs:28-45 (`foreign import ccall`, slices). */
typedef struct am_multi am_multi;
#define AM_UNIQUE_ID_BYTES 128
AM_API int am_multi_unique_id(uint8_t id_out[AM_UNIQUE_ID_BYTES]);
AM_API int am_multi_create(int n_devices, am_multi** out);
AM_API int am_multi_create_rank(int n_ranks, int rank, const uint8_t id[AM_UNIQUE_ID_BYTES], am_multi** out);
AM_API void am_multi_destroy(am_multi* m);
AM_API int am_multi_local_devices(const am_multi* m);     /* devices this process drives */
AM_API int am_multi_world_size(const am_multi* m);        /* devices in all */
AM_API int am_multi_device(const am_multi* m, int i);     /* HIP device id of local device i */
/* ncclBroadcast of the flattened image of `a` (held by global rank `root`; NULL in processes that do not hold the
 * root) to every device; autos_out[i] = a handle on local device i attached to its copy (am_automaton_destroy each). */
AM_API int am_multi_broadcast_automaton(am_multi* m, const am_automaton* a, int case_mode, int root, am_automaton** autos_out);
/* ncclAllReduce(sum) of `count` (<= 512) uint64 per device: values = local_devices x count, row i belongs to local
 * device i; every row holds the sums afterwards. */
AM_API int am_multi_allreduce_sum(am_multi* m, uint64_t* values, size_t count);
/* This process's haystacks cut into contiguous blocks, one per local device (block i = haystacks [n*i/D, n*(i+1)/D)),
 * scanned concurrently; counts_out (nullable) per haystack in order; *total_out = sum over ALL devices (all-reduce):
 * countMatches (benchmark/haskell/app/Main.hs:67-76) of the whole job. */
AM_API int am_multi_count(am_multi* m, am_automaton* const* autos, int case_mode, const am_slice* hay, size_t n_hay, uint64_t* counts_out, uint64_t* total_out);
/* runWithCase on every local device's block; the records of all blocks concatenated on the host in haystack order
 * (haystack = index into `hay`).  Free with am_multi_matches_free. */
AM_API int am_multi_run(am_multi* m, am_automaton* const* autos, int case_mode, const am_slice* hay, size_t n_hay, am_match** matches_out, size_t* n_out);
AM_API void am_multi_matches_free(am_match* p);
/* The same on DEVICE-RESIDENT batches: batches[i] lives on local device i (made there by am_batch_upload or
 * am_batch_from_device; NULL = no work for that device); one host thread and stream per device; nothing but the counts
 * leaves the devices.  BASELINE configs[3] (100 GiB of haystacks spread over the HBM of 8 GPUs) from a C / Haskell host.
 *   am_multi_count_batch  counts_out (nullable; then one nullable array of batch-i-many counts per local device);
 *                         local_totals_out (nullable): one sum per local device; *total_out: all devices (all-reduce)
 *   am_multi_run_batch    results_out[i] = the records of batch i, left in device i's HBM (am_matches_free each);
 *                         *total_records_out = records on all devices (all-reduce)
 * Error behaviour of every am_multi_* collective: a failure on one device is carried into the collective as a flag, so
 * every rank returns (none blocks) and every rank returns an error. */
/* am_batch_upload onto local device i of m (a host without HIP of its own has no other way to choose the device). */
AM_API int am_multi_batch_upload(const am_multi* m, int local_device, const am_slice* hay, size_t n_hay, am_batch** out);
AM_API int am_multi_count_batch(am_multi* m, am_automaton* const* autos, int case_mode, am_batch* const* batches, uint64_t* const* counts_out,
                         uint64_t* local_totals_out, uint64_t* total_out);
AM_API int am_multi_run_batch(am_multi* m, am_automaton* const* autos, int case_mode, am_batch* const* batches, am_matches** results_out, uint64_t* total_records_out);

/* ONE haystack on all devices ("a single huge haystack splits into G ranges with maxNeedleCodePoints overlap"; BASELINE configs[1]
 * shape (i), 1 x 1 GiB): global device g of W owns the end positions in (len * g / W, len * (g + 1) / W] (am_run_range on its own device).
 * Every process passes the WHOLE haystack; each device uploads only its window of it.
 *   am_multi_count_single  local_counts_out (nullable): one count per local device; *total_out: the whole haystack (all-reduce)
 *   am_multi_run_single    *matches_out = the records of this process's ranges in position order (haystack = 0, end_pos relative to the whole
 *                          haystack; am_multi_matches_free); with one process per GPU the launcher concatenates the ranks' arrays in rank
 *                          order.  *total_records_out (nullable) = records on all devices. */
AM_API int am_multi_count_single(am_multi* m, am_automaton* const* autos, int case_mode, const am_slice* hay, uint64_t* local_counts_out, uint64_t* total_out);
AM_API int am_multi_run_single(am_multi* m, am_automaton* const* autos, int case_mode, const am_slice* hay, am_match** matches_out, size_t* n_out, uint64_t* total_records_out);

/* ---- multi-GPU: move the flattened automaton between devices -----------------------------------
 * The image is one position-independent blob, so rank 0 flattens once and the blob is broadcast
 * over xGMI (RCCL broadcast of a byte tensor); every other rank attaches to its received copy. */
AM_API int am_automaton_image_size(const am_automaton* a, int case_mode, size_t* nbytes);
AM_API int am_automaton_image_copy(const am_automaton* a, int case_mode, void* d_dst, size_t nbytes);   /* device -> device */
AM_API int am_automaton_from_image(const void* d_image, size_t nbytes, am_automaton** out);            /* copies the blob; TRUSTED input: a blob made by am_automaton_image_copy in this job (header checked only) */
/* Serialised automaton: the same blob in host memory (write it to a file as is).  Loading checks the
 * header (magic, version, every section inside the blob), a checksum of the body and every index the kernels
 * follow (states, node and edge ids, table slots), so a truncated, damaged or stale file is refused with
 * AM_ERR_INVALID; it skips build + flatten entirely (the reference's JSON instances store only
 * the needles and rebuild, Searcher.hs:68-77).  A handle made from an image serves the image's case mode. */
AM_API int am_automaton_image_read(const am_automaton* a, int case_mode, void* host_dst, size_t nbytes);   /* device -> host */
AM_API int am_automaton_from_host_image(const void* image, size_t nbytes, am_automaton** out);

/* ---- UTF-8 helpers on the path -------------------------------------------------------------------
 * am_lower_code_point: Utf8.lowerCodePoint (src/Data/Text/Utf8.hs:145-151) with the BUILT-IN table: simple mapping of
 * Unicode 14.0 (am_unicode_version() = 0x0E00: major << 8 | minor).  The reference uses GHC base's Data.Char.toLower, whose
 * Unicode version follows the compiler: a caller that needs exactly its own hands its table to am_automaton_create_ex; a
 * flattened image carries the table it was baked with and a hash of it (am_automaton_lower_hash).
 * am_unlower_code_point: Utf8.unlowerCodePoint (src/Data/Text/Utf8/Unlower.hs:26-28) as an ascending set, built-in table;
 * returns the set size (may exceed cap).
 * am_image_version: layout version of the flattened image (am_automaton_image_*); images of another version are refused. */
AM_API uint32_t am_lower_code_point(uint32_t cp);
AM_API uint32_t am_unicode_version(void);
AM_API uint32_t am_image_version(void);
AM_API size_t am_unlower_code_point(uint32_t cp, uint32_t* out, size_t cap);

/* ---- runtime knobs ------------------------------------------------------------------------------ */
AM_API int am_set_stream(void* hip_stream);   /* hipStream_t for all subsequent launches OF THE CALLING THREAD; NULL = the thread's library stream */
AM_API int am_get_stream(void** hip_stream);  /* the stream the calling thread's launches on the current device go to (to order other work after them) */
AM_API int am_device_info(int* n_cu, size_t* hbm_bytes, char* name, size_t name_cap);
/* Host blocks of large results are the library's own (am_matches_data); one freed block -- page-locked, up to 1 GiB -- is kept for the next
 * large result, and so is one pageable block of a result beyond that (up to 8 GiB: its pages are there, the next result of that size is
 * copied into it at the speed of the wire).  This gives both back to the system now (page-locked memory is a shared, limited resource). */
AM_API int am_release_host_memory(void);
/* Device memory the library keeps between calls: up to four device arrays of freed results per device (16 GiB in all) and, per calling thread and device, the
 * one-shot batch of am_run / am_count / am_contains_any (text + workspaces, up to a sixteenth of the device's memory) -- freeing VRAM is not free (the driver wipes it
 * on the engines the host copies use), so arrays that will be wanted again are kept.  This frees the kept result arrays of every device and the CALLING thread's
 * one-shot batches now. */
AM_API int am_release_device_memory(void);
/* Per-kernel timing with HIP events on the launch stream (off by default). */
AM_API int am_profile_enable(int on);
AM_API int am_profile_reset(void);
AM_API int am_profile_read(const char* kernel /* "sf" | "ac" | "hidx" | "scan" | "permute" | "rp_pass" | "rp_splice" | ... */, double* total_ms, uint64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* AM_H */
