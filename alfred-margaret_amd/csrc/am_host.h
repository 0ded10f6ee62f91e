// am_host.h -- what the host-side translation units of libam share (am_abi.cpp: runtime, automata, batches, scans, results; am_replacer.cpp: the
// Replacer; am_contains_all.cpp: containsAll and the fold checksum): error handling, the per-device runtime, device buffers, the handle structs of
// include/am.h and the few scan entry points the Replacer drives.  Internal: nothing here is part of the C ABI.
#pragma once
#include "../../include/am.h"
#include "../../include/am_debug.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "am_config.h"
#include "am_device.h"
#include "am_flatten.h"

namespace am { int abi_fail(int code, const std::string& msg); }      // sets the calling thread's am_last_error message (am_abi.cpp)

namespace am {
namespace host {

using namespace am::dev;

inline int fail(int code, const std::string& msg) { return am::abi_fail(code, msg); }
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(AM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define AM_TRY(expr) do { int rc_ = (expr); if (rc_ != AM_OK) return rc_; } while (0)

// ---- per-device runtime.  libam serves every visible HIP device from one process: a handle (automaton, batch, result,
// replacer) lives on the device that was current when it was made (or that its memory belongs to), every entry point makes
// that device current for the calling thread while it runs, and launches go to a stream that belongs to the CALLING THREAD
// (one library stream per thread and device, or the stream the thread gave with am_set_stream): calls from different
// threads do not serialise on a shared stream or lock.
constexpr int kMaxDev = 16;
struct DeviceInfo { int n_cu = 0; size_t hbm = 0; std::string name; };
struct Runtime {
    std::mutex mu;
    bool probed = false;
    int n_dev = 0;
    std::string why;
    DeviceInfo dev[kMaxDev];
    // profiling (process-wide totals per kernel name)
    std::atomic<bool> prof_on{false};
    struct Pending { std::string k; hipEvent_t a, b; int dev; };
    std::vector<Pending> pending;
    std::map<std::string, std::pair<double, uint64_t>> prof;
};
extern Runtime g_rt;

int ensure_runtime();
int get_stream(int dev, hipStream_t* st);                 // the calling thread's stream on `dev` (its own, or the one it gave with am_set_stream)

// RAII: makes `dev` current for the calling thread while an entry point runs
struct OnDevice {
    int prev = -1; bool switched = false; int rc = AM_OK;
    explicit OnDevice(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) { rc = fail(AM_ERR_HIP, "hipGetDevice failed"); return; }
        if (prev != dev) {
            hipError_t e = hipSetDevice(dev);
            if (e != hipSuccess) { rc = fail(AM_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e)); return; }
            switched = true;
        }
    }
    ~OnDevice() { if (switched) (void)hipSetDevice(prev); }
};
#define ON_DEVICE(dev) OnDevice on_device_guard_(dev); AM_TRY(on_device_guard_.rc)

// RAII HIP-event bracket around one kernel launch (only when profiling is enabled)
struct Prof {
    bool on; hipStream_t st; Runtime::Pending p;
    Prof(const char* k, hipStream_t s) : on(g_rt.prof_on.load(std::memory_order_relaxed)), st(s)
    {
        if (!on) return;
        p.k = k; p.dev = 0;
        (void)hipGetDevice(&p.dev);
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(p.a, st);
    }
    ~Prof()
    {
        if (!on) return;
        (void)hipEventRecord(p.b, st);
        std::lock_guard<std::mutex> lk(g_rt.mu);
        g_rt.pending.push_back(p);
    }
};

struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    int ensure(size_t n)
    {
        if (n <= cap) return AM_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = n + n / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return fail(e == hipErrorOutOfMemory ? AM_ERR_OOM : AM_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
        cap = want;
        return AM_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct Flavor {
    bool ready = false;
    void* d_image = nullptr;
    size_t bytes = 0;
    uint64_t generation = 0;     // unique per uploaded image (an address can be reused): what a batch remembers its route by
    ImageHeader h;
};
uint64_t next_image_generation();

}  // namespace host
}  // namespace am

struct am_automaton {
    int dev = 0;                 // the device its images live on
    std::vector<uint64_t> transitions, root_ascii;
    std::vector<uint32_t> offsets, values_len;
    bool has_ref = false;        // false for handles attached to a received image
    std::shared_ptr<const am::LowerTable> lower;   // the caller's lower-case table (am_automaton_create_ex); null: the built-in one
    std::vector<uint8_t> cs_image;   // CaseSensitive image flattened (= validated) at creation, uploaded on first use
    // the IgnoreCase image is flattened on a thread of its own WHILE am_automaton_create flattens (= validates) the CaseSensitive one: whichever mode the first scan
    // asks for, its image is there or nearly there (the reference's protocol times build + run together, benchmark/haskell/app/Main.hs:62-64,73)
    struct Flat { int rc = 0; std::string err; std::vector<uint8_t> img; };
    std::future<Flat> ic_pending;
    int kernel_pref = 0;
    std::mutex mu;
    am::host::Flavor fl[2];
};

struct am_batch {
    int dev = 0;
    void* d_text = nullptr; uint64_t* d_offsets = nullptr;
    bool owns = false;
    bool hidx_ready = false;     // the per-KiB haystack index depends only on the offsets: built once per batch
    uint64_t total = 0; uint32_t n_hay = 0;
    std::mutex mu;              // guards the workspaces below (calls on one batch serialise)
    am::host::DevBuf text_buf, offs_buf;  // backing store of d_text / d_offsets when the batch owns them
    am::host::DevBuf combo;               // ... or ONE buffer [offsets | text] for small batches that went up with a single copy
    am::host::DevBuf hidx, unit_counts, unit_offsets, scan_tmp, small, hay_counts, flags, unit_first, pool, block_next;
    am::host::DevBuf sparse, dense_counts, dense_offsets, dense_out;      // automata with the empty needle (dense pass)
    // the route a dictionary's image took on this batch the last time it was asked (am_abi.cpp make_plan: a sample walk decides once per batch and image)
    uint64_t route_image = 0; bool route_dfa = false;        // (the image's generation; 0: not asked)
    uint32_t route_ends_per_kib = 0;                         // what the sample walk counted: sizes the token pool's first guess
};

struct am_matches {
    int dev = 0;
    am::dev::Record* d_records = nullptr; uint64_t n = 0; size_t cap_bytes = 0;
    uint64_t first = 0;                                  // the result is records [first, first + n) of the array (am_run_range keeps a sub-range)
    std::vector<am_match> host; bool fetched = false;
    am_match* big = nullptr; size_t big_cap = 0;         // large results: a host block of the library's own -- page-locked (big_pinned: the records are
    bool big_pinned = false;                             // DMA'd straight into it), or pageable and filled through pinned staging
};

namespace am {
namespace host {

// am_abi.cpp: what the Replacer and containsAll drive
int prepare(const am_automaton* ca, int case_mode, const Flavor** out);                      // the automaton's image for a case mode, on its device
int finish_batch(am_batch* b);                                                               // workspaces of a batch whose text and offsets are in place
// sorted records of a batch: sink_final(n, &ptr) names the destination once the count is known
// Searcher.containsAll without records (k_sf's ids mode): every reported needle id into the haystack's row of d_bits (n_hay x words, cleared here),
// flags_out[h] = 1 iff all n_needles ids were seen; *taken = false when the automaton does not go this way (general kernel forced, the empty needle's
// dense pass, nothing to scan) and the caller folds the records instead.
int scan_needle_ids(const am_automaton* a, int case_mode, am_batch* b, const uint64_t* d_vals_off, const uint32_t* d_vals, uint32_t n_needles,
                    uint32_t* d_bits, uint32_t words, uint32_t* d_missing, uint8_t* flags_out, bool* taken);
int run_records(const am_automaton* a, int case_mode, am_batch* b, const std::function<int(uint64_t, Record**)>& sink_final, uint64_t* n_out, bool have_lock = false);
// the same without a host round trip (suffix-filter route, worst-case pool): the count stays on the device
int run_records_async(const am_automaton* a, int case_mode, am_batch* b, Record* d_out, const uint64_t** n_dev, hipStream_t st);
inline size_t padded_text(uint64_t total) { return (size_t)((total + 15) & ~15ull) + 16; }

}  // namespace host
}  // namespace am
