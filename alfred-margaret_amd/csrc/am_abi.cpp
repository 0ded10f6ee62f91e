// am_abi.cpp -- the C ABI of include/am.h: handles, device memory, launch orchestration.
// There is deliberately no CPU execution path here: every run entry point needs a HIP device.
#include "../../include/am.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "am_config.h"
#include "am_device.h"
#include "am_flatten.h"

using namespace am;
using namespace am::dev;

static_assert(sizeof(am_match) == sizeof(Record), "am_match must mirror the device record");
static_assert(offsetof(am_match, end_pos) == offsetof(Record, end_pos) && offsetof(am_match, haystack) == offsetof(Record, haystack) &&
                  offsetof(am_match, state) == offsetof(Record, state), "am_match layout");

// ------------------------------------------------------------------ errors, runtime

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
namespace am { int abi_fail(int code, const std::string& msg) { return fail(code, msg); } }      // for am_multi.cpp (same thread-local message)
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(AM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define AM_TRY(expr) do { int rc_ = (expr); if (rc_ != AM_OK) return rc_; } while (0)

namespace {

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
Runtime g_rt;

int ensure_runtime()
{
    std::lock_guard<std::mutex> lk(g_rt.mu);
    if (!g_rt.probed) {
        g_rt.probed = true;
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) {
            g_rt.why = std::string("no HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") + "); libam has no CPU path";
        } else {
            if (n > kMaxDev) n = kMaxDev;
            for (int d = 0; d < n; d++) {
                hipDeviceProp_t p;
                if (hipGetDeviceProperties(&p, d) != hipSuccess) { g_rt.why = "hipGetDeviceProperties failed"; n = d; break; }
                g_rt.dev[d].n_cu = p.multiProcessorCount; g_rt.dev[d].hbm = p.totalGlobalMem; g_rt.dev[d].name = p.gcnArchName;
            }
            g_rt.n_dev = n;
        }
    }
    if (g_rt.n_dev <= 0) return fail(AM_ERR_NO_DEVICE, g_rt.why);
    return AM_OK;
}

// the device that is current for the calling thread
int current_device(int* dev)
{
    AM_TRY(ensure_runtime());
    int d = 0;
    HIP_TRY(hipGetDevice(&d));
    if (d < 0 || d >= g_rt.n_dev) return fail(AM_ERR_UNSUPPORTED, "current HIP device is beyond the devices libam serves");
    *dev = d;
    return AM_OK;
}
int ensure_device() { int d; return current_device(&d); }

// the device a device pointer belongs to (falls back to the current device for pointers HIP does not know)
int device_of_pointer(const void* p, int* dev)
{
    AM_TRY(current_device(dev));
    hipPointerAttribute_t at;
    if (p && hipPointerGetAttributes(&at, p) == hipSuccess) { if (at.device >= 0 && at.device < g_rt.n_dev) *dev = at.device; }
    else (void)hipGetLastError();
    return AM_OK;
}

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

struct am_batch_fwd;
// per calling thread: its library streams (one per device, made on first use), its stream override, its one-shot batches
struct ThreadState {
    hipStream_t own[kMaxDev] = {};
    hipStream_t user = nullptr; bool use_user = false;
    am_batch* oneshot[kMaxDev] = {};
    // one-shot calls on small inputs: pinned host memory of the calling thread, so that offsets + text go up with ONE asynchronous copy on
    // the thread's stream and the results come back the same way -- a call costs one stream synchronisation, not four blocking copies
    uint8_t* pin = nullptr; size_t pin_cap = 0;          // upload staging (two halves that take turns for inputs above kPinPiece)
    uint8_t* pin_res = nullptr; size_t pin_res_cap = 0;  // results
    hipEvent_t pin_ev[2] = {nullptr, nullptr}; int pin_ev_dev = -1;      // (events belong to the device they were made on)
    bool pins_adopted = false;                           // looked for an ended thread's pinned buffers and events already
    ~ThreadState();
};
thread_local ThreadState tl_state;
hipStream_t adopt_stream(int dev);
am_batch* adopt_batch(int dev);

int get_stream(int dev, hipStream_t* st)
{
    if (tl_state.use_user) { *st = tl_state.user; return AM_OK; }
    if (!tl_state.own[dev]) {
        tl_state.own[dev] = adopt_stream(dev);                                                               // one left by a thread that ended
        if (!tl_state.own[dev]) HIP_TRY(hipStreamCreateWithFlags(&tl_state.own[dev], hipStreamNonBlocking));     // the device is current (ON_DEVICE)
    }
    *st = tl_state.own[dev];
    return AM_OK;
}

constexpr size_t kSmallUpload = 4u << 20;       // one-shot batches up to this size take the pinned single-copy path
constexpr size_t kPinPiece = 256u << 10;        // above this the gather into pinned memory and the DMA of the previous piece overlap

void adopt_pins();
int pin_events(int dev);
std::atomic<size_t> g_pinned_staging_bytes{0};           // page-locked staging memory of all threads, living or parked (am_debug_pinned_bytes: the leak test)
int pin_ensure(uint8_t*& p, size_t& cap, size_t need)
{
    if (!tl_state.pins_adopted) adopt_pins();             // the page-locked buffers of a thread that has ended, before any new ones are made
    if (need <= cap) return AM_OK;
    if (p) { (void)hipHostFree(p); g_pinned_staging_bytes.fetch_sub(cap, std::memory_order_relaxed); p = nullptr; cap = 0; }
    const size_t want = need + need / 4 + 4096;
    if (hipHostMalloc((void**)&p, want, hipHostMallocPortable) != hipSuccess) { p = nullptr; (void)hipGetLastError(); return fail(AM_ERR_OOM, "hipHostMalloc(pinned staging) failed"); }
    cap = want;
    g_pinned_staging_bytes.fetch_add(want, std::memory_order_relaxed);
    return AM_OK;
}

// Results of a call, device -> caller: small ones travel through the thread's pinned result buffer (an asynchronous copy into pageable
// memory is a blocking, staged copy inside the runtime), the caller's buffers are filled after the call's ONE stream synchronisation.
struct ResultCopies {
    struct Item { void* dst; size_t off, n; };
    Item items[4]; int n_items = 0; size_t used = 0;
    int add(void* dst, const void* d_src, size_t n, hipStream_t st)
    {
        if (n == 0) return AM_OK;
        if (n > (64u << 10) || n_items == 4) { HIP_TRY(hipMemcpyAsync(dst, d_src, n, hipMemcpyDeviceToHost, st)); return AM_OK; }
        const size_t off = (used + 15) & ~(size_t)15;
        if (off + n > tl_state.pin_res_cap) {
            if (n_items) { HIP_TRY(hipMemcpyAsync(dst, d_src, n, hipMemcpyDeviceToHost, st)); return AM_OK; }      // the buffer is in use by this call: do not move it
            AM_TRY(pin_ensure(tl_state.pin_res, tl_state.pin_res_cap, (size_t)256 << 10));
        }
        HIP_TRY(hipMemcpyAsync(tl_state.pin_res + off, d_src, n, hipMemcpyDeviceToHost, st));
        items[n_items++] = Item{dst, off, n};
        used = off + n;
        return AM_OK;
    }
    int finish(hipStream_t st)
    {
        HIP_TRY(hipStreamSynchronize(st));
        for (int i = 0; i < n_items; i++) std::memcpy(items[i].dst, tl_state.pin_res + items[i].off, items[i].n);
        return AM_OK;
    }
};

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
    ImageHeader h;
};

}  // namespace

struct am_automaton {
    int dev = 0;                 // the device its images live on
    std::vector<uint64_t> transitions, root_ascii;
    std::vector<uint32_t> offsets, values_len;
    bool has_ref = false;        // false for handles attached to a received image
    std::shared_ptr<const LowerTable> lower;   // the caller's lower-case table (am_automaton_create_ex); null: the built-in one
    std::vector<uint8_t> cs_image;   // CaseSensitive image flattened (= validated) at creation, uploaded on first use
    int kernel_pref = 0;
    std::mutex mu;
    Flavor fl[2];
};

struct am_batch {
    int dev = 0;
    void* d_text = nullptr; uint64_t* d_offsets = nullptr;
    bool owns = false;
    bool hidx_ready = false;     // the per-KiB haystack index depends only on the offsets: built once per batch
    uint64_t total = 0; uint32_t n_hay = 0;
    std::mutex mu;              // guards the workspaces below (calls on one batch serialise)
    DevBuf text_buf, offs_buf;  // backing store of d_text / d_offsets when the batch owns them
    DevBuf combo;               // ... or ONE buffer [offsets | text] for small batches that went up with a single copy
    DevBuf hidx, unit_counts, unit_offsets, scan_tmp, small, hay_counts, flags, unit_first, pool, block_next;
    DevBuf sparse, dense_counts, dense_offsets, dense_out;      // automata with the empty needle (dense pass)
};

struct am_matches {
    int dev = 0;
    Record* d_records = nullptr; uint64_t n = 0; size_t cap_bytes = 0;
    uint64_t first = 0;                                  // the result is records [first, first + n) of the array (am_run_range keeps a sub-range)
    std::vector<am_match> host; bool fetched = false;
    am_match* big = nullptr; size_t big_cap = 0;         // large results: a host block of the library's own -- page-locked (big_pinned: the records are
    bool big_pinned = false;                             // DMA'd straight into it), or pageable and filled through pinned staging
};

// The record array of the last freed result is kept for the next call (one buffer, reused when it is large enough
// and not more than twice what is needed): a caller that scans batch after batch does not pay hipMalloc/hipFree of
// hundreds of megabytes per call.
namespace {
struct RecordCache {
    std::mutex mu; void* p = nullptr; size_t cap = 0;
    void* take(size_t need, size_t* cap_out)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (p && cap >= need && cap <= 2 * need + (1u << 20)) { void* r = p; *cap_out = cap; p = nullptr; cap = 0; return r; }
        return nullptr;
    }
    void give(void* q, size_t c)
    {
        void* old = nullptr;
        { std::lock_guard<std::mutex> lk(mu); old = p; p = q; cap = c; }
        if (old) (void)hipFree(old);
    }
};
RecordCache g_record_cache[kMaxDev];

// Streams and one-shot batches of threads that have ended wait here for the next new thread: a thread-exit destructor must
// not call into HIP (it may run while the runtime is being torn down at process exit), and a server that starts and ends
// many threads must not leak a stream and a batch per thread.
struct Orphans {
    std::mutex mu;
    std::vector<hipStream_t> streams[kMaxDev];
    std::vector<am_batch*> batches[kMaxDev];
    // page-locked staging of ended threads (am_multi_* and the Replacer's group threads start fresh threads per call: without this every
    // call left ~0.3 .. 20 MiB of page-locked host memory behind per device -- ADVICE r3)
    struct Pins { uint8_t* pin; size_t pin_cap; uint8_t* pin_res; size_t pin_res_cap; };
    std::vector<Pins> pins;
    std::vector<hipEvent_t> events[kMaxDev];
};
Orphans& orphans() { static Orphans* o = new Orphans(); return *o; }      // never destroyed: no static-destruction order to worry about
hipStream_t adopt_stream(int dev)
{
    Orphans& o = orphans();
    std::lock_guard<std::mutex> lk(o.mu);
    if (o.streams[dev].empty()) return nullptr;
    hipStream_t s = o.streams[dev].back(); o.streams[dev].pop_back();
    return s;
}
am_batch* adopt_batch(int dev)
{
    Orphans& o = orphans();
    std::lock_guard<std::mutex> lk(o.mu);
    if (o.batches[dev].empty()) return nullptr;
    am_batch* b = o.batches[dev].back(); o.batches[dev].pop_back();
    return b;
}

void adopt_pins()
{
    tl_state.pins_adopted = true;
    if (tl_state.pin || tl_state.pin_res) return;
    Orphans& o = orphans();
    std::lock_guard<std::mutex> lk(o.mu);
    if (o.pins.empty()) return;
    const Orphans::Pins p = o.pins.back(); o.pins.pop_back();
    tl_state.pin = p.pin; tl_state.pin_cap = p.pin_cap; tl_state.pin_res = p.pin_res; tl_state.pin_res_cap = p.pin_res_cap;
}

// the calling thread's two staging events, on device `dev` (events of an ended thread are taken over; a thread that moves to another device
// leaves its old ones for that device's next user)
int pin_events(int dev)
{
    if (tl_state.pin_ev[0] && tl_state.pin_ev_dev == dev) return AM_OK;
    {
        Orphans& o = orphans();
        std::lock_guard<std::mutex> lk(o.mu);
        if (tl_state.pin_ev[0] && tl_state.pin_ev_dev >= 0) { o.events[tl_state.pin_ev_dev].push_back(tl_state.pin_ev[0]); o.events[tl_state.pin_ev_dev].push_back(tl_state.pin_ev[1]); }
        tl_state.pin_ev[0] = tl_state.pin_ev[1] = nullptr; tl_state.pin_ev_dev = dev;
        for (int k = 0; k < 2 && !o.events[dev].empty(); k++) { tl_state.pin_ev[k] = o.events[dev].back(); o.events[dev].pop_back(); }
    }
    for (int k = 0; k < 2; k++) if (!tl_state.pin_ev[k]) HIP_TRY(hipEventCreateWithFlags(&tl_state.pin_ev[k], hipEventDisableTiming));
    return AM_OK;
}

ThreadState::~ThreadState()
{
    Orphans& o = orphans();
    std::lock_guard<std::mutex> lk(o.mu);
    for (int d = 0; d < kMaxDev; d++) {
        if (oneshot[d]) { o.batches[d].push_back(oneshot[d]); oneshot[d] = nullptr; }
        if (own[d]) { o.streams[d].push_back(own[d]); own[d] = nullptr; }
    }
    // the pinned buffers and events wait for the next new thread, too (a thread-exit destructor must not call into HIP, so they are not freed)
    if (pin || pin_res) { o.pins.push_back(Orphans::Pins{pin, pin_cap, pin_res, pin_res_cap}); pin = pin_res = nullptr; }
    for (int k = 0; k < 2; k++) if (pin_ev[k] && pin_ev_dev >= 0) { o.events[pin_ev_dev].push_back(pin_ev[k]); pin_ev[k] = nullptr; }
}
}  // namespace

// ------------------------------------------------------------------ automaton

static int prepare(const am_automaton* ca, int case_mode, const Flavor** out)
{
    if (!ca) return fail(AM_ERR_INVALID, "null automaton");
    if (case_mode != AM_CASE_SENSITIVE && case_mode != AM_IGNORE_CASE) return fail(AM_ERR_INVALID, "bad case_mode");
    am_automaton* a = const_cast<am_automaton*>(ca);
    std::lock_guard<std::mutex> lk(a->mu);
    Flavor& f = a->fl[case_mode];
    if (!f.ready) {
        if (!a->has_ref) return fail(AM_ERR_UNSUPPORTED, "this handle was attached to an image of the other case mode");
        AM_TRY(ensure_runtime());
        ON_DEVICE(a->dev);
        std::vector<uint8_t> img; std::string err;
        if (case_mode == AM_CASE_SENSITIVE && !a->cs_image.empty()) img.swap(a->cs_image);
        else {
            RefArrays ref{a->transitions.data(), a->transitions.size(), a->offsets.data(), a->offsets.size() - 1, a->root_ascii.data(), a->values_len.data()};
            if (flatten(ref, case_mode, img, err, a->lower.get()) != 0) return fail(AM_ERR_INVALID, err);
        }
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, img.size());
        if (e != hipSuccess) return fail(AM_ERR_OOM, std::string("hipMalloc(image): ") + hipGetErrorString(e));
        e = hipMemcpy(d, img.data(), img.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d); return fail(AM_ERR_HIP, std::string("hipMemcpy(image): ") + hipGetErrorString(e)); }
        std::memcpy(&f.h, img.data(), sizeof(ImageHeader));
        f.d_image = d; f.bytes = img.size(); f.ready = true;
    }
    *out = &f;
    return AM_OK;
}

extern "C" const char* am_last_error(void) { return g_err.c_str(); }

// lower_from / lower_to (n_lower_pairs of them; null: the built-in Unicode 14.0 table): what `Data.Char.toLower` of the caller's GHC does,
// as (c, toLower c) pairs -- Utf8.hs:145-151 lowerCodePoint, consumed by the IgnoreCase image (unlower sets baked into the suffix
// structure's byte edges, the general kernel's delta table).
extern "C" int am_automaton_create_ex(const uint64_t* transitions, size_t n_transitions, const uint32_t* offsets, size_t n_states,
                                      const uint64_t* root_ascii, const uint32_t* values_len,
                                      const uint32_t* lower_from, const uint32_t* lower_to, size_t n_lower_pairs, am_automaton** out)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!transitions || !offsets || !root_ascii || !values_len || n_states == 0) return fail(AM_ERR_INVALID, "null or empty automaton arrays");
    std::shared_ptr<const LowerTable> lower;
    if (lower_from || lower_to || n_lower_pairs) {
        if (!lower_from || !lower_to) return fail(AM_ERR_INVALID, "lower_from and lower_to must both be given");
        auto lt = std::make_shared<LowerTable>();
        std::string err;
        if (LowerTable::make(lower_from, lower_to, n_lower_pairs, *lt, err) != 0) return fail(AM_ERR_INVALID, err);
        if (lt->hash != builtin_lower_table().hash) lower = lt;            // the built-in table handed back to us: nothing to keep
    }
    // validate on the host right away (flatten checks every index); the image is uploaded on first use
    std::vector<uint8_t> img;
    {
        std::string err;
        RefArrays ref{transitions, n_transitions, offsets, n_states, root_ascii, values_len};
        if (flatten(ref, AM_CASE_SENSITIVE, img, err, lower.get()) != 0) return fail(AM_ERR_INVALID, err);
    }
    am_automaton* a = new am_automaton();
    a->lower = lower;
    a->cs_image.swap(img);
    a->transitions.assign(transitions, transitions + n_transitions);
    a->offsets.assign(offsets, offsets + n_states + 1);
    a->root_ascii.assign(root_ascii, root_ascii + 128);
    a->values_len.assign(values_len, values_len + n_states);
    a->has_ref = true;
    // the automaton belongs to the device that is current now (its images are uploaded there on first use); without a
    // device the handle can still be made and inspected, every run entry point then fails with AM_ERR_NO_DEVICE
    { int d = 0; if (current_device(&d) == AM_OK) a->dev = d; else g_err.clear(); }
    *out = a;
    return AM_OK;
}

extern "C" int am_automaton_create(const uint64_t* transitions, size_t n_transitions, const uint32_t* offsets, size_t n_states,
                                   const uint64_t* root_ascii, const uint32_t* values_len, am_automaton** out)
{
    return am_automaton_create_ex(transitions, n_transitions, offsets, n_states, root_ascii, values_len, nullptr, nullptr, 0, out);
}

// identifies the lower-case table of the handle's IgnoreCase image (ImageHeader::flags); am_lower_table_hash of the same pairs agrees
extern "C" uint32_t am_automaton_lower_hash(const am_automaton* a)
{
    if (!a) return 0;
    for (const Flavor& f : a->fl) if (f.ready && !a->has_ref) return f.h.flags;      // attached to an image: what the image says
    return a->lower ? a->lower->hash : builtin_lower_table().hash;
}

extern "C" uint32_t am_lower_table_hash(const uint32_t* lower_from, const uint32_t* lower_to, size_t n_pairs)
{
    if (!lower_from || !lower_to) return builtin_lower_table().hash;
    LowerTable lt; std::string err;
    if (LowerTable::make(lower_from, lower_to, n_pairs, lt, err) != 0) { g_err = err; return 0; }
    return lt.hash;
}

extern "C" void am_automaton_destroy(am_automaton* a)
{
    if (!a) return;
    for (Flavor& f : a->fl) if (f.d_image) (void)hipFree(f.d_image);      // hipFree finds the owning device itself
    delete a;
}

extern "C" int am_automaton_set_kernel(am_automaton* a, int k)
{
    if (!a || k < 0 || k > 2) return fail(AM_ERR_INVALID, "bad arguments");
    a->kernel_pref = k;
    return AM_OK;
}

extern "C" int am_automaton_image_size(const am_automaton* a, int case_mode, size_t* nbytes)
{
    const Flavor* f; AM_TRY(prepare(a, case_mode, &f));
    if (!nbytes) return fail(AM_ERR_INVALID, "nbytes is null");
    *nbytes = f->bytes;
    return AM_OK;
}

extern "C" int am_automaton_image_copy(const am_automaton* a, int case_mode, void* d_dst, size_t nbytes)
{
    const Flavor* f; AM_TRY(prepare(a, case_mode, &f));
    if (!d_dst || nbytes < f->bytes) return fail(AM_ERR_INVALID, "destination too small");
    ON_DEVICE(a->dev);
    HIP_TRY(hipMemcpy(d_dst, f->d_image, f->bytes, hipMemcpyDefault));       // d_dst may live on another device (peer copy)
    return AM_OK;
}

extern "C" int am_automaton_from_image(const void* d_image, size_t nbytes, am_automaton** out)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!d_image || nbytes < sizeof(ImageHeader)) return fail(AM_ERR_INVALID, "image too small");
    int dev = 0;
    AM_TRY(device_of_pointer(d_image, &dev));             // the new handle lives where the received image lies
    ON_DEVICE(dev);
    ImageHeader h;
    HIP_TRY(hipMemcpy(&h, d_image, sizeof(h), hipMemcpyDeviceToHost));
    if (!image_sections_in_bounds(h) || h.total_bytes > nbytes) return fail(AM_ERR_INVALID, "not an automaton image");
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, h.total_bytes));
    hipError_t e = hipMemcpy(d, d_image, h.total_bytes, hipMemcpyDeviceToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return fail(AM_ERR_HIP, hipGetErrorString(e)); }
    am_automaton* a = new am_automaton();
    a->dev = dev;
    Flavor& f = a->fl[h.case_mode];
    f.h = h; f.d_image = d; f.bytes = h.total_bytes; f.ready = true;
    *out = a;
    return AM_OK;
}

// Serialised form = the image blob itself (position independent; magic, version and a checksum in its header).
extern "C" int am_automaton_image_read(const am_automaton* a, int case_mode, void* host_dst, size_t nbytes)
{
    const Flavor* f; AM_TRY(prepare(a, case_mode, &f));
    if (!host_dst || nbytes < f->bytes) return fail(AM_ERR_INVALID, "destination too small");
    ON_DEVICE(a->dev);
    HIP_TRY(hipMemcpy(host_dst, f->d_image, f->bytes, hipMemcpyDeviceToHost));
    return AM_OK;
}

extern "C" int am_automaton_from_host_image(const void* image, size_t nbytes, am_automaton** out)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!image || nbytes < sizeof(ImageHeader)) return fail(AM_ERR_INVALID, "image too small");
    ImageHeader h;
    std::memcpy(&h, image, sizeof(h));
    if (!image_sections_in_bounds(h) || h.total_bytes > nbytes) return fail(AM_ERR_INVALID, "not an automaton image (magic, version or section bounds)");
    // the validation below reads the sections through their own types (16-byte vectors, 64-byte lines): a caller's buffer need not
    // be aligned for that, so a misaligned one is copied first
    std::vector<uint64_t> aligned_copy;
    if (((uintptr_t)image & 63u) != 0) {
        try { aligned_copy.resize(((size_t)h.total_bytes + 7) / 8 + 8); } catch (const std::exception&) { return fail(AM_ERR_OOM, "no memory for an aligned copy of the image"); }
        uint8_t* base = (uint8_t*)aligned_copy.data();
        base += (64 - ((uintptr_t)base & 63u)) & 63u;
        std::memcpy(base, image, (size_t)h.total_bytes);
        image = base;
    }
    if (image_checksum((const uint8_t*)image + sizeof(h), (size_t)h.total_bytes - sizeof(h)) != h.checksum) return fail(AM_ERR_INVALID, "automaton image is corrupt (checksum)");
    { std::string why; if (!image_body_valid((const uint8_t*)image, h, why)) return fail(AM_ERR_INVALID, why); }       // the checksum is no proof of origin: check what the kernels will follow
    int dev = 0;
    AM_TRY(current_device(&dev));
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, h.total_bytes));
    hipError_t e = hipMemcpy(d, image, h.total_bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return fail(AM_ERR_HIP, hipGetErrorString(e)); }
    am_automaton* a = new am_automaton();
    a->dev = dev;
    Flavor& f = a->fl[h.case_mode];
    f.h = h; f.d_image = d; f.bytes = h.total_bytes; f.ready = true;
    *out = a;
    return AM_OK;
}

// ------------------------------------------------------------------ batches

static int finish_batch(am_batch* b)
{
    b->hidx_ready = false;
    if (b->total > 0) AM_TRY(b->hidx.ensure(((b->total >> kHidxShift) + 2) * sizeof(uint32_t)));
    // the batch's block of counters is allocated HERE, once, before the batch is visible to other threads: make_plan reads its
    // address without the batch lock, so it must never be re-allocated later (every later ensure(64) is a no-op)
    AM_TRY(b->small.ensure(64));
    return AM_OK;
}

// Uploads the slices into `b` (re-using its device buffers when they are large enough).
static int upload_slices(const am_slice* hay, size_t n_hay, am_batch* b, bool oneshot = false)
{
    if (n_hay && !hay) return fail(AM_ERR_INVALID, "hay is null");
    if (n_hay >= 0xFFFFFFFFull) return fail(AM_ERR_INVALID, "too many haystacks");
    AM_TRY(ensure_runtime());
    ON_DEVICE(b->dev);
    std::vector<uint64_t> offs(n_hay + 1, 0);
    for (size_t i = 0; i < n_hay; i++) {
        if (hay[i].len && !hay[i].ptr) return fail(AM_ERR_INVALID, "slice with null ptr");
        offs[i + 1] = offs[i] + hay[i].len;
    }
    const uint64_t total = offs[n_hay];
    const size_t padded = (size_t)((total + 15) & ~15ull) + 16;
    b->owns = true; b->total = total; b->n_hay = (uint32_t)n_hay;
    if (total <= kSmallUpload && n_hay <= (1u << 16)) {
        // Small batches (the one-document-per-call pattern): [offsets | text] is put together in pinned memory of the calling thread and goes
        // up with asynchronous copies on the thread's stream; the kernels of the call follow on the same stream, so nothing waits here.
        // Above kPinPiece the text is cut into pieces: the DMA of one piece runs while the host gathers the next into the other half.
        const size_t text_off = (offs.size() * sizeof(uint64_t) + 63) & ~(size_t)63;
        const size_t bytes = text_off + padded;
        hipStream_t st; AM_TRY(get_stream(b->dev, &st));
        AM_TRY(b->combo.ensure(bytes));
        b->d_offsets = (uint64_t*)b->combo.p; b->d_text = (uint8_t*)b->combo.p + text_off;
        const bool pieces = padded > kPinPiece;
        AM_TRY(pin_ensure(tl_state.pin, tl_state.pin_cap, pieces ? text_off + 2 * kPinPiece : bytes));
        uint8_t* pin = tl_state.pin;
        std::memcpy(pin, offs.data(), offs.size() * sizeof(uint64_t));
        // copies bytes [lo, hi) of the concatenated (zero-padded) text into dst
        auto gather = [&](uint64_t lo, uint64_t hi, uint8_t* dst) {
            uint64_t at = lo;
            if (at < total) {
                size_t i = (size_t)(std::upper_bound(offs.begin(), offs.end(), at) - offs.begin()) - 1;
                const uint64_t stop_all = std::min<uint64_t>(hi, total);
                while (at < stop_all) {
                    while (offs[i + 1] <= at) i++;
                    const uint64_t stop = std::min<uint64_t>(stop_all, offs[i + 1]);
                    std::memcpy(dst + (at - lo), hay[i].ptr + hay[i].off + (at - offs[i]), (size_t)(stop - at));
                    at = stop;
                }
            }
            if (at < hi) std::memset(dst + (at - lo), 0, (size_t)(hi - at));        // zero tail: kernels read whole 16-byte groups
        };
        if (!pieces) {
            gather(0, padded, pin + text_off);
            HIP_TRY(hipMemcpyAsync(b->combo.p, pin, bytes, hipMemcpyHostToDevice, st));
        } else {
            AM_TRY(pin_events(b->dev));
            HIP_TRY(hipMemcpyAsync(b->combo.p, pin, text_off, hipMemcpyHostToDevice, st));
            bool used[2] = {false, false};
            int turn = 0;
            for (uint64_t lo = 0; lo < padded; lo += kPinPiece, turn ^= 1) {
                const uint64_t hi = std::min<uint64_t>(padded, lo + kPinPiece);
                uint8_t* half = pin + text_off + (size_t)turn * kPinPiece;
                if (used[turn]) HIP_TRY(hipEventSynchronize(tl_state.pin_ev[turn]));       // the copy out of this half has finished
                gather(lo, hi, half);
                HIP_TRY(hipMemcpyAsync((uint8_t*)b->d_text + lo, half, (size_t)(hi - lo), hipMemcpyHostToDevice, st));
                HIP_TRY(hipEventRecord(tl_state.pin_ev[turn], st));
                used[turn] = true;
            }
        }
        if (!oneshot) HIP_TRY(hipStreamSynchronize(st));       // a batch object may be used from any thread and stream afterwards
        return finish_batch(b);
    }
    AM_TRY(b->text_buf.ensure(padded));
    AM_TRY(b->offs_buf.ensure(offs.size() * sizeof(uint64_t)));
    b->d_text = b->text_buf.p; b->d_offsets = (uint64_t*)b->offs_buf.p;
    hipError_t e = hipMemcpy(b->d_offsets, offs.data(), offs.size() * sizeof(uint64_t), hipMemcpyHostToDevice);
    // The slices are gathered piece by piece (several threads) into two pinned staging buffers that take turns:
    // while the DMA engine uploads one piece, the host fills the other.  The buffers stay for the next call.
    {
        struct UploadStage { std::mutex mu; uint8_t* stage[2] = {nullptr, nullptr}; hipStream_t copy_stream = nullptr; hipEvent_t done[2] = {nullptr, nullptr}; };
        static UploadStage per_device[kMaxDev];                  // pinned staging + copy stream of each device; big uploads to one device take turns (they share its PCIe link anyway)
        UploadStage& us = per_device[b->dev];
        uint8_t* (&stage)[2] = us.stage;
        hipStream_t& copy_stream = us.copy_stream;
        hipEvent_t (&done)[2] = us.done;
        constexpr size_t kPiece = 32u << 20;
        std::lock_guard<std::mutex> stage_lk(us.mu);
        if (!stage[0]) {
            if (hipHostMalloc((void**)&stage[0], kPiece, hipHostMallocPortable) != hipSuccess || hipHostMalloc((void**)&stage[1], kPiece, hipHostMallocPortable) != hipSuccess ||
                hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&done[0]) != hipSuccess || hipEventCreate(&done[1]) != hipSuccess) {
                if (stage[0]) (void)hipHostFree(stage[0]);
                if (stage[1]) (void)hipHostFree(stage[1]);
                stage[0] = stage[1] = nullptr;
                return fail(AM_ERR_OOM, "pinned staging buffers / copy stream could not be created");
            }
        }
        const unsigned hw = std::thread::hardware_concurrency();
        const unsigned n_threads = std::max(1u, std::min(8u, hw ? hw : 1u));
        // copies bytes [lo, hi) of the concatenated batch into dst
        auto gather = [&](uint64_t lo, uint64_t hi, uint8_t* dst) {
            size_t i = (size_t)(std::upper_bound(offs.begin(), offs.end(), lo) - offs.begin()) - 1;
            uint64_t at = lo;
            while (at < hi) {
                while (offs[i + 1] <= at) i++;
                const uint64_t stop = std::min<uint64_t>(hi, offs[i + 1]);
                std::memcpy(dst + (at - lo), hay[i].ptr + hay[i].off + (at - offs[i]), (size_t)(stop - at));
                at = stop;
            }
        };
        bool used[2] = {false, false};
        int turn = 0;
        for (uint64_t lo = 0; lo < total && e == hipSuccess; lo += kPiece, turn ^= 1) {
            const uint64_t hi = std::min<uint64_t>(total, lo + kPiece);
            if (used[turn]) e = hipEventSynchronize(done[turn]);          // the previous upload out of this buffer has finished
            if (e != hipSuccess) break;
            const uint64_t len = hi - lo;
            if (n_threads == 1 || len < (4u << 20)) gather(lo, hi, stage[turn]);
            else {
                std::vector<std::thread> pool;
                const uint64_t step = (len + n_threads - 1) / n_threads;
                for (unsigned t = 0; t < n_threads; t++) {
                    const uint64_t a = lo + t * step, z = std::min<uint64_t>(hi, a + step);
                    if (a < z) pool.emplace_back(gather, a, z, stage[turn] + (a - lo));
                }
                for (auto& th : pool) th.join();
            }
            e = hipMemcpyAsync((uint8_t*)b->d_text + lo, stage[turn], (size_t)len, hipMemcpyHostToDevice, copy_stream);
            if (e == hipSuccess) e = hipEventRecord(done[turn], copy_stream);
            used[turn] = true;
        }
        if (e == hipSuccess) e = hipMemsetAsync((uint8_t*)b->d_text + total, 0, padded - (size_t)total, copy_stream);    // zero tail: kernels read whole 16-byte groups
        if (e == hipSuccess) e = hipStreamSynchronize(copy_stream);
    }
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? AM_ERR_OOM : AM_ERR_HIP, std::string("batch upload: ") + hipGetErrorString(e));
    return finish_batch(b);
}

extern "C" int am_batch_upload(const am_slice* hay, size_t n_hay, am_batch** out)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    am_batch* b = new am_batch();
    int rc = current_device(&b->dev);
    if (rc == AM_OK) rc = upload_slices(hay, n_hay, b);
    if (rc != AM_OK) { am_batch_destroy(b); return rc; }
    *out = b;
    return AM_OK;
}

// One-shot entry points keep one batch object (device text + workspaces) per calling thread and device between calls, so
// that a caller that scans one document per call does not pay a dozen hipMalloc/hipFree each time -- and calls from
// different threads share nothing.  Anything larger than 256 MiB is let go.
namespace {
am_batch* oneshot_get(int dev)
{
    am_batch*& b = tl_state.oneshot[dev];
    if (!b) b = adopt_batch(dev);
    if (!b) { b = new am_batch(); b->dev = dev; }
    return b;
}
void oneshot_trim(int dev)
{
    am_batch*& b = tl_state.oneshot[dev];
    if (!b) return;
    const size_t held = b->text_buf.cap + b->pool.cap + b->hidx.cap + b->unit_offsets.cap + b->hay_counts.cap;
    if (held > (256ull << 20)) { am_batch_destroy(b); b = nullptr; }
}
}  // namespace

extern "C" int am_batch_from_device(const void* d_bytes, const void* d_offsets, size_t n_hay, uint64_t total_bytes, am_batch** out)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!d_offsets || (total_bytes && !d_bytes)) return fail(AM_ERR_INVALID, "null device pointers");
    int dev = 0;
    AM_TRY(device_of_pointer(total_bytes ? d_bytes : d_offsets, &dev));      // the batch lives where its memory does
    ON_DEVICE(dev);
    if (((uintptr_t)d_bytes & 15) != 0) return fail(AM_ERR_INVALID, "d_bytes must be 16-byte aligned");
    if (n_hay >= 0xFFFFFFFFull) return fail(AM_ERR_INVALID, "too many haystacks");
    uint64_t first = 1, last = 0;
    HIP_TRY(hipMemcpy(&first, d_offsets, 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&last, (const uint64_t*)d_offsets + n_hay, 8, hipMemcpyDeviceToHost));
    if (first != 0 || last != total_bytes) return fail(AM_ERR_INVALID, "d_offsets[0] must be 0 and d_offsets[n_hay] must equal total_bytes");
    am_batch* b = new am_batch();
    b->dev = dev;
    b->owns = false; b->d_text = const_cast<void*>(d_bytes); b->d_offsets = (uint64_t*)const_cast<void*>(d_offsets);
    b->total = total_bytes; b->n_hay = (uint32_t)n_hay;
    int rc = finish_batch(b);
    if (rc != AM_OK) { am_batch_destroy(b); return rc; }
    *out = b;
    return AM_OK;
}

extern "C" void am_batch_destroy(am_batch* b)
{
    if (!b) return;
    for (DevBuf* d : {&b->text_buf, &b->offs_buf, &b->combo, &b->hidx, &b->unit_counts, &b->unit_offsets, &b->scan_tmp, &b->small, &b->hay_counts, &b->flags, &b->unit_first, &b->pool, &b->block_next,
                      &b->sparse, &b->dense_counts, &b->dense_offsets, &b->dense_out}) d->release();
    delete b;
}

extern "C" uint64_t am_batch_total_bytes(const am_batch* b) { return b ? b->total : 0; }

// ------------------------------------------------------------------ scanning

namespace {

struct Plan {
    const Flavor* f; bool ic; bool use_sf; bool nothing; uint64_t n_units; uint32_t unit_chunks; int n_cu;
    bool dense;          // automaton with the empty needle on the suffix-filter route: k_sf's records + the dense pass (am_dense.hip)
    AcView ac; SfView sf; BatchView bv;
    am_batch* batch;
    uint32_t* next_unit; // k_sf's unit counter (in the batch's `small` block: [0..1] total_values, [4] block counter, [5] overflow, [8] this)
};

int make_plan(const am_automaton* a, int case_mode, am_batch* b, Plan& p)
{
    if (!b) return fail(AM_ERR_INVALID, "null batch");
    if (!a) return fail(AM_ERR_INVALID, "null automaton");
    if (a->dev != b->dev) return fail(AM_ERR_INVALID, "automaton and batch live on different devices");
    AM_TRY(prepare(a, case_mode, &p.f));
    p.ic = case_mode == AM_IGNORE_CASE;
    if (a->kernel_pref == 2 && !p.f->h.sf_enabled) return fail(AM_ERR_UNSUPPORTED, "suffix-filter kernel cannot run this automaton (empty needle with too many prefix terminals)");
    p.use_sf = p.f->h.sf_enabled && a->kernel_pref != 1;
    p.dense = p.use_sf && p.f->h.root_vlen > 0;
    p.ac = make_ac_view(p.f->d_image, p.f->h);
    p.sf = make_sf_view(p.f->d_image, p.f->h);
    p.bv = BatchView{(const uint8_t*)b->d_text, b->d_offsets, (const uint32_t*)b->hidx.p, b->total, b->n_hay, 0};
    // no goto edge at all (no needles, or only empty needles): the reference never reports anything
    const bool no_edges = p.f->h.n_transitions == p.f->h.n_states;
    p.nothing = b->total == 0 || no_edges || (p.use_sf && p.f->h.sf_tiers == 0 && !p.dense);      // dense: first code points still report the root's values
    p.unit_chunks = p.use_sf ? sf_unit_chunks(p.bv, g_rt.dev[b->dev].n_cu) : 0;
    p.next_unit = nullptr;
    if (p.use_sf) { if (!b->small.p) return fail(AM_ERR_INVALID, "batch without its counter block (not made by am_batch_upload / am_batch_from_device)"); p.next_unit = (uint32_t*)b->small.p + 8; }
    p.n_cu = g_rt.dev[b->dev].n_cu;
    p.batch = b;
    p.n_units = p.nothing ? 0 : (p.use_sf ? (sf_chunks(p.bv) + p.unit_chunks - 1) / p.unit_chunks : ac_units(p.ac, p.bv));
    if (p.n_units >= 0x7FFFFFF0ull) return fail(AM_ERR_UNSUPPORTED, "batch too large for one launch; split it");
    return AM_OK;
}

int launch_scan_kernel(const Plan& p, int mode, const ScanOut& o, hipStream_t st)
{
    if (p.use_sf) {
        ScanOut os = o;
        os.next_unit = p.next_unit;
        Prof pr("sf", st);
        HIP_TRY(launch_sf(p.ic, mode, p.sf, p.bv, os, p.n_cu, st));
    }
    else { Prof pr("ac", st); HIP_TRY(launch_ac(p.ic, mode, p.ac, p.bv, o, st)); }
    return AM_OK;
}

int build_hidx(const Plan& p, am_batch* b, hipStream_t st)
{
    if (b->hidx_ready) return AM_OK;
    Prof pr("hidx", st);
    HIP_TRY(launch_hidx(p.bv, (uint32_t*)b->hidx.p, (b->total >> kHidxShift) + 2, st));
    b->hidx_ready = true;
    return AM_OK;
}

// the haystack index and the clearing of (up to two) arrays in ONE launch when the index has to be built anyway -- the one-document call;
// with the index in place the arrays are cleared by memsets.  bytes0 / bytes1 are multiples of 4.
int build_hidx_and_clear(const Plan& p, am_batch* b, hipStream_t st, void* z0, size_t bytes0, void* z1, size_t bytes1)
{
    if (b->hidx_ready) {
        if (bytes0) HIP_TRY(hipMemsetAsync(z0, 0, bytes0, st));
        if (bytes1) HIP_TRY(hipMemsetAsync(z1, 0, bytes1, st));
        return AM_OK;
    }
    Prof pr("hidx", st);
    HIP_TRY(launch_hidx(p.bv, (uint32_t*)b->hidx.p, (b->total >> kHidxShift) + 2, st, (uint32_t*)z0, bytes0 / 4, (uint32_t*)z1, bytes1 / 4));
    b->hidx_ready = true;
    return AM_OK;
}

}  // namespace

static int run_records(const am_automaton* a, int case_mode, am_batch* b, const std::function<int(uint64_t, Record**)>& sink_final, uint64_t* n_out, bool have_lock = false);

// count / containsAny of an automaton with the empty needle on the suffix-filter route: a record at almost every position, so
// the records are made (k_sf + dense pass) and reduced
static int reduce_dense(const am_automaton* a, int case_mode, am_batch* b, uint64_t* counts_out, uint64_t* total_out, uint8_t* flags_out)
{
    uint64_t n_rec = 0;
    auto sink = [&](uint64_t n, Record** ptr) -> int { AM_TRY(b->dense_out.ensure(n * sizeof(Record))); *ptr = (Record*)b->dense_out.p; return AM_OK; };
    const Flavor* f = nullptr;
    AM_TRY(prepare(a, case_mode, &f));
    std::lock_guard<std::mutex> lk(b->mu);          // ONE lock over the scan and the reduction: b->dense_out must not be refilled by another thread in between
    AM_TRY(run_records(a, case_mode, b, sink, &n_rec, true));
    if (n_rec == 0) return AM_OK;
    ON_DEVICE(b->dev);
    hipStream_t st; AM_TRY(get_stream(b->dev, &st));
    AM_TRY(b->small.ensure(64));
    HIP_TRY(hipMemsetAsync(b->small.p, 0, 64, st));
    if (counts_out) { AM_TRY(b->hay_counts.ensure((size_t)b->n_hay * 8)); HIP_TRY(hipMemsetAsync(b->hay_counts.p, 0, (size_t)b->n_hay * 8, st)); }
    if (flags_out) { AM_TRY(b->flags.ensure(b->n_hay)); HIP_TRY(hipMemsetAsync(b->flags.p, 0, b->n_hay, st)); }
    const AcView ac = make_ac_view(f->d_image, f->h);
    HIP_TRY(launch_records_reduce((const Record*)b->dense_out.p, n_rec, ac.vlen, counts_out ? (uint64_t*)b->hay_counts.p : nullptr, (uint64_t*)b->small.p,
                                  flags_out ? (uint8_t*)b->flags.p : nullptr, st));
    uint64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, b->small.p, 8, hipMemcpyDeviceToHost, st));
    if (counts_out) HIP_TRY(hipMemcpyAsync(counts_out, b->hay_counts.p, (size_t)b->n_hay * 8, hipMemcpyDeviceToHost, st));
    if (flags_out) HIP_TRY(hipMemcpyAsync(flags_out, b->flags.p, b->n_hay, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (total_out) *total_out = total;
    return AM_OK;
}

extern "C" int am_count_batch(const am_automaton* a, int case_mode, const am_batch* cb, uint64_t* counts_out, uint64_t* total_out)
{
    am_batch* b = const_cast<am_batch*>(cb);
    Plan p; AM_TRY(make_plan(a, case_mode, b, p));
    if (total_out) *total_out = 0;
    if (counts_out && b->n_hay) std::memset(counts_out, 0, (size_t)b->n_hay * sizeof(uint64_t));
    if (p.nothing) return AM_OK;
    if (p.dense) return reduce_dense(a, case_mode, b, counts_out, total_out, nullptr);
    std::lock_guard<std::mutex> lk(b->mu);
    ON_DEVICE(b->dev);
    hipStream_t st; AM_TRY(get_stream(b->dev, &st));
    AM_TRY(b->small.ensure(64));
    ScanOut o{};
    o.unit_chunks = p.unit_chunks;
    if (!p.use_sf) { AM_TRY(b->unit_counts.ensure((p.n_units + 1) * sizeof(uint32_t))); o.unit_counts = (uint32_t*)b->unit_counts.p; }
    o.total_values = (uint64_t*)b->small.p;
    if (counts_out) {
        AM_TRY(b->hay_counts.ensure((size_t)b->n_hay * sizeof(uint64_t)));
        o.hay_counts = (uint64_t*)b->hay_counts.p;
    }
    AM_TRY(build_hidx_and_clear(p, b, st, b->small.p, 64, counts_out ? b->hay_counts.p : nullptr, counts_out ? (size_t)b->n_hay * sizeof(uint64_t) : 0));
    if (p.use_sf) o.pool_ctrl = (uint32_t*)b->small.p + 4;          // ([2]: the role-specialised kernel's watchdog reports here)
    AM_TRY(launch_scan_kernel(p, kModeCount, o, st));
    uint64_t head[4] = {0, 0, 0, 0};                                // total_values, -, {pool counter, overflow}, {watchdog, -}
    ResultCopies rc;
    AM_TRY(rc.add(head, b->small.p, 32, st));
    if (counts_out) AM_TRY(rc.add(counts_out, b->hay_counts.p, (size_t)b->n_hay * sizeof(uint64_t), st));
    AM_TRY(rc.finish(st));
    if ((uint32_t)head[3] != 0) return fail(AM_ERR_HIP, "suffix-filter kernel: internal hand-over between its wavefronts timed out (watchdog)");
    if (total_out) *total_out = head[0];
    return AM_OK;
}

extern "C" int am_contains_any_batch(const am_automaton* a, int case_mode, const am_batch* cb, uint8_t* flags_out)
{
    am_batch* b = const_cast<am_batch*>(cb);
    Plan p; AM_TRY(make_plan(a, case_mode, b, p));
    if (!flags_out && b->n_hay) return fail(AM_ERR_INVALID, "flags_out is null");
    if (b->n_hay) std::memset(flags_out, 0, b->n_hay);
    if (p.nothing) return AM_OK;
    if (p.dense) return reduce_dense(a, case_mode, b, nullptr, nullptr, flags_out);
    std::lock_guard<std::mutex> lk(b->mu);
    ON_DEVICE(b->dev);
    hipStream_t st; AM_TRY(get_stream(b->dev, &st));
    AM_TRY(b->flags.ensure(((size_t)b->n_hay + 3) & ~(size_t)3));
    AM_TRY(b->small.ensure(64));
    ScanOut o{};
    o.unit_chunks = p.unit_chunks;
    o.flags = (uint8_t*)b->flags.p;
    AM_TRY(build_hidx_and_clear(p, b, st, b->flags.p, ((size_t)b->n_hay + 3) & ~(size_t)3, b->small.p, 64));      // (the counter block: k_sf's unit ticket)
    AM_TRY(launch_scan_kernel(p, kModeAny, o, st));
    ResultCopies rc;
    AM_TRY(rc.add(flags_out, b->flags.p, b->n_hay, st));
    AM_TRY(rc.finish(st));
    return AM_OK;
}

// The whole scan: leaves every record of the batch, sorted by (haystack, end_pos), in device memory
// obtained from `sink(total, &ptr)` (called once, only when total > 0); *n_out = number of records.
static int run_records(const am_automaton* a, int case_mode, am_batch* b, const std::function<int(uint64_t, Record**)>& sink_final, uint64_t* n_out, bool have_lock)
{
    *n_out = 0;
    Plan p; AM_TRY(make_plan(a, case_mode, b, p));
    if (p.nothing) return AM_OK;
    std::unique_lock<std::mutex> lk(b->mu, std::defer_lock);
    if (!have_lock) lk.lock();
    ON_DEVICE(b->dev);
    hipStream_t st; AM_TRY(get_stream(b->dev, &st));
    // automata with the empty needle: k_sf's (sparse) records go to a buffer of the batch, the dense pass writes the result
    uint64_t n_sparse = 0;
    auto sink_sparse = [&](uint64_t n, Record** ptr) -> int { AM_TRY(b->sparse.ensure(n * sizeof(Record))); *ptr = (Record*)b->sparse.p; return AM_OK; };
    const std::function<int(uint64_t, Record**)>& sink = p.dense ? std::function<int(uint64_t, Record**)>(sink_sparse) : sink_final;
    uint64_t* n_scan = p.dense ? &n_sparse : n_out;
    const uint64_t n = p.n_units + 1;           // trailing zero: offsets[n_units] = total
    AM_TRY(b->unit_counts.ensure(n * sizeof(uint32_t)));
    AM_TRY(b->unit_offsets.ensure(n * sizeof(uint64_t)));
    AM_TRY(b->small.ensure(64));
    size_t tmp_bytes = 0;
    if (scan_temp_bytes(n, &tmp_bytes) != hipSuccess) return fail(AM_ERR_HIP, "scan sizing failed");
    AM_TRY(b->scan_tmp.ensure(tmp_bytes + 16));
    Record* d_records = nullptr;
    // general kernel: count pass -> exclusive scan -> emit pass (unit = one lane's chunk)
    auto body_ac = [&]() -> int {
        ScanOut o{};
        o.unit_counts = (uint32_t*)b->unit_counts.p;
        o.total_values = (uint64_t*)b->small.p;
        HIP_TRY(hipMemsetAsync(b->small.p, 0, 64, st));
        HIP_TRY(hipMemsetAsync((uint32_t*)b->unit_counts.p + p.n_units, 0, sizeof(uint32_t), st));
        AM_TRY(build_hidx(p, b, st));
        AM_TRY(launch_scan_kernel(p, kModeCount, o, st));
        { Prof pr("scan", st); HIP_TRY(launch_scan(b->scan_tmp.p, tmp_bytes, (const uint32_t*)b->unit_counts.p, (uint64_t*)b->unit_offsets.p, n, st)); }
        uint64_t total = 0;
        HIP_TRY(hipMemcpyAsync(&total, (uint64_t*)b->unit_offsets.p + p.n_units, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        *n_scan = total;
        if (total == 0) return AM_OK;
        AM_TRY(sink(total, &d_records));
        ScanOut w{};
        w.unit_offsets = (const uint64_t*)b->unit_offsets.p;
        w.records = d_records;
        AM_TRY(launch_scan_kernel(p, kModeEmit, w, st));
        HIP_TRY(hipStreamSynchronize(st));
        return AM_OK;
    };
    // suffix-filter kernel: ONE scan pass writes records into pool blocks (chained per unit), then
    // scan(unit_counts) + k_permute put them in order.  The pool size is a guess (1 record per 128
    // haystack bytes + one block per unit); if it overflows the kernel still counts, and the pass is
    // repeated once with the exact number of blocks.
    auto body_sf = [&]() -> int {
        AM_TRY(b->unit_first.ensure(2 * p.n_units * sizeof(uint32_t)));          // first block + slot count per unit
        uint64_t want_blocks = b->total / (128 * kPoolBlock) + p.n_units + 1024 + pool_grant_slack(p.n_cu, p.n_units, sf_lds_bytes(p.sf) <= 80 * 1024);
        if (b->pool.cap / (kPoolBlock * sizeof(Record)) > want_blocks) want_blocks = b->pool.cap / (kPoolBlock * sizeof(Record));
        if (cfg::get(cfg::kSfPoolBlocks) > 0) want_blocks = (uint64_t)cfg::get(cfg::kSfPoolBlocks);   // tests: force the overflow/retry path
        for (int attempt = 0; attempt < 4; attempt++) {
            if (want_blocks >= (1ull << 26)) return fail(AM_ERR_UNSUPPORTED, "too many match records for one call (2^32 record slots); split the batch");      // k_sf addresses record slots with 32 bits
            AM_TRY(b->pool.ensure(want_blocks * kPoolBlock * sizeof(Record)));
            AM_TRY(b->block_next.ensure(want_blocks * sizeof(uint32_t)));
            ScanOut o{};
            o.unit_chunks = p.unit_chunks;
            o.unit_counts = (uint32_t*)b->unit_counts.p;
            o.unit_first = (uint32_t*)b->unit_first.p;
            o.unit_slots = (uint32_t*)b->unit_first.p + p.n_units;
            o.pool = (Record*)b->pool.p;
            o.block_next = (uint32_t*)b->block_next.p;
            o.pool_ctrl = (uint32_t*)b->small.p + 4;            // small: [0..1] total_values, [4] block counter, [5] overflow
            o.n_blocks = (uint32_t)want_blocks;
            HIP_TRY(hipMemsetAsync(b->small.p, 0, 64, st));
            HIP_TRY(hipMemsetAsync((uint32_t*)b->unit_counts.p + p.n_units, 0, sizeof(uint32_t), st));
            AM_TRY(build_hidx(p, b, st));
            AM_TRY(launch_scan_kernel(p, kModeEmit, o, st));
            { Prof pr("scan", st); HIP_TRY(launch_scan(b->scan_tmp.p, tmp_bytes, (const uint32_t*)b->unit_counts.p, (uint64_t*)b->unit_offsets.p, n, st)); }
            uint64_t total = 0; uint32_t ctrl[4] = {0, 0, 0, 0};
            HIP_TRY(hipMemcpyAsync(&total, (uint64_t*)b->unit_offsets.p + p.n_units, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(ctrl, o.pool_ctrl, 16, hipMemcpyDeviceToHost, st));      // [0] blocks drawn, [1] overflow, [2] kernel watchdog
            HIP_TRY(hipStreamSynchronize(st));
            if (ctrl[2]) return fail(AM_ERR_HIP, "suffix-filter kernel: internal hand-over between its wavefronts timed out (watchdog)");
            if (ctrl[1]) { want_blocks = (uint64_t)ctrl[0] + 64 + pool_grant_slack(p.n_cu, p.n_units, sf_lds_bytes(p.sf) <= 80 * 1024); continue; }    // pool too small: ctrl[0] = blocks actually needed
            *n_scan = total;
            if (total == 0) return AM_OK;
            AM_TRY(sink(total, &d_records));
            { Prof pr("permute", st); HIP_TRY(launch_permute(o, (const uint64_t*)b->unit_offsets.p, d_records, p.n_units, st)); }
            HIP_TRY(hipStreamSynchronize(st));
            return AM_OK;
        }
        return fail(AM_ERR_HIP, "record pool overflowed repeatedly (internal error)");
    };
    if (!p.dense) return p.use_sf ? body_sf() : body_ac();
    if (p.f->h.sf_tiers != 0) AM_TRY(body_sf());
    else {                                                  // no needle end is reachable (e.g. upper-case needles under IgnoreCase): only the dense part
        HIP_TRY(hipMemsetAsync(b->unit_offsets.p, 0, n * sizeof(uint64_t), st));
        AM_TRY(build_hidx(p, b, st));
    }
    // dense pass: count per unit -> scan -> write (the unit boundaries and b->unit_offsets are those of the k_sf pass)
    AM_TRY(b->sparse.ensure(sizeof(Record)));
    AM_TRY(b->dense_counts.ensure(n * sizeof(uint32_t)));
    AM_TRY(b->dense_offsets.ensure(n * sizeof(uint64_t)));
    HIP_TRY(hipMemsetAsync((uint32_t*)b->dense_counts.p + p.n_units, 0, sizeof(uint32_t), st));
    { Prof pr("dense", st);
      HIP_TRY(launch_dense(p.ic, false, p.ac, p.bv, (const Record*)b->sparse.p, (const uint64_t*)b->unit_offsets.p, p.unit_chunks, p.n_units, (uint32_t*)b->dense_counts.p, nullptr, nullptr, st)); }
    { Prof pr("scan", st); HIP_TRY(launch_scan(b->scan_tmp.p, tmp_bytes, (const uint32_t*)b->dense_counts.p, (uint64_t*)b->dense_offsets.p, n, st)); }
    uint64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, (uint64_t*)b->dense_offsets.p + p.n_units, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *n_out = total;
    if (total == 0) return AM_OK;
    Record* d_out = nullptr;
    AM_TRY(sink_final(total, &d_out));
    { Prof pr("dense", st);
      HIP_TRY(launch_dense(p.ic, true, p.ac, p.bv, (const Record*)b->sparse.p, (const uint64_t*)b->unit_offsets.p, p.unit_chunks, p.n_units, nullptr, (const uint64_t*)b->dense_offsets.p, d_out, st)); }
    HIP_TRY(hipStreamSynchronize(st));
    return AM_OK;
}

// Small batches (the one-document call): the whole chain -- clears + haystack index, scan, unit offsets, k_permute into a record array of
// the worst-case size (a record per byte) -- and the copies of the count and of the first records are enqueued at once, so the call has
// ONE stream synchronisation.  (The general path needs the count on the host before it sizes the record array: two, and a third when the
// caller reads the records.)  *done = false: not taken, or the record pool overflowed -- the general path runs.
constexpr uint64_t kSmallRunBytes = 64u << 10;
constexpr uint64_t kSmallRunEager = 256;                  // records that travel with the count (2048 of them: 10 us slower on a 10-KB document with 1 187 matches than a second copy)

static int run_records_small(const am_automaton* a, int case_mode, am_batch* b, am_matches* m, bool* done)
{
    *done = false;
    if (b->total == 0 || b->total > kSmallRunBytes) return AM_OK;
    Plan p; AM_TRY(make_plan(a, case_mode, b, p));
    if (p.nothing || p.dense || !p.use_sf) return AM_OK;
    if (cfg::on(cfg::kNoSmallRun)) return AM_OK;                             // A/B
    std::lock_guard<std::mutex> lk(b->mu);
    ON_DEVICE(b->dev);
    hipStream_t st; AM_TRY(get_stream(b->dev, &st));
    const uint64_t n = p.n_units + 1;
    AM_TRY(b->unit_counts.ensure(n * sizeof(uint32_t)));
    AM_TRY(b->unit_offsets.ensure(n * sizeof(uint64_t)));
    AM_TRY(b->small.ensure(64));
    AM_TRY(b->unit_first.ensure(2 * p.n_units * sizeof(uint32_t)));
    size_t tmp_bytes = 0;
    if (scan_temp_bytes(n, &tmp_bytes) != hipSuccess) return fail(AM_ERR_HIP, "scan sizing failed");
    AM_TRY(b->scan_tmp.ensure(tmp_bytes + 16));
    uint64_t want_blocks = b->total / (128 * kPoolBlock) + p.n_units + 1024 + pool_grant_slack(p.n_cu, p.n_units, sf_lds_bytes(p.sf) <= 80 * 1024);
    if (b->pool.cap / (kPoolBlock * sizeof(Record)) > want_blocks) want_blocks = b->pool.cap / (kPoolBlock * sizeof(Record));
    if (cfg::get(cfg::kSfPoolBlocks) > 0) return AM_OK;                      // (tests of the overflow / retry path: the general path has it)
    AM_TRY(b->pool.ensure(want_blocks * kPoolBlock * sizeof(Record)));
    AM_TRY(b->block_next.ensure(want_blocks * sizeof(uint32_t)));
    const size_t need = (size_t)b->total * sizeof(Record);
    size_t cap_bytes = 0;
    Record* d_records = (Record*)g_record_cache[b->dev].take(need, &cap_bytes);
    if (!d_records) {
        cap_bytes = need + need / 16;
        hipError_t e = hipMalloc((void**)&d_records, cap_bytes);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? AM_ERR_OOM : AM_ERR_HIP, std::string("hipMalloc(records): ") + hipGetErrorString(e));
    }
    auto body = [&]() -> int {
        ScanOut o{};
        o.unit_chunks = p.unit_chunks;
        o.unit_counts = (uint32_t*)b->unit_counts.p;
        o.unit_first = (uint32_t*)b->unit_first.p;
        o.unit_slots = (uint32_t*)b->unit_first.p + p.n_units;
        o.pool = (Record*)b->pool.p;
        o.block_next = (uint32_t*)b->block_next.p;
        o.pool_ctrl = (uint32_t*)b->small.p + 4;
        o.n_blocks = (uint32_t)want_blocks;
        AM_TRY(build_hidx_and_clear(p, b, st, b->small.p, 64, (uint32_t*)b->unit_counts.p + p.n_units, sizeof(uint32_t)));
        AM_TRY(launch_scan_kernel(p, kModeEmit, o, st));
        { Prof pr("scan", st); HIP_TRY(launch_scan(b->scan_tmp.p, tmp_bytes, (const uint32_t*)b->unit_counts.p, (uint64_t*)b->unit_offsets.p, n, st)); }
        { Prof pr("permute", st); HIP_TRY(launch_permute(o, (const uint64_t*)b->unit_offsets.p, d_records, p.n_units, st)); }
        uint64_t total = 0; uint32_t ctrl[2] = {0, 0};
        const uint64_t eager = b->total < kSmallRunEager ? b->total : kSmallRunEager;
        m->host.resize(eager);
        ResultCopies rc;
        AM_TRY(rc.add(&total, (uint64_t*)b->unit_offsets.p + p.n_units, 8, st));
        AM_TRY(rc.add(ctrl, o.pool_ctrl, 8, st));
        AM_TRY(rc.add(m->host.data(), d_records, eager * sizeof(Record), st));
        AM_TRY(rc.finish(st));
        if (ctrl[1]) return AM_OK;                            // record pool too small (cannot happen with this guess on <= 64 KiB, but the general path knows what to do)
        m->n = total;
        if (total <= eager) { m->host.resize(total); m->fetched = true; }
        else { m->host.clear(); m->fetched = false; }
        *done = true;
        return AM_OK;
    };
    const int rc = body();
    if (rc == AM_OK && *done && m->n) { m->d_records = d_records; m->cap_bytes = cap_bytes; }
    else g_record_cache[b->dev].give(d_records, cap_bytes);
    return rc;
}

static int run_batch_impl(const am_automaton* a, int case_mode, const am_batch* cb, am_matches** out, bool allow_small)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!cb) return fail(AM_ERR_INVALID, "null batch");
    am_matches* m = new am_matches();
    m->dev = cb->dev;
    if (allow_small) {
        bool done = false;
        const int rc = run_records_small(a, case_mode, const_cast<am_batch*>(cb), m, &done);
        if (rc != AM_OK) { am_matches_free(m); return rc; }
        if (done) { *out = m; return AM_OK; }
        m->host.clear(); m->fetched = false; m->n = 0;
    }
    auto sink = [&](uint64_t total, Record** ptr) -> int {
        const size_t need = total * sizeof(Record);
        m->d_records = (Record*)g_record_cache[m->dev].take(need, &m->cap_bytes);
        if (!m->d_records) {
            m->cap_bytes = need + need / 16;
            hipError_t e = hipMalloc((void**)&m->d_records, m->cap_bytes);
            if (e != hipSuccess) { m->d_records = nullptr; return fail(e == hipErrorOutOfMemory ? AM_ERR_OOM : AM_ERR_HIP, std::string("hipMalloc(records): ") + hipGetErrorString(e)); }
        }
        *ptr = m->d_records;
        return AM_OK;
    };
    const int rc = run_records(a, case_mode, const_cast<am_batch*>(cb), sink, &m->n);
    if (rc != AM_OK) { am_matches_free(m); return rc; }
    *out = m;
    return AM_OK;
}

extern "C" int am_run_batch(const am_automaton* a, int case_mode, const am_batch* cb, am_matches** out) { return run_batch_impl(a, case_mode, cb, out, true); }

// ------------------------------------------------------------------ one-shot host entry points

extern "C" int am_count(const am_automaton* a, int case_mode, const am_slice* hay, size_t n_hay, uint64_t* counts_out)
{
    if (n_hay && !counts_out) return fail(AM_ERR_INVALID, "counts_out is null");
    if (!a) return fail(AM_ERR_INVALID, "null automaton");
    am_batch* b = oneshot_get(a->dev);                    // this thread's batch on the automaton's device
    int rc = upload_slices(hay, n_hay, b, true);
    if (rc == AM_OK) rc = am_count_batch(a, case_mode, b, counts_out, nullptr);
    oneshot_trim(a->dev);
    return rc;
}

extern "C" int am_contains_any(const am_automaton* a, int case_mode, const am_slice* hay, size_t n_hay, uint8_t* flags_out)
{
    if (!a) return fail(AM_ERR_INVALID, "null automaton");
    am_batch* b = oneshot_get(a->dev);                    // this thread's batch on the automaton's device
    int rc = upload_slices(hay, n_hay, b, true);
    if (rc == AM_OK) rc = am_contains_any_batch(a, case_mode, b, flags_out);
    oneshot_trim(a->dev);
    return rc;
}

extern "C" int am_run(const am_automaton* a, int case_mode, const am_slice* hay, size_t n_hay, am_matches** out)
{
    if (!a) return fail(AM_ERR_INVALID, "null automaton");
    am_batch* b = oneshot_get(a->dev);                    // this thread's batch on the automaton's device
    int rc = upload_slices(hay, n_hay, b, true);
    if (rc == AM_OK) rc = am_run_batch(a, case_mode, b, out);
    oneshot_trim(a->dev);
    return rc;
}

// ---- ONE haystack in ranges ("a single huge haystack splits into G ranges with maxNeedleCodePoints overlap" -- the same rule as the
// chunking inside the kernels).  Whether a needle ends at a position depends only on the bytes of one maximal match before it, so scanning
// text[start, scan_hi) with start = lo - overlap reports exactly the reference's matches with end positions in (lo, hi]; overlap = 4 bytes
// per code point of the longest needle (under IgnoreCase a haystack code point may be longer than the needle code point it lowers to:
// KELVIN SIGN, 3 bytes, lowers to k); start is moved back and scan_hi forward to a code point boundary (a slice that ended inside a code
// point would hand the general kernel a truncated sequence).  The own range is cut out of the sorted records ON THE DEVICE (two binary
// searches) and rebased to the whole haystack.
static int range_window(const am_automaton* a, int case_mode, const am_slice* hay, uint64_t lo, uint64_t hi, uint64_t* start, uint64_t* scan_hi)
{
    if (!a || !hay || (hay->len && !hay->ptr)) return fail(AM_ERR_INVALID, "null arguments");
    if (lo > hi || hi > hay->len) return fail(AM_ERR_INVALID, "range outside the haystack");
    const Flavor* f; AM_TRY(prepare(a, case_mode, &f));
    const uint64_t overlap = 4ull * (f->h.max_needle_cps ? f->h.max_needle_cps : 1u);
    const uint8_t* t = hay->ptr + hay->off;
    uint64_t s = lo > overlap ? lo - overlap : 0;
    while (s > 0 && (t[s] & 0xC0u) == 0x80u) s--;
    uint64_t e = hi;
    while (e < hay->len && (t[e] & 0xC0u) == 0x80u) e++;
    *start = s; *scan_hi = e;
    return AM_OK;
}

extern "C" int am_run_range(const am_automaton* a, int case_mode, const am_slice* hay, uint64_t lo, uint64_t hi, am_matches** out)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    uint64_t start = 0, scan_hi = 0;
    AM_TRY(range_window(a, case_mode, hay, lo, hi, &start, &scan_hi));
    if (hi == lo) { am_matches* m = new am_matches(); m->dev = a->dev; m->fetched = true; *out = m; return AM_OK; }
    const am_slice win{hay->ptr, hay->off + start, scan_hi - start};
    am_batch* b = oneshot_get(a->dev);
    int rc = upload_slices(&win, 1, b, true);
    am_matches* m = nullptr;
    if (rc == AM_OK) rc = run_batch_impl(a, case_mode, b, &m, false);          // (the general path: the records stay on the device, unfetched)
    oneshot_trim(a->dev);
    if (rc != AM_OK) return rc;
    if (m->n) {
        OnDevice od(m->dev);
        if (od.rc != AM_OK) { am_matches_free(m); return od.rc; }
        hipStream_t st;
        rc = get_stream(m->dev, &st);
        uint64_t* d_b = nullptr;
        if (rc == AM_OK && hipMalloc((void**)&d_b, 16) != hipSuccess) rc = fail(AM_ERR_OOM, "hipMalloc failed");
        uint64_t bounds[2] = {0, 0};
        if (rc == AM_OK) {
            hipError_t e = launch_range_bounds(m->d_records, m->n, lo - start, hi - start, d_b, st);
            if (e == hipSuccess) e = hipMemcpyAsync(bounds, d_b, 16, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e == hipSuccess) e = launch_range_rebase(m->d_records + bounds[0], bounds[1] - bounds[0], start, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) rc = fail(AM_ERR_HIP, std::string("am_run_range: ") + hipGetErrorString(e));
        }
        if (d_b) (void)hipFree(d_b);
        if (rc != AM_OK) { am_matches_free(m); return rc; }
        m->first = bounds[0]; m->n = bounds[1] - bounds[0];
    }
    *out = m;
    return AM_OK;
}

// countMatches (benchmark/haskell/app/Main.hs:67-76) over the end positions in (lo, hi] of ONE haystack
extern "C" int am_count_range(const am_automaton* a, int case_mode, const am_slice* hay, uint64_t lo, uint64_t hi, uint64_t* count_out)
{
    if (!count_out) return fail(AM_ERR_INVALID, "count_out is null");
    *count_out = 0;
    am_matches* m = nullptr;
    AM_TRY(am_run_range(a, case_mode, hay, lo, hi, &m));
    int rc = AM_OK;
    if (m->n) {
        const Flavor* f = nullptr;
        rc = prepare(a, case_mode, &f);
        OnDevice od(m->dev);
        if (rc == AM_OK) rc = od.rc;
        hipStream_t st;
        if (rc == AM_OK) rc = get_stream(m->dev, &st);
        uint64_t* d_t = nullptr;
        if (rc == AM_OK && hipMalloc((void**)&d_t, 8) != hipSuccess) rc = fail(AM_ERR_OOM, "hipMalloc failed");
        if (rc == AM_OK) {
            const AcView ac = make_ac_view(f->d_image, f->h);
            hipError_t e = hipMemsetAsync(d_t, 0, 8, st);
            if (e == hipSuccess) e = launch_records_reduce(m->d_records + m->first, m->n, ac.vlen, nullptr, d_t, nullptr, st);
            if (e == hipSuccess) e = hipMemcpyAsync(count_out, d_t, 8, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) rc = fail(AM_ERR_HIP, std::string("am_count_range: ") + hipGetErrorString(e));
        }
        if (d_t) (void)hipFree(d_t);
    }
    am_matches_free(m);
    return rc;
}

// ------------------------------------------------------------------ results

// one host block of a freed large result is kept for the next one (up to 256 MiB; larger ones go back to the allocator)
struct HostCache {
    std::mutex mu; void* p = nullptr; size_t cap = 0;
    void* take(size_t need, size_t* cap_out)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (p && cap >= need && cap <= 2 * need + (1u << 20)) { void* r = p; *cap_out = cap; p = nullptr; cap = 0; return r; }
        return nullptr;
    }
    void give(void* q, size_t c)
    {
        if (c > ((size_t)256 << 20)) { std::free(q); return; }
        void* old = nullptr;
        { std::lock_guard<std::mutex> lk(mu); old = p; p = q; cap = c; }
        std::free(old);
    }
    ~HostCache() { std::free(p); }
};
static HostCache g_host_cache;

// Large results go into PAGE-LOCKED host blocks of the library's own (the caller only ever sees the pointer am_matches_data returns): the
// records are DMA'd straight into them, no staging copy -- a match-dense result is several times the size of the text that produced it
// (natural language: 2.5 x) and used to crawl through two 8-MiB staging halves and a single-threaded memcpy (2 GiB of text: 926 ms,
// round 3).  Page-locking is slow (~1 GiB/s) and page-locked memory is a limited resource, so one freed block is kept for the next result
// (up to kPinnedKeep) and larger ones are given back at once; if the runtime refuses a block the pageable path below still works.
struct PinnedCache {
    std::mutex mu; void* p = nullptr; size_t cap = 0;
    void* take(size_t need, size_t* cap_out)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (p && cap >= need) { void* r = p; *cap_out = cap; p = nullptr; cap = 0; return r; }
        return nullptr;
    }
    void give(void* q, size_t c, size_t keep_limit)
    {
        void* old = q;
        if (c <= keep_limit) { std::lock_guard<std::mutex> lk(mu); if (c > cap) { old = p; p = q; cap = c; } }
        if (old) (void)hipHostFree(old);
    }
};
static PinnedCache& pinned_cache() { static PinnedCache* c = new PinnedCache(); return *c; }      // never destroyed: no HIP call in a static destructor
constexpr size_t kPinnedKeep = (size_t)8 << 30;

constexpr size_t kRecordsDirect = 1u << 20;              // results up to this size: one plain copy
constexpr size_t kFetchPiece = 8u << 20;

// device -> pageable host memory through the calling thread's pinned staging area (two halves that take turns)
static int fetch_through_pinned(void* dst, const void* d_src, size_t bytes, int dev)
{
    hipStream_t st; AM_TRY(get_stream(dev, &st));
    AM_TRY(pin_ensure(tl_state.pin, tl_state.pin_cap, 2 * kFetchPiece));
    AM_TRY(pin_events(dev));
    // pieces of an eighth of the result (256 KiB .. 8 MiB): a result of a few MiB still overlaps its copy-out with the transfer
    size_t piece = (bytes / 8 + 4095) & ~(size_t)4095;
    if (piece < ((size_t)256 << 10)) piece = (size_t)256 << 10;
    if (piece > kFetchPiece) piece = kFetchPiece;
    const size_t n_pieces = (bytes + piece - 1) / piece;
    auto issue = [&](size_t i) -> int {
        const size_t lo = i * piece, len = std::min(piece, bytes - lo);
        HIP_TRY(hipMemcpyAsync(tl_state.pin + (i & 1) * kFetchPiece, (const uint8_t*)d_src + lo, len, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(tl_state.pin_ev[i & 1], st));
        return AM_OK;
    };
    auto take = [&](size_t i) -> int {
        const size_t lo = i * piece, len = std::min(piece, bytes - lo);
        HIP_TRY(hipEventSynchronize(tl_state.pin_ev[i & 1]));
        std::memcpy((uint8_t*)dst + lo, tl_state.pin + (i & 1) * kFetchPiece, len);
        return AM_OK;
    };
    AM_TRY(issue(0));
    for (size_t i = 1; i < n_pieces; i++) { AM_TRY(issue(i)); AM_TRY(take(i - 1)); }
    AM_TRY(take(n_pieces - 1));
    return AM_OK;
}

extern "C" uint64_t am_matches_size(const am_matches* m) { return m ? m->n : 0; }

extern "C" const am_match* am_matches_data(am_matches* m)
{
    if (!m) return nullptr;
    if (!m->fetched) {
        const size_t bytes = (size_t)m->n * sizeof(Record);
        OnDevice od(m->dev);
        if (bytes <= kRecordsDirect) {
            m->host.resize(m->n);
            if (m->n) {
                hipError_t e = hipMemcpy(m->host.data(), m->d_records + m->first, bytes, hipMemcpyDeviceToHost);
                if (e != hipSuccess) { fail(AM_ERR_HIP, std::string("hipMemcpy(records): ") + hipGetErrorString(e)); return nullptr; }
            }
        } else {
            // a large result: no zero-filled vector and no staged copy into pageable memory inside the runtime -- the records cross PCIe in
            // pieces into the calling thread's pinned staging area, and piece i is copied out while piece i + 1 is on its way
            if (bytes > kFetchPiece) {
                // page-locked block + one DMA (in 256-MiB requests, so that a huge result does not sit in one multi-second call of the runtime)
                m->big = (am_match*)pinned_cache().take(bytes, &m->big_cap);
                if (!m->big) {
                    void* q = nullptr;
                    const size_t want = bytes + bytes / 16 + 4096;
                    if (hipHostMalloc(&q, want, hipHostMallocPortable) == hipSuccess) { m->big = (am_match*)q; m->big_cap = want; }
                    else (void)hipGetLastError();
                }
                if (m->big) {
                    m->big_pinned = true;
                    hipStream_t st;
                    bool good = get_stream(m->dev, &st) == AM_OK;
                    constexpr size_t kReq = (size_t)256 << 20;
                    for (size_t lo = 0; good && lo < bytes; lo += kReq)
                        good = hipMemcpyAsync((uint8_t*)m->big + lo, (const uint8_t*)(m->d_records + m->first) + lo, std::min(kReq, bytes - lo), hipMemcpyDeviceToHost, st) == hipSuccess;
                    if (good) good = hipStreamSynchronize(st) == hipSuccess;
                    if (!good) { fail(AM_ERR_HIP, "copying the match records to the host failed"); (void)hipHostFree(m->big); m->big = nullptr; return nullptr; }
                    m->fetched = true;
                    return m->big;
                }
            }
            m->big = (am_match*)g_host_cache.take(bytes, &m->big_cap);      // (a block used before has its pages: a fresh 100-MB block costs 10 ms of page faults)
            if (!m->big) { m->big_cap = bytes + bytes / 16; m->big = (am_match*)std::malloc(m->big_cap); }
            if (!m->big) { fail(AM_ERR_OOM, "out of host memory for the match records"); return nullptr; }
            if (bytes <= kFetchPiece) {                       // a few MiB: the runtime's own staged copy is faster than two pieces of ours (1.8 MB: 290 against 410 us per am_run)
                hipError_t e = hipMemcpy(m->big, m->d_records + m->first, bytes, hipMemcpyDeviceToHost);
                if (e != hipSuccess) { fail(AM_ERR_HIP, std::string("hipMemcpy(records): ") + hipGetErrorString(e)); std::free(m->big); m->big = nullptr; return nullptr; }
            } else if (fetch_through_pinned(m->big, m->d_records + m->first, bytes, m->dev) != AM_OK) { std::free(m->big); m->big = nullptr; return nullptr; }
        }
        m->fetched = true;
    }
    return m->big ? m->big : m->host.data();
}

extern "C" const void* am_matches_device_data(const am_matches* m) { return m ? (m->d_records ? m->d_records + m->first : nullptr) : nullptr; }

extern "C" void am_matches_free(am_matches* m)
{
    if (!m) return;
    if (m->d_records) g_record_cache[m->dev].give(m->d_records, m->cap_bytes);
    if (m->big) { if (m->big_pinned) pinned_cache().give(m->big, m->big_cap, kPinnedKeep); else g_host_cache.give(m->big, m->big_cap); }
    delete m;
}

// ------------------------------------------------------------------ UTF-8 helpers

extern "C" uint32_t am_lower_code_point(uint32_t cp) { return cp < 128 ? fold_byte(cp) : simple_lower(cp); }
extern "C" uint32_t am_unicode_version(void) { return kUnicodeLowerVersion; }
extern "C" uint32_t am_image_version(void) { return kImageVersion; }

extern "C" size_t am_unlower_code_point(uint32_t cp, uint32_t* out, size_t cap)
{
    std::vector<uint32_t> v;
    unlower(cp, v);
    for (size_t i = 0; i < v.size() && i < cap; i++) out[i] = v[i];
    return v.size();
}

// ------------------------------------------------------------------ runtime knobs

extern "C" int am_set_stream(void* hip_stream)
{
    tl_state.user = (hipStream_t)hip_stream;          // per calling thread
    tl_state.use_user = hip_stream != nullptr;
    return AM_OK;
}

extern "C" int am_get_stream(void** hip_stream)
{
    if (!hip_stream) return fail(AM_ERR_INVALID, "hip_stream is null");
    AM_TRY(ensure_runtime());
    int dev = 0;
    AM_TRY(current_device(&dev));
    ON_DEVICE(dev);
    hipStream_t st; AM_TRY(get_stream(dev, &st));
    *hip_stream = (void*)st;
    return AM_OK;
}

extern "C" int am_device_info(int* n_cu, size_t* hbm_bytes, char* name, size_t name_cap)
{
    int dev = 0;
    AM_TRY(current_device(&dev));
    if (n_cu) *n_cu = g_rt.dev[dev].n_cu;
    if (hbm_bytes) *hbm_bytes = g_rt.dev[dev].hbm;
    if (name && name_cap) { std::strncpy(name, g_rt.dev[dev].name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    return AM_OK;
}

// debug only (not declared in am.h): cycle sums per k_sf phase for launches made under AM_SF_ABLATE=9
extern "C" int am_debug_sf_phase_cycles(uint64_t* out5)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(read_sf_phase_cycles(out5));
    return AM_OK;
}

// debug only: a test / measurement switch of am_config.h by the name of its environment variable ("AM_RP_FULL_SCANS", ...); value -1 = unset
extern "C" int am_debug_set(const char* name, long value)
{
    if (!name || !cfg::set(name, value)) return fail(AM_ERR_INVALID, "no such switch");
    return AM_OK;
}

extern "C" uint64_t am_debug_pinned_bytes(void) { return (uint64_t)g_pinned_staging_bytes.load(std::memory_order_relaxed); }

extern "C" int am_debug_sfx_roles(uint64_t* out24)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(read_sfx_roles(out24));
    return AM_OK;
}

extern "C" int am_debug_sf_wave_records(uint64_t* out, size_t n_waves)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(read_sf_wave_records(out, n_waves));
    return AM_OK;
}

extern "C" int am_profile_enable(int on) { g_rt.prof_on.store(on != 0); return AM_OK; }

static void drain_profile_locked()
{
    for (auto& p : g_rt.pending) {
        float ms = 0;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto& acc = g_rt.prof[p.k]; acc.first += ms; acc.second += 1;
        }
        (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b);
    }
    g_rt.pending.clear();
}

extern "C" int am_profile_reset(void)
{
    std::lock_guard<std::mutex> lk(g_rt.mu);
    drain_profile_locked();
    g_rt.prof.clear();
    return AM_OK;
}

extern "C" int am_profile_read(const char* kernel, double* total_ms, uint64_t* launches)
{
    if (!kernel) return fail(AM_ERR_INVALID, "kernel is null");
    std::lock_guard<std::mutex> lk(g_rt.mu);
    drain_profile_locked();
    auto it = g_rt.prof.find(kernel);
    if (total_ms) *total_ms = it == g_rt.prof.end() ? 0.0 : it->second.first;
    if (launches) *launches = it == g_rt.prof.end() ? 0 : it->second.second;
    return AM_OK;
}

// ------------------------------------------------------------------ Replacer (Replacer.hs:97-274), device-resident passes

static_assert(sizeof(am_payload) == sizeof(RpPayload) && offsetof(am_payload, repl_off) == offsetof(RpPayload, repl_off) &&
                  offsetof(am_payload, len_code_points) == offsetof(RpPayload, len_code_points) && offsetof(am_payload, repl_len) == offsetof(RpPayload, repl_len),
              "am_payload must mirror the device payload");

struct am_replacer {
    const am_automaton* a = nullptr;
    int case_mode = 0;
    DevBuf vals_off, vals, payloads, repl, one;
    RpTables t{};
    uint32_t max_repl_len = 0;                        // longest replacement (bounds the re-scan window of the one-kernel loop)
    // the workspace of the last run (device buffers, pinned scratch, copy stream) is kept for the next one: a caller that
    // rewrites one document per call would otherwise pay ~40 hipMalloc/hipFree (4 ms) each time
    mutable std::mutex session_mu;
    mutable std::vector<void*> sessions;              // workspaces of finished runs, kept for the next ones (several: concurrent groups / threads)
    void (*session_delete)(void*) = nullptr;
};

// Finished texts are copied D2H straight into pinned slabs that the result object keeps (no second host
// copy); am_replaced_free hands the slabs back to a small process-wide pool so that repeated calls do
// not pay for pinning again.
namespace {
struct Slab { uint8_t* p = nullptr; size_t cap = 0, used = 0; };
struct SlabPool {
    std::mutex mu;
    std::vector<Slab> free_list;
    bool device = false;           // slabs in the current device's HBM (results that stay on the device) instead of pinned host memory
    static constexpr size_t kSlab = 256ull << 20, kKeep = 8;
    int take(size_t need, Slab* out)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < free_list.size(); i++)
                if (free_list[i].cap >= need) { *out = free_list[i]; out->used = 0; free_list.erase(free_list.begin() + i); return AM_OK; }
        }
        Slab s; s.cap = need > kSlab ? need : kSlab;
        if (device) { if (hipMalloc((void**)&s.p, s.cap) != hipSuccess) return fail(AM_ERR_OOM, "hipMalloc(result slab) failed"); }
        else if (hipHostMalloc((void**)&s.p, s.cap, hipHostMallocPortable) != hipSuccess) return fail(AM_ERR_OOM, "hipHostMalloc(result slab) failed");
        *out = s;
        return AM_OK;
    }
    void give(const Slab& s)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (free_list.size() < kKeep) { free_list.push_back(s); return; }
        }
        if (device) (void)hipFree(s.p); else (void)hipHostFree(s.p);
    }
};
SlabPool g_slabs;
struct DevSlabPools { SlabPool p[kMaxDev]; DevSlabPools() { for (SlabPool& x : p) x.device = true; } } g_dev_slabs;
}  // namespace

struct am_replaced {
    struct Item { const uint8_t* p = nullptr; size_t len = 0; };
    std::vector<Item> text;
    std::vector<uint8_t> just;
    std::vector<Slab> slabs;
    uint64_t passes = 0, scanned = 0, spliced = 0;
    int dev = -1;                  // >= 0: the texts stay in that device's memory (am_replacer_run_batch_device)
    SlabPool& pool() const { return dev >= 0 ? g_dev_slabs.p[dev] : g_slabs; }
    ~am_replaced() { for (const Slab& s : slabs) pool().give(s); }
    // room for n contiguous bytes in the current slab, or a new slab
    int room(size_t n, uint8_t** out)
    {
        if (slabs.empty() || slabs.back().cap - slabs.back().used < n) { Slab s; AM_TRY(pool().take(n, &s)); slabs.push_back(s); }
        *out = slabs.back().p + slabs.back().used;
        slabs.back().used += (n + 63) & ~(size_t)63;
        if (slabs.back().used > slabs.back().cap) slabs.back().used = slabs.back().cap;
        return AM_OK;
    }
};

extern "C" int am_replacer_create(const am_automaton* a, int case_mode, const uint64_t* values_offsets, const uint32_t* values,
                                  const am_payload* payloads, size_t n_payloads, const uint8_t* repl_bytes, size_t n_repl_bytes,
                                  int64_t min_priority, am_replacer** out)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    const Flavor* f = nullptr;
    AM_TRY(prepare(a, case_mode, &f));
    ON_DEVICE(a->dev);
    const uint64_t n_states = f->h.n_states;
    if (!values_offsets || values_offsets[0] != 0) return fail(AM_ERR_INVALID, "values_offsets[0] must be 0");
    uint32_t max_repl = 0;
    const uint64_t n_values = values_offsets[n_states];
    if ((n_values && !values) || (n_payloads && !payloads) || (n_repl_bytes && !repl_bytes)) return fail(AM_ERR_INVALID, "null table");
    for (uint64_t s = 0; s < n_states; s++) {
        if (values_offsets[s + 1] < values_offsets[s]) return fail(AM_ERR_INVALID, "values_offsets must be non-decreasing");
        if (a->has_ref && values_offsets[s + 1] - values_offsets[s] != a->values_len[s])
            return fail(AM_ERR_INVALID, "values_offsets disagrees with the values_len given to am_automaton_create");
    }
    for (uint64_t k = 0; k < n_values; k++) if (values[k] >= n_payloads) return fail(AM_ERR_INVALID, "payload index out of range");
    {
        // Replacer.hs:100-104 / :127-131: priorities are 0, -1, -2, ...; the device pass relies on them being distinct
        std::vector<int64_t> pr(n_payloads);
        for (size_t i = 0; i < n_payloads; i++) {
            pr[i] = payloads[i].priority;
            if (payloads[i].repl_len > max_repl) max_repl = payloads[i].repl_len;
            if (pr[i] > 0) return fail(AM_ERR_INVALID, "priorities must be <= 0 (the initial threshold is 1, Replacer.hs:211)");
            if ((uint64_t)payloads[i].repl_off + payloads[i].repl_len > n_repl_bytes) return fail(AM_ERR_INVALID, "replacement slice out of range");
            if (case_mode == AM_IGNORE_CASE && payloads[i].len_code_points == 0)
                return fail(AM_ERR_UNSUPPORTED, "empty needle under IgnoreCase: the reference's skipCodePointsBackwards has no answer (Utf8.hs:259)");
        }
        std::sort(pr.begin(), pr.end());
        for (size_t i = 1; i < n_payloads; i++) if (pr[i] == pr[i - 1]) return fail(AM_ERR_INVALID, "payload priorities must be distinct");
    }
    am_replacer* r = new am_replacer();
    r->a = a; r->case_mode = case_mode; r->max_repl_len = max_repl;
    auto up = [&](DevBuf& d, const void* src, size_t bytes) -> int {
        AM_TRY(d.ensure(bytes + 64));
        if (bytes) HIP_TRY(hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice));
        return AM_OK;
    };
    int rc = up(r->vals_off, values_offsets, (n_states + 1) * sizeof(uint64_t));
    if (rc == AM_OK) rc = up(r->vals, values, n_values * sizeof(uint32_t));
    if (rc == AM_OK) rc = up(r->payloads, payloads, n_payloads * sizeof(am_payload));
    if (rc == AM_OK && n_payloads == 0) { hipError_t e = hipMemset(r->payloads.p, 0, sizeof(am_payload)); if (e != hipSuccess) rc = fail(AM_ERR_HIP, hipGetErrorString(e)); }
    if (rc == AM_OK) rc = up(r->repl, repl_bytes, n_repl_bytes);
    if (rc == AM_OK) {
        std::vector<RpStateOne> one(n_states);
        for (uint64_t s = 0; s < n_states; s++) {
            const uint64_t n = values_offsets[s + 1] - values_offsets[s];
            RpStateOne e{0, 0, (uint32_t)(n > 0xFFFFFFFFull ? 0xFFFFFFFFull : n), 0, 0, 0, 0};
            if (n == 1) {
                const am_payload& pl = payloads[values[values_offsets[s]]];
                e.priority = pl.priority; e.payload = values[values_offsets[s]]; e.len_bytes = pl.len_bytes; e.repl_len = pl.repl_len; e.len_code_points = pl.len_code_points;
            }
            one[s] = e;
        }
        rc = up(r->one, one.data(), one.size() * sizeof(RpStateOne));
    }
    if (rc != AM_OK) { am_replacer_destroy(r); return rc; }
    r->t = RpTables{(const uint64_t*)r->vals_off.p, (const uint32_t*)r->vals.p, (const RpPayload*)r->payloads.p, (const uint8_t*)r->repl.p, min_priority, (const RpStateOne*)r->one.p};
    *out = r;
    return AM_OK;
}

extern "C" void am_replacer_destroy(am_replacer* r)
{
    if (!r) return;
    if (r->session_delete) for (void* p : r->sessions) r->session_delete(p);
    for (DevBuf* d : {&r->vals_off, &r->vals, &r->payloads, &r->repl, &r->one}) d->release();
    delete r;
}

namespace {

struct RpSession {
    DevBuf text[2], offs[2], orig[2], thr[2];
    DevBuf totals; uint64_t* tot_host = nullptr; uint64_t tot_seq = 0;       // the per-pass totals, read back through pinned memory (tot_host[15]: sequence number of the last pass written)
    hipStream_t copy_stream = nullptr; hipEvent_t ev_spliced = nullptr;     // finished texts travel home next to the window scans
    RpFin* fin_host = nullptr; size_t fin_host_cap = 0;                     // pinned
    int pin_meta(size_t bytes)
    {
        if (bytes <= fin_host_cap) return AM_OK;
        if (fin_host) (void)hipHostFree(fin_host);
        fin_host = nullptr; fin_host_cap = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        if (hipHostMalloc((void**)&fin_host, want, hipHostMallocPortable) != hipSuccess) { fin_host = nullptr; return fail(AM_ERR_OOM, "hipHostMalloc failed"); }
        fin_host_cap = want;
        return AM_OK;
    }
    DevBuf recbuf[2];                    // sorted records of the current pass / of the next one (incremental re-scan)
    DevBuf nwin, win_off, wins, wlen, woffs, wtext, wrec, wrec_first, mcount, moff, tile_hay;
    am_batch ws2;                        // workspace of the window scans
    DevBuf rec_first, rec_first2, kept, hs, len_next, len_fin, tiles, act, fin, off_next, off_fin, tile_off, act_idx, fin_idx, scan_tmp, fin_text, fin_meta;      // (rec_first2: the piece-table loop's second ranges buffer -- a pass's merge writes the next pass's ranges)
    am_batch ws;                         // workspace holder for the scans; never owns its text
    DevBuf first_orig, first_thr;
    DevBuf pt_pieces[2], pt_start[2], pt_cnt[2], pt_need, pt_need_off, pt_fin_start, pt_fin_cnt;      // piece-table path
    DevBuf lp_rec, lp_pc, lp_kept, lp_wtext, lp_out, lp_ctrl, lp_cap_r, lp_cap_p, lp_rec_base, lp_pc_base, lp_fin, lp_fin_start, lp_fin_cnt;      // one-kernel loop (am_rploop.hip)
    void* lp_host = nullptr; size_t lp_host_cap = 0;                        // pinned: the loop's per-haystack results, then the materialise tables
    int pin_loop(size_t bytes)
    {
        if (bytes <= lp_host_cap) return AM_OK;
        if (lp_host) (void)hipHostFree(lp_host);
        lp_host = nullptr; lp_host_cap = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        if (hipHostMalloc(&lp_host, want, hipHostMallocPortable) != hipSuccess) { lp_host = nullptr; return fail(AM_ERR_OOM, "hipHostMalloc failed"); }
        lp_host_cap = want;
        return AM_OK;
    }
    DevBuf pf_best, pf_delta, pf_payload, pf_selflag, pf_sidx, pf_cand, pf_sel, pf_keep, pf_kflag, pf_kdelta, pf_kidx, pf_kdpre, pf_tmp;   // record-parallel fold
    size_t device_bytes() const
    {
        size_t n = 0;
        for (const DevBuf* d : {&text[0], &text[1], &recbuf[0], &recbuf[1], &kept, &wins, &wtext, &wrec, &fin_text, &ws.pool, &ws2.pool, &ws.hidx, &ws2.hidx, &pf_cand, &pf_sel, &pf_sidx,
                                &lp_rec, &lp_pc, &lp_kept, &lp_wtext}) n += d->cap;
        return n;
    }
    ~RpSession()
    {
        for (DevBuf* d : {&text[0], &text[1], &offs[0], &offs[1], &orig[0], &orig[1], &thr[0], &thr[1], &rec_first, &kept, &hs, &len_next, &len_fin,
                          &recbuf[0], &recbuf[1], &nwin, &win_off, &wins, &wlen, &woffs, &wtext, &wrec, &wrec_first, &mcount, &moff, &tile_hay,
                          &totals, &tiles, &act, &fin, &off_next, &off_fin, &tile_off, &act_idx, &fin_idx, &scan_tmp, &fin_text, &fin_meta, &first_orig, &first_thr,
                          &pf_best, &pf_delta, &pf_payload, &pf_selflag, &pf_sidx, &pf_cand, &pf_sel, &pf_keep, &pf_kflag, &pf_kdelta, &pf_kidx, &pf_kdpre, &pf_tmp,
                          &pt_pieces[0], &pt_pieces[1], &pt_start[0], &pt_start[1], &pt_cnt[0], &pt_cnt[1], &pt_need, &pt_need_off, &pt_fin_start, &pt_fin_cnt,
                          &lp_rec, &lp_pc, &lp_kept, &lp_wtext, &lp_out, &lp_ctrl, &lp_cap_r, &lp_cap_p, &lp_rec_base, &lp_pc_base, &lp_fin, &lp_fin_start, &lp_fin_cnt}) d->release();
        if (lp_host) (void)hipHostFree(lp_host);
        if (tot_host) (void)hipHostFree(tot_host);
        if (fin_host) (void)hipHostFree(fin_host);
        if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); }
        if (ev_spliced) (void)hipEventDestroy(ev_spliced);
        for (am_batch* w : {&ws, &ws2})
            for (DevBuf* d : {&w->hidx, &w->unit_counts, &w->unit_offsets, &w->scan_tmp, &w->small, &w->hay_counts, &w->flags, &w->unit_first, &w->pool, &w->block_next}) d->release();
    }
};

size_t padded_text(uint64_t total) { return (size_t)((total + 15) & ~15ull) + 16; }

// The suffix-filter scan of a (small) batch WITHOUT a host round trip: the record pool is sized for the worst case -- a record at
// every byte -- so the pass cannot overflow and needs no retry; the sorted records go to d_out (room for b->total records), their
// number stays on the device (*n_dev points at it).  Used between Replacer passes, where a sync per scan would cost more than
// the scan.
static int run_records_async(const am_automaton* a, int case_mode, am_batch* b, Record* d_out, const uint64_t** n_dev, hipStream_t st)
{
    Plan p; AM_TRY(make_plan(a, case_mode, b, p));
    if (!p.use_sf || p.dense) return fail(AM_ERR_UNSUPPORTED, "internal: asynchronous scan needs the plain suffix-filter route");
    std::lock_guard<std::mutex> lk(b->mu);
    const uint64_t n = p.n_units + 1;
    AM_TRY(b->unit_counts.ensure(n * sizeof(uint32_t)));
    AM_TRY(b->unit_offsets.ensure(n * sizeof(uint64_t)));
    *n_dev = (const uint64_t*)b->unit_offsets.p + p.n_units;
    if (p.nothing) { HIP_TRY(hipMemsetAsync(b->unit_offsets.p, 0, n * sizeof(uint64_t), st)); return AM_OK; }
    AM_TRY(b->small.ensure(64));
    AM_TRY(b->unit_first.ensure(2 * p.n_units * sizeof(uint32_t)));
    size_t tmp_bytes = 0;
    if (scan_temp_bytes(n, &tmp_bytes) != hipSuccess) return fail(AM_ERR_HIP, "scan sizing failed");
    AM_TRY(b->scan_tmp.ensure(tmp_bytes + 16));
    const uint64_t want_blocks = b->total / kPoolBlock + p.n_units + 8 + pool_grant_slack(p.n_cu, p.n_units, sf_lds_bytes(p.sf) <= 80 * 1024);          // ceil(records / 64) per unit, records <= bytes; + the grants' unused remainders
    if (want_blocks >= (1ull << 26)) return fail(AM_ERR_UNSUPPORTED, "too many match records for one call (2^32 record slots); split the batch");      // k_sf addresses record slots with 32 bits
    AM_TRY(b->pool.ensure(want_blocks * kPoolBlock * sizeof(Record)));
    AM_TRY(b->block_next.ensure(want_blocks * sizeof(uint32_t)));
    ScanOut o{};
    o.unit_chunks = p.unit_chunks;
    o.unit_counts = (uint32_t*)b->unit_counts.p;
    o.unit_first = (uint32_t*)b->unit_first.p;
    o.unit_slots = (uint32_t*)b->unit_first.p + p.n_units;
    o.pool = (Record*)b->pool.p;
    o.block_next = (uint32_t*)b->block_next.p;
    o.pool_ctrl = (uint32_t*)b->small.p + 4;
    o.n_blocks = (uint32_t)want_blocks;
    AM_TRY(build_hidx_and_clear(p, b, st, b->small.p, 64, (uint32_t*)b->unit_counts.p + p.n_units, sizeof(uint32_t)));      // (the Replacer's window batches are new every pass: one launch)
    AM_TRY(launch_scan_kernel(p, kModeEmit, o, st));
    { Prof pr("scan", st);
      if (n <= (1u << 16)) {                                  // few units: the single-workgroup scan (one dispatch, no library sizing / configuration on the host)
          ScanJobs jobs{};
          jobs.j[0] = ScanJob{(const uint32_t*)b->unit_counts.p, nullptr, (uint64_t*)b->unit_offsets.p, n, nullptr};
          jobs.n_jobs = 1;
          HIP_TRY(launch_scan_jobs(jobs, st));
      } else HIP_TRY(launch_scan(b->scan_tmp.p, tmp_bytes, (const uint32_t*)b->unit_counts.p, (uint64_t*)b->unit_offsets.p, n, st)); }
    { Prof pr("permute", st); HIP_TRY(launch_permute(o, (const uint64_t*)b->unit_offsets.p, d_out, p.n_units, st)); }
    return AM_OK;
}

// prependMatch + makeMatch + removeOverlap of one pass (Replacer.hs:252-274,191-198): one wavefront per haystack, or -- few
// haystacks with very many matches each -- parallel over the records.  Writes kept[], hs[] and the route arrays.
static int rp_fold(RpSession& s, const am_replacer* r, bool ic, const uint8_t* text, const uint64_t* offs, const Record* recs, uint64_t n_rec, const int64_t* thr,
                   uint64_t max_length, const RpRoute& route, uint32_t n_act, hipStream_t st, const uint64_t* rec_first)
{
    const uint64_t n1 = (uint64_t)n_act + 1;
    // one wavefront per haystack, or -- few haystacks with very many matches each -- parallel over the records
        bool par_fold = n_rec > 2048ull * n_act;
        if (cfg::get(cfg::kRpParallelFold) != cfg::kUnset) par_fold = cfg::get(cfg::kRpParallelFold) != 0;        // tests force either path
        if (!par_fold) {
            Prof pr("rp_pass", st);
            HIP_TRY(launch_rp_pass(ic, r->t, text, offs, recs, rec_first, thr,
                                   max_length, (RpKept*)s.kept.p, (RpHay*)s.hs.p, route, n_act, 0u, st));
        } else {
            Prof pr("rp_pass", st);
            const uint64_t nb = n_rec + 2;
            AM_TRY(s.pf_best.ensure(n1 * 8)); AM_TRY(s.pf_delta.ensure(n1 * 8)); AM_TRY(s.pf_payload.ensure(n1 * 4));
            AM_TRY(s.pf_selflag.ensure(nb * 4)); AM_TRY(s.pf_sidx.ensure(nb * 8)); AM_TRY(s.pf_cand.ensure(nb * sizeof(RpSel))); AM_TRY(s.pf_sel.ensure(nb * sizeof(RpSel)));
            AM_TRY(s.pf_keep.ensure(nb * 4)); AM_TRY(s.pf_kflag.ensure(nb * 4)); AM_TRY(s.pf_kdelta.ensure(nb * 8)); AM_TRY(s.pf_kidx.ensure(nb * 8)); AM_TRY(s.pf_kdpre.ensure(nb * 8));
            size_t t32b = 0, t64b = 0;
            if (scan_temp_bytes(nb, &t32b) != hipSuccess || scan64_temp_bytes(nb, &t64b) != hipSuccess) return fail(AM_ERR_HIP, "scan sizing failed");
            AM_TRY(s.pf_tmp.ensure(std::max(t32b, t64b) + 16));
            const size_t ptmp = s.pf_tmp.cap - 16;
            HIP_TRY(hipMemsetAsync(s.pf_delta.p, 0, n1 * 8, st)); HIP_TRY(hipMemsetAsync(s.pf_payload.p, 0, n1 * 4, st));
            HIP_TRY(hipMemsetAsync(s.pf_kflag.p, 0, nb * 4, st)); HIP_TRY(hipMemsetAsync(s.pf_kdelta.p, 0, nb * 8, st)); HIP_TRY(hipMemsetAsync(s.pf_keep.p, 0, nb * 4, st));
            HIP_TRY(launch_rpp_best(r->t, recs, n_rec, thr, (int64_t*)s.pf_best.p, n_act, st));
            HIP_TRY(launch_rpp_select(ic, r->t, text, offs, recs, n_rec, (const int64_t*)s.pf_best.p,
                                      (uint32_t*)s.pf_selflag.p, (RpSel*)s.pf_cand.p, (int64_t*)s.pf_delta.p, (uint32_t*)s.pf_payload.p, st));
            HIP_TRY(launch_scan(s.pf_tmp.p, ptmp, (const uint32_t*)s.pf_selflag.p, (uint64_t*)s.pf_sidx.p, n_rec + 1, st));
            const uint64_t* n_sel_dev = (const uint64_t*)s.pf_sidx.p + n_rec;
            HIP_TRY(launch_rpp_compact((const uint32_t*)s.pf_selflag.p, (const uint64_t*)s.pf_sidx.p, (const RpSel*)s.pf_cand.p, n_rec, (RpSel*)s.pf_sel.p, st));
            HIP_TRY(launch_rpp_overlaps((const RpSel*)s.pf_sel.p, n_sel_dev, n_rec, (uint32_t*)s.pf_keep.p, st));
            HIP_TRY(launch_rpp_kflags((const RpSel*)s.pf_sel.p, n_sel_dev, n_rec, (const uint32_t*)s.pf_keep.p, r->t, (const uint32_t*)s.pf_payload.p,
                                      (uint32_t*)s.pf_kflag.p, (uint64_t*)s.pf_kdelta.p, st));
            HIP_TRY(launch_scan(s.pf_tmp.p, ptmp, (const uint32_t*)s.pf_kflag.p, (uint64_t*)s.pf_kidx.p, n_rec + 2, st));
            HIP_TRY(launch_scan64(s.pf_tmp.p, ptmp, (const uint64_t*)s.pf_kdelta.p, (uint64_t*)s.pf_kdpre.p, n_rec + 2, st));
            HIP_TRY(launch_rpp_finish(r->t, (const RpSel*)s.pf_sel.p, n_sel_dev, n_rec, (const uint32_t*)s.pf_kflag.p, (const uint64_t*)s.pf_kidx.p,
                                      (const uint64_t*)s.pf_kdpre.p, (const uint64_t*)s.pf_sidx.p, offs, rec_first, (const int64_t*)s.pf_best.p,
                                      (const int64_t*)s.pf_delta.p, (const uint32_t*)s.pf_payload.p, max_length, (RpKept*)s.kept.p, (RpHay*)s.hs.p, route, n_act, st));
        }
    return AM_OK;
}

// The same loop with the text of the active haystacks kept as PIECE TABLES (am_replace.hip): no pass rewrites a text; bytes
// move into the re-scanned windows and, once per haystack, into the result.  CaseSensitive replacers on the suffix-filter
// route (the incremental re-scan is part of the design: after the first pass only windows are scanned).
int replacer_run_pt(const am_replacer* r, const am_batch* in, uint64_t max_length, am_replaced* res, const Flavor* flavor)
{
    const uint32_t n_hay = in->n_hay;
    ON_DEVICE(in->dev);
    hipStream_t st; AM_TRY(get_stream(in->dev, &st));
    RpSession* sp = nullptr;
    { std::lock_guard<std::mutex> lk(r->session_mu); if (!r->sessions.empty()) { sp = static_cast<RpSession*>(r->sessions.back()); r->sessions.pop_back(); } }
    if (!sp) sp = new RpSession();
    struct Return {
        const am_replacer* r; RpSession* sp;
        ~Return()
        {
            if (sp->copy_stream) (void)hipStreamSynchronize(sp->copy_stream);
            if (sp->device_bytes() > (2048ull << 20)) { delete sp; return; }      // keep workspaces of up to 2 GiB between calls
            std::vector<RpSession*> doomed;
            { std::lock_guard<std::mutex> lk(r->session_mu);
              const_cast<am_replacer*>(r)->session_delete = [](void* p) { delete static_cast<RpSession*>(p); };
              r->sessions.push_back(sp);
              // at most 8 cached workspaces and at most 4 GiB of device memory in all of them (each is below 2 GiB): the oldest go first
              for (;;) {
                  size_t held = 0;
                  for (void* q : r->sessions) held += static_cast<RpSession*>(q)->device_bytes();
                  if (r->sessions.size() <= 1 || (r->sessions.size() <= 8 && held <= (4096ull << 20))) break;
                  doomed.push_back(static_cast<RpSession*>(r->sessions.front()));
                  r->sessions.erase(r->sessions.begin());
              } }
            for (RpSession* q : doomed) delete q;
        }
    } give_back{r, sp};
    RpSession& s = *sp;
    AM_TRY(s.totals.ensure(128));
    if (!s.tot_host) {
        if (hipHostMalloc((void**)&s.tot_host, 128, hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess) { s.tot_host = nullptr; return fail(AM_ERR_OOM, "hipHostMalloc failed"); }
        std::memset(s.tot_host, 0, 128);                  // (fine-grained: a device store is visible to the host while the kernel is still running)
    }
    if (!s.copy_stream && (hipStreamCreateWithFlags(&s.copy_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&s.ev_spliced, hipEventDisableTiming) != hipSuccess))
        return fail(AM_ERR_HIP, "could not create the copy stream");
    const uint8_t* base_text = (const uint8_t*)in->d_text;               // never modified: every text piece points into it
    const uint64_t* cur_offs = in->d_offsets;                            // logical offsets of the active haystacks (lengths only after pass 0)
    uint32_t n_act = n_hay;
    int nxt = 0;
    {
        std::vector<uint32_t> o(n_hay); std::vector<int64_t> t(n_hay, 1);      // initialThreshold = 1 (Replacer.hs:211)
        for (uint32_t i = 0; i < n_hay; i++) o[i] = i;
        AM_TRY(s.first_orig.ensure(n_hay * sizeof(uint32_t))); AM_TRY(s.first_thr.ensure(n_hay * sizeof(int64_t)));
        HIP_TRY(hipMemcpyAsync(s.first_orig.p, o.data(), n_hay * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(s.first_thr.p, t.data(), n_hay * sizeof(int64_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    const uint32_t* cur_orig = (const uint32_t*)s.first_orig.p;
    const int64_t* cur_thr = (const int64_t*)s.first_thr.p;
    const uint32_t ov = 4u * (flavor->h.max_needle_cps ? flavor->h.max_needle_cps : 1u) + 4u;
    // piece lists of pass 0: one piece per haystack
    int cur_pt = 0;
    AM_TRY(s.pt_pieces[0].ensure(((size_t)n_hay * 2 + 2) * sizeof(RpPiece)));
    AM_TRY(s.pt_start[0].ensure(((size_t)n_hay + 1) * 8)); AM_TRY(s.pt_cnt[0].ensure(((size_t)n_hay + 1) * 4));
    HIP_TRY(launch_pt_init(in->d_offsets, n_hay, (RpPiece*)s.pt_pieces[0].p, (uint64_t*)s.pt_start[0].p, (uint32_t*)s.pt_cnt[0].p, st));
    // pass 0 scans the caller's batch; afterwards the records come from the window scans + the shifted old records
    uint64_t n_rec = 0;                                   // records of the current pass: exact when n_rec_dev == nullptr, else an upper bound ...
    const uint64_t* n_rec_dev = nullptr;                  // ... and the exact count is still on the device
    int cur_rec = 0;
    {
        res->scanned += in->total;
        s.ws.dev = in->dev; s.ws.d_text = in->d_text; s.ws.d_offsets = in->d_offsets; s.ws.owns = false; s.ws.total = in->total; s.ws.n_hay = n_hay;
        AM_TRY(finish_batch(&s.ws));
        auto sink = [&](uint64_t n, Record** ptr) -> int { AM_TRY(s.recbuf[0].ensure(n * sizeof(Record))); *ptr = (Record*)s.recbuf[0].p; return AM_OK; };
        AM_TRY(run_records(r->a, r->case_mode, &s.ws, sink, &n_rec));
    }
    const bool trace = cfg::on(cfg::kRpTrace);
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_a = 0, t_b = 0, t_c = 0, t_sync = 0;
    // finished haystacks of the previous pass: their bytes are on their way home on the copy stream; the host looks at the list after
    // the next pass's (only) synchronisation
    uint64_t prev_n_fin = 0, prev_total_fin = 0; uint8_t* prev_home = nullptr;
    hipEvent_t ev_copied = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&ev_copied, hipEventDisableTiming));
    struct EvGuard { hipEvent_t e; ~EvGuard() { (void)hipEventDestroy(e); } } ev_guard{ev_copied};
    bool copies_pending = false, ev_copied_used = false;
    auto finished_home = [&]() -> int {
        if (!copies_pending) return AM_OK;
        HIP_TRY(hipStreamSynchronize(s.copy_stream));
        copies_pending = false;
        for (uint64_t i = 0; i < prev_n_fin; i++) {
            const RpFin& f = s.fin_host[i];
            if (f.orig >= n_hay || f.off + f.len > prev_total_fin) return fail(AM_ERR_HIP, "replacer pass produced inconsistent metadata (internal error)");
            if (f.status == kRpNothing) res->just[f.orig] = 0;
            else res->text[f.orig] = am_replaced::Item{prev_home + f.off, (size_t)f.len};
        }
        return AM_OK;
    };

    int cur_rf = 0; bool have_ranges = false;           // (see the ranges buffers below)
    while (n_act > 0) {
        double t0 = now();
        res->passes++;
        DevBuf& records = s.recbuf[cur_rec];
        const uint64_t n1 = (uint64_t)n_act + 1;
        // the record-parallel fold needs the exact count on the host: fetch it when that regime is possible
        if (n_rec_dev && (n_rec > 2048ull * n_act || cfg::get(cfg::kRpParallelFold) != cfg::kUnset)) {
            HIP_TRY(hipMemcpyAsync(&s.tot_host[9], n_rec_dev, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            n_rec = s.tot_host[9]; n_rec_dev = nullptr;
        }
        // record ranges of the haystacks: two buffers that take turns -- the merge at the end of a pass leaves the offsets of the records it
        // writes (per haystack of the next pass) in the other one, which ARE the next pass's ranges: no search then
        DevBuf& rfb = cur_rf ? s.rec_first2 : s.rec_first; DevBuf& rfb_next = cur_rf ? s.rec_first : s.rec_first2;
        AM_TRY(rfb.ensure(n1 * 8)); AM_TRY(s.kept.ensure((n_rec + 1) * sizeof(RpKept))); AM_TRY(s.hs.ensure(n1 * sizeof(RpHay)));
        AM_TRY(s.len_next.ensure(n1 * 8)); AM_TRY(s.len_fin.ensure(n1 * 8)); AM_TRY(s.tiles.ensure(n1 * 4)); AM_TRY(s.act.ensure(n1 * 4)); AM_TRY(s.fin.ensure(n1 * 4));
        AM_TRY(s.off_next.ensure(n1 * 8)); AM_TRY(s.off_fin.ensure(n1 * 8)); AM_TRY(s.tile_off.ensure(n1 * 8)); AM_TRY(s.act_idx.ensure(n1 * 8)); AM_TRY(s.fin_idx.ensure(n1 * 8));
        AM_TRY(s.nwin.ensure(n1 * 4)); AM_TRY(s.win_off.ensure(n1 * 8)); AM_TRY(s.pt_need.ensure(n1 * 4)); AM_TRY(s.pt_need_off.ensure(n1 * 8));
        AM_TRY(s.wins.ensure((n_rec + 1) * sizeof(RpWin))); AM_TRY(s.wlen.ensure((n_rec + 2) * 4)); AM_TRY(s.woffs.ensure((n_rec + 2) * 8));
        size_t t32 = 0, t64 = 0, tw = 0;
        if (scan_temp_bytes(n1, &t32) != hipSuccess || scan64_temp_bytes(n1, &t64) != hipSuccess || scan_temp_bytes(n_rec + 2, &tw) != hipSuccess) return fail(AM_ERR_HIP, "scan sizing failed");
        AM_TRY(s.scan_tmp.ensure(std::max(std::max(t32, t64), tw) + 16));
        const size_t tmp2 = s.scan_tmp.cap - 16;
        AM_TRY(records.ensure(sizeof(Record)));
        RpRoute route{(uint64_t*)s.len_next.p, (uint64_t*)s.len_fin.p, (uint32_t*)s.tiles.p, (uint32_t*)s.act.p, (uint32_t*)s.fin.p};
        RpRouted rt{(const uint64_t*)s.off_next.p, (const uint64_t*)s.off_fin.p, (const uint64_t*)s.tile_off.p, (const uint64_t*)s.act_idx.p, (const uint64_t*)s.fin_idx.p};
        // the per-haystack fold also finds its record range and writes the piece / window counts (one dispatch instead of three in the pass's
        // chain); the record-parallel fold keeps the separate launches
        bool par_fold = (n_rec_dev ? 0 : n_rec) > 2048ull * n_act;
        if (cfg::get(cfg::kRpParallelFold) != cfg::kUnset) par_fold = cfg::get(cfg::kRpParallelFold) != 0;
        const bool no_fuse = cfg::on(cfg::kRpNoFuse);                                             // A/B
        const bool fused = !par_fold && !no_fuse;
        if (fused) {
            Prof pr("rp_pass", st);
            const RpFused fu{have_ranges ? nullptr : (uint64_t*)rfb.p, n_rec_dev ? 0 : n_rec, n_rec_dev, (const uint32_t*)s.pt_cnt[cur_pt].p, (uint32_t*)s.pt_need.p, (uint32_t*)s.nwin.p};
            HIP_TRY(launch_rp_pass(false, r->t, base_text, cur_offs, (const Record*)records.p, (const uint64_t*)rfb.p, cur_thr, max_length, (RpKept*)s.kept.p, (RpHay*)s.hs.p, route, n_act, 0u, st, &fu));
        } else {
            { Prof pr("rp_ranges", st);
              if (n_rec_dev) HIP_TRY(launch_rp_ranges_dev((const Record*)records.p, n_rec_dev, (uint64_t*)rfb.p, route, n_act, st));
              else HIP_TRY(launch_rp_ranges((const Record*)records.p, n_rec, (uint64_t*)rfb.p, route, n_act, st)); }
            AM_TRY(rp_fold(s, r, false, base_text, cur_offs, (const Record*)records.p, n_rec_dev ? 0 : n_rec, cur_thr, max_length, route, n_act, st, (const uint64_t*)rfb.p));
        }
        const bool small = n1 <= (1u << 18);
        { Prof pr("rp_scans", st);
          if (!fused) HIP_TRY(launch_pt_count((const RpHay*)s.hs.p, (const uint32_t*)s.pt_cnt[cur_pt].p, n_act, (uint32_t*)s.pt_need.p, (uint32_t*)s.nwin.p, st));
          if (small) {
              ScanJobs jobs{};
              jobs.j[0] = ScanJob{nullptr, route.len_next, (uint64_t*)s.off_next.p, n1, nullptr};
              jobs.j[1] = ScanJob{nullptr, route.len_fin, (uint64_t*)s.off_fin.p, n1, nullptr};
              jobs.j[2] = ScanJob{(const uint32_t*)s.pt_need.p, nullptr, (uint64_t*)s.pt_need_off.p, n1, nullptr};
              jobs.j[3] = ScanJob{route.act, nullptr, (uint64_t*)s.act_idx.p, n1, nullptr};
              jobs.j[4] = ScanJob{route.fin, nullptr, (uint64_t*)s.fin_idx.p, n1, nullptr};
              jobs.j[5] = ScanJob{(const uint32_t*)s.nwin.p, nullptr, (uint64_t*)s.win_off.p, n1, nullptr};
              jobs.n_jobs = 6;
              HIP_TRY(launch_scan_jobs(jobs, st));
          } else {
              HIP_TRY(launch_scan64(s.scan_tmp.p, tmp2, route.len_next, (uint64_t*)s.off_next.p, n1, st));
              HIP_TRY(launch_scan64(s.scan_tmp.p, tmp2, route.len_fin, (uint64_t*)s.off_fin.p, n1, st));
              HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, (const uint32_t*)s.pt_need.p, (uint64_t*)s.pt_need_off.p, n1, st));
              HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, route.act, (uint64_t*)s.act_idx.p, n1, st));
              HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, route.fin, (uint64_t*)s.fin_idx.p, n1, st));
              HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, (const uint32_t*)s.nwin.p, (uint64_t*)s.win_off.p, n1, st));
          } }
        uint64_t woffs_last = n_rec;
        { Prof pr("rp_windows", st);
          if (!small) HIP_TRY(hipMemsetAsync(s.wlen.p, 0, (n_rec + 2) * 4, st));
          HIP_TRY(launch_rp_win_meta(r->t, rt, (const RpHay*)s.hs.p, (const uint64_t*)rfb.p, (const RpKept*)s.kept.p,
                                     (const uint64_t*)s.win_off.p, ov, (RpWin*)s.wins.p, (uint32_t*)s.wlen.p, n_act, st, true));
          if (small) {
              ScanJobs jobs{};
              jobs.j[0] = ScanJob{(const uint32_t*)s.wlen.p, nullptr, (uint64_t*)s.woffs.p, 1, (const uint64_t*)s.win_off.p + n_act};
              jobs.n_jobs = 1;
              HIP_TRY(launch_scan_jobs(jobs, st));
              woffs_last = ~0ull;
          } else HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, (const uint32_t*)s.wlen.p, (uint64_t*)s.woffs.p, n_rec + 1, st)); }
        // the pass's ONE synchronisation: bytes of next text, bytes of finished text, -, haystacks still active, haystacks finished,
        // windows, window bytes, piece entries, and the exact record count of this pass when it was still on the device
        // (k_rp_totals writes straight into the pinned host block -- 80 bytes of posted PCIe writes -- instead of into device memory that a
        // 16-us copy dispatch would then move)
        // ... and the host waits for the LAST word of that block (a sequence number the kernel stores after a system-scope fence) by spinning on
        // it for a while before it falls back to hipStreamSynchronize: the blocking wait's wake-up cost 20-30 us of every pass's ~250
        const uint64_t seq = ++s.tot_seq;
        HIP_TRY(launch_rp_totals(rt, n_act, (const uint64_t*)s.win_off.p, (const uint64_t*)s.woffs.p, woffs_last, s.tot_host, st,
                                 (const uint64_t*)s.pt_need_off.p + n_act, n_rec_dev, seq));
        const double t_s0 = now();
        {
            const bool no_spin = cfg::on(cfg::kRpNoSpin);                                 // A/B
            bool seen = false;
            if (!no_spin) {
                const double give_up = t_s0 + 2e-3;
                for (uint32_t it = 0; !seen; it++) {
                    seen = __atomic_load_n(&s.tot_host[15], __ATOMIC_ACQUIRE) == seq;
                    if (!seen && (it & 1023u) == 1023u && now() > give_up) break;
                }
            }
            if (!seen) HIP_TRY(hipStreamSynchronize(st));
        }
        if (trace) t_sync += now() - t_s0;
        const uint64_t* tot = s.tot_host;
        const uint64_t total_next = tot[0], total_fin = tot[1], n_next = tot[3], n_fin = tot[4], n_win = tot[5], total_w = tot[6], n_pieces = tot[8];
        if (n_rec_dev) { n_rec = tot[9]; n_rec_dev = nullptr; }
        AM_TRY(finished_home());                              // the previous pass's finished haystacks (their copies have had a whole pass)
        t_a += now() - t0; t0 = now();
        if (n_win >= 0xFFFFFFF0ull) return fail(AM_ERR_UNSUPPORTED, "too many replacements in one pass; split the batch");
        // ---- the next pass's piece lists; finished haystacks are materialised and go home
        AM_TRY(s.offs[nxt].ensure((n_next + 1) * 8)); AM_TRY(s.orig[nxt].ensure((n_next + 1) * 4)); AM_TRY(s.thr[nxt].ensure((n_next + 1) * 8));
        AM_TRY(s.fin_text.ensure(total_fin + 16)); AM_TRY(s.fin_meta.ensure((n_fin + 1) * sizeof(RpFin)));
        AM_TRY(s.pt_pieces[cur_pt ^ 1].ensure((n_pieces + 2) * sizeof(RpPiece)));
        AM_TRY(s.pt_start[cur_pt ^ 1].ensure((n_next + 1) * 8)); AM_TRY(s.pt_cnt[cur_pt ^ 1].ensure((n_next + 1) * 4));
        AM_TRY(s.pt_fin_start.ensure((n_fin + 1) * 8)); AM_TRY(s.pt_fin_cnt.ensure((n_fin + 1) * 4));
        if (ev_copied_used) HIP_TRY(hipStreamWaitEvent(st, ev_copied, 0));      // the previous pass's finished texts are written and their metadata has left fin_meta
        { Prof pr("rp_route", st);
          HIP_TRY(launch_rp_route((const RpHay*)s.hs.p, rt, cur_orig, n_act, (uint64_t*)s.offs[nxt].p, (uint32_t*)s.orig[nxt].p, (int64_t*)s.thr[nxt].p, (RpFin*)s.fin_meta.p, st)); }
        { Prof pr("pt_build", st);
          HIP_TRY(launch_pt_build(r->t, (const RpHay*)s.hs.p, (const uint64_t*)rfb.p, (const RpKept*)s.kept.p, (const RpPiece*)s.pt_pieces[cur_pt].p,
                                  (const uint64_t*)s.pt_start[cur_pt].p, (const uint32_t*)s.pt_cnt[cur_pt].p, (const uint64_t*)s.pt_need_off.p, rt, n_act,
                                  (RpPiece*)s.pt_pieces[cur_pt ^ 1].p, (uint64_t*)s.pt_start[cur_pt ^ 1].p, (uint32_t*)s.pt_cnt[cur_pt ^ 1].p,
                                  (uint64_t*)s.pt_fin_start.p, (uint32_t*)s.pt_fin_cnt.p, st)); }
        res->spliced += total_fin;
        if (n_fin) {
            uint8_t* home = nullptr;
            if (total_fin) AM_TRY(res->room((size_t)total_fin, &home));
            AM_TRY(s.pin_meta((n_fin + 1) * sizeof(RpFin)));
            // the finished texts are written out on the COPY stream (64 us of a 270-us pass that nothing of the next pass waits for): it starts when
            // this pass's piece lists and metadata are complete; what it reads is not touched before the next pass's host-side look at the copy
            // stream (finished_home, after the totals) -- and the next rp_route waits for the event as well
            const bool mat_main = cfg::on(cfg::kRpMatMain);                              // A/B: on the pass's own stream, as before
            hipStream_t mst = mat_main ? st : s.copy_stream;
            if (!mat_main) { HIP_TRY(hipEventRecord(s.ev_spliced, st)); HIP_TRY(hipStreamWaitEvent(s.copy_stream, s.ev_spliced, 0)); }
            { Prof pr("pt_materialise", mst);
              HIP_TRY(launch_pt_materialise((const RpPiece*)s.pt_pieces[cur_pt ^ 1].p, (const uint64_t*)s.pt_fin_start.p, (const uint32_t*)s.pt_fin_cnt.p, (const RpFin*)s.fin_meta.p,
                                            (uint32_t)n_fin, base_text, r->t.repl, res->dev >= 0 && total_fin ? home : (uint8_t*)s.fin_text.p, mst)); }
            if (mat_main) { HIP_TRY(hipEventRecord(s.ev_spliced, st)); HIP_TRY(hipStreamWaitEvent(s.copy_stream, s.ev_spliced, 0)); }
            if (total_fin && res->dev < 0) HIP_TRY(hipMemcpyAsync(home, s.fin_text.p, total_fin, hipMemcpyDeviceToHost, s.copy_stream));
            HIP_TRY(hipMemcpyAsync(s.fin_host, s.fin_meta.p, n_fin * sizeof(RpFin), hipMemcpyDeviceToHost, s.copy_stream));
            HIP_TRY(hipEventRecord(ev_copied, s.copy_stream));
            ev_copied_used = true;
            prev_n_fin = n_fin; prev_total_fin = total_fin; prev_home = home; copies_pending = true;
        }
        t_b += now() - t0; t0 = now();
        // ---- the next pass's records
        uint64_t next_n_rec = 0; const uint64_t* next_n_rec_dev = nullptr; bool next_have_ranges = false;
        if (n_next > 0) {
            DevBuf& next_records = s.recbuf[cur_rec ^ 1];
            if (total_w > total_next) {
                // tiny texts: the windows would be larger than the texts themselves -- materialise the next texts and scan them whole
                AM_TRY(s.text[0].ensure(padded_text(total_next)));
                HIP_TRY(launch_pt_materialise_next((const RpPiece*)s.pt_pieces[cur_pt ^ 1].p, (const uint64_t*)s.pt_start[cur_pt ^ 1].p, (const uint32_t*)s.pt_cnt[cur_pt ^ 1].p,
                                                   (const uint64_t*)s.offs[nxt].p, (uint32_t)n_next, base_text, r->t.repl, (uint8_t*)s.text[0].p, st));
                HIP_TRY(hipMemsetAsync((uint8_t*)s.text[0].p + total_next, 0, padded_text(total_next) - (size_t)total_next, st));
                s.ws.dev = in->dev; s.ws.d_text = s.text[0].p; s.ws.d_offsets = (uint64_t*)s.offs[nxt].p; s.ws.owns = false; s.ws.total = total_next; s.ws.n_hay = (uint32_t)n_next;
                AM_TRY(finish_batch(&s.ws));
                auto sink = [&](uint64_t n, Record** ptr) -> int { AM_TRY(next_records.ensure(n * sizeof(Record))); *ptr = (Record*)next_records.p; return AM_OK; };
                AM_TRY(run_records(r->a, r->case_mode, &s.ws, sink, &next_n_rec));
                res->scanned += total_next;
            } else {
                // windows around the replacements (gathered from the new piece lists) + the shifted old records; no host round trip when the
                // worst-case record pool of the window scan stays small
                const bool lean = total_w <= (64ull << 20);
                uint64_t n_wrec = 0; const uint64_t* n_wrec_dev = nullptr;
                AM_TRY(s.wtext.ensure(padded_text(total_w)));
                AM_TRY(s.wrec.ensure(((lean ? total_w : 0) + 1) * sizeof(Record)));
                if (n_win > 0 && total_w > 0) {
                    { Prof pr("rp_windows", st);
                      HIP_TRY(launch_pt_win_copy((const RpWin*)s.wins.p, (const uint64_t*)s.woffs.p, (const RpPiece*)s.pt_pieces[cur_pt ^ 1].p, (const uint64_t*)s.pt_start[cur_pt ^ 1].p,
                                                 (const uint32_t*)s.pt_cnt[cur_pt ^ 1].p, base_text, r->t.repl, (uint8_t*)s.wtext.p, n_win, total_w, padded_text(total_w), st)); }
                    s.ws2.dev = in->dev; s.ws2.d_text = s.wtext.p; s.ws2.d_offsets = (uint64_t*)s.woffs.p; s.ws2.owns = false; s.ws2.total = total_w; s.ws2.n_hay = (uint32_t)n_win;
                    AM_TRY(finish_batch(&s.ws2));
                    if (lean) AM_TRY(run_records_async(r->a, r->case_mode, &s.ws2, (Record*)s.wrec.p, &n_wrec_dev, st));
                    else {
                        auto wsink = [&](uint64_t n, Record** ptr) -> int { AM_TRY(s.wrec.ensure(n * sizeof(Record))); *ptr = (Record*)s.wrec.p; return AM_OK; };
                        AM_TRY(run_records(r->a, r->case_mode, &s.ws2, wsink, &n_wrec));
                    }
                    res->scanned += total_w;
                }
                const uint64_t wrec_bound = n_wrec_dev ? total_w : n_wrec;
                AM_TRY(s.wrec_first.ensure((n_win + 2) * 8)); AM_TRY(s.mcount.ensure((n_next + 1) * 4)); AM_TRY(rfb_next.ensure((n_next + 1) * 8));
                AM_TRY(next_records.ensure((n_rec + wrec_bound + 1) * sizeof(Record)));
                Prof pr("rp_merge", st);
                if (n_wrec_dev) HIP_TRY(launch_rp_ranges_dev((const Record*)s.wrec.p, n_wrec_dev, (uint64_t*)s.wrec_first.p, RpRoute{nullptr, nullptr, nullptr, nullptr, nullptr}, (uint32_t)n_win, st));
                else HIP_TRY(launch_rp_ranges((const Record*)s.wrec.p, n_wrec, (uint64_t*)s.wrec_first.p, RpRoute{nullptr, nullptr, nullptr, nullptr, nullptr}, (uint32_t)n_win, st));
                HIP_TRY(launch_rp_merge(false, (const Record*)records.p, (const uint64_t*)rfb.p, (const RpKept*)s.kept.p, (const RpHay*)s.hs.p, cur_offs, rt,
                                        (const uint64_t*)s.win_off.p, (const RpWin*)s.wins.p, (const Record*)s.wrec.p, (const uint64_t*)s.wrec_first.p, ov, n_act,
                                        (uint32_t*)s.mcount.p, nullptr, nullptr, st));
                if (n_next + 1 <= (1u << 18)) {
                    ScanJobs jobs{};
                    jobs.j[0] = ScanJob{(const uint32_t*)s.mcount.p, nullptr, (uint64_t*)rfb_next.p, n_next + 1, nullptr};
                    jobs.n_jobs = 1;
                    HIP_TRY(launch_scan_jobs(jobs, st));
                } else HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, (const uint32_t*)s.mcount.p, (uint64_t*)rfb_next.p, n_next + 1, st));
                HIP_TRY(launch_rp_merge(true, (const Record*)records.p, (const uint64_t*)rfb.p, (const RpKept*)s.kept.p, (const RpHay*)s.hs.p, cur_offs, rt,
                                        (const uint64_t*)s.win_off.p, (const RpWin*)s.wins.p, (const Record*)s.wrec.p, (const uint64_t*)s.wrec_first.p, ov, n_act,
                                        (uint32_t*)s.mcount.p, (const uint64_t*)rfb_next.p, (Record*)next_records.p, st));
                next_n_rec = n_rec + wrec_bound;                 // an upper bound; the exact count is read with the next pass's totals
                next_n_rec_dev = (const uint64_t*)rfb_next.p + n_next;
                next_have_ranges = true;
            }
        }
        t_c += now() - t0;
        if (trace && cfg::get(cfg::kRpTrace) == 2)
            std::fprintf(stderr, "[am_replacer pt pass %u] active %u -> %llu, finished %llu, records <= %llu, windows %llu (%llu B), next text %llu B\n", (unsigned)res->passes, n_act,
                         (unsigned long long)n_next, (unsigned long long)n_fin, (unsigned long long)n_rec, (unsigned long long)n_win, (unsigned long long)total_w, (unsigned long long)total_next);
        cur_rec ^= 1; cur_pt ^= 1; n_rec = next_n_rec; n_rec_dev = next_n_rec_dev;
        const bool no_reuse = cfg::on(cfg::kRpNoRangeReuse);                              // A/B
        have_ranges = next_have_ranges && !no_reuse; cur_rf ^= 1;
        cur_offs = (const uint64_t*)s.offs[nxt].p; cur_orig = (const uint32_t*)s.orig[nxt].p; cur_thr = (const int64_t*)s.thr[nxt].p;
        n_act = (uint32_t)n_next; nxt ^= 1;
    }
    HIP_TRY(hipStreamSynchronize(st));
    AM_TRY(finished_home());
    if (trace) std::fprintf(stderr, "[am_replacer pt] fold+scans %.1f ms (of which waiting for the device %.1f), pieces+materialise %.1f ms, windows+merge %.1f ms\n", t_a * 1e3, t_sync * 1e3, t_b * 1e3, t_c * 1e3);
    return AM_OK;
}

// Replacer.hs:203-242 runWithLimit for every haystack of `in`, all passes on the device.
int replacer_run(const am_replacer* r, const am_batch* in, uint64_t max_length, am_replaced* res)
{
    const uint32_t n_hay = in->n_hay;
    res->text.assign(n_hay, am_replaced::Item());
    res->just.assign(n_hay, 1);
    if (n_hay == 0) return AM_OK;
    if (in->dev != r->a->dev) return fail(AM_ERR_INVALID, "replacer and batch live on different devices");
    {
        // CaseSensitive replacers on the suffix-filter route keep the texts as piece tables (AM_RP_SPLICE=1: the splicing loop, for A/B and tests)
        const Flavor* fl = nullptr;
        AM_TRY(prepare(r->a, r->case_mode, &fl));
        // ... when the batch is made of many documents: the piece-table kernels give a haystack to ONE wavefront, the splicing loop
        // cuts every text into 16-KiB tiles.  One 1-MB document with half a million replacements per pass: 472 ms vs 90 ms (measured).
        const bool many_documents = n_hay >= 64 && in->total / n_hay <= (1ull << 20);
        const bool pt = r->case_mode == AM_CASE_SENSITIVE && fl->h.sf_enabled && fl->h.root_vlen == 0 && r->a->kernel_pref != 1 &&
                        !cfg::on(cfg::kRpFullScans) && !cfg::on(cfg::kRpSplice) && (many_documents || cfg::on(cfg::kRpPieces)) &&
                        n_hay < (1u << 24) && in->total < (1ull << 40);        // RpWin::src_abs packs (haystack index << 40 | start): beyond that the splicing loop runs
        if (pt) return replacer_run_pt(r, in, max_length, res, fl);
    }
    ON_DEVICE(in->dev);
    hipStream_t st; AM_TRY(get_stream(in->dev, &st));
    // take the replacer's cached workspace (or make one); it goes back at the end unless it has grown large
    RpSession* sp = nullptr;
    { std::lock_guard<std::mutex> lk(r->session_mu); if (!r->sessions.empty()) { sp = static_cast<RpSession*>(r->sessions.back()); r->sessions.pop_back(); } }
    if (!sp) sp = new RpSession();
    struct Return {
        const am_replacer* r; RpSession* sp;
        ~Return()
        {
            if (sp->copy_stream) (void)hipStreamSynchronize(sp->copy_stream);
            if (sp->device_bytes() > (2048ull << 20)) { delete sp; return; }      // keep workspaces of up to 2 GiB between calls
            std::vector<RpSession*> doomed;
            { std::lock_guard<std::mutex> lk(r->session_mu);
              const_cast<am_replacer*>(r)->session_delete = [](void* p) { delete static_cast<RpSession*>(p); };
              r->sessions.push_back(sp);
              // at most 8 cached workspaces and at most 4 GiB of device memory in all of them (each is below 2 GiB): the oldest go first
              for (;;) {
                  size_t held = 0;
                  for (void* q : r->sessions) held += static_cast<RpSession*>(q)->device_bytes();
                  if (r->sessions.size() <= 1 || (r->sessions.size() <= 8 && held <= (4096ull << 20))) break;
                  doomed.push_back(static_cast<RpSession*>(r->sessions.front()));
                  r->sessions.erase(r->sessions.begin());
              } }
            for (RpSession* q : doomed) delete q;
        }
    } give_back{r, sp};
    RpSession& s = *sp;
    AM_TRY(s.totals.ensure(128));
    if (!s.tot_host) {
        if (hipHostMalloc((void**)&s.tot_host, 128, hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess) { s.tot_host = nullptr; return fail(AM_ERR_OOM, "hipHostMalloc failed"); }
        std::memset(s.tot_host, 0, 128);                  // (fine-grained: a device store is visible to the host while the kernel is still running)
    }
    // pass 0 reads the caller's batch in place; afterwards the text ping-pongs between s.text[0] and s.text[1]
    const uint8_t* cur_text = (const uint8_t*)in->d_text;
    const uint64_t* cur_offs = in->d_offsets;
    uint64_t total = in->total;
    uint32_t n_act = n_hay;
    int nxt = 0;
    DevBuf& first_orig = s.first_orig; DevBuf& first_thr = s.first_thr;
    {
        std::vector<uint32_t> o(n_hay); std::vector<int64_t> t(n_hay, 1);      // initialThreshold = 1 (Replacer.hs:211)
        for (uint32_t i = 0; i < n_hay; i++) o[i] = i;
        AM_TRY(first_orig.ensure(n_hay * sizeof(uint32_t))); AM_TRY(first_thr.ensure(n_hay * sizeof(int64_t)));
        HIP_TRY(hipMemcpy(first_orig.p, o.data(), n_hay * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(first_thr.p, t.data(), n_hay * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    const uint32_t* cur_orig = (const uint32_t*)first_orig.p;
    const int64_t* cur_thr = (const int64_t*)first_thr.p;
    if (!s.copy_stream && (hipStreamCreateWithFlags(&s.copy_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&s.ev_spliced, hipEventDisableTiming) != hipSuccess))
        return fail(AM_ERR_HIP, "could not create the copy stream");
    // Incremental re-scan (am_replace.hip): after the first pass only windows around the replacements are scanned and
    // merged with the shifted records of the previous pass.  Needs the suffix-filter kernel's position-local semantics
    // (automata with the empty needle re-scan everything); AM_RP_FULL_SCANS=1 turns it off (A/B, tests).
    const Flavor* flavor = nullptr;
    AM_TRY(prepare(r->a, r->case_mode, &flavor));
    const uint32_t ov = 4u * (flavor->h.max_needle_cps ? flavor->h.max_needle_cps : 1u) + 4u;
    const bool inc_enabled = flavor->h.sf_enabled && flavor->h.root_vlen == 0 && r->a->kernel_pref != 1 && !cfg::on(cfg::kRpFullScans);
    bool have_inc = false;
    uint64_t inc_n_rec = 0;
    int cur_rec = 0;
    // AM_RP_TRACE=1: wall-clock split of the loop on stderr (development aid)
    const bool trace = cfg::on(cfg::kRpTrace);
    double t_scan = 0, t_fold = 0, t_splice = 0, t_home = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    struct Report { bool on; double &a, &b, &c, &d; ~Report() { if (on) std::fprintf(stderr, "[am_replacer] scan %.1f ms, fold+scans %.1f ms, splice+D2H %.1f ms, scatter %.1f ms\n", a * 1e3, b * 1e3, c * 1e3, d * 1e3); } } report{trace, t_scan, t_fold, t_splice, t_home};

    while (n_act > 0) {
        double t0 = now();
        res->passes++;
        // ---- the scan (Replacer.hs:223-225): everything, unless the previous pass already derived this pass's records
        uint64_t n_rec = 0;
        DevBuf& records = s.recbuf[cur_rec];
        if (have_inc) { n_rec = inc_n_rec; have_inc = false; }
        else {
            res->scanned += total;
            s.ws.dev = in->dev;
            s.ws.d_text = const_cast<uint8_t*>(cur_text); s.ws.d_offsets = const_cast<uint64_t*>(cur_offs); s.ws.owns = false;
            s.ws.total = total; s.ws.n_hay = n_act;
            AM_TRY(finish_batch(&s.ws));
            auto sink = [&](uint64_t n, Record** ptr) -> int { AM_TRY(records.ensure(n * sizeof(Record))); *ptr = (Record*)records.p; return AM_OK; };
            AM_TRY(run_records(r->a, r->case_mode, &s.ws, sink, &n_rec));
        }
        t_scan += now() - t0; t0 = now();
        // ---- per-haystack fold of the records
        const uint64_t n1 = (uint64_t)n_act + 1;
        AM_TRY(s.rec_first.ensure(n1 * 8)); AM_TRY(s.kept.ensure((n_rec + 1) * sizeof(RpKept))); AM_TRY(s.hs.ensure(n1 * sizeof(RpHay)));
        AM_TRY(s.len_next.ensure(n1 * 8)); AM_TRY(s.len_fin.ensure(n1 * 8)); AM_TRY(s.tiles.ensure(n1 * 4)); AM_TRY(s.act.ensure(n1 * 4)); AM_TRY(s.fin.ensure(n1 * 4));
        AM_TRY(s.off_next.ensure(n1 * 8)); AM_TRY(s.off_fin.ensure(n1 * 8)); AM_TRY(s.tile_off.ensure(n1 * 8)); AM_TRY(s.act_idx.ensure(n1 * 8)); AM_TRY(s.fin_idx.ensure(n1 * 8));
        size_t t32 = 0, t64 = 0;
        if (scan_temp_bytes(n1, &t32) != hipSuccess || scan64_temp_bytes(n1, &t64) != hipSuccess) return fail(AM_ERR_HIP, "scan sizing failed");
        const size_t tmp_bytes = t32 > t64 ? t32 : t64;
        AM_TRY(s.scan_tmp.ensure(tmp_bytes + 16));
        AM_TRY(records.ensure(sizeof(Record)));            // a valid pointer even when nothing matched
        RpRoute route{(uint64_t*)s.len_next.p, (uint64_t*)s.len_fin.p, (uint32_t*)s.tiles.p, (uint32_t*)s.act.p, (uint32_t*)s.fin.p};
        { Prof pr("rp_ranges", st); HIP_TRY(launch_rp_ranges((const Record*)records.p, n_rec, (uint64_t*)s.rec_first.p, route, n_act, st)); }
        AM_TRY(rp_fold(s, r, r->case_mode == AM_IGNORE_CASE, cur_text, cur_offs, (const Record*)records.p, n_rec, cur_thr, max_length, route, n_act, st, (const uint64_t*)s.rec_first.p));
        RpRouted rt{(const uint64_t*)s.off_next.p, (const uint64_t*)s.off_fin.p, (const uint64_t*)s.tile_off.p, (const uint64_t*)s.act_idx.p, (const uint64_t*)s.fin_idx.p};
        // windows of the incremental re-scan (their geometry follows from the kept matches alone, the text is copied after the splice)
        const bool try_inc = inc_enabled && n_rec > 0;
        const bool small = n1 <= (1u << 18);          // bookkeeping sums in one launch (k_scan_jobs) instead of a dozen scan launches
        size_t tmp2 = tmp_bytes;
        uint64_t woffs_last = n_rec;
        if (try_inc) {
            AM_TRY(s.nwin.ensure(n1 * 4)); AM_TRY(s.win_off.ensure(n1 * 8));
            AM_TRY(s.wins.ensure((n_rec + 1) * sizeof(RpWin))); AM_TRY(s.wlen.ensure((n_rec + 2) * 4)); AM_TRY(s.woffs.ensure((n_rec + 2) * 8));
            size_t tw = 0;
            if (scan_temp_bytes(n_rec + 1, &tw) != hipSuccess) return fail(AM_ERR_HIP, "scan sizing failed");
            AM_TRY(s.scan_tmp.ensure(std::max(tw, tmp_bytes) + 16));
            tmp2 = s.scan_tmp.cap - 16;
            HIP_TRY(launch_rp_win_count((const RpHay*)s.hs.p, n_act, (uint32_t*)s.nwin.p, st));
        }
        { Prof pr("rp_scans", st);
          if (small) {
              ScanJobs jobs{};
              jobs.j[0] = ScanJob{nullptr, route.len_next, (uint64_t*)s.off_next.p, n1, nullptr};
              jobs.j[1] = ScanJob{nullptr, route.len_fin, (uint64_t*)s.off_fin.p, n1, nullptr};
              jobs.j[2] = ScanJob{route.tiles, nullptr, (uint64_t*)s.tile_off.p, n1, nullptr};
              jobs.j[3] = ScanJob{route.act, nullptr, (uint64_t*)s.act_idx.p, n1, nullptr};
              jobs.j[4] = ScanJob{route.fin, nullptr, (uint64_t*)s.fin_idx.p, n1, nullptr};
              jobs.n_jobs = 5;
              if (try_inc) { jobs.j[5] = ScanJob{(const uint32_t*)s.nwin.p, nullptr, (uint64_t*)s.win_off.p, n1, nullptr}; jobs.n_jobs = 6; }
              HIP_TRY(launch_scan_jobs(jobs, st));
          } else {
              HIP_TRY(launch_scan64(s.scan_tmp.p, tmp2, route.len_next, (uint64_t*)s.off_next.p, n1, st));
              HIP_TRY(launch_scan64(s.scan_tmp.p, tmp2, route.len_fin, (uint64_t*)s.off_fin.p, n1, st));
              HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, route.tiles, (uint64_t*)s.tile_off.p, n1, st));
              HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, route.act, (uint64_t*)s.act_idx.p, n1, st));
              HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, route.fin, (uint64_t*)s.fin_idx.p, n1, st));
              if (try_inc) HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, (const uint32_t*)s.nwin.p, (uint64_t*)s.win_off.p, n1, st));
          } }
        if (try_inc) {
            Prof pr("rp_windows", st);
            if (!small) HIP_TRY(hipMemsetAsync(s.wlen.p, 0, (n_rec + 2) * 4, st));     // at most one window per record; unused entries scan as zeros
            HIP_TRY(launch_rp_win_meta(r->t, rt, (const RpHay*)s.hs.p, (const uint64_t*)s.rec_first.p, (const RpKept*)s.kept.p,
                                       (const uint64_t*)s.win_off.p, ov, (RpWin*)s.wins.p, (uint32_t*)s.wlen.p, n_act, st));
            if (small) {          // exactly n_win + 1 elements: the count is read on the device
                ScanJobs jobs{};
                jobs.j[0] = ScanJob{(const uint32_t*)s.wlen.p, nullptr, (uint64_t*)s.woffs.p, 1, (const uint64_t*)s.win_off.p + n_act};
                jobs.n_jobs = 1;
                HIP_TRY(launch_scan_jobs(jobs, st));
                woffs_last = ~0ull;
            } else HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, (const uint32_t*)s.wlen.p, (uint64_t*)s.woffs.p, n_rec + 1, st));
        }
        // bytes of next text, bytes of finished text, tiles, haystacks still active, haystacks finished, windows, window bytes
        HIP_TRY(launch_rp_totals(rt, n_act, try_inc ? (const uint64_t*)s.win_off.p : nullptr, try_inc ? (const uint64_t*)s.woffs.p : nullptr, woffs_last,
                                 (uint64_t*)s.totals.p, st));
        HIP_TRY(hipMemcpyAsync(s.tot_host, s.totals.p, 56, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const uint64_t* tot = s.tot_host;
        const uint64_t total_next = tot[0], total_fin = tot[1], n_tiles = tot[2], n_next = tot[3], n_fin = tot[4], n_win = tot[5], total_w = tot[6];
        t_fold += now() - t0; t0 = now();
        res->spliced += total_next + total_fin;
        if (n_tiles >= 0x7FFFFFF0ull) return fail(AM_ERR_UNSUPPORTED, "replacement output too large for one launch; split the batch");
        // ---- replace (Replacer.hs:163-180) into the next batch / the finished buffer
        AM_TRY(s.text[nxt].ensure(padded_text(total_next))); AM_TRY(s.offs[nxt].ensure((n_next + 1) * 8));
        AM_TRY(s.orig[nxt].ensure((n_next + 1) * 4)); AM_TRY(s.thr[nxt].ensure((n_next + 1) * 8));
        AM_TRY(s.fin_text.ensure(total_fin + 16)); AM_TRY(s.fin_meta.ensure((n_fin + 1) * sizeof(RpFin))); AM_TRY(s.tile_hay.ensure((n_tiles + 1) * 4));
        { Prof pr("rp_route", st);
          HIP_TRY(launch_rp_route((const RpHay*)s.hs.p, rt, cur_orig, n_act, (uint64_t*)s.offs[nxt].p, (uint32_t*)s.orig[nxt].p, (int64_t*)s.thr[nxt].p, (RpFin*)s.fin_meta.p, st)); }
        { Prof pr("rp_splice", st);
          HIP_TRY(launch_rp_splice(r->t, cur_text, cur_offs, (const uint64_t*)s.rec_first.p, (const RpKept*)s.kept.p, (const RpHay*)s.hs.p, rt, n_act, n_tiles,
                                   (uint32_t*)s.tile_hay.p, (uint8_t*)s.text[nxt].p, (uint8_t*)s.fin_text.p, st)); }
        HIP_TRY(hipMemsetAsync((uint8_t*)s.text[nxt].p + total_next, 0, padded_text(total_next) - (size_t)total_next, st));
        // ---- finished haystacks go home: the copy runs on its own stream, next to the window scans below
        uint8_t* home = nullptr;
        if (total_fin) AM_TRY(res->room((size_t)total_fin, &home));
        AM_TRY(s.pin_meta((n_fin + 1) * sizeof(RpFin)));
        HIP_TRY(hipEventRecord(s.ev_spliced, st));
        HIP_TRY(hipStreamWaitEvent(s.copy_stream, s.ev_spliced, 0));
        if (total_fin) HIP_TRY(hipMemcpyAsync(home, s.fin_text.p, total_fin, hipMemcpyDefault, s.copy_stream));      // the slab is pinned host memory, or device memory for results that stay there
        if (n_fin) HIP_TRY(hipMemcpyAsync(s.fin_host, s.fin_meta.p, n_fin * sizeof(RpFin), hipMemcpyDeviceToHost, s.copy_stream));
        auto finished_home = [&]() -> int {
            HIP_TRY(hipStreamSynchronize(s.copy_stream));
            for (uint64_t i = 0; i < n_fin; i++) {
                const RpFin& f = s.fin_host[i];
                if (f.orig >= n_hay || f.off + f.len > total_fin) return fail(AM_ERR_HIP, "replacer pass produced inconsistent metadata (internal error)");
                if (f.status == kRpNothing) res->just[f.orig] = 0;
                else res->text[f.orig] = am_replaced::Item{home + f.off, (size_t)f.len};
            }
            return AM_OK;
        };
        t_splice += now() - t0; t0 = now();
        t_home += now() - t0; t0 = now();
        // ---- next pass's records without a full scan: windows around the replacements + the shifted old records
        if (try_inc && n_next > 0 && n_win > 0 && n_win < 0xFFFFFFF0ull && total_w <= total_next / 2) {
            const uint8_t* text_next = (const uint8_t*)s.text[nxt].p;
            AM_TRY(s.wtext.ensure(padded_text(total_w)));
            { Prof pr("rp_windows", st);
              HIP_TRY(launch_rp_win_copy((const RpWin*)s.wins.p, (const uint64_t*)s.woffs.p, text_next, (uint8_t*)s.wtext.p, n_win, st));
              HIP_TRY(hipMemsetAsync((uint8_t*)s.wtext.p + total_w, 0, padded_text(total_w) - (size_t)total_w, st)); }
            s.ws2.dev = in->dev;
            s.ws2.d_text = s.wtext.p; s.ws2.d_offsets = (uint64_t*)s.woffs.p; s.ws2.owns = false; s.ws2.total = total_w; s.ws2.n_hay = (uint32_t)n_win;
            AM_TRY(finish_batch(&s.ws2));
            uint64_t n_wrec = 0;
            auto wsink = [&](uint64_t n, Record** ptr) -> int { AM_TRY(s.wrec.ensure(n * sizeof(Record))); *ptr = (Record*)s.wrec.p; return AM_OK; };
            AM_TRY(run_records(r->a, r->case_mode, &s.ws2, wsink, &n_wrec));
            res->scanned += total_w;
            AM_TRY(s.wrec.ensure(sizeof(Record)));
            AM_TRY(s.wrec_first.ensure((n_win + 1) * 8)); AM_TRY(s.mcount.ensure((n_next + 1) * 4)); AM_TRY(s.moff.ensure((n_next + 1) * 8));
            DevBuf& next_records = s.recbuf[cur_rec ^ 1];
            AM_TRY(next_records.ensure((n_rec + n_wrec + 1) * sizeof(Record)));          // upper bound; the exact count arrives with the end-of-pass sync
            Prof pr("rp_merge", st);
            HIP_TRY(launch_rp_ranges((const Record*)s.wrec.p, n_wrec, (uint64_t*)s.wrec_first.p, RpRoute{nullptr, nullptr, nullptr, nullptr, nullptr}, (uint32_t)n_win, st));
            HIP_TRY(hipMemsetAsync((uint32_t*)s.mcount.p + n_next, 0, 4, st));
            HIP_TRY(launch_rp_merge(false, (const Record*)records.p, (const uint64_t*)s.rec_first.p, (const RpKept*)s.kept.p, (const RpHay*)s.hs.p, cur_offs, rt,
                                    (const uint64_t*)s.win_off.p, (const RpWin*)s.wins.p, (const Record*)s.wrec.p, (const uint64_t*)s.wrec_first.p, ov, n_act,
                                    (uint32_t*)s.mcount.p, nullptr, nullptr, st));
            if (n_next + 1 <= (1u << 18)) {
                ScanJobs jobs{};
                jobs.j[0] = ScanJob{(const uint32_t*)s.mcount.p, nullptr, (uint64_t*)s.moff.p, n_next + 1, nullptr};
                jobs.n_jobs = 1;
                HIP_TRY(launch_scan_jobs(jobs, st));
            } else HIP_TRY(launch_scan(s.scan_tmp.p, tmp2, (const uint32_t*)s.mcount.p, (uint64_t*)s.moff.p, n_next + 1, st));
            HIP_TRY(launch_rp_merge(true, (const Record*)records.p, (const uint64_t*)s.rec_first.p, (const RpKept*)s.kept.p, (const RpHay*)s.hs.p, cur_offs, rt,
                                    (const uint64_t*)s.win_off.p, (const RpWin*)s.wins.p, (const Record*)s.wrec.p, (const uint64_t*)s.wrec_first.p, ov, n_act,
                                    (uint32_t*)s.mcount.p, (const uint64_t*)s.moff.p, (Record*)next_records.p, st));
            HIP_TRY(hipMemcpyAsync(&s.tot_host[7], (uint64_t*)s.moff.p + n_next, 8, hipMemcpyDeviceToHost, st));
            have_inc = true;
        }
        HIP_TRY(hipStreamSynchronize(st));            // end of pass: the merged record count (if any) is on the host now
        if (have_inc) inc_n_rec = s.tot_host[7];
        t_scan += now() - t0; t0 = now();
        AM_TRY(finished_home());
        t_home += now() - t0; t0 = now();
        cur_rec ^= 1;
        cur_text = (const uint8_t*)s.text[nxt].p; cur_offs = (const uint64_t*)s.offs[nxt].p;
        cur_orig = (const uint32_t*)s.orig[nxt].p; cur_thr = (const int64_t*)s.thr[nxt].p;
        total = total_next; n_act = (uint32_t)n_next; nxt ^= 1;
        t_scan += now() - t0;
    }
    return AM_OK;
}

}  // namespace

// All passes of every haystack in ONE kernel (am_rploop.hip): a wavefront takes a haystack and runs its loop to the end.  *handled = false:
// the batch is not for this path (or a haystack outgrew its regions) and nothing of `res` was touched: the caller takes the pass-by-pass paths.
static int replacer_run_loop(const am_replacer* r, const am_batch* in, uint64_t max_length, am_replaced* res, bool* handled)
{
    *handled = false;
    const uint32_t n_hay = in->n_hay;
    if (n_hay == 0 || in->dev != r->a->dev) return AM_OK;
    const long sw = cfg::get(cfg::kRpLoop);
    if (sw == 0) return AM_OK;
    const Flavor* fl = nullptr;
    AM_TRY(prepare(r->a, r->case_mode, &fl));
    if (!fl->h.sf_enabled || fl->h.root_vlen != 0 || r->a->kernel_pref == 1) return AM_OK;
    if (sw != 1) {
        // unset: batches of many documents, and no switch that asks for one of the other loops
        if (!(n_hay >= 64 && in->total / n_hay <= (1ull << 20))) return AM_OK;
        for (cfg::Key k : {cfg::kRpFullScans, cfg::kRpSplice, cfg::kRpPieces, cfg::kRpParallelFold, cfg::kRpGroups, cfg::kRpNoFuse, cfg::kRpNoRangeReuse, cfg::kRpNoSpin, cfg::kRpMatMain})
            if (cfg::get(k) != cfg::kUnset) return AM_OK;
    }
    const uint32_t ov = 4u * (fl->h.max_needle_cps ? fl->h.max_needle_cps : 1u) + 4u;
    const uint64_t wcap64 = ((2ull * ov + r->max_repl_len + 16ull) + 63ull) & ~63ull;
    if (wcap64 > 4096 || wcap64 * n_hay > (1ull << 30) || in->total >= (1ull << 40)) return AM_OK;
    ON_DEVICE(in->dev);
    hipStream_t st; AM_TRY(get_stream(in->dev, &st));
    RpSession* sp = nullptr;
    { std::lock_guard<std::mutex> lk(r->session_mu); if (!r->sessions.empty()) { sp = static_cast<RpSession*>(r->sessions.back()); r->sessions.pop_back(); } }
    if (!sp) sp = new RpSession();
    struct Return {
        const am_replacer* r; RpSession* sp;
        ~Return()
        {
            if (sp->copy_stream) (void)hipStreamSynchronize(sp->copy_stream);
            if (sp->device_bytes() > (2048ull << 20)) { delete sp; return; }
            std::vector<RpSession*> doomed;
            { std::lock_guard<std::mutex> lk(r->session_mu);
              const_cast<am_replacer*>(r)->session_delete = [](void* p) { delete static_cast<RpSession*>(p); };
              r->sessions.push_back(sp);
              for (;;) {
                  size_t held = 0;
                  for (void* q : r->sessions) held += static_cast<RpSession*>(q)->device_bytes();
                  if (r->sessions.size() <= 1 || (r->sessions.size() <= 8 && held <= (4096ull << 20))) break;
                  doomed.push_back(static_cast<RpSession*>(r->sessions.front()));
                  r->sessions.erase(r->sessions.begin());
              } }
            for (RpSession* q : doomed) delete q;
        }
    } give_back{r, sp};
    RpSession& s = *sp;
    const bool trace = cfg::on(cfg::kRpTrace);
    auto say = [&](const char* what) { if (trace) { (void)hipStreamSynchronize(st); std::fprintf(stderr, "[am_replacer loop] %s\n", what); std::fflush(stderr); } };
    // the first (and only full) scan
    say("first scan");
    uint64_t n_rec = 0;
    s.ws.dev = in->dev; s.ws.d_text = in->d_text; s.ws.d_offsets = in->d_offsets; s.ws.owns = false; s.ws.total = in->total; s.ws.n_hay = n_hay;
    AM_TRY(finish_batch(&s.ws));
    {
        auto sink = [&](uint64_t n, Record** ptr) -> int { AM_TRY(s.recbuf[0].ensure((n + 1) * sizeof(Record))); *ptr = (Record*)s.recbuf[0].p; return AM_OK; };
        AM_TRY(run_records(r->a, r->case_mode, &s.ws, sink, &n_rec));
    }
    if (n_rec >= (1ull << 26)) return AM_OK;                 // (the regions below would not fit: the pass-by-pass loop scans again)
    const uint64_t n1 = (uint64_t)n_hay + 1;
    const uint64_t rec_total = 4 * n_rec + 128ull * n_hay, pc_total = 8 * n_rec + 128ull * n_hay;      // = the sums of k_rp_loop_caps' region sizes
    AM_TRY(s.recbuf[0].ensure(sizeof(Record)));
    AM_TRY(s.rec_first.ensure(n1 * 8));
    AM_TRY(s.lp_cap_r.ensure(n1 * 4)); AM_TRY(s.lp_cap_p.ensure(n1 * 4)); AM_TRY(s.lp_rec_base.ensure(n1 * 8)); AM_TRY(s.lp_pc_base.ensure(n1 * 8));
    AM_TRY(s.lp_rec.ensure((rec_total + 1) * sizeof(Record))); AM_TRY(s.lp_pc.ensure((pc_total + 1) * sizeof(RpPiece)));
    AM_TRY(s.lp_kept.ensure((rec_total / 2 + 1) * sizeof(RpKept)));
    AM_TRY(s.lp_wtext.ensure(wcap64 * n_hay + 64)); AM_TRY(s.lp_out.ensure(n1 * sizeof(RpLoopOut))); AM_TRY(s.lp_ctrl.ensure(64));
    size_t t32 = 0;
    if (scan_temp_bytes(n1, &t32) != hipSuccess) return fail(AM_ERR_HIP, "scan sizing failed");
    AM_TRY(s.scan_tmp.ensure(t32 + 16));
    { Prof pr("rp_ranges", st);
      HIP_TRY(launch_rp_ranges((const Record*)s.recbuf[0].p, n_rec, (uint64_t*)s.rec_first.p, RpRoute{nullptr, nullptr, nullptr, nullptr, nullptr}, n_hay, st)); }
    { Prof pr("rp_scans", st);
      HIP_TRY(launch_rp_loop_caps((const uint64_t*)s.rec_first.p, n_hay, (uint32_t*)s.lp_cap_r.p, (uint32_t*)s.lp_cap_p.p, st));
      HIP_TRY(launch_scan(s.scan_tmp.p, t32, (const uint32_t*)s.lp_cap_r.p, (uint64_t*)s.lp_rec_base.p, n1, st));
      HIP_TRY(launch_scan(s.scan_tmp.p, t32, (const uint32_t*)s.lp_cap_p.p, (uint64_t*)s.lp_pc_base.p, n1, st)); }
    say("ranges + region sizes");
    HIP_TRY(hipMemsetAsync(s.lp_ctrl.p, 0, 64, st));
    RpLoop a{};
    a.t = r->t; a.s = make_sf_view(fl->d_image, fl->h);
    a.text = (const uint8_t*)in->d_text; a.offsets = in->d_offsets; a.n_hay = n_hay; a.ov = ov;
    a.recs0 = (const Record*)s.recbuf[0].p; a.rec_first0 = (const uint64_t*)s.rec_first.p;
    a.rec_buf = (Record*)s.lp_rec.p; a.rec_base = (const uint64_t*)s.lp_rec_base.p;
    a.pc_buf = (RpPiece*)s.lp_pc.p; a.pc_base = (const uint64_t*)s.lp_pc_base.p;
    a.kept_buf = (RpKept*)s.lp_kept.p; a.wtext = (uint8_t*)s.lp_wtext.p; a.wcap = (uint32_t)wcap64;
    a.max_len = max_length; a.out = (RpLoopOut*)s.lp_out.p; a.ctrl = (uint32_t*)s.lp_ctrl.p;
    say("launch");
    { Prof pr("rp_loop", st); HIP_TRY(launch_rp_loop(r->case_mode == AM_IGNORE_CASE, a, (int)cfg::get(cfg::kRpLoopWaves), st)); }
    say("launched");
    // what every haystack ended as
    const size_t out_bytes = (size_t)n_hay * sizeof(RpLoopOut);
    const size_t tab_bytes = (size_t)n_hay * (sizeof(RpFin) + 8 + 4) + 64;
    AM_TRY(s.pin_loop(64 + out_bytes + tab_bytes));
    uint32_t* ctrl_h = (uint32_t*)s.lp_host;
    RpLoopOut* out_h = (RpLoopOut*)((uint8_t*)s.lp_host + 64);
    HIP_TRY(hipMemcpyAsync(ctrl_h, s.lp_ctrl.p, 64, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_h, s.lp_out.p, out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (trace) { std::fprintf(stderr, "[am_replacer loop] kernel done: overflow %u passes %u watchdog %u\n", ctrl_h[0], ctrl_h[1], ctrl_h[5]); std::fflush(stderr); }
    if (ctrl_h[0] != 0) return AM_OK;                        // a haystack outgrew its regions: the pass-by-pass loop takes the batch
    // the finished texts: one materialise launch over the final piece lists
    RpFin* fin_h = (RpFin*)((uint8_t*)s.lp_host + 64 + out_bytes);
    uint64_t* fstart_h = (uint64_t*)(fin_h + n_hay);
    uint32_t* fcnt_h = (uint32_t*)(fstart_h + n_hay);
    uint64_t total_fin = 0;
    for (uint32_t i = 0; i < n_hay; i++) {
        const RpLoopOut& o = out_h[i];
        if (o.status > kRpNothing || o.pieces_at + o.n_pieces + 1 > pc_total) return fail(AM_ERR_HIP, "replacer loop produced inconsistent metadata (internal error)");
        fin_h[i] = RpFin{total_fin, o.len, i, o.status};
        fstart_h[i] = o.pieces_at; fcnt_h[i] = o.n_pieces;
        total_fin += o.len;
    }
    AM_TRY(s.lp_fin.ensure((size_t)n_hay * sizeof(RpFin))); AM_TRY(s.lp_fin_start.ensure((size_t)n_hay * 8)); AM_TRY(s.lp_fin_cnt.ensure((size_t)n_hay * 4));
    HIP_TRY(hipMemcpyAsync(s.lp_fin.p, fin_h, (size_t)n_hay * sizeof(RpFin), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s.lp_fin_start.p, fstart_h, (size_t)n_hay * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s.lp_fin_cnt.p, fcnt_h, (size_t)n_hay * 4, hipMemcpyHostToDevice, st));
    res->text.assign(n_hay, am_replaced::Item());
    res->just.assign(n_hay, 1);
    uint8_t* home = nullptr;
    if (total_fin) AM_TRY(res->room((size_t)total_fin, &home));
    uint8_t* d_fin = home;
    if (res->dev < 0) { AM_TRY(s.fin_text.ensure(total_fin + 16)); d_fin = (uint8_t*)s.fin_text.p; }
    { Prof pr("pt_materialise", st);
      HIP_TRY(launch_pt_materialise((const RpPiece*)s.lp_pc.p, (const uint64_t*)s.lp_fin_start.p, (const uint32_t*)s.lp_fin_cnt.p, (const RpFin*)s.lp_fin.p, n_hay,
                                    (const uint8_t*)in->d_text, r->t.repl, d_fin, st)); }
    if (res->dev < 0 && total_fin) {
        // home in requests of 256 MiB (one huge request keeps the copy engine from overlapping with anything else queued behind it)
        for (uint64_t off = 0; off < total_fin; off += (256ull << 20)) {
            const uint64_t n = std::min<uint64_t>(256ull << 20, total_fin - off);
            HIP_TRY(hipMemcpyAsync(home + off, d_fin + off, n, hipMemcpyDeviceToHost, st));
        }
    }
    say("materialise queued");
    HIP_TRY(hipStreamSynchronize(st));
    say("done");
    for (uint32_t i = 0; i < n_hay; i++) {
        if (fin_h[i].status == kRpNothing) res->just[i] = 0;
        else res->text[i] = am_replaced::Item{home + fin_h[i].off, (size_t)fin_h[i].len};
    }
    res->passes = ctrl_h[1];
    res->scanned += in->total + (((uint64_t)ctrl_h[3] << 32) | ctrl_h[2]);
    res->spliced += total_fin;
    *handled = true;
    return AM_OK;
}

// Large batches are cut into a few groups of haystacks that run the pass loop CONCURRENTLY, one host thread and HIP stream per
// group: a pass is a chain of small kernels bound by launch and dependency latency, not by throughput, so the chains of
// different groups overlap on the GPU.  The groups share nothing but the (read-only) batch text and the replacer tables.
static int replacer_run_groups(const am_replacer* r, const am_batch* in, uint64_t max_length, am_replaced* res)
{
    const uint32_t n_hay = in->n_hay;
    { bool handled = false; AM_TRY(replacer_run_loop(r, in, max_length, res, &handled)); if (handled) return AM_OK; }
    uint32_t groups = n_hay / 2048u;
    if (groups > 2) groups = 2;        // measured on config 5: 1 -> 82 ms, 2 -> 54 ms, 4 -> 77 ms, 8 -> 109 ms (the groups' kernels start to queue behind each other)
    { const long v = cfg::get(cfg::kRpGroups); if (v >= 1 && v <= 16) groups = (uint32_t)v; }
    if (groups < 2 || n_hay < groups) return replacer_run(r, in, max_length, res);
    ON_DEVICE(in->dev);
    std::vector<uint64_t> offs((size_t)n_hay + 1);
    HIP_TRY(hipMemcpy(offs.data(), in->d_offsets, offs.size() * 8, hipMemcpyDeviceToHost));
    // group boundaries at haystacks whose text starts 16-byte aligned (the scan kernels load aligned 16-byte groups)
    std::vector<uint32_t> cut(1, 0);
    for (uint32_t g = 1; g < groups; g++) {
        uint32_t h = (uint32_t)((uint64_t)n_hay * g / groups);
        while (h < n_hay && (offs[h] & 15u)) h++;
        if (h > cut.back() && h < n_hay) cut.push_back(h);
    }
    cut.push_back(n_hay);
    const size_t G = cut.size() - 1;
    if (G < 2) return replacer_run(r, in, max_length, res);
    struct Group { am_batch b; am_replaced part; int rc = AM_OK; std::string err; DevBuf offs; };
    const int res_dev = res->dev;
    std::vector<std::unique_ptr<Group>> gs;
    for (size_t g = 0; g < G; g++) {
        auto gp = std::make_unique<Group>();
        gp->part.dev = res_dev;
        const uint32_t h0 = cut[g], h1 = cut[g + 1];
        std::vector<uint64_t> sub(h1 - h0 + 1);
        for (uint32_t i = 0; i <= h1 - h0; i++) sub[i] = offs[h0 + i] - offs[h0];
        AM_TRY(gp->offs.ensure(sub.size() * 8));
        HIP_TRY(hipMemcpy(gp->offs.p, sub.data(), sub.size() * 8, hipMemcpyHostToDevice));
        gp->b.dev = in->dev; gp->b.owns = false; gp->b.d_text = (uint8_t*)in->d_text + offs[h0]; gp->b.d_offsets = (uint64_t*)gp->offs.p;
        gp->b.total = sub.back(); gp->b.n_hay = h1 - h0;
        gs.push_back(std::move(gp));
    }
    std::vector<std::thread> pool;
    // The group threads launch on their own streams.  Work the caller queued on ITS stream before this call (a producer still
    // writing the text of an am_batch_from_device batch) must come first: an event on the caller's stream, waited for by every group stream.
    hipEvent_t caller_done = nullptr;
    {
        hipStream_t caller_st; AM_TRY(get_stream(in->dev, &caller_st));
        HIP_TRY(hipEventCreateWithFlags(&caller_done, hipEventDisableTiming));
        hipError_t e = hipEventRecord(caller_done, caller_st);
        if (e != hipSuccess) { (void)hipEventDestroy(caller_done); return fail(AM_ERR_HIP, std::string("hipEventRecord: ") + hipGetErrorString(e)); }
    }
    auto work = [&](size_t g) {
        Group& x = *gs[g];
        OnDevice od(in->dev);                                   // a fresh thread's current device is 0: the group's buffers and launches belong to the batch's device
        x.rc = od.rc;
        if (x.rc == AM_OK) x.rc = finish_batch(&x.b);
        if (x.rc == AM_OK) {
            hipStream_t st;
            x.rc = get_stream(in->dev, &st);
            if (x.rc == AM_OK && hipStreamWaitEvent(st, caller_done, 0) != hipSuccess) x.rc = fail(AM_ERR_HIP, "hipStreamWaitEvent failed");
        }
        if (x.rc == AM_OK) x.rc = replacer_run(r, &x.b, max_length, &x.part);
        if (x.rc != AM_OK) x.err = am_last_error();
    };
    // every group on a thread of its own (letting the calling thread take one of them serialised the two: 73 ms instead of 38, measured)
    for (size_t g = 0; g < G; g++) {
        try { pool.emplace_back(work, g); }
        catch (const std::exception&) { work(g); }          // no thread to be had: this group runs here (nothing may throw across the C ABI)
    }
    for (auto& t : pool) t.join();
    (void)hipEventDestroy(caller_done);
    res->text.assign(n_hay, am_replaced::Item());
    res->just.assign(n_hay, 1);
    int rc = AM_OK;
    for (size_t g = 0; g < G; g++) {
        Group& x = *gs[g];
        if (x.rc != AM_OK && rc == AM_OK) rc = fail(x.rc, x.err);
        for (uint32_t i = 0; i < x.b.n_hay && i < x.part.text.size(); i++) { res->text[cut[g] + i] = x.part.text[i]; res->just[cut[g] + i] = x.part.just[i]; }
        for (const Slab& sl : x.part.slabs) res->slabs.push_back(sl);      // the result keeps the group's pinned slabs (its texts point into them)
        x.part.slabs.clear();
        res->passes = std::max(res->passes, x.part.passes); res->scanned += x.part.scanned; res->spliced += x.part.spliced;
        for (DevBuf* d : {&x.b.hidx, &x.b.unit_counts, &x.b.unit_offsets, &x.b.scan_tmp, &x.b.small, &x.b.hay_counts, &x.b.flags, &x.b.unit_first, &x.b.pool, &x.b.block_next,
                          &x.b.sparse, &x.b.dense_counts, &x.b.dense_offsets, &x.b.dense_out}) d->release();
        x.offs.release();
    }
    return rc;
}

static int replacer_run_to(const am_replacer* r, const am_batch* b, uint64_t max_length, am_replaced** out, bool on_device)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!r || !b) return fail(AM_ERR_INVALID, "null replacer or batch");
    AM_TRY(ensure_runtime());
    am_replaced* res = new am_replaced();
    if (on_device) res->dev = b->dev;
    const int rc = replacer_run_groups(r, b, max_length, res);
    if (rc != AM_OK) { delete res; return rc; }
    *out = res;
    return AM_OK;
}

extern "C" int am_replacer_run_batch(const am_replacer* r, const am_batch* b, uint64_t max_length, am_replaced** out) { return replacer_run_to(r, b, max_length, out, false); }
extern "C" int am_replacer_run_batch_device(const am_replacer* r, const am_batch* b, uint64_t max_length, am_replaced** out) { return replacer_run_to(r, b, max_length, out, true); }

extern "C" int am_replacer_run(const am_replacer* r, const am_slice* hay, size_t n_hay, uint64_t max_length, am_replaced** out)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!r) return fail(AM_ERR_INVALID, "null replacer");
    ON_DEVICE(r->a->dev);
    am_batch* b = nullptr;
    AM_TRY(am_batch_upload(hay, n_hay, &b));
    const int rc = am_replacer_run_batch(r, b, max_length, out);
    am_batch_destroy(b);
    return rc;
}

// One pass of the fold only (SURVEY 8b am_run_priority): prependMatch + makeMatch (Replacer.hs:252-274) on the device,
// sort / removeOverlap / replace stay with the caller.
static_assert(sizeof(am_prio_match) == sizeof(RpSelected) && offsetof(am_prio_match, haystack) == offsetof(RpSelected, haystack), "am_prio_match layout");

extern "C" int am_run_priority(const am_replacer* r, const am_slice* hay, size_t n_hay, const int64_t* thresholds, int64_t* best_out,
                               am_prio_match** matches_out, size_t* n_matches_out)
{
    if (!matches_out || !n_matches_out) return fail(AM_ERR_INVALID, "out pointers are null");
    *matches_out = nullptr; *n_matches_out = 0;
    if (!r) return fail(AM_ERR_INVALID, "null replacer");
    if (n_hay && (!thresholds || !best_out)) return fail(AM_ERR_INVALID, "thresholds / best_out are null");
    if (n_hay == 0) return AM_OK;
    ON_DEVICE(r->a->dev);
    am_batch* b = nullptr;
    AM_TRY(am_batch_upload(hay, n_hay, &b));
    std::unique_ptr<am_batch, void (*)(am_batch*)> guard(b, am_batch_destroy);
    hipStream_t st; AM_TRY(get_stream(b->dev, &st));
    const uint32_t n = (uint32_t)n_hay;
    const uint64_t n1 = (uint64_t)n + 1;
    DevBuf records, rec_first, kept, hs, nk, off, thr, best, out, tmp;
    struct Release { std::vector<DevBuf*> l; ~Release() { for (DevBuf* d : l) d->release(); } } rel{{&records, &rec_first, &kept, &hs, &nk, &off, &thr, &best, &out, &tmp}};
    uint64_t n_rec = 0;
    auto sink = [&](uint64_t k, Record** ptr) -> int { AM_TRY(records.ensure(k * sizeof(Record))); *ptr = (Record*)records.p; return AM_OK; };
    AM_TRY(run_records(r->a, r->case_mode, b, sink, &n_rec));
    AM_TRY(records.ensure(sizeof(Record)));
    AM_TRY(rec_first.ensure(n1 * 8)); AM_TRY(kept.ensure((n_rec + 1) * sizeof(RpKept))); AM_TRY(hs.ensure(n1 * sizeof(RpHay)));
    AM_TRY(nk.ensure(n1 * 4)); AM_TRY(off.ensure(n1 * 8)); AM_TRY(thr.ensure(n1 * 8)); AM_TRY(best.ensure(n1 * 8));
    size_t tmp_bytes = 0;
    if (scan_temp_bytes(n1, &tmp_bytes) != hipSuccess) return fail(AM_ERR_HIP, "scan sizing failed");
    AM_TRY(tmp.ensure(tmp_bytes + 16));
    HIP_TRY(hipMemcpyAsync(thr.p, thresholds, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync((uint32_t*)nk.p + n, 0, 4, st));
    RpRoute route{nullptr, nullptr, (uint32_t*)nk.p, nullptr, nullptr};
    HIP_TRY(launch_rp_ranges((const Record*)records.p, n_rec, (uint64_t*)rec_first.p, RpRoute{nullptr, nullptr, nullptr, nullptr, nullptr}, n, st));
    HIP_TRY(launch_rp_pass(r->case_mode == AM_IGNORE_CASE, r->t, (const uint8_t*)b->d_text, b->d_offsets, (const Record*)records.p, (const uint64_t*)rec_first.p,
                           (const int64_t*)thr.p, UINT64_MAX, (RpKept*)kept.p, (RpHay*)hs.p, route, n, 1u, st));
    HIP_TRY(launch_scan(tmp.p, tmp_bytes, (const uint32_t*)nk.p, (uint64_t*)off.p, n1, st));
    uint64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, (uint64_t*)off.p + n, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    AM_TRY(out.ensure((total + 1) * sizeof(RpSelected)));
    HIP_TRY(launch_rp_gather((const RpHay*)hs.p, (const uint64_t*)rec_first.p, (const RpKept*)kept.p, (const uint64_t*)off.p, (RpSelected*)out.p, (int64_t*)best.p, n, st));
    am_prio_match* host = (am_prio_match*)std::malloc((total ? total : 1) * sizeof(am_prio_match));
    if (!host) return fail(AM_ERR_OOM, "malloc(matches) failed");
    hipError_t e = hipMemcpyAsync(best_out, best.p, (size_t)n * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && total) e = hipMemcpyAsync(host, out.p, total * sizeof(am_prio_match), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { std::free(host); return fail(AM_ERR_HIP, hipGetErrorString(e)); }
    *matches_out = host; *n_matches_out = (size_t)total;
    return AM_OK;
}

extern "C" void am_prio_matches_free(am_prio_match* m) { std::free(m); }

extern "C" uint64_t am_replaced_size(const am_replaced* r) { return r ? r->text.size() : 0; }
extern "C" uint64_t am_replaced_passes(const am_replaced* r) { return r ? r->passes : 0; }
extern "C" uint64_t am_replaced_scanned_bytes(const am_replaced* r) { return r ? r->scanned : 0; }
extern "C" uint64_t am_replaced_spliced_bytes(const am_replaced* r) { return r ? r->spliced : 0; }

extern "C" int am_replaced_get(const am_replaced* r, size_t i, const uint8_t** ptr, size_t* len)
{
    if (!r || i >= r->text.size() || !ptr || !len) return fail(AM_ERR_INVALID, "bad argument");
    *ptr = r->text[i].p ? r->text[i].p : (const uint8_t*)""; *len = r->text[i].len;
    return r->just[i] ? 1 : 0;
}

extern "C" int am_replaced_device(const am_replaced* r) { return r ? r->dev : -1; }

// copies text i to host memory, wherever the result lives
extern "C" int am_replaced_read(const am_replaced* r, size_t i, uint8_t* dst, size_t cap, size_t* len)
{
    if (!r || i >= r->text.size()) return fail(AM_ERR_INVALID, "index out of range");
    if (len) *len = r->just[i] ? r->text[i].len : 0;
    if (!r->just[i]) return 0;
    const size_t n = r->text[i].len;
    if (n > cap || (n && !dst)) return fail(AM_ERR_INVALID, "destination too small");
    if (n == 0) return 1;
    if (r->dev < 0) { std::memcpy(dst, r->text[i].p, n); return 1; }
    ON_DEVICE(r->dev);
    HIP_TRY(hipMemcpy(dst, r->text[i].p, n, hipMemcpyDeviceToHost));
    return 1;
}

extern "C" void am_replaced_free(am_replaced* r) { delete r; }

// ------------------------------------------------------------------ Searcher.containsAll (Searcher.hs:167-187)

struct am_needle_ids {
    const am_automaton* a = nullptr;
    uint32_t n_needles = 0;
    DevBuf vals_off, vals;
};

extern "C" int am_needle_ids_create(const am_automaton* a, const uint64_t* values_offsets, const uint32_t* values, uint32_t n_needles, am_needle_ids** out)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!a) return fail(AM_ERR_INVALID, "null automaton");
    AM_TRY(ensure_runtime());
    ON_DEVICE(a->dev);
    // a handle attached to a received image (multi-GPU ranks) has no reference arrays: the state count comes from the image
    uint64_t n_states = 0;
    if (a->has_ref) n_states = a->offsets.size() - 1;
    else {
        std::lock_guard<std::mutex> lk(const_cast<am_automaton*>(a)->mu);
        for (const Flavor& f : a->fl) if (f.ready) n_states = f.h.n_states;
        if (!n_states) return fail(AM_ERR_INVALID, "automaton handle has no image");
    }
    if (!values_offsets || values_offsets[0] != 0) return fail(AM_ERR_INVALID, "values_offsets[0] must be 0");
    for (uint64_t s = 0; s < n_states; s++)
        if (values_offsets[s + 1] < values_offsets[s] || (a->has_ref && values_offsets[s + 1] - values_offsets[s] != a->values_len[s]))
            return fail(AM_ERR_INVALID, "values_offsets disagrees with the values_len given to am_automaton_create");
    const uint64_t n_values = values_offsets[n_states];
    if (n_values && !values) return fail(AM_ERR_INVALID, "values is null");
    am_needle_ids* ids = new am_needle_ids();
    ids->a = a; ids->n_needles = n_needles;
    int rc = ids->vals_off.ensure((n_states + 1) * 8);
    if (rc == AM_OK) rc = ids->vals.ensure(n_values * 4 + 4);
    if (rc == AM_OK && hipMemcpy(ids->vals_off.p, values_offsets, (n_states + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) rc = fail(AM_ERR_HIP, "upload failed");
    if (rc == AM_OK && n_values && hipMemcpy(ids->vals.p, values, n_values * 4, hipMemcpyHostToDevice) != hipSuccess) rc = fail(AM_ERR_HIP, "upload failed");
    if (rc != AM_OK) { am_needle_ids_destroy(ids); return rc; }
    *out = ids;
    return AM_OK;
}

extern "C" void am_needle_ids_destroy(am_needle_ids* ids)
{
    if (!ids) return;
    ids->vals_off.release(); ids->vals.release();
    delete ids;
}

extern "C" int am_contains_all_batch(const am_needle_ids* ids, int case_mode, const am_batch* cb, uint8_t* flags_out)
{
    if (!ids || !cb) return fail(AM_ERR_INVALID, "null needle ids or batch");
    am_batch* b = const_cast<am_batch*>(cb);
    const uint32_t n_hay = b->n_hay;
    if (n_hay && !flags_out) return fail(AM_ERR_INVALID, "flags_out is null");
    if (ids->n_needles == 0) { if (n_hay) std::memset(flags_out, 1, n_hay); return AM_OK; }     // IS.null of the empty set (Searcher.hs:176,184)
    if (n_hay == 0) return AM_OK;
    ON_DEVICE(b->dev);
    hipStream_t st; AM_TRY(get_stream(b->dev, &st));
    DevBuf records, rec_first, bits, flags;
    struct Release { DevBuf &a, &b, &c, &d; ~Release() { a.release(); b.release(); c.release(); d.release(); } } rel{records, rec_first, bits, flags};
    uint64_t n_rec = 0;
    auto sink = [&](uint64_t n, Record** ptr) -> int { AM_TRY(records.ensure(n * sizeof(Record))); *ptr = (Record*)records.p; return AM_OK; };
    AM_TRY(run_records(ids->a, case_mode, b, sink, &n_rec));
    if (n_rec == 0) { std::memset(flags_out, 0, n_hay); return AM_OK; }
    const uint32_t words = (ids->n_needles + 31) / 32;
    // one bitmap row per haystack; very wide batches go through in groups of haystacks (records are sorted by haystack)
    const uint64_t budget = 1ull << 30;
    const uint32_t group = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_hay, budget / ((uint64_t)words * 4)));
    AM_TRY(rec_first.ensure(((uint64_t)n_hay + 1) * 8));
    AM_TRY(bits.ensure((uint64_t)group * words * 4));
    AM_TRY(flags.ensure(n_hay));
    HIP_TRY(launch_rp_ranges((const Record*)records.p, n_rec, (uint64_t*)rec_first.p, RpRoute{nullptr, nullptr, nullptr, nullptr, nullptr}, n_hay, st));
    std::vector<uint64_t> first;
    if (group < n_hay) {
        first.resize((size_t)n_hay + 1);
        HIP_TRY(hipMemcpyAsync(first.data(), rec_first.p, first.size() * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    for (uint32_t h0 = 0; h0 < n_hay; h0 += group) {
        const uint32_t h1 = std::min<uint64_t>(n_hay, (uint64_t)h0 + group);
        const uint64_t r0 = first.empty() ? 0 : first[h0], r1 = first.empty() ? n_rec : first[h1];
        HIP_TRY(hipMemsetAsync(bits.p, 0, (uint64_t)(h1 - h0) * words * 4, st));
        { Prof pr("idset", st);
          HIP_TRY(launch_idset((const Record*)records.p, r0, r1, (const uint64_t*)ids->vals_off.p, (const uint32_t*)ids->vals.p, ids->n_needles, h0, words, (uint32_t*)bits.p, st));
          HIP_TRY(launch_idset_all((const uint32_t*)bits.p, words, ids->n_needles, h1 - h0, (uint8_t*)flags.p + h0, st)); }
    }
    HIP_TRY(hipMemcpyAsync(flags_out, flags.p, n_hay, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return AM_OK;
}

extern "C" int am_matches_fold_hash(const am_matches* m, const am_needle_ids* ids, size_t n_hay, uint64_t* hash_out, uint64_t* count_out)
{
    if (!m || !ids) return fail(AM_ERR_INVALID, "null matches or values table");
    if (n_hay && !hash_out) return fail(AM_ERR_INVALID, "hash_out is null");
    if (n_hay >= 0xFFFFFFFFull) return fail(AM_ERR_INVALID, "too many haystacks");
    if (n_hay == 0) return AM_OK;
    AM_TRY(ensure_runtime());
    if (m->dev != ids->a->dev) return fail(AM_ERR_INVALID, "result and values table live on different devices");
    ON_DEVICE(m->dev);
    hipStream_t st; AM_TRY(get_stream(m->dev, &st));
    DevBuf rec_first, out, dummy;
    struct Release { DevBuf &a, &b, &c; ~Release() { a.release(); b.release(); c.release(); } } rel{rec_first, out, dummy};
    AM_TRY(rec_first.ensure((n_hay + 1) * 8));
    AM_TRY(out.ensure(n_hay * 16));
    AM_TRY(dummy.ensure(sizeof(Record)));
    const Record* recs = m->n ? m->d_records + m->first : (const Record*)dummy.p;
    HIP_TRY(launch_rp_ranges(recs, m->n, (uint64_t*)rec_first.p, RpRoute{nullptr, nullptr, nullptr, nullptr, nullptr}, (uint32_t)n_hay, st));
    { Prof pr("fold_hash", st);
      HIP_TRY(launch_fold_hash(recs, (const uint64_t*)rec_first.p, (const uint64_t*)ids->vals_off.p, (const uint32_t*)ids->vals.p, (uint32_t)n_hay,
                               (uint64_t*)out.p, (uint64_t*)out.p + n_hay, st)); }
    HIP_TRY(hipMemcpyAsync(hash_out, out.p, n_hay * 8, hipMemcpyDeviceToHost, st));
    if (count_out) HIP_TRY(hipMemcpyAsync(count_out, (uint64_t*)out.p + n_hay, n_hay * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return AM_OK;
}

extern "C" int am_contains_all(const am_needle_ids* ids, int case_mode, const am_slice* hay, size_t n_hay, uint8_t* flags_out)
{
    if (!ids) return fail(AM_ERR_INVALID, "null needle ids");
    ON_DEVICE(ids->a->dev);
    am_batch* b = nullptr;
    AM_TRY(am_batch_upload(hay, n_hay, &b));
    const int rc = am_contains_all_batch(ids, case_mode, b, flags_out);
    am_batch_destroy(b);
    return rc;
}
