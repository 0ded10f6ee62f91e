// am_abi.cpp -- the C ABI of include/am.h: handles, device memory, launch orchestration.
// There is deliberately no CPU execution path here: every run entry point needs a HIP device.
#include "am_host.h"

#include <condition_variable>
#include <deque>
#include <sys/mman.h>

using namespace am;
using namespace am::dev;
using namespace am::host;

static_assert(sizeof(am_match) == sizeof(Record), "am_match must mirror the device record");
static_assert(offsetof(am_match, end_pos) == offsetof(Record, end_pos) && offsetof(am_match, haystack) == offsetof(Record, haystack) &&
                  offsetof(am_match, state) == offsetof(Record, state), "am_match layout");

// ------------------------------------------------------------------ errors, runtime

static thread_local std::string g_err;
namespace am { int abi_fail(int code, const std::string& msg) { g_err = msg; return code; } }      // (am_host.h's fail(); am_multi.cpp uses it too: one thread-local message)

namespace am {
namespace host {

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

}  // namespace host
}  // namespace am

namespace {

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

}  // namespace

namespace am {
namespace host {
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
}  // namespace host
}  // namespace am

namespace {

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

}  // namespace


// The record array of the last freed result is kept for the next call (one buffer, reused when it is large enough
// and not more than twice what is needed): a caller that scans batch after batch does not pay hipMalloc/hipFree of
// hundreds of megabytes per call.
namespace {
// Device arrays of freed results, kept for the next ones: up to four (a segmented am_run has three in flight), 16 GiB in all; the oldest goes first.  (One block until
// round 6.  Every hipFree of a large array is followed by amdgpu's wipe of the freed VRAM on the SDMA engines, which halves the rate of the copies to the host for about a
// second -- tools/experiments/host_results/README.md -- so arrays that will be wanted again in a moment are not freed.)
struct RecordCache {
    struct Block { void* p; size_t cap; };
    std::mutex mu; std::vector<Block> kept;
    void* take(size_t need, size_t* cap_out)
    {
        std::lock_guard<std::mutex> lk(mu);
        size_t best = kept.size();
        for (size_t i = 0; i < kept.size(); i++)
            if (kept[i].cap >= need && kept[i].cap <= 2 * need + (1u << 20) && (best == kept.size() || kept[i].cap < kept[best].cap)) best = i;
        if (best == kept.size()) return nullptr;
        void* r = kept[best].p; *cap_out = kept[best].cap;
        kept.erase(kept.begin() + (std::ptrdiff_t)best);
        return r;
    }
    void trim()
    {
        std::vector<Block> out;
        { std::lock_guard<std::mutex> lk(mu); out.swap(kept); }
        for (const Block& b : out) (void)hipFree(b.p);
    }
    void give(void* q, size_t c)
    {
        std::vector<Block> out;
        {
            std::lock_guard<std::mutex> lk(mu);
            kept.push_back(Block{q, c});
            size_t total = 0;
            for (const Block& b : kept) total += b.cap;
            while (kept.size() > 4 || (kept.size() > 1 && total > ((size_t)16 << 30))) { total -= kept.front().cap; out.push_back(kept.front()); kept.erase(kept.begin()); }
        }
        for (const Block& b : out) (void)hipFree(b.p);
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

uint64_t am::host::next_image_generation() { static std::atomic<uint64_t> g{0}; return g.fetch_add(1, std::memory_order_relaxed) + 1u; }

int am::host::prepare(const am_automaton* ca, int case_mode, const Flavor** out)
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
        else if (case_mode == AM_IGNORE_CASE && a->ic_pending.valid()) {
            am_automaton::Flat fl = a->ic_pending.get();                // made since am_automaton_create, on a thread of its own
            if (fl.rc != 0) return fail(AM_ERR_INVALID, fl.err);
            img.swap(fl.img);
        }
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
        f.d_image = d; f.bytes = img.size(); f.generation = next_image_generation(); f.ready = true;
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
    // validate on the host right away (flatten checks every index); the images are uploaded on first use.  The handle's own copy of the arrays is what both
    // flattens read: the IgnoreCase one runs on a thread of its own while this thread makes (and thereby validates) the CaseSensitive one.
    std::unique_ptr<am_automaton> a(new am_automaton());
    a->lower = lower;
    a->transitions.assign(transitions, transitions + n_transitions);
    a->offsets.assign(offsets, offsets + n_states + 1);
    a->root_ascii.assign(root_ascii, root_ascii + 128);
    a->values_len.assign(values_len, values_len + n_states);
    const RefArrays ref{a->transitions.data(), n_transitions, a->offsets.data(), n_states, a->root_ascii.data(), a->values_len.data()};
    const LowerTable* lt = lower.get();
    if (!cfg::on(cfg::kFlattenSerial)) try {
        a->ic_pending = std::async(std::launch::async, [ref, lt] { am_automaton::Flat f; f.rc = flatten(ref, AM_IGNORE_CASE, f.img, f.err, lt); return f; });
    } catch (const std::system_error&) { /* no thread to be had: prepare() flattens the IgnoreCase image when it is asked for */ }
    {
        std::string err;
        if (flatten(ref, AM_CASE_SENSITIVE, a->cs_image, err, lt) != 0) {
            if (a->ic_pending.valid()) a->ic_pending.wait();          // (it reads the arrays of the handle that is about to go)
            return fail(AM_ERR_INVALID, err);
        }
    }
    a->has_ref = true;
    // the automaton belongs to the device that is current now (its images are uploaded there on first use); without a
    // device the handle can still be made and inspected, every run entry point then fails with AM_ERR_NO_DEVICE
    { int d = 0; if (current_device(&d) == AM_OK) a->dev = d; else g_err.clear(); }
    *out = a.release();
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
    if (a->has_ref) return a->lower ? a->lower->hash : builtin_lower_table().hash;   // immutable after creation: no lock
    std::lock_guard<std::mutex> lk(const_cast<am_automaton*>(a)->mu);                 // prepare() writes fl[] under this lock
    for (const Flavor& f : a->fl) if (f.ready) return f.h.flags;                      // attached to an image: what the image says
    return builtin_lower_table().hash;
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
    if (a->ic_pending.valid()) a->ic_pending.wait();                      // (the flatten thread reads the handle's arrays)
    for (Flavor& f : a->fl) if (f.d_image) (void)hipFree(f.d_image);      // hipFree finds the owning device itself
    delete a;
}

extern "C" int am_automaton_set_kernel(am_automaton* a, int k)
{
    if (!a || k < 0 || k > 3) return fail(AM_ERR_INVALID, "bad arguments");
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
    f.h = h; f.d_image = d; f.bytes = h.total_bytes; f.generation = next_image_generation(); f.ready = true;
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
    f.h = h; f.d_image = d; f.bytes = h.total_bytes; f.generation = next_image_generation(); f.ready = true;
    *out = a;
    return AM_OK;
}

// ------------------------------------------------------------------ batches

int am::host::finish_batch(am_batch* b)
{
    b->hidx_ready = false;
    b->route_image = 0;                        // new text: the route is asked again
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
    // What a thread's one-shot batch may keep between calls: a sixteenth of the device's memory (18 GB of 288), at least 256 MiB.  (Until round 6: 256 MiB -- every
    // large call then freed and re-allocated its text and pools, and amdgpu wipes freed VRAM through the SDMA engines that the records' copy to the host uses
    // right afterwards: 29 GB/s instead of 50, tools/experiments/host_results/README.md.)
    const size_t held = b->text_buf.cap + b->pool.cap + b->hidx.cap + b->unit_offsets.cap + b->hay_counts.cap;
    if (held <= ((size_t)256 << 20)) return;
    static std::atomic<size_t> keep[kMaxDev];
    size_t limit = keep[dev].load(std::memory_order_relaxed);
    if (limit == 0) {
        size_t free_b = 0, total_b = 0;
        OnDevice od(dev);
        limit = (od.rc == AM_OK && hipMemGetInfo(&free_b, &total_b) == hipSuccess) ? std::max<size_t>(total_b / 16, (size_t)256 << 20) : (size_t)256 << 20;
        keep[dev].store(limit, std::memory_order_relaxed);
    }
    if (held > limit) { am_batch_destroy(b); b = nullptr; }
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

// k_ac's launcher, handed over by libam_check.so when a test or bench.py's parity gate loads it (am_debug_set_general_kernel); never set in a product process
using AcLauncher = hipError_t (*)(bool ic, int mode, const AcView& a, const BatchView& b, const ScanOut& o, hipStream_t st);
std::atomic<AcLauncher> g_ac_launcher{nullptr};

constexpr uint64_t kDfaMinBytes = 32ull << 20;     // below this the suffix-filter route is the faster one even on natural text: a lane's walk of its unit (>= 128 bytes + warm-up, ~1 us per step) has a floor of 0.3 ms (natural text, 16 MiB: k_sf 0.26 / 0.36 ms counting / emitting, k_dfa 0.29 / 0.41; 32 MiB: 0.44 / 0.57 against 0.33 / 0.47)
constexpr uint64_t kDfaSampleBytes = 32ull << 20;   // from here on -- i.e. whenever the table walk is in question -- a sample walk asks the text which route it wants (round 5 asked from
                                                    // 64 MiB on: a dictionary over text that is not its language took the table walk at 32-64 MiB, 6 x slower than the filter there)
constexpr uint32_t kDfaEndsPerKiB = 48;            // needle ends per KiB from which the table walk wins (k_sf: 670 GiB/s at 10 per KiB, 215 at 63, 76 at 156; k_dfa: ~155 flat)

struct Plan {
    const Flavor* f; bool ic; bool use_sf; bool nothing; uint64_t n_units; uint32_t unit_chunks; int n_cu;
    bool use_dfa;        // the table-walk kernel (am_dfa.hip) on the general route's two passes; never together with use_sf
    DfaView dfa;
    bool dense;          // automaton with the empty needle on the suffix-filter route: k_sf's records + the dense pass (am_dense.hip)
    AcView ac; SfView sf; BatchView bv;
    am_batch* batch;
    uint32_t* next_unit; // k_sf's unit counter (in the batch's `small` block: [0..1] total_values, [4] block counter, [5] overflow, [8] this)
};

// allow_dfa: the caller's route works for the general two-pass protocol (am_count_batch, am_contains_any_batch, run_records)
// have_lock: the caller holds b->mu already (run_records under reduce_dense)
int make_plan(const am_automaton* a, int case_mode, am_batch* b, Plan& p, bool allow_dfa = false, bool have_lock = false)
{
    if (!b) return fail(AM_ERR_INVALID, "null batch");
    if (!a) return fail(AM_ERR_INVALID, "null automaton");
    if (a->dev != b->dev) return fail(AM_ERR_INVALID, "automaton and batch live on different devices");
    AM_TRY(prepare(a, case_mode, &p.f));
    p.ic = case_mode == AM_IGNORE_CASE;
    if (a->kernel_pref == 2 && !p.f->h.sf_enabled) return fail(AM_ERR_UNSUPPORTED, "suffix-filter kernel cannot run this automaton (empty needle with too many prefix terminals)");
    p.dfa = make_dfa_view(p.f->d_image, p.f->h);
    bool has_dfa = p.f->h.dfa_n_states != 0 && p.f->h.root_vlen == 0;
    if (a->kernel_pref == 3 && !has_dfa) return fail(AM_ERR_UNSUPPORTED, "am_automaton_set_kernel(a, 3): this automaton's image has no DFA section");
    if (has_dfa) {
        ON_DEVICE(b->dev);
        if (!dfa_usable(p.dfa)) {                            // (a section this device cannot walk -- its LDS attribute refused, offsets beyond 32 bits -- is no error: the filter takes the batch)
            if (a->kernel_pref == 3) return fail(AM_ERR_UNSUPPORTED, "am_automaton_set_kernel(a, 3): the table-walk kernel cannot run this image on this device");
            has_dfa = false;
        }
    }
    if (has_dfa) {
        // A lane walks its unit byte after byte (~1 us per step): a unit of 2 048 bytes takes milliseconds however small the batch is.  The image's unit is for batches that
        // fill the machine (n_cu x 32 wavefronts x 64 lanes) with it; smaller batches get smaller units, down to 128 bytes (where the warm-up is a third of the walk).
        const long forced = cfg::get(cfg::kDfaChunk);
        if (forced < 64) {
            const uint64_t lanes = (uint64_t)g_rt.dev[b->dev].n_cu * 32u * 64u;
            uint64_t unit = ((b->total / (lanes ? lanes : 1)) + 15u) & ~15ull;
            if (unit < 128) unit = 128;
            if (unit < 4ull * p.dfa.warm) unit = (4ull * p.dfa.warm + 15u) & ~15ull;
            if (unit < p.dfa.chunk) p.dfa.chunk = (uint32_t)unit;
        }
    }
    p.use_dfa = has_dfa && allow_dfa && (a->kernel_pref == 3 || (a->kernel_pref == 0 && b->total >= (cfg::get(cfg::kDfaMinKiB) >= 0 ? (uint64_t)cfg::get(cfg::kDfaMinKiB) << 10 : kDfaMinBytes) && cfg::get(cfg::kDfa) != 0));
    if (p.use_dfa && a->kernel_pref == 0 && b->total >= kDfaSampleBytes) {
        // The table walk costs the same whatever the text is; the suffix filter is 6 x faster where needles are rare and slower where one ends every few bytes.  A large batch
        // is asked: 4 096 lanes spread over it walk 128 bytes each (0.15 ms); below kDfaEndsPerKiB needle ends per KiB the filter takes it.  Decided once per batch and image.
        std::unique_lock<std::mutex> lk(b->mu, std::defer_lock);
        if (!have_lock) lk.lock();
        if (b->route_image != p.f->generation) {
            ON_DEVICE(b->dev);
            hipStream_t st; AM_TRY(get_stream(b->dev, &st));
            AM_TRY(b->small.ensure(64));
            HIP_TRY(hipMemsetAsync(b->small.p, 0, 64, st));
            const uint32_t kSamples = b->total >= (64ull << 20) ? 4096u : 1024u, kLen = 128;       // (a smaller batch is asked with fewer lanes: 0.05 ms of a 0.3-ms scan)
            HIP_TRY(launch_dfa_sample(p.dfa, (const uint8_t*)b->d_text, b->total, kSamples, kLen, (uint32_t*)b->small.p, st));
            uint32_t ends = 0;
            HIP_TRY(hipMemcpyAsync(&ends, b->small.p, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            b->route_dfa = (uint64_t)ends * 1024u >= (uint64_t)kDfaEndsPerKiB * kSamples * kLen;
            b->route_ends_per_kib = (uint32_t)(((uint64_t)ends * 1024u + (uint64_t)kSamples * kLen - 1u) / ((uint64_t)kSamples * kLen));
            b->route_image = p.f->generation;
        }
        p.use_dfa = b->route_dfa;
    }
    p.use_sf = p.f->h.sf_enabled && a->kernel_pref != 1 && !p.use_dfa;
    if (!p.use_sf && !p.use_dfa && !g_ac_launcher.load(std::memory_order_acquire))
        return fail(AM_ERR_UNSUPPORTED, a->kernel_pref == 1 ? "am_automaton_set_kernel(a, 1): the general AC kernel is test infrastructure (libam_check.so) and is not loaded in this process"
                                                           : "this image has no suffix-filter section (sf_enabled == 0) and no table-walk section this batch could take: no kernel of the library can scan with it");
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
    p.n_units = p.nothing ? 0 : (p.use_sf ? (sf_chunks(p.bv) + p.unit_chunks - 1) / p.unit_chunks : p.use_dfa ? dfa_units(p.dfa, p.bv) : ac_units(p.ac, p.bv));
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
    else if (p.use_dfa) {
        Prof pr("dfa", st);
        HIP_TRY(launch_dfa(mode, p.dfa, p.bv, o, p.n_cu, st));
    }
    else {
        // the general AC-walk kernel is test infrastructure (libam_check.so, tests/native/am_ac.hip): make_plan refused the scan if it is not loaded
        const AcLauncher ac = g_ac_launcher.load(std::memory_order_acquire);
        if (!ac) return fail(AM_ERR_UNSUPPORTED, "the general AC kernel is not loaded");
        Prof pr("ac", st);
        HIP_TRY(ac(p.ic, mode, p.ac, p.bv, o, st));
    }
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
    Plan p; AM_TRY(make_plan(a, case_mode, b, p, true));
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
    Plan p; AM_TRY(make_plan(a, case_mode, b, p, true));
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

int am::host::scan_needle_ids(const am_automaton* a, int case_mode, am_batch* b, const uint64_t* d_vals_off, const uint32_t* d_vals, uint32_t n_needles,
                              uint32_t* d_bits, uint32_t words, uint32_t* d_missing, uint8_t* flags_out, bool* taken)
{
    *taken = false;
    Plan p; AM_TRY(make_plan(a, case_mode, b, p));
    if (p.nothing || p.dense || !p.use_sf) return AM_OK;
    std::lock_guard<std::mutex> lk(b->mu);
    ON_DEVICE(b->dev);
    hipStream_t st; AM_TRY(get_stream(b->dev, &st));
    AM_TRY(b->flags.ensure(((size_t)b->n_hay + 3) & ~(size_t)3));
    AM_TRY(b->small.ensure(64));
    ScanOut o{};
    o.unit_chunks = p.unit_chunks;
    o.flags = (uint8_t*)b->flags.p;
    o.ids_vals_off = d_vals_off; o.ids_vals = d_vals; o.ids_bits = d_bits; o.ids_missing = d_missing; o.ids_words = words; o.ids_n = n_needles;
    HIP_TRY(hipMemsetAsync(d_bits, 0, (size_t)b->n_hay * words * 4, st));
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)d_missing, (int)n_needles, b->n_hay, st));
    AM_TRY(build_hidx_and_clear(p, b, st, b->flags.p, ((size_t)b->n_hay + 3) & ~(size_t)3, b->small.p, 64));      // (the counter block: k_sf's unit ticket)
    AM_TRY(launch_scan_kernel(p, kModeIds, o, st));
    ResultCopies rc;
    AM_TRY(rc.add(flags_out, b->flags.p, b->n_hay, st));
    AM_TRY(rc.finish(st));
    *taken = true;
    return AM_OK;
}

// The whole scan: leaves every record of the batch, sorted by (haystack, end_pos), in device memory
// obtained from `sink(total, &ptr)` (called once, only when total > 0); *n_out = number of records.
int am::host::run_records(const am_automaton* a, int case_mode, am_batch* b, const std::function<int(uint64_t, Record**)>& sink_final, uint64_t* n_out, bool have_lock)
{
    *n_out = 0;
    Plan p; AM_TRY(make_plan(a, case_mode, b, p, true, have_lock));
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
    // table-walk kernel: ONE walk drops a token per match into the pool (superblocks per wavefront, any order) and counts per unit; scan(unit_counts) + k_dfa_place
    // put token (unit, seq) where its record belongs.  The pool is sized by a guess (a record per 6 haystack bytes: the density these automata are made for);
    // if it is exhausted the counts are still exact and the walk is repeated once with the pool they ask for.
    auto body_dfa = [&]() -> int {
        const uint32_t n_waves = dfa_token_waves(p.dfa, p.bv, p.n_cu);
        const uint64_t sb_bytes = dfa_superblock_bytes();
        // (the guess stays below 32 GiB of pool; a batch that needs more finds out with exact counts in hand, and one that needs more than the device has left
        // takes the plain count -> scan -> emit protocol, which needs no pool)
        constexpr uint64_t kFirstPoolBytes = 16ull << 30;
        // (the sample walk of make_plan has counted the needle ends of this batch: a quarter above its estimate; a batch too small to have been asked: a record per 6 bytes)
        const uint64_t guess = b->route_image == p.f->generation && b->route_ends_per_kib ? (b->total >> 10) * b->route_ends_per_kib * 5u / 4u + 4096u : b->total / 6u;
        uint64_t want = dfa_token_superblocks(guess, n_waves, p.n_units);
        if (want * sb_bytes > kFirstPoolBytes) want = std::max<uint64_t>(kFirstPoolBytes / sb_bytes, (uint64_t)n_waves + 16);
        if (b->pool.cap / sb_bytes > want) want = b->pool.cap / sb_bytes;
        if (cfg::get(cfg::kSfPoolBlocks) > 0) want = (uint64_t)cfg::get(cfg::kSfPoolBlocks);       // tests: force the exhausted-pool path
        for (int attempt = 0; attempt < 3; attempt++) {
            if (want >= (1ull << 31)) return body_ac();
            if (want * sb_bytes > b->pool.cap) {
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || want * sb_bytes + (want * sb_bytes) / 8 + (1ull << 30) > (uint64_t)free_b + b->pool.cap) return body_ac();
            }
            AM_TRY(b->pool.ensure(want * sb_bytes));
            AM_TRY(b->block_next.ensure(2 * want * sizeof(uint32_t)));
            ScanOut o{};
            o.unit_counts = (uint32_t*)b->unit_counts.p;
            o.pool = (Record*)b->pool.p;
            o.block_next = (uint32_t*)b->block_next.p;           // here: tokens in each superblock, then each superblock's first group
            o.pool_ctrl = (uint32_t*)b->small.p + 4;             // small: [4] superblocks drawn, [5] pool exhausted
            o.n_blocks = (uint32_t)want;
            HIP_TRY(hipMemsetAsync(b->small.p, 0, 64, st));
            HIP_TRY(hipMemsetAsync(b->block_next.p, 0, want * sizeof(uint32_t), st));
            HIP_TRY(hipMemsetAsync((uint32_t*)b->unit_counts.p + p.n_units, 0, sizeof(uint32_t), st));
            AM_TRY(build_hidx(p, b, st));
            { Prof pr("dfa", st); HIP_TRY(launch_dfa_tokens(p.dfa, p.bv, o, p.n_cu, st)); }
            { Prof pr("scan", st); HIP_TRY(launch_scan(b->scan_tmp.p, tmp_bytes, (const uint32_t*)b->unit_counts.p, (uint64_t*)b->unit_offsets.p, n, st)); }
            uint64_t total = 0; uint32_t ctrl[2] = {0, 0};
            HIP_TRY(hipMemcpyAsync(&total, (uint64_t*)b->unit_offsets.p + p.n_units, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(ctrl, o.pool_ctrl, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (ctrl[1]) { want = dfa_token_superblocks(total, n_waves, p.n_units); continue; }
            *n_scan = total;
            if (total == 0) return AM_OK;
            AM_TRY(sink(total, &d_records));
            { Prof pr("dfa_place", st); HIP_TRY(launch_dfa_place(p.dfa, p.bv, o, ctrl[0] < o.n_blocks ? ctrl[0] : o.n_blocks, (const uint64_t*)b->unit_offsets.p, p.n_cu, n_waves, p.f->h.n_states, d_records, st)); }
            HIP_TRY(hipStreamSynchronize(st));
            return AM_OK;
        }
        return fail(AM_ERR_HIP, "token pool exhausted repeatedly (internal error)");
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
    if (!p.dense) return p.use_sf ? body_sf() : (p.use_dfa && dfa_tokens_ok(p.dfa)) ? body_dfa() : body_ac();
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

// The suffix-filter scan of a (small) batch WITHOUT a host round trip: the record pool is sized for the worst case -- a record at
// every byte -- so the pass cannot overflow and needs no retry; the sorted records go to d_out (room for b->total records), their
// number stays on the device (*n_dev points at it).  Used between Replacer passes, where a sync per scan would cost more than
// the scan.
int am::host::run_records_async(const am_automaton* a, int case_mode, am_batch* b, Record* d_out, const uint64_t** n_dev, hipStream_t st)
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

static int run_records_small(const am_automaton* a, int case_mode, am_batch* b, am_matches* m, bool* done)
{
    *done = false;
    if (b->total == 0 || b->total > kSmallRunBytes || a->kernel_pref == 3) return AM_OK;
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

static int run_segmented(const am_automaton* a, int case_mode, const am_slice* hay, size_t n_hay, uint64_t total, am_matches** out);
constexpr uint64_t kRunSegmentedFrom = 1ull << 30;        // host batches from here on are scanned in segments whose records travel back while the next segment goes up
constexpr uint64_t kRunSegment = 256ull << 20;

extern "C" int am_run(const am_automaton* a, int case_mode, const am_slice* hay, size_t n_hay, am_matches** out)
{
    if (!a) return fail(AM_ERR_INVALID, "null automaton");
    if (n_hay >= 2 && hay && out && cfg::get(cfg::kRunSegments) != 0) {
        uint64_t total = 0;
        bool sound = true;
        for (size_t i = 0; i < n_hay && sound; i++) { sound = !(hay[i].len && !hay[i].ptr); total += hay[i].len; }
        // (only automata whose image carries a DFA section -- dictionaries, small automata: the ones that meet match-dense text; for the others a result is a few per
        // cent of its text, the upload is the bound, and a first segment scanned apart costs 4-15 % of the call: 46 against 48 GiB/s on cfg2's 2 GiB, measured)
        const Flavor* f = nullptr;
        if (sound && n_hay < 0xFFFFFFFFull && (cfg::get(cfg::kRunSegments) > 0 || (total >= kRunSegmentedFrom && prepare(a, case_mode, &f) == AM_OK && f->h.dfa_n_states != 0)))
            return run_segmented(a, case_mode, hay, n_hay, total, out);
    }
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

// countMatches (benchmark/haskell/app/Main.hs:67-76) over the end positions in (lo, hi] of ONE haystack.  No record is written: the count-mode scan
// runs over two slices of the window in one call -- text[start, hi') and its prefix text[start, lo') (lo', hi' = lo, hi moved back to a code point
// boundary: no match ends inside a code point) -- and the answer is their difference.  A scan of a prefix reports exactly the matches of the longer
// scan that end inside the prefix, and whatever the window misses before `start` (matches that begin before it) it misses in both; every match
// ending behind lo' lies inside the window (range_window's overlap).  The prefix is at most the overlap: a few dozen bytes.
extern "C" int am_count_range(const am_automaton* a, int case_mode, const am_slice* hay, uint64_t lo, uint64_t hi, uint64_t* count_out)
{
    if (!count_out) return fail(AM_ERR_INVALID, "count_out is null");
    *count_out = 0;
    uint64_t start = 0, scan_hi = 0;
    AM_TRY(range_window(a, case_mode, hay, lo, hi, &start, &scan_hi));
    if (hi == lo) return AM_OK;
    const uint8_t* t = hay->ptr + hay->off;
    uint64_t lo_b = lo, hi_b = hi;
    while (lo_b > start && lo_b < hay->len && (t[lo_b] & 0xC0u) == 0x80u) lo_b--;
    while (hi_b > start && hi_b < hay->len && (t[hi_b] & 0xC0u) == 0x80u) hi_b--;
    if (hi_b <= lo_b) return AM_OK;
    const am_slice two[2] = {{hay->ptr, hay->off + start, hi_b - start}, {hay->ptr, hay->off + start, lo_b - start}};
    uint64_t counts[2] = {0, 0};
    AM_TRY(am_count(a, case_mode, two, 2, counts));
    *count_out = counts[0] - counts[1];
    return AM_OK;
}

// ------------------------------------------------------------------ results

// one host block of a freed large result is kept for the next one (up to 8 GiB; larger ones go back to the allocator): mapping, first touch and release of a
// 5-GB block cost as much as copying the records into it (0.18 s + 0.18 s, measured)
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
        if (c > ((size_t)8 << 30)) { std::free(q); return; }
        void* old = nullptr;
        { std::lock_guard<std::mutex> lk(mu); old = p; p = q; cap = c; }
        std::free(old);
    }
    void trim()
    {
        void* old = nullptr;
        { std::lock_guard<std::mutex> lk(mu); old = p; p = nullptr; cap = 0; }
        std::free(old);
    }
    ~HostCache() { std::free(p); }
};
static HostCache g_host_cache;

// Large results go into PAGE-LOCKED host blocks of the library's own (the caller only ever sees the pointer am_matches_data returns): the
// records are DMA'd straight into them, no staging copy -- a match-dense result is several times the size of the text that produced it
// (natural language: 2.5 x) and used to crawl through two 8-MiB staging halves and a single-threaded memcpy (2 GiB of text: 926 ms,
// round 3).  Page-locking is slow (~1 GiB/s) and page-locked memory is a limited resource, so one freed block is kept for the next result
// (up to kPinnedKeep = 1 GiB) and larger ones are given back at once; a kept block serves a result of at least half its size (a 9-MiB result
// does not sit in a 1-GiB block); am_release_host_memory() gives the kept block back; if the runtime refuses a block the pageable path below
// still works.
struct PinnedCache {
    std::mutex mu; void* p = nullptr; size_t cap = 0;
    void* take(size_t need, size_t* cap_out)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (p && cap >= need && cap <= 2 * need + ((size_t)16 << 20)) { void* r = p; *cap_out = cap; p = nullptr; cap = 0; return r; }
        return nullptr;
    }
    void trim()
    {
        void* old = nullptr;
        { std::lock_guard<std::mutex> lk(mu); old = p; p = nullptr; cap = 0; }
        if (old) (void)hipHostFree(old);
    }
    void give(void* q, size_t c, size_t keep_limit)
    {
        void* old = q;
        if (c <= keep_limit) { std::lock_guard<std::mutex> lk(mu); if (c > cap) { old = p; p = q; cap = c; } }
        if (old) (void)hipHostFree(old);
    }
};
static PinnedCache& pinned_cache() { static PinnedCache* c = new PinnedCache(); return *c; }      // never destroyed: no HIP call in a static destructor
constexpr size_t kPinnedKeep = (size_t)1 << 30;

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

// device -> a pageable host block of several GiB (a match-dense result: natural text yields 2.5 x its own size in records).  Page-locking such a block costs more than
// moving it (hipHostMalloc of 5.5 GB: 1.1 s, hipHostFree 0.7 s, measured in round 6 -- the whole call ran at 1.7 GiB/s of text), so the records cross PCIe into three
// page-locked 32-MiB pieces that take turns (kept per device), and a few threads copy each piece out while the next two are on their way; the block itself is
// asked for in huge pages (its first touch and its release are then cheap).
static int fetch_parallel(void* dst, const void* d_src, size_t bytes, int dev)
{
    struct DownloadStage { std::mutex mu; uint8_t* buf[3] = {nullptr, nullptr, nullptr}; hipEvent_t ev[3] = {nullptr, nullptr, nullptr}; };
    static DownloadStage per_device[kMaxDev];
    DownloadStage& ds = per_device[dev];
    constexpr size_t kPiece = 32u << 20;                         // (8 ... 128 MiB: the same rate)
    std::lock_guard<std::mutex> stage_lk(ds.mu);                 // big downloads from one device take turns (they share its PCIe link anyway)
    if (!ds.buf[0]) {
        bool good = true;
        for (int k = 0; k < 3 && good; k++) good = hipHostMalloc((void**)&ds.buf[k], kPiece, hipHostMallocPortable) == hipSuccess && hipEventCreateWithFlags(&ds.ev[k], hipEventDisableTiming) == hipSuccess;
        if (!good) {
            (void)hipGetLastError();
            for (int k = 0; k < 3; k++) { if (ds.buf[k]) (void)hipHostFree(ds.buf[k]); if (ds.ev[k]) (void)hipEventDestroy(ds.ev[k]); ds.buf[k] = nullptr; ds.ev[k] = nullptr; }
            return fail(AM_ERR_OOM, "pinned pieces for a large download could not be created");
        }
    }
    hipStream_t st; AM_TRY(get_stream(dev, &st));                // the calling thread's stream: behind the kernels that wrote the records
    const size_t n_pieces = (bytes + kPiece - 1) / kPiece;
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned n_threads = std::max(1u, std::min(16u, hw ? hw / 2u : 1u));      // (8 threads copy 29 GB/s out of the pieces, the wire brings 50)
    std::mutex mu; std::condition_variable cv;
    size_t ready = 0;                                            // pieces whose DMA has finished
    bool failed = false;
    std::vector<unsigned> copied(n_pieces, 0);                   // threads that have copied their share of piece i out
    auto worker = [&](unsigned t) {
        for (size_t i = 0; i < n_pieces; i++) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return ready > i || failed; }); if (failed) return; }
            const size_t lo = i * kPiece, len = std::min(kPiece, bytes - lo);
            const size_t step = ((len + n_threads - 1) / n_threads + 4095) & ~(size_t)4095, a = std::min(len, t * step), z = std::min(len, a + step);
            if (a < z) std::memcpy((uint8_t*)dst + lo + a, ds.buf[i % 3] + a, z - a);
            { std::lock_guard<std::mutex> lk(mu); if (++copied[i] == n_threads) cv.notify_all(); }
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < n_threads; t++) pool.emplace_back(worker, t);
    hipError_t e = hipSuccess;
    auto issue = [&](size_t i) {
        const size_t lo = i * kPiece, len = std::min(kPiece, bytes - lo);
        e = hipMemcpyAsync(ds.buf[i % 3], (const uint8_t*)d_src + lo, len, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipEventRecord(ds.ev[i % 3], st);
    };
    for (size_t i = 0; i < std::min<size_t>(3, n_pieces) && e == hipSuccess; i++) issue(i);
    for (size_t i = 0; i < n_pieces && e == hipSuccess; i++) {
        e = hipEventSynchronize(ds.ev[i % 3]);
        if (e != hipSuccess) break;
        { std::lock_guard<std::mutex> lk(mu); ready = i + 1; }
        cv.notify_all();
        if (i + 3 < n_pieces) {                                  // its buffer takes piece i + 3 once every thread has copied its share of piece i out
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return copied[i] == n_threads; }); }
            issue(i + 3);
        }
    }
    if (e != hipSuccess) { { std::lock_guard<std::mutex> lk(mu); failed = true; } cv.notify_all(); }
    for (auto& th : pool) th.join();
    if (e != hipSuccess) { (void)hipStreamSynchronize(st); return fail(AM_ERR_HIP, std::string("copying the match records to the host: ") + hipGetErrorString(e)); }
    return AM_OK;
}
// a host block for a large result: 2-MiB aligned and advised into huge pages (first touch and release of several GiB of 4-KiB pages cost tenths of a second)
static void* big_block_alloc(size_t bytes)
{
    constexpr size_t kHuge = (size_t)2 << 20;
    if (bytes < 8 * kHuge) return std::malloc(bytes);
    void* q = std::aligned_alloc(kHuge, (bytes + kHuge - 1) & ~(kHuge - 1));
#ifdef MADV_HUGEPAGE
    if (q) (void)madvise(q, (bytes + kHuge - 1) & ~(kHuge - 1), MADV_HUGEPAGE);
#endif
    return q;
}

// am_run on host slices of 1 GiB and more.  A call used to be upload -> scan -> (am_matches_data) download, one after the other, and on match-dense text the records
// are several times the text (natural language against a dictionary: 2.5 x): the wire stood still in one direction while the other worked.  Here the haystacks go up in
// segments of >= 256 MiB (whole haystacks); a segment is scanned as a batch of its own, its records get their haystack numbers rebased on the device, and a second thread
// brings them into their place in the host block (fetch_parallel, on its own stream) while the calling thread gathers, uploads and scans the next segment -- PCIe is full
// duplex.  The block is sized from the first segment's density (and grown if the text turns denser).  A first segment with few records (< 1/8 of its text in bytes: the
// upload is the bound, nothing to overlap) sends all the rest up as ONE segment.  The result lives on the host only: am_matches_data is immediate, am_matches_device_data
// is NULL, am_matches_haystack_range / am_matches_copy search and copy the host block.
static int run_segmented(const am_automaton* a, int case_mode, const am_slice* hay, size_t n_hay, uint64_t total, am_matches** out)
{
    *out = nullptr;
    const int dev = a->dev;
    AM_TRY(ensure_runtime());
    ON_DEVICE(dev);
    am_batch* b = oneshot_get(dev);
    struct Job { am_matches* m; uint64_t at; };
    static const bool trace = std::getenv("AM_RUN_TRACE") != nullptr;      // (measurements: when each segment's steps begin and end, ms since the call began, on stderr)
    const auto t_begin = std::chrono::steady_clock::now();
    auto now_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    std::mutex mu; std::condition_variable cv;
    std::deque<Job> jobs; bool closing = false; int dl_rc = AM_OK; std::string dl_err;
    am_match* block = nullptr; size_t block_cap = 0;              // bytes
    std::thread downloader([&] {
        OnDevice od(dev);
        for (;;) {
            Job j;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return closing || !jobs.empty(); }); if (jobs.empty()) return; j = jobs.front(); }
            int rc = od.rc;
            const double t0 = now_ms();
            if (rc == AM_OK && j.m->n) rc = fetch_parallel(block + j.at, j.m->d_records + j.m->first, (size_t)j.m->n * sizeof(Record), dev);      // (block is not moved while a job is queued)
            const std::string msg = rc != AM_OK ? std::string(am_last_error()) : std::string();
            const double t1 = now_ms();
            const uint64_t n_j = j.m->n;
            am_matches_free(j.m);
            if (trace) std::fprintf(stderr, "[am_run] download of %llu records: %.1f .. %.1f ms, freed at %.1f\n", (unsigned long long)n_j, t0, t1, now_ms());
            { std::lock_guard<std::mutex> lk(mu); jobs.pop_front(); if (rc != AM_OK && dl_rc == AM_OK) { dl_rc = rc; dl_err = msg; } }
            cv.notify_all();
        }
    });
    auto drain = [&] { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return jobs.empty(); }); };
    int rc = AM_OK;
    uint64_t n_total = 0;
    bool rest_at_once = false;
    const uint64_t segment = cfg::get(cfg::kRunSegments) > 0 ? (uint64_t)cfg::get(cfg::kRunSegments) << 10 : kRunSegment;
    hipStream_t st = nullptr;
    rc = get_stream(dev, &st);
    for (size_t i = 0; i < n_hay && rc == AM_OK;) {
        size_t j = i; uint64_t bytes = 0;
        // (the first segment a quarter of the others: its upload and scan are the only ones no download runs beside)
        while (j < n_hay && (rest_at_once || bytes < (i == 0 ? segment / 4 : segment))) bytes += hay[j++].len;
        const double t_up = now_ms();
        rc = upload_slices(hay + i, j - i, b, true);
        const double t_scan = now_ms();
        am_matches* m = nullptr;
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return jobs.size() < 2; }); }      // at most three record arrays alive: one on its way, one waiting, this one
        if (rc == AM_OK) rc = run_batch_impl(a, case_mode, b, &m, false);
        if (rc != AM_OK) break;
        if (trace) std::fprintf(stderr, "[am_run] segment of %zu haystacks, %llu bytes: upload %.1f .. %.1f, scan .. %.1f ms\n", j - i, (unsigned long long)bytes, t_up, t_scan, now_ms());
        hipError_t e = m->n ? launch_hay_rebase(m->d_records + m->first, m->n, (uint32_t)i, st) : hipSuccess;
        if (e == hipSuccess) e = hipStreamSynchronize(st);          // the records are final before another thread's stream reads them
        if (e != hipSuccess) { am_matches_free(m); rc = fail(AM_ERR_HIP, std::string("am_run (segments): ") + hipGetErrorString(e)); break; }
        const size_t need = (size_t)(n_total + m->n) * sizeof(Record);
        if (need > block_cap) {
            // the first segment's density, carried over the whole batch, + a fifth; later: half as much again
            const double scale = i == 0 && bytes ? (double)total / (double)bytes * 1.2 : 1.5;
            const size_t want = std::max<size_t>((size_t)((double)need * scale) + ((size_t)1 << 20), need);
            drain();                                                // nothing copies into the old block while it moves
            size_t got_cap = 0;
            am_match* nb = (am_match*)g_host_cache.take(want, &got_cap);
            if (!nb) { got_cap = want; nb = (am_match*)big_block_alloc(want); }
            if (!nb) { am_matches_free(m); rc = fail(AM_ERR_OOM, "out of host memory for the match records"); break; }
            if (block) { std::memcpy(nb, block, (size_t)n_total * sizeof(Record)); g_host_cache.give(block, block_cap); }
            block = nb; block_cap = got_cap;
        }
        if (i == 0 && (uint64_t)m->n * sizeof(Record) * 8u < bytes) rest_at_once = true;
        { std::lock_guard<std::mutex> lk(mu); jobs.push_back(Job{m, n_total}); }
        cv.notify_all();
        n_total += m->n;
        i = j;
    }
    { std::lock_guard<std::mutex> lk(mu); closing = true; }
    cv.notify_all();
    downloader.join();
    oneshot_trim(dev);
    if (rc == AM_OK && dl_rc != AM_OK) rc = fail(dl_rc, dl_err);
    if (rc != AM_OK) { if (block) g_host_cache.give(block, block_cap); return rc; }
    am_matches* res = new am_matches();
    res->dev = dev; res->n = n_total; res->fetched = true;
    if (n_total) { res->big = block; res->big_cap = block_cap; res->big_pinned = false; }
    else if (block) g_host_cache.give(block, block_cap);
    *out = res;
    return AM_OK;
}

extern "C" int am_release_device_memory(void)
{
    if (ensure_runtime() != AM_OK) return AM_OK;                 // (no device: nothing is held)
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess) { (void)hipGetLastError(); n_dev = 0; }
    for (int d = 0; d < n_dev && d < kMaxDev; d++) {
        OnDevice od(d);
        if (od.rc != AM_OK) continue;
        g_record_cache[d].trim();
        am_batch*& b = tl_state.oneshot[d];
        if (b) { am_batch_destroy(b); b = nullptr; }
    }
    return AM_OK;
}

extern "C" int am_release_host_memory(void)
{
    pinned_cache().trim();
    g_host_cache.trim();
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
                if (!m->big && bytes <= kPinnedKeep) {              // (a block that would not be kept is not page-locked either: fetch_parallel below)
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
            if (!m->big) { m->big_cap = bytes + bytes / 16; m->big = (am_match*)big_block_alloc(m->big_cap); }
            if (!m->big) { fail(AM_ERR_OOM, "out of host memory for the match records"); return nullptr; }
            if (bytes <= kFetchPiece) {                       // a few MiB: the runtime's own staged copy is faster than two pieces of ours (1.8 MB: 290 against 410 us per am_run)
                hipError_t e = hipMemcpy(m->big, m->d_records + m->first, bytes, hipMemcpyDeviceToHost);
                if (e != hipSuccess) { fail(AM_ERR_HIP, std::string("hipMemcpy(records): ") + hipGetErrorString(e)); std::free(m->big); m->big = nullptr; return nullptr; }
            } else if ((bytes > kPinnedKeep ? fetch_parallel(m->big, m->d_records + m->first, bytes, m->dev) : fetch_through_pinned(m->big, m->d_records + m->first, bytes, m->dev)) != AM_OK) {
                std::free(m->big); m->big = nullptr; return nullptr;
            }
        }
        m->fetched = true;
    }
    return m->big ? m->big : m->host.data();
}

extern "C" const void* am_matches_device_data(const am_matches* m) { return m ? (m->d_records ? m->d_records + m->first : nullptr) : nullptr; }

// A slice of a result without bringing all of it to the host: the records are sorted by (haystack, end_pos), so one haystack's records are a contiguous run.
// The run is found by a binary search over the records in HBM (4-byte probes of the haystack field: ~2 log2(n) small copies), the copy is one DMA.
extern "C" int am_matches_haystack_range(const am_matches* m, uint32_t haystack, uint64_t* first_out, uint64_t* count_out)
{
    if (!m || !first_out || !count_out) return fail(AM_ERR_INVALID, "am_matches_haystack_range: null argument");
    *first_out = 0; *count_out = 0;
    if (m->n == 0) return AM_OK;
    if (!m->d_records) {                                          // a result assembled on the host (am_run on a large host batch)
        if (!m->big) return fail(AM_ERR_INVALID, "am_matches_haystack_range: the result has no records");
        const am_match* lo = std::lower_bound(m->big, m->big + m->n, haystack, [](const am_match& r, uint32_t h) { return r.haystack < h; });
        const am_match* hi = std::upper_bound(lo, (const am_match*)(m->big + m->n), haystack, [](uint32_t h, const am_match& r) { return h < r.haystack; });
        *first_out = (uint64_t)(lo - m->big); *count_out = (uint64_t)(hi - lo);
        return AM_OK;
    }
    OnDevice od(m->dev);
    const Record* r = m->d_records + m->first;
    auto lower = [&](uint64_t key, uint64_t* out) -> int {        // first index whose haystack >= key
        uint64_t lo = 0, hi = m->n;
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo) / 2;
            uint32_t h = 0;
            HIP_TRY(hipMemcpy(&h, &r[mid].haystack, sizeof(h), hipMemcpyDeviceToHost));
            if (h < key) lo = mid + 1; else hi = mid;
        }
        *out = lo;
        return AM_OK;
    };
    uint64_t a = 0, b = 0;
    AM_TRY(lower(haystack, &a));
    AM_TRY(lower((uint64_t)haystack + 1u, &b));
    *first_out = a; *count_out = b - a;
    return AM_OK;
}

extern "C" int am_matches_copy(const am_matches* m, uint64_t first, uint64_t count, am_match* out)
{
    if (!m || (count && !out)) return fail(AM_ERR_INVALID, "am_matches_copy: null argument");
    if (first > m->n || count > m->n - first) return fail(AM_ERR_INVALID, "am_matches_copy: range outside the result");
    if (count == 0) return AM_OK;
    if (!m->d_records) {                                          // a result assembled on the host
        if (!m->big) return fail(AM_ERR_INVALID, "am_matches_copy: the result has no records");
        std::memcpy(out, m->big + first, (size_t)count * sizeof(Record));
        return AM_OK;
    }
    OnDevice od(m->dev);
    HIP_TRY(hipMemcpy(out, m->d_records + m->first + first, (size_t)count * sizeof(Record), hipMemcpyDeviceToHost));
    return AM_OK;
}

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

// ---- include/am_debug.h: tests and measurements only
extern "C" int am_debug_set_general_kernel(void* launcher, uint32_t image_version)
{
    if (launcher && image_version != kImageVersion) return fail(AM_ERR_INVALID, "libam_check.so was built against another image version");
    g_ac_launcher.store(reinterpret_cast<AcLauncher>(launcher), std::memory_order_release);
    return AM_OK;
}

// cycle sums per k_sf phase for launches made under AM_SF_ABLATE=9
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

// index assertions of a -DAM_BOUNDS_CHECK build (am_bounds.h): every kernel translation unit that carries them registers a reader of its device-side words at load time
namespace {
struct BoundsUnit { const char* file; hipError_t (*read)(uint32_t*); };
std::vector<BoundsUnit>& bounds_units() { static std::vector<BoundsUnit> v; return v; }
}  // namespace
namespace am { namespace dev { void bounds_register(const char* file, hipError_t (*read)(uint32_t* out2)) { bounds_units().push_back(BoundsUnit{file, read}); } } }

extern "C" int am_debug_bounds_report(uint64_t* failed_out, uint32_t* first_line_out, uint32_t* checked_units_out)
{
    uint64_t failed = 0; uint32_t line = 0;
    if (!bounds_units().empty()) HIP_TRY(hipDeviceSynchronize());
    for (const BoundsUnit& u : bounds_units()) {
        uint32_t w[2] = {0, 0};
        HIP_TRY(u.read(w));
        if (w[0] && !failed) { line = w[1]; g_err = std::string("first failed index assertion: ") + u.file + ":" + std::to_string(w[1]); }
        failed += w[0];
    }
    if (failed_out) *failed_out = failed;
    if (first_line_out) *first_line_out = line;
    if (checked_units_out) *checked_units_out = (uint32_t)bounds_units().size();
    return AM_OK;
}

extern "C" uint64_t am_debug_pinned_bytes(void) { return (uint64_t)g_pinned_staging_bytes.load(std::memory_order_relaxed); }

extern "C" int am_debug_resident_waves(float* one_ms_out, float* two_ms_out)
{
    if (!one_ms_out || !two_ms_out) return fail(AM_ERR_INVALID, "null argument");
    AM_TRY(ensure_runtime());
    int dev = 0; AM_TRY(current_device(&dev));
    hipDeviceProp_t prop; HIP_TRY(hipGetDeviceProperties(&prop, dev));
    const uint32_t n_cu = (uint32_t)prop.multiProcessorCount;
    hipStream_t st; AM_TRY(get_stream(dev, &st));
    uint32_t* d = nullptr; HIP_TRY(hipMalloc((void**)&d, 64));
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t e = hipEventCreate(&a); if (e == hipSuccess) e = hipEventCreate(&b);
    float ms[2] = {0, 0};
    const uint64_t cycles = 1000000;                             // ~0.4 ms
    for (int k = 0; k < 2 && e == hipSuccess; k++) {
        // 16 wavefronts per CU: one workgroup of 1024 threads each; 32: eight workgroups of 256 threads each
        const uint32_t wgs = k == 0 ? n_cu : 8u * n_cu, threads = k == 0 ? 1024u : 256u;
        e = launch_spin(wgs, threads, 1000, d, st);              // (warm-up)
        if (e == hipSuccess) e = hipEventRecord(a, st);
        if (e == hipSuccess) e = launch_spin(wgs, threads, cycles, d, st);
        if (e == hipSuccess) e = hipEventRecord(b, st);
        if (e == hipSuccess) e = hipEventSynchronize(b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms[k], a, b);
    }
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(AM_ERR_HIP, std::string("am_debug_resident_waves: ") + hipGetErrorString(e));
    *one_ms_out = ms[0]; *two_ms_out = ms[1];
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

