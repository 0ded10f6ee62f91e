// am_abi.cpp -- the C ABI of include/am.h: handles, device memory, launch orchestration.
// There is deliberately no CPU execution path here: every run entry point needs a HIP device.
#include "../../include/am.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

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
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(AM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define AM_TRY(expr) do { int rc_ = (expr); if (rc_ != AM_OK) return rc_; } while (0)

namespace {

struct Runtime {
    std::mutex mu;
    bool probed = false;
    bool have_device = false;
    int n_cu = 0;
    size_t hbm = 0;
    std::string name, why;
    hipStream_t own_stream = nullptr;
    hipStream_t user_stream = nullptr;
    bool use_user = false;
    // profiling
    bool prof_on = false;
    struct Pending { std::string k; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::map<std::string, std::pair<double, uint64_t>> prof;
};
Runtime g_rt;

int ensure_device()
{
    std::lock_guard<std::mutex> lk(g_rt.mu);
    if (!g_rt.probed) {
        g_rt.probed = true;
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) {
            g_rt.why = std::string("no HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") + "); libam has no CPU path";
        } else {
            int dev = 0;
            hipDeviceProp_t p;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) {
                g_rt.have_device = true;
                g_rt.n_cu = p.multiProcessorCount;
                g_rt.hbm = p.totalGlobalMem;
                g_rt.name = p.gcnArchName;
            } else {
                g_rt.why = "hipGetDeviceProperties failed";
            }
        }
    }
    if (!g_rt.have_device) return fail(AM_ERR_NO_DEVICE, g_rt.why);
    return AM_OK;
}

int get_stream(hipStream_t* st)
{
    std::lock_guard<std::mutex> lk(g_rt.mu);
    if (g_rt.use_user) { *st = g_rt.user_stream; return AM_OK; }
    if (!g_rt.own_stream) HIP_TRY(hipStreamCreateWithFlags(&g_rt.own_stream, hipStreamNonBlocking));
    *st = g_rt.own_stream;
    return AM_OK;
}

// RAII HIP-event bracket around one kernel launch (only when profiling is enabled)
struct Prof {
    bool on; hipStream_t st; Runtime::Pending p;
    Prof(const char* k, hipStream_t s) : on(g_rt.prof_on), st(s)
    {
        if (!on) return;
        p.k = k;
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
    std::vector<uint64_t> transitions, root_ascii;
    std::vector<uint32_t> offsets, values_len;
    bool has_ref = false;        // false for handles attached to a received image
    std::vector<uint8_t> cs_image;   // CaseSensitive image flattened (= validated) at creation, uploaded on first use
    int kernel_pref = 0;
    std::mutex mu;
    Flavor fl[2];
};

struct am_batch {
    void* d_text = nullptr; uint64_t* d_offsets = nullptr;
    bool owns = false;
    uint64_t total = 0; uint32_t n_hay = 0;
    std::mutex mu;              // guards the workspaces below (calls on one batch serialise)
    DevBuf hidx, unit_counts, unit_offsets, scan_tmp, small, hay_counts, flags, unit_first, pool, block_next;
};

struct am_matches {
    Record* d_records = nullptr; uint64_t n = 0;
    std::vector<am_match> host; bool fetched = false;
};

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
        AM_TRY(ensure_device());
        std::vector<uint8_t> img; std::string err;
        if (case_mode == AM_CASE_SENSITIVE && !a->cs_image.empty()) img.swap(a->cs_image);
        else {
            RefArrays ref{a->transitions.data(), a->transitions.size(), a->offsets.data(), a->offsets.size() - 1, a->root_ascii.data(), a->values_len.data()};
            if (flatten(ref, case_mode, img, err) != 0) return fail(AM_ERR_INVALID, err);
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

extern "C" int am_automaton_create(const uint64_t* transitions, size_t n_transitions, const uint32_t* offsets, size_t n_states,
                                   const uint64_t* root_ascii, const uint32_t* values_len, am_automaton** out)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!transitions || !offsets || !root_ascii || !values_len || n_states == 0) return fail(AM_ERR_INVALID, "null or empty automaton arrays");
    // validate on the host right away (flatten checks every index); the image is uploaded on first use
    std::vector<uint8_t> img;
    {
        std::string err;
        RefArrays ref{transitions, n_transitions, offsets, n_states, root_ascii, values_len};
        if (flatten(ref, AM_CASE_SENSITIVE, img, err) != 0) return fail(AM_ERR_INVALID, err);
    }
    am_automaton* a = new am_automaton();
    a->cs_image.swap(img);
    a->transitions.assign(transitions, transitions + n_transitions);
    a->offsets.assign(offsets, offsets + n_states + 1);
    a->root_ascii.assign(root_ascii, root_ascii + 128);
    a->values_len.assign(values_len, values_len + n_states);
    a->has_ref = true;
    *out = a;
    return AM_OK;
}

extern "C" void am_automaton_destroy(am_automaton* a)
{
    if (!a) return;
    for (Flavor& f : a->fl) if (f.d_image) (void)hipFree(f.d_image);
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
    HIP_TRY(hipMemcpy(d_dst, f->d_image, f->bytes, hipMemcpyDeviceToDevice));
    return AM_OK;
}

extern "C" int am_automaton_from_image(const void* d_image, size_t nbytes, am_automaton** out)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    AM_TRY(ensure_device());
    if (!d_image || nbytes < sizeof(ImageHeader)) return fail(AM_ERR_INVALID, "image too small");
    ImageHeader h;
    HIP_TRY(hipMemcpy(&h, d_image, sizeof(h), hipMemcpyDeviceToHost));
    if (h.magic != kImageMagic || h.version != kImageVersion || h.total_bytes > nbytes || h.case_mode > 1) return fail(AM_ERR_INVALID, "not an automaton image");
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, h.total_bytes));
    hipError_t e = hipMemcpy(d, d_image, h.total_bytes, hipMemcpyDeviceToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return fail(AM_ERR_HIP, hipGetErrorString(e)); }
    am_automaton* a = new am_automaton();
    Flavor& f = a->fl[h.case_mode];
    f.h = h; f.d_image = d; f.bytes = h.total_bytes; f.ready = true;
    *out = a;
    return AM_OK;
}

// ------------------------------------------------------------------ batches

static int finish_batch(am_batch* b)
{
    if (b->total > 0) AM_TRY(b->hidx.ensure(((b->total >> kHidxShift) + 2) * sizeof(uint32_t)));
    return AM_OK;
}

extern "C" int am_batch_upload(const am_slice* hay, size_t n_hay, am_batch** out)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    if (n_hay && !hay) return fail(AM_ERR_INVALID, "hay is null");
    if (n_hay >= 0xFFFFFFFFull) return fail(AM_ERR_INVALID, "too many haystacks");
    AM_TRY(ensure_device());
    std::vector<uint64_t> offs(n_hay + 1, 0);
    for (size_t i = 0; i < n_hay; i++) {
        if (hay[i].len && !hay[i].ptr) return fail(AM_ERR_INVALID, "slice with null ptr");
        offs[i + 1] = offs[i] + hay[i].len;
    }
    const uint64_t total = offs[n_hay];
    const size_t padded = (size_t)((total + 15) & ~15ull) + 16;
    // gather the slices into a pinned staging buffer (kept for the next call) so that the H2D copy is one DMA
    static std::mutex stage_mu;
    static uint8_t* stage = nullptr;
    static size_t stage_cap = 0;
    std::lock_guard<std::mutex> stage_lk(stage_mu);
    if (padded > stage_cap) {
        if (stage) { (void)hipHostFree(stage); stage = nullptr; stage_cap = 0; }
        const size_t want = padded + padded / 4;
        if (hipHostMalloc((void**)&stage, want, hipHostMallocDefault) != hipSuccess) { stage = nullptr; return fail(AM_ERR_OOM, "hipHostMalloc(staging) failed"); }
        stage_cap = want;
    }
    for (size_t i = 0; i < n_hay; i++) if (hay[i].len) std::memcpy(stage + offs[i], hay[i].ptr + hay[i].off, hay[i].len);
    std::memset(stage + total, 0, padded - (size_t)total);
    am_batch* b = new am_batch();
    b->owns = true; b->total = total; b->n_hay = (uint32_t)n_hay;
    hipError_t e = hipMalloc(&b->d_text, padded);
    if (e == hipSuccess) e = hipMalloc((void**)&b->d_offsets, offs.size() * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemcpy(b->d_text, stage, padded, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(b->d_offsets, offs.data(), offs.size() * sizeof(uint64_t), hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? finish_batch(b) : fail(e == hipErrorOutOfMemory ? AM_ERR_OOM : AM_ERR_HIP, std::string("batch upload: ") + hipGetErrorString(e));
    if (rc != AM_OK) { am_batch_destroy(b); return rc; }
    *out = b;
    return AM_OK;
}

extern "C" int am_batch_from_device(const void* d_bytes, const void* d_offsets, size_t n_hay, uint64_t total_bytes, am_batch** out)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    AM_TRY(ensure_device());
    if (!d_offsets || (total_bytes && !d_bytes)) return fail(AM_ERR_INVALID, "null device pointers");
    if (((uintptr_t)d_bytes & 15) != 0) return fail(AM_ERR_INVALID, "d_bytes must be 16-byte aligned");
    if (n_hay >= 0xFFFFFFFFull) return fail(AM_ERR_INVALID, "too many haystacks");
    uint64_t first = 1, last = 0;
    HIP_TRY(hipMemcpy(&first, d_offsets, 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&last, (const uint64_t*)d_offsets + n_hay, 8, hipMemcpyDeviceToHost));
    if (first != 0 || last != total_bytes) return fail(AM_ERR_INVALID, "d_offsets[0] must be 0 and d_offsets[n_hay] must equal total_bytes");
    am_batch* b = new am_batch();
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
    if (b->owns) { if (b->d_text) (void)hipFree(b->d_text); if (b->d_offsets) (void)hipFree(b->d_offsets); }
    for (DevBuf* d : {&b->hidx, &b->unit_counts, &b->unit_offsets, &b->scan_tmp, &b->small, &b->hay_counts, &b->flags, &b->unit_first, &b->pool, &b->block_next}) d->release();
    delete b;
}

extern "C" uint64_t am_batch_total_bytes(const am_batch* b) { return b ? b->total : 0; }

// ------------------------------------------------------------------ scanning

namespace {

struct Plan {
    const Flavor* f; bool ic; bool use_sf; bool nothing; uint64_t n_units; uint32_t unit_chunks;
    AcView ac; SfView sf; BatchView bv;
};

int make_plan(const am_automaton* a, int case_mode, am_batch* b, Plan& p)
{
    if (!b) return fail(AM_ERR_INVALID, "null batch");
    AM_TRY(prepare(a, case_mode, &p.f));
    p.ic = case_mode == AM_IGNORE_CASE;
    if (a->kernel_pref == 2 && !p.f->h.sf_enabled) return fail(AM_ERR_UNSUPPORTED, "suffix-filter kernel cannot run automata that contain the empty needle");
    p.use_sf = p.f->h.sf_enabled && a->kernel_pref != 1;
    p.ac = make_ac_view(p.f->d_image, p.f->h);
    p.sf = make_sf_view(p.f->d_image, p.f->h);
    p.bv = BatchView{(const uint8_t*)b->d_text, b->d_offsets, (const uint32_t*)b->hidx.p, b->total, b->n_hay, 0};
    // no goto edge at all (no needles, or only empty needles): the reference never reports anything
    const bool no_edges = p.f->h.n_transitions == p.f->h.n_states;
    p.nothing = b->total == 0 || no_edges || (p.use_sf && p.f->h.sf_tiers == 0);
    p.unit_chunks = p.use_sf ? sf_unit_chunks(p.bv, g_rt.n_cu) : 0;
    p.n_units = p.nothing ? 0 : (p.use_sf ? (sf_chunks(p.bv) + p.unit_chunks - 1) / p.unit_chunks : ac_units(p.ac, p.bv));
    if (p.n_units >= 0x7FFFFFF0ull) return fail(AM_ERR_UNSUPPORTED, "batch too large for one launch; split it");
    return AM_OK;
}

int launch_scan_kernel(const Plan& p, int mode, const ScanOut& o, hipStream_t st)
{
    if (p.use_sf) { Prof pr("sf", st); HIP_TRY(launch_sf(p.ic, mode, p.sf, p.bv, o, g_rt.n_cu, st)); }
    else { Prof pr("ac", st); HIP_TRY(launch_ac(p.ic, mode, p.ac, p.bv, o, st)); }
    return AM_OK;
}

int build_hidx(const Plan& p, am_batch* b, hipStream_t st)
{
    Prof pr("hidx", st);
    HIP_TRY(launch_hidx(p.bv, (uint32_t*)b->hidx.p, (b->total >> kHidxShift) + 2, st));
    return AM_OK;
}

}  // namespace

extern "C" int am_count_batch(const am_automaton* a, int case_mode, const am_batch* cb, uint64_t* counts_out, uint64_t* total_out)
{
    am_batch* b = const_cast<am_batch*>(cb);
    Plan p; AM_TRY(make_plan(a, case_mode, b, p));
    if (total_out) *total_out = 0;
    if (counts_out && b->n_hay) std::memset(counts_out, 0, (size_t)b->n_hay * sizeof(uint64_t));
    if (p.nothing) return AM_OK;
    std::lock_guard<std::mutex> lk(b->mu);
    hipStream_t st; AM_TRY(get_stream(&st));
    AM_TRY(b->small.ensure(64));
    ScanOut o{};
    o.unit_chunks = p.unit_chunks;
    if (!p.use_sf) { AM_TRY(b->unit_counts.ensure((p.n_units + 1) * sizeof(uint32_t))); o.unit_counts = (uint32_t*)b->unit_counts.p; }
    o.total_values = (uint64_t*)b->small.p;
    HIP_TRY(hipMemsetAsync(b->small.p, 0, 64, st));
    if (counts_out) {
        AM_TRY(b->hay_counts.ensure((size_t)b->n_hay * sizeof(uint64_t)));
        HIP_TRY(hipMemsetAsync(b->hay_counts.p, 0, (size_t)b->n_hay * sizeof(uint64_t), st));
        o.hay_counts = (uint64_t*)b->hay_counts.p;
    }
    AM_TRY(build_hidx(p, b, st));
    AM_TRY(launch_scan_kernel(p, kModeCount, o, st));
    uint64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, b->small.p, 8, hipMemcpyDeviceToHost, st));
    if (counts_out) HIP_TRY(hipMemcpyAsync(counts_out, b->hay_counts.p, (size_t)b->n_hay * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (total_out) *total_out = total;
    return AM_OK;
}

extern "C" int am_contains_any_batch(const am_automaton* a, int case_mode, const am_batch* cb, uint8_t* flags_out)
{
    am_batch* b = const_cast<am_batch*>(cb);
    Plan p; AM_TRY(make_plan(a, case_mode, b, p));
    if (!flags_out && b->n_hay) return fail(AM_ERR_INVALID, "flags_out is null");
    if (b->n_hay) std::memset(flags_out, 0, b->n_hay);
    if (p.nothing) return AM_OK;
    std::lock_guard<std::mutex> lk(b->mu);
    hipStream_t st; AM_TRY(get_stream(&st));
    AM_TRY(b->flags.ensure(b->n_hay));
    HIP_TRY(hipMemsetAsync(b->flags.p, 0, b->n_hay, st));
    ScanOut o{};
    o.unit_chunks = p.unit_chunks;
    o.flags = (uint8_t*)b->flags.p;
    AM_TRY(build_hidx(p, b, st));
    AM_TRY(launch_scan_kernel(p, kModeAny, o, st));
    HIP_TRY(hipMemcpyAsync(flags_out, b->flags.p, b->n_hay, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return AM_OK;
}

extern "C" int am_run_batch(const am_automaton* a, int case_mode, const am_batch* cb, am_matches** out)
{
    if (!out) return fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    am_batch* b = const_cast<am_batch*>(cb);
    Plan p; AM_TRY(make_plan(a, case_mode, b, p));
    am_matches* m = new am_matches();
    if (p.nothing) { *out = m; return AM_OK; }
    std::lock_guard<std::mutex> lk(b->mu);
    hipStream_t st;
    int rc = get_stream(&st);
    auto bail = [&](int code) { am_matches_free(m); return code; };
    if (rc != AM_OK) return bail(rc);
    const uint64_t n = p.n_units + 1;           // trailing zero: offsets[n_units] = total
    if ((rc = b->unit_counts.ensure(n * sizeof(uint32_t))) != AM_OK) return bail(rc);
    if ((rc = b->unit_offsets.ensure(n * sizeof(uint64_t))) != AM_OK) return bail(rc);
    if ((rc = b->small.ensure(64)) != AM_OK) return bail(rc);
    size_t tmp_bytes = 0;
    if (scan_temp_bytes(n, &tmp_bytes) != hipSuccess) return bail(fail(AM_ERR_HIP, "hipcub scan sizing failed"));
    if ((rc = b->scan_tmp.ensure(tmp_bytes + 16)) != AM_OK) return bail(rc);

    auto alloc_records = [&](uint64_t total) -> int {
        hipError_t e = hipMalloc((void**)&m->d_records, total * sizeof(Record));
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? AM_ERR_OOM : AM_ERR_HIP, std::string("hipMalloc(records): ") + hipGetErrorString(e));
        return AM_OK;
    };
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
        m->n = total;
        if (total == 0) return AM_OK;
        AM_TRY(alloc_records(total));
        ScanOut w{};
        w.unit_offsets = (const uint64_t*)b->unit_offsets.p;
        w.records = m->d_records;
        AM_TRY(launch_scan_kernel(p, kModeEmit, w, st));
        HIP_TRY(hipStreamSynchronize(st));
        return AM_OK;
    };
    // suffix-filter kernel: ONE scan pass writes records into pool blocks (chained per unit), then
    // scan(unit_counts) + k_permute put them in order.  The pool size is a guess (1 record per 128
    // haystack bytes + one block per unit); if it overflows the kernel still counts, and the pass is
    // repeated once with the exact number of blocks.
    auto body_sf = [&]() -> int {
        AM_TRY(b->unit_first.ensure(p.n_units * sizeof(uint32_t)));
        uint64_t want_blocks = b->total / (128 * kPoolBlock) + p.n_units + 1024;
        if (b->pool.cap / (kPoolBlock * sizeof(Record)) > want_blocks) want_blocks = b->pool.cap / (kPoolBlock * sizeof(Record));
        if (const char* env = std::getenv("AM_SF_POOL_BLOCKS")) { long v = std::atol(env); if (v > 0) want_blocks = (uint64_t)v; }   // tests: force the overflow/retry path
        for (int attempt = 0; attempt < 3; attempt++) {
            if (want_blocks >= 0xFFFFFFF0ull) return fail(AM_ERR_UNSUPPORTED, "too many match records for one call; split the batch");
            AM_TRY(b->pool.ensure(want_blocks * kPoolBlock * sizeof(Record)));
            AM_TRY(b->block_next.ensure(want_blocks * sizeof(uint32_t)));
            ScanOut o{};
            o.unit_chunks = p.unit_chunks;
            o.unit_counts = (uint32_t*)b->unit_counts.p;
            o.unit_first = (uint32_t*)b->unit_first.p;
            o.pool = (Record*)b->pool.p;
            o.block_next = (uint32_t*)b->block_next.p;
            o.pool_ctrl = (uint32_t*)b->small.p + 4;            // small: [0..1] total_values, [4] block counter, [5] overflow
            o.n_blocks = (uint32_t)want_blocks;
            HIP_TRY(hipMemsetAsync(b->small.p, 0, 64, st));
            HIP_TRY(hipMemsetAsync((uint32_t*)b->unit_counts.p + p.n_units, 0, sizeof(uint32_t), st));
            AM_TRY(build_hidx(p, b, st));
            AM_TRY(launch_scan_kernel(p, kModeEmit, o, st));
            { Prof pr("scan", st); HIP_TRY(launch_scan(b->scan_tmp.p, tmp_bytes, (const uint32_t*)b->unit_counts.p, (uint64_t*)b->unit_offsets.p, n, st)); }
            uint64_t total = 0; uint32_t ctrl[2] = {0, 0};
            HIP_TRY(hipMemcpyAsync(&total, (uint64_t*)b->unit_offsets.p + p.n_units, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(ctrl, o.pool_ctrl, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (ctrl[1]) { want_blocks = (uint64_t)ctrl[0] + 64; continue; }    // pool too small: ctrl[0] = blocks actually needed
            m->n = total;
            if (total == 0) return AM_OK;
            AM_TRY(alloc_records(total));
            { Prof pr("permute", st); HIP_TRY(launch_permute(o, (const uint64_t*)b->unit_offsets.p, m->d_records, p.n_units, st)); }
            HIP_TRY(hipStreamSynchronize(st));
            return AM_OK;
        }
        return fail(AM_ERR_HIP, "record pool overflowed repeatedly (internal error)");
    };
    auto body = [&]() -> int { return p.use_sf ? body_sf() : body_ac(); };
    rc = body();
    if (rc != AM_OK) return bail(rc);
    *out = m;
    return AM_OK;
}

// ------------------------------------------------------------------ one-shot host entry points

extern "C" int am_count(const am_automaton* a, int case_mode, const am_slice* hay, size_t n_hay, uint64_t* counts_out)
{
    if (n_hay && !counts_out) return fail(AM_ERR_INVALID, "counts_out is null");
    am_batch* b = nullptr;
    AM_TRY(am_batch_upload(hay, n_hay, &b));
    int rc = am_count_batch(a, case_mode, b, counts_out, nullptr);
    am_batch_destroy(b);
    return rc;
}

extern "C" int am_contains_any(const am_automaton* a, int case_mode, const am_slice* hay, size_t n_hay, uint8_t* flags_out)
{
    am_batch* b = nullptr;
    AM_TRY(am_batch_upload(hay, n_hay, &b));
    int rc = am_contains_any_batch(a, case_mode, b, flags_out);
    am_batch_destroy(b);
    return rc;
}

extern "C" int am_run(const am_automaton* a, int case_mode, const am_slice* hay, size_t n_hay, am_matches** out)
{
    am_batch* b = nullptr;
    AM_TRY(am_batch_upload(hay, n_hay, &b));
    int rc = am_run_batch(a, case_mode, b, out);
    am_batch_destroy(b);
    return rc;
}

// ------------------------------------------------------------------ results

extern "C" uint64_t am_matches_size(const am_matches* m) { return m ? m->n : 0; }

extern "C" const am_match* am_matches_data(am_matches* m)
{
    if (!m) return nullptr;
    if (!m->fetched) {
        m->host.resize(m->n);
        if (m->n) {
            hipError_t e = hipMemcpy(m->host.data(), m->d_records, m->n * sizeof(Record), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { fail(AM_ERR_HIP, std::string("hipMemcpy(records): ") + hipGetErrorString(e)); return nullptr; }
        }
        m->fetched = true;
    }
    return m->host.data();
}

extern "C" const void* am_matches_device_data(const am_matches* m) { return m ? m->d_records : nullptr; }

extern "C" void am_matches_free(am_matches* m)
{
    if (!m) return;
    if (m->d_records) (void)hipFree(m->d_records);
    delete m;
}

// ------------------------------------------------------------------ UTF-8 helpers

extern "C" uint32_t am_lower_code_point(uint32_t cp) { return cp < 128 ? fold_byte(cp) : simple_lower(cp); }

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
    std::lock_guard<std::mutex> lk(g_rt.mu);
    g_rt.user_stream = (hipStream_t)hip_stream;
    g_rt.use_user = hip_stream != nullptr;
    return AM_OK;
}

extern "C" int am_device_info(int* n_cu, size_t* hbm_bytes, char* name, size_t name_cap)
{
    AM_TRY(ensure_device());
    if (n_cu) *n_cu = g_rt.n_cu;
    if (hbm_bytes) *hbm_bytes = g_rt.hbm;
    if (name && name_cap) { std::strncpy(name, g_rt.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    return AM_OK;
}

// debug only (not declared in am.h): cycle sums per k_sf phase for launches made under AM_SF_ABLATE=9
extern "C" int am_debug_sf_phase_cycles(uint64_t* out5)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(read_sf_phase_cycles(out5));
    return AM_OK;
}

extern "C" int am_profile_enable(int on) { std::lock_guard<std::mutex> lk(g_rt.mu); g_rt.prof_on = on != 0; return AM_OK; }

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
