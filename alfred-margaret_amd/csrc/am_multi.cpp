// am_multi.cpp -- several GPUs behind the C ABI (include/am.h "several GPUs").
//
// The path shards trivially: haystacks are independent and the automaton is read-only.  So the only traffic between
// devices is (1) ONE broadcast of the flattened automaton image over xGMI (ncclBroadcast of the position-independent
// blob) and (2) an all-reduce of match counts (ncclAllReduce, a handful of uint64).  Match lists never cross devices:
// every device's records go to the host and are concatenated in haystack order.
//
// Two ways to span the devices, same entry points afterwards:
//   am_multi_create       one process drives devices 0..n-1 (ncclCommInitAll; one host thread per device while scanning)
//   am_multi_create_rank  one process per GPU (ncclCommInitRank with an id made by am_multi_unique_id on rank 0 and
//                         handed round by the launcher -- bench.py uses torch.distributed only for those 128 bytes)
// This file sits ABOVE the single-device ABI: it only calls what am.h declares.
#include "../../include/am.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <mutex>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>

namespace am { int abi_fail(int code, const std::string& msg); }
using am::abi_fail;

// RCCL is bound at first use (dlopen), not at link time: single-GPU callers never load it, and a process that already
// holds an RCCL (PyTorch ships its own copy) keeps using that one instead of getting a second copy of the library.
// AM_RCCL_LIBRARY=<path> names the library to bind instead (a site's own RCCL build; the test suite's file-based stand-in).
namespace {
struct Rccl {
    void* lib = nullptr;
    decltype(&::ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&::ncclCommInitAll) CommInitAll = nullptr;
    decltype(&::ncclCommInitRank) CommInitRank = nullptr;
    decltype(&::ncclCommDestroy) CommDestroy = nullptr;
    decltype(&::ncclBroadcast) Broadcast = nullptr;
    decltype(&::ncclAllReduce) AllReduce = nullptr;
    decltype(&::ncclGroupStart) GroupStart = nullptr;
    decltype(&::ncclGroupEnd) GroupEnd = nullptr;
    decltype(&::ncclGetErrorString) GetErrorString = nullptr;
    std::string why;
    bool load()
    {
        if (lib) return true;
        if (!why.empty()) return false;
        if (const char* named = std::getenv("AM_RCCL_LIBRARY")) {
            if (*named) {
                lib = dlopen(named, RTLD_NOW | RTLD_LOCAL);
                if (!lib) { why = std::string("AM_RCCL_LIBRARY: ") + dlerror(); return false; }
            }
        }
        if (!lib) for (const char* name : {"librccl.so.1", "librccl.so"}) { lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD); if (lib) break; }      // one that is already in the process
        if (!lib) for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) { why = std::string("RCCL not found: ") + dlerror(); return false; }
        bool ok = true;
        auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) { ok = false; why = std::string("RCCL symbol missing: ") + n; } return p; };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        Broadcast = (decltype(Broadcast))sym("ncclBroadcast");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!ok) { lib = nullptr; return false; }
        return true;
    }
};
Rccl g_rccl;
std::once_flag g_rccl_once;
int need_rccl()
{
    std::call_once(g_rccl_once, [] { g_rccl.load(); });
    if (!g_rccl.lib) return abi_fail(AM_ERR_UNSUPPORTED, g_rccl.why.empty() ? std::string("RCCL could not be loaded") : g_rccl.why);
    return AM_OK;
}
}  // namespace
#define ncclGetUniqueId g_rccl.GetUniqueId
#define ncclCommInitAll g_rccl.CommInitAll
#define ncclCommInitRank g_rccl.CommInitRank
#define ncclCommDestroy g_rccl.CommDestroy
#define ncclBroadcast g_rccl.Broadcast
#define ncclAllReduce g_rccl.AllReduce
#define ncclGroupStart g_rccl.GroupStart
#define ncclGroupEnd g_rccl.GroupEnd
#define ncclGetErrorString g_rccl.GetErrorString

static_assert(AM_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "am_multi_unique_id hands out an ncclUniqueId");

#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) return abi_fail(AM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define NCCL_TRY(expr)                                                                                          \
    do {                                                                                                        \
        ncclResult_t r_ = (expr);                                                                               \
        if (r_ != ncclSuccess) return abi_fail(AM_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
    } while (0)
#define AM_TRY(expr) do { int rc_ = (expr); if (rc_ != AM_OK) return rc_; } while (0)

struct am_multi {
    int world = 0;                    // devices in all
    int first_rank = 0;               // global rank of local device 0
    std::vector<int> devs;            // HIP device ids this process drives
    std::vector<ncclComm_t> comms;    // one communicator per local device
    std::vector<hipStream_t> streams;
    std::vector<uint64_t*> small;     // 4 KiB of device memory per local device (sizes, counters)
};

namespace {
constexpr size_t kSmallBytes = 4096 + 64;      // 512 values of an all-reduce + the error flag word that travels with them

struct DeviceGuard {
    int prev = 0;
    DeviceGuard() { (void)hipGetDevice(&prev); }
    ~DeviceGuard() { (void)hipSetDevice(prev); }
};

int finish_create(am_multi* m)
{
    const size_t n = m->devs.size();
    m->streams.assign(n, nullptr); m->small.assign(n, nullptr);
    for (size_t i = 0; i < n; i++) {
        HIP_TRY(hipSetDevice(m->devs[i]));
        HIP_TRY(hipStreamCreateWithFlags(&m->streams[i], hipStreamNonBlocking));
        HIP_TRY(hipMalloc((void**)&m->small[i], kSmallBytes));
        HIP_TRY(hipMemset(m->small[i], 0, kSmallBytes));
    }
    return AM_OK;
}

int rccl_self_check(am_multi* m);      // (below, behind allreduce_flagged)

int sync_all(const am_multi* m)
{
    for (size_t i = 0; i < m->devs.size(); i++) { HIP_TRY(hipSetDevice(m->devs[i])); HIP_TRY(hipStreamSynchronize(m->streams[i])); }
    return AM_OK;
}
}  // namespace

extern "C" int am_multi_unique_id(uint8_t id_out[AM_UNIQUE_ID_BYTES])
{
    if (!id_out) return abi_fail(AM_ERR_INVALID, "id_out is null");
    AM_TRY(need_rccl());
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    std::memcpy(id_out, &id, AM_UNIQUE_ID_BYTES);
    return AM_OK;
}

extern "C" int am_multi_create(int n_devices, am_multi** out)
{
    if (!out) return abi_fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return abi_fail(AM_ERR_NO_DEVICE, "no HIP device; libam has no CPU path");
    if (n_devices == 0) n_devices = have;
    if (n_devices < 0 || n_devices > have) return abi_fail(AM_ERR_INVALID, "n_devices exceeds the visible devices");
    AM_TRY(need_rccl());
    DeviceGuard guard;
    am_multi* m = new am_multi();
    m->world = n_devices; m->first_rank = 0;
    for (int d = 0; d < n_devices; d++) m->devs.push_back(d);
    m->comms.assign(n_devices, nullptr);
    ncclResult_t r = ncclCommInitAll(m->comms.data(), n_devices, m->devs.data());
    if (r != ncclSuccess) { m->comms.clear(); am_multi_destroy(m); return abi_fail(AM_ERR_HIP, std::string("ncclCommInitAll: ") + ncclGetErrorString(r)); }
    int rc = finish_create(m);
    if (rc == AM_OK) rc = rccl_self_check(m);
    if (rc != AM_OK) { am_multi_destroy(m); return rc; }
    *out = m;
    return AM_OK;
}

extern "C" int am_multi_create_rank(int n_ranks, int rank, const uint8_t id[AM_UNIQUE_ID_BYTES], am_multi** out)
{
    if (!out) return abi_fail(AM_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!id || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return abi_fail(AM_ERR_INVALID, "bad rank / id");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return abi_fail(AM_ERR_NO_DEVICE, "no HIP device; libam has no CPU path");
    AM_TRY(need_rccl());
    am_multi* m = new am_multi();
    m->world = n_ranks; m->first_rank = rank;
    m->devs.push_back(dev);
    m->comms.assign(1, nullptr);
    ncclUniqueId uid;
    std::memcpy(&uid, id, AM_UNIQUE_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&m->comms[0], n_ranks, uid, rank);
    if (r != ncclSuccess) { m->comms.clear(); am_multi_destroy(m); return abi_fail(AM_ERR_HIP, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); }
    int rc = finish_create(m);
    if (rc == AM_OK) rc = rccl_self_check(m);
    if (rc != AM_OK) { am_multi_destroy(m); return rc; }
    *out = m;
    return AM_OK;
}

extern "C" void am_multi_destroy(am_multi* m)
{
    if (!m) return;
    DeviceGuard guard;
    for (size_t i = 0; i < m->devs.size(); i++) {
        (void)hipSetDevice(m->devs[i]);
        if (i < m->streams.size() && m->streams[i]) { (void)hipStreamSynchronize(m->streams[i]); (void)hipStreamDestroy(m->streams[i]); }
        if (i < m->small.size() && m->small[i]) (void)hipFree(m->small[i]);
        if (i < m->comms.size() && m->comms[i]) (void)ncclCommDestroy(m->comms[i]);
    }
    delete m;
}

extern "C" int am_multi_local_devices(const am_multi* m) { return m ? (int)m->devs.size() : 0; }
extern "C" int am_multi_world_size(const am_multi* m) { return m ? m->world : 0; }
extern "C" int am_multi_device(const am_multi* m, int i) { return (m && i >= 0 && i < (int)m->devs.size()) ? m->devs[i] : -1; }

namespace {
// Collectives must be entered by EVERY rank, whatever went wrong locally: a rank that returns before the matching
// ncclBroadcast / ncclAllReduce leaves the others blocked for ever.  So local failures are carried INTO the
// collective as a flag and the decision is taken afterwards, identically on every rank; and a group that was started
// is always ended.
template <class F> ncclResult_t in_group(int n, F f)
{
    ncclResult_t first = ncclGroupStart();
    if (first != ncclSuccess) return first;
    for (int i = 0; i < n; i++) { const ncclResult_t r = f(i); if (r != ncclSuccess && first == ncclSuccess) first = r; }
    const ncclResult_t e = ncclGroupEnd();
    return first != ncclSuccess ? first : e;
}

// values[i * count .. +count) of local device i are summed over all devices; word `count` of every row is the error flag:
// local_rc != AM_OK contributes 1.  Returns local_rc if that failed, the collective's own error, or AM_ERR_HIP "a peer failed".
int allreduce_flagged(am_multi* m, uint64_t* values, size_t count, int local_rc)
{
    const int n = (int)m->devs.size();
    DeviceGuard guard;
    const std::string local_msg = local_rc != AM_OK ? std::string(am_last_error()) : std::string();
    std::vector<uint64_t> row(count + 1);
    int rc = AM_OK;
    // (a staging failure on one device must not leave the others with the previous call's row -- possibly flag 0: from the first failure on
    // every device of this process gets {0..., 1}, so the peers see the failure whatever this process could still stage)
    for (int i = 0; i < n; i++) {
        const bool bad = local_rc != AM_OK || rc != AM_OK;
        for (size_t k = 0; k < count; k++) row[k] = bad ? 0 : values[(size_t)i * count + k];
        row[count] = bad ? 1 : 0;
        hipError_t e = hipSetDevice(m->devs[i]);
        if (e == hipSuccess) e = hipMemcpy(m->small[i], row.data(), (count + 1) * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess && rc == AM_OK) {
            rc = abi_fail(AM_ERR_HIP, std::string("all-reduce staging: ") + hipGetErrorString(e));      // still enter the collective below
            for (int j = 0; j < i; j++) {                        // the rows already staged as "fine": flag them, too
                const uint64_t one = 1;
                if (hipSetDevice(m->devs[j]) == hipSuccess) (void)hipMemcpy(m->small[j] + count, &one, 8, hipMemcpyHostToDevice);
            }
        }
    }
    const ncclResult_t r = in_group(n, [&](int i) { return ncclAllReduce(m->small[i], m->small[i], count + 1, ncclUint64, ncclSum, m->comms[i], m->streams[i]); });
    if (r != ncclSuccess && rc == AM_OK) rc = abi_fail(AM_ERR_HIP, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
    if (r == ncclSuccess) { const int s = sync_all(m); if (rc == AM_OK) rc = s; }
    uint64_t failed = 0;
    for (int i = 0; i < n && rc == AM_OK; i++) {
        hipError_t e = hipSetDevice(m->devs[i]);
        if (e == hipSuccess) e = hipMemcpy(row.data(), m->small[i], (count + 1) * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rc = abi_fail(AM_ERR_HIP, std::string("all-reduce result: ") + hipGetErrorString(e)); break; }
        for (size_t k = 0; k < count; k++) values[(size_t)i * count + k] = row[k];
        failed = row[count];
    }
    if (local_rc != AM_OK) return abi_fail(local_rc, local_msg);
    if (rc != AM_OK) return rc;
    if (failed) return abi_fail(AM_ERR_HIP, std::to_string(failed) + " device(s) of the job failed before the collective; see their own error messages");
    return AM_OK;
}
}  // namespace

// The flattened automaton over xGMI: size first (8 bytes; 0 = the root could not produce its image), then an all-reduced
// "everybody has room" flag, then the blob, then an all-reduced "everybody attached" flag -- so every rank takes part in
// every collective and all ranks return the same verdict.
extern "C" int am_multi_broadcast_automaton(am_multi* m, const am_automaton* a, int case_mode, int root, am_automaton** autos_out)
{
    if (!m || !autos_out) return abi_fail(AM_ERR_INVALID, "null arguments");
    if (root < 0 || root >= m->world) return abi_fail(AM_ERR_INVALID, "root out of range");       // same arguments on every rank: all return here
    const int n = (int)m->devs.size();
    for (int i = 0; i < n; i++) autos_out[i] = nullptr;
    const int root_local = root - m->first_rank;                  // index of the root among this process's devices, if it is here
    const bool have_root = root_local >= 0 && root_local < n;
    DeviceGuard guard;
    int rc = AM_OK;
    uint64_t nbytes = 0;
    if (have_root) {
        size_t sz = 0;
        if (!a) rc = abi_fail(AM_ERR_INVALID, "the root needs the automaton");
        else if (hipSetDevice(m->devs[root_local]) != hipSuccess) rc = abi_fail(AM_ERR_HIP, "hipSetDevice(root) failed");
        else rc = am_automaton_image_size(a, case_mode, &sz);
        nbytes = rc == AM_OK ? sz : 0;                             // 0 tells every rank that the root failed
        if (hipSetDevice(m->devs[root_local]) != hipSuccess || hipMemcpy(m->small[root_local], &nbytes, 8, hipMemcpyHostToDevice) != hipSuccess) {
            if (rc == AM_OK) rc = abi_fail(AM_ERR_HIP, "automaton broadcast: could not stage the image size");   // the peers then receive whatever small[] held: the flag round below catches it
        }
    }
    {
        const ncclResult_t r = in_group(n, [&](int i) { return ncclBroadcast(m->small[i], m->small[i], 8, ncclUint8, root, m->comms[i], m->streams[i]); });
        if (r != ncclSuccess && rc == AM_OK) rc = abi_fail(AM_ERR_HIP, std::string("ncclBroadcast(size): ") + ncclGetErrorString(r));
        if (r == ncclSuccess) { const int s = sync_all(m); if (rc == AM_OK) rc = s; }
    }
    if (rc == AM_OK && !have_root) {
        if (hipSetDevice(m->devs[0]) != hipSuccess || hipMemcpy(&nbytes, m->small[0], 8, hipMemcpyDeviceToHost) != hipSuccess) rc = abi_fail(AM_ERR_HIP, "automaton broadcast: could not read the image size");
    }
    if (rc == AM_OK && nbytes == 0) rc = abi_fail(AM_ERR_HIP, "automaton broadcast: the root rank could not produce its image");
    if (rc == AM_OK && (nbytes < 64 || nbytes > (1ull << 40))) rc = abi_fail(AM_ERR_HIP, "automaton broadcast: implausible image size received");
    std::vector<void*> blob(n, nullptr);
    auto release = [&]() { for (int i = 0; i < n; i++) if (blob[i]) { (void)hipSetDevice(m->devs[i]); (void)hipFree(blob[i]); blob[i] = nullptr; } };
    for (int i = 0; i < n && rc == AM_OK; i++) {
        hipError_t e = hipSetDevice(m->devs[i]);
        if (e == hipSuccess) e = hipMalloc(&blob[i], nbytes);
        if (e != hipSuccess) rc = abi_fail(AM_ERR_OOM, std::string("hipMalloc(image copy): ") + hipGetErrorString(e));
    }
    if (rc == AM_OK && have_root) { (void)hipSetDevice(m->devs[root_local]); rc = am_automaton_image_copy(a, case_mode, blob[root_local], nbytes); }
    rc = allreduce_flagged(m, nullptr, 0, rc);                     // does every rank have the size, the room and (the root) the image?
    if (rc != AM_OK) { release(); return rc; }                     // the same decision on every rank: nobody waits in the blob broadcast
    {
        const ncclResult_t r = in_group(n, [&](int i) { return ncclBroadcast(blob[i], blob[i], nbytes, ncclUint8, root, m->comms[i], m->streams[i]); });
        if (r != ncclSuccess) rc = abi_fail(AM_ERR_HIP, std::string("ncclBroadcast(image): ") + ncclGetErrorString(r));
        else rc = sync_all(m);
    }
    for (int i = 0; i < n && rc == AM_OK; i++) {
        (void)hipSetDevice(m->devs[i]);
        rc = am_automaton_from_image(blob[i], nbytes, &autos_out[i]);        // copies the blob: the handle owns its image
    }
    release();
    rc = allreduce_flagged(m, nullptr, 0, rc);                     // all ranks agree on whether the job has its automata
    if (rc != AM_OK) for (int i = 0; i < n; i++) { am_automaton_destroy(autos_out[i]); autos_out[i] = nullptr; }
    return rc;
}

namespace {
// A handle is only handed out when its communicator has carried one collective whose answer is known: every rank adds (its number + 1) in a one-word all-reduce and
// must read world * (world + 1) / 2 on every local device.  A communicator that connects the wrong ranks, sums on the wrong stream or returns before the data has
// arrived fails here, loudly, and not as a wrong match count at the end of a job (the real ncclAllReduce had never run with more than one rank before round 6's 8-GPU run).
int rccl_self_check(am_multi* m)
{
    const int n = (int)m->devs.size();
    std::vector<uint64_t> v((size_t)n);
    for (int i = 0; i < n; i++) v[(size_t)i] = (uint64_t)(m->first_rank + i) + 1u;
    AM_TRY(allreduce_flagged(m, v.data(), 1, AM_OK));
    const uint64_t want = (uint64_t)m->world * ((uint64_t)m->world + 1u) / 2u;
    for (int i = 0; i < n; i++)
        if (v[(size_t)i] != want)
            return abi_fail(AM_ERR_HIP, "RCCL self-check failed: the all-reduce of (rank + 1) over " + std::to_string(m->world) + " ranks gave " + std::to_string(v[(size_t)i]) +
                                        " on local device " + std::to_string(i) + ", not " + std::to_string(want));
    return AM_OK;
}
}  // namespace

extern "C" int am_multi_allreduce_sum(am_multi* m, uint64_t* values, size_t count)
{
    if (!m || (count && !values)) return abi_fail(AM_ERR_INVALID, "null arguments");
    if (count > 512) return abi_fail(AM_ERR_INVALID, "too many values for one all-reduce (512 at most)");
    if (count == 0) return AM_OK;
    return allreduce_flagged(m, values, count, AM_OK);
}

namespace {
// contiguous block of haystacks of local device i: blocks differ by at most one haystack
void block_of(size_t n_hay, int i, int n, size_t* lo, size_t* hi) { *lo = n_hay * (size_t)i / (size_t)n; *hi = n_hay * (size_t)(i + 1) / (size_t)n; }

struct Job { int rc = AM_OK; std::string err; };

// runs f(i) for every local device on its own host thread with that device current
template <class F> int per_device(const am_multi* m, F f)
{
    const int n = (int)m->devs.size();
    std::vector<Job> jobs(n);
    auto body = [&](int i) {
        if (hipSetDevice(m->devs[i]) != hipSuccess) { jobs[i].rc = AM_ERR_HIP; jobs[i].err = "hipSetDevice failed"; return; }
        jobs[i].rc = f(i);
        if (jobs[i].rc != AM_OK) jobs[i].err = am_last_error();           // thread-local: carry it to the caller's thread
    };
    if (n == 1) { DeviceGuard guard; body(0); }
    else {
        std::vector<std::thread> pool;
        for (int i = 0; i < n; i++) pool.emplace_back(body, i);
        for (auto& t : pool) t.join();
    }
    for (int i = 0; i < n; i++) if (jobs[i].rc != AM_OK) return abi_fail(jobs[i].rc, "device " + std::to_string(m->devs[i]) + ": " + jobs[i].err);
    return AM_OK;
}
}  // namespace

extern "C" int am_multi_count(am_multi* m, am_automaton* const* autos, int case_mode, const am_slice* hay, size_t n_hay, uint64_t* counts_out, uint64_t* total_out)
{
    if (!m || !autos || (n_hay && !hay)) return abi_fail(AM_ERR_INVALID, "null arguments");
    const int n = (int)m->devs.size();
    std::vector<uint64_t> totals(n, 0);
    const int local = per_device(m, [&](int i) -> int {
        size_t lo, hi; block_of(n_hay, i, n, &lo, &hi);
        if (hi == lo) return AM_OK;
        am_batch* b = nullptr;
        AM_TRY(am_batch_upload(hay + lo, hi - lo, &b));
        const int rc = am_count_batch(autos[i], case_mode, b, counts_out ? counts_out + lo : nullptr, &totals[i]);
        am_batch_destroy(b);
        return rc;
    });
    AM_TRY(allreduce_flagged(m, totals.data(), 1, local));                // final gather of match counts; entered even after a local failure
    if (total_out) *total_out = totals[0];
    return AM_OK;
}

extern "C" int am_multi_batch_upload(const am_multi* m, int local_device, const am_slice* hay, size_t n_hay, am_batch** out)
{
    if (!m || !out) return abi_fail(AM_ERR_INVALID, "null arguments");
    *out = nullptr;
    if (local_device < 0 || local_device >= (int)m->devs.size()) return abi_fail(AM_ERR_INVALID, "local_device out of range");
    DeviceGuard guard;
    HIP_TRY(hipSetDevice(m->devs[local_device]));
    return am_batch_upload(hay, n_hay, out);                 // a batch lives on the device that is current when it is made
}

// Device-resident variants: batches[i] already lives on local device i (am_batch_upload / am_batch_from_device made there),
// one host thread and stream per device, nothing crosses PCIe but the counts.  This is BASELINE configs[3] from C: every
// GPU holds its share of the haystacks in HBM.
extern "C" int am_multi_count_batch(am_multi* m, am_automaton* const* autos, int case_mode, am_batch* const* batches, uint64_t* const* counts_out,
                                    uint64_t* local_totals_out, uint64_t* total_out)
{
    if (!m || !autos || !batches) return abi_fail(AM_ERR_INVALID, "null arguments");
    const int n = (int)m->devs.size();
    std::vector<uint64_t> totals(n, 0);
    const int local = per_device(m, [&](int i) -> int {
        if (!batches[i]) return AM_OK;                                     // a device without work still takes part in the all-reduce
        return am_count_batch(autos[i], case_mode, batches[i], counts_out ? counts_out[i] : nullptr, &totals[i]);
    });
    if (local_totals_out && local == AM_OK) for (int i = 0; i < n; i++) local_totals_out[i] = totals[i];
    AM_TRY(allreduce_flagged(m, totals.data(), 1, local));
    if (total_out) *total_out = totals[0];
    return AM_OK;
}

extern "C" int am_multi_run_batch(am_multi* m, am_automaton* const* autos, int case_mode, am_batch* const* batches, am_matches** results_out, uint64_t* total_records_out)
{
    if (!m || !autos || !batches || !results_out) return abi_fail(AM_ERR_INVALID, "null arguments");
    const int n = (int)m->devs.size();
    for (int i = 0; i < n; i++) results_out[i] = nullptr;
    const int local = per_device(m, [&](int i) -> int {
        if (!batches[i]) return AM_OK;
        return am_run_batch(autos[i], case_mode, batches[i], &results_out[i]);      // the records stay in that device's HBM
    });
    std::vector<uint64_t> sizes(n, 0);
    if (local == AM_OK) for (int i = 0; i < n; i++) sizes[i] = results_out[i] ? am_matches_size(results_out[i]) : 0;
    const int rc = allreduce_flagged(m, sizes.data(), 1, local);
    if (rc != AM_OK) { for (int i = 0; i < n; i++) { am_matches_free(results_out[i]); results_out[i] = nullptr; } return rc; }
    if (total_records_out) *total_records_out = sizes[0];
    return AM_OK;
}

extern "C" int am_multi_run(am_multi* m, am_automaton* const* autos, int case_mode, const am_slice* hay, size_t n_hay, am_match** matches_out, size_t* n_out)
{
    if (!m || !autos || !matches_out || !n_out || (n_hay && !hay)) return abi_fail(AM_ERR_INVALID, "null arguments");
    *matches_out = nullptr; *n_out = 0;
    const int n = (int)m->devs.size();
    std::vector<am_matches*> res(n, nullptr);
    struct Free { std::vector<am_matches*>& r; ~Free() { for (am_matches* x : r) am_matches_free(x); } } free_results{res};
    AM_TRY(per_device(m, [&](int i) -> int {
        size_t lo, hi; block_of(n_hay, i, n, &lo, &hi);
        if (hi == lo) return AM_OK;
        AM_TRY(am_run(autos[i], case_mode, hay + lo, hi - lo, &res[i]));
        return am_matches_data(res[i]) || am_matches_size(res[i]) == 0 ? AM_OK : AM_ERR_HIP;      // D2H on the device's own thread
    }));
    size_t total = 0;
    for (int i = 0; i < n; i++) total += (size_t)am_matches_size(res[i]);
    am_match* all = (am_match*)std::malloc((total ? total : 1) * sizeof(am_match));
    if (!all) return abi_fail(AM_ERR_OOM, "malloc(matches) failed");
    size_t at = 0;
    for (int i = 0; i < n; i++) {                                         // host concatenation in haystack order
        const size_t k = (size_t)am_matches_size(res[i]);
        if (!k) continue;
        size_t lo, hi; block_of(n_hay, i, n, &lo, &hi);
        const am_match* src = am_matches_data(res[i]);
        for (size_t j = 0; j < k; j++) { all[at + j] = src[j]; all[at + j].haystack += (uint32_t)lo; }
        at += k;
    }
    *matches_out = all; *n_out = total;
    return AM_OK;
}

extern "C" void am_multi_matches_free(am_match* p) { std::free(p); }

// ---- ONE haystack on all devices: global device g of W owns the end positions in (len * g / W, len * (g + 1) / W]; it uploads and
// scans only its window of the text (am_run_range: the range plus one maximal match of overlap before it), cuts its own range out of the sorted
// records on the device and rebases them.  No data-path collective; the counts are all-reduced.
namespace {
void range_of(uint64_t len, int g, int world, uint64_t* lo, uint64_t* hi) { *lo = (uint64_t)((unsigned __int128)len * (unsigned)g / (unsigned)world); *hi = (uint64_t)((unsigned __int128)len * (unsigned)(g + 1) / (unsigned)world); }
}  // namespace

extern "C" int am_multi_count_single(am_multi* m, am_automaton* const* autos, int case_mode, const am_slice* hay, uint64_t* local_counts_out, uint64_t* total_out)
{
    if (!m || !autos || !hay) return abi_fail(AM_ERR_INVALID, "null arguments");
    const int n = (int)m->devs.size();
    std::vector<uint64_t> totals(n, 0);
    const int local = per_device(m, [&](int i) -> int {
        uint64_t lo, hi; range_of(hay->len, m->first_rank + i, m->world, &lo, &hi);
        return am_count_range(autos[i], case_mode, hay, lo, hi, &totals[i]);
    });
    if (local_counts_out && local == AM_OK) for (int i = 0; i < n; i++) local_counts_out[i] = totals[i];
    AM_TRY(allreduce_flagged(m, totals.data(), 1, local));
    if (total_out) *total_out = totals[0];
    return AM_OK;
}

extern "C" int am_multi_run_single(am_multi* m, am_automaton* const* autos, int case_mode, const am_slice* hay, am_match** matches_out, size_t* n_out, uint64_t* total_records_out)
{
    if (!m || !autos || !hay || !matches_out || !n_out) return abi_fail(AM_ERR_INVALID, "null arguments");
    *matches_out = nullptr; *n_out = 0;
    const int n = (int)m->devs.size();
    std::vector<am_matches*> res(n, nullptr);
    struct Free { std::vector<am_matches*>& r; ~Free() { for (am_matches* x : r) am_matches_free(x); } } free_results{res};
    const int local = per_device(m, [&](int i) -> int {
        uint64_t lo, hi; range_of(hay->len, m->first_rank + i, m->world, &lo, &hi);
        AM_TRY(am_run_range(autos[i], case_mode, hay, lo, hi, &res[i]));
        return am_matches_data(res[i]) || am_matches_size(res[i]) == 0 ? AM_OK : AM_ERR_HIP;      // D2H on the device's own thread
    });
    std::vector<uint64_t> sizes(n, 0);
    if (local == AM_OK) for (int i = 0; i < n; i++) sizes[i] = am_matches_size(res[i]);
    size_t total = 0;
    for (int i = 0; i < n; i++) total += (size_t)sizes[i];
    AM_TRY(allreduce_flagged(m, sizes.data(), 1, local));                 // (every rank enters it, whatever happened locally)
    am_match* all = (am_match*)std::malloc((total ? total : 1) * sizeof(am_match));
    if (!all) return abi_fail(AM_ERR_OOM, "malloc(matches) failed");
    size_t at = 0;
    for (int i = 0; i < n; i++) {                                         // the local devices' ranges follow each other: position order
        const size_t k = (size_t)am_matches_size(res[i]);
        if (k) std::memcpy(all + at, am_matches_data(res[i]), k * sizeof(am_match));
        at += k;
    }
    *matches_out = all; *n_out = total;
    if (total_records_out) *total_records_out = sizes[0];
    return AM_OK;
}
