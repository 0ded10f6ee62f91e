// rccl_stub.cpp -- TEST INFRASTRUCTURE: a stand-in for the nine RCCL entry points csrc/am_multi.cpp binds (its dlopen table), so that the
// product's multi-GPU code -- rank bookkeeping, block bounds, the flag word that carries a local failure into every collective, image
// broadcast + attach, the count all-reduce -- runs with 2 / 4 / 8 RANKS on a box with ONE GPU: one process per rank, all on device 0, the
// "collectives" exchanged through files in a directory under /dev/shm named by the unique id.  It is built with the soname librccl.so.1 and
// loaded into the rank processes before libam touches RCCL, so am_multi.cpp's `dlopen("librccl.so.1", RTLD_NOLOAD)` finds it.  Nothing
// here measures anything, and none of it ships: what it leaves untested are exactly the real ncclBroadcast / ncclAllReduce calls.
//
// Semantics kept from RCCL as far as am_multi.cpp relies on them: every rank must enter every collective (a missing rank = the others time
// out after 60 s with ncclSystemError instead of hanging for ever), in-place buffers, stream order (the stub synchronises the stream, moves
// the bytes through the host, and returns with the result in place).  One communicator per process (ncclCommInitRank; ncclCommInitAll only
// for one device), groups are no-ops.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

struct Comm { std::string dir; int n = 1, rank = 0; uint64_t seq = 0; };

std::string dir_of(const ncclUniqueId& id)
{
    char name[64]; std::memcpy(name, id.internal, 40); name[40] = 0;
    return std::string("/dev/shm/") + name;
}

bool write_file(const std::string& path, const void* p, size_t n)
{
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = n == 0 || std::fwrite(p, 1, n, f) == n;
    std::fclose(f);
    return ok && std::rename(tmp.c_str(), path.c_str()) == 0;       // rename: a reader never sees half a file
}

bool read_file(const std::string& path, void* p, size_t n, double timeout_s = 60.0)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (f) {
            const bool ok = n == 0 || std::fread(p, 1, n, f) == n;
            std::fclose(f);
            return ok;
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

size_t size_of(ncclDataType_t t)
{
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

}  // namespace

extern "C" {

__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    std::memset(id, 0, sizeof(*id));
    std::snprintf(id->internal, 40, "am_rccl_stub_%ld_%lx", (long)getpid(), (unsigned long)std::chrono::steady_clock::now().time_since_epoch().count());
    return mkdir(dir_of(*id).c_str(), 0700) == 0 ? ncclSuccess : ncclSystemError;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    Comm* c = new Comm();
    c->dir = dir_of(id); c->n = nranks; c->rank = rank;
    struct stat st;
    if (stat(c->dir.c_str(), &st) != 0) { delete c; return ncclInvalidArgument; }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int*)
{
    if (ndev != 1) return ncclInvalidUsage;                         // one process per rank only
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    return r != ncclSuccess ? r : ncclCommInitRank(&comms[0], 1, id, 0);
}

__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete reinterpret_cast<Comm*>(comm); return ncclSuccess; }
__attribute__((visibility("default"))) ncclResult_t ncclGroupStart(void) { return ncclSuccess; }
__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd(void) { return ncclSuccess; }
__attribute__((visibility("default"))) const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : r == ncclSystemError ? "stub: a rank did not arrive (timeout) or a file could not be written" : "stub: invalid use"; }

__attribute__((visibility("default"))) ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t type, int root, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t bytes = count * size_of(type);
    if (!size_of(type) || root < 0 || root >= c->n) return ncclInvalidArgument;
    const uint64_t seq = c->seq++;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    const std::string path = c->dir + "/bc_" + std::to_string(seq);
    std::vector<uint8_t> host(bytes);
    if (c->rank == root) {
        if (bytes && hipMemcpy(host.data(), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        if (!write_file(path, host.data(), bytes)) return ncclSystemError;
        if (recvbuff != sendbuff && bytes && hipMemcpy(recvbuff, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    } else {
        if (!read_file(path, host.data(), bytes)) return ncclSystemError;
        if (bytes && hipMemcpy(recvbuff, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (type != ncclUint64 || op != ncclSum) return ncclInvalidArgument;          // what am_multi.cpp uses
    const uint64_t seq = c->seq++;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<uint64_t> mine(count), sum(count, 0), other(count);
    if (count && hipMemcpy(mine.data(), sendbuff, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    const std::string base = c->dir + "/ar_" + std::to_string(seq) + "_";
    if (!write_file(base + std::to_string(c->rank), mine.data(), count * 8)) return ncclSystemError;
    for (int r = 0; r < c->n; r++) {
        if (!read_file(base + std::to_string(r), other.data(), count * 8)) return ncclSystemError;
        for (size_t k = 0; k < count; k++) sum[k] += other[k];
    }
    // (files are a few KiB and never deleted here -- a broadcast does not synchronise the ranks, so no rank knows when the others are done reading;
    // the test harness removes the directory when every rank process has exited)
    if (count && hipMemcpy(recvbuff, sum.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

}  // extern "C"
