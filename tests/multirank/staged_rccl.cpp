// staged_rccl.cpp — TEST DOUBLE for librccl, used only by tests/test_multirank_gpu.py (never shipped, never the product path).
//
// RCCL refuses two ranks on the same device, and the test boxes have ONE GPU.  To run the library's real multi-rank code
// (row shards, q_eff, the lock-step launch schedule, one all-reduce per J'·) with real kernels in TWO PROCESSES on one
// GPU, the library is pointed (BH_RCCL_LIB) at this stand-in, which exports the five RCCL entry points bh_api.hip binds
// and implements ncclAllReduce(double, sum) by staging through POSIX shared memory:
//     drain the stream -> D2H into slot[rank] -> barrier -> sum the slots in RANK ORDER -> H2D -> barrier.
// Every rank adds the slots in the same order, so the result is bit-identical on all ranks (what RCCL guarantees).
// A rank that waits longer than 60 s for its peers returns ncclSystemError: a launch-schedule mismatch between ranks
// (the hazard DESIGN.md §7 describes) fails the test instead of hanging it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

namespace {
constexpr int kMaxRanks = 8;
constexpr size_t kMaxCount = 1 << 16;
struct Shm {
    std::atomic<int> arrive;
    std::atomic<int> generation;
    std::atomic<long long> n_calls[kMaxRanks];
    double slots[kMaxRanks][kMaxCount];
};
struct Comm {
    Shm* shm;
    int rank, nranks;
    double* bounce;
};

bool barrier(Comm* c) {
    const int gen = c->shm->generation.load(std::memory_order_acquire);
    if (c->shm->arrive.fetch_add(1, std::memory_order_acq_rel) + 1 == c->nranks) {
        c->shm->arrive.store(0, std::memory_order_relaxed);
        c->shm->generation.store(gen + 1, std::memory_order_release);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (c->shm->generation.load(std::memory_order_acquire) == gen) {
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return false;
    }
    return true;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "staged-%d", (int)getpid());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId, int rank) {
    const char* name = getenv("BH_STAGED_RCCL_SHM");
    if (!name || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, sizeof(Shm)) != 0) { close(fd); return ncclSystemError; }
    void* p = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    Comm* c = new Comm{static_cast<Shm*>(p), rank, nranks, nullptr};
    if (hipHostMalloc(reinterpret_cast<void**>(&c->bounce), kMaxCount * sizeof(double), hipHostMallocDefault) != hipSuccess) return ncclUnhandledCudaError;
    *comm = reinterpret_cast<ncclComm_t>(c);
    return barrier(c) ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclSuccess;
    (void)hipHostFree(c->bounce);
    munmap(c->shm, sizeof(Shm));
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || datatype != ncclDouble || op != ncclSum || count > kMaxCount) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->bounce, sendbuff, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    memcpy(c->shm->slots[c->rank], c->bounce, count * sizeof(double));
    c->shm->n_calls[c->rank].fetch_add(1, std::memory_order_relaxed);
    if (!barrier(c)) { fprintf(stderr, "staged_rccl: rank %d timed out waiting for its peers (launch schedules differ?)\n", c->rank); return ncclSystemError; }
    for (size_t i = 0; i < count; ++i) {
        double s = c->shm->slots[0][i];
        for (int r = 1; r < c->nranks; ++r) s += c->shm->slots[r][i];
        c->bounce[i] = s;
    }
    if (!barrier(c)) return ncclSystemError;     // nobody overwrites a slot before everyone has summed
    if (hipMemcpy(recvbuff, c->bounce, count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "staged_rccl error"; }

long long staged_rccl_calls(ncclComm_t comm, int rank) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    return c ? c->shm->n_calls[rank].load() : -1;
}

}  // extern "C"
