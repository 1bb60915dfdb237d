#include "comm.hpp"
#include <rccl/rccl.h>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace i3d {

// ---- RCCL -------------------------------------------------------------------------------------------------------------
struct RcclComm : Comm {
    ncclComm_t comm = nullptr;
    ~RcclComm() override { if (comm) ncclCommDestroy(comm); }
    int allreduce_sum(double* dev, size_t n, hipStream_t st) override {
        return ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, comm, st) == ncclSuccess ? 0 : 1;
    }
    int allgather(float* dev, size_t count, hipStream_t st) override {
        return ncclAllGather(dev + (size_t)rank * count, dev, count, ncclFloat, comm, st) == ncclSuccess ? 0 : 1;
    }
    int allreduce_allgather(double* red, size_t n, float* vec, size_t count, hipStream_t st) override {      // one grouped launch
        if (ncclGroupStart() != ncclSuccess) return 1;
        const ncclResult_t a = ncclAllReduce(red, red, n, ncclDouble, ncclSum, comm, st);
        const ncclResult_t b = ncclAllGather(vec + (size_t)rank * count, vec, count, ncclFloat, comm, st);
        const ncclResult_t e = ncclGroupEnd();
        return (a == ncclSuccess && b == ncclSuccess && e == ncclSuccess) ? 0 : 1;
    }
};

int rccl_unique_id(void* out, size_t* bytes) {
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
    std::memcpy(out, &id, sizeof(id)); *bytes = sizeof(id);
    return 0;
}

Comm* make_rccl_comm(int rank, int world, const void* unique_id, size_t id_bytes, hipStream_t, char* err, size_t errlen) {
    if (id_bytes != sizeof(ncclUniqueId)) { std::snprintf(err, errlen, "RCCL unique id has %zu bytes, expected %zu", id_bytes, sizeof(ncclUniqueId)); return nullptr; }
    ncclUniqueId id; std::memcpy(&id, unique_id, sizeof(id));
    auto* c = new RcclComm; c->rank = rank; c->world = world;
    const ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { std::snprintf(err, errlen, "ncclCommInitRank: %s", ncclGetErrorString(r)); c->comm = nullptr; delete c; return nullptr; }
    return c;
}

// ---- single-GPU simulation ----------------------------------------------------------------------------------------------
struct SimShared {
    int world = 1;
    std::mutex m; std::condition_variable cv;
    int arrived = 0; long generation = 0;
    std::vector<void*> ptr; std::vector<double> sum;
    void barrier() {
        std::unique_lock<std::mutex> lk(m);
        const long gen = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};
SimShared* sim_create(int world) { auto* s = new SimShared; s->world = world; s->ptr.assign(world, nullptr); return s; }
void sim_destroy(SimShared* s) { delete s; }

struct SimComm : Comm {
    SimShared* sh = nullptr;
    int allreduce_sum(double* dev, size_t n, hipStream_t st) override {
        if (hipStreamSynchronize(st) != hipSuccess) return 1;
        std::vector<double> mine(n);
        if (hipMemcpy(mine.data(), dev, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
        { std::lock_guard<std::mutex> lk(sh->m); if (sh->sum.size() != n) sh->sum.assign(n, 0.0); }
        sh->barrier();
        // fixed rank order -> every rank sees the same bits
        for (int k = 0; k < world; ++k) { if (k == rank) { std::lock_guard<std::mutex> lk(sh->m); for (size_t i = 0; i < n; ++i) sh->sum[i] += mine[i]; } sh->barrier(); }
        std::vector<double> out(sh->sum.begin(), sh->sum.begin() + n);
        sh->barrier();
        if (rank == 0) { std::lock_guard<std::mutex> lk(sh->m); std::fill(sh->sum.begin(), sh->sum.end(), 0.0); }
        sh->barrier();
        return hipMemcpy(dev, out.data(), n * sizeof(double), hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
    }
    int allgather(float* dev, size_t count, hipStream_t st) override {
        if (hipStreamSynchronize(st) != hipSuccess) return 1;
        sh->ptr[rank] = dev;
        sh->barrier();
        for (int k = 0; k < world; ++k) if (k != rank) {
            const float* src = (const float*)sh->ptr[k] + (size_t)k * count;       // same device: plain device-to-device copy
            if (hipMemcpy(dev + (size_t)k * count, src, count * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess) return 1;
        }
        sh->barrier();
        return 0;
    }
};
Comm* make_sim_comm(SimShared* s, int rank) { auto* c = new SimComm; c->sh = s; c->rank = rank; c->world = s->world; return c; }

}  // namespace i3d
