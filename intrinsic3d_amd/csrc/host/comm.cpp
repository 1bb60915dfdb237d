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
        ++reduce_calls; reduce_bytes += 8ll * (long long)n;
        return ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, comm, st) == ncclSuccess ? 0 : 1;
    }
    int allgather(float* dev, size_t count, hipStream_t st) override {
        return ncclAllGather(dev + (size_t)rank * count, dev, count, ncclFloat, comm, st) == ncclSuccess ? 0 : 1;
    }
    // neighbours only: one grouped launch of sends / receives of the packed rim values (a few tens of KB per pair)
    int push_halo(float* vec, const HaloPlan& h, hipStream_t st) override {
        ++halo_calls; halo_bytes_sent += 8ll * h.n_send;
        if (h.n_send == 0 && h.n_recv == 0) return 0;
        launch_halo_pack(st, h.n_send, h.d_send_idx, vec, h.chunk, h.d_send_buf);
        if (ncclGroupStart() != ncclSuccess) return 1;
        bool ok = true;
        for (int k = 0; k < world; ++k) {
            if (k == rank) continue;
            if (h.send_cnt[k] > 0) ok &= ncclSend(h.d_send_buf + 2 * (size_t)h.send_off[k], 2 * (size_t)h.send_cnt[k], ncclFloat, k, comm, st) == ncclSuccess;
            if (h.recv_cnt[k] > 0) ok &= ncclRecv(h.d_recv_buf + 2 * (size_t)h.recv_off[k], 2 * (size_t)h.recv_cnt[k], ncclFloat, k, comm, st) == ncclSuccess;
        }
        if (ncclGroupEnd() != ncclSuccess || !ok) return 1;
        launch_halo_unpack(st, h.n_recv, h.d_recv_idx, h.d_recv_buf, h.chunk, vec);
        return 0;
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
    std::vector<const HaloPlan*> halo;
    void barrier() {
        std::unique_lock<std::mutex> lk(m);
        const long gen = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};
SimShared* sim_create(int world) { auto* s = new SimShared; s->world = world; s->ptr.assign(world, nullptr); s->halo.assign(world, nullptr); return s; }
void sim_destroy(SimShared* s) { delete s; }

struct SimComm : Comm {
    SimShared* sh = nullptr;
    int allreduce_sum(double* dev, size_t n, hipStream_t st) override {
        ++reduce_calls; reduce_bytes += 8ll * (long long)n;
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
    // same device, one process: every rank packs, then copies what the peers packed for it
    int push_halo(float* vec, const HaloPlan& h, hipStream_t st) override {
        ++halo_calls; halo_bytes_sent += 8ll * h.n_send;
        launch_halo_pack(st, h.n_send, h.d_send_idx, vec, h.chunk, h.d_send_buf);
        if (hipStreamSynchronize(st) != hipSuccess) return 1;
        sh->halo[rank] = &h;
        sh->barrier();
        for (int k = 0; k < world; ++k) {
            if (k == rank || h.recv_cnt[k] == 0) continue;
            const HaloPlan* pk = sh->halo[k];
            if (pk->send_cnt[rank] != h.recv_cnt[k]) return 1;                  // the two ends of a pair disagree about their list
            if (hipMemcpy(h.d_recv_buf + 2 * (size_t)h.recv_off[k], pk->d_send_buf + 2 * (size_t)pk->send_off[rank], sizeof(float) * 2 * (size_t)h.recv_cnt[k],
                          hipMemcpyDeviceToDevice) != hipSuccess) return 1;
        }
        sh->barrier();
        launch_halo_unpack(st, h.n_recv, h.d_recv_idx, h.d_recv_buf, h.chunk, vec);
        return 0;
    }
};
Comm* make_sim_comm(SimShared* s, int rank) { auto* c = new SimComm; c->sh = s; c->rank = rank; c->world = s->world; return c; }

}  // namespace i3d
