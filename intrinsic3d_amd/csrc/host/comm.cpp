#include "comm.hpp"
#include "p2p.hpp"
#include <cstdlib>
#include <rccl/rccl.h>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace i3d {

constexpr int P2P_RED_CAP = 6 * 2000 + 16;      // the camera block of the largest supported keyframe count + p.q
constexpr int P2P_HALO_CAP = 1 << 17;           // rim entries per pair and pass

// ---- RCCL -------------------------------------------------------------------------------------------------------------
struct RcclComm : Comm {
    ncclComm_t comm = nullptr;
    P2PEngine p2p; bool use_p2p = false;       // peer-to-peer mailboxes for the per-pass exchanges (verified at start-up, else RCCL carries them too)
    bool halo_rccl = false; double* d_agree = nullptr;      // this outer iteration's rim exceeds a mailbox somewhere: every rank pushes it through RCCL instead
    ~RcclComm() override { p2p.destroy(); if (d_agree) (void)hipFree(d_agree); if (comm) ncclCommDestroy(comm); }
    // a rim that does not fit the mailbox of one rank pair is not an error: the ranks agree (max over ranks) and the grouped send / receive path carries
    // this iteration's rim pushes (HALO_CAP entries instead of P2P_HALO_CAP)
    int plan_changed(const HaloPlan& h, hipStream_t st) override {
        if (!use_p2p) return 0;
        // 0 ok | 2 the rim of a pair exceeds a mailbox (re-routed through RCCL, agreed below) | anything else: a HIP / copy error, which must surface —
        // but only AFTER the agreement all-reduce, so that the collectives of the ranks stay matched (max over ranks: 1 = re-route, 2 = somebody failed)
        const int rc = p2p.set_halo_lists(h, st);
        double bad = rc == 0 ? 0.0 : (rc == 2 ? 1.0 : 2.0);
        if (!d_agree && hipMalloc((void**)&d_agree, sizeof(double)) != hipSuccess) return 1;
        if (hipMemcpyAsync(d_agree, &bad, sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) return 1;
        if (ncclAllReduce(d_agree, d_agree, 1, ncclDouble, ncclMax, comm, st) != ncclSuccess) return 1;
        if (hipMemcpyAsync(&bad, d_agree, sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 1;
        if (bad >= 2.0) return 3;                  // a rank could not install its lists (HIP error): every rank reports it
        halo_rccl = bad != 0.0;
        return 0;
    }
    int health(hipStream_t st) override { return use_p2p ? p2p.check(st) : 0; }
    bool device_reduce(P2PDev* dev) override { if (!use_p2p) return false; *dev = p2p.device(); return true; }
    bool fused_exchange(P2PDev* dev) override { if (!use_p2p || halo_rccl) return false; *dev = p2p.device(); return true; }
    int allreduce_sum(double* dev, size_t n, hipStream_t st) override {
        ++reduce_calls; reduce_bytes += 8ll * (long long)n;
        if (use_p2p && (int)n <= p2p.L.red_cap) return p2p.allreduce(dev, n, st);
        return ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, comm, st) == ncclSuccess ? 0 : 1;
    }
    int allgather(float* dev, size_t count, hipStream_t st) override {
        return ncclAllGather(dev + (size_t)rank * count, dev, count, ncclFloat, comm, st) == ncclSuccess ? 0 : 1;
    }
    // neighbours only: one grouped launch of sends / receives of the packed rim values (a few tens of KB per pair)
    int push_halo(float* vec, const HaloPlan& h, hipStream_t st) override {
        ++halo_calls; halo_bytes_sent += 8ll * h.n_send;
        if (use_p2p && !halo_rccl) return p2p.push_halo(vec, h, st);
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
    // a ladder batch: ONE grouped launch of sends / receives, nsys values per rim entry
    int push_halo_multi(float* vec0, size_t stride, const int* sys, int nsys, const HaloPlan& h, hipStream_t st) override {
        if ((use_p2p && !halo_rccl) || nsys > HALO_MULTI_MAX) return Comm::push_halo_multi(vec0, stride, sys, nsys, h, st);
        ++halo_calls; halo_bytes_sent += 8ll * h.n_send * nsys;
        if ((h.n_send == 0 && h.n_recv == 0) || nsys <= 0) return 0;
        HaloSys hs; for (int s = 0; s < HALO_MULTI_MAX; ++s) hs.id[s] = sys[s < nsys ? s : nsys - 1];
        launch_halo_pack_multi(st, h.n_send, h.d_send_idx, vec0, stride, hs, nsys, h.chunk, h.d_send_buf);
        if (ncclGroupStart() != ncclSuccess) return 1;
        bool ok = true;
        const size_t m = 2 * (size_t)nsys;
        for (int k = 0; k < world; ++k) {
            if (k == rank) continue;
            if (h.send_cnt[k] > 0) ok &= ncclSend(h.d_send_buf + m * (size_t)h.send_off[k], m * (size_t)h.send_cnt[k], ncclFloat, k, comm, st) == ncclSuccess;
            if (h.recv_cnt[k] > 0) ok &= ncclRecv(h.d_recv_buf + m * (size_t)h.recv_off[k], m * (size_t)h.recv_cnt[k], ncclFloat, k, comm, st) == ncclSuccess;
        }
        if (ncclGroupEnd() != ncclSuccess || !ok) return 1;
        launch_halo_unpack_multi(st, h.n_recv, h.d_recv_idx, h.d_recv_buf, stride, hs, nsys, h.chunk, vec0);
        return 0;
    }
};

int rccl_unique_id(void* out, size_t* bytes) {
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
    std::memcpy(out, &id, sizeof(id)); *bytes = sizeof(id);
    return 0;
}

// Mailboxes for the per-pass exchanges: created on every rank, IPC handles all-gathered through RCCL, then verified with real exchanges
// (known sums, bounded waits).  All ranks take the same decision (all-reduced); anything short of a clean pass leaves RCCL in charge.
// With I3D_TRANSPORT=p2p the mailbox transport carries the per-pass exchanges whenever this start-up test passes on every rank (default: RCCL for everything): with it
// the sharded PCG pass is the three launches of the single-rank pass, its two exchanges running INSIDE k_pcg_dir3 / k_pcg_step3 (pcg_fused.hip) — the test below
// therefore also runs that multi-workgroup exchange pattern (P2PEngine::selftest_fused).  It has run between processes on ONE device and between the ranks of the
// same-process simulation; on a multi-GPU node it is decided by this test, with bounded waits, and anything short of a clean pass leaves RCCL in charge.
// *fatal is set when this rank could not even take part in the agreement collectives (scratch allocation): the caller aborts the init instead of
// letting the ranks issue mismatched collectives.
static bool bootstrap_p2p(RcclComm* c, hipStream_t st, bool* fatal) {
    *fatal = false;
    // RCCL carries everything unless the mailboxes are ASKED for (I3D_TRANSPORT=p2p): they have run between processes on one device and between the ranks of the
    // same-process simulation, never between devices — until tests/test_gpu_multi_device.py and bench.py --gpus N have passed on a multi-GPU node the library does
    // not start a production run on a transport whose failure mode is a bounded-wait time-out in the middle of a solve (advisor finding of round 4).
    { const char* e = std::getenv("I3D_TRANSPORT"); if (!e || std::strcmp(e, "p2p") != 0) return false; }      // every rank reads the same environment: no collective is skipped one-sidedly
    // the scratch of the agreement collectives comes first and unconditionally: every rank issues the all-gather and both min-reductions whatever happens to it locally
    unsigned char* d_handles = nullptr; double* d_test = nullptr;
    if (hipMalloc((void**)&d_handles, 64 * (size_t)c->world) != hipSuccess || hipMalloc((void**)&d_test, sizeof(double) * 64) != hipSuccess) {
        if (d_handles) (void)hipFree(d_handles);
        *fatal = true; return false;
    }
    (void)hipMemset(d_handles, 0, 64 * (size_t)c->world);
    bool ok = c->p2p.create(c->rank, c->world, P2P_RED_CAP, P2P_HALO_CAP) == 0;
    unsigned char mine[64] = {0};
    if (ok) ok = c->p2p.export_handle(mine) == 0;
    if (ok) ok = hipMemcpy(d_handles + 64 * (size_t)c->rank, mine, 64, hipMemcpyHostToDevice) == hipSuccess;
    // every rank must take part in the collectives below even if it failed locally (ok is agreed on at the end)
    std::vector<unsigned char> all(64 * (size_t)c->world, 0);
    {
        const bool gathered = ncclAllGather(d_handles + 64 * (size_t)c->rank, d_handles, 64, ncclUint8, c->comm, st) == ncclSuccess;
        (void)hipStreamSynchronize(st);
        ok = gathered && ok && hipMemcpy(all.data(), d_handles, all.size(), hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (ok && c->world > 1) ok = c->p2p.attach_ipc(all.data()) == 0;
    if (ok && c->world == 1) c->p2p.ready = true;
    // agree that everyone has mapped everyone before the first peer store
    double flag = ok ? 1.0 : 0.0;
    (void)hipMemcpy(d_test, &flag, sizeof(double), hipMemcpyHostToDevice); (void)ncclAllReduce(d_test, d_test, 1, ncclDouble, ncclMin, c->comm, st); (void)hipStreamSynchronize(st);
    (void)hipMemcpy(&flag, d_test, sizeof(double), hipMemcpyDeviceToHost);
    ok = ok && flag == 1.0;
    c->p2p.spin_limit = P2P_SPIN_LIMIT_SELFTEST;
    if (ok) {        // self-test with real exchanges and bounded waits: small and full-size all-reduces, then rim pushes between every pair
        const int big = P2P_RED_CAP;
        double* d_big = nullptr; std::vector<double> hb(big);
        ok = hipMalloc((void**)&d_big, sizeof(double) * big) == hipSuccess;
        for (int rep = 0; rep < 12 && ok; ++rep) {
            const int n = rep < 8 ? 64 : big;
            for (int i = 0; i < n; ++i) hb[i] = (double)(c->rank + 1) * (i + 1) + rep;
            ok = hipMemcpy(d_big, hb.data(), sizeof(double) * n, hipMemcpyHostToDevice) == hipSuccess && c->p2p.allreduce(d_big, n, st) == 0 && c->p2p.check(st) == 0
                 && hipMemcpy(hb.data(), d_big, sizeof(double) * n, hipMemcpyDeviceToHost) == hipSuccess;
            const double tri = 0.5 * c->world * (c->world + 1);
            for (int i = 0; i < n && ok; ++i) ok = hb[i] == tri * (i + 1) + (double)rep * c->world;
        }
        if (d_big) (void)hipFree(d_big);
        // rim push: every rank owns entries [0, B) of a test vector and receives the B entries of peer k at [(k + 1) B, (k + 2) B)
        const int B = 512, W = c->world, chunk = B * (W + 1);
        HaloPlan h; h.world = W; h.chunk = chunk; h.send_cnt.assign(W, 0); h.send_off.assign(W, 0); h.recv_cnt.assign(W, 0); h.recv_off.assign(W, 0);
        std::vector<int> sidx, ridx;
        for (int k = 0; k < W; ++k) if (k != c->rank) {
            h.send_off[k] = (int)sidx.size(); h.send_cnt[k] = B; for (int i = 0; i < B; ++i) sidx.push_back(i);
            h.recv_off[k] = (int)ridx.size(); h.recv_cnt[k] = B; for (int i = 0; i < B; ++i) ridx.push_back((k + 1) * B + i);
        }
        h.n_send = (int)sidx.size(); h.n_recv = (int)ridx.size();
        int* d_idx = nullptr; float* d_vec = nullptr;
        if (ok && W > 1) {
            std::vector<float> hv(2 * (size_t)chunk);
            ok = hipMalloc((void**)&d_idx, sizeof(int) * (sidx.size() + ridx.size())) == hipSuccess && hipMalloc((void**)&d_vec, sizeof(float) * hv.size()) == hipSuccess
                 && hipMemcpy(d_idx, sidx.data(), sizeof(int) * sidx.size(), hipMemcpyHostToDevice) == hipSuccess
                 && hipMemcpy(d_idx + sidx.size(), ridx.data(), sizeof(int) * ridx.size(), hipMemcpyHostToDevice) == hipSuccess;
            h.d_send_idx = d_idx; h.d_recv_idx = d_idx ? d_idx + sidx.size() : nullptr;
            ok = ok && c->p2p.set_halo_lists(h, st) == 0;
            for (int rep = 0; rep < 4 && ok; ++rep) {
                std::fill(hv.begin(), hv.end(), -1.0f);
                for (int i = 0; i < B; ++i) { hv[i] = (float)(1000 * c->rank + i + rep); hv[(size_t)chunk + i] = -(float)(1000 * c->rank + i + rep); }
                ok = hipMemcpy(d_vec, hv.data(), sizeof(float) * hv.size(), hipMemcpyHostToDevice) == hipSuccess && c->p2p.push_halo(d_vec, h, st) == 0 && c->p2p.check(st) == 0
                     && hipMemcpy(hv.data(), d_vec, sizeof(float) * hv.size(), hipMemcpyDeviceToHost) == hipSuccess;
                for (int k = 0; k < W && ok; ++k) if (k != c->rank) for (int i = 0; i < B && ok; ++i)
                    ok = hv[(size_t)(k + 1) * B + i] == (float)(1000 * k + i + rep) && hv[(size_t)chunk + (size_t)(k + 1) * B + i] == -(float)(1000 * k + i + rep);
            }
        }
        if (d_idx) (void)hipFree(d_idx); if (d_vec) (void)hipFree(d_vec);
        // the exchange pattern of the three-launch sharded pass: many workgroups, one writer, everybody reads; rim words pushed and consumed in the same launch
        if (ok) ok = c->p2p.selftest_fused(st) == 0;
        flag = ok ? 1.0 : 0.0;
        (void)hipMemcpy(d_test, &flag, sizeof(double), hipMemcpyHostToDevice); (void)ncclAllReduce(d_test, d_test, 1, ncclDouble, ncclMin, c->comm, st); (void)hipStreamSynchronize(st);
        (void)hipMemcpy(&flag, d_test, sizeof(double), hipMemcpyDeviceToHost);
        ok = flag == 1.0;
    }
    c->p2p.spin_limit = P2P_SPIN_LIMIT;
    if (d_handles) (void)hipFree(d_handles); if (d_test) (void)hipFree(d_test);
    if (!ok) c->p2p.destroy();
    return ok;
}

Comm* make_rccl_comm(int rank, int world, const void* unique_id, size_t id_bytes, hipStream_t st, char* err, size_t errlen) {
    if (id_bytes != sizeof(ncclUniqueId)) { std::snprintf(err, errlen, "RCCL unique id has %zu bytes, expected %zu", id_bytes, sizeof(ncclUniqueId)); return nullptr; }
    ncclUniqueId id; std::memcpy(&id, unique_id, sizeof(id));
    auto* c = new RcclComm; c->rank = rank; c->world = world;
    const ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { std::snprintf(err, errlen, "ncclCommInitRank: %s", ncclGetErrorString(r)); c->comm = nullptr; delete c; return nullptr; }
    bool fatal = false;
    c->use_p2p = bootstrap_p2p(c, st, &fatal);
    if (fatal) { std::snprintf(err, errlen, "communicator bootstrap: scratch allocation failed on rank %d", rank); delete c; return nullptr; }
    c->transport = c->use_p2p ? "p2p-mailbox (per-pass exchanges) + rccl" : "rccl";
    return c;
}

// ---- single-GPU simulation ----------------------------------------------------------------------------------------------
struct SimShared {
    int world = 1;
    std::mutex m; std::condition_variable cv;
    int arrived = 0; long generation = 0;
    std::vector<void*> ptr; std::vector<double> sum;
    std::vector<const HaloPlan*> halo;
    std::vector<unsigned char*> mailbox;      // P2P mode: every rank's mailbox (same process: plain pointers)
    void barrier() {
        std::unique_lock<std::mutex> lk(m);
        const long gen = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};
SimShared* sim_create(int world) { auto* s = new SimShared; s->world = world; s->ptr.assign(world, nullptr); s->halo.assign(world, nullptr); s->mailbox.assign(world, nullptr); return s; }
void sim_destroy(SimShared* s) { delete s; }

struct SimComm : Comm {
    SimShared* sh = nullptr;
    P2PEngine p2p; bool use_p2p = false, want_p2p = false, attached = false;      // I3D_SIM_P2P=1: the per-pass exchanges through the mailbox kernels (<= 4 ranks: one hardware queue per spinning rank)
    ~SimComm() override { p2p.destroy(); }
    void attach() {        // first collective of every rank (they run concurrently): all mailboxes are registered by now
        if (attached) return; attached = true;
        sh->barrier();
        bool all = want_p2p; for (int k = 0; k < world; ++k) all = all && sh->mailbox[k] != nullptr;
        if (all) { for (int k = 0; k < world; ++k) p2p.attach_pointer(k, sh->mailbox[k]); p2p.ready = true; use_p2p = true; transport = "p2p-mailbox (rank simulation)"; }
        sh->barrier();
    }
    bool halo_too_big = false;
    int plan_changed(const HaloPlan& h, hipStream_t st) override { attach(); if (!use_p2p) return 0; const int rc = p2p.set_halo_lists(h, st); halo_too_big = rc == 2; return rc; }
    int health(hipStream_t st) override { return use_p2p ? p2p.check(st) : 0; }
    bool device_reduce(P2PDev* dev) override { attach(); if (!use_p2p) return false; *dev = p2p.device(); return true; }
    bool fused_exchange(P2PDev* dev) override { attach(); if (!use_p2p || halo_too_big) return false; *dev = p2p.device(); return true; }
    int allreduce_sum(double* dev, size_t n, hipStream_t st) override {
        ++reduce_calls; reduce_bytes += 8ll * (long long)n;
        attach();
        if (use_p2p && (int)n <= p2p.L.red_cap) return p2p.allreduce(dev, n, st);
        if (hipStreamSynchronize(st) != hipSuccess) return 1;
        std::vector<double> mine(n);
        if (hipMemcpyAsync(mine.data(), dev, n * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 1;
        { std::lock_guard<std::mutex> lk(sh->m); if (sh->sum.size() != n) sh->sum.assign(n, 0.0); }
        sh->barrier();
        // fixed rank order -> every rank sees the same bits
        for (int k = 0; k < world; ++k) { if (k == rank) { std::lock_guard<std::mutex> lk(sh->m); for (size_t i = 0; i < n; ++i) sh->sum[i] += mine[i]; } sh->barrier(); }
        std::vector<double> out(sh->sum.begin(), sh->sum.begin() + n);
        sh->barrier();
        if (rank == 0) { std::lock_guard<std::mutex> lk(sh->m); std::fill(sh->sum.begin(), sh->sum.end(), 0.0); }
        sh->barrier();
        // (every copy of the simulation goes through the rank's OWN stream and is waited for: the library's streams are non-blocking, so a null-stream
        //  hipMemcpy is not ordered against them, and a device-to-device hipMemcpy need not have finished when it returns — a rim or a gathered slice
        //  could be read before it had landed: one sharded run in a few dozen ended 6e-3 off in the cost, round 5)
        if (hipMemcpyAsync(dev, out.data(), n * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) return 1;
        return hipStreamSynchronize(st) == hipSuccess ? 0 : 1;
    }
    int allgather(float* dev, size_t count, hipStream_t st) override {
        if (hipStreamSynchronize(st) != hipSuccess) return 1;
        sh->ptr[rank] = dev;
        sh->barrier();
        for (int k = 0; k < world; ++k) if (k != rank) {
            const float* src = (const float*)sh->ptr[k] + (size_t)k * count;       // same device: plain device-to-device copy
            if (hipMemcpyAsync(dev + (size_t)k * count, src, count * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return 1;
        }
        if (hipStreamSynchronize(st) != hipSuccess) return 1;      // before the peers may go on and change their slices
        sh->barrier();
        return 0;
    }
    // same device, one process: every rank packs, then copies what the peers packed for it
    int push_halo(float* vec, const HaloPlan& h, hipStream_t st) override {
        ++halo_calls; halo_bytes_sent += 8ll * h.n_send;
        attach();
        if (use_p2p) return p2p.push_halo(vec, h, st);
        launch_halo_pack(st, h.n_send, h.d_send_idx, vec, h.chunk, h.d_send_buf);
        if (hipStreamSynchronize(st) != hipSuccess) return 1;
        sh->halo[rank] = &h;
        sh->barrier();
        for (int k = 0; k < world; ++k) {
            if (k == rank || h.recv_cnt[k] == 0) continue;
            const HaloPlan* pk = sh->halo[k];
            if (pk->send_cnt[rank] != h.recv_cnt[k]) return 1;                  // the two ends of a pair disagree about their list
            if (hipMemcpyAsync(h.d_recv_buf + 2 * (size_t)h.recv_off[k], pk->d_send_buf + 2 * (size_t)pk->send_off[rank], sizeof(float) * 2 * (size_t)h.recv_cnt[k],
                               hipMemcpyDeviceToDevice, st) != hipSuccess) return 1;
        }
        if (hipStreamSynchronize(st) != hipSuccess) return 1;      // the peers' send buffers are read: they may pack the next pass after the barrier
        sh->barrier();
        launch_halo_unpack(st, h.n_recv, h.d_recv_idx, h.d_recv_buf, h.chunk, vec);
        return 0;
    }
    int push_halo_multi(float* vec0, size_t stride, const int* sys, int nsys, const HaloPlan& h, hipStream_t st) override {
        attach();
        if (use_p2p || nsys > HALO_MULTI_MAX) return Comm::push_halo_multi(vec0, stride, sys, nsys, h, st);
        ++halo_calls; halo_bytes_sent += 8ll * h.n_send * nsys;
        HaloSys hs; for (int s = 0; s < HALO_MULTI_MAX; ++s) hs.id[s] = sys[s < nsys ? s : nsys - 1];
        launch_halo_pack_multi(st, h.n_send, h.d_send_idx, vec0, stride, hs, nsys, h.chunk, h.d_send_buf);
        if (hipStreamSynchronize(st) != hipSuccess) return 1;
        sh->halo[rank] = &h;
        sh->barrier();
        const size_t m = 2 * (size_t)nsys;
        for (int k = 0; k < world; ++k) {
            if (k == rank || h.recv_cnt[k] == 0) continue;
            const HaloPlan* pk = sh->halo[k];
            if (pk->send_cnt[rank] != h.recv_cnt[k]) return 1;
            if (hipMemcpyAsync(h.d_recv_buf + m * (size_t)h.recv_off[k], pk->d_send_buf + m * (size_t)pk->send_off[rank], sizeof(float) * m * (size_t)h.recv_cnt[k],
                               hipMemcpyDeviceToDevice, st) != hipSuccess) return 1;
        }
        if (hipStreamSynchronize(st) != hipSuccess) return 1;
        sh->barrier();
        launch_halo_unpack_multi(st, h.n_recv, h.d_recv_idx, h.d_recv_buf, stride, hs, nsys, h.chunk, vec0);
        return 0;
    }
};
Comm* make_sim_comm(SimShared* s, int rank) {
    auto* c = new SimComm; c->sh = s; c->rank = rank; c->world = s->world; c->transport = "host-mediated (rank simulation)";
    const char* e = std::getenv("I3D_SIM_P2P");
    if (e && e[0] == '1' && s->world <= 4) {      // mailboxes are registered here; the peers are attached at the first collective (all ranks are running by then)
        c->want_p2p = c->p2p.create(rank, s->world, P2P_RED_CAP, P2P_HALO_CAP) == 0;
        c->p2p.wg_cap = 256 / s->world;      // the multi-workgroup exchange kernels of ALL ranks must be resident on this one device together
        std::lock_guard<std::mutex> lk(s->m); s->mailbox[rank] = c->want_p2p ? c->p2p.mailbox : nullptr;
    }
    return c;
}

}  // namespace i3d
