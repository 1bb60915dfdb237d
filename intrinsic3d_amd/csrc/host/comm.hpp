// Collective layer of the sharded path: one process (or, for the single-GPU simulation used by the tests, one host thread) per rank.
//   RcclComm : RCCL (ncclAllReduce / ncclAllGather over xGMI) on the context's stream.
//   SimComm  : W host threads of ONE process, each with its own i3d_context on the same device; collectives through a shared
//              host-side rendezvous.  It exists so that the SPMD control flow (ownership, compute lists, rank-major vector layout,
//              reduction order) can be exercised on a 1-GPU box; it is not a production transport.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <vector>
#include "../device/p2p_device.hpp"

namespace i3d {

// What a rank exchanges with its neighbours once per PCG pass: the operator input on the rim its rows read (common.hpp, sharding).
// Built once per outer iteration, identically on both ends of every pair (every rank derives it from the replicated work list):
// send_idx[send_off[k] .. +send_cnt[k]) = work-list entries this rank OWNS and rank k needs, ascending; recv_idx likewise for what it needs of k.
struct HaloPlan {
    int world = 1, chunk = 0;
    std::vector<int> send_cnt, send_off, recv_cnt, recv_off;      // per peer (entries; every entry carries its sdf and its albedo value)
    const int* d_send_idx = nullptr; const int* d_recv_idx = nullptr;
    float* d_send_buf = nullptr; float* d_recv_buf = nullptr;     // 2 floats per entry
    int n_send = 0, n_recv = 0;
};

struct Comm {
    int rank = 0, world = 1;
    bool force = false;        // run the sharded code path (and its collectives) even with one rank: exercises the real RCCL calls on a 1-GPU box
    long long halo_bytes_sent = 0, halo_calls = 0, reduce_calls = 0, reduce_bytes = 0;      // traffic log (i3d_comm_stats)
    virtual ~Comm() {}
    // in-place sum over ranks of n doubles in device memory
    virtual int allreduce_sum(double* dev, size_t n, hipStream_t st) = 0;
    // in-place all-gather of one part of a solver vector: every rank owns count floats at dev + rank*count
    virtual int allgather(float* dev, size_t count, hipStream_t st) = 0;
    // vec[e], vec[chunk + e] of the entries in the send lists -> the same positions of the peers' copies of vec (their recv lists)
    virtual int push_halo(float* vec, const HaloPlan& h, hipStream_t st) = 0;
    // the same for the nsys (<= HALO_MULTI_MAX) vectors vec0 + sys[s] * stride of a ladder batch in ONE message per peer: every rim entry carries the values of all of them
    // (the buffers of `h` hold HALO_MULTI_MAX values per entry).  Default: one exchange per vector.
    virtual int push_halo_multi(float* vec0, size_t stride, const int* sys, int nsys, const HaloPlan& h, hipStream_t st) {
        for (int s = 0; s < nsys; ++s) { const int rc = push_halo(vec0 + (size_t)sys[s] * stride, h, st); if (rc) return rc; }
        return 0;
    }
    // true: all-reduces of a few doubles can run INSIDE single-workgroup kernels (p2p_allreduce_wg with *dev); the caller logs them with count_reduce
    virtual bool device_reduce(P2PDev* dev) { (void)dev; return false; }
    // true: this outer iteration's per-pass exchanges may run inside the MULTI-workgroup PCG kernels (pcg_fused.hip: the three-launch sharded pass) — the
    // mailbox transport is up and every rank pair's rim fits its mailbox
    virtual bool fused_exchange(P2PDev* dev) { (void)dev; return false; }
    void count_reduce(size_t n) { ++reduce_calls; reduce_bytes += 8ll * (long long)n; }
    void count_halo(long long entries) { ++halo_calls; halo_bytes_sent += 8ll * entries; }
    // called once per outer iteration after the halo plan changed; 0 = ok
    virtual int plan_changed(const HaloPlan&, hipStream_t) { return 0; }
    // 0 = healthy; non-zero after a peer-to-peer wait timed out (synchronises the stream)
    virtual int health(hipStream_t) { return 0; }
    const char* transport = "";      // what carries the per-pass exchanges (for logs / bench)
};

void launch_halo_pack_multi(hipStream_t st, int n, const int* idx, const float* vec0, size_t stride, HaloSys sys, int nsys, int chunk, float* buf);
void launch_halo_unpack_multi(hipStream_t st, int n, const int* idx, const float* buf, size_t stride, HaloSys sys, int nsys, int chunk, float* vec0);
// pack / unpack kernels of the halo exchange (operator.hip)
void launch_halo_pack(hipStream_t st, int n, const int* idx, const float* vec, int chunk, float* buf);
void launch_halo_unpack(hipStream_t st, int n, const int* idx, const float* buf, int chunk, float* vec);

Comm* make_rccl_comm(int rank, int world, const void* unique_id, size_t id_bytes, hipStream_t st, char* err, size_t errlen);
int   rccl_unique_id(void* out, size_t* bytes);

struct SimShared;
SimShared* sim_create(int world);
void sim_destroy(SimShared* s);
Comm* make_sim_comm(SimShared* s, int rank);

}  // namespace i3d
