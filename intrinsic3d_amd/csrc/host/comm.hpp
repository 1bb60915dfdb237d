// Collective layer of the sharded path: one process (or, for the single-GPU simulation used by the tests, one host thread) per rank.
//   RcclComm : RCCL (ncclAllReduce / ncclAllGather over xGMI) on the context's stream.
//   SimComm  : W host threads of ONE process, each with its own i3d_context on the same device; collectives through a shared
//              host-side rendezvous.  It exists so that the SPMD control flow (ownership, compute lists, rank-major vector layout,
//              reduction order) can be exercised on a 1-GPU box; it is not a production transport.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace i3d {

struct Comm {
    int rank = 0, world = 1;
    bool force = false;        // run the sharded code path (and its collectives) even with one rank: exercises the real RCCL calls on a 1-GPU box
    virtual ~Comm() {}
    // in-place sum over ranks of n doubles in device memory
    virtual int allreduce_sum(double* dev, size_t n, hipStream_t st) = 0;
    // in-place all-gather: every rank owns count floats at dev + rank*count; afterwards all world*count floats are valid everywhere
    virtual int allgather(float* dev, size_t count, hipStream_t st) = 0;
    // both of the above as one exchange (the PCG iteration boundary: 4 scalars + the preconditioned residual slices)
    virtual int allreduce_allgather(double* red, size_t n, float* vec, size_t count, hipStream_t st) {
        const int rc = allreduce_sum(red, n, st); return rc ? rc : allgather(vec, count, st);
    }
};

Comm* make_rccl_comm(int rank, int world, const void* unique_id, size_t id_bytes, hipStream_t st, char* err, size_t errlen);
int   rccl_unique_id(void* out, size_t* bytes);

struct SimShared;
SimShared* sim_create(int world);
void sim_destroy(SimShared* s);
Comm* make_sim_comm(SimShared* s, int rank);

}  // namespace i3d
