// Device side of the peer-to-peer mailbox transport (host/p2p.cpp): the all-reduce of a few doubles as a function a SINGLE-WORKGROUP
// kernel can call in the middle of its own work.  The PCG's boundary kernels (k_pcg_tail_a / k_pcg_tail_b, operator.hip) are such kernels:
// with the exchange inside them a sharded pass has no separate reduction launches at all (every tiny launch costs 5-10 us of GPU timeline,
// as much as a rank's share of the operator at 8 GPUs).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace i3d {
constexpr int HALO_MULTI_MAX = 6;      // = LADDER_MAX (common.hpp): vectors one rim message of a ladder batch can carry (Comm::push_halo_multi)
struct HaloSys { int id[HALO_MULTI_MAX]; };
}


namespace i3d {

constexpr int P2P_MAX_RANKS = 64;
constexpr unsigned long long P2P_SPIN_LIMIT = 60000000000ull;      // ~30 s of shader-clock ticks: a peer that is merely late (first launch in a fresh process loads the code object) is not a dead peer
constexpr unsigned long long P2P_SPIN_LIMIT_SELFTEST = 10000000000ull;   // ~5 s while the start-up self-test decides whether the transport works at all

struct P2PLayout { int world, red_cap, halo_cap; size_t off_red, off_halo, bytes; };
struct PeerPtrs { unsigned char* m[P2P_MAX_RANKS]; };             // every rank's mailbox as seen from this device
// handle passed BY VALUE to kernels; on == 0: no transport (the kernel behaves as on a single rank)
struct P2PDev { int on, me; P2PLayout L; int* err; unsigned long long* epoch_red /* device counter of the all-reduces performed so far */; unsigned long long spin_limit; PeerPtrs peers;
                int wg_cap; /* > 0: multi-workgroup exchange kernels launch at most this many workgroups (rank simulation: all ranks' polling kernels must be resident on ONE device together) */ };
// the rim of one rank for the exchanges that run inside the PCG kernels (pcg_fused.hip): flat lists over all peers, built once per outer iteration (solver.cpp shard_plan)
struct RimLists {
    int n_send, n_recv;
    const int* send_idx; const int* send_peer;      // [n_send] owned entries a peer's rows read, ascending per peer; the peer of every item
    const int* recv_idx; const int* recv_peer;      // [n_recv] foreign entries this rank's rows read; their owner
    const int* send_off; const int* recv_off;       // [world] start of every peer's run in the two lists (item i of peer k travels in word pair i - off[k] of the pair's mailbox)
};

// Mailbox words are SELF-VALIDATING (the "LL" idea of the collective libraries): every 8-byte word carries 4 bytes of payload and the low 32
// bits of the exchange's epoch, written by ONE 8-byte store and polled by 8-byte loads.  An 8-byte store is atomic, so a reader that sees the
// epoch sees the payload: no release fence (at agent / system scope that is a write-back of the whole L2, microseconds on every exchange), no
// separate flag and no acquire fence (an L2 invalidate) — the latency of an exchange is one store crossing the link plus the poll.
static __device__ inline unsigned long long* p2p_red_words(unsigned char* mb, const P2PLayout& L, int par, int sender) { return reinterpret_cast<unsigned long long*>(mb + L.off_red) + ((size_t)par * L.world + sender) * 2 * L.red_cap; }
static __device__ inline unsigned long long* p2p_halo_words(unsigned char* mb, const P2PLayout& L, int par, int sender) { return reinterpret_cast<unsigned long long*>(mb + L.off_halo) + ((size_t)par * L.world + sender) * 2 * L.halo_cap; }
static __device__ inline void p2p_put(unsigned long long* w, unsigned payload, unsigned epoch32) {
    __hip_atomic_store(w, ((unsigned long long)epoch32 << 32) | (unsigned long long)payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// bounded wait for the word of this epoch; a timeout latches *err (the host turns it into I3D_ERR_COMM) and returns 0
static __device__ inline unsigned p2p_get(unsigned long long* w, unsigned epoch32, int* err, unsigned long long spin_limit) {
    unsigned long long x = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned)(x >> 32) == epoch32) return (unsigned)x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (;;) {
        __builtin_amdgcn_s_sleep(1);
        x = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(x >> 32) == epoch32) return (unsigned)x;
        if (__builtin_readcyclecounter() - t0 > spin_limit) { atomicExch(err, 1); return 0u; }
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return 0u;      // another wait of this rank has already given up: fail fast
    }
}

// dev[0..n) <- sum over ranks in rank order (bit-identical on every rank).  Called by ALL threads of a single-workgroup kernel; n <= L.red_cap.
// The epoch is a DEVICE counter advanced by every exchange actually performed: ranks that skip the same exchanges (a solve that has
// converged skips its remaining boundary kernels on every rank alike) stay in step, and two consecutive exchanges always use the two
// different parity buffers (a rank cannot run two exchanges ahead of a peer: it needs that peer's contribution to the one in between, and the
// peer sends that only after it has consumed this one).  Epoch 0 is never used (the mailbox starts zeroed).
static __device__ inline void p2p_allreduce_wg(const P2PDev& d, double* dev, int n) {
    __shared__ unsigned long long epoch_s;
    if (threadIdx.x == 0) { unsigned long long e = *d.epoch_red + 1; if ((unsigned)e == 0u) ++e; epoch_s = e; }
    __syncthreads();
    const unsigned long long epoch = epoch_s;
    const unsigned e32 = (unsigned)epoch;
    const int par = (int)(epoch & 1ull), W = d.L.world, me = d.me;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double x = dev[i];
        const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
        for (int k = 0; k < W; ++k) { unsigned long long* dst = p2p_red_words(d.peers.m[k], d.L, par, me); p2p_put(&dst[2 * i], lo, e32); p2p_put(&dst[2 * i + 1], hi, e32); }
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double s = 0.0;
        for (int j = 0; j < W; ++j) {
            unsigned long long* src = p2p_red_words(d.peers.m[me], d.L, par, j);
            const unsigned lo = p2p_get(&src[2 * i], e32, d.err, d.spin_limit), hi = p2p_get(&src[2 * i + 1], e32, d.err, d.spin_limit);
            s += __hiloint2double((int)hi, (int)lo);
        }
        dev[i] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) *d.epoch_red = epoch;
    __syncthreads();
}

// ---- exchanges INSIDE multi-workgroup kernels (the three-launch PCG pass of a sharded run, pcg_fused.hip) ---------------------------------------------
// The workgroups of one launch cannot agree on a device counter one of them advances, so the epoch of these exchanges is handed in by the host: the pass
// number, identical on all ranks (every rank queues the same passes; a finished solve skips its exchanges on every rank alike).  Four epochs per pass:
enum { P2P_X_DIR = 0, P2P_X_STEP = 1, P2P_X_RESET_RIM = 2, P2P_X_RESET_STEP = 3 };
// (top bit set: the epochs of the counter-driven all-reduce, p2p_allreduce_wg, count up from 1 in the SAME mailbox words and never reach 2^31 — a stale pass epoch
// left in a word can therefore never equal a counter epoch after a switch from the fused pass to the legacy pass; advisor finding of round 4)
static __device__ inline unsigned p2p_pass_epoch(int seq, int which) { return 0x80000000u | (4u * (unsigned)seq + (unsigned)which); }      // never 0; parity buffer = which & 1
// A rank cannot overwrite words a peer still has to read: buffers alternate (dir: 0, step: 1, reset rim: 0, reset step: 1), and before a rank reaches the next
// exchange on the same buffer it has completed one on the other buffer, which needed every peer's contribution — sent by a LATER kernel of that peer's
// stream than the one that read the words in question.

// this rank's double #idx -> every rank's mailbox (own included); one thread per (double)
static __device__ inline void p2p_put_double_all(const P2PDev& d, int par, unsigned e32, int idx, double x) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    for (int k = 0; k < d.L.world; ++k) { unsigned long long* dst = p2p_red_words(d.peers.m[k], d.L, par, d.me); p2p_put(&dst[2 * idx], lo, e32); p2p_put(&dst[2 * idx + 1], hi, e32); }
}
// double #idx of rank j, read from this rank's own mailbox (bounded wait)
static __device__ inline double p2p_get_double(const P2PDev& d, int par, unsigned e32, int j, int idx) {
    unsigned long long* src = p2p_red_words(d.peers.m[d.me], d.L, par, j);
    const unsigned lo = p2p_get(&src[2 * idx], e32, d.err, d.spin_limit), hi = p2p_get(&src[2 * idx + 1], e32, d.err, d.spin_limit);
    return __hiloint2double((int)hi, (int)lo);
}
// NC doubles summed over the ranks in rank order, identical in every workgroup of every rank: workgroup `writer` has stored this rank's values (p2p_put_double_all);
// all threads of the workgroup call this; sm: NC * world doubles of LDS
template <int NC>
static __device__ inline void p2p_sum_all(const P2PDev& d, int par, unsigned e32, double (&tot)[NC], double* sm) {
    const int W = d.L.world;
    for (int t = threadIdx.x; t < NC * W; t += blockDim.x) sm[t] = p2p_get_double(d, par, e32, t / NC, t % NC);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NC; ++k) { double s = 0.0; for (int j = 0; j < W; ++j) s += sm[j * NC + k]; tot[k] = s; }
    __syncthreads();
}
// rim values: item i of the run this rank sends to peer k -> word pair i of [par][sender me] in k's halo region
static __device__ inline void p2p_put_rim(const P2PDev& d, int par, unsigned e32, int peer, int i, float vs, float va) {
    unsigned long long* dst = p2p_halo_words(d.peers.m[peer], d.L, par, d.me);
    p2p_put(&dst[2 * i], __float_as_uint(vs), e32); p2p_put(&dst[2 * i + 1], __float_as_uint(va), e32);
}
static __device__ inline void p2p_get_rim(const P2PDev& d, int par, unsigned e32, int peer, int i, float& vs, float& va) {
    unsigned long long* src = p2p_halo_words(d.peers.m[d.me], d.L, par, peer);
    vs = __uint_as_float(p2p_get(&src[2 * i], e32, d.err, d.spin_limit)); va = __uint_as_float(p2p_get(&src[2 * i + 1], e32, d.err, d.spin_limit));
}

}  // namespace i3d
