// Peer-to-peer mailbox transport for what the sharded PCG exchanges every pass (comm.hpp): the small fp64 all-reduce
// [camera block | p.q] / [4 iteration scalars] and the rim of the operator input.  xGMI is point to point and fully connected
// inside a node, and both messages are latency-bound (10 KB / a few tens of KB per pair): instead of a collective library's
// kernels (~40 us per launch here) every rank STORES its contribution straight into a mailbox in each peer's HBM, publishes an
// epoch flag, waits for the peers' flags and consumes — one single-workgroup kernel per exchange, no host involvement.
//   * mailbox memory is fine-grained (uncached in the consumer's L2), mapped into the peers by HIP IPC (one process per GPU) or
//     shared by pointer (rank simulation on one GPU);
//   * publish = system-scope release fence + relaxed system-scope flag store per peer; consume = relaxed polling of the own
//     flags, one system-scope acquire fence, then plain loads (MI355X_MICROARCH.md, inter-workgroup visibility — system instead
//     of agent scope because the producer is another device);
//   * two buffers by epoch parity: a rank cannot run two exchanges ahead of a peer (it needs that peer's contribution to the
//     exchange in between), so the buffer of epoch e is free again when e + 2 is written;
//   * the sum runs over the ranks in rank order on every rank: all ranks hold bit-identical results (the replicated camera
//     tail of the solver vectors stays identical without a broadcast);
//   * every spin is bounded; a timeout latches an error flag the host turns into I3D_ERR_COMM.
#include "p2p.hpp"
#include <cstdio>
#include <cstring>

namespace i3d {

constexpr unsigned long long P2P_SPIN_LIMIT = 4000000000ull;       // ~2 s of s_memtime ticks

struct PeerPtrs { unsigned char* m[P2P_MAX_RANKS]; };

static __device__ inline unsigned long long* red_flag(unsigned char* mb, int par, int sender) { return reinterpret_cast<unsigned long long*>(mb) + par * P2P_MAX_RANKS + sender; }
static __device__ inline unsigned long long* halo_flag(unsigned char* mb, int par, int sender) { return reinterpret_cast<unsigned long long*>(mb) + (2 + par) * P2P_MAX_RANKS + sender; }
static __device__ inline double* red_in(unsigned char* mb, const P2PLayout& L, int par, int sender) { return reinterpret_cast<double*>(mb + L.off_red) + ((size_t)par * L.world + sender) * L.red_cap; }
static __device__ inline float* halo_in(unsigned char* mb, const P2PLayout& L, int par, int sender) { return reinterpret_cast<float*>(mb + L.off_halo) + ((size_t)par * L.world + sender) * 2 * L.halo_cap; }

static __device__ inline bool wait_flag(unsigned long long* f, unsigned long long epoch, int* err) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_readcyclecounter() - t0 > P2P_SPIN_LIMIT) { atomicExch(err, 1); return false; }
    }
    return true;
}

// dev[0..n) <- sum over ranks, in rank order
__global__ void __launch_bounds__(1024) k_p2p_allreduce(double* __restrict__ dev, int n, int me, P2PLayout L, PeerPtrs peers, unsigned long long epoch, int* err) {
    const int par = (int)(epoch & 1ull), W = L.world;
    for (int k = 0; k < W; ++k) {
        double* dst = red_in(peers.m[k], L, par, me);
        for (int i = threadIdx.x; i < n; i += blockDim.x) __hip_atomic_store(&dst[i], dev[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if ((int)threadIdx.x < W) __hip_atomic_store(red_flag(peers.m[threadIdx.x], par, me), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)threadIdx.x < W) wait_flag(red_flag(peers.m[me], par, threadIdx.x), epoch, err);
    __syncthreads();
    __threadfence_system();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double s = 0.0;
        for (int j = 0; j < W; ++j) s += __hip_atomic_load(&red_in(peers.m[me], L, par, j)[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        dev[i] = s;
    }
}

// rim of the operator input: vec[e], vec[chunk + e] of this rank's send list -> the peers' mailboxes; their values for this rank -> vec
__global__ void __launch_bounds__(1024) k_p2p_halo(float* __restrict__ vec, int chunk, int me, P2PLayout L, PeerPtrs peers, unsigned long long epoch,
                                                   const int* __restrict__ send_idx, const int* __restrict__ send_off, const int* __restrict__ send_cnt,
                                                   const int* __restrict__ recv_idx, const int* __restrict__ recv_off, const int* __restrict__ recv_cnt, int* err) {
    const int par = (int)(epoch & 1ull), W = L.world;
    for (int k = 0; k < W; ++k) {
        const int cnt = send_cnt[k]; if (k == me || cnt == 0) continue;
        const int off = send_off[k]; float* dst = halo_in(peers.m[k], L, par, me);
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const int e = send_idx[off + i];
            __hip_atomic_store(&dst[2 * i], vec[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&dst[2 * i + 1], vec[(size_t)chunk + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __threadfence_system();
    __syncthreads();
    if ((int)threadIdx.x < W && (int)threadIdx.x != me && send_cnt[threadIdx.x] > 0)
        __hip_atomic_store(halo_flag(peers.m[threadIdx.x], par, me), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)threadIdx.x < W && (int)threadIdx.x != me && recv_cnt[threadIdx.x] > 0) wait_flag(halo_flag(peers.m[me], par, threadIdx.x), epoch, err);
    __syncthreads();
    __threadfence_system();
    for (int k = 0; k < W; ++k) {
        const int cnt = recv_cnt[k]; if (k == me || cnt == 0) continue;
        const int off = recv_off[k]; const float* src = halo_in(peers.m[me], L, par, k);
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const int e = recv_idx[off + i];
            vec[e] = __hip_atomic_load(&src[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            vec[(size_t)chunk + e] = __hip_atomic_load(&src[2 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

P2PLayout p2p_layout(int world, int red_cap, int halo_cap) {
    P2PLayout L; L.world = world; L.red_cap = red_cap; L.halo_cap = halo_cap;
    L.off_red = 4 * P2P_MAX_RANKS * sizeof(unsigned long long);
    L.off_halo = L.off_red + (size_t)2 * world * red_cap * sizeof(double);
    L.bytes = L.off_halo + (size_t)2 * world * 2 * halo_cap * sizeof(float);
    return L;
}

int P2PEngine::create(int rank_, int world_, int red_cap, int halo_cap) {
    rank = rank_; world = world_;
    if (world > P2P_MAX_RANKS) return 1;
    L = p2p_layout(world, red_cap, halo_cap);
    // fine-grained: peers' stores must not be shadowed by stale lines of this device's L2
    if (hipExtMallocWithFlags((void**)&mailbox, L.bytes, hipDeviceMallocFinegrained) != hipSuccess) { mailbox = nullptr; return 1; }
    if (hipMemset(mailbox, 0, L.bytes) != hipSuccess) return 1;
    if (hipMalloc((void**)&d_err, sizeof(int)) != hipSuccess || hipMemset(d_err, 0, sizeof(int)) != hipSuccess) return 1;
    if (hipMalloc((void**)&d_lists, sizeof(int) * 4 * P2P_MAX_RANKS) != hipSuccess) return 1;
    for (int k = 0; k < P2P_MAX_RANKS; ++k) peer[k] = nullptr;
    peer[rank] = mailbox;
    return 0;
}
int P2PEngine::export_handle(void* out64) { hipIpcMemHandle_t h; if (hipIpcGetMemHandle(&h, mailbox) != hipSuccess) return 1; std::memcpy(out64, &h, sizeof(h)); return 0; }
int P2PEngine::attach_ipc(const void* handles /* world x 64 bytes */) {
    for (int k = 0; k < world; ++k) {
        if (k == rank) continue;
        hipIpcMemHandle_t h; std::memcpy(&h, (const char*)handles + (size_t)k * sizeof(h), sizeof(h));
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return 1;
        peer[k] = (unsigned char*)p; opened[k] = true;
    }
    ready = true; return 0;
}
void P2PEngine::attach_pointer(int k, unsigned char* p) { peer[k] = p; }
void P2PEngine::destroy() {
    for (int k = 0; k < P2P_MAX_RANKS; ++k) if (opened[k] && peer[k]) { (void)hipIpcCloseMemHandle(peer[k]); opened[k] = false; }
    if (mailbox) (void)hipFree(mailbox); if (d_err) (void)hipFree(d_err); if (d_lists) (void)hipFree(d_lists);
    mailbox = nullptr; d_err = nullptr; d_lists = nullptr; ready = false;
}

static PeerPtrs ptrs_of(const P2PEngine& e) { PeerPtrs p; for (int k = 0; k < P2P_MAX_RANKS; ++k) p.m[k] = e.peer[k]; return p; }

int P2PEngine::allreduce(double* dev, size_t n, hipStream_t st) {
    if ((int)n > L.red_cap) return 1;
    ++epoch_red;
    k_p2p_allreduce<<<1, 1024, 0, st>>>(dev, (int)n, rank, L, ptrs_of(*this), epoch_red, d_err);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int P2PEngine::set_halo_lists(const HaloPlan& h, hipStream_t st) {      // once per outer iteration
    if (h.n_send > 0 || h.n_recv > 0) for (int k = 0; k < world; ++k) if (h.send_cnt[k] > L.halo_cap || h.recv_cnt[k] > L.halo_cap) return 1;
    int host[4 * P2P_MAX_RANKS]; std::memset(host, 0, sizeof(host));
    for (int k = 0; k < world; ++k) { host[k] = h.send_off[k]; host[P2P_MAX_RANKS + k] = h.send_cnt[k]; host[2 * P2P_MAX_RANKS + k] = h.recv_off[k]; host[3 * P2P_MAX_RANKS + k] = h.recv_cnt[k]; }
    if (hipMemcpyAsync(d_lists, host, sizeof(host), hipMemcpyHostToDevice, st) != hipSuccess) return 1;
    return hipStreamSynchronize(st) == hipSuccess ? 0 : 1;
}
int P2PEngine::push_halo(float* vec, const HaloPlan& h, hipStream_t st) {
    ++epoch_halo;
    k_p2p_halo<<<1, 1024, 0, st>>>(vec, h.chunk, rank, L, ptrs_of(*this), epoch_halo, h.d_send_idx, d_lists, d_lists + P2P_MAX_RANKS, h.d_recv_idx, d_lists + 2 * P2P_MAX_RANKS,
                                   d_lists + 3 * P2P_MAX_RANKS, d_err);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int P2PEngine::check(hipStream_t st) {      // has any spin timed out?  (synchronises the stream)
    int e = 0;
    if (hipMemcpyAsync(&e, d_err, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 1;
    return e;
}

}  // namespace i3d
