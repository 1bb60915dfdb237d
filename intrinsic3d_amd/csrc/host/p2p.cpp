// Peer-to-peer mailbox transport for what the sharded PCG exchanges every pass (comm.hpp): the small fp64 all-reduce
// [camera block | p.q] / [4 iteration scalars] and the rim of the operator input.  xGMI is point to point and fully connected
// inside a node, and both messages are latency-bound (10 KB / a few tens of KB per pair): instead of a collective library's
// kernels (~40 us per launch here) every rank STORES its contribution straight into a mailbox in each peer's HBM and polls its
// own mailbox for the peers' — no host involvement, and for the reductions not even a kernel of their own (p2p_device.hpp:
// p2p_allreduce_wg runs inside the PCG's boundary kernels).
//   * mailbox memory is fine-grained (uncached in the consumer's L2), mapped into the peers by HIP IPC (one process per GPU) or
//     shared by pointer (rank simulation on one GPU);
//   * every 8-byte mailbox word carries 4 payload bytes + the low 32 bits of the exchange's epoch and is written by one atomic
//     8-byte system-scope store / read by 8-byte system-scope loads: a word that shows the epoch shows the payload.  No fences —
//     a release fence at agent or system scope writes back the whole L2 and an acquire invalidates it, microseconds per exchange
//     (measured on one GPU: 9.7 us for the fenced flag protocol as a kernel of its own);
//   * two buffers by epoch parity: a rank cannot run two exchanges ahead of a peer (it needs that peer's contribution to the
//     exchange in between), so the buffer of epoch e is free again when e + 2 is written;
//   * the sum runs over the ranks in rank order on every rank: all ranks hold bit-identical results (the replicated camera
//     tail of the solver vectors stays identical without a broadcast);
//   * every spin is bounded; a timeout latches an error flag the host turns into I3D_ERR_COMM.
#include "p2p.hpp"
#include <cstdio>
#include <cstring>

namespace i3d {

__global__ void k_p2p_warmup(int* p) { if (p && threadIdx.x == 12345) *p = 0; }

// dev[0..n) <- sum over ranks, in rank order
__global__ void __launch_bounds__(1024) k_p2p_allreduce(double* __restrict__ dev, int n, P2PDev d) { p2p_allreduce_wg(d, dev, n); }

// rim of the operator input: vec[e], vec[chunk + e] of this rank's send list -> the peers' mailboxes; their values for this rank -> vec
__global__ void __launch_bounds__(1024) k_p2p_halo(float* __restrict__ vec, int chunk, int me, P2PLayout L, PeerPtrs peers, unsigned long long epoch,
                                                   const int* __restrict__ send_idx, const int* __restrict__ send_off, const int* __restrict__ send_cnt,
                                                   const int* __restrict__ recv_idx, const int* __restrict__ recv_off, const int* __restrict__ recv_cnt, int* err, unsigned long long spin_limit) {
    const int par = (int)(epoch & 1ull), W = L.world;
    const unsigned e32 = (unsigned)epoch;
    for (int k = 0; k < W; ++k) {
        const int cnt = send_cnt[k]; if (k == me || cnt == 0) continue;
        const int off = send_off[k]; unsigned long long* dst = p2p_halo_words(peers.m[k], L, par, me);
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const int e = send_idx[off + i];
            p2p_put(&dst[2 * i], __float_as_uint(vec[e]), e32);
            p2p_put(&dst[2 * i + 1], __float_as_uint(vec[(size_t)chunk + e]), e32);
        }
    }
    for (int k = 0; k < W; ++k) {
        const int cnt = recv_cnt[k]; if (k == me || cnt == 0) continue;
        const int off = recv_off[k]; unsigned long long* src = p2p_halo_words(peers.m[me], L, par, k);
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const int e = recv_idx[off + i];
            vec[e] = __uint_as_float(p2p_get(&src[2 * i], e32, err, spin_limit));
            vec[(size_t)chunk + e] = __uint_as_float(p2p_get(&src[2 * i + 1], e32, err, spin_limit));
        }
    }
}

P2PLayout p2p_layout(int world, int red_cap, int halo_cap) {
    P2PLayout L; L.world = world; L.red_cap = red_cap; L.halo_cap = halo_cap;
    // [2 parities][world senders][2 words per double]  |  [2 parities][world senders][2 words per rim entry (sdf, albedo)]
    L.off_red = 0;
    L.off_halo = L.off_red + (size_t)2 * world * 2 * red_cap * sizeof(unsigned long long);
    L.bytes = L.off_halo + (size_t)2 * world * 2 * halo_cap * sizeof(unsigned long long);
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
    if (hipMalloc((void**)&d_epoch_red, sizeof(unsigned long long)) != hipSuccess || hipMemset(d_epoch_red, 0, sizeof(unsigned long long)) != hipSuccess) return 1;
    for (int k = 0; k < P2P_MAX_RANKS; ++k) peer[k] = nullptr;
    peer[rank] = mailbox;
    // the first launch from this library loads its code object (hundreds of ms in a fresh process, more when several ranks start together): do it
    // here, before any peer can be waiting for this rank inside an exchange
    k_p2p_warmup<<<1, 64>>>(d_err);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
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
    if (mailbox) (void)hipFree(mailbox); if (d_err) (void)hipFree(d_err); if (d_lists) (void)hipFree(d_lists); if (d_epoch_red) (void)hipFree(d_epoch_red);
    mailbox = nullptr; d_err = nullptr; d_lists = nullptr; d_epoch_red = nullptr; ready = false;
}

static PeerPtrs ptrs_of(const P2PEngine& e) { PeerPtrs p; for (int k = 0; k < P2P_MAX_RANKS; ++k) p.m[k] = e.peer[k]; return p; }

P2PDev P2PEngine::device() const { P2PDev d; d.on = ready ? 1 : 0; d.me = rank; d.L = L; d.err = d_err; d.epoch_red = d_epoch_red; d.spin_limit = spin_limit; d.peers = ptrs_of(*this); d.wg_cap = wg_cap; return d; }

int P2PEngine::allreduce(double* dev, size_t n, hipStream_t st) {
    if ((int)n > L.red_cap) return 1;
    k_p2p_allreduce<<<1, 1024, 0, st>>>(dev, (int)n, device());
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int P2PEngine::set_halo_lists(const HaloPlan& h, hipStream_t st) {      // once per outer iteration
    if (h.n_send > 0 || h.n_recv > 0) for (int k = 0; k < world; ++k) if (h.send_cnt[k] > L.halo_cap || h.recv_cnt[k] > L.halo_cap) return 2;      // 2 = the rim does not fit a mailbox (not an error: the caller re-routes it)
    int host[4 * P2P_MAX_RANKS]; std::memset(host, 0, sizeof(host));
    for (int k = 0; k < world; ++k) { host[k] = h.send_off[k]; host[P2P_MAX_RANKS + k] = h.send_cnt[k]; host[2 * P2P_MAX_RANKS + k] = h.recv_off[k]; host[3 * P2P_MAX_RANKS + k] = h.recv_cnt[k]; }
    if (hipMemcpyAsync(d_lists, host, sizeof(host), hipMemcpyHostToDevice, st) != hipSuccess) return 1;
    return hipStreamSynchronize(st) == hipSuccess ? 0 : 1;
}
int P2PEngine::push_halo(float* vec, const HaloPlan& h, hipStream_t st) {
    ++epoch_halo; if ((unsigned)epoch_halo == 0u) ++epoch_halo;
    k_p2p_halo<<<1, 1024, 0, st>>>(vec, h.chunk, rank, L, ptrs_of(*this), epoch_halo, h.d_send_idx, d_lists, d_lists + P2P_MAX_RANKS, h.d_recv_idx, d_lists + 2 * P2P_MAX_RANKS,
                                   d_lists + 3 * P2P_MAX_RANKS, d_err, spin_limit);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
// Start-up test of the exchanges that run inside the multi-workgroup PCG kernels (p2p_device.hpp: p2p_put_double_all / p2p_sum_all / p2p_put_rim / p2p_get_rim):
// per round, in ONE launch of several workgroups — the first two push B rim items to every peer before waiting for anything, workgroup 2 stores this rank's four
// doubles into all mailboxes, every workgroup reads and checks the rank-ordered sums, the first two then consume and check the peers' rim items.  Rounds use the
// pass numbers 1..8 (real solves start far above), alternating the two buffers like a pass does (dir | step).
__global__ void __launch_bounds__(256) k_p2p_selftest_fused(P2PDev d, int seq, int B, int* bad) {
    __shared__ double smx[4 * P2P_MAX_RANKS];
    const int W = d.L.world, me = d.me;
    for (int which = 0; which < 2; ++which) {
        const unsigned e32 = p2p_pass_epoch(seq, which == 0 ? P2P_X_DIR : P2P_X_STEP); const int par = which;
        if (which == 0 && blockIdx.x < 2)
            for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < B * W; j += 2 * blockDim.x) { const int k = j / B, i = j % B; if (k != me) p2p_put_rim(d, par, e32, k, i, (float)(1000 * me + i + seq), -(float)(7 * k + i)); }
        if (blockIdx.x == 2 && threadIdx.x < 4) p2p_put_double_all(d, par, e32, threadIdx.x, (double)(me + 1) * (threadIdx.x + 1) + 0.25 * seq + which);
        double tot[4]; p2p_sum_all<4>(d, par, e32, tot, smx);
        const double tri = 0.5 * W * (W + 1);
        for (int k = 0; k < 4; ++k) if (tot[k] != tri * (k + 1) + W * (0.25 * seq + which)) atomicExch(bad, 1);
        if (which == 0 && blockIdx.x < 2)
            for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < B * W; j += 2 * blockDim.x) {
                const int k = j / B, i = j % B; if (k == me) continue;
                float a, b; p2p_get_rim(d, par, e32, k, i, a, b);
                if (a != (float)(1000 * k + i + seq) || b != -(float)(7 * me + i)) atomicExch(bad, 2);
            }
    }
}
int P2PEngine::selftest_fused(hipStream_t st) {
    int* d_bad = nullptr;
    if (hipMalloc((void**)&d_bad, sizeof(int)) != hipSuccess || hipMemset(d_bad, 0, sizeof(int)) != hipSuccess) return 1;
    int bad = 0; bool ok = true;
    for (int seq = 1; seq <= 8 && ok; ++seq) {
        k_p2p_selftest_fused<<<24, 256, 0, st>>>(device(), seq, 96, d_bad);
        ok = hipGetLastError() == hipSuccess && check(st) == 0 && hipMemcpy(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess && bad == 0;
    }
    (void)hipFree(d_bad);
    return ok ? 0 : 1;
}

int P2PEngine::check(hipStream_t st) {      // has any spin timed out?  (synchronises the stream)
    int e = 0;
    if (hipMemcpyAsync(&e, d_err, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 1;
    return e;
}

}  // namespace i3d
