// Sharding plan of one outer iteration (one process per GPU; common.hpp "sharding"): which rim values of the operator input travel
// between which ranks, and which foreign tiles of the operator pass hold ghost entries.  Every rank derives the whole plan from the
// replicated work list — no communication, and both ends of a pair obtain the same lists by construction.
#include <cstring>
#include <cstdlib>
#include "kernels.hpp"
#include <rocprim/rocprim.hpp>

namespace i3d {

// need[e] bit k: rank k computes rows that READ the unknowns of entry e (the entry itself or one of the 12 ring / forward-stencil
// columns of an active entry on its compute list) and does not own e.
__global__ void __launch_bounds__(256) k_need_mask(RowView r, int slice, unsigned long long* __restrict__ need) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= r.A || !(r.aflags[a] & F_ACTIVE)) return;              // entries without rows read nothing
    const int o = a / slice;
    int col[12]; bool same;
    const unsigned long long ranks = shard_entry_ranks(a, slice, r.anbr, r.Acap, col, same);
    if (same) return;                                               // interior entry: rows, columns and owner on one rank
    for (unsigned long long m = ranks; m; m &= m - 1) {
        const int k = __ffsll((long long)m) - 1;                    // rank k has entry a on its compute list (shard_needs_entry)
        if (o != k) atomicOr(&need[a], 1ull << k);
#pragma unroll
        for (int i = 0; i < 12; ++i) if (col[i] >= 0 && col[i] / slice != k) atomicOr(&need[col[i]], 1ull << k);
    }
}

// (direction, peer, entry) items of rank `me`: direction 0 = send (entry owned by me, needed by peer), 1 = receive
__global__ void __launch_bounds__(256) k_halo_items(int A, int slice, int me, const unsigned long long* __restrict__ need, unsigned long long* __restrict__ items,
                                                    int* __restrict__ count, int cap) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= A) return;
    const unsigned long long m = need[e];
    if (!m) return;
    const int o = e / slice;
    if (o == me) {
        for (unsigned long long b = m & ~(1ull << me); b; b &= b - 1) {
            const unsigned long long k = (unsigned long long)(__ffsll((long long)b) - 1);
            const int pos = atomicAdd(count, 1); if (pos < cap) items[pos] = (k << 32) | (unsigned)e;
        }
    } else if ((m >> me) & 1ull) {
        const int pos = atomicAdd(count, 1); if (pos < cap) items[pos] = (1ull << 40) | ((unsigned long long)o << 32) | (unsigned)e;
    }
}

__global__ void __launch_bounds__(256) k_tile_flags(int A, int T, const int* __restrict__ cflag, int* __restrict__ tileflag) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < A && cflag[a]) tileflag[a / T] = 1;
}

void launch_need_mask(hipStream_t st, RowView r, int slice, unsigned long long* need) { if (r.A > 0) k_need_mask<<<(r.A + 255) / 256, 256, 0, st>>>(r, slice, need); }
void launch_halo_items(hipStream_t st, int A, int slice, int me, const unsigned long long* need, unsigned long long* items, int* count, int cap) {
    if (A > 0) k_halo_items<<<(A + 255) / 256, 256, 0, st>>>(A, slice, me, need, items, count, cap);
}
void launch_tile_flags(hipStream_t st, int A, int T, const int* cflag, int* tileflag) { if (A > 0) k_tile_flags<<<(A + 255) / 256, 256, 0, st>>>(A, T, cflag, tileflag); }
size_t halo_sort_temp_bytes(int cap) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (size_t)cap, 0, 48, (hipStream_t)0);
    return bytes;
}
hipError_t launch_halo_sort(hipStream_t st, void* temp, size_t temp_bytes, const unsigned long long* in, unsigned long long* out, int n) {
    if (n <= 0) return hipSuccess;
    return rocprim::radix_sort_keys(temp, temp_bytes, in, out, (size_t)n, 0, 48, st);
}

}  // namespace i3d
