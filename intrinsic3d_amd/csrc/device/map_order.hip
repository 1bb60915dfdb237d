// Iteration order of the reference's voxel map, computed on the device.
//
// host/map_order.hpp replays libstdc++'s list operations one insertion at a time (random access into a bucket array per key: ~1 s for the
// 15 M voxels of a fused volume).  The same order has a closed form per REHASH EPOCH (the span of insertions between two rehashes, bucket
// count nb): the list is a concatenation of bucket groups, a node always enters at the FRONT of its group, a new group always at the front
// of the list — for the nodes relinked by the rehash (processed in list order) and for the insertions after it alike.  With
//     stamp(v)  = position of v in the list before the rehash (old nodes) | its insertion index (new nodes: larger than every position)
//     gstamp(b) = min stamp over the members of bucket b (the arrival that created the group)
// the list after the epoch is the elements sorted by (gstamp descending, stamp descending).  An epoch is therefore: one modulo + atomicMin
// per element, one radix sort of 2 x ceil(log2 m) key bits, one scatter; epochs double in size, the whole replay sorts ~2 n elements.
// The epoch schedule (when libstdc++'s _Prime_rehash_policy grows, and to which prime) comes from the library itself (map_epochs, host).
#include "map_order_device.hpp"
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace i3d {

__global__ void __launch_bounds__(256) k_mo_codes(size_t n, const int* __restrict__ keys, unsigned long long* __restrict__ code) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // mat.h:117-124: int -> size_t sign-extends before the multiply
    code[i] = ((unsigned long long)(long long)keys[3 * i] * 73856093ull) ^ ((unsigned long long)(long long)keys[3 * i + 1] * 19349669ull) ^ ((unsigned long long)(long long)keys[3 * i + 2] * 83492791ull);
}
__global__ void __launch_bounds__(256) k_mo_group(size_t m, size_t m_prev, unsigned long long nb, const unsigned long long* __restrict__ code, const int* __restrict__ pos,
                                                  unsigned* __restrict__ bkt, int* __restrict__ gmin) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const unsigned b = (unsigned)(code[i] % nb);
    bkt[i] = b;
    atomicMin(&gmin[b], i < m_prev ? pos[i] : (int)i);
}
__global__ void __launch_bounds__(256) k_mo_keys(size_t m, size_t m_prev, int bits, const int* __restrict__ pos, const unsigned* __restrict__ bkt, const int* __restrict__ gmin,
                                                 unsigned long long* __restrict__ key, int* __restrict__ val) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const unsigned long long stamp = i < m_prev ? (unsigned long long)pos[i] : (unsigned long long)i, g = (unsigned long long)gmin[bkt[i]];
    key[i] = (((unsigned long long)m - g) << bits) | ((unsigned long long)m - stamp);      // ascending key = descending (gstamp, stamp)
    val[i] = (int)i;
}
__global__ void __launch_bounds__(256) k_mo_scatter(size_t m, const int* __restrict__ order, int* __restrict__ pos) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < m) pos[order[r]] = (int)r;
}

hipError_t map_order_device(hipStream_t st, const int* d_keys, size_t n, const MapEpoch* epochs, int n_epochs, int* d_order) {
    if (n == 0) return hipSuccess;
    if (n >= (1ull << 31)) return hipErrorInvalidValue;
    size_t nb_max = 0; for (int e = 0; e < n_epochs; ++e) nb_max = epochs[e].nb > nb_max ? epochs[e].nb : nb_max;
    unsigned long long *code = nullptr, *key0 = nullptr, *key1 = nullptr; int *pos = nullptr, *gmin = nullptr, *val0 = nullptr, *val1 = nullptr; unsigned* bkt = nullptr; void* tmp = nullptr;
    size_t tmp_bytes = 0;
    hipError_t err = rocprim::radix_sort_pairs(nullptr, tmp_bytes, key0, key1, val0, val1, n, 0, 64, st);
    auto fail = [&](hipError_t e) { for (void* p : {(void*)code, (void*)key0, (void*)key1, (void*)pos, (void*)gmin, (void*)val0, (void*)val1, (void*)bkt, tmp}) if (p) (void)hipFree(p); return e; };
    if (err != hipSuccess) return err;
#define MO_TRY(expr) do { const hipError_t _e = (expr); if (_e != hipSuccess) return fail(_e); } while (0)
    MO_TRY(hipMalloc((void**)&code, sizeof(unsigned long long) * n)); MO_TRY(hipMalloc((void**)&key0, sizeof(unsigned long long) * n)); MO_TRY(hipMalloc((void**)&key1, sizeof(unsigned long long) * n));
    MO_TRY(hipMalloc((void**)&pos, sizeof(int) * n)); MO_TRY(hipMalloc((void**)&gmin, sizeof(int) * nb_max)); MO_TRY(hipMalloc((void**)&val0, sizeof(int) * n)); MO_TRY(hipMalloc((void**)&val1, sizeof(int) * n));
    MO_TRY(hipMalloc((void**)&bkt, sizeof(unsigned) * n)); MO_TRY(hipMalloc(&tmp, tmp_bytes));
    const auto grid = [](size_t m) { return dim3((unsigned)((m + 255) / 256)); };
    k_mo_codes<<<grid(n), 256, 0, st>>>(n, d_keys, code);
    size_t m_prev = 0;
    for (int e = 0; e < n_epochs; ++e) {
        const size_t m = epochs[e].m_end < n ? epochs[e].m_end : n;
        if (m == 0 || m == m_prev) { m_prev = m; continue; }
        int bits = 1; while ((1ull << bits) <= m) ++bits;                         // m - stamp and m - gstamp are in [1, m]
        MO_TRY(hipMemsetAsync(gmin, 0x7f, sizeof(int) * epochs[e].nb, st));
        k_mo_group<<<grid(m), 256, 0, st>>>(m, m_prev, (unsigned long long)epochs[e].nb, code, pos, bkt, gmin);
        k_mo_keys<<<grid(m), 256, 0, st>>>(m, m_prev, bits, pos, bkt, gmin, key0, val0);
        MO_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, key0, key1, val0, val1, m, 0, 2 * bits, st));
        if (e + 1 < n_epochs) k_mo_scatter<<<grid(m), 256, 0, st>>>(m, val1, pos);
        m_prev = m;
    }
    MO_TRY(hipMemcpyAsync(d_order, val1, sizeof(int) * n, hipMemcpyDeviceToDevice, st));
    MO_TRY(hipStreamSynchronize(st));
#undef MO_TRY
    return fail(hipSuccess);
}

}  // namespace i3d
