// Voxel-grid kernels: brick-Morton sort keys, field permutation, device hash table + neighbour table,
// per-iteration voxel classification and active-list compaction.
//
// Replaces the std::unordered_map<Vec3i,VoxelSBR> accesses of the reference (sparse_voxel_grid.cpp:166-259:
// voxel / exists / valid; ~32 hash finds per Eg row, shading_cost.cpp:65-118) by ONE hash build + ONE
// neighbour-table build per grid; every later kernel reads neighbours through the int32 table.
#include "kernels.hpp"

namespace i3d {

thread_local char g_errbuf[512] = {0};
static thread_local char g_launch_err[256] = {0};
bool set_dynamic_lds(const void* kernel, const char* name, size_t bytes, int K) {
    hipError_t e = bytes <= I3D_LDS_LIMIT ? hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) : hipErrorInvalidValue;
    if (e == hipSuccess) return true;
    (void)hipGetLastError();
    if (!g_launch_err[0]) std::snprintf(g_launch_err, sizeof(g_launch_err), "%s needs %zu bytes of LDS per workgroup with K = %d keyframes (limit %zu): %s", name, bytes, K, I3D_LDS_LIMIT, hipGetErrorString(e));
    return false;
}
bool take_launch_error(char* msg, size_t n) {
    if (!g_launch_err[0]) return false;
    std::snprintf(msg, n, "%s", g_launch_err); g_launch_err[0] = 0;
    return true;
}

static __device__ __host__ inline unsigned long long pack_key(int x, int y, int z) {
    const unsigned long long B = 1ull << 20;
    return ((unsigned long long)(x + (long long)B) & 0x1fffffull) | (((unsigned long long)(y + (long long)B) & 0x1fffffull) << 21) |
           (((unsigned long long)(z + (long long)B) & 0x1fffffull) << 42);
}
static __device__ inline unsigned int mix64(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned int)k;
}
static __device__ inline unsigned long long spread3(unsigned long long v) {   // 17 bits -> every 3rd bit
    v &= 0x1ffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

// sort key = Morton code of the 8^3 brick, then x-fastest position inside the brick: neighbouring voxels of the
// forward stencil land in the same or an adjacent 2 KB run of every SoA plane (L2-resident gathers).
__global__ void k_sort_keys(int N, const int* __restrict__ kxyz, unsigned long long* __restrict__ keys, int* __restrict__ iota) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int x = kxyz[3 * i] + (1 << 19), y = kxyz[3 * i + 1] + (1 << 19), z = kxyz[3 * i + 2] + (1 << 19);
    const unsigned long long m = spread3((unsigned)(x >> 3)) | (spread3((unsigned)(y >> 3)) << 1) | (spread3((unsigned)(z >> 3)) << 2);
    keys[i] = (m << 9) | (unsigned long long)(((z & 7) << 6) | ((y & 7) << 3) | (x & 7));
    iota[i] = i;
}
void launch_sort_keys(hipStream_t st, int N, const int* kxyz, unsigned long long* keys, int* iota) {
    if (N > 0) k_sort_keys<<<(N + 255) / 256, 256, 0, st>>>(N, kxyz, keys, iota);
}

__global__ void k_permute(int N, const int* __restrict__ perm, const int* __restrict__ kxyz, const double* __restrict__ sdf,
                          const double* __restrict__ sdf_ref, const double* __restrict__ alb, const float* __restrict__ w,
                          const uint8_t* __restrict__ rgb, int* cx, int* cy, int* cz, int* rank, double* sdf0, double* x_sdf,
                          double* x_alb, float* f_sdf, float* f_alb, float* weight, uchar4* color) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const int v = perm[s];
    cx[s] = kxyz[3 * v]; cy[s] = kxyz[3 * v + 1]; cz[s] = kxyz[3 * v + 2]; rank[s] = v;
    sdf0[s] = sdf[v]; x_sdf[s] = sdf_ref[v]; x_alb[s] = alb[v];
    f_sdf[s] = (float)sdf_ref[v]; f_alb[s] = (float)alb[v]; weight[s] = w[v];
    color[s] = make_uchar4(rgb[3 * v], rgb[3 * v + 1], rgb[3 * v + 2], 0);
}
void launch_permute_grid(hipStream_t st, int N, const int* perm, const int* kxyz, const double* sdf, const double* sdf_ref,
                         const double* alb, const float* w, const uint8_t* rgb, int* cx, int* cy, int* cz, int* rank, double* sdf0,
                         double* x_sdf, double* x_alb, float* f_sdf, float* f_alb, float* weight, uchar4* color) {
    if (N > 0) k_permute<<<(N + 255) / 256, 256, 0, st>>>(N, perm, kxyz, sdf, sdf_ref, alb, w, rgb, cx, cy, cz, rank, sdf0, x_sdf, x_alb, f_sdf, f_alb, weight, color);
}

__global__ void k_hash_build(int N, const int* __restrict__ cx, const int* __restrict__ cy, const int* __restrict__ cz, HashTable t) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const unsigned long long key = pack_key(cx[s], cy[s], cz[s]);
    unsigned int h = mix64(key) & t.mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&t.keys[h], ~0ull, key);
        if (prev == ~0ull || prev == key) { t.vals[h] = s; return; }
        h = (h + 1) & t.mask;
    }
}
void launch_hash_build(hipStream_t st, int N, const int* cx, const int* cy, const int* cz, HashTable t) {
    if (N > 0) k_hash_build<<<(N + 255) / 256, 256, 0, st>>>(N, cx, cy, cz, t);
}

static __device__ inline int hash_find(const HashTable& t, int x, int y, int z) {
    const unsigned long long key = pack_key(x, y, z);
    unsigned int h = mix64(key) & t.mask;
    for (;;) {
        const unsigned long long k = t.keys[h];
        if (k == key) return t.vals[h];
        if (k == ~0ull) return -1;
        h = (h + 1) & t.mask;
    }
}

__global__ void k_nbr_build(int N, const int* __restrict__ cx, const int* __restrict__ cy, const int* __restrict__ cz, HashTable t, int* __restrict__ nbr) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const int x = cx[s], y = cy[s], z = cz[s];
#pragma unroll
    for (int i = 0; i < NUM_NBR; ++i) {
        int dx, dy, dz; nbr_offset(i, dx, dy, dz);
        nbr[(size_t)i * N + s] = hash_find(t, x + dx, y + dy, z + dz);
    }
}
void launch_nbr_build(hipStream_t st, int N, const int* cx, const int* cy, const int* cz, HashTable t, int* nbr) {
    if (N > 0) k_nbr_build<<<(N + 255) / 256, 256, 0, st>>>(N, cx, cy, cz, t, nbr);
}

// optimizer.cpp:183-193 (valid, shell, normal != 0), :312-361 (fixed sets), operators.cpp:58-77 (float normal)
__global__ void k_classify(GridView g, OptParams p, int* __restrict__ active_flag) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= g.N) return;
    const int N = g.N;
    const bool valid = g.weight[s] > 0.0f;
    const bool inshell = !(fabs(g.x_sdf[s]) > p.thres_shell);
    int nb[6]; bool nbv[6];
#pragma unroll
    for (int d = 0; d < 6; ++d) { nb[d] = g.nbr[(size_t)d * N + s]; nbv[d] = nb[d] >= 0 && g.weight[nb[d]] > 0.0f; }
    const bool ring = nbv[0] && nbv[1] && nbv[2] && nbv[3] && nbv[4] && nbv[5];
    bool normal_ok = false;
    if (valid && nbv[NB_PX] && nbv[NB_PY] && nbv[NB_PZ]) {
        const float s0 = g.f_sdf[s];
        float nx = g.f_sdf[nb[NB_PX]] - s0, ny = g.f_sdf[nb[NB_PY]] - s0, nz = g.f_sdf[nb[NB_PZ]] - s0;
        const float len = sqrtf(nx * nx + (ny * ny + nz * nz));
        if (len != 0.0f) { nx /= len; ny /= len; nz /= len; }
        normal_ok = !(fabsf(nx) <= 1e-5f && fabsf(ny) <= 1e-5f && fabsf(nz) <= 1e-5f);
    }
    const bool active = valid && inshell && normal_ok;
    const bool free_vox = valid && inshell && ring;             // Optimizer::fixVoxelParams (optimizer.cpp:312-361)
    const bool free_sdf = free_vox && !p.fix_sdf;
    const bool free_alb = free_vox && !(p.lambda_a < 0.0);
    g.flags[s] = (valid ? F_VALID : 0) | (active ? F_ACTIVE : 0) | (ring ? F_RING : 0) | (free_sdf ? F_FREE_SDF : 0) | (free_alb ? F_FREE_ALB : 0);
    active_flag[s] = (active || free_sdf || free_alb) ? 1 : 0;      // work list = voxels that own rows or unknowns
}
void launch_classify(hipStream_t st, GridView g, OptParams p, int* active_flag) {
    if (g.N > 0) k_classify<<<(g.N + 255) / 256, 256, 0, st>>>(g, p, active_flag);
}

__global__ void k_compact(int N, const int* __restrict__ flag, const int* __restrict__ scan, const uint8_t* __restrict__ flags,
                          int* __restrict__ aidx, int* __restrict__ alist, uint8_t* __restrict__ aflags) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    if (flag[s]) { const int a = scan[s]; aidx[s] = a; alist[a] = s; aflags[a] = flags[s]; } else aidx[s] = -1;
}
void launch_compact(hipStream_t st, int N, const int* flag, const int* scan, const uint8_t* flags, int* aidx, int* alist, uint8_t* aflags) {
    if (N > 0) k_compact<<<(N + 255) / 256, 256, 0, st>>>(N, flag, scan, flags, aidx, alist, aflags);
}

// Stable partition inside every block of 512 consecutive work-list entries (one workgroup per block): first the entries that can own Eg rows — active, with the
// whole forward stencil of the shading term stored (shading_cost.cpp:65-70; what k_build calls `eligible`) — then the rest (free-only entries, active voxels on
// the rim of the stored band).  The brick-Morton order inside both halves is kept; tile and slice membership (multiples of 512 entries) does not change.  Where a
// band of the stored shell cannot own rows (SURVEY.md 8(d)'s 4-voxel shell: 36 % of the active voxels) the row-less entries now fill whole waves instead of
// being sprinkled over all of them.
__global__ void __launch_bounds__(512) k_partition_blocks(GridView g, int A, int* __restrict__ alist, uint8_t* __restrict__ aflags, int* __restrict__ aidx) {
    __shared__ int wcount[8];
    const int base = blockIdx.x * 512, a = base + (int)threadIdx.x;
    const bool in = a < A;
    const int s = in ? alist[a] : -1;
    const uint8_t fl = in ? aflags[a] : 0;
    bool rows = in && (fl & F_ACTIVE);
    if (rows) {
#pragma unroll
        for (int c = 1; c < 10; ++c) rows &= g.nbr[(size_t)slot_fwd_nbr(c) * g.N + s] >= 0;
    }
    const unsigned long long m = __ballot(rows);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wcount[w] = __popcll(m);
    __syncthreads();                                                 // (also: every entry of the block is read before any is rewritten)
    int before = 0, total = 0;
    for (int i = 0; i < 8; ++i) { const int cnt = wcount[i]; if (i < w) before += cnt; total += cnt; }
    const int rank_rows = before + __popcll(m & ((1ull << lane) - 1ull));             // entries with rows in front of this one
    const int pos = rows ? rank_rows : total + ((int)threadIdx.x - rank_rows);         // stable on both sides
    if (in) { const int na = base + pos; alist[na] = s; aflags[na] = fl; aidx[s] = na; }
}
void launch_partition_blocks(hipStream_t st, GridView g, int A, int* alist, uint8_t* aflags, int* aidx) {
    if (A > 0) k_partition_blocks<<<(A + 511) / 512, 512, 0, st>>>(g, A, alist, aflags, aidx);
}
__global__ void k_group_rows(int A, const uint8_t* __restrict__ nrows, int* __restrict__ gmax) {
    const int gq = blockIdx.x * blockDim.x + threadIdx.x;
    if (64 * gq >= A) return;
    int m = 0;
    for (int i = 0; i < 64 && 64 * gq + i < A; ++i) m = max(m, (int)nrows[64 * (size_t)gq + i]);
    gmax[gq] = m;
}
void launch_group_rows(hipStream_t st, int A, const uint8_t* nrows, int* gmax) { if (A > 0) k_group_rows<<<((A + 63) / 64 + 255) / 256, 256, 0, st>>>(A, nrows, gmax); }

// neighbour table of the work list in LIST space: the solver's vectors live there (2A + 6K + 9 entries instead of 2N + ...)
__global__ void k_anbr(int N, int A, int Acap, const int* __restrict__ alist, const int* __restrict__ nbr, const int* __restrict__ aidx, int* __restrict__ anbr) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= A) return;
    const int s = alist[a];
#pragma unroll
    for (int i = 0; i < NUM_NBR; ++i) { const int nb = nbr[(size_t)i * N + s]; anbr[(size_t)i * Acap + a] = nb >= 0 ? aidx[nb] : -1; }
}
void launch_anbr(hipStream_t st, int N, int A, int Acap, const int* alist, const int* nbr, const int* aidx, int* anbr) {
    if (A > 0) k_anbr<<<(A + 255) / 256, 256, 0, st>>>(N, A, Acap, alist, nbr, aidx, anbr);
}

__global__ void k_scatter_sh(int N, const int* __restrict__ rank, const double* __restrict__ shv, float* __restrict__ sh) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const double* src = shv + (size_t)rank[s] * 9;
#pragma unroll
    for (int j = 0; j < 9; ++j) sh[(size_t)j * N + s] = (float)src[j];
}
void launch_scatter_sh(hipStream_t st, int N, const int* rank, const double* shv, float* sh) {
    if (N > 0) k_scatter_sh<<<(N + 255) / 256, 256, 0, st>>>(N, rank, shv, sh);
}

__global__ void k_gather_visit(int N, const int* __restrict__ rank, const double* __restrict__ xs, const double* __restrict__ xa, double* os, double* oa) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const int v = rank[s];
    if (os) os[v] = xs[s];
    if (oa) oa[v] = xa[s];
}
void launch_gather_visit(hipStream_t st, int N, const int* rank, const double* xs, const double* xa, double* os, double* oa) {
    if (N > 0) k_gather_visit<<<(N + 255) / 256, 256, 0, st>>>(N, rank, xs, xa, os, oa);
}

__global__ void k_update_fields(int N, const int* __restrict__ rank, const double* sr, const double* al, const uint8_t* rgb,
                                double* x_sdf, double* x_alb, float* f_sdf, float* f_alb, uchar4* color) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const int v = rank[s];
    if (sr) { x_sdf[s] = sr[v]; f_sdf[s] = (float)sr[v]; }
    if (al) { x_alb[s] = al[v]; f_alb[s] = (float)al[v]; }
    if (rgb) color[s] = make_uchar4(rgb[3 * v], rgb[3 * v + 1], rgb[3 * v + 2], 0);
}
void launch_update_fields(hipStream_t st, int N, const int* rank, const double* sr, const double* al, const uint8_t* rgb,
                          double* x_sdf, double* x_alb, float* f_sdf, float* f_alb, uchar4* color) {
    if (N > 0) k_update_fields<<<(N + 255) / 256, 256, 0, st>>>(N, rank, sr, al, rgb, x_sdf, x_alb, f_sdf, f_alb, color);
}

}  // namespace i3d
