// TSDF fusion kernels (the stage in front of the path).  Compiled with -ffp-contract=off: allocation and
// integration take discrete decisions on float values (voxel rounding, pixel rounding, truncation tests), so every expression is
// evaluated in the reference's operation order without FMA contraction.
//   SparseVoxelGrid<Voxel>::alloc / integrate            sparse_voxel_grid.cpp:301-467
//   erodeDiscontinuities / computeVertexMap / computeNormals   rgbd/processing.cpp:49-127,184-232
//   SDFAlgorithms::correctSDF                             sdf/algorithms.cpp:260-331
// The reference's map is replaced by an open-addressing table in HBM (64-bit packed keys, linear probing).  What the reference derives
// from its map — the order in which voxels were first inserted, hence the unordered_map iteration order of the saved volume and of
// correctSDF's in-place sweep — is carried by a per-voxel insertion rank that allocation maintains with atomicMin.
#include "fusion_kernels.hpp"

namespace i3d {
namespace {

constexpr int TPB = 256;

__device__ inline unsigned long long pack_key(int x, int y, int z) {
    return (unsigned long long)(unsigned)(x + FUSION_COORD_OFFSET) | ((unsigned long long)(unsigned)(y + FUSION_COORD_OFFSET) << 21) |
           ((unsigned long long)(unsigned)(z + FUSION_COORD_OFFSET) << 42);
}
__device__ inline void unpack_key(unsigned long long k, int& x, int& y, int& z) {
    x = (int)(k & 0x1FFFFFull) - FUSION_COORD_OFFSET; y = (int)((k >> 21) & 0x1FFFFFull) - FUSION_COORD_OFFSET; z = (int)((k >> 42) & 0x1FFFFFull) - FUSION_COORD_OFFSET;
}
// Home slot: a multiplicative hash of the packed key.  (A brick-local layout — 512 contiguous slots per 8x8x8 brick — was measured and
// rejected: the surface shell fills long runs of such a group, colliding bricks then probe linearly through hundreds of occupied slots,
// and both allocation and correctSDF became ~40x slower.)
__device__ inline unsigned long long slot_of(unsigned long long key, unsigned long long mask) { return ((key * 0x9E3779B97F4A7C15ull) >> 17) & mask; }
__device__ inline long long find_slot(const FusionTable& t, unsigned long long key) {
    unsigned long long s = slot_of(key, t.mask);
    for (unsigned long long probes = 0; probes <= t.mask; ++probes) {      // bounded: a completely full table has no empty slot to stop at
        const unsigned long long k = t.keys[s];
        if (k == key) return (long long)s;
        if (k == FUSION_EMPTY) return -1;
        s = (s + 1) & t.mask;
    }
    return -1;
}
__device__ inline int round_trunc(float v) { return (int)(v + 0.5f); }                                   // mat.h:90
// pose.topLeftCorner<3,3>() * p + pose.topRightCorner<3,1>() (sparse_voxel_grid.cpp:328,425,584): a fixed-size Eigen product, every coefficient
// the halving reduction a0 + (a1 + a2) (Eigen Core/Redux.h; the tests hold this against the reference's own integrate / alloc code)
__device__ inline void xform(const float* T, float px, float py, float pz, float q[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) q[i] = (T[4 * i] * px + (T[4 * i + 1] * py + T[4 * i + 2] * pz)) + T[4 * i + 3];
}
__device__ inline float robust_kernel(float val) { const float div = 1.0f + 2.0f * val; return 1.0f / (div * div * div); }   // math.cpp:43-47
__device__ inline bool within(const int* b, int x, int y, int z) { return !(x < b[0] || x > b[1] || y < b[2] || y > b[3] || z < b[4] || z > b[5]); }

__global__ void k_clear(FusionTable t) {
    const unsigned long long i = (unsigned long long)blockIdx.x * TPB + threadIdx.x;
    if (i > t.mask) return;
    t.keys[i] = FUSION_EMPTY; t.sdf[i] = 0.0f; t.weight[i] = 0.0f; t.color[i] = make_uchar4(0, 0, 0, 0); t.rank[i] = ~0ull; t.crank[i] = ~0ull;
}
__global__ void k_rehash(FusionTable src, FusionTable dst) {
    const unsigned long long i = (unsigned long long)blockIdx.x * TPB + threadIdx.x;
    if (i > src.mask) return;
    const unsigned long long key = src.keys[i];
    if (key == FUSION_EMPTY) return;
    unsigned long long s = slot_of(key, dst.mask);
    for (;;) {
        if (atomicCAS(&dst.keys[s], FUSION_EMPTY, key) == FUSION_EMPTY) break;       // keys are unique in src
        s = (s + 1) & dst.mask;
    }
    dst.sdf[s] = src.sdf[i]; dst.weight[s] = src.weight[i]; dst.color[s] = src.color[i]; dst.rank[s] = src.rank[i]; dst.crank[s] = src.crank[i];
}

__global__ void k_erode(int w, int h, const float* __restrict__ in, int window, float max_diff, float* __restrict__ out) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= w || y >= h) return;
    const float d_ref = in[(size_t)y * w + x];
    bool valid = d_ref != 0.0f;
    if (valid && window > 0)
        for (int v = max(0, y - window); v <= min(y + window, h - 1); ++v)
            for (int u = max(0, x - window); u <= min(x + window, w - 1); ++u) {
                const float d = in[(size_t)v * w + u];
                if (d == 0.0f || fabsf(d - d_ref) > max_diff) valid = false;
            }
    out[(size_t)y * w + x] = valid ? d_ref : 0.0f;
}
__device__ inline void vertex(const FusionCam& c, const float* depth, int x, int y, float fx_inv, float fy_inv, float v[3]) {
    const float d = depth[(size_t)y * c.w + x];
    const float x0 = ((float)x - c.cx) * fx_inv, y0 = ((float)y - c.cy) * fy_inv;
    v[0] = x0 * d; v[1] = y0 * d; v[2] = d;
}
__global__ void k_normals(FusionCam c, const float* __restrict__ depth, float thr, float* __restrict__ normals) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= c.w || y >= c.h) return;
    float n[3] = {0.0f, 0.0f, 0.0f};
    if (x >= 1 && y >= 1 && x < c.w - 1 && y < c.h - 1) {
        const float fx_inv = 1.0f / c.fx, fy_inv = 1.0f / c.fy;
        float v[3], x0[3], x1[3], y0[3], y1[3];
        vertex(c, depth, x, y, fx_inv, fy_inv, v); vertex(c, depth, x - 1, y, fx_inv, fy_inv, x0); vertex(c, depth, x + 1, y, fx_inv, fy_inv, x1);
        vertex(c, depth, x, y - 1, fx_inv, fy_inv, y0); vertex(c, depth, x, y + 1, fx_inv, fy_inv, y1);
        if (v[2] != 0.0f && x0[2] != 0.0f && x1[2] != 0.0f && y0[2] != 0.0f && y1[2] != 0.0f) {
            const float tx[3] = {x1[0] - x0[0], x1[1] - x0[1], x1[2] - x0[2]}, ty[3] = {y1[0] - y0[0], y1[1] - y0[1], y1[2] - y0[2]};
            const float ntx = sqrtf(tx[0] * tx[0] + (tx[1] * tx[1] + tx[2] * tx[2])), nty = sqrtf(ty[0] * ty[0] + (ty[1] * ty[1] + ty[2] * ty[2]));
            if (ntx < thr && nty < thr) {
                n[0] = ty[1] * tx[2] - ty[2] * tx[1]; n[1] = ty[2] * tx[0] - ty[0] * tx[2]; n[2] = ty[0] * tx[1] - ty[1] * tx[0];
                const float sq = n[0] * n[0] + (n[1] * n[1] + n[2] * n[2]);
                if (sq > 0.0f) { const float l = sqrtf(sq); n[0] /= l; n[1] /= l; n[2] /= l; }
            }
        }
    }
    float* o = &normals[((size_t)y * c.w + x) * 3]; o[0] = n[0]; o[1] = n[1]; o[2] = n[2];
}

// Allocation (SparseVoxelGrid::alloc) in two launches per frame.
// The reference walks every depth pixel's ray through the truncation band and inserts the 3x3x3 block around every voxel the ray enters —
// 27 map probes per ray sample, ~3.4e8 per VGA frame, nearly all of them finding the voxel present.  Here
//   1. k_alloc_centres (one lane per pixel) only records the CENTRE voxels: it makes sure the centre exists and keeps, per centre, the rank
//      of its first visit in the reference's sequential order (frame, pixel, ray step) with atomicMin — one probe per ray sample;
//   2. k_alloc_blocks (one lane per table slot) expands the block of every centre that has not been expanded in an earlier frame.
// A cell is first inserted by the earliest visit whose block covers it, and for one centre the earliest visit is its first, so the
// insertion rank of a new cell is  min over the centres c around it of (first visit of c, index of the cell in c's block)  — exactly
// what the atomicMin over the expanding centres computes.  A centre expanded once never needs expanding again (its cells exist):
// crank = 0 marks it, ~0 = never a centre, anything else = pending first-visit rank.  Both launches are idempotent, so a frame whose
// allocation ran out of table space is simply repeated after growth (the blocks launch does nothing when the centres launch overflowed).
// `fresh` counts the cells this lane created; the lanes of a wave add their totals to the global counter with ONE atomic at the end of
// the kernel (a same-address atomic per new voxel — ~1e6 per frame — serialises at ~10 ns each and was 3/4 of the allocation time).
__device__ inline bool insert_cell(const FusionTable& t, unsigned long long key, unsigned long long my_rank, unsigned long long limit, const unsigned long long* count,
                                   unsigned& fresh, int* overflow, unsigned long long* slot_out) {
    unsigned long long sl = slot_of(key, t.mask);
    for (unsigned long long probes = 0; probes <= t.mask; ++probes) {
        unsigned long long k = t.keys[sl];
        if (k == FUSION_EMPTY) {
            if (*count + fresh >= limit) break;                                   // the counter lags by what the waves in flight have not yet added: the
            k = atomicCAS(&t.keys[sl], FUSION_EMPTY, key);                        // probe bound above keeps a table that filled up meanwhile from spinning
            if (k == FUSION_EMPTY) { ++fresh; k = key; }
        }
        if (k == key) { if (my_rank < t.rank[sl]) atomicMin(&t.rank[sl], my_rank); *slot_out = sl; return true; }
        sl = (sl + 1) & t.mask;
    }
    *overflow = 1;
    return false;
}
__device__ inline void add_fresh(unsigned fresh, unsigned long long* count) {     // every lane of the wave must call this
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) fresh += __shfl_down(fresh, off);
    if ((threadIdx.x & 63) == 0 && fresh) atomicAdd(count, (unsigned long long)fresh);
}
__global__ void k_alloc_centres(FusionTable t, FusionFrame f, FusionCam cam, const float* __restrict__ depth, unsigned long long limit,
                                unsigned long long* count, int* overflow) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    unsigned fresh = 0;
    const float d = (x < cam.w && y < cam.h) ? depth[(size_t)y * cam.w + x] : 0.0f;
    if (d != 0.0f) {
        const float pcx = 1.0f * (((float)x - cam.cx) / cam.fx), pcy = 1.0f * (((float)y - cam.cy) / cam.fy), pcz = 1.0f;     // unproject2(x, y, 1)
        const float ray_step = f.voxel_size * 0.25f, inv_vs = 1.0f / f.voxel_size;
        const unsigned long long pixel = (unsigned long long)y * cam.w + x;
        int lx = 0, ly = 0, lz = 0; unsigned step = 0;
        for (float d_off = -f.truncation; d_off <= f.truncation; d_off += ray_step, ++step) {
            const float s = d + d_off;
            float pw[3]; xform(f.c2w, pcx * s, pcy * s, pcz * s, pw);
            const int gx = round_trunc(pw[0] * inv_vs), gy = round_trunc(pw[1] * inv_vs), gz = round_trunc(pw[2] * inv_vs);
            if (gx == lx && gy == ly && gz == lz) continue;
            lx = gx; ly = gy; lz = gz;
            if (!within(f.bounds, gx, gy, gz)) continue;
            if (f.use_clip) {
                const float wx = (float)gx * f.voxel_size, wy = (float)gy * f.voxel_size, wz = (float)gz * f.voxel_size;
                if (wx < f.clip[0] || wx > f.clip[1] || wy < f.clip[2] || wy > f.clip[3] || wz < f.clip[4] || wz > f.clip[5]) continue;
            }
            const unsigned long long visit = (f.frame << 41) | (pixel << 13) | ((unsigned long long)(step & 0xFF) << 5);
            unsigned long long sl;
            if (!insert_cell(t, pack_key(gx, gy, gz), visit | 13ull, limit, count, fresh, overflow, &sl)) break;      // the centre is cell 13 of its own block
            const unsigned long long pending = visit | 31ull;                                                       // never 0, never a cell rank
            const unsigned long long c = t.crank[sl];
            if (c != 0ull && pending < c) atomicMin(&t.crank[sl], pending);
        }
    }
    add_fresh(fresh, count);
}
__global__ void k_alloc_blocks(FusionTable t, unsigned long long limit, unsigned long long* count, int* overflow) {
    const unsigned long long i = (unsigned long long)blockIdx.x * TPB + threadIdx.x;
    unsigned fresh = 0;
    // a centre may only be expanded once its first-visit rank is final: if k_alloc_centres ran out of space, some lanes stopped early and a
    // pending rank may not be the minimum yet — leave everything pending for the repeat after growth
    const unsigned long long c = (i <= t.mask && !*overflow) ? t.crank[i] : 0ull;
    if (c != 0ull && c != ~0ull) {
        int gx, gy, gz; unpack_key(t.keys[i], gx, gy, gz);
        const unsigned long long visit = c & ~31ull;
        int blk = 0; bool ok = true;
        for (int bz = -1; bz <= 1 && ok; ++bz) for (int by = -1; by <= 1 && ok; ++by) for (int bx = -1; bx <= 1 && ok; ++bx, ++blk) {
            unsigned long long sl;
            ok = insert_cell(t, pack_key(gx + bx, gy + by, gz + bz), visit | (unsigned long long)blk, limit, count, fresh, overflow, &sl);
        }
        if (ok) t.crank[i] = 0ull;
    }
    add_fresh(fresh, count);
}

// one lane per table slot: the running weighted mean of one frame (sparse_voxel_grid.cpp:315-395)
__global__ void k_integrate(FusionTable t, FusionFrame f, FusionCam dc, FusionCam cc, const float* __restrict__ depth, const float* __restrict__ normals,
                            const uint8_t* __restrict__ bgr) {
    const unsigned long long i = (unsigned long long)blockIdx.x * TPB + threadIdx.x;
    if (i > t.mask) return;
    const unsigned long long key = t.keys[i];
    if (key == FUSION_EMPTY) return;
    int gx, gy, gz; unpack_key(key, gx, gy, gz);
    if (!within(f.bounds, gx, gy, gz)) return;
    float p[3]; xform(f.w2c, (float)gx * f.voxel_size, (float)gy * f.voxel_size, (float)gz * f.voxel_size, p);
    if (p[2] < 0.0f) return;
    int px = round_trunc((p[0] * dc.fx) / p[2] + dc.cx), py = round_trunc((p[1] * dc.fy) / p[2] + dc.cy);
    if (px < 0 || py < 0 || px >= dc.w || py >= dc.h) return;
    const float d = depth[(size_t)py * dc.w + px];
    if (d <= 0.0f) return;
    const float sdf = d - p[2];
    if (sdf <= -f.truncation) return;
    const float tsdf = sdf >= 0.0f ? fminf(f.truncation, sdf) : fmaxf(-f.truncation, sdf);
    float wu = 1.0f;
    if (f.weight_sample > 0.0f) {
        const float* n = &normals[((size_t)py * dc.w + px) * 3];
        const float sq = p[0] * p[0] + (p[1] * p[1] + p[2] * p[2]);
        float pn[3] = {p[0], p[1], p[2]};
        if (sq > 0.0f) { const float l = sqrtf(sq); pn[0] /= l; pn[1] /= l; pn[2] /= l; }
        float wn = 1.0f - fabsf(pn[0] * n[0] + (pn[1] * n[1] + pn[2] * n[2]));
        wn = fmaxf(fminf(wn, 1.0f), 0.0f);
        wn = fmaxf(f.weight_sample * robust_kernel(wn), 1.0f);
        const float wd = fmaxf(f.weight_sample * robust_kernel(2.0f * fabsf(tsdf) / f.truncation), 1.0f);
        const float dn = (d - f.depth_min) / (f.depth_max - f.depth_min);
        const float wz = fmaxf(f.weight_sample * (1.0f - dn), 1.0f);
        wu = fmaxf(((wn + wd) + wz) / 3.0f, 3.0f);
    }
    const float w_old = t.weight[i], w_new = w_old + wu;
    t.sdf[i] = (t.sdf[i] * w_old + sdf * wu) / w_new;
    px = round_trunc((p[0] * cc.fx) / p[2] + cc.cx); py = round_trunc((p[1] * cc.fy) / p[2] + cc.cy);
    if (px >= 0 && py >= 0 && px < cc.w && py < cc.h) {
        const uint8_t* c = &bgr[((size_t)py * cc.w + px) * 3];
        uchar4 col = t.color[i];
        col.x = (unsigned char)(((float)col.x * w_old + (float)c[2] * wu) / w_new);
        col.y = (unsigned char)(((float)col.y * w_old + (float)c[1] * wu) / w_new);
        col.z = (unsigned char)(((float)col.z * w_old + (float)c[0] * wu) / w_new);
        t.color[i] = col;
    }
    t.weight[i] = w_new;
}

__global__ void k_occupied(FusionTable t, int* flags) {
    const unsigned long long i = (unsigned long long)blockIdx.x * TPB + threadIdx.x;
    if (i > t.mask) return;
    flags[i] = t.keys[i] != FUSION_EMPTY;
}
__global__ void k_gather_rank(FusionTable t, const int* flags, const int* offs, unsigned long long* rank, unsigned int* slot) {
    const unsigned long long i = (unsigned long long)blockIdx.x * TPB + threadIdx.x;
    if (i > t.mask || !flags[i]) return;
    rank[offs[i]] = t.rank[i]; slot[offs[i]] = (unsigned int)i;
}
__global__ void k_keys(FusionTable t, long long m, const unsigned int* slot_sorted, int* kxyz) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= m) return;
    int x, y, z; unpack_key(t.keys[slot_sorted[i]], x, y, z);
    kxyz[3 * i] = x; kxyz[3 * i + 1] = y; kxyz[3 * i + 2] = z;
}
__global__ void k_positions(long long m, const unsigned int* slot_sorted, const int* order, unsigned int* visit_slot, int* pos_of_slot) {
    const long long v = (long long)blockIdx.x * TPB + threadIdx.x;
    if (v >= m) return;
    const unsigned int s = slot_sorted[order[v]];
    visit_slot[v] = s; pos_of_slot[s] = (int)v;
}

// ---- correctSDF (sdf/algorithms.cpp:260-331) ----------------------------------------------------------------------------------
// The sweep is an in-place Gauss-Seidel pass in iteration order: voxel v sees the NEW value of neighbours visited before it and the
// OLD value of the others.  The sequential result is the unique fixed point of  cur[v] = F(v; cur[nb < v], old[nb > v]),  so k_correct is
// relaunched on `cur` until a launch changes nothing (voxel number k is final after at most k launches; in practice about ten).
// Hash probing 26 neighbours per voxel per launch costs 43 ms on 14M voxels (random 64-byte lines); instead the allocated voxels are
// sorted once by (brick, cell) so that spatial neighbours are neighbours in memory, the 26 neighbour indices are resolved once into a
// [26][m] table, and every launch is a coalesced read of that table plus short-range gathers.
__global__ void k_spatial_keys(FusionTable t, long long m, const unsigned int* slots, unsigned long long* skey) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= m) return;
    const unsigned long long k = t.keys[slots[i]];
    const unsigned long long x = k & 0x1FFFFFull, y = (k >> 21) & 0x1FFFFFull, z = (k >> 42) & 0x1FFFFFull;
    skey[i] = ((z >> 3) << 45) | ((y >> 3) << 27) | ((x >> 3) << 9) | ((z & 7) << 6) | ((y & 7) << 3) | (x & 7);
}
__global__ void k_compact_init(FusionTable t, long long m, const unsigned int* slot_c, const int* pos_of_slot, int* compact_of_slot, float* c_sdf, int* c_pos,
                               unsigned char* c_valid, unsigned char* c_touched) {
    const long long c = (long long)blockIdx.x * TPB + threadIdx.x;
    if (c >= m) return;
    const unsigned int s = slot_c[c];
    compact_of_slot[s] = (int)c; c_sdf[c] = t.sdf[s]; c_pos[c] = pos_of_slot[s]; c_valid[c] = t.weight[s] > 0.0f; c_touched[c] = 0;
}
__global__ void k_build_nbr(FusionTable t, long long m, const unsigned int* slot_c, const int* compact_of_slot, const unsigned char* c_valid, int* nbr) {
    const long long c = (long long)blockIdx.x * TPB + threadIdx.x;
    if (c >= m) return;
    int gx, gy, gz; unpack_key(t.keys[slot_c[c]], gx, gy, gz);
    int n = 0;
    for (int k = -1; k <= 1; ++k) for (int j = -1; j <= 1; ++j) for (int i = -1; i <= 1; ++i) {
        if (k == 0 && j == 0 && i == 0) continue;
        int out = -1;
        if (c_valid[c]) {
            const long long nb = find_slot(t, pack_key(gx + i, gy + j, gz + k));
            if (nb >= 0) { const int cn = compact_of_slot[nb]; if (c_valid[cn]) out = cn; }
        }
        nbr[(long long)n * m + c] = out; ++n;
    }
}
__global__ void k_correct(FusionTable t, long long m, float voxel_size, const unsigned int* __restrict__ slot_c, const int* __restrict__ nbr, const int* __restrict__ c_pos,
                          const unsigned char* __restrict__ c_valid, const float* __restrict__ c_sdf, float* c_cur, unsigned char* c_upd, int* changed) {
    const long long c = (long long)blockIdx.x * TPB + threadIdx.x;
    if (c >= m || !c_valid[c]) return;
    const int v = c_pos[c];
    int gx, gy, gz; unpack_key(t.keys[slot_c[c]], gx, gy, gz);
    const float cx = (float)gx * voxel_size, cy = (float)gy * voxel_size, cz = (float)gz * voxel_size;
    const float old_f = c_sdf[c];
    const double sdf = (double)old_f, sgn = sdf >= 0.0 ? 1.0 : -1.0;
    float res = old_f; bool updated = false; int n = 0;
    // the 26 neighbours in THREE batches of unconditional loads (a missing neighbour reads this voxel's own slots): table entries, their visit positions, then the value each one
    // contributes — current sweep for voxels visited earlier, previous sweep otherwise.  One neighbour at a time this was a chain of three dependent loads behind a condition, 78
    // round trips per voxel and sweep (tools/isa_drains.py).  The compare-and-keep-the-last loop below is unchanged.
    int nbv[26]; float val[26];
#pragma unroll
    for (int q = 0; q < 26; ++q) nbv[q] = nbr[(long long)q * m + c];
    {
        int pos[26];
#pragma unroll
        for (int q = 0; q < 26; ++q) pos[q] = c_pos[nbv[q] >= 0 ? nbv[q] : (int)c];
#pragma unroll
        for (int q = 0; q < 26; ++q) { const int e = nbv[q] >= 0 ? nbv[q] : (int)c; val[q] = (pos[q] < v ? (const float*)c_cur : c_sdf)[e]; }
    }
#pragma unroll
    for (int k = -1; k <= 1; ++k)
#pragma unroll
    for (int j = -1; j <= 1; ++j)
#pragma unroll
    for (int i = -1; i <= 1; ++i) {
        if (k == 0 && j == 0 && i == 0) continue;
        const int nb = nbv[n]; const float vnb = val[n]; ++n;
        if (nb < 0) continue;
        const double sdf_nb = (double)vnb, sgn_nb = sdf_nb >= 0.0 ? 1.0 : -1.0;
        const float dx = cx - (float)(gx + i) * voxel_size, dy = cy - (float)(gy + j) * voxel_size, dz = cz - (float)(gz + k) * voxel_size;
        const double dist_nb = sdf_nb + sgn_nb * (double)sqrtf(dx * dx + (dy * dy + dz * dz));
        if (fabs(dist_nb) < fabs(sdf) && sgn == sgn_nb) { res = (float)dist_nb; updated = true; }
    }
    c_upd[c] = updated ? 1 : 0;
    if (__float_as_uint(c_cur[c]) != __float_as_uint(res)) { c_cur[c] = res; *changed = 1; }
}
__global__ void k_commit(long long m, const unsigned char* c_valid, const float* c_cur, const unsigned char* c_upd, float* c_sdf, unsigned char* c_touched, int* has_update) {
    const long long c = (long long)blockIdx.x * TPB + threadIdx.x;
    if (c >= m || !c_valid[c]) return;
    if (c_upd[c]) { c_sdf[c] = c_cur[c]; c_touched[c] = 1; *has_update = 1; }
}
__global__ void k_write_back(FusionTable t, long long m, const unsigned int* slot_c, const float* c_sdf, const unsigned char* c_touched) {
    const long long c = (long long)blockIdx.x * TPB + threadIdx.x;
    if (c >= m || !c_touched[c]) return;
    const unsigned int s = slot_c[c];
    t.sdf[s] = c_sdf[c]; t.weight[s] = 1.0f;
}
__global__ void k_valid(FusionTable t, long long m, const unsigned int* visit_slot, int* flags) {
    const long long v = (long long)blockIdx.x * TPB + threadIdx.x;
    if (v >= m) return;
    flags[v] = t.weight[visit_slot[v]] > 0.0f;
}
__global__ void k_export(FusionTable t, long long m, const unsigned int* visit_slot, const int* flags, const int* offs, int* kxyz, float* sdf, float* weight, uint8_t* rgb) {
    const long long v = (long long)blockIdx.x * TPB + threadIdx.x;
    if (v >= m || !flags[v]) return;
    const unsigned int s = visit_slot[v]; const long long o = offs[v];
    int x, y, z; unpack_key(t.keys[s], x, y, z);
    kxyz[3 * o] = x; kxyz[3 * o + 1] = y; kxyz[3 * o + 2] = z;
    sdf[o] = t.sdf[s]; weight[o] = t.weight[s];
    const uchar4 c = t.color[s]; rgb[3 * o] = c.x; rgb[3 * o + 1] = c.y; rgb[3 * o + 2] = c.z;
}

inline unsigned blocks(unsigned long long n) { return (unsigned)((n + TPB - 1) / TPB); }
inline dim3 image_grid(int w, int h) { return dim3((w + 31) / 32, (h + 7) / 8); }

}  // namespace

void launch_fusion_clear(hipStream_t st, FusionTable t) { hipLaunchKernelGGL(k_clear, blocks(t.mask + 1), TPB, 0, st, t); }
void launch_fusion_rehash(hipStream_t st, FusionTable src, FusionTable dst) { hipLaunchKernelGGL(k_rehash, blocks(src.mask + 1), TPB, 0, st, src, dst); }
void launch_erode(hipStream_t st, int w, int h, const float* in, int window, float max_diff, float* out) {
    hipLaunchKernelGGL(k_erode, image_grid(w, h), TPB, 0, st, w, h, in, window, max_diff, out);
}
void launch_normals(hipStream_t st, FusionCam cam, const float* depth, float thr, float* normals) {
    hipLaunchKernelGGL(k_normals, image_grid(cam.w, cam.h), TPB, 0, st, cam, depth, thr, normals);
}
void launch_fusion_alloc(hipStream_t st, FusionTable t, FusionFrame f, FusionCam cam, const float* depth, unsigned long long limit, unsigned long long* count, int* overflow) {
    hipLaunchKernelGGL(k_alloc_centres, image_grid(cam.w, cam.h), TPB, 0, st, t, f, cam, depth, limit, count, overflow);
    hipLaunchKernelGGL(k_alloc_blocks, blocks(t.mask + 1), TPB, 0, st, t, limit, count, overflow);
}
void launch_fusion_integrate(hipStream_t st, FusionTable t, FusionFrame f, FusionCam dcam, FusionCam ccam, const float* depth, const float* normals, const uint8_t* bgr) {
    hipLaunchKernelGGL(k_integrate, blocks(t.mask + 1), TPB, 0, st, t, f, dcam, ccam, depth, normals, bgr);
}
void launch_fusion_occupied(hipStream_t st, FusionTable t, int* flags) { hipLaunchKernelGGL(k_occupied, blocks(t.mask + 1), TPB, 0, st, t, flags); }
void launch_fusion_gather_rank(hipStream_t st, FusionTable t, const int* flags, const int* offsets, unsigned long long* rank, unsigned int* slot) {
    hipLaunchKernelGGL(k_gather_rank, blocks(t.mask + 1), TPB, 0, st, t, flags, offsets, rank, slot);
}
void launch_fusion_keys(hipStream_t st, FusionTable t, long long m, const unsigned int* slot_sorted, int* kxyz) {
    if (m > 0) hipLaunchKernelGGL(k_keys, blocks(m), TPB, 0, st, t, m, slot_sorted, kxyz);
}
void launch_fusion_positions(hipStream_t st, long long m, const unsigned int* slot_sorted, const int* order, unsigned int* visit_slot, int* pos_of_slot) {
    if (m > 0) hipLaunchKernelGGL(k_positions, blocks(m), TPB, 0, st, m, slot_sorted, order, visit_slot, pos_of_slot);
}
void launch_fusion_spatial_keys(hipStream_t st, FusionTable t, long long m, const unsigned int* slots, unsigned long long* skey) {
    if (m > 0) hipLaunchKernelGGL(k_spatial_keys, blocks(m), TPB, 0, st, t, m, slots, skey);
}
void launch_fusion_compact_init(hipStream_t st, FusionTable t, long long m, const unsigned int* slot_c, const int* pos_of_slot, int* compact_of_slot, float* c_sdf, int* c_pos,
                                unsigned char* c_valid, unsigned char* c_touched) {
    if (m > 0) hipLaunchKernelGGL(k_compact_init, blocks(m), TPB, 0, st, t, m, slot_c, pos_of_slot, compact_of_slot, c_sdf, c_pos, c_valid, c_touched);
}
void launch_fusion_build_nbr(hipStream_t st, FusionTable t, long long m, const unsigned int* slot_c, const int* compact_of_slot, const unsigned char* c_valid, int* nbr) {
    if (m > 0) hipLaunchKernelGGL(k_build_nbr, blocks(m), TPB, 0, st, t, m, slot_c, compact_of_slot, c_valid, nbr);
}
void launch_fusion_correct(hipStream_t st, FusionTable t, long long m, float voxel_size, const unsigned int* slot_c, const int* nbr, const int* c_pos, const unsigned char* c_valid,
                           const float* c_sdf, float* c_cur, unsigned char* c_upd, int* changed) {
    if (m > 0) hipLaunchKernelGGL(k_correct, blocks(m), TPB, 0, st, t, m, voxel_size, slot_c, nbr, c_pos, c_valid, c_sdf, c_cur, c_upd, changed);
}
void launch_fusion_commit(hipStream_t st, long long m, const unsigned char* c_valid, const float* c_cur, const unsigned char* c_upd, float* c_sdf, unsigned char* c_touched, int* has_update) {
    if (m > 0) hipLaunchKernelGGL(k_commit, blocks(m), TPB, 0, st, m, c_valid, c_cur, c_upd, c_sdf, c_touched, has_update);
}
void launch_fusion_write_back(hipStream_t st, FusionTable t, long long m, const unsigned int* slot_c, const float* c_sdf, const unsigned char* c_touched) {
    if (m > 0) hipLaunchKernelGGL(k_write_back, blocks(m), TPB, 0, st, t, m, slot_c, c_sdf, c_touched);
}
void launch_fusion_valid(hipStream_t st, FusionTable t, long long m, const unsigned int* visit_slot, int* flags) {
    if (m > 0) hipLaunchKernelGGL(k_valid, blocks(m), TPB, 0, st, t, m, visit_slot, flags);
}
void launch_fusion_export(hipStream_t st, FusionTable t, long long m, const unsigned int* visit_slot, const int* flags, const int* offsets, int* kxyz, float* sdf, float* weight, uint8_t* rgb) {
    if (m > 0) hipLaunchKernelGGL(k_export, blocks(m), TPB, 0, st, t, m, visit_slot, flags, offsets, kxyz, sdf, weight, rgb);
}

}  // namespace i3d
