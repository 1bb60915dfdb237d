// Conservative culling in front of the observation pass (K1, observe.hip).  The pass evaluates every (active voxel, keyframe) pair — 2.3 M x 200 on the bench
// workload, 7.9e8 VALU wave-instructions, issue-bound by the exact float divisions its discrete decisions need — although for about half of the pairs the voxel
// lies behind the surface the keyframe sees (occlusion test, colorization.cpp:254-271) or outside its image.  A wave of k_observe handles 64 consecutive
// work-list entries, a compact patch of the surface (brick-Morton order): here every such GROUP gets a bounding sphere of its iso-projected points, and every
// (group, keyframe) pair a conservative test against an 8 x 8-block min / max pyramid of the keyframe's depth image.  A pair is culled only when NO point of the
// sphere can produce a non-zero observation weight:
//   * the whole pixel footprint (with the distortion's displacement bound and two pixels of slack for the rounding of (int)(u + 0.5f)) misses the image, or
//   * no block under the footprint holds a valid depth (the weight needs d > 0), or
//   * occlusion test on: every depth under the footprint is nearer than the sphere by more than the occlusion distance, or farther by more than it.
// Float round-off of the pass's own arithmetic is covered by relative 1e-4 / absolute 1e-6 margins on every bound; a sphere that reaches the camera plane is
// never culled.  What survives is evaluated by k_observe exactly as before: the (voxel, keyframe) row sets stay bit-identical (tests/test_gpu_parity.py).
#include "kernels.hpp"
#include "wave_ops.hpp"

namespace i3d {


// Depth-range pyramid of a keyframe: level 0 = (min valid depth, max depth) of every 8 x 8 pixel block (min = +inf, max = 0 where no pixel is valid), level l + 1 = the
// ranges of 2 x 2 cells of level l, up to a single cell.  A footprint is looked up at the level whose cells are at least as large as it is: at most 2 x 2 cells,
// four INDEPENDENT loads (one memory round trip per keyframe instead of a walk over up to 64 blocks).
CullPyramid cull_pyramid(int w, int h) {
    CullPyramid py{};
    int bw = (w + CULL_BLOCK - 1) / CULL_BLOCK, bh = (h + CULL_BLOCK - 1) / CULL_BLOCK, off = 0, l = 0;
    for (;; ++l) {
        py.bw[l] = bw; py.bh[l] = bh; py.off[l] = off; off += bw * bh;
        if ((bw == 1 && bh == 1) || l + 1 == CULL_MAX_LEVELS) break;
        bw = (bw + 1) / 2; bh = (bh + 1) / 2;
    }
    py.levels = l + 1; py.cells = off;
    return py;
}

__global__ void __launch_bounds__(256) k_depth_blocks(const FrameConst* __restrict__ frames, int K, int w, int h, CullPyramid py, float2* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    const int bw = py.bw[0], bh = py.bh[0];
    if (b >= bw * bh || f >= K) return;
    const float* __restrict__ depth = frames[f].depth;
    const int bx = b % bw, by = b / bw;
    float lo = INFINITY, hi = 0.0f;
    for (int y = by * CULL_BLOCK; y < min(h, (by + 1) * CULL_BLOCK); ++y)
        for (int x = bx * CULL_BLOCK; x < min(w, (bx + 1) * CULL_BLOCK); ++x) {
            const float d = depth[(size_t)y * w + x];
            if (d > 0.0f) { lo = fminf(lo, d); hi = fmaxf(hi, d); }
        }
    out[(size_t)f * py.cells + b] = make_float2(lo, hi);
}
__global__ void __launch_bounds__(256) k_depth_mip(int K, CullPyramid py, int l, float2* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    const int bw = py.bw[l], bh = py.bh[l], pw = py.bw[l - 1], ph = py.bh[l - 1];
    if (b >= bw * bh || f >= K) return;
    const float2* __restrict__ src = out + (size_t)f * py.cells + py.off[l - 1];
    const int bx = b % bw, by = b / bw;
    float lo = INFINITY, hi = 0.0f;
    for (int y = 2 * by; y < min(ph, 2 * by + 2); ++y) for (int x = 2 * bx; x < min(pw, 2 * bx + 2); ++x) { const float2 d = src[y * pw + x]; lo = fminf(lo, d.x); hi = fmaxf(hi, d.y); }
    out[(size_t)f * py.cells + py.off[l] + b] = make_float2(lo, hi);
}
void launch_depth_blocks(hipStream_t st, const FrameConst* frames, int K, int w, int h, float2* out) {
    const CullPyramid py = cull_pyramid(w, h);
    if (K <= 0 || py.cells <= 0) return;
    k_depth_blocks<<<dim3((py.bw[0] * py.bh[0] + 255) / 256, K), 256, 0, st>>>(frames, K, w, h, py, out);
    for (int l = 1; l < py.levels; ++l) k_depth_mip<<<dim3((py.bw[l] * py.bh[l] + 255) / 256, K), 256, 0, st>>>(K, py, l, out);
}

// bounding sphere (centre xyz, radius) of the iso-projected points of the ACTIVE entries of every group of 64 compute-list entries; radius < 0: no active entry
__global__ void __launch_bounds__(256) k_group_bounds(GridView g, RowView r, float4* __restrict__ bounds) {
    const int ci = blockIdx.x * blockDim.x + threadIdx.x;
    const int grp = ci >> 6;
    bool act = false; float px = 0.0f, py = 0.0f, pz = 0.0f;
    if (ci < r.nC) {
        const int a = r.clist ? r.clist[ci] : ci;
        if (r.aflags[a] & F_ACTIVE) {
            const int s = r.alist[a], N = g.N;
            const float s0 = g.f_sdf[s];
            float nx = g.f_sdf[g.nbr[(size_t)NB_PX * N + s]] - s0, ny = g.f_sdf[g.nbr[(size_t)NB_PY * N + s]] - s0, nz = g.f_sdf[g.nbr[(size_t)NB_PZ * N + s]] - s0;
            const float len = sqrtf(nx * nx + (ny * ny + nz * nz));
            if (len != 0.0f) { nx /= len; ny /= len; nz /= len; }
            px = (float)g.cx[s] * g.voxel_size - nx * s0; py = (float)g.cy[s] * g.voxel_size - ny * s0; pz = (float)g.cz[s] * g.voxel_size - nz * s0;
            act = px == px && py == py && pz == pz && !isinf(px) && !isinf(py) && !isinf(pz);
            if (!act) { px = py = pz = NAN; }                         // a non-finite point makes the group unboundable: radius = NaN -> never culled
        }
    }
    auto wmin = [](float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64)); return v; };
    auto wmax = [](float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; };
    const bool any_nan = __ballot(px != px) != 0ull;
    const bool any = __ballot(act) != 0ull;
    const float x0 = wmin(act ? px : INFINITY), x1 = wmax(act ? px : -INFINITY), y0 = wmin(act ? py : INFINITY), y1 = wmax(act ? py : -INFINITY),
                z0 = wmin(act ? pz : INFINITY), z1 = wmax(act ? pz : -INFINITY);
    const float cx = 0.5f * (x0 + x1), cy = 0.5f * (y0 + y1), cz = 0.5f * (z0 + z1);
    const float dx = px - cx, dy = py - cy, dz = pz - cz;
    const float rad = wmax(act ? sqrtf(dx * dx + dy * dy + dz * dz) : 0.0f);
    if ((threadIdx.x & 63) == 0 && (size_t)grp * 64 < (size_t)r.nC)
        bounds[grp] = any_nan ? make_float4(0.0f, 0.0f, 0.0f, NAN) : (any ? make_float4(cx, cy, cz, rad * 1.0001f + 1e-6f) : make_float4(0.0f, 0.0f, 0.0f, -1.0f));
}

// One thread per (group, mask word): lanes are CONSECUTIVE groups — neighbouring surface patches, whose footprints fall into the same cache lines of a keyframe's
// block pyramid — and the keyframe is wave-uniform, so its constants arrive by scalar loads.  (Round 4 first had one thread per (group, keyframe), 32 keyframes
// side by side: every lane then read its own keyframe's constants and its own image's blocks, 0.39 ms of uncoalesced L1 traffic for 0.54 ms saved.)
// Bit f of the group's mask = 1 when no point of the group can be observed by keyframe f.
static __device__ inline bool group_culled(const float4 b, const FrameConst& fc, const OptParams& p, const float2* __restrict__ db, const CullPyramid& py) {
    if (b.w < 0.0f) return true;                                    // no active entry: nothing to observe
    if (!(b.w == b.w)) return false;                                // unboundable group
    const float qx = fc.Rf[0] * b.x + fc.Rf[1] * b.y + fc.Rf[2] * b.z + fc.tf[0];
    const float qy = fc.Rf[3] * b.x + fc.Rf[4] * b.y + fc.Rf[5] * b.z + fc.tf[1];
    const float qz = fc.Rf[6] * b.x + fc.Rf[7] * b.y + fc.Rf[8] * b.z + fc.tf[2];
    const float mag = fabsf(b.x) + fabsf(b.y) + fabsf(b.z) + fabsf(fc.tf[0]) + fabsf(fc.tf[1]) + fabsf(fc.tf[2]) + 1.0f;
    const float rho = b.w + 1e-5f * mag;                            // sphere radius + round-off of both evaluations of R p + t
    const float zmin = qz - rho, zmax = qz + rho;
    if (!(zmin > 1e-3f)) return false;                              // a sphere that reaches the camera plane is never culled
    const float ia = 1.0f / zmin, ib = 1.0f / zmax;
    float xlo = fminf((qx - rho) * ia, (qx - rho) * ib), xhi = fmaxf((qx + rho) * ia, (qx + rho) * ib);
    float ylo = fminf((qy - rho) * ia, (qy - rho) * ib), yhi = fmaxf((qy + rho) * ia, (qy + rho) * ib);
    if (!p.dist_zero) {                                             // displacement bound of the Brown model over the box (camera.cpp:135-147: y uses the distorted x)
        const float X = fmaxf(fabsf(xlo), fabsf(xhi)), Y = fmaxf(fabsf(ylo), fabsf(yhi)), r2 = X * X + Y * Y;
        if (!(r2 < 4.0f)) return false;
        const float radial = fabsf(p.dist_f[0]) * r2 + fabsf(p.dist_f[1]) * r2 * r2 + fabsf(p.dist_f[2]) * r2 * r2 * r2;
        const float ddx = X * radial + 2.0f * fabsf(p.dist_f[3]) * X * Y + fabsf(p.dist_f[4]) * (r2 + 2.0f * X * X);
        const float Xd = X + ddx;
        const float ddy = Y * radial + 2.0f * fabsf(p.dist_f[4]) * Xd * Y + fabsf(p.dist_f[3]) * (r2 + 2.0f * Y * Y);
        xlo -= ddx * 1.0001f; xhi += ddx * 1.0001f; ylo -= ddy * 1.0001f; yhi += ddy * 1.0001f;
    }
    const float fx = p.cam_f[0], fy = p.cam_f[1];
    float ulo = fminf(fx * xlo, fx * xhi) + p.cam_f[2], uhi = fmaxf(fx * xlo, fx * xhi) + p.cam_f[2];
    float vlo = fminf(fy * ylo, fy * yhi) + p.cam_f[3], vhi = fmaxf(fy * ylo, fy * yhi) + p.cam_f[3];
    const float slack_u = 2.0f + 1e-4f * (fabsf(ulo) + fabsf(uhi)), slack_v = 2.0f + 1e-4f * (fabsf(vlo) + fabsf(vhi));      // (int)(u + 0.5f) + round-off
    ulo -= slack_u; uhi += slack_u; vlo -= slack_v; vhi += slack_v;
    if (uhi < 0.0f || vhi < 0.0f || ulo > (float)(p.w - 1) || vlo > (float)(p.h - 1)) return true;      // the footprint misses the image
    if (!(ulo == ulo && uhi == uhi && vlo == vlo && vhi == vhi)) return false;
    const int x0 = max(0, (int)floorf(fmaxf(ulo, -1.0f))), x1 = min(p.w - 1, (int)ceilf(fminf(uhi, (float)p.w))),
              y0 = max(0, (int)floorf(fmaxf(vlo, -1.0f))), y1 = min(p.h - 1, (int)ceilf(fminf(vhi, (float)p.h)));
    // the pyramid level whose cells are larger than the footprint: at most two cells per axis
    const int ext = max(x1 - x0, y1 - y0);
    const int l = min(py.levels - 1, ext < CULL_BLOCK ? 0 : 32 - __clz(ext >> 3)), sh = 3 + l;
    if (((x1 >> sh) - (x0 >> sh)) > 1 || ((y1 >> sh) - (y0 >> sh)) > 1) return false;      // (only when the level was capped)
    const float2* __restrict__ lv = db + py.off[l];
    const int lw = py.bw[l], cx0 = x0 >> sh, cx1 = x1 >> sh, cy0 = y0 >> sh, cy1 = y1 >> sh;
    const float2 d00 = lv[cy0 * lw + cx0], d10 = lv[cy0 * lw + cx1], d01 = lv[cy1 * lw + cx0], d11 = lv[cy1 * lw + cx1];
    const float dlo = fminf(fminf(d00.x, d10.x), fminf(d01.x, d11.x)), dhi = fmaxf(fmaxf(d00.y, d10.y), fmaxf(d01.y, d11.y));
    if (!(dhi > 0.0f)) return true;                                 // no valid depth under the footprint
    if (p.occlusion > 0.0f) {
        const float occ = p.occlusion * 1.0001f + 1e-6f + 1e-6f * (fabsf(zmax) + dhi);
        if (zmin - dhi > occ || dlo - zmax > occ) return true;      // behind what the keyframe sees / in front of all of it
    }
    return false;
}

__global__ void __launch_bounds__(256) k_group_cull(int ngroups, int K, OptParams p, const FrameConst* __restrict__ frames, const float4* __restrict__ bounds,
                                                    const float2* __restrict__ dblocks, CullPyramid py, unsigned char* __restrict__ mask, int ncw) {
    const int grp = blockIdx.x * blockDim.x + threadIdx.x, byte = blockIdx.y;         // 8 keyframes per thread: enough waves to hide the round trips
    if (grp >= ngroups) return;
    const float4 b = bounds[grp];
    unsigned bits = 0;
    const int f1 = min(K, 8 * (byte + 1));
    for (int f = 8 * byte; f < f1; ++f)
        if (group_culled(b, frames[f], p, dblocks + (size_t)f * py.cells, py)) bits |= 1u << (f & 7);
    mask[(size_t)grp * (4 * ncw) + byte] = (unsigned char)bits;                       // little-endian: byte f / 8 of the group's ncw 32-bit words
}

void launch_group_cull(hipStream_t st, GridView g, RowView r, OptParams p, const FrameConst* frames, const float2* dblocks, float4* bounds, unsigned* mask) {
    if (r.nC <= 0) return;
    const int ngroups = (r.nC + 63) / 64, ncw = (p.K + 31) / 32;
    const CullPyramid py = cull_pyramid(p.w, p.h);
    k_group_bounds<<<(ngroups * 64 + 255) / 256, 256, 0, st>>>(g, r, bounds);
    k_group_cull<<<dim3((ngroups + 255) / 256, 4 * ncw), 256, 0, st>>>(ngroups, p.K, p, frames, bounds, dblocks, py, reinterpret_cast<unsigned char*>(mask), ncw);
}

}  // namespace i3d
