// Level-transition kernels of Intrinsic3D::refine and the recolourisation that runs between levels.
//   k_recolor        SDFColorization::add + compute (sdf/colorization.cpp:113-189,318-354), driven by Intrinsic3D::recomputeColors
//                    (refinement/intrinsic3d.cpp:381-409): best-n weighted mean of the keyframe colours seen by every voxel.
//   k_shell_*        SDFAlgorithms::clearVoxelsOutsideThinShell (sdf/algorithms.cpp:368-458).
//   k_upsample       SDFAlgorithms::upsample + interpolate (sdf/algorithms.cpp:118-235): 8 children per voxel, trilinear blend of the
//                    VALID corners with fp32 accumulators (the reference squeezes its double fields through float here).
// Compiled with -ffp-contract=off: colours are truncated to 8 bit and child weights are thresholded, so the float operation order
// of the reference is kept.
#include "kernels.hpp"
#include "level_kernels.hpp"
#include "observe_device.hpp"

namespace i3d {

static __device__ inline unsigned long long pack_key_l(int x, int y, int z) {
    const long long B = 1ll << 20;
    return ((unsigned long long)(x + B) & 0x1fffffull) | (((unsigned long long)(y + B) & 0x1fffffull) << 21) | (((unsigned long long)(z + B) & 0x1fffffull) << 42);
}
static __device__ inline unsigned int mix64_l(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned int)k;
}
static __device__ inline int hash_find_l(const HashTable& t, int x, int y, int z) {
    const unsigned long long key = pack_key_l(x, y, z);
    unsigned int h = mix64_l(key) & t.mask;
    for (;;) {
        const unsigned long long k = t.keys[h];
        if (k == key) return t.vals[h];
        if (k == ~0ull) return -1;
        h = (h + 1) & t.mask;
    }
}

// ---- recolourisation -----------------------------------------------------------------------------------------------------
template <int NOBS>
__global__ void __launch_bounds__(256) k_recolor(GridView g, OptParams p, const FrameConst* __restrict__ frames, int nobs, uchar4* __restrict__ color_out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= g.N) return;
    const int N = g.N;
    // computeSurfaceNormal (operators.cpp:58-77): zero unless the voxel and its +x,+y,+z neighbours are valid
    const int nbx = g.nbr[(size_t)NB_PX * N + s], nby = g.nbr[(size_t)NB_PY * N + s], nbz = g.nbr[(size_t)NB_PZ * N + s];
    if (!(g.weight[s] > 0.0f) || nbx < 0 || nby < 0 || nbz < 0 || !(g.weight[nbx] > 0.0f) || !(g.weight[nby] > 0.0f) || !(g.weight[nbz] > 0.0f)) return;
    const float s0 = g.f_sdf[s];
    float nx = g.f_sdf[nbx] - s0, ny = g.f_sdf[nby] - s0, nz = g.f_sdf[nbz] - s0;
    const float len = sqrtf(nx * nx + (ny * ny + nz * nz));
    if (len != 0.0f) { nx /= len; ny /= len; nz /= len; }
    if (fabsf(nx) <= 1e-5f && fabsf(ny) <= 1e-5f && fabsf(nz) <= 1e-5f) return;
    const float px = (float)g.cx[s] * g.voxel_size - nx * s0, py = (float)g.cy[s] * g.voxel_size - ny * s0, pz = (float)g.cz[s] * g.voxel_size - nz * s0;

    // two candidate lists: the first NOBS observations in frame order (used when there are <= n of them: filter() returns unsorted)
    // and the NOBS heaviest in ascending order (what std::sort + zeroing leaves with non-zero weight)
    float fw[NOBS], bw[NOBS]; uchar4 fc4[NOBS], bc4[NOBS];
#pragma unroll
    for (int i = 0; i < NOBS; ++i) { fw[i] = 0.0f; bw[i] = 0.0f; fc4[i] = make_uchar4(0, 0, 0, 0); bc4[i] = make_uchar4(0, 0, 0, 0); }
    int count = 0;
    for (int f = 0; f < p.K; ++f) {
        const FrameConst& fc = frames[f];
        float u, v;
        const float w = observation_weight(fc, p, px, py, pz, nx, ny, nz, fc.depth, u, v);
        if (!(w > 0.0f)) continue;
        const uchar4 c = make_uchar4(bilinear_u8(fc.bgr, fc.w, fc.h, u, v, 2), bilinear_u8(fc.bgr, fc.w, fc.h, u, v, 1), bilinear_u8(fc.bgr, fc.w, fc.h, u, v, 0), 0);
#pragma unroll
        for (int i = 0; i < NOBS; ++i) if (i == count) { fw[i] = w; fc4[i] = c; }
        ++count;
        if (w > bw[0]) {
            bw[0] = w; bc4[0] = c;
#pragma unroll
            for (int i = 0; i + 1 < NOBS; ++i)
                if (bw[i] > bw[i + 1]) { const float tw = bw[i]; bw[i] = bw[i + 1]; bw[i + 1] = tw; const uchar4 tc = bc4[i]; bc4[i] = bc4[i + 1]; bc4[i + 1] = tc; }
        }
    }
    if (count == 0) return;                                  // voxel keeps its colour (colorization.cpp:171)
    const bool sorted = (nobs > 0 && nobs < count);          // filter(): n == 0 or n >= #obs leaves everything, unsorted
    // computeColor (colorization.cpp:318-354)
    const float scale_color = 1.0f / 255.0f;
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, wsum = 0.0f;
#pragma unroll
    for (int i = 0; i < NOBS; ++i) {
        const float w = sorted ? bw[i] : fw[i]; const uchar4 c = sorted ? bc4[i] : fc4[i];
        if (w == 0.0f || (sorted && i < NOBS - nobs)) continue;   // zeroed / absent entries add exactly 0; only the n heaviest survive filter()
        c0 += (float)c.x * (w * scale_color); c1 += (float)c.y * (w * scale_color); c2 += (float)c.z * (w * scale_color);
        wsum = wsum + w;
    }
    if (wsum > 0.0f) { const float q = 255.0f / wsum; c0 = c0 * q; c1 = c1 * q; c2 = c2 * q; }
    color_out[s] = make_uchar4((unsigned char)c0, (unsigned char)c1, (unsigned char)c2, 0);
}
void launch_recolor(hipStream_t st, GridView g, OptParams p, const FrameConst* frames, int nobs, uchar4* color_out) {
    if (g.N <= 0) return;
    k_recolor<MAX_SLOTS><<<(g.N + 255) / 256, 256, 0, st>>>(g, p, frames, nobs, color_out);
}

// ---- thin shell ------------------------------------------------------------------------------------------------------------
// pass 1: every valid in-shell voxel keeps itself, its 6-ring and its +2x/+2y/+2z neighbours (algorithms.cpp:374-398)
__global__ void __launch_bounds__(256) k_shell_mark(GridView g, double thres, int* __restrict__ keep) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= g.N) return;
    if (!(g.weight[s] > 0.0f) || fabs(g.x_sdf[s]) > thres) return;
    keep[s] = 1;
    const int N = g.N;
    const int sel[9] = {NB_PX, NB_MX, NB_PY, NB_MY, NB_PZ, NB_MZ, NB_P2X, NB_P2Y, NB_P2Z};
#pragma unroll
    for (int i = 0; i < 9; ++i) { const int nb = g.nbr[(size_t)sel[i] * N + s]; if (nb >= 0) keep[nb] = 1; }
}
// pass 2: anything else survives only if a stored voxel of opposite sign lies within its 5^3 neighbourhood (algorithms.cpp:401-452)
__global__ void __launch_bounds__(256) k_shell_crossing(GridView g, HashTable t, int* __restrict__ keep) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= g.N || keep[s]) return;
    const bool neg = g.x_sdf[s] < 0.0;
    const int x = g.cx[s], y = g.cy[s], z = g.cz[s];
    for (int dz = -2; dz <= 2; ++dz) for (int dy = -2; dy <= 2; ++dy) for (int dx = -2; dx <= 2; ++dx) {
        if (!dx && !dy && !dz) continue;
        const int nb = hash_find_l(t, x + dx, y + dy, z + dz);
        if (nb < 0) continue;
        const double v = g.x_sdf[nb];
        if (neg ? (v >= 0.0) : (v < 0.0)) { keep[s] = 1; return; }
    }
}
void launch_shell_mark(hipStream_t st, GridView g, double thres, int* keep) { if (g.N > 0) k_shell_mark<<<(g.N + 255) / 256, 256, 0, st>>>(g, thres, keep); }
void launch_shell_crossing(hipStream_t st, GridView g, HashTable t, int* keep) { if (g.N > 0) k_shell_crossing<<<(g.N + 255) / 256, 256, 0, st>>>(g, t, keep); }

// visit-order staging of the resident grid (optionally filtered): out arrays are indexed by the NEW visit index
__global__ void __launch_bounds__(256) k_export_visit(GridView g, const int* __restrict__ inv_rank /*visit -> device*/, const int* __restrict__ keep_dev,
                                                      const int* __restrict__ scan_visit, int* kxyz, double* sdf, double* sdf_ref, double* alb, float* w, uint8_t* rgb) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= g.N) return;
    const int s = inv_rank[v];
    if (keep_dev && !keep_dev[s]) return;
    const int o = scan_visit ? scan_visit[v] : v;
    kxyz[3 * o] = g.cx[s]; kxyz[3 * o + 1] = g.cy[s]; kxyz[3 * o + 2] = g.cz[s];
    sdf[o] = g.sdf0[s]; sdf_ref[o] = g.x_sdf[s]; alb[o] = g.x_alb[s]; w[o] = g.weight[s];
    const uchar4 c = g.color[s]; rgb[3 * o] = c.x; rgb[3 * o + 1] = c.y; rgb[3 * o + 2] = c.z;
}
__global__ void k_inv_rank(int N, const int* __restrict__ rank, int* __restrict__ inv) { const int s = blockIdx.x * blockDim.x + threadIdx.x; if (s < N) inv[rank[s]] = s; }
__global__ void k_keep_visit(int N, const int* __restrict__ inv, const int* __restrict__ keep_dev, int* __restrict__ keep_visit) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x; if (v < N) keep_visit[v] = keep_dev[inv[v]] ? 1 : 0; }
void launch_inv_rank(hipStream_t st, int N, const int* rank, int* inv) { if (N > 0) k_inv_rank<<<(N + 255) / 256, 256, 0, st>>>(N, rank, inv); }
void launch_keep_visit(hipStream_t st, int N, const int* inv, const int* keep_dev, int* keep_visit) { if (N > 0) k_keep_visit<<<(N + 255) / 256, 256, 0, st>>>(N, inv, keep_dev, keep_visit); }
void launch_export_visit(hipStream_t st, GridView g, const int* inv_rank, const int* keep_dev, const int* scan_visit, int* kxyz, double* sdf, double* sdf_ref,
                         double* alb, float* w, uint8_t* rgb) {
    if (g.N > 0) k_export_visit<<<(g.N + 255) / 256, 256, 0, st>>>(g, inv_rank, keep_dev, scan_visit, kxyz, sdf, sdf_ref, alb, w, rgb);
}

// ---- upsample -----------------------------------------------------------------------------------------------------------------
// one lane per (parent, child): children are emitted in the reference's INSERTION sequence (parent visit order, z-y-x loops);
// the iteration order of the new map is derived from that sequence on the host.
__global__ void __launch_bounds__(256) k_upsample(GridView g, HashTable t, const int* __restrict__ inv_rank, int* kxyz, double* sdf, double* sdf_ref,
                                                  double* alb, float* w_out, uint8_t* rgb) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 8ll * g.N) return;
    const int v = (int)(i >> 3), ch = (int)(i & 7);
    const int cxo = ch & 1, cyo = (ch >> 1) & 1, czo = (ch >> 2) & 1;          // for z: for y: for x  => child index = z*4 + y*2 + x
    const int s = inv_rank[v];
    const int x = g.cx[s], y = g.cy[s], z = g.cz[s];
    // math::interpolationWeights (math.cpp:103-128) at pos = p + 0.5*(cx,cy,cz): base = p, fractional part 0 or 0.5
    const float wx = (float)cxo * 0.5f, wy = (float)cyo * 0.5f, wz = (float)czo * 0.5f;
    const int ox[8] = {0, 1, 0, 0, 1, 0, 1, 1}, oy[8] = {0, 0, 1, 0, 1, 1, 0, 1}, oz[8] = {0, 0, 0, 1, 0, 1, 1, 1};
    const float w8[8] = {(1.0f - wx) * (1.0f - wy) * (1.0f - wz), wx * (1.0f - wy) * (1.0f - wz), (1.0f - wx) * wy * (1.0f - wz), (1.0f - wx) * (1.0f - wy) * wz,
                         wx * wy * (1.0f - wz), (1.0f - wx) * wy * wz, wx * (1.0f - wy) * wz, wx * wy * wz};
    float a_w = 0.0f, a_sdf = 0.0f, a_alb = 0.0f, a_ref = 0.0f, a_c0 = 0.0f, a_c1 = 0.0f, a_c2 = 0.0f, sum_w = 0.0f;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int nb = (k == 0) ? s : hash_find_l(t, x + ox[k], y + oy[k], z + oz[k]);
        if (nb < 0 || !(g.weight[nb] > 0.0f)) continue;
        const float wk = w8[k];
        a_sdf += wk * (float)g.sdf0[nb];
        const uchar4 c = g.color[nb];
        a_c0 += wk * (float)c.x; a_c1 += wk * (float)c.y; a_c2 += wk * (float)c.z;
        a_w += wk * g.weight[nb];
        a_alb += wk * (float)g.x_alb[nb];
        a_ref += wk * (float)g.x_sdf[nb];
        sum_w += wk; ++cnt;
    }
    if (sum_w > 0.0f) { a_sdf /= sum_w; a_c0 /= sum_w; a_c1 /= sum_w; a_c2 /= sum_w; a_w /= sum_w; a_alb /= sum_w; a_ref /= sum_w; }
    if (cnt <= 4) a_w = 0.0f;
    kxyz[3 * i] = 2 * x + cxo; kxyz[3 * i + 1] = 2 * y + cyo; kxyz[3 * i + 2] = 2 * z + czo;
    sdf[i] = (double)a_sdf; sdf_ref[i] = (double)a_ref; alb[i] = (double)a_alb; w_out[i] = fmaxf(a_w, 0.0f);
    rgb[3 * i] = (unsigned char)(int)(a_c0 + 0.5f); rgb[3 * i + 1] = (unsigned char)(int)(a_c1 + 0.5f); rgb[3 * i + 2] = (unsigned char)(int)(a_c2 + 0.5f);
}
void launch_upsample(hipStream_t st, GridView g, HashTable t, const int* inv_rank, int* kxyz, double* sdf, double* sdf_ref, double* alb, float* w, uint8_t* rgb) {
    const long long n = 8ll * g.N;
    if (n > 0) k_upsample<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(g, t, inv_rank, kxyz, sdf, sdf_ref, alb, w, rgb);
}

}  // namespace i3d

namespace i3d {
// new visit order = perm over the insertion sequence (derived on the host from the map's iteration order)
__global__ void __launch_bounds__(256) k_permute_staging(long long n, const int* __restrict__ perm, const int* __restrict__ kin, const double* __restrict__ s0, const double* __restrict__ s1,
                                                         const double* __restrict__ al, const float* __restrict__ w, const uint8_t* __restrict__ rgb,
                                                         int* kout, double* o0, double* o1, double* oal, float* ow, uint8_t* orgb) {
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const long long i = perm[v];
    kout[3 * v] = kin[3 * i]; kout[3 * v + 1] = kin[3 * i + 1]; kout[3 * v + 2] = kin[3 * i + 2];
    o0[v] = s0[i]; o1[v] = s1[i]; oal[v] = al[i]; ow[v] = w[i];
    orgb[3 * v] = rgb[3 * i]; orgb[3 * v + 1] = rgb[3 * i + 1]; orgb[3 * v + 2] = rgb[3 * i + 2];
}
void launch_permute_staging(hipStream_t st, long long n, const int* perm, const int* kin, const double* s0, const double* s1, const double* al, const float* w,
                            const uint8_t* rgb, int* kout, double* o0, double* o1, double* oal, float* ow, uint8_t* orgb) {
    if (n > 0) k_permute_staging<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(n, perm, kin, s0, s1, al, w, rgb, kout, o0, o1, oal, ow, orgb);
}
}  // namespace i3d

// ---- keyframe pyramids (Pyramid::create, rgbd/pyramid.cpp:59-166) -------------------------------------------------------------------
// level 0 luminance: color.convertTo(CV_32FC3, 1/255) then cv::cvtColor(COLOR_BGR2GRAY) on floats  [OpenCV, un-vendored: b*0.114f + g*0.587f +
// r*0.299f summed left to right]; further levels: cv::pyrDown = separable [1 4 6 4 1] with BORDER_REFLECT_101, horizontal pass first, the
// 1/256 applied once at the end; depth levels: mean of the valid taps of each 2x2 block (pyramid.cpp:115-143).
namespace i3d {
__global__ void __launch_bounds__(256) k_lum_from_bgr(int n, const uint8_t* __restrict__ bgr, float* __restrict__ lum) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = (float)(1.0 / 255.0);
    const float b = (float)bgr[3 * i] * s, g = (float)bgr[3 * i + 1] * s, r = (float)bgr[3 * i + 2] * s;
    lum[i] = (b * 0.114f + g * 0.587f) + r * 0.299f;
}
static __device__ inline int reflect101(int i, int n) { if (n == 1) return 0; while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i; return i; }
__global__ void __launch_bounds__(256) k_pyr_down(int w, int h, const float* __restrict__ src, int ow, int oh, float* __restrict__ dst) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= ow || y >= oh) return;
    float row[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const float* line = src + (size_t)reflect101(2 * y - 2 + j, h) * w;
        const float m2 = line[reflect101(2 * x - 2, w)], m1 = line[reflect101(2 * x - 1, w)], c0 = line[reflect101(2 * x, w)],
                    p1 = line[reflect101(2 * x + 1, w)], p2 = line[reflect101(2 * x + 2, w)];
        row[j] = ((c0 * 6.0f + (m1 + p1) * 4.0f) + m2) + p2;
    }
    dst[(size_t)y * ow + x] = (((row[2] * 6.0f + (row[1] + row[3]) * 4.0f) + row[0]) + row[4]) * (1.0f / 256.0f);
}
__global__ void __launch_bounds__(256) k_depth_down(int w, const float* __restrict__ src, int ow, int oh, float* __restrict__ dst) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= ow || y >= oh) return;
    int cnt = 0; float sum = 0.0f;
    const float d0 = src[(size_t)(2 * y) * w + 2 * x], d1 = src[(size_t)(2 * y) * w + 2 * x + 1], d2 = src[(size_t)(2 * y + 1) * w + 2 * x], d3 = src[(size_t)(2 * y + 1) * w + 2 * x + 1];
    if (d0 > 0.0f) { sum += d0; ++cnt; } if (d1 > 0.0f) { sum += d1; ++cnt; } if (d2 > 0.0f) { sum += d2; ++cnt; } if (d3 > 0.0f) { sum += d3; ++cnt; }
    dst[(size_t)y * ow + x] = cnt > 0 ? sum / (float)cnt : 0.0f;
}
void launch_lum_from_bgr(hipStream_t st, int n, const uint8_t* bgr, float* lum) { if (n > 0) k_lum_from_bgr<<<(n + 255) / 256, 256, 0, st>>>(n, bgr, lum); }
void launch_pyr_down(hipStream_t st, int w, int h, const float* src, int ow, int oh, float* dst) { if (ow > 0 && oh > 0) k_pyr_down<<<dim3((ow + 255) / 256, oh), 256, 0, st>>>(w, h, src, ow, oh, dst); }
void launch_depth_down(hipStream_t st, int w, const float* src, int ow, int oh, float* dst) { if (ow > 0 && oh > 0) k_depth_down<<<dim3((ow + 255) / 256, oh), 256, 0, st>>>(w, src, ow, oh, dst); }
}  // namespace i3d

// resizeDepth (rgbd/processing.cpp:129-181): depth image resampled into the colour camera's geometry (pinhole to pinhole at z = 1,
// pixel test on round-half-up coordinates, bilinear lookup with out-of-image taps dropped and the weights renormalised: interpolate<float>,
// processing.cpp:236-283); pixels whose lookup is exactly 0 stay 0.
namespace i3d {
__global__ void __launch_bounds__(256) k_resize_depth(int iw, int ih, const float* __restrict__ din, float in_fx, float in_fy, float in_cx, float in_cy,
                                                      int ow, int oh, float out_fx_inv, float out_fy_inv, float out_cx, float out_cy, float* __restrict__ dout) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= ow || y >= oh) return;
    const float x0n = ((float)x - out_cx) * out_fx_inv, y0n = ((float)y - out_cy) * out_fy_inv;
    const float px = (in_fx * x0n / 1.0f) + in_cx, py = (in_fy * y0n / 1.0f) + in_cy;
    const int pxi = (int)(px + 0.5f), pyi = (int)(py + 0.5f);
    float out = 0.0f;
    if (!(pxi < 0 || pyi < 0 || pxi >= iw || pyi >= ih)) {
        const int x0 = (int)floorf(px), y0 = (int)floorf(py), x1 = x0 + 1, y1 = y0 + 1;
        float x1w = px - (float)x0, y1w = py - (float)y0, x0w = 1.0f - x1w, y0w = 1.0f - y1w;
        if (x0 < 0 || x0 >= iw) x0w = 0.0f; if (x1 < 0 || x1 >= iw) x1w = 0.0f; if (y0 < 0 || y0 >= ih) y0w = 0.0f; if (y1 < 0 || y1 >= ih) y1w = 0.0f;
        const float w00 = x0w * y0w, w10 = x1w * y0w, w01 = x0w * y1w, w11 = x1w * y1w;
        const float sw = ((w00 + w10) + w01) + w11;
        float sum = 0.0f;
        if (w00 > 0.0f) sum += din[(size_t)y0 * iw + x0] * w00;
        if (w01 > 0.0f) sum += din[(size_t)y1 * iw + x0] * w01;
        if (w10 > 0.0f) sum += din[(size_t)y0 * iw + x1] * w10;
        if (w11 > 0.0f) sum += din[(size_t)y1 * iw + x1] * w11;
        if (sw > 0.0f) out = sum / sw;
    }
    dout[(size_t)y * ow + x] = out;
}
void launch_resize_depth(hipStream_t st, int iw, int ih, const float* din, const float in_intr[4], int ow, int oh, const float out_intr[4], float* dout) {
    if (ow > 0 && oh > 0) k_resize_depth<<<dim3((ow + 255) / 256, oh), 256, 0, st>>>(iw, ih, din, in_intr[0], in_intr[1], in_intr[2], in_intr[3], ow, oh,
                                                                                      1.0f / out_intr[0], 1.0f / out_intr[1], out_intr[2], out_intr[3], dout);
}
}  // namespace i3d
