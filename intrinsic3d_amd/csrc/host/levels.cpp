// Host side of the level transitions of Intrinsic3D::refine (refinement/intrinsic3d.cpp:206-409) around the device kernels of
// level_kernels.hip: recolourisation, thin-shell sparsification, x2 upsampling, and the refine schedule itself.
//
// Visit order (SURVEY.md hazard H1).  The reference keeps its voxels in std::unordered_map<Vec3i,...> (hash mat.h:117-124,
// reserve(64), max_load_factor 0.6: sparse_voxel_grid.cpp:52-53) and several results depend on that container's ITERATION order.
// Erasing keeps the relative order, so sparsification is a stable filter; upsampling and the initial Voxel -> VoxelSBR conversion
// build NEW maps, whose iteration order is a property of libstdc++'s container given the insertion sequence.  That order is
// obtained by replaying the container's list operations on the same key sequence (map_order.hpp; keys only, no voxel payload).
#include "context.hpp"
#include "../device/level_kernels.hpp"
#include <rocprim/rocprim.hpp>
#include "map_order.hpp"

namespace i3d {

namespace {
OptParams color_params(const i3d_context* c, float occlusion) {
    OptParams p; std::memset(&p, 0, sizeof(p));
    p.K = c->K; p.level = 0; p.pyr_scale = 1.0; p.occlusion = occlusion;
    bool dz = true;
    for (int i = 0; i < 4; ++i) { p.intr[i] = c->intr[i]; p.cam_f[i] = (float)c->intr[i]; }          // Camera::setIntrinsics(Vec4) casts to float (camera.cpp:95-98)
    for (int i = 0; i < 5; ++i) { p.dist[i] = c->dist[i]; p.dist_f[i] = (float)c->dist[i]; if (std::fabs(p.dist_f[i]) > 1e-5f) dz = false; }
    p.dist_zero = dz ? 1 : 0; p.w = c->fw[0]; p.h = c->fh[0];
    return p;
}
}  // namespace

// Intrinsic3D::recomputeColors (intrinsic3d.cpp:381-409): SDFColorization::add for every keyframe at pyramid level 0, then compute()
int recompute_colors(i3d_context* c, float occlusion_distance, int num_observations) {
    if (!c->have_grid || !c->have_frames || !c->have_camera) return ctx_fail(c, I3D_ERR_STATE, "recompute_colors: grid, keyframes and camera must be set");
    for (int f = 0; f < c->K; ++f) if (!c->bgr[(size_t)f * c->levels].p) return ctx_fail(c, I3D_ERR_STATE, "recompute_colors: keyframes were uploaded without colour images");
    if (num_observations > MAX_SLOTS || (num_observations <= 0 && c->K > MAX_SLOTS)) return ctx_fail(c, I3D_ERR_CAPACITY, "recompute_colors: more than 8 observations per voxel");
    CTX_HIP(c, hipSetDevice(c->device));
    OptParams p = color_params(c, occlusion_distance);
    std::vector<FrameConst> fc; build_frame_consts(c, 0, c->poses.data(), fc);
    CTX_HIP(c, hipMemcpyAsync(c->d_frames.p, fc.data(), sizeof(FrameConst) * fc.size(), hipMemcpyHostToDevice, c->stream));
    { TimedScope t(c, I3D_K_OBSERVE); launch_recolor(c->stream, c->grid_view(), p, c->d_frames.p, num_observations, c->color.p); }
    CTX_HIP(c, hipStreamSynchronize(c->stream));
    c->assembled = false;
    return I3D_OK;
}

// SDFAlgorithms::clearVoxelsOutsideThinShell (algorithms.cpp:368-458); erase keeps the iteration order of the survivors
int clear_outside_thin_shell(i3d_context* c, double thres_shell, int64_t* new_count) {
    if (!c->have_grid) return ctx_fail(c, I3D_ERR_STATE, "clear_outside_thin_shell: no grid");
    CTX_HIP(c, hipSetDevice(c->device));
    hipStream_t st = c->stream; const int N = c->N;
    GridView g = c->grid_view(); HashTable t{c->hkeys.p, c->hvals.p, c->hmask};
    DevBuf<int> keep, inv, keep_v, scan_v;
    CTX_HIP(c, keep.alloc(N)); CTX_HIP(c, inv.alloc(N)); CTX_HIP(c, keep_v.alloc(N)); CTX_HIP(c, scan_v.alloc(N));
    CTX_HIP(c, hipMemsetAsync(keep.p, 0, sizeof(int) * (size_t)N, st));
    launch_shell_mark(st, g, thres_shell, keep.p);
    launch_shell_crossing(st, g, t, keep.p);
    launch_inv_rank(st, N, c->rank.p, inv.p);
    launch_keep_visit(st, N, inv.p, keep.p, keep_v.p);
    CTX_HIP(c, rocprim::exclusive_scan(c->scan_tmp.p, c->scan_tmp_bytes, keep_v.p, scan_v.p, 0, (size_t)N, rocprim::plus<int>(), st));
    int tail[2];
    CTX_HIP(c, hipMemcpyAsync(&tail[0], scan_v.p + (N - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    CTX_HIP(c, hipMemcpyAsync(&tail[1], keep_v.p + (N - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    CTX_HIP(c, hipStreamSynchronize(st));
    const int M = tail[0] + tail[1];
    if (new_count) *new_count = M;
    if (M == N) return I3D_OK;
    if (M <= 0) return ctx_fail(c, I3D_ERR_STATE, "clear_outside_thin_shell: no voxel left");
    GridStaging s;
    CTX_HIP(c, s.kxyz.alloc((size_t)3 * M)); CTX_HIP(c, s.sdf.alloc(M)); CTX_HIP(c, s.sdf_ref.alloc(M)); CTX_HIP(c, s.alb.alloc(M)); CTX_HIP(c, s.w.alloc(M)); CTX_HIP(c, s.rgb.alloc((size_t)3 * M));
    launch_export_visit(st, g, inv.p, keep.p, scan_v.p, s.kxyz.p, s.sdf.p, s.sdf_ref.p, s.alb.p, s.w.p, s.rgb.p);
    return set_grid_device(c, M, c->voxel_size, c->truncation, s);
}

// SDFAlgorithms::upsample (algorithms.cpp:202-235): the new grid has voxel_size/2 and truncation 5*voxel_size/2 (sparse_voxel_grid.cpp:44-49)
int upsample_grid(i3d_context* c, int64_t* new_count) {
    if (!c->have_grid) return ctx_fail(c, I3D_ERR_STATE, "upsample_grid: no grid");
    if ((long long)c->N * 8 > (1ll << 30)) return ctx_fail(c, I3D_ERR_CAPACITY, "upsample_grid: more than 2^30 voxels");
    CTX_HIP(c, hipSetDevice(c->device));
    hipStream_t st = c->stream; const int N = c->N; const long long M = 8ll * N;
    GridView g = c->grid_view(); HashTable t{c->hkeys.p, c->hvals.p, c->hmask};
    DevBuf<int> inv, perm; CTX_HIP(c, inv.alloc(N)); CTX_HIP(c, perm.alloc(M));
    GridStaging a, b;
    CTX_HIP(c, a.kxyz.alloc((size_t)3 * M)); CTX_HIP(c, a.sdf.alloc(M)); CTX_HIP(c, a.sdf_ref.alloc(M)); CTX_HIP(c, a.alb.alloc(M)); CTX_HIP(c, a.w.alloc(M)); CTX_HIP(c, a.rgb.alloc((size_t)3 * M));
    CTX_HIP(c, b.kxyz.alloc((size_t)3 * M)); CTX_HIP(c, b.sdf.alloc(M)); CTX_HIP(c, b.sdf_ref.alloc(M)); CTX_HIP(c, b.alb.alloc(M)); CTX_HIP(c, b.w.alloc(M)); CTX_HIP(c, b.rgb.alloc((size_t)3 * M));
    launch_inv_rank(st, N, c->rank.p, inv.p);
    launch_upsample(st, g, t, inv.p, a.kxyz.p, a.sdf.p, a.sdf_ref.p, a.alb.p, a.w.p, a.rgb.p);
    // iteration order of the new map, on the device (child keys are distinct: every parent has its own 2x2x2 block)
    { const std::vector<MapEpoch> ep = map_epochs((size_t)M);
      CTX_HIP(c, map_order_device(st, a.kxyz.p, (size_t)M, ep.data(), (int)ep.size(), perm.p)); }
    launch_permute_staging(st, M, perm.p, a.kxyz.p, a.sdf.p, a.sdf_ref.p, a.alb.p, a.w.p, a.rgb.p, b.kxyz.p, b.sdf.p, b.sdf_ref.p, b.alb.p, b.w.p, b.rgb.p);
    CTX_HIP(c, hipStreamSynchronize(st));
    const float vs = c->voxel_size * 0.5f;
    if (new_count) *new_count = M;
    return set_grid_device(c, (int)M, vs, vs * 5.0f, b);
}

}  // namespace i3d

using namespace i3d;

extern "C" {

// parity probe: the replayed iteration order (mode 0: keys may repeat, 1: caller guarantees distinct keys) or the order of a real
// std::unordered_map (mode 2); returns the number of visited elements
int64_t i3d_debug_map_order(const int32_t* keys, int64_t n, int32_t mode, int32_t* order) {
    if (n < 0 || (n && (!keys || !order))) return -1;
    std::vector<int> o;
    if (mode == 2) map_iteration_order_stl(keys, (size_t)n, o);
    else if (mode == 3) map_iteration_order_epochs(keys, (size_t)n, o);
    else if (mode == 4) {                                     // the device path (distinct keys)
        if (n == 0) return 0;
        int *dk = nullptr, *dout = nullptr;
        const std::vector<MapEpoch> ep = map_epochs((size_t)n);
        bool ok = hipMalloc((void**)&dk, sizeof(int) * 3 * (size_t)n) == hipSuccess && hipMalloc((void**)&dout, sizeof(int) * (size_t)n) == hipSuccess
                  && hipMemcpy(dk, keys, sizeof(int) * 3 * (size_t)n, hipMemcpyHostToDevice) == hipSuccess
                  && map_order_device(nullptr, dk, (size_t)n, ep.data(), (int)ep.size(), dout) == hipSuccess
                  && hipMemcpy(order, dout, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess;
        if (dk) (void)hipFree(dk); if (dout) (void)hipFree(dout);
        return ok ? n : -1;
    }
    else map_iteration_order_replay(keys, (size_t)n, o, mode == 1);
    for (size_t i = 0; i < o.size(); ++i) order[i] = o[i];
    return (int64_t)o.size();
}
int i3d_recompute_colors(i3d_context* c, float occlusion_distance, int32_t num_observations) {
    if (!c) return I3D_ERR_INVALID_ARGUMENT;
    return recompute_colors(c, occlusion_distance, num_observations);
}
int i3d_clear_outside_thin_shell(i3d_context* c, double thres_shell, int64_t* new_count) {
    if (!c) return I3D_ERR_INVALID_ARGUMENT;
    return clear_outside_thin_shell(c, thres_shell, new_count);
}
int i3d_upsample(i3d_context* c, int64_t* new_count) {
    if (!c) return I3D_ERR_INVALID_ARGUMENT;
    return upsample_grid(c, new_count);
}
// Intrinsic3D::init's per-keyframe Pyramid(num_rgbd_levels, color, depth) (intrinsic3d.cpp:182-187, rgbd/pyramid.cpp:59-166) on the device:
// level-0 colour + depth in, float luminance / depth pyramids out (colour is kept at level 0 only: recomputeColors reads nothing else)
int i3d_set_frames_rgbd(i3d_context* c, int32_t K, int32_t levels, int32_t width, int32_t height, const uint8_t* const* bgr, const float* const* depth) {
    if (!c || K <= 0 || levels <= 0 || width <= 0 || height <= 0 || !bgr || !depth) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_set_frames_rgbd: bad arguments");
    if ((width >> (levels - 1)) < 1 || (height >> (levels - 1)) < 1) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_set_frames_rgbd: more levels than the image size allows");
    CTX_HIP(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    c->have_frames = false; c->K = K; c->levels = levels; c->cull_level = -1;
    c->slots = 0; c->assembled = false;            // K sizes the camera blocks and solver vectors: force alloc_rows() to run again
    c->fw.resize(levels); c->fh.resize(levels);
    for (int l = 0; l < levels; ++l) { c->fw[l] = l ? c->fw[l - 1] / 2 : width; c->fh[l] = l ? c->fh[l - 1] / 2 : height; }
    c->lum.clear(); c->depth.clear(); c->bgr.clear();
    c->lum.resize((size_t)K * levels); c->depth.resize((size_t)K * levels); c->bgr.resize((size_t)K * levels);
    for (int f = 0; f < K; ++f) {
        if (!bgr[f] || !depth[f]) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_set_frames_rgbd: null image");
        const size_t k0 = (size_t)f * levels, px = (size_t)width * height;
        CTX_HIP(c, c->bgr[k0].alloc(px * 3)); CTX_HIP(c, c->lum[k0].alloc(px + 4)); CTX_HIP(c, c->depth[k0].alloc(px));
        CTX_HIP(c, hipMemcpyAsync(c->bgr[k0].p, bgr[f], px * 3, hipMemcpyHostToDevice, st));
        CTX_HIP(c, hipMemcpyAsync(c->depth[k0].p, depth[f], px * sizeof(float), hipMemcpyHostToDevice, st));
        launch_lum_from_bgr(st, (int)px, c->bgr[k0].p, c->lum[k0].p);
        for (int l = 1; l < levels; ++l) {
            const size_t k = k0 + l, n = (size_t)c->fw[l] * c->fh[l];
            CTX_HIP(c, c->lum[k].alloc(n + 4)); CTX_HIP(c, c->depth[k].alloc(n));
            launch_pyr_down(st, c->fw[l - 1], c->fh[l - 1], c->lum[k - 1].p, c->fw[l], c->fh[l], c->lum[k].p);
            launch_depth_down(st, c->fw[l - 1], c->depth[k - 1].p, c->fw[l], c->fh[l], c->depth[k].p);
        }
    }
    CTX_HIP(c, c->d_frames.alloc(K)); CTX_HIP(c, c->d_frames_cand.alloc(K));
    CTX_HIP(c, hipStreamSynchronize(st));
    CTX_HIP(c, hipGetLastError());
    c->have_frames = true;
    if ((int)c->poses.size() != 6 * K) { c->poses.assign((size_t)6 * K, 0.0); c->have_camera = false; }
    return I3D_OK;
}
// resizeDepth(depth camera, depth, colour camera) (rgbd/processing.cpp:129-181), called per keyframe by Intrinsic3D::init (intrinsic3d.cpp:182-184)
// before the pyramid is built; intrinsics are (fx, fy, cx, cy) as floats.  Same size in and out: a plain copy, like the reference.
int i3d_resize_depth(int32_t device_ordinal, int32_t in_w, int32_t in_h, const float* depth_in, const float* in_intr, int32_t out_w, int32_t out_h,
                     const float* out_intr, float* depth_out) {
    if (!depth_in || !depth_out || !in_intr || !out_intr || in_w <= 0 || in_h <= 0 || out_w <= 0 || out_h <= 0) return I3D_ERR_INVALID_ARGUMENT;
    if (in_w == out_w && in_h == out_h) { std::memcpy(depth_out, depth_in, sizeof(float) * (size_t)in_w * in_h); return I3D_OK; }
    int ndev = 0; if (hipGetDeviceCount(&ndev) != hipSuccess || device_ordinal < 0 || device_ordinal >= ndev) return I3D_ERR_NO_DEVICE;
    if (hipSetDevice(device_ordinal) != hipSuccess) return I3D_ERR_HIP;
    DevBuf<float> a, b;
    if (a.alloc((size_t)in_w * in_h) != hipSuccess || b.alloc((size_t)out_w * out_h) != hipSuccess) return I3D_ERR_HIP;
    if (hipMemcpy(a.p, depth_in, sizeof(float) * (size_t)in_w * in_h, hipMemcpyHostToDevice) != hipSuccess) return I3D_ERR_HIP;
    launch_resize_depth(nullptr, in_w, in_h, a.p, in_intr, out_w, out_h, out_intr, b.p);
    if (hipMemcpy(depth_out, b.p, sizeof(float) * (size_t)out_w * out_h, hipMemcpyDeviceToHost) != hipSuccess) return I3D_ERR_HIP;
    return hipGetLastError() == hipSuccess ? I3D_OK : I3D_ERR_HIP;
}
// one pyramid image back to the host (parity probe / callers that still want the images)
int i3d_get_frame_image(i3d_context* c, int32_t frame, int32_t level, float* lum, float* depth) {
    if (!c || !c->have_frames || frame < 0 || frame >= c->K || level < 0 || level >= c->levels) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_get_frame_image: bad frame / level");
    const size_t k = (size_t)frame * c->levels + level, n = (size_t)c->fw[level] * c->fh[level];
    CTX_HIP(c, hipSetDevice(c->device));
    if (lum) CTX_HIP(c, hipMemcpy(lum, c->lum[k].p, n * sizeof(float), hipMemcpyDeviceToHost));
    if (depth) CTX_HIP(c, hipMemcpy(depth, c->depth[k].p, n * sizeof(float), hipMemcpyDeviceToHost));
    return I3D_OK;
}
int i3d_grid_info(i3d_context* c, int64_t* num_voxels, float* voxel_size, float* truncation) {
    if (!c || !c->have_grid) return ctx_fail(c, I3D_ERR_STATE, "i3d_grid_info: no grid");
    if (num_voxels) *num_voxels = c->N; if (voxel_size) *voxel_size = c->voxel_size; if (truncation) *truncation = c->truncation;
    return I3D_OK;
}
int i3d_export_grid(i3d_context* c, int32_t* keys, double* sdf, double* sdf_refined, double* albedo, float* weight, uint8_t* color) {
    if (!c || !c->have_grid) return ctx_fail(c, I3D_ERR_STATE, "i3d_export_grid: no grid");
    CTX_HIP(c, hipSetDevice(c->device));
    hipStream_t st = c->stream; const int N = c->N;
    DevBuf<int> inv; GridStaging s;
    CTX_HIP(c, inv.alloc(N)); CTX_HIP(c, s.kxyz.alloc((size_t)3 * N)); CTX_HIP(c, s.sdf.alloc(N)); CTX_HIP(c, s.sdf_ref.alloc(N)); CTX_HIP(c, s.alb.alloc(N)); CTX_HIP(c, s.w.alloc(N)); CTX_HIP(c, s.rgb.alloc((size_t)3 * N));
    launch_inv_rank(st, N, c->rank.p, inv.p);
    launch_export_visit(st, c->grid_view(), inv.p, nullptr, nullptr, s.kxyz.p, s.sdf.p, s.sdf_ref.p, s.alb.p, s.w.p, s.rgb.p);
    if (keys) CTX_HIP(c, hipMemcpyAsync(keys, s.kxyz.p, sizeof(int) * 3 * (size_t)N, hipMemcpyDeviceToHost, st));
    if (sdf) CTX_HIP(c, hipMemcpyAsync(sdf, s.sdf.p, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, st));
    if (sdf_refined) CTX_HIP(c, hipMemcpyAsync(sdf_refined, s.sdf_ref.p, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, st));
    if (albedo) CTX_HIP(c, hipMemcpyAsync(albedo, s.alb.p, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, st));
    if (weight) CTX_HIP(c, hipMemcpyAsync(weight, s.w.p, sizeof(float) * (size_t)N, hipMemcpyDeviceToHost, st));
    if (color) CTX_HIP(c, hipMemcpyAsync(color, s.rgb.p, (size_t)3 * N, hipMemcpyDeviceToHost, st));
    CTX_HIP(c, hipStreamSynchronize(st));
    return I3D_OK;
}

// SparseVoxelGrid<Voxel>::load (sparse_voxel_grid.cpp:545-568, records inserted in file order) followed by SDFAlgorithms::convert
// (algorithms.cpp:47-72: re-insertion in the Voxel map's iteration order, sdf_refined = sdf, albedo = 0.6, then clearInvalidVoxels)
int i3d_set_grid_from_tsdf_records(i3d_context* c, float voxel_size, int64_t n, const int32_t* keys, const float* sdf, const float* weight, const uint8_t* color) {
    if (!c || n <= 0 || !keys || !sdf || !weight || !color) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_set_grid_from_tsdf_records: bad arguments");
    // map 1: file order -> Voxel map (later records with the same key overwrite the payload, the node keeps its place)
    std::vector<int> o1;
    map_iteration_order_replay(keys, (size_t)n, o1, false);
    // map 2: VoxelSBR map filled in that order; invalid voxels (weight <= 0) are erased afterwards (order of the rest is unchanged)
    std::vector<int> k2(3 * o1.size()), o2(o1.size());
    for (size_t v = 0; v < o1.size(); ++v) for (int a = 0; a < 3; ++a) k2[3 * v + a] = keys[3 * (size_t)o1[v] + a];
    {   // distinct keys now: on the device (map_order.hip)
        CTX_HIP(c, hipSetDevice(c->device));
        DevBuf<int> dk, dord; CTX_HIP(c, dk.alloc(k2.size())); CTX_HIP(c, dord.alloc(o1.size()));
        CTX_HIP(c, hipMemcpyAsync(dk.p, k2.data(), sizeof(int) * k2.size(), hipMemcpyHostToDevice, c->stream));
        const std::vector<MapEpoch> ep = map_epochs(o1.size());
        CTX_HIP(c, map_order_device(c->stream, dk.p, o1.size(), ep.data(), (int)ep.size(), dord.p));
        CTX_HIP(c, hipMemcpy(o2.data(), dord.p, sizeof(int) * o2.size(), hipMemcpyDeviceToHost));
    }
    std::vector<int32_t> k; std::vector<double> s, a; std::vector<float> w; std::vector<uint8_t> col;
    for (size_t v = 0; v < o2.size(); ++v) {
        const size_t i = (size_t)o1[(size_t)o2[v]];
        if (!(weight[i] > 0.0f)) continue;
        k.push_back(keys[3 * i]); k.push_back(keys[3 * i + 1]); k.push_back(keys[3 * i + 2]);
        s.push_back((double)sdf[i]); a.push_back(0.6); w.push_back(weight[i]);
        col.push_back(color[3 * i]); col.push_back(color[3 * i + 1]); col.push_back(color[3 * i + 2]);
    }
    if (w.empty()) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_set_grid_from_tsdf_records: no valid voxel");
    i3d_grid_view gv; gv.num_voxels = (int64_t)w.size(); gv.voxel_size = voxel_size; gv.truncation = voxel_size * 5.0f;
    gv.keys = k.data(); gv.sdf = s.data(); gv.sdf_refined = s.data(); gv.albedo = a.data(); gv.weight = w.data(); gv.color = col.data();
    return i3d_set_grid(c, &gv);
}

static double refine_lambda(int it, int n, double l0, double l1) { if (n <= 1) return l0; return l0 + ((l1 - l0) / (double)(n - 1)) * (double)it; }

// Intrinsic3D::refine (intrinsic3d.cpp:206-290) on the resident grid / keyframes / camera
int i3d_refine(i3d_context* c, const i3d_refine_config* rc, const i3d_optimizer_config* oc, i3d_refine_callback cb, void* user) {
    if (!c || !rc || !oc) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_refine: null argument");
    if (!c->have_grid || rc->num_grid_levels <= 0 || rc->num_rgbd_levels <= 0) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_refine: no grid or no levels");   // :208-211
    if (rc->num_rgbd_levels > c->levels) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_refine: more rgbd levels than uploaded pyramid levels");
    int rcode = recompute_colors(c, rc->occlusion_distance, rc->num_observations);            // init(): initial SDF recolorization (:196-201)
    if (rcode) return rcode;
    const int coarsest = rc->num_grid_levels - 1;
    for (int gl = coarsest; gl >= 0; --gl) {
        double factor = rc->thin_shell_factor;                                                 // prepareGridLevel (:298-316)
        if (rc->thin_shell_factor_final > 0.0) factor = refine_lambda(coarsest - gl, rc->num_grid_levels, rc->thin_shell_factor, rc->thin_shell_factor_final);
        const double thres = factor * (double)c->voxel_size;
        if (rc->clear_distant_voxels) { rcode = clear_outside_thin_shell(c, thres, nullptr); if (rcode) return rcode; }
        for (int pl = rc->num_rgbd_levels - 1; pl >= 0; --pl) {
            if (pl > 0 && gl < coarsest) continue;                                             // all pyramid levels only on the coarsest grid (:245)
            i3d_sh_stats shst; int32_t S = 0;
            rcode = estimate_sh(c, rc->subvolume_size_sh, rc->sh_lambda_reg, thres, &S, nullptr, nullptr, 1 << 30, &shst);
            if (rcode) break;                                                                  // "lighting estimation not successful": leave the rgbd loop (:257-261)
            i3d_optimizer_config o = *oc; o.thres_shell = thres; o.grid_level = gl; o.rgbd_level = pl;
            o.occlusion_distance = rc->occlusion_distance; o.num_observations = rc->num_observations;
            rcode = optimize(c, o, nullptr);
            if (rcode && rcode != I3D_ERR_INVALID_ARGUMENT) return rcode;                      // the reference logs a failed optimize and goes on (:273-276)
            rcode = recompute_colors(c, rc->occlusion_distance, rc->num_observations);         // finishRgbdLevel (:353-378)
            if (rcode) return rcode;
            if (cb) cb(user, gl, rc->num_grid_levels, pl, rc->num_rgbd_levels);                // notifyCallbacks -> onSDFRefined (:282)
        }
        if (gl > 0) { rcode = upsample_grid(c, nullptr); if (rcode) return rcode; }            // finishGridLevel (:320-333)
    }
    return I3D_OK;
}

}  // extern "C"
