// TSDF fusion on the device — AppFusion::fuseSDF's volume (apps/src/app_fusion.cpp:107-200) behind a C handle.
//   i3d_fusion_integrate = erodeDiscontinuities + computeNormals + SparseVoxelGrid<Voxel>::integrate (alloc + update) for one frame
//   i3d_fusion_finish    = SDFAlgorithms::correctSDF + clearInvalidVoxels, and the reference's record order
// Frames are integrated in call order, one allocation launch and one integration launch per frame (fusion_kernels.hip).  The saved
// volume's record order is the iteration order of the reference's unordered_map; it is reproduced from the order of first insertion
// (a per-voxel rank kept by the allocation kernel, radix-sorted here) by replaying the insertions epoch by epoch on the device (map_order.hip) — the same replay levels.cpp uses for the level
// transitions.
#include "../../../include/intrinsic3d_hip.h"
#include "../device/fusion_kernels.hpp"
#include "context.hpp"
#include "map_order.hpp"
#include <rocprim/rocprim.hpp>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

using namespace i3d;

namespace {

// Matrix4f::inverse(): adjugate over determinant from the 2x2 minors of the row pairs, in float
void inverse4f(const float* m, float* inv) {
    const float s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2], s2 = m[0] * m[7] - m[4] * m[3];
    const float s3 = m[1] * m[6] - m[5] * m[2], s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
    const float c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11], c3 = m[9] * m[14] - m[13] * m[10];
    const float c2 = m[8] * m[15] - m[12] * m[11], c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
    const float id = 1.0f / (((((s0 * c5 - s1 * c4) + s2 * c3) + s3 * c2) - s4 * c1) + s5 * c0);
    inv[0] = ((m[5] * c5 - m[6] * c4) + m[7] * c3) * id;      inv[1] = ((-m[1] * c5 + m[2] * c4) - m[3] * c3) * id;
    inv[2] = ((m[13] * s5 - m[14] * s4) + m[15] * s3) * id;   inv[3] = ((-m[9] * s5 + m[10] * s4) - m[11] * s3) * id;
    inv[4] = ((-m[4] * c5 + m[6] * c2) - m[7] * c1) * id;     inv[5] = ((m[0] * c5 - m[2] * c2) + m[3] * c1) * id;
    inv[6] = ((-m[12] * s5 + m[14] * s2) - m[15] * s1) * id;  inv[7] = ((m[8] * s5 - m[10] * s2) + m[11] * s1) * id;
    inv[8] = ((m[4] * c4 - m[5] * c2) + m[7] * c0) * id;      inv[9] = ((-m[0] * c4 + m[1] * c2) - m[3] * c0) * id;
    inv[10] = ((m[12] * s4 - m[13] * s2) + m[15] * s0) * id;  inv[11] = ((-m[8] * s4 + m[9] * s2) - m[11] * s0) * id;
    inv[12] = ((-m[4] * c3 + m[5] * c1) - m[6] * c0) * id;    inv[13] = ((m[0] * c3 - m[1] * c1) + m[2] * c0) * id;
    inv[14] = ((-m[12] * s3 + m[13] * s1) - m[14] * s0) * id; inv[15] = ((m[8] * s3 - m[9] * s1) + m[10] * s0) * id;
}

}  // namespace

struct i3d_fusion {
    int device = 0; hipStream_t stream = nullptr;
    float voxel_size = 0, truncation = 0, depth_min = 0, depth_max = 0, weight_sample = 10.0f;      // sparse_voxel_grid.cpp:44-51
    float clip[6] = {0, 0, 0, 0, 0, 0}; bool use_clip = false;
    unsigned long long capacity = 0, frames = 0;
    DevBuf<unsigned long long> keys, rank, crank; DevBuf<float> sdf, weight; DevBuf<uchar4> color;
    DevBuf<unsigned long long> d_count; DevBuf<int> d_flag;
    DevBuf<float> d_depth_raw, d_depth, d_normals; DevBuf<uint8_t> d_bgr;
    // result of finish()
    bool finished = false, corrected = false;      // corrected: correctSDF has been written into the table (finish is not re-runnable past that point)
    std::vector<int32_t> out_keys; std::vector<float> out_sdf, out_weight; std::vector<uint8_t> out_color;
    unsigned long long allocated = 0; int correct_launches = 0;
    std::string error;
    FusionTable table() { return FusionTable{keys.p, sdf.p, weight.p, color.p, rank.p, crank.p, capacity - 1}; }
};

namespace {

int fail(i3d_fusion* f, int code, const std::string& msg) { if (f) f->error = msg; return code; }
#define F_HIP(f, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(f, I3D_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

int alloc_table(i3d_fusion* f, unsigned long long cap, DevBuf<unsigned long long>& keys, DevBuf<unsigned long long>& rank, DevBuf<unsigned long long>& crank,
                DevBuf<float>& sdf, DevBuf<float>& weight, DevBuf<uchar4>& color) {
    F_HIP(f, keys.alloc(cap)); F_HIP(f, rank.alloc(cap)); F_HIP(f, crank.alloc(cap)); F_HIP(f, sdf.alloc(cap)); F_HIP(f, weight.alloc(cap)); F_HIP(f, color.alloc(cap));
    launch_fusion_clear(f->stream, FusionTable{keys.p, sdf.p, weight.p, color.p, rank.p, crank.p, cap - 1});
    return I3D_OK;
}
int grow(i3d_fusion* f) {
    if (f->capacity >= (1ull << 31)) return fail(f, I3D_ERR_CAPACITY, "fusion: more than 2^31 table slots");
    DevBuf<unsigned long long> keys, rank, crank; DevBuf<float> sdf, weight; DevBuf<uchar4> color;
    const unsigned long long cap = f->capacity * 2;
    const int rc = alloc_table(f, cap, keys, rank, crank, sdf, weight, color); if (rc) return rc;
    launch_fusion_rehash(f->stream, f->table(), FusionTable{keys.p, sdf.p, weight.p, color.p, rank.p, crank.p, cap - 1});
    F_HIP(f, hipStreamSynchronize(f->stream));
    f->keys = std::move(keys); f->rank = std::move(rank); f->crank = std::move(crank); f->sdf = std::move(sdf); f->weight = std::move(weight); f->color = std::move(color);
    f->capacity = cap;
    return I3D_OK;
}

// SparseVoxelGrid::computeFrustumBounds (sparse_voxel_grid.cpp:573-606); floor / ceil act on metres before the voxel conversion
void frustum_bounds(const i3d_fusion* f, const FusionCam& cam, const float* pose, int b[6]) {
    const int lo = std::numeric_limits<int>::min(), hi = std::numeric_limits<int>::max();
    b[0] = hi; b[1] = lo; b[2] = hi; b[3] = lo; b[4] = hi; b[5] = lo;
    const int px[4] = {0, cam.w - 1, cam.w - 1, 0}, py[4] = {0, 0, cam.h - 1, cam.h - 1};
    const float inv = 1.0f / f->voxel_size;
    auto to_voxel = [&](float v) { return (int)(v * inv + 0.5f); };
    for (int i = 0; i < 8; ++i) {
        const float depth = i < 4 ? f->depth_min : f->depth_max;
        float c[3] = {0, 0, 0};
        if (depth != 0.0f) { const float x = ((float)px[i & 3] - cam.cx) / cam.fx, y = ((float)py[i & 3] - cam.cy) / cam.fy; c[0] = depth * x; c[1] = depth * y; c[2] = depth; }
        for (int a = 0; a < 3; ++a) {
            const float pt = (pose[4 * a] * c[0] + (pose[4 * a + 1] * c[1] + pose[4 * a + 2] * c[2])) + pose[4 * a + 3];    // fixed-size Eigen product: halving reduction
            const int pl = to_voxel((float)(int)std::floor(pt)), pu = to_voxel((float)(int)std::ceil(pt));
            b[2 * a] = std::min(b[2 * a], std::min(pl, pu)); b[2 * a + 1] = std::max(b[2 * a + 1], std::max(pl, pu));
        }
    }
}

}  // namespace

extern "C" {

int i3d_fusion_create(int32_t device_ordinal, float voxel_size, float depth_min, float depth_max, const float* clip6, uint64_t initial_capacity, i3d_fusion** out) {
    if (!out) return I3D_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (voxel_size <= 0.00001f) return I3D_ERR_INVALID_ARGUMENT;                   // SparseVoxelGrid::create returns nullptr
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_ordinal < 0 || device_ordinal >= ndev) return I3D_ERR_NO_DEVICE;
    if (hipSetDevice(device_ordinal) != hipSuccess) return I3D_ERR_NO_DEVICE;
    i3d_fusion* f = new i3d_fusion();
    f->device = device_ordinal; f->voxel_size = voxel_size; f->truncation = voxel_size * 5.0f; f->depth_min = depth_min; f->depth_max = depth_max;
    if (clip6) { float n = 0.0f; for (int i = 0; i < 6; ++i) { f->clip[i] = clip6[i]; n += clip6[i] * clip6[i]; } f->use_clip = std::sqrt(n) > 0.0f; }
    unsigned long long cap = 1ull << 12;
    while (cap < initial_capacity * 2 && cap < (1ull << 31)) cap <<= 1;
    f->capacity = cap;
    if (hipStreamCreate(&f->stream) != hipSuccess) { delete f; return I3D_ERR_HIP; }
    int rc = alloc_table(f, cap, f->keys, f->rank, f->crank, f->sdf, f->weight, f->color);
    if (rc == I3D_OK && (f->d_count.alloc(1) != hipSuccess || f->d_flag.alloc(2) != hipSuccess)) rc = I3D_ERR_HIP;
    if (rc == I3D_OK && hipMemsetAsync(f->d_count.p, 0, sizeof(unsigned long long), f->stream) != hipSuccess) rc = I3D_ERR_HIP;
    if (rc == I3D_OK && hipStreamSynchronize(f->stream) != hipSuccess) rc = I3D_ERR_HIP;
    if (rc != I3D_OK) { (void)hipStreamDestroy(f->stream); delete f; return rc; }
    *out = f;
    return I3D_OK;
}

void i3d_fusion_destroy(i3d_fusion* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->stream) (void)hipStreamDestroy(f->stream);
    delete f;
}
const char* i3d_fusion_last_error(const i3d_fusion* f) { return f ? f->error.c_str() : "null fusion handle"; }

int i3d_fusion_integrate(i3d_fusion* f, int32_t dw, int32_t dh, const float* dcam4, int32_t cw, int32_t ch, const float* ccam4, const float* depth, const uint8_t* bgr,
                         const float* pose16, int32_t erode_window) {
    if (!f) return I3D_ERR_INVALID_ARGUMENT;
    if (dw <= 0 || dh <= 0 || cw <= 0 || ch <= 0 || !dcam4 || !ccam4 || !depth || !bgr || !pose16) return fail(f, I3D_ERR_INVALID_ARGUMENT, "i3d_fusion_integrate: bad arguments");
    if (f->finished || f->corrected) return fail(f, I3D_ERR_STATE, "i3d_fusion_integrate: the volume has been finished (or a finish failed after correcting the table)");
    F_HIP(f, hipSetDevice(f->device));
    hipStream_t st = f->stream;
    const size_t dn = (size_t)dw * dh, cn = (size_t)cw * ch;
    F_HIP(f, f->d_depth_raw.alloc(dn)); F_HIP(f, f->d_depth.alloc(dn)); F_HIP(f, f->d_normals.alloc(dn * 3)); F_HIP(f, f->d_bgr.alloc(cn * 3));
    F_HIP(f, hipMemcpyAsync(f->d_depth_raw.p, depth, dn * sizeof(float), hipMemcpyHostToDevice, st));
    F_HIP(f, hipMemcpyAsync(f->d_bgr.p, bgr, cn * 3, hipMemcpyHostToDevice, st));
    const FusionCam dcam{dcam4[0], dcam4[1], dcam4[2], dcam4[3], dw, dh}, ccam{ccam4[0], ccam4[1], ccam4[2], ccam4[3], cw, ch};
    launch_erode(st, dw, dh, f->d_depth_raw.p, erode_window, 0.5f, f->d_depth.p);                  // processing.h:57 default max_depth_diff
    launch_normals(st, dcam, f->d_depth.p, 0.3f, f->d_normals.p);                                  // processing.h:53 default depth_threshold
    FusionFrame fr; std::memset(&fr, 0, sizeof(fr));
    fr.voxel_size = f->voxel_size; fr.truncation = f->truncation; fr.depth_min = f->depth_min; fr.depth_max = f->depth_max; fr.weight_sample = f->weight_sample;
    for (int i = 0; i < 6; ++i) fr.clip[i] = f->clip[i];
    fr.use_clip = f->use_clip ? 1 : 0; fr.frame = f->frames;
    std::memcpy(fr.c2w, pose16, sizeof(fr.c2w)); inverse4f(pose16, fr.w2c);
    frustum_bounds(f, dcam, pose16, fr.bounds);
    for (int i = 0; i < 6; ++i) if (fr.bounds[i] <= -FUSION_COORD_OFFSET + 2 || fr.bounds[i] >= FUSION_COORD_OFFSET - 2)
        fr.bounds[i] = fr.bounds[i] < 0 ? -FUSION_COORD_OFFSET + 2 : FUSION_COORD_OFFSET - 2;     // keys are packed in 21 bits per axis
    for (;;) {                                                                                       // allocation is idempotent: repeat after growth
        F_HIP(f, hipMemsetAsync(f->d_flag.p, 0, sizeof(int), st));
        launch_fusion_alloc(st, f->table(), fr, dcam, f->d_depth.p, (unsigned long long)(0.6 * (double)f->capacity), f->d_count.p, f->d_flag.p);
        int overflow = 0;
        F_HIP(f, hipMemcpyAsync(&overflow, f->d_flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
        F_HIP(f, hipStreamSynchronize(st));
        if (overflow) { const int rc = grow(f); if (rc) return rc; continue; }
        // the in-kernel load test lags by the inserts of the waves in flight: keep the load factor of the finished frame below 0.6 as well
        unsigned long long have = 0;
        F_HIP(f, hipMemcpyAsync(&have, f->d_count.p, sizeof(have), hipMemcpyDeviceToHost, st));
        F_HIP(f, hipStreamSynchronize(st));
        if ((double)have <= 0.6 * (double)f->capacity) break;
        const int rc = grow(f); if (rc) return rc;         // rehash only: every cell of this frame already exists
        break;
    }
    launch_fusion_integrate(st, f->table(), fr, dcam, ccam, f->d_depth.p, f->d_normals.p, f->d_bgr.p);
    F_HIP(f, hipStreamSynchronize(st));                                                              // the host buffers may be reused by the caller
    ++f->frames;
    return I3D_OK;
}

int i3d_fusion_finish(i3d_fusion* f, int32_t correct_iterations, uint64_t* count) {
    if (!f) return I3D_ERR_INVALID_ARGUMENT;
    if (f->finished) { if (count) *count = f->out_sdf.size(); return I3D_OK; }
    if (f->corrected) return fail(f, I3D_ERR_STATE, "i3d_fusion_finish: an earlier finish failed after correctSDF had been written into the table; the volume cannot be finished twice");
    F_HIP(f, hipSetDevice(f->device));
    hipStream_t st = f->stream; FusionTable t = f->table();
    const unsigned long long cap = f->capacity;
    // 1. occupied slots, sorted by first-insertion rank = the reference's insertion sequence
    DevBuf<int> flags, offs; DevBuf<char> tmp; size_t bytes = 0;
    F_HIP(f, flags.alloc(cap)); F_HIP(f, offs.alloc(cap));
    launch_fusion_occupied(st, t, flags.p);
    F_HIP(f, rocprim::exclusive_scan(nullptr, bytes, flags.p, offs.p, 0, (size_t)cap, rocprim::plus<int>(), st));
    F_HIP(f, tmp.alloc(bytes));
    F_HIP(f, rocprim::exclusive_scan(tmp.p, bytes, flags.p, offs.p, 0, (size_t)cap, rocprim::plus<int>(), st));
    unsigned long long m = 0;
    F_HIP(f, hipMemcpyAsync(&m, f->d_count.p, sizeof(m), hipMemcpyDeviceToHost, st));
    F_HIP(f, hipStreamSynchronize(st));
    f->allocated = m;
    if (m > 0x7FFFFFFFull) return fail(f, I3D_ERR_CAPACITY, "fusion: more than 2^31 voxels");
    f->out_keys.clear(); f->out_sdf.clear(); f->out_weight.clear(); f->out_color.clear();
    if (m == 0) { f->finished = true; if (count) *count = 0; return I3D_OK; }
    DevBuf<unsigned long long> rank0, rank1; DevBuf<unsigned int> slot0, slot1;
    F_HIP(f, rank0.alloc(m)); F_HIP(f, rank1.alloc(m)); F_HIP(f, slot0.alloc(m)); F_HIP(f, slot1.alloc(m));
    launch_fusion_gather_rank(st, t, flags.p, offs.p, rank0.p, slot0.p);
    bytes = 0;
    F_HIP(f, rocprim::radix_sort_pairs(nullptr, bytes, rank0.p, rank1.p, slot0.p, slot1.p, (size_t)m, 0, 64, st));
    F_HIP(f, tmp.alloc(bytes));
    F_HIP(f, rocprim::radix_sort_pairs(tmp.p, bytes, rank0.p, rank1.p, slot0.p, slot1.p, (size_t)m, 0, 64, st));
    // 2. replay the insertions: iteration order of the reference's map
    DevBuf<int> kxyz; F_HIP(f, kxyz.alloc(3 * m));
    launch_fusion_keys(st, t, (long long)m, slot1.p, kxyz.p);
    DevBuf<int> d_order, pos_of_slot; DevBuf<unsigned int> visit_slot;
    F_HIP(f, d_order.alloc(m)); F_HIP(f, pos_of_slot.alloc(cap)); F_HIP(f, visit_slot.alloc(m));
    { const std::vector<MapEpoch> ep = map_epochs((size_t)m);                                            // keys of a hash table: distinct by construction
      F_HIP(f, map_order_device(st, kxyz.p, (size_t)m, ep.data(), (int)ep.size(), d_order.p)); }
    launch_fusion_positions(st, (long long)m, slot1.p, d_order.p, visit_slot.p, pos_of_slot.p);
    // 3. correctSDF: up to `correct_iterations` in-place sweeps, each evaluated as a fixed point (see k_correct), in a spatially sorted
    //    compact index space with the 26 neighbour indices resolved once
    if (correct_iterations > 0) {
        DevBuf<unsigned long long> sk0, sk1; DevBuf<unsigned int> slot_c; DevBuf<int> compact_of_slot, c_pos, nbr; DevBuf<float> c_sdf, c_cur;
        DevBuf<unsigned char> c_valid, c_touched, c_upd;
        F_HIP(f, sk0.alloc(m)); F_HIP(f, sk1.alloc(m)); F_HIP(f, slot_c.alloc(m)); F_HIP(f, compact_of_slot.alloc(cap)); F_HIP(f, c_pos.alloc(m));
        F_HIP(f, nbr.alloc(26 * m)); F_HIP(f, c_sdf.alloc(m)); F_HIP(f, c_cur.alloc(m)); F_HIP(f, c_valid.alloc(m)); F_HIP(f, c_touched.alloc(m)); F_HIP(f, c_upd.alloc(m));
        launch_fusion_spatial_keys(st, t, (long long)m, slot1.p, sk0.p);
        bytes = 0;
        F_HIP(f, rocprim::radix_sort_pairs(nullptr, bytes, sk0.p, sk1.p, slot1.p, slot_c.p, (size_t)m, 0, 64, st));
        F_HIP(f, tmp.alloc(bytes));
        F_HIP(f, rocprim::radix_sort_pairs(tmp.p, bytes, sk0.p, sk1.p, slot1.p, slot_c.p, (size_t)m, 0, 64, st));
        launch_fusion_compact_init(st, t, (long long)m, slot_c.p, pos_of_slot.p, compact_of_slot.p, c_sdf.p, c_pos.p, c_valid.p, c_touched.p);
        launch_fusion_build_nbr(st, t, (long long)m, slot_c.p, compact_of_slot.p, c_valid.p, nbr.p);
        f->correct_launches = 0;
        for (int iter = 0; iter < correct_iterations; ++iter) {
            F_HIP(f, hipMemcpyAsync(c_cur.p, c_sdf.p, sizeof(float) * m, hipMemcpyDeviceToDevice, st));
            F_HIP(f, hipMemsetAsync(c_upd.p, 0, m, st));
            for (;;) {
                F_HIP(f, hipMemsetAsync(f->d_flag.p, 0, 2 * sizeof(int), st));
                launch_fusion_correct(st, t, (long long)m, f->voxel_size, slot_c.p, nbr.p, c_pos.p, c_valid.p, c_sdf.p, c_cur.p, c_upd.p, f->d_flag.p);
                int changed = 0;
                F_HIP(f, hipMemcpyAsync(&changed, f->d_flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
                F_HIP(f, hipStreamSynchronize(st));
                ++f->correct_launches;
                if (!changed) break;
            }
            launch_fusion_commit(st, (long long)m, c_valid.p, c_cur.p, c_upd.p, c_sdf.p, c_touched.p, f->d_flag.p + 1);
            int has_update = 0;
            F_HIP(f, hipMemcpyAsync(&has_update, f->d_flag.p + 1, sizeof(int), hipMemcpyDeviceToHost, st));
            F_HIP(f, hipStreamSynchronize(st));
            if (!has_update) break;
        }
        f->corrected = true;                              // from here on the table holds corrected values: a retry would correct them twice
        launch_fusion_write_back(st, t, (long long)m, slot_c.p, c_sdf.p, c_touched.p);
        F_HIP(f, hipStreamSynchronize(st));
    }
    // 4. clearInvalidVoxels + records in iteration order
    DevBuf<int> vflags, voffs;
    F_HIP(f, vflags.alloc(m)); F_HIP(f, voffs.alloc(m));
    launch_fusion_valid(st, t, (long long)m, visit_slot.p, vflags.p);
    bytes = 0;
    F_HIP(f, rocprim::exclusive_scan(nullptr, bytes, vflags.p, voffs.p, 0, (size_t)m, rocprim::plus<int>(), st));
    F_HIP(f, tmp.alloc(bytes));
    F_HIP(f, rocprim::exclusive_scan(tmp.p, bytes, vflags.p, voffs.p, 0, (size_t)m, rocprim::plus<int>(), st));
    int last_off = 0, last_flag = 0;
    F_HIP(f, hipMemcpyAsync(&last_off, voffs.p + (m - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    F_HIP(f, hipMemcpyAsync(&last_flag, vflags.p + (m - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    F_HIP(f, hipStreamSynchronize(st));
    const size_t nv = (size_t)last_off + (size_t)last_flag;
    if (nv) {
        DevBuf<int> ok; DevBuf<float> os, ow; DevBuf<uint8_t> oc;
        F_HIP(f, ok.alloc(3 * nv)); F_HIP(f, os.alloc(nv)); F_HIP(f, ow.alloc(nv)); F_HIP(f, oc.alloc(3 * nv));
        launch_fusion_export(st, t, (long long)m, visit_slot.p, vflags.p, voffs.p, ok.p, os.p, ow.p, oc.p);
        f->out_keys.resize(3 * nv); f->out_sdf.resize(nv); f->out_weight.resize(nv); f->out_color.resize(3 * nv);
        F_HIP(f, hipMemcpyAsync(f->out_keys.data(), ok.p, sizeof(int) * 3 * nv, hipMemcpyDeviceToHost, st));
        F_HIP(f, hipMemcpyAsync(f->out_sdf.data(), os.p, sizeof(float) * nv, hipMemcpyDeviceToHost, st));
        F_HIP(f, hipMemcpyAsync(f->out_weight.data(), ow.p, sizeof(float) * nv, hipMemcpyDeviceToHost, st));
        F_HIP(f, hipMemcpyAsync(f->out_color.data(), oc.p, 3 * nv, hipMemcpyDeviceToHost, st));
        F_HIP(f, hipStreamSynchronize(st));
    }
    f->finished = true;                                  // only now: a failed finish can be diagnosed (i3d_fusion_last_error) and is not mistaken for an empty volume
    if (count) *count = nv;
    return I3D_OK;
}

int i3d_fusion_info(const i3d_fusion* f, uint64_t* frames, uint64_t* allocated, uint64_t* capacity, int32_t* correct_launches) {
    if (!f) return I3D_ERR_INVALID_ARGUMENT;
    if (frames) *frames = f->frames;
    if (allocated) *allocated = f->allocated;
    if (capacity) *capacity = f->capacity;
    if (correct_launches) *correct_launches = f->correct_launches;
    return I3D_OK;
}

int i3d_fusion_get(const i3d_fusion* f, int32_t* keys, float* sdf, float* weight, uint8_t* color) {
    if (!f || !f->finished) return I3D_ERR_STATE;
    const size_t n = f->out_sdf.size();
    if (keys) std::memcpy(keys, f->out_keys.data(), sizeof(int32_t) * 3 * n);
    if (sdf) std::memcpy(sdf, f->out_sdf.data(), sizeof(float) * n);
    if (weight) std::memcpy(weight, f->out_weight.data(), sizeof(float) * n);
    if (color) std::memcpy(color, f->out_color.data(), 3 * n);
    return I3D_OK;
}

// SparseVoxelGrid<Voxel>::save of the finished volume (sparse_voxel_grid.cpp:484-520)
int i3d_fusion_save(const i3d_fusion* f, const char* path) {
    if (!f || !path) return I3D_ERR_INVALID_ARGUMENT;
    if (!f->finished) return I3D_ERR_STATE;
    return i3d_tsdf_write(path, f->voxel_size, f->truncation, f->weight_sample, 0.6f, f->out_sdf.size(), f->out_keys.data(), f->out_sdf.data(), f->out_weight.data(),
                          f->out_color.data());
}

}  // extern "C"
