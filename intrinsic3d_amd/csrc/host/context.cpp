// Context lifetime, uploads (grid / keyframes / camera / per-voxel SH), write-back, per-keyframe constants, timing.
#include "context.hpp"
#include "../device/frame_math.hpp"
#include <rocprim/rocprim.hpp>
#include <algorithm>

using namespace i3d;

static thread_local std::string g_create_error;

namespace i3d {

int ctx_fail(i3d_context* c, int code, const std::string& msg) { if (c) c->err = msg; else g_create_error = msg; return code; }
int ctx_hip(i3d_context* c, hipError_t e, const char* what) {
    return ctx_fail(c, I3D_ERR_HIP, std::string(what) + " -> " + hipGetErrorString(e));
}

int ctx_launch_check(i3d_context* c) {
    char msg[256];
    if (take_launch_error(msg, sizeof(msg))) return ctx_fail(c, I3D_ERR_CAPACITY, msg);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? I3D_OK : ctx_hip(c, e, "kernel launch");
}

int ensure_pinned(i3d_context* c, size_t n) {
    if (n <= c->h_pinned_n) return I3D_OK;
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    c->h_pinned = nullptr; c->h_pinned_n = 0;
    CTX_HIP(c, hipHostMalloc((void**)&c->h_pinned, n * sizeof(double), hipHostMallocDefault));
    c->h_pinned_n = n;
    return I3D_OK;
}

// ---- timing: one HIP event pair per launch on the context's stream, resolved at flush -------------------------
bool timing_begin(i3d_context* c, int cat) {
    if (!c->timing.on || !((c->timing.mask >> cat) & 1u)) return false;
    Timing::Pending p; p.cat = cat;
    auto get = [&]() { hipEvent_t e; if (!c->timing.pool.empty()) { e = c->timing.pool.back(); c->timing.pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
    p.a = get(); p.b = get();
    (void)hipEventRecord(p.a, c->stream);
    c->timing.pending.push_back(p);
    return true;
}
void timing_end(i3d_context* c) {
    if (c->timing.pending.empty()) return;
    (void)hipEventRecord(c->timing.pending.back().b, c->stream);
}
void timing_flush(i3d_context* c) {
    if (c->timing.pending.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->timing.pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { c->timing.ms[p.cat] += ms; c->timing.launches[p.cat] += 1; c->timing.each[p.cat].push_back(ms); }
        c->timing.pool.push_back(p.a); c->timing.pool.push_back(p.b);
    }
    c->timing.pending.clear();
}

// ---- per-keyframe constants ------------------------------------------------------------------------------------
// (the formulas live in device/frame_math.hpp: the device builds the same constants for the candidate point of every LM attempt)
void build_frame_consts(const i3d_context* c, int level, const double* poses, std::vector<FrameConst>& out) {
    out.resize(c->K);
    for (int f = 0; f < c->K; ++f) {
        FrameConst& fc = out[f];
        fm::frame_from_pose(poses + 6 * f, fc);
        const size_t k = (size_t)f * c->levels + level;
        fc.hot.lum = c->lum[k].p; fc.depth = c->depth[k].p; fc.bgr = c->bgr[k].p;
        fc.w = c->fw[level]; fc.h = c->fh[level];
    }
}

}  // namespace i3d

i3d::GridView i3d_context::grid_view() const {
    GridView g;
    g.N = N; g.voxel_size = voxel_size; g.truncation = truncation;
    g.cx = cx.p; g.cy = cy.p; g.cz = cz.p; g.rank = rank.p; g.nbr = nbr.p; g.weight = weight.p; g.color = color.p;
    g.sdf0 = sdf0.p; g.x_sdf = x_sdf.p; g.x_alb = x_alb.p; g.f_sdf = f_sdf.p; g.f_alb = f_alb.p; g.sh = sh.p;
    g.flags = flags.p; g.aidx = aidx.p;
    return g;
}
i3d::RowView i3d_context::row_view() const {
    RowView r;
    r.A = A; r.Acap = Acap; r.slots = slots; r.chunk = chunk; r.world = comm ? comm->world : 1; r.own0 = own0; r.own1 = own1;
    r.clist = (comm && (comm->world > 1 || comm->force)) ? clist.p : nullptr; r.nC = nC; r.alist = alist.p; r.aflags = aflags.p; r.anbr = anbr.p; r.obs_frame = obs_frame.p; r.obs_w = obs_w.p;
    r.rows = rows.p; r.row_wr = row_wr.p; r.nrows = nrows.p; r.gmax = gmax.p; r.regflags = regflags.p; r.ea_w = ea_w.p; r.ea_free = ea_free.p;
    return r;
}

extern "C" {

const char* i3d_version(void) { return "intrinsic3d_hip 0.1 (gfx950)"; }
const char* i3d_last_error(const i3d_context* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int i3d_create(int32_t device_ordinal, i3d_context** out) {
    if (!out) return ctx_fail(nullptr, I3D_ERR_INVALID_ARGUMENT, "i3d_create: null output pointer");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return ctx_fail(nullptr, I3D_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU fallback");
    if (device_ordinal < 0 || device_ordinal >= count) return ctx_fail(nullptr, I3D_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    e = hipSetDevice(device_ordinal);
    if (e != hipSuccess) return ctx_hip(nullptr, e, "hipSetDevice");
    auto* c = new i3d_context();
    c->device = device_ordinal;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return ctx_hip(nullptr, e, "hipStreamCreate"); }
    *out = c;
    return I3D_OK;
}

void i3d_destroy(i3d_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    timing_flush(c);
    for (auto e : c->timing.pool) (void)hipEventDestroy(e);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->h_pcg) (void)hipHostFree(c->h_pcg);
    if (c->h_flags) (void)hipHostFree(c->h_flags);
    if (c->h_lmrec) (void)hipHostFree(c->h_lmrec);
    for (auto e : c->ev_asm) if (e) (void)hipEventDestroy(e);
    for (auto e : c->pcg_ev) if (e) (void)hipEventDestroy(e);
    delete c->comm;
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int i3d_set_grid(i3d_context* c, const i3d_grid_view* gv) {
    if (!c || !gv || gv->num_voxels <= 0 || !gv->keys || !gv->sdf || !gv->sdf_refined || !gv->albedo || !gv->weight || !gv->color)
        return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_set_grid: null grid or empty voxel list");
    if (gv->num_voxels > (1ll << 30)) return ctx_fail(c, I3D_ERR_CAPACITY, "i3d_set_grid: more than 2^30 voxels");
    CTX_HIP(c, hipSetDevice(c->device));
    const int N = (int)gv->num_voxels;
    hipStream_t st = c->stream;
    GridStaging g;     // staging copies of the caller's arrays (visit order)
    CTX_HIP(c, g.kxyz.alloc((size_t)3 * N)); CTX_HIP(c, g.sdf.alloc(N)); CTX_HIP(c, g.sdf_ref.alloc(N)); CTX_HIP(c, g.alb.alloc(N)); CTX_HIP(c, g.w.alloc(N)); CTX_HIP(c, g.rgb.alloc((size_t)3 * N));
    CTX_HIP(c, hipMemcpyAsync(g.kxyz.p, gv->keys, sizeof(int) * 3 * (size_t)N, hipMemcpyHostToDevice, st));
    CTX_HIP(c, hipMemcpyAsync(g.sdf.p, gv->sdf, sizeof(double) * (size_t)N, hipMemcpyHostToDevice, st));
    CTX_HIP(c, hipMemcpyAsync(g.sdf_ref.p, gv->sdf_refined, sizeof(double) * (size_t)N, hipMemcpyHostToDevice, st));
    CTX_HIP(c, hipMemcpyAsync(g.alb.p, gv->albedo, sizeof(double) * (size_t)N, hipMemcpyHostToDevice, st));
    CTX_HIP(c, hipMemcpyAsync(g.w.p, gv->weight, sizeof(float) * (size_t)N, hipMemcpyHostToDevice, st));
    CTX_HIP(c, hipMemcpyAsync(g.rgb.p, gv->color, (size_t)3 * N, hipMemcpyHostToDevice, st));
    return set_grid_device(c, N, gv->voxel_size, gv->truncation, g);
}

}  // extern "C"

namespace i3d {
int set_grid_device(i3d_context* c, int N, float voxel_size, float truncation, GridStaging& g) {
    c->N = N; c->voxel_size = voxel_size; c->truncation = truncation; c->have_grid = false; c->have_sh = false; c->have_subvolumes = false; c->assembled = false;
    c->slots = 0;      // row storage is sized by N: force a re-allocation for the new grid
    hipStream_t st = c->stream;
    DevBuf<int> perm, iota; DevBuf<unsigned long long> skeys, skeys2;
    CTX_HIP(c, perm.alloc(N)); CTX_HIP(c, iota.alloc(N)); CTX_HIP(c, skeys.alloc(N)); CTX_HIP(c, skeys2.alloc(N));
    // brick-Morton sort (device order), then permute every field into SoA planes
    launch_sort_keys(st, N, g.kxyz.p, skeys.p, iota.p);
    size_t tmp_bytes = 0;
    CTX_HIP(c, rocprim::radix_sort_pairs(nullptr, tmp_bytes, skeys.p, skeys2.p, iota.p, perm.p, (size_t)N, 0, 64, st));
    DevBuf<unsigned char> tmp; CTX_HIP(c, tmp.alloc(tmp_bytes));
    CTX_HIP(c, rocprim::radix_sort_pairs(tmp.p, tmp_bytes, skeys.p, skeys2.p, iota.p, perm.p, (size_t)N, 0, 64, st));
    // grid-sized buffers: release and re-allocate when the voxel count changes (level transitions shrink and grow the grid)
    auto fit = [&](auto& buf, size_t n) -> hipError_t { if (buf.n != n) buf.release(); return buf.alloc(n); };
    CTX_HIP(c, fit(c->cx, N)); CTX_HIP(c, fit(c->cy, N)); CTX_HIP(c, fit(c->cz, N)); CTX_HIP(c, fit(c->rank, N));
    CTX_HIP(c, fit(c->sdf0, N)); CTX_HIP(c, fit(c->x_sdf, N)); CTX_HIP(c, fit(c->x_alb, N)); CTX_HIP(c, fit(c->xc_sdf, N)); CTX_HIP(c, fit(c->xc_alb, N));
    CTX_HIP(c, fit(c->f_sdf, N)); CTX_HIP(c, fit(c->f_alb, N)); CTX_HIP(c, fit(c->weight, N)); CTX_HIP(c, fit(c->color, N));
    CTX_HIP(c, fit(c->flags, N)); CTX_HIP(c, fit(c->aidx, N)); CTX_HIP(c, fit(c->alist, N)); CTX_HIP(c, fit(c->aflag, N)); CTX_HIP(c, fit(c->ascan, N));
    CTX_HIP(c, fit(c->sh, (size_t)9 * N)); CTX_HIP(c, fit(c->nbr, (size_t)NUM_NBR * N));
    launch_permute_grid(st, N, perm.p, g.kxyz.p, g.sdf.p, g.sdf_ref.p, g.alb.p, g.w.p, g.rgb.p, c->cx.p, c->cy.p, c->cz.p, c->rank.p, c->sdf0.p,
                        c->x_sdf.p, c->x_alb.p, c->f_sdf.p, c->f_alb.p, c->weight.p, c->color.p);
    CTX_HIP(c, hipMemcpyAsync(c->xc_sdf.p, c->x_sdf.p, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice, st));
    CTX_HIP(c, hipMemcpyAsync(c->xc_alb.p, c->x_alb.p, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice, st));
    // device hash (2x load-factor headroom, power of two) + neighbour table; the hash stays resident for the level kernels
    unsigned int cap = 1; while (cap < (unsigned int)N * 2u) cap <<= 1;
    CTX_HIP(c, fit(c->hkeys, cap)); CTX_HIP(c, fit(c->hvals, cap)); c->hmask = cap - 1;
    CTX_HIP(c, hipMemsetAsync(c->hkeys.p, 0xff, sizeof(unsigned long long) * (size_t)cap, st));
    HashTable t{c->hkeys.p, c->hvals.p, cap - 1};
    launch_hash_build(st, N, c->cx.p, c->cy.p, c->cz.p, t);
    launch_nbr_build(st, N, c->cx.p, c->cy.p, c->cz.p, t, c->nbr.p);
    CTX_HIP(c, hipMemsetAsync(c->sh.p, 0, sizeof(float) * 9 * (size_t)N, st));
    size_t sb = 0;
    CTX_HIP(c, rocprim::exclusive_scan(nullptr, sb, c->aflag.p, c->ascan.p, 0, (size_t)N, rocprim::plus<int>(), st));
    c->scan_tmp.release(); CTX_HIP(c, c->scan_tmp.alloc(sb ? sb : 1)); c->scan_tmp_bytes = sb;
    CTX_HIP(c, hipStreamSynchronize(st));
    CTX_HIP(c, hipGetLastError());
    c->have_grid = true;
    return I3D_OK;
}
}  // namespace i3d

extern "C" {

int i3d_get_grid(i3d_context* c, double* sdf_refined, double* albedo) {
    if (!c || !c->have_grid) return ctx_fail(c, I3D_ERR_STATE, "i3d_get_grid: no grid");
    CTX_HIP(c, hipSetDevice(c->device));
    const int N = c->N;
    DevBuf<double> a, b; CTX_HIP(c, a.alloc(N)); CTX_HIP(c, b.alloc(N));
    launch_gather_visit(c->stream, N, c->rank.p, c->x_sdf.p, c->x_alb.p, a.p, b.p);
    if (sdf_refined) CTX_HIP(c, hipMemcpyAsync(sdf_refined, a.p, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, c->stream));
    if (albedo) CTX_HIP(c, hipMemcpyAsync(albedo, b.p, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, c->stream));
    CTX_HIP(c, hipStreamSynchronize(c->stream));
    return I3D_OK;
}

int i3d_update_grid(i3d_context* c, const double* sdf_refined, const double* albedo, const uint8_t* color) {
    if (!c || !c->have_grid) return ctx_fail(c, I3D_ERR_STATE, "i3d_update_grid: no grid");
    CTX_HIP(c, hipSetDevice(c->device));
    const int N = c->N;
    DevBuf<double> a, b; DevBuf<uint8_t> col;
    if (sdf_refined) { CTX_HIP(c, a.alloc(N)); CTX_HIP(c, hipMemcpyAsync(a.p, sdf_refined, sizeof(double) * (size_t)N, hipMemcpyHostToDevice, c->stream)); }
    if (albedo) { CTX_HIP(c, b.alloc(N)); CTX_HIP(c, hipMemcpyAsync(b.p, albedo, sizeof(double) * (size_t)N, hipMemcpyHostToDevice, c->stream)); }
    if (color) { CTX_HIP(c, col.alloc((size_t)3 * N)); CTX_HIP(c, hipMemcpyAsync(col.p, color, (size_t)3 * N, hipMemcpyHostToDevice, c->stream)); }
    launch_update_fields(c->stream, N, c->rank.p, a.p, b.p, col.p, c->x_sdf.p, c->x_alb.p, c->f_sdf.p, c->f_alb.p, c->color.p);
    CTX_HIP(c, hipStreamSynchronize(c->stream));
    c->assembled = false;
    return I3D_OK;
}

int i3d_set_frames(i3d_context* c, int32_t K, int32_t levels, const int32_t* widths, const int32_t* heights,
                   const float* const* lum, const float* const* depth, const uint8_t* const* bgr) {
    if (!c || K <= 0 || levels <= 0 || !widths || !heights || !lum || !depth) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_set_frames: bad arguments");
    CTX_HIP(c, hipSetDevice(c->device));
    c->have_frames = false; c->K = K; c->levels = levels; c->cull_level = -1;
    c->slots = 0; c->assembled = false;            // K sizes the camera blocks and solver vectors: force alloc_rows() to run again
    c->fw.assign(widths, widths + levels); c->fh.assign(heights, heights + levels);
    c->lum.clear(); c->depth.clear(); c->bgr.clear();
    c->lum.resize((size_t)K * levels); c->depth.resize((size_t)K * levels); c->bgr.resize((size_t)K * levels);
    for (int f = 0; f < K; ++f) for (int l = 0; l < levels; ++l) {
        const size_t k = (size_t)f * levels + l, px = (size_t)widths[l] * heights[l];
        if (!lum[k] || !depth[k]) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_set_frames: null image");
        CTX_HIP(c, c->lum[k].alloc(px + 4));       /* + 16 B: the 16-byte tap loads of the cost kernels may run past the last row of an image narrower than 4 pixels (build.hip) */ CTX_HIP(c, c->depth[k].alloc(px));
        CTX_HIP(c, hipMemcpyAsync(c->lum[k].p, lum[k], px * sizeof(float), hipMemcpyHostToDevice, c->stream));
        CTX_HIP(c, hipMemcpyAsync(c->depth[k].p, depth[k], px * sizeof(float), hipMemcpyHostToDevice, c->stream));
        if (bgr && bgr[k]) { CTX_HIP(c, c->bgr[k].alloc(px * 3)); CTX_HIP(c, hipMemcpyAsync(c->bgr[k].p, bgr[k], px * 3, hipMemcpyHostToDevice, c->stream)); }
    }
    CTX_HIP(c, c->d_frames.alloc(K)); CTX_HIP(c, c->d_frames_cand.alloc(K));
    CTX_HIP(c, hipStreamSynchronize(c->stream));
    c->have_frames = true;
    if ((int)c->poses.size() != 6 * K) { c->poses.assign((size_t)6 * K, 0.0); c->have_camera = false; }
    return I3D_OK;
}

int i3d_set_camera(i3d_context* c, const double* intr, const double* dist, const double* poses) {
    if (!c || !intr || !dist || !poses) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_set_camera: null pointer");
    if (!c->have_frames) return ctx_fail(c, I3D_ERR_STATE, "i3d_set_camera: set the keyframes first");
    std::memcpy(c->intr, intr, sizeof(double) * 4); std::memcpy(c->dist, dist, sizeof(double) * 5);
    c->poses.assign(poses, poses + (size_t)6 * c->K);
    c->have_camera = true;
    return I3D_OK;
}
int i3d_get_camera(i3d_context* c, double* intr, double* dist, double* poses) {
    if (!c || !c->have_camera) return ctx_fail(c, I3D_ERR_STATE, "i3d_get_camera: no camera");
    if (intr) std::memcpy(intr, c->intr, sizeof(double) * 4);
    if (dist) std::memcpy(dist, c->dist, sizeof(double) * 5);
    if (poses) std::memcpy(poses, c->poses.data(), sizeof(double) * 6 * (size_t)c->K);
    return I3D_OK;
}

int i3d_set_voxel_sh(i3d_context* c, const double* voxel_sh) {
    if (!c || !voxel_sh) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_set_voxel_sh: null pointer");
    if (!c->have_grid) return ctx_fail(c, I3D_ERR_STATE, "i3d_set_voxel_sh: no grid");
    CTX_HIP(c, hipSetDevice(c->device));
    DevBuf<double> tmp; CTX_HIP(c, tmp.alloc((size_t)9 * c->N));
    CTX_HIP(c, hipMemcpyAsync(tmp.p, voxel_sh, sizeof(double) * 9 * (size_t)c->N, hipMemcpyHostToDevice, c->stream));
    launch_scatter_sh(c->stream, c->N, c->rank.p, tmp.p, c->sh.p);
    CTX_HIP(c, hipStreamSynchronize(c->stream));
    c->have_sh = true;
    return I3D_OK;
}
int i3d_get_voxel_sh(i3d_context* c, double* out) {
    if (!c || !out) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_get_voxel_sh: null pointer");
    if (!c->have_grid) return ctx_fail(c, I3D_ERR_STATE, "i3d_get_voxel_sh: no grid");
    CTX_HIP(c, hipSetDevice(c->device));
    const int N = c->N;
    std::vector<float> sh((size_t)9 * N); std::vector<int> rank(N);
    CTX_HIP(c, hipMemcpy(sh.data(), c->sh.p, sizeof(float) * 9 * (size_t)N, hipMemcpyDeviceToHost));
    CTX_HIP(c, hipMemcpy(rank.data(), c->rank.p, sizeof(int) * (size_t)N, hipMemcpyDeviceToHost));
    for (int s = 0; s < N; ++s) for (int j = 0; j < 9; ++j) out[(size_t)rank[s] * 9 + j] = (double)sh[(size_t)j * N + s];
    return I3D_OK;
}

int i3d_timing_enable(i3d_context* c, int32_t on) { if (!c) return I3D_ERR_INVALID_ARGUMENT; timing_flush(c); c->timing.on = on != 0; return I3D_OK; }
int i3d_timing_select(i3d_context* c, uint32_t category_mask) { if (!c) return I3D_ERR_INVALID_ARGUMENT; timing_flush(c); c->timing.mask = category_mask; return I3D_OK; }
int i3d_timing_get(i3d_context* c, double* ms, int64_t* launches, int32_t reset) {
    if (!c) return I3D_ERR_INVALID_ARGUMENT;
    timing_flush(c);
    for (int i = 0; i < I3D_K_COUNT; ++i) { if (ms) ms[i] = c->timing.ms[i]; if (launches) launches[i] = c->timing.launches[i]; }
    if (reset) for (int i = 0; i < I3D_K_COUNT; ++i) { c->timing.ms[i] = 0; c->timing.launches[i] = 0; c->timing.each[i].clear(); }
    return I3D_OK;
}
int i3d_timing_get_work(i3d_context* c, double* ms, int64_t* launches) { return i3d_timing_get_work_ex(c, ms, launches, nullptr, nullptr); }
int i3d_timing_get_work_ex(i3d_context* c, double* ms, int64_t* launches, double* slow_ms, int64_t* slow_launches) {
    if (!c) return I3D_ERR_INVALID_ARGUMENT;
    timing_flush(c);
    for (int i = 0; i < I3D_K_COUNT; ++i) {
        // reference = the 90th percentile of the launch durations: launches queued behind the convergence flag return at once (< 25 % of it),
        // and a launch that straddles a hiccup of the device (seen once: 19.9 ms for a 0.33 ms kernel) is not the kernel's duration either (> 4x)
        std::vector<float> sorted(c->timing.each[i]); std::sort(sorted.begin(), sorted.end());
        const float ref = sorted.empty() ? 0.0f : sorted[(size_t)(0.9 * (double)(sorted.size() - 1))];
        double s = 0.0, slow = 0.0; int64_t n = 0, nslow = 0;
        const bool waits_for_peers = i == I3D_K_COMM;        // exchange launches mostly WAIT: their spread is the peers', there is no "hiccup" to cut off
        for (float v : c->timing.each[i]) {
            if (v < 0.25f * ref) continue;
            if (v > 4.0f * ref && !waits_for_peers) { slow += v; ++nslow; continue; }
            s += v; ++n;
        }
        if (ms) ms[i] = s; if (launches) launches[i] = n;
        if (slow_ms) slow_ms[i] = slow; if (slow_launches) slow_launches[i] = nslow;      // reported beside the average, never silently dropped
    }
    return I3D_OK;
}
const char* i3d_kernel_name(int32_t k) {
    static const char* names[I3D_K_COUNT] = {"classify", "observe", "build", "eg_pass", "gather", "cost", "vector", "sh", "eg_aux", "comm", "eg_mr2", "eg_mr3"};
    return (k >= 0 && k < I3D_K_COUNT) ? names[k] : "?";
}
int i3d_problem_sizes(i3d_context* c, int64_t out[6]) { if (!c || !out) return I3D_ERR_INVALID_ARGUMENT; for (int i = 0; i < 6; ++i) out[i] = c->last_sizes[i]; return I3D_OK; }

void i3d_optimizer_config_default(i3d_optimizer_config* cfg) {     // optimizer.h:69-79, intrinsic3d.h:67-84
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->iterations = 10; cfg->lm_steps = 50; cfg->lambda_g = 0.2; cfg->lambda_r0 = 20.0; cfg->lambda_r1 = 160.0;
    cfg->lambda_s0 = 10.0; cfg->lambda_s1 = 120.0; cfg->lambda_a = 0.1;
    cfg->occlusion_distance = 0.02f; cfg->num_observations = 5; cfg->pcg_fixed_iterations = -1;
}

}  // extern "C"
