// Host-side state of one device context: resident voxel grid, keyframes, camera model, row storage and solver vectors.
#pragma once
#include <string>
#include <vector>
#include <cstring>
#include <cmath>
#include "../device/kernels.hpp"
#include "comm.hpp"
#include "../../../include/intrinsic3d_hip.h"

namespace i3d {

template <class T>
struct DevBuf {
    T* p = nullptr; size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~DevBuf() { release(); }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    hipError_t alloc(size_t count) {          // grow-only
        if (count <= n && p) return hipSuccess;
        release();
        if (count == 0) return hipSuccess;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
};

struct Timing {
    bool on = false;
    unsigned mask = ~0u;                            // categories that get HIP events (an event pair per launch is not free: ~8 % with all of them on)
    double ms[I3D_K_COUNT] = {0};
    long long launches[I3D_K_COUNT] = {0};
    std::vector<float> each[I3D_K_COUNT];          // every launch duration (for the work-only average)
    struct Pending { hipEvent_t a, b; int cat; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
};

}  // namespace i3d

struct i3d_context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // ---- grid (device order = brick-Morton sorted) ----
    int N = 0; float voxel_size = 0, truncation = 0;
    i3d::DevBuf<int> cx, cy, cz, rank, nbr, aidx, alist, aflag, ascan;
    i3d::DevBuf<double> sdf0, x_sdf, x_alb, xc_sdf, xc_alb;
    i3d::DevBuf<float> f_sdf, f_alb, weight, sh;
    i3d::DevBuf<uchar4> color;
    i3d::DevBuf<uint8_t> flags;
    i3d::DevBuf<unsigned char> scan_tmp; size_t scan_tmp_bytes = 0;
    i3d::DevBuf<unsigned long long> hkeys; i3d::DevBuf<int> hvals; unsigned int hmask = 0;     // device hash of the resident grid (kept for the level kernels)
    bool have_grid = false, have_sh = false;
    // the lighting estimate behind `sh` (LightingSVSH::subvolumes() / shCoeffs()): packed subvolume indices (ascending), nine coefficients each, the subvolume size —
    // what the "shading" colour modes of the mesh export interpolate at every voxel (SDFVisualization::applyColorShading)
    std::vector<unsigned long long> sv_keys; std::vector<double> sv_sh; float sv_size = 0.0f; bool have_subvolumes = false;

    // ---- keyframes ----
    int K = 0, levels = 0;
    std::vector<int> fw, fh;                        // per level
    std::vector<i3d::DevBuf<float>> lum, depth;     // [K*levels]
    std::vector<i3d::DevBuf<uint8_t>> bgr;
    i3d::DevBuf<i3d::FrameConst> d_frames, d_frames_cand;
    bool have_frames = false;

    // ---- camera (fp64 master copy on the host) ----
    double intr[4] = {0, 0, 0, 0}, dist[5] = {0, 0, 0, 0, 0};
    std::vector<double> poses;
    bool have_camera = false;

    // ---- rows (work-list space) ----
    int Acap = 0, slots = 0, A = 0; long long n_active = 0;
    // sharding: owned range / compute list of this rank (see common.hpp)
    i3d::Comm* comm = nullptr; int chunk = 1, own0 = 0, own1 = 0, nC = 0;
    i3d::DevBuf<int> clist, cflag, cscan;
    // halo exchange plan of the current work list (shard_kernels.hip) and the foreign tiles with ghost entries
    i3d::HaloPlan halo; i3d::DevBuf<unsigned long long> need_mask, halo_items, halo_sorted; i3d::DevBuf<int> halo_count, halo_send_idx, halo_recv_idx, halo_send_peer, halo_recv_peer, halo_offs, tile_flag, ghost_tiles;
    i3d::DevBuf<float> halo_send_buf, halo_recv_buf; i3d::DevBuf<unsigned char> halo_temp; int n_ghost_tiles = 0, slice = 0;
    i3d::DevBuf<int> obs_frame, anbr; i3d::DevBuf<float> obs_w, ea_w, C, treg;
    i3d::DevBuf<float4> rows; i3d::DevBuf<float2> row_wr;
    i3d::DevBuf<uint8_t> aflags, nrows, regflags, ea_free; i3d::DevBuf<int> gmax;
    // culling in front of the observation pass (cull_kernels.hip): 8x8-block depth ranges of every keyframe at level cull_level (-1: not built), bounding spheres and
    // keyframe masks of the 64-entry groups of the compute list
    i3d::DevBuf<float2> cull_blocks; i3d::DevBuf<float4> cull_bounds; i3d::DevBuf<unsigned> cull_mask; int cull_level = -1; bool cull_on = false;
    // tiled operator pass (tile_pass.hip): plan of the current work list
    double cost_at_build = 0.0;       // 0.5 sum_t type_w[t] sum(w r^2) at the point the rows were built at (assemble)
    i3d::DevBuf<float> C2, treg2, gc_part;      // the second set of staging planes + the camera rows of the one-stream gradient / column-norm pass (gradcol.hip)
    i3d::DevBuf<float> aux_part;      // one float row of camera totals per workgroup of the gradient / column-norm passes (summed in a fixed order)
    i3d::DevBuf<unsigned> tp_lnbr; i3d::DevBuf<int> tp_halo_idx, tp_halo_cnt, tp_iota, tp_ext_e, tp_ext_pos, tp_ext_off, tp_overflow; i3d::DevBuf<float> tp_qh, tp_eaw, cam_part;
    i3d::DevBuf<unsigned char> tp_temp; bool tile_ok = false;
    i3d::DevBuf<unsigned short> tp_hp_off, tp_hp_src; bool halo_pull = false;      // halo pull lists of the plan (tile_pass.hip k_tile_pull_plan): I3D_HALO_PULL=1 and the bit-reproducible mode
    bool ladder_lists = false;      // the lists are built for the multi-system operator pass of the ladder (tile_pass_mr.hip) although the single-system pass pushes its halo
    // ---- the damping ladder (solver.cpp lm_solve / pcg_solve_ladder): per-system slabs of the PCG vectors and partial sums ----
    bool mr1_serial = false;        // I3D_EGT_MR1=1: the serial loop's operator pass is k_eg_tile_mr<1> (A/B runs, the control of the ladder tests)
    int ladder_max = 1;             // I3D_LADDER (read at every assemble): attempts solved together, 1 = the serial loop
    int ladder_hint = 0, ladder_hint_prev = 0;      // LM attempts of the last two outer iterations (the first batch speculates as deep as the larger)
    i3d::LadVec lad{};              // strides of the slabs below
    i3d::DevBuf<float> lad_vec;     // [LADDER_MAX][6][lad.vec]: x, r, p, z, u, qacc of every system
    i3d::DevBuf<float> lad_qh, lad_cam, lad_mblk, lad_tail;
    i3d::DevBuf<double> lad_part;   // [LADDER_MAX][lad.part]: step partials [4 * 1024] | p.q partials [1024] | D^2 p^2 partials [1024]
    i3d::DevBuf<double> lad_red;    // sharded ladder: what a pass all-reduces — [LADDER_MAX][4] slice sums | [LADDER_MAX][6K + 10, padded] camera block + p.q
    i3d::DevBuf<i3d::PcgState> lad_st;      // [LADDER_MAX][2]
    long long lad_batches = 0, lad_streams = 0, lad_system_passes = 0, lad_resyncs = 0, lad_wasted = 0;      // counters (i3d_debug_ladder_stats)
    double t_add_end = 0.0;         // host clock at the end of the residual collection of the current outer iteration (time_add | time_build)
    bool deterministic = false;     // read at every assemble: bit-reproducible operator pass (default on one rank, I3D_DETERMINISTIC=0 / =1 override)
    int tile_T = 0;                 // geometry of the current plan (0 = the default, 1024); single rank: 512 when a 1024-entry tile's halo does not fit; sharded: 512 first, then 1024
    int plan_T() const { return tile_T > 0 ? tile_T : i3d::tile_plan_T(); }
    i3d::TilePlan tile_plan() const {
        const int T = plan_T();
        const bool sh = comm && (comm->world > 1 || comm->force);
        const int t0 = sh ? own0 / T : 0, t1 = sh ? (own1 + T - 1) / T : i3d::tile_plan_tiles_of(A, T);
        return i3d::TilePlan{tp_lnbr.p, tp_eaw.p, tp_halo_idx.p, tp_halo_cnt.p, tp_iota.p, tp_ext_e.p, tp_ext_pos.p, tp_qh.p, tp_ext_off.p, tp_overflow.p, T, i3d::tile_plan_hmax_of(T), t0, t1 > t0 ? t1 - t0 : 0,
                             ghost_tiles.p, sh ? n_ghost_tiles : 0, deterministic ? 1 : 0, (halo_pull || ladder_lists) ? tp_hp_off.p : nullptr, (halo_pull || ladder_lists) ? tp_hp_src.p : nullptr, halo_pull ? 1 : 0};
    }

    // ---- solver vectors (length NP = 2N + 6K + 9) ----
    i3d::DevBuf<float> v_mask, v_c, v_S, v_cm, v_D2, v_Minv, v_b, v_x, v_r, v_p, v_z, v_q, v_u, v_acc, v_tmp, v_qacc;
    i3d::DevBuf<float> Minv_blocks;
    i3d::DevBuf<double> d_shared, d_blocks, d_scal, d_xshared, d_xcshared;
    i3d::DevBuf<i3d::PcgState> d_pcg, d_pcg2; i3d::DevBuf<double> d_partials; i3d::PcgState* h_pcg = nullptr; hipEvent_t pcg_ev[2] = {nullptr, nullptr};
    int* h_flags = nullptr; int* d_flags = nullptr; int pcg_seq = 1024;      // pinned (seq, done) ring written by k_pcg_tail_a, polled by the host
    // the trust-region loop on the device (lm_kernels.hip): its state, one record per attempt in mapped host memory, the camera blocks of J^T W J it damps
    i3d::DevBuf<i3d::LmState> d_lm; i3d::LmRecord* h_lmrec = nullptr; i3d::LmRecord* d_lmrec = nullptr; int lm_seq = 1;
    i3d::DevBuf<double> d_cam_c, d_cam_H;
    hipEvent_t ev_asm[2] = {nullptr, nullptr};      // start of an outer iteration's assembly | end of its residual collection (time_add / time_build without a synchronisation)
    long long n_syncs = 0;                          // hipStreamSynchronize calls of the solver path (i3d_debug_sync_count)
    double* h_pinned = nullptr; size_t h_pinned_n = 0;

    i3d::OptParams last_params; bool assembled = false;
    long long last_sizes[6] = {0, 0, 0, 0, 0, 0};
    i3d::Timing timing;

    // views
    i3d::GridView grid_view() const;
    i3d::RowView row_view() const;
};

namespace i3d {

// helpers implemented in context.cpp
int ctx_fail(i3d_context* c, int code, const std::string& msg);
int ctx_hip(i3d_context* c, hipError_t e, const char* what);
int ctx_launch_check(i3d_context* c);      // hipGetLastError + the latched launch-configuration error (common.hpp: set_dynamic_lds) -> I3D_ERR_HIP / I3D_ERR_CAPACITY
#define CTX_HIP(c, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return i3d::ctx_hip((c), _e, #expr); } while (0)
void build_frame_consts(const i3d_context* c, int level, const double* poses, std::vector<FrameConst>& out);
int ensure_pinned(i3d_context* c, size_t n);
bool timing_begin(i3d_context* c, int cat);
void timing_end(i3d_context* c);
void timing_flush(i3d_context* c);
struct TimedScope { i3d_context* c; bool active; TimedScope(i3d_context* c_, int cat) : c(c_), active(timing_begin(c_, cat)) {} ~TimedScope() { if (active) timing_end(c); } };

// context.cpp — (re)build the resident grid from device arrays in visit order
struct GridStaging { DevBuf<int> kxyz; DevBuf<double> sdf, sdf_ref, alb; DevBuf<float> w; DevBuf<uint8_t> rgb; };
int set_grid_device(i3d_context* c, int N, float voxel_size, float truncation, GridStaging& st);

// levels.cpp
int recompute_colors(i3d_context* c, float occlusion_distance, int num_observations);
int clear_outside_thin_shell(i3d_context* c, double thres_shell, int64_t* new_count);
int upsample_grid(i3d_context* c, int64_t* new_count);

// solver.cpp
int assemble(i3d_context* c, const i3d_optimizer_config& cfg, int iteration, OptParams& p, i3d_iteration_stats* st);
int optimize(i3d_context* c, const i3d_optimizer_config& cfg, i3d_iteration_stats* stats);
int normal_eq_debug(i3d_context* c, double* gradient, double* jtj_diag, double* cost);
int jtj_apply_debug(i3d_context* c, const double* x, double* y);

// lighting.cpp
int estimate_sh(i3d_context* c, float subvolume_size, double lambda_reg, double thres_shell, int* num_subvolumes, double* sh,
                int32_t* sub_index, int cap, i3d_sh_stats* stats);

}  // namespace i3d
