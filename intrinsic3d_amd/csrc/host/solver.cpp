// Host control of one Optimizer::optimize call on the device:
//   assemble()  = Optimizer::addVoxelResiduals over all voxels + NLSSolver::buildProblem (optimizer.cpp:119-165, nls_solver.cpp:190-276)
//   lm_solve()  = NLSSolver::solve (nls_solver.cpp:296-367): Ceres 2.1.0 TrustRegionMinimizer + LevenbergMarquardtStrategy + CGNR with
//                 block-Jacobi preconditioning [Ceres is un-vendored; semantics per SURVEY.md Appendix B], stopping after the first
//                 successful step (SuccessfulStepCallback, nls_solver.cpp:279-293).
// Solver vectors are fp32 in work-list space; every reduction (dots, cost, weight sums, camera blocks) accumulates in fp64; the
// unknowns keep an fp64 master copy.  The PCG scalars stay on the device (PcgState); the host polls them one iteration behind.
#include "context.hpp"
#include <rocprim/rocprim.hpp>
#include <chrono>
#include <limits>
#include <algorithm>

namespace i3d {

#ifdef I3D_MR_PHASES
void mr_phase_report_now();     // tile_pass_mr.hip, variant build only
#endif
#ifdef I3D_MR_BLOCKTIME
void mr_blocktime_report_now(); // tile_pass_mr.hip, variant build only
#endif

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static OptParams make_params(const i3d_context* c, const i3d_optimizer_config& cfg, const double* intr, const double* dist) {
    OptParams p; std::memset(&p, 0, sizeof(p));
    p.thres_shell = cfg.thres_shell; p.lambda_a = cfg.lambda_a; p.K = c->K; p.level = cfg.rgbd_level;
    p.pyr_scale = 1.0 / std::pow(2.0, cfg.rgbd_level);                      // cost.h:146-150
    p.occlusion = cfg.occlusion_distance;
    p.use_er = (cfg.lambda_r0 > 0.0 && cfg.lambda_r1 > 0.0); p.use_es = (cfg.lambda_s0 > 0.0 && cfg.lambda_s1 > 0.0); p.use_ea = cfg.lambda_a > 0.0;
    for (int i = 0; i < 4; ++i) { p.intr[i] = intr[i]; p.cam_f[i] = (float)(intr[i] * p.pyr_scale); }   // optimizer.cpp:125, camera.cpp:95-98
    bool dz = true;
    for (int i = 0; i < 5; ++i) { p.dist[i] = dist[i]; p.dist_f[i] = (float)dist[i]; if (std::fabs(p.dist_f[i]) > 1e-5f) dz = false; }
    p.dist_zero = dz ? 1 : 0;
    p.w = c->fw[cfg.rgbd_level]; p.h = c->fh[cfg.rgbd_level];
    p.fix_poses = cfg.fix_poses; p.fix_intr = cfg.fix_intrinsics; p.fix_dist = cfg.fix_distortion; p.fix_sdf = cfg.fix_sdf;
    return p;
}

static double varying_lambda(int it, int n, double l0, double l1) {        // cost.h:130-143
    if (n <= 1) return l0;
    return l0 + ((l1 - l0) / (double)(n - 1)) * (double)it;
}

constexpr int HALO_CAP = 1 << 20;      // (direction, peer, entry) items of a rank's halo plan
constexpr int AUX_PART_ROWS = 320;     // >= the workgroups of a gradient / column-norm pass (one per CU)
static size_t aux_part_stride(int K) { return ((size_t)21 * K + 34 + 3) & ~(size_t)3; }
constexpr int GC_PART_ROWS = 1024;     // >= the workgroups of the one-stream gradient + column-norm pass (two per CU)
static size_t gc_col_off(int K) { return ((size_t)6 * K + 9 + 3) & ~(size_t)3; }
static size_t gc_part_stride(int K) { return gc_col_off(K) + aux_part_stride(K); }
constexpr int PCG_SEQ_STRIDE = 1024;   // pass numbers a PCG solve may use (<= 521 passes): the numbering of the next solve does not depend on how many passes a host queued
constexpr int LM_REC_SLOTS = 64;       // LmRecord ring: the initial tests + one record per LM attempt (lm_steps <= LM_REC_SLOTS - 2)

// every stream synchronisation of the solver path goes through here (counted: i3d_debug_counters; the review bar is <= 8 per Gauss-Newton iteration)
static hipError_t sync_stream(i3d_context* c) { ++c->n_syncs; return hipStreamSynchronize(c->stream); }

static int alloc_rows(i3d_context* c, int slots) {
    const size_t Acap = (size_t)c->N;
    c->Acap = (int)Acap; c->slots = slots;
    CTX_HIP(c, c->obs_frame.alloc(Acap * slots)); CTX_HIP(c, c->obs_w.alloc(Acap * slots));
    { const size_t nrow = ((Acap + 63) / 64) * 64 * (size_t)slots;
      CTX_HIP(c, c->rows.alloc(nrow / 64 * ROW_BLOCK_F4)); CTX_HIP(c, c->row_wr.alloc(nrow)); }
    CTX_HIP(c, c->aflags.alloc(Acap)); CTX_HIP(c, c->nrows.alloc(Acap)); CTX_HIP(c, c->gmax.alloc(Acap / 64 + 2)); CTX_HIP(c, c->anbr.alloc(Acap * NUM_NBR));
    CTX_HIP(c, c->cull_bounds.alloc(Acap / 64 + 2)); CTX_HIP(c, c->cull_mask.alloc((Acap / 64 + 2) * (size_t)((c->K + 31) / 32)));
    CTX_HIP(c, c->regflags.alloc(Acap)); CTX_HIP(c, c->ea_free.alloc(Acap)); CTX_HIP(c, c->ea_w.alloc(Acap * 6));
    CTX_HIP(c, c->C.alloc(Acap * P_VOX)); CTX_HIP(c, c->treg.alloc(Acap * 8)); CTX_HIP(c, c->aux_part.alloc((size_t)AUX_PART_ROWS * aux_part_stride(c->K)));
    CTX_HIP(c, c->C2.alloc(Acap * P_VOX)); CTX_HIP(c, c->treg2.alloc(Acap * 8)); CTX_HIP(c, c->gc_part.alloc((size_t)GC_PART_ROWS * gc_part_stride(c->K)));
    { // sized for BOTH tile geometries: the 512-entry one needs the most slots on large grids (3 per entry; 1024: 2), but a grid of <= 512 entries is ONE
      // 1024-entry tile with 2048 halo slots against one 512-entry tile with 1536
      const size_t nt = (size_t)tile_plan_tiles_of((int)Acap, 512);
      const size_t nh = std::max(nt * (size_t)tile_plan_hmax_of(512), (size_t)tile_plan_tiles_of((int)Acap, 1024) * (size_t)tile_plan_hmax_of(1024));
      CTX_HIP(c, c->tp_lnbr.alloc(Acap * LNBR_WORDS)); CTX_HIP(c, c->tp_eaw.alloc(Acap * 6)); CTX_HIP(c, c->tp_halo_idx.alloc(nh)); CTX_HIP(c, c->tp_halo_cnt.alloc(nt)); CTX_HIP(c, c->tp_iota.alloc(nh));
      CTX_HIP(c, c->tp_ext_e.alloc(nh)); CTX_HIP(c, c->tp_ext_pos.alloc(nh)); CTX_HIP(c, c->tp_qh.alloc(2 * nh)); CTX_HIP(c, c->tp_overflow.alloc(1));
      CTX_HIP(c, c->tp_ext_off.alloc(Acap + (size_t)SHARD_ALIGN * ((c->comm ? c->comm->world : 1) + 1) + 8));      // chunk + 1 offsets (chunk >= A, a multiple of the slice alignment)
      CTX_HIP(c, c->cam_part.alloc((size_t)2048 * (((size_t)6 * c->K + 9 + 3) & ~(size_t)3)));                     // one float row of the camera block per operator workgroup
      CTX_HIP(c, c->tp_temp.alloc(tile_plan_temp_bytes((int)nt))); }
    const int world_a = c->comm ? c->comm->world : 1;
    const size_t NP = 2 * ((size_t)c->N + (size_t)SHARD_ALIGN * (world_a + 1)) + 6 * (size_t)c->K + 9;      // chunk = world * slice >= A, slice a multiple of SHARD_ALIGN
    for (DevBuf<float>* v : {&c->v_mask, &c->v_c, &c->v_S, &c->v_cm, &c->v_D2, &c->v_Minv, &c->v_b, &c->v_x, &c->v_r, &c->v_p, &c->v_z, &c->v_q, &c->v_u, &c->v_acc, &c->v_tmp, &c->v_qacc})
        { const bool fresh = v->n < NP || !v->p; CTX_HIP(c, v->alloc(NP)); if (fresh) CTX_HIP(c, hipMemsetAsync(v->p, 0, sizeof(float) * v->n, c->stream)); }   // padding entries stay finite (on the library's stream: it is non-blocking, a null-stream memset is not ordered against it)
    CTX_HIP(c, c->Minv_blocks.alloc((size_t)36 * c->K + 16 + 25));
    CTX_HIP(c, c->d_shared.alloc((size_t)6 * c->K + 9 + 1)); CTX_HIP(c, c->d_blocks.alloc((size_t)21 * c->K + 25));      // +1: p.q partial rides with the camera block
    CTX_HIP(c, c->clist.alloc(Acap)); CTX_HIP(c, c->cflag.alloc(Acap)); CTX_HIP(c, c->cscan.alloc(Acap));
    if (c->comm) {      // sharding plan storage (small: the rim of a rank is a few percent of what it owns)
        const size_t cap = HALO_CAP;
        CTX_HIP(c, c->need_mask.alloc(Acap)); CTX_HIP(c, c->halo_items.alloc(cap)); CTX_HIP(c, c->halo_sorted.alloc(cap)); CTX_HIP(c, c->halo_count.alloc(1));
        CTX_HIP(c, c->halo_send_idx.alloc(cap)); CTX_HIP(c, c->halo_recv_idx.alloc(cap)); CTX_HIP(c, c->halo_send_peer.alloc(cap)); CTX_HIP(c, c->halo_recv_peer.alloc(cap)); CTX_HIP(c, c->halo_offs.alloc(2 * P2P_MAX_RANKS)); CTX_HIP(c, c->halo_send_buf.alloc((size_t)HALO_MULTI_MAX * (2 * cap))); CTX_HIP(c, c->halo_recv_buf.alloc((size_t)HALO_MULTI_MAX * (2 * cap)));      /* a ladder batch pushes HALO_MULTI_MAX values per rim entry in one message */
        CTX_HIP(c, c->halo_temp.alloc(halo_sort_temp_bytes((int)cap)));
        const size_t nt = (size_t)tile_plan_tiles_of((int)Acap, 512) + 1; CTX_HIP(c, c->tile_flag.alloc(nt)); CTX_HIP(c, c->ghost_tiles.alloc(nt));
    }
    CTX_HIP(c, c->d_scal.alloc(32)); CTX_HIP(c, c->d_xshared.alloc((size_t)6 * c->K + 9)); CTX_HIP(c, c->d_xcshared.alloc((size_t)6 * c->K + 9));
    CTX_HIP(c, c->d_pcg.alloc(1)); CTX_HIP(c, c->d_pcg2.alloc(2));
    CTX_HIP(c, c->d_partials.alloc((Acap / 256 + 2048) * 9));       // per-workgroup partial sums of the fp64 reductions
    if (!c->h_pcg) CTX_HIP(c, hipHostMalloc((void**)&c->h_pcg, 2 * sizeof(PcgState), hipHostMallocDefault));
    for (auto& e : c->pcg_ev) if (!e) CTX_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (!c->h_flags) { CTX_HIP(c, hipHostMalloc((void**)&c->h_flags, 32 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));      // (pass, done) rings: 4 ints per system of a ladder batch
                       for (int i = 0; i < 32; ++i) c->h_flags[i] = -1;
                       CTX_HIP(c, hipHostGetDevicePointer((void**)&c->d_flags, c->h_flags, 0)); }
    CTX_HIP(c, c->d_lm.alloc(1)); CTX_HIP(c, c->d_cam_c.alloc((size_t)6 * c->K + 9)); CTX_HIP(c, c->d_cam_H.alloc((size_t)21 * c->K + 25));
    if (!c->h_lmrec) { CTX_HIP(c, hipHostMalloc((void**)&c->h_lmrec, LM_REC_SLOTS * sizeof(LmRecord), hipHostMallocMapped | hipHostMallocCoherent));
                       std::memset(c->h_lmrec, 0, LM_REC_SLOTS * sizeof(LmRecord));
                       CTX_HIP(c, hipHostGetDevicePointer((void**)&c->d_lmrec, c->h_lmrec, 0)); }
    for (auto& e : c->ev_asm) if (!e) CTX_HIP(c, hipEventCreate(&e));
    return ensure_pinned(c, 64 + (size_t)27 * c->K + 64);
}

// vector layout (common.hpp): [sdf chunk | albedo chunk | camera tail]; a rank's slice = the same segment of both parts
struct Layout { int A, K, NS, chunk, world, rank, slice; Seg2 own; size_t tail_off, NP; };
static Layout layout_of(const i3d_context* c) {
    Layout L; L.A = c->A; L.K = c->K; L.NS = 6 * c->K + 9; L.chunk = c->chunk; L.world = c->comm ? c->comm->world : 1; L.rank = c->comm ? c->comm->rank : 0;
    L.slice = L.chunk / L.world;
    L.own = Seg2{(size_t)L.rank * L.slice, (size_t)L.chunk + (size_t)L.rank * L.slice, L.slice};
    L.tail_off = 2 * (size_t)L.chunk; L.NP = L.tail_off + L.NS;
    return L;
}
static bool sharded(const i3d_context* c) { return c->comm && (c->comm->world > 1 || c->comm->force); }
// Where the cost the trust-region loop starts from comes from (it is the cost at the point the rows were built at; up to round 4: a residual-only pass k_build<false>):
//   one rank           : the gradient pass (gradcol.hip) adds up 0.5 w r^2 of the rows it streams anyway;
//   sharded / I3D_GRADCOL=0 : k_weight_sums adds the per-type sums while it walks the row weights (assemble);
//   I3D_COST0=0        : the residual-only pass.
static bool env_off(const char* name) { const char* e = std::getenv(name); return e && e[0] == '0'; }
static bool one_stream_gradcol(const i3d_context* c) { return !sharded(c) && !env_off("I3D_GRADCOL"); }
static bool cost_from_sums(const i3d_context* c) { return !env_off("I3D_COST0") && !one_stream_gradcol(c); }

static int allreduce(i3d_context* c, double* dev, size_t n) {
    if (!sharded(c)) return I3D_OK;
    TimedScope t(c, I3D_K_COMM);
    return c->comm->allreduce_sum(dev, n, c->stream) ? ctx_fail(c, I3D_ERR_COMM, "all-reduce failed") : I3D_OK;
}
static int allgather(i3d_context* c, float* vec) {       // every rank contributes its two segments of a solver vector
    if (!sharded(c)) return I3D_OK;
    const size_t slice = (size_t)(c->chunk / c->comm->world);
    TimedScope t(c, I3D_K_COMM);
    if (c->comm->allgather(vec, slice, c->stream) || c->comm->allgather(vec + c->chunk, slice, c->stream)) return ctx_fail(c, I3D_ERR_COMM, "all-gather failed");
    return I3D_OK;
}
static int push_halo(i3d_context* c, float* vec) {       // the rim of the operator input (common.hpp: sharding)
    if (!sharded(c)) return I3D_OK;
    TimedScope t(c, I3D_K_COMM);
    return c->comm->push_halo(vec, c->halo, c->stream) ? ctx_fail(c, I3D_ERR_COMM, "halo exchange failed") : I3D_OK;
}

// the sharding plan of this outer iteration: compute list, halo lists, ghost tiles — derived from the replicated work list, no communication
static int shard_plan(i3d_context* c) {
    hipStream_t s = c->stream;
    const int world = c->comm->world, rank = c->comm->rank;
    if (world > 64) return ctx_fail(c, I3D_ERR_CAPACITY, "sharding: more than 64 ranks");
    shard_range(c->A, world, rank, c->chunk, c->own0, c->own1); c->nC = 0; c->slice = c->chunk / world;
    RowView r0 = c->row_view();
    TimedScope t(c, I3D_K_CLASSIFY);
    launch_mark_compute(s, r0, c->cflag.p);
    CTX_HIP(c, rocprim::exclusive_scan(c->scan_tmp.p, c->scan_tmp_bytes, c->cflag.p, c->cscan.p, 0, (size_t)c->A, rocprim::plus<int>(), s));
    launch_compact_list(s, c->A, c->cflag.p, c->cscan.p, c->clist.p);
    const int T = c->plan_T(), ntiles = tile_plan_tiles_of(c->A, T);
    CTX_HIP(c, hipMemsetAsync(c->need_mask.p, 0, sizeof(unsigned long long) * (size_t)c->A, s));
    CTX_HIP(c, hipMemsetAsync(c->halo_count.p, 0, sizeof(int), s));
    CTX_HIP(c, hipMemsetAsync(c->tile_flag.p, 0, sizeof(int) * (size_t)(ntiles + 1), s));
    launch_need_mask(s, r0, c->slice, c->need_mask.p);
    launch_halo_items(s, c->A, c->slice, rank, c->need_mask.p, c->halo_items.p, c->halo_count.p, HALO_CAP);
    launch_tile_flags(s, c->A, T, c->cflag.p, c->tile_flag.p);
    int tl[2] = {0, 0}, nitems = 0;
    if (c->A > 0) { CTX_HIP(c, hipMemcpyAsync(&tl[0], c->cscan.p + (c->A - 1), sizeof(int), hipMemcpyDeviceToHost, s));
                    CTX_HIP(c, hipMemcpyAsync(&tl[1], c->cflag.p + (c->A - 1), sizeof(int), hipMemcpyDeviceToHost, s)); }
    CTX_HIP(c, hipMemcpyAsync(&nitems, c->halo_count.p, sizeof(int), hipMemcpyDeviceToHost, s));
    std::vector<int> tf((size_t)ntiles + 1, 0);
    CTX_HIP(c, hipMemcpyAsync(tf.data(), c->tile_flag.p, sizeof(int) * (size_t)ntiles, hipMemcpyDeviceToHost, s));
    CTX_HIP(c, sync_stream(c));
    c->nC = tl[0] + tl[1];
    if (nitems > HALO_CAP) return ctx_fail(c, I3D_ERR_CAPACITY, "sharding: the halo plan does not fit its buffers");
    // ghost tiles: foreign tiles that hold compute-list entries
    std::vector<int> ghosts; const int t0 = c->own0 / T, t1 = (c->own1 + T - 1) / T;
    for (int i = 0; i < ntiles; ++i) if (tf[i] && (i < t0 || i >= t1)) ghosts.push_back(i);
    c->n_ghost_tiles = (int)ghosts.size();
    if (!ghosts.empty()) CTX_HIP(c, hipMemcpyAsync(c->ghost_tiles.p, ghosts.data(), sizeof(int) * ghosts.size(), hipMemcpyHostToDevice, s));
    // halo lists: items sorted by (direction, peer, entry) -> per-peer runs
    HaloPlan& h = c->halo; h.world = world; h.chunk = c->chunk;
    h.send_cnt.assign(world, 0); h.send_off.assign(world, 0); h.recv_cnt.assign(world, 0); h.recv_off.assign(world, 0); h.n_send = h.n_recv = 0;
    std::vector<unsigned long long> items((size_t)nitems);
    if (nitems > 0) {
        CTX_HIP(c, launch_halo_sort(s, c->halo_temp.p, c->halo_temp.n, c->halo_items.p, c->halo_sorted.p, nitems));
        CTX_HIP(c, hipMemcpyAsync(items.data(), c->halo_sorted.p, sizeof(unsigned long long) * (size_t)nitems, hipMemcpyDeviceToHost, s));
        CTX_HIP(c, sync_stream(c));
    }
    std::vector<int> sidx, ridx, speer, rpeer;       // + the peer of every item: the flat lists the in-kernel exchanges of the three-launch pass walk (RimLists)
    for (unsigned long long it : items) {
        const int dir = (int)(it >> 40) & 1, peer = (int)((it >> 32) & 0xFF), e = (int)(it & 0xFFFFFFFFull);
        if (dir == 0) { if (h.send_cnt[peer]++ == 0) h.send_off[peer] = (int)sidx.size(); sidx.push_back(e); speer.push_back(peer); }
        else          { if (h.recv_cnt[peer]++ == 0) h.recv_off[peer] = (int)ridx.size(); ridx.push_back(e); rpeer.push_back(peer); }
    }
    h.n_send = (int)sidx.size(); h.n_recv = (int)ridx.size();
    if (h.n_send) { CTX_HIP(c, hipMemcpyAsync(c->halo_send_idx.p, sidx.data(), sizeof(int) * sidx.size(), hipMemcpyHostToDevice, s));
                    CTX_HIP(c, hipMemcpyAsync(c->halo_send_peer.p, speer.data(), sizeof(int) * speer.size(), hipMemcpyHostToDevice, s)); }
    if (h.n_recv) { CTX_HIP(c, hipMemcpyAsync(c->halo_recv_idx.p, ridx.data(), sizeof(int) * ridx.size(), hipMemcpyHostToDevice, s));
                    CTX_HIP(c, hipMemcpyAsync(c->halo_recv_peer.p, rpeer.data(), sizeof(int) * rpeer.size(), hipMemcpyHostToDevice, s)); }
    int offs[2 * P2P_MAX_RANKS] = {0};
    for (int k = 0; k < world; ++k) { offs[k] = h.send_off[k]; offs[P2P_MAX_RANKS + k] = h.recv_off[k]; }
    CTX_HIP(c, hipMemcpyAsync(c->halo_offs.p, offs, sizeof(offs), hipMemcpyHostToDevice, s));
    CTX_HIP(c, sync_stream(c));                  // (the host vectors above go out of scope)
    h.d_send_idx = c->halo_send_idx.p; h.d_recv_idx = c->halo_recv_idx.p; h.d_send_buf = c->halo_send_buf.p; h.d_recv_buf = c->halo_recv_buf.p;
    { const int prc = c->comm->plan_changed(h, s);
      if (prc == 2) return ctx_fail(c, I3D_ERR_CAPACITY, "sharding: the rim of a rank pair exceeds the peer-to-peer mailbox");
      if (prc) return ctx_fail(c, I3D_ERR_COMM, "sharding: installing the halo lists of the peer-to-peer transport failed on a rank (HIP / copy error)"); }
    return I3D_OK;
}

// read `n` doubles from device memory (after everything queued on the stream)
static int read_doubles(i3d_context* c, const double* dptr, size_t n, double* out) {
    CTX_HIP(c, hipMemcpyAsync(c->h_pinned, dptr, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CTX_HIP(c, sync_stream(c));
    std::memcpy(out, c->h_pinned, n * sizeof(double));
    return I3D_OK;
}

int assemble(i3d_context* c, const i3d_optimizer_config& cfg, int iteration, OptParams& p, i3d_iteration_stats* st) {
    if (!c->have_grid || !c->have_frames || !c->have_camera) return ctx_fail(c, I3D_ERR_STATE, "optimize: grid, keyframes and camera must be set");
    if (cfg.rgbd_level < 0 || cfg.rgbd_level >= c->levels) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "optimize: rgbd_level outside the uploaded pyramid");
    int slots = (cfg.num_observations <= 0 || cfg.num_observations >= c->K) ? c->K : cfg.num_observations;
    if (slots > MAX_SLOTS) return ctx_fail(c, I3D_ERR_CAPACITY, "optimize: more than 8 observations per voxel requested");
    if (c->K > 2000) return ctx_fail(c, I3D_ERR_CAPACITY, "optimize: more than 2000 keyframes (the camera accumulators of the operator pass live in 160 KB of LDS)");
    if (c->slots != slots || c->Acap != c->N || !c->rows.p) { int rc = alloc_rows(c, slots); if (rc) return rc; }
    hipStream_t s = c->stream;
    CTX_HIP(c, hipEventRecord(c->ev_asm[0], s));
    p = make_params(c, cfg, c->intr, c->dist);
    GridView g = c->grid_view();
    { TimedScope t(c, I3D_K_CLASSIFY);
      launch_classify(s, g, p, c->aflag.p);
      CTX_HIP(c, rocprim::exclusive_scan(c->scan_tmp.p, c->scan_tmp_bytes, c->aflag.p, c->ascan.p, 0, (size_t)c->N, rocprim::plus<int>(), s));
      launch_compact(s, c->N, c->aflag.p, c->ascan.p, c->flags.p, c->aidx.p, c->alist.p, c->aflags.p); }
    int tail[2];
    CTX_HIP(c, hipMemcpyAsync(&tail[0], c->ascan.p + (c->N - 1), sizeof(int), hipMemcpyDeviceToHost, s));
    CTX_HIP(c, hipMemcpyAsync(&tail[1], c->aflag.p + (c->N - 1), sizeof(int), hipMemcpyDeviceToHost, s));
    // the keyframe constants of this iteration's poses (host, fp64, the reference's libm) while the device classifies: the host would wait here anyway
    std::vector<FrameConst> fc; build_frame_consts(c, cfg.rgbd_level, c->poses.data(), fc);
    CTX_HIP(c, sync_stream(c));
    CTX_HIP(c, hipMemcpyAsync(c->d_frames.p, fc.data(), sizeof(FrameConst) * fc.size(), hipMemcpyHostToDevice, s));
    c->A = tail[0] + tail[1];
    { TimedScope t(c, I3D_K_CLASSIFY);
      static const bool no_partition = [] { const char* e = std::getenv("I3D_NO_PARTITION"); return e && e[0] == '1'; }();      // A/B runs
      if (!no_partition) launch_partition_blocks(s, g, c->A, c->alist.p, c->aflags.p, c->aidx.p);
      launch_anbr(s, c->N, c->A, c->Acap, c->alist.p, c->nbr.p, c->aidx.p, c->anbr.p); }
    c->tile_ok = false; c->tile_T = 0;
    // Bit-reproducible operator pass (fixed-order sums inside the workgroups of k_eg_tile as well): the default on one rank, where the multi-system pass is fixed-order
    // by construction and only the lone-system passes pay for it (0.7 % of an iteration, profiles/r05_bench_deterministic.json); I3D_DETERMINISTIC=0 selects the
    // LDS-atomic pass.  A sharded run iterates serially through k_eg_tile (5 % there): opt-in with I3D_DETERMINISTIC=1.
    { const char* e = std::getenv("I3D_DETERMINISTIC"); c->deterministic = e ? e[0] == '1' : !sharded(c); }
    {   // the damping ladder (lm_solve): up to I3D_LADDER consecutive LM attempts solved together, one stream of the rows per group of <= 3 of them (tile_pass_mr.hip);
        // 1 = the serial trust-region loop.  Needs the single-rank tiled pass in its 512-entry geometry with pull lists and the shipped 5 observation slots.
        const char* e = std::getenv("I3D_LADDER"); const int v = e ? std::atoi(e) : LADDER_MAX;
        c->ladder_max = v < 1 ? 1 : (v > LADDER_MAX ? LADDER_MAX : v);
        if (slots != 5 || eg_tile_mr_max_systems(c->K) < 2) c->ladder_max = 1;
        // sharded (round 6): the ladder runs with the exchanges as launches of the transport (RCCL, the rank simulation): pcg_solve_ladder all-reduces the batch's sums in
        // one message per exchange.  Over the mailboxes (the in-kernel exchanges of the three-launch serial pass, I3D_TRANSPORT=p2p) the serial loop stays.
        if (sharded(c)) { P2PDev probe; if (c->comm->fused_exchange(&probe)) c->ladder_max = 1; }
        { const char* l = std::getenv("I3D_PCG_LEGACY"); if (l && l[0] == '1') c->ladder_max = 1; }
        // I3D_EGT_MR1=1: the SERIAL loop streams its rows through k_eg_tile_mr<1> (the multi-system kernel with one system) instead of k_eg_tile — the control of the
        // ladder tests (same kernel family: a system solved alone vs in a batch, bit for bit) and an A/B switch
        { const char* m1 = std::getenv("I3D_EGT_MR1"); c->mr1_serial = m1 && m1[0] == '1' && !sharded(c) && slots == 5 && eg_tile_mr_max_systems(c->K) >= 1; }
        c->ladder_lists = c->ladder_max > 1 || c->mr1_serial;
    }
    {   // halo sums of the operator pass pulled over plan lists instead of pushed with LDS atomics: always in the bit-reproducible mode, else on request
        const char* e = std::getenv("I3D_HALO_PULL");
        c->halo_pull = c->deterministic || (e && e[0] == '1');
        if (c->halo_pull || c->ladder_lists) {      // sized for both tile geometries (Acap entries)
            const size_t n512 = (size_t)tile_plan_tiles_of(c->Acap, 512), n1024 = (size_t)tile_plan_tiles_of(c->Acap, 1024);
            CTX_HIP(c, c->tp_hp_off.alloc(std::max(n512 * (size_t)(1536 + 1), n1024 * (size_t)(2048 + 1)) + 8));
            CTX_HIP(c, c->tp_hp_src.alloc(std::max(n512 * (size_t)tile_plan_pull_cap(1536), n1024 * (size_t)tile_plan_pull_cap(2048)) + 8));
        }
    }
    if (!sharded(c)) {
        // tile geometry: 1024-entry tiles (one workgroup of 16 waves per CU) unless the work list is row-poor — SURVEY.md 8(d)'s 4-voxel shell, real sequences with few
        // observations: a third of the entries own no Eg rows, their waves idle while the others stream, and two smaller workgroups per CU keep more row blocks in
        // flight (measured on --band 2: 0.43 -> 0.39 ms per pass; on the 5.0-rows workload 512-entry tiles are 2.5 % slower).  The density is last iteration's
        // (rows per entry are a property of the grid, not of the iteration); either geometry is the other's fallback when a tile's halo does not fit.
        { const long long la = c->last_sizes[0], lr = c->last_sizes[1];
          const bool forced = std::getenv("I3D_EGT_TILE") != nullptr;
          if (!forced && la > 0 && (double)lr < 0.8 * (double)slots * (double)la) c->tile_T = 512;
          if (c->ladder_max > 1 || c->mr1_serial) c->tile_T = 512; }      // the multi-system pass exists in the 512-entry geometry (8 waves at 2 per SIMD: the column sums of 3 systems live in registers)
        shard_range(c->A, 1, 0, c->chunk, c->own0, c->own1); c->nC = c->A; c->slice = c->chunk;
        RowView r0 = c->row_view(); TimedScope t(c, I3D_K_CLASSIFY);
        CTX_HIP(c, launch_tile_plan(s, r0, c->tile_plan(), c->tp_temp.p, c->tp_temp.n));
    } else {
        // owned range + compute list + halo lists + ghost tiles (every rank derives them from the replicated work list: no communication), then the tile plan.
        // A sharded run has no untiled operator: when a tile's halo does not fit anywhere (the ranks agree: max over ranks), all of them plan again with the
        // other geometry before anything is built on the plan.  First choice: 1024-entry tiles (2048 halo slots; a third fewer tile boundaries, one
        // workgroup per CU) when a rank's share fills at least two rounds of them, else 512-entry tiles (1536 halo slots, two workgroups per CU): a small
        // share balances better over twice the tiles.  Every rank derives the choice from the replicated work list.
        // (256-entry tiles for shares below two rounds were built and measured in round 4: the operator gains 2 us per pass, the halo fold loses more —
        // profiles/r04_share_tile_ab.json — removed.)
        const int first_T = (c->ladder_max > 1) ? 512 : ((c->A / c->comm->world >= 2 * 256 * 1024) ? 1024 : 512);      // the multi-system pass exists in the 512-entry geometry
        c->tile_T = first_T;
        for (int attempt = 0; attempt < 2; ++attempt) {
            int rc = shard_plan(c); if (rc) return rc;
            { RowView r0 = c->row_view(); TimedScope t(c, I3D_K_CLASSIFY);
              CTX_HIP(c, launch_tile_plan(s, r0, c->tile_plan(), c->tp_temp.p, c->tp_temp.n)); }
            launch_fill_d(s, 1, c->d_scal.p + 20, 0.0); launch_int_to_double(s, c->tp_overflow.p, c->d_scal.p + 20);
            rc = allreduce(c, c->d_scal.p + 20, 1); if (rc) return rc;
            double over = 0.0; rc = read_doubles(c, c->d_scal.p + 20, 1, &over); if (rc) return rc;
            if (over == 0.0) break;
            if (attempt == 1) return ctx_fail(c, I3D_ERR_CAPACITY, "sharded optimize: a tile of the operator pass reaches more foreign entries than either tile geometry has halo slots");
            const int other = c->plan_T() == 512 ? 1024 : 512;
            std::fprintf(stderr, "[i3d] sharded operator pass: a %d-entry tile's halo does not fit, planning again with %d-entry tiles\n", c->plan_T(), other);
            c->tile_T = other;
        }
        // rows are only built on the compute list: everything else on the tiles this rank runs must be inert
        CTX_HIP(c, hipMemsetAsync(c->nrows.p, 0, (size_t)c->A, s)); CTX_HIP(c, hipMemsetAsync(c->regflags.p, 0, (size_t)c->A, s));
    }
    RowView r = c->row_view();
    { TimedScope t(c, I3D_K_OBSERVE);
      // conservative (group of 64 entries, keyframe) culling: the depth-range pyramid once per keyframe set and level, spheres and masks per iteration
      // I3D_NO_CULL (A/B runs and the with / without test; read at every assemble): 1 = neither, 2 = no group masks, 3 = no weight-bound prefilter
      const char* e = std::getenv("I3D_NO_CULL"); const char mode = e ? e[0] : '0';
      const unsigned* mask = nullptr; c->cull_on = !(mode == '1' || mode == '2'); const bool prefilter = !(mode == '1' || mode == '3');
      if (c->cull_on) {
          if (c->cull_level != cfg.rgbd_level || !c->cull_blocks.p) {
              CTX_HIP(c, c->cull_blocks.alloc((size_t)c->K * cull_pyramid(p.w, p.h).cells));
              launch_depth_blocks(s, c->d_frames.p, c->K, p.w, p.h, c->cull_blocks.p); c->cull_level = cfg.rgbd_level; }
          launch_group_cull(s, g, r, p, c->d_frames.p, c->cull_blocks.p, c->cull_bounds.p, c->cull_mask.p); mask = c->cull_mask.p; }
      launch_observe(s, g, r, p, c->d_frames.p, mask, prefilter); }
    // the reference's three-way split (nls_solver.cpp:66-67,101): time_add = collecting the residuals (addVoxelResiduals: classification, observations),
    // time_build = buildProblem (cost functions, weight normalisation), time_solve.  The boundary is an event on the stream (round 3: a synchronisation).
    CTX_HIP(c, hipEventRecord(c->ev_asm[1], s));
    { TimedScope t(c, I3D_K_BUILD); launch_build(s, g, r, p, c->d_frames.p, true, nullptr, c->d_partials.p); }
    { TimedScope t(c, I3D_K_CLASSIFY); launch_group_rows(s, c->A, c->nrows.p, c->gmax.p); }
    { TimedScope t(c, I3D_K_CLASSIFY); launch_eaw_sym(s, r, c->tile_plan(), sharded(c) ? c->cflag.p : nullptr); }
    CTX_HIP(c, hipMemsetAsync(c->d_scal.p, 0, sizeof(double) * 32, s));
    { TimedScope t(c, I3D_K_VECTOR); launch_weight_sums(s, r, g, cost_from_sums(c), c->d_scal.p, c->d_partials.p); }
    { int rc = allreduce(c, c->d_scal.p, 13); if (rc) return rc; }
    int tp_over = 1;
    CTX_HIP(c, hipMemcpyAsync(&tp_over, c->tp_overflow.p, sizeof(int), hipMemcpyDeviceToHost, s));
    double sums[13]; { int rc = read_doubles(c, c->d_scal.p, 13, sums); if (rc) return rc; }
    if (sharded(c) && c->comm->health(s)) return ctx_fail(c, I3D_ERR_COMM, "assemble: a peer-to-peer exchange timed out (a rank stopped taking part)");
    if (!sharded(c) && tp_over != 0) {
        // single rank: the other geometry before giving up on the tiled pass
        const int other = c->plan_T() == 1024 ? 512 : 1024;
        std::fprintf(stderr, "[i3d] operator pass: a %d-entry tile's halo does not fit, planning again with %d-entry tiles\n", c->plan_T(), other);
        c->tile_T = other;
        { RowView r0 = c->row_view(); TimedScope t(c, I3D_K_CLASSIFY);
          CTX_HIP(c, launch_tile_plan(s, r0, c->tile_plan(), c->tp_temp.p, c->tp_temp.n)); }      // (the symmetric Ea weights do not depend on the geometry)
        CTX_HIP(c, hipMemcpyAsync(&tp_over, c->tp_overflow.p, sizeof(int), hipMemcpyDeviceToHost, s));
        CTX_HIP(c, sync_stream(c));
    }
    c->tile_ok = tp_over == 0;       // a halo that does not fit (pathological grids) -> the untiled operator pass (single rank only)
    if (!sharded(c) && !c->tile_ok) std::fprintf(stderr, "[i3d] operator pass: a tile's halo does not fit, using the untiled pass (k_eg_jtjp + k_gather)\n");
    if (sharded(c) && !c->tile_ok) return ctx_fail(c, I3D_ERR_CAPACITY, "sharded optimize: the tile plan overflowed after it had been accepted");      // (cannot happen: agreed on above)
    if (!sharded(c)) { const char* e = std::getenv("I3D_NO_TILE"); if (e && e[0] == '1') c->tile_ok = false; }      // tests of the single-rank fallback (k_eg_jtjp + k_gather); a sharded run has no untiled pass
    sums[5] = sums[1]; sums[6] = sums[2];
    const double lambda[4] = {cfg.lambda_g, varying_lambda(iteration, cfg.iterations, cfg.lambda_r0, cfg.lambda_r1),
                              varying_lambda(iteration, cfg.iterations, cfg.lambda_s0, cfg.lambda_s1), cfg.lambda_a};
    for (int t = 0; t < 4; ++t) { p.type_w[t] = sums[t] != 0.0 ? (lambda[t] / sums[t]) * 1000.0 : 0.0; p.type_wf[t] = (float)p.type_w[t]; }     // nls_solver.cpp:379-394
    c->n_active = (long long)(sums[8] + 0.5);
    // the cost at the point the rows were built at (k_weight_sums): what the trust-region loop starts from
    c->cost_at_build = 0.5 * (p.type_w[0] * sums[9] + p.type_w[1] * sums[10] + p.type_w[2] * sums[11] + p.type_w[3] * sums[12]);
    if (st) { for (int t = 0; t < 4; ++t) { st->rows[t] = (int64_t)(sums[4 + t] + 0.5); st->weight_sum[t] = sums[t]; st->type_weight[t] = p.type_w[t]; } st->valid_voxels = c->n_active; }
    c->last_sizes[0] = c->n_active; for (int t = 0; t < 4; ++t) c->last_sizes[1 + t] = (long long)(sums[4 + t] + 0.5);
    c->last_params = p; c->assembled = true;
    { float ms = 0.0f; c->t_add_end = (hipEventElapsedTime(&ms, c->ev_asm[0], c->ev_asm[1]) == hipSuccess) ? (double)ms * 1e-3 : -1.0; }      // (both events lie before the read-back above)
    { const int lrc = ctx_launch_check(c); if (lrc) return lrc; }
    return I3D_OK;
}

// ---- normal-equation pieces (GRAD / COLNORM; the PCG operator is inlined in pcg_solve) ----------------------------------------
static int run_pass(i3d_context* c, PassMode mode, const OptParams& p, const float* u, float* out /*[NP]*/) {
    hipStream_t s = c->stream; GridView g = c->grid_view(); RowView r = c->row_view(); const Layout L = layout_of(c);
    const int stride = (int)aux_part_stride(c->K);
    PassBuffers b{c->C.p, c->treg.p, c->d_shared.p, c->d_blocks.p, c->aux_part.p, stride};
    CTX_HIP(c, hipMemsetAsync(c->d_shared.p, 0, sizeof(double) * ((size_t)L.NS + 1), s));
    if (mode == PASS_COLNORM) CTX_HIP(c, hipMemsetAsync(c->d_blocks.p, 0, sizeof(double) * (21 * (size_t)c->K + 25), s));
    // the camera block leaves every launch as one float row per workgroup, added up in workgroup order (launch_sum_rows): no global atomics, bit-reproducible
    if (mode == PASS_JTJP && c->tile_ok && !sharded(c)) {        // the PCG's tiled operator pass (raw accumulators straight into `out`)
        const int NSP = (L.NS + 3) & ~3; int rows = 0;
        { TimedScope t(c, I3D_K_EG_PASS); rows = launch_eg_tile(s, r, p, u, c->tile_plan(), nullptr, out, nullptr, nullptr, c->cam_part.p, NSP); }
        { TimedScope t(c, I3D_K_GATHER); launch_halo_fold(s, r, c->tile_plan(), out, nullptr); }
        { TimedScope t(c, I3D_K_VECTOR); launch_mul(s, 2 * c->chunk, c->v_mask.p, out, out);       // the raw accumulators also cover fixed unknowns (the PCG multiplies them by S = 0)
          if (rows > 0) launch_sum_rows(s, PASS_JTJP, c->K, c->cam_part.p, rows, NSP, c->d_shared.p, c->d_blocks.p); }
    } else {
        int rows = 0;
        { TimedScope t(c, mode == PASS_JTJP ? I3D_K_EG_PASS : I3D_K_EG_AUX); rows = launch_eg_pass(s, mode, g, r, p, u, b, nullptr);
          if (rows > AUX_PART_ROWS) return ctx_fail(c, I3D_ERR_CAPACITY, "run_pass: more workgroups than rows of the camera partial buffer");
          if (rows > 0) launch_sum_rows(s, mode, c->K, c->aux_part.p, rows, stride, c->d_shared.p, c->d_blocks.p); }
        { TimedScope t(c, I3D_K_GATHER); launch_gather(s, mode, r, b, out); }
    }
    { int rc = allreduce(c, c->d_shared.p, (size_t)L.NS); if (rc) return rc; }
    if (mode == PASS_COLNORM) { int rc = allreduce(c, c->d_blocks.p, 21 * (size_t)c->K + 25); if (rc) return rc; }
    { TimedScope t(c, I3D_K_VECTOR); launch_shared_finalize(s, L.tail_off, c->K, p, c->d_shared.p, out, false, nullptr, nullptr, nullptr, nullptr, nullptr); }
    return I3D_OK;
}

// Column norms -> c->v_c (+ d_shared / d_blocks: the camera part and the camera blocks) and the raw gradient J^T W r -> c->v_acc from ONE stream of the rows
// (gradcol.hip).  One rank only (a sharded run keeps the two passes: their all-reduces sit between the launches).  `between` runs after the column norms are
// complete and before d_shared is reused for the gradient's camera part.
template <class F> static int run_gradcol(i3d_context* c, const OptParams& p, F between) {
    hipStream_t s = c->stream; GridView g = c->grid_view(); RowView r = c->row_view(); const Layout L = layout_of(c);
    const int stride = (int)gc_part_stride(c->K), col_off = (int)gc_col_off(c->K);
    GradColBuffers gb{c->C.p, c->C2.p, c->treg.p, c->treg2.p, c->gc_part.p, stride, col_off, c->d_partials.p, c->d_scal.p + 16};
    CTX_HIP(c, hipMemsetAsync(c->d_scal.p + 16, 0, sizeof(double), s));
    CTX_HIP(c, hipMemsetAsync(c->d_shared.p, 0, sizeof(double) * ((size_t)L.NS + 1), s));
    CTX_HIP(c, hipMemsetAsync(c->d_blocks.p, 0, sizeof(double) * (21 * (size_t)c->K + 25), s));
    int rows = 0;
    { TimedScope t(c, I3D_K_EG_AUX); rows = launch_eg_gradcol(s, g, r, p, gb);
      if (rows > GC_PART_ROWS) return ctx_fail(c, I3D_ERR_CAPACITY, "run_gradcol: more workgroups than rows of the camera partial buffer");
      if (rows > 0) launch_sum_rows(s, PASS_COLNORM, c->K, c->gc_part.p + col_off, rows, stride, c->d_shared.p, c->d_blocks.p); }
    { PassBuffers b{c->C2.p, c->treg2.p, c->d_shared.p, c->d_blocks.p, nullptr, 0};
      TimedScope t(c, I3D_K_GATHER); launch_gather(s, PASS_COLNORM, r, b, c->v_c.p); }
    { TimedScope t(c, I3D_K_VECTOR); launch_shared_finalize(s, L.tail_off, c->K, p, c->d_shared.p, c->v_c.p, false, nullptr, nullptr, nullptr, nullptr, nullptr); }
    { int rc = between(); if (rc) return rc; }
    CTX_HIP(c, hipMemsetAsync(c->d_shared.p, 0, sizeof(double) * ((size_t)L.NS + 1), s));
    { TimedScope t(c, I3D_K_EG_AUX); if (rows > 0) launch_sum_rows(s, PASS_GRAD, c->K, c->gc_part.p, rows, stride, c->d_shared.p, c->d_blocks.p); }
    { PassBuffers b{c->C.p, c->treg.p, c->d_shared.p, c->d_blocks.p, nullptr, 0};
      TimedScope t(c, I3D_K_GATHER); launch_gather(s, PASS_GRAD, r, b, c->v_acc.p); }
    { TimedScope t(c, I3D_K_VECTOR); launch_shared_finalize(s, L.tail_off, c->K, p, c->d_shared.p, c->v_acc.p, false, nullptr, nullptr, nullptr, nullptr, nullptr); }
    return I3D_OK;
}

// a.b over the distributed vector: own slice -> all-reduce -> + the replicated camera tail; the result stays on the device (*slot, one of d_scal's doubles)
static int dot_dev(i3d_context* c, const float* a, const float* b, double* slot) {
    const Layout L = layout_of(c);
    CTX_HIP(c, hipMemsetAsync(slot, 0, sizeof(double), c->stream));
    { TimedScope t(c, I3D_K_VECTOR); launch_dot2(c->stream, L.own, a, b, slot, c->d_partials.p); }
    { int rc = allreduce(c, slot, 1); if (rc) return rc; }
    { TimedScope t(c, I3D_K_VECTOR); launch_dot(c->stream, L.NS, a + L.tail_off, b + L.tail_off, slot, c->d_partials.p); }
    return I3D_OK;
}

// slot = number of owned entries (+ the replicated camera tail, counted once) with |a m| > tol
static int count_above_dev(i3d_context* c, const float* a, const float* m, float tol, double* slot) {
    const Layout L = layout_of(c);
    CTX_HIP(c, hipMemsetAsync(slot, 0, sizeof(double), c->stream));
    { TimedScope t(c, I3D_K_VECTOR); launch_count_above2(c->stream, L.own, a, m, tol, slot, c->d_partials.p); }
    { int rc = allreduce(c, slot, 1); if (rc) return rc; }
    { TimedScope t(c, I3D_K_VECTOR); launch_count_above(c->stream, L.NS, a + L.tail_off, m + L.tail_off, tol, slot, c->d_partials.p); }
    return I3D_OK;
}

static int eval_cost_launch(i3d_context* c, const OptParams& p, bool candidate, const FrameConst* frames, const double* cam9 = nullptr, const LmState* lm = nullptr) {
    GridView g = c->grid_view();
    if (candidate) { g.x_sdf = c->xc_sdf.p; g.x_alb = c->xc_alb.p; }
    { TimedScope t(c, I3D_K_COST); launch_build(c->stream, g, c->row_view(), p, frames, false, c->d_scal.p + 16, c->d_partials.p, cam9, lm); }
    return allreduce(c, c->d_scal.p + 16, 1);
}
static int eval_cost(i3d_context* c, const OptParams& p, bool candidate, const FrameConst* frames, double* cost) {
    int rc = eval_cost_launch(c, p, candidate, frames); if (rc) return rc;
    rc = read_doubles(c, c->d_scal.p + 16, 1, cost); if (rc) return rc;
    if (sharded(c) && c->comm->health(c->stream)) return ctx_fail(c, I3D_ERR_COMM, "cost evaluation: a peer-to-peer exchange timed out (a rank stopped taking part)");
    return I3D_OK;
}

// CGNR (ConjugateGradientsSolver) on (S J^T W J S + D^2) x = b, x0 = 0.  No host synchronisation inside an iteration.
// Sharded (one process per GPU): a rank iterates on its two owned segments of every vector + the replicated camera tail.  Per pass:
//   * ONE neighbour exchange: the operator input u = S p on the rim the rank's rows read (Comm::push_halo, a few tens of KB per pair);
//   * ONE small all-reduce after the operator: [camera block 6K+9 | p.q] (fp64);
//   * ONE all-reduce of the 4 iteration scalars (r.z, x.(b+r), x.r, sum D^2 x^2) at the iteration boundary.
// No vector is gathered: everything that lands on an owned unknown is computed from rows the rank holds itself (owned + ghost entries).
// The terminal state stays on the device: k_lm_decide reads it on the stream.
static int pcg_solve(i3d_context* c, const i3d_optimizer_config& cfg, const OptParams& p, const PcgState** final_state) {
    hipStream_t s = c->stream;
    const Layout L = layout_of(c);
    const int K = c->K; const bool multi = sharded(c), tiled = c->tile_ok;
    GridView g = c->grid_view(); RowView r = c->row_view();
    PassBuffers pb{c->C.p, c->treg.p, c->d_shared.p, c->d_blocks.p, c->aux_part.p, (int)aux_part_stride(c->K)};
    PcgState* st = c->d_pcg.p;
    double* pq_slot = c->d_shared.p + L.NS;
    const size_t to = L.tail_off; const Seg2 own = L.own;
    { TimedScope t(c, I3D_K_VECTOR); launch_fill(s, (int)L.NP, c->v_x.p, 0.0f); launch_pcg_init(s, st, cfg.pcg_fixed_iterations, 500, c->d_lm.p); }
    CTX_HIP(c, hipMemcpyAsync(c->v_r.p, c->v_b.p, sizeof(float) * L.NP, hipMemcpyDeviceToDevice, s));
    // fp64 partial sums: [0, 4*2048) slice sums of k_pcg_step, then the p.q partials of the operator pass, then the D^2 p^2 partials of k_pcg_direction
    double* const step_part = c->d_partials.p; double* const pq_part = c->d_partials.p + 4 * 2048; double* const d2_part = pq_part + 1024;
    int n_pq = 0, n_step = 0, n_d2 = 0;
    // Sharded over the peer-to-peer transport: the two reductions of a pass run INSIDE the boundary kernels (k_pcg_tail_a / k_pcg_tail_b), a pass
    // has no reduction launches of its own.  Any other transport: separate all-reduce launches around the same kernels (pd.on = 0).
    P2PDev pd; std::memset(&pd, 0, sizeof(pd));
    const bool fused = multi && tiled && c->comm->device_reduce(&pd) && L.NS + 1 <= pd.L.red_cap;
    if (!fused) pd.on = 0;
    // The operator on the rows of this rank.  Tiled (tile_pass.hip): raw accumulators J^T W J u in v_qacc (camera block in d_shared, p.q partials
    // row by row); the vector q = S acc + D^2 v is formed inside k_pcg_step.  Untiled fallback (single rank): k_eg_jtjp + k_gather -> out.
    auto rows_apply = [&](const float* v, float* out, bool with_dot, bool zero_first) -> int {
        if (zero_first) CTX_HIP(c, hipMemsetAsync(c->d_shared.p, 0, sizeof(double) * ((size_t)L.NS + 1), s));      // (the pass boundary kernel zeroes it otherwise)
        if (multi) { int rc = push_halo(c, c->v_u.p); if (rc) return rc; }
        if (tiled) {
            { TimedScope t(c, I3D_K_EG_PASS);
              n_pq = launch_eg_tile(s, r, p, c->v_u.p, c->tile_plan(), c->d_shared.p, c->v_qacc.p, with_dot ? pq_part : nullptr, st); }
            { TimedScope t(c, I3D_K_GATHER); launch_halo_fold(s, r, c->tile_plan(), c->v_qacc.p, st); }
            if (!with_dot) n_pq = 0;
            if (!multi || (fused && with_dot)) return I3D_OK;      // fused: k_pcg_tail_b adds the partials and sums [camera block | p.q] over the ranks
            // the rank's p.q (rows + D^2 p^2 of its slice) rides with the camera block
            if (with_dot) { TimedScope t(c, I3D_K_VECTOR); launch_reduce_partials(s, pq_part, n_pq, 1, pq_slot, st); launch_reduce_partials(s, d2_part, n_d2, 1, pq_slot, st); }
            n_pq = 0;
            return allreduce(c, c->d_shared.p, (size_t)L.NS + 1);
        }
        { TimedScope t(c, I3D_K_EG_PASS); launch_eg_pass(s, PASS_JTJP, g, r, p, c->v_u.p, pb, st); }
        { TimedScope t(c, I3D_K_GATHER); n_pq = launch_gather_tail(s, r, pb, out, c->v_S.p, c->v_D2.p, v, with_dot ? pq_part : nullptr, false, st); }
        if (!with_dot) n_pq = 0;
        return I3D_OK;
    };
    // sharded: the slice sums of k_pcg_step go into acc (one small launch: thousands of workgroups adding into four doubles serialise at the L2,
    // measured 71 vs 32 us for the step kernel), which is all-reduced at the iteration boundary; single rank: k_pcg_tail_a adds the partials itself
    auto fold_step = [&]() { if (multi && !fused) { launch_reduce_partials(s, step_part, n_step, 4, st->acc, st); n_step = 0; } };
    const float* const Sq = tiled ? c->v_S.p : nullptr;          // tiled pass: k_pcg_step forms q from the accumulators
    { TimedScope t(c, I3D_K_VECTOR);
      n_step = launch_pcg_step(s, 0 /*init*/, own, c->v_p.p, c->v_q.p, c->v_x.p, c->v_r.p, c->v_b.p, c->v_D2.p, c->v_Minv.p, c->v_z.p, nullptr, step_part, st);
      fold_step(); }
    int tail_mode = 0;
    const int seq0 = c->pcg_seq;               // pass numbers are unique across solves: a stale ring entry can never match
    int it = 1;
    for (;; ++it) {
        // iteration boundary.  Sharded: the 4 slice sums (r.z, x.(b+r), x.r, sum D2 x^2) are summed over the ranks — inside k_pcg_tail_a over the
        // mailbox transport, by a reduction + all-reduce launch otherwise
        if (multi && !fused) { int rc = allreduce(c, st->acc, 4); if (rc) return rc; }
        if (fused) { c->comm->count_reduce(4); c->comm->count_reduce((size_t)L.NS + 1); }
        { TimedScope t(c, I3D_K_VECTOR);
          launch_pcg_tail_a(s, tail_mode, to, K, c->Minv_blocks.p, c->v_p.p, tail_mode == 3 ? c->v_tmp.p : c->v_q.p, c->v_x.p, c->v_r.p, c->v_b.p, c->v_D2.p, c->v_z.p,
                            step_part, n_step, st, c->d_shared.p, L.NS + 1, c->d_flags, seq0 + it, pd); }
        { TimedScope t(c, I3D_K_VECTOR);        // p = z + beta p, u = S p on the owned segments and the (replicated) camera tail; sharded: D^2 p^2 of the tail is added by k_pcg_tail_b
          n_d2 = launch_pcg_direction(s, own, to, L.NS, c->v_z.p, c->v_p.p, c->v_S.p, c->v_u.p, c->v_D2.p, tiled ? d2_part : nullptr, st); }
        { int rc = rows_apply(c->v_p.p, c->v_q.p, true, false); if (rc) return rc; }
        { TimedScope t(c, I3D_K_VECTOR); launch_pcg_tail_b(s, to, K, p, c->d_shared.p, pq_slot, pq_part, n_pq, d2_part, (tiled && (!multi || fused)) ? n_d2 : 0, tiled,
                                                           c->v_q.p, c->v_S.p, c->v_D2.p, c->v_p.p, st, pd); }
        const bool reset = (it % 10 == 0);                                       // residual_reset_period
        if (!reset) {
            TimedScope t(c, I3D_K_VECTOR);
            n_step = launch_pcg_step(s, 1, own, c->v_p.p, tiled ? c->v_qacc.p : c->v_q.p, c->v_x.p, c->v_r.p, c->v_b.p, c->v_D2.p, c->v_Minv.p, c->v_z.p, Sq, step_part, st);
            fold_step();
            tail_mode = 1;
        } else {                                                                 // r = b - A x instead of r -= alpha q
            { TimedScope t(c, I3D_K_VECTOR);
              launch_pcg_step(s, 2, own, c->v_p.p, c->v_q.p, c->v_x.p, c->v_r.p, c->v_b.p, c->v_D2.p, c->v_Minv.p, c->v_z.p, nullptr, step_part, st);
              launch_pcg_tail_x(s, to, K, c->v_p.p, c->v_x.p, st);
              launch_mul2(s, own, c->v_S.p, c->v_x.p, c->v_u.p); launch_mul(s, L.NS, c->v_S.p + to, c->v_x.p + to, c->v_u.p + to); }
            { int rc = rows_apply(c->v_x.p, c->v_tmp.p, false, true); if (rc) return rc; }
            { TimedScope t(c, I3D_K_VECTOR);
              launch_shared_finalize(s, to, K, p, c->d_shared.p, c->v_tmp.p, true, c->v_S.p, c->v_D2.p, c->v_x.p, nullptr, st);
              n_step = launch_pcg_step(s, 3, own, c->v_p.p, tiled ? c->v_qacc.p : c->v_tmp.p, c->v_x.p, c->v_r.p, c->v_b.p, c->v_D2.p, c->v_Minv.p, c->v_z.p, Sq, step_part, st);
              fold_step(); }
            tail_mode = 3;
        }
        if (it >= 2) {                                                           // look at the boundary of pass it-1 while pass it runs
            const int want = seq0 + it - 1; volatile int* ring = c->h_flags + 2 * (want & 1);
            const double t_wait = now_s();
            while (__atomic_load_n((int*)&ring[0], __ATOMIC_ACQUIRE) != want) {
                if (now_s() - t_wait > 30.0) { CTX_HIP(c, sync_stream(c)); if (__atomic_load_n((int*)&ring[0], __ATOMIC_ACQUIRE) != want) return ctx_fail(c, I3D_ERR_HIP, "pcg_solve: the device stopped publishing its state"); }
            }
            if (__atomic_load_n((int*)&ring[1], __ATOMIC_ACQUIRE)) break;
        }
        if (it > 520) break;
    }
    c->pcg_seq = seq0 + PCG_SEQ_STRIDE;
    *final_state = st;                   // kernels after `done` were no-ops, so this is the terminal state (read by k_lm_decide on the stream)
    return I3D_OK;
}

// The same solve in THREE launches per pass (pcg_fused.hip): k_pcg_dir3 | k_eg_tile | k_pcg_step3.  Single rank, tiled operator.  The scalar state is
// double-buffered by pass parity: boundary `it` reads st2[(it + 1) & 1] and writes st2[it & 1], which the operator and the step of pass `it` read.
// Sharded (one process per GPU, mailbox transport): the SAME three launches — the boundary's exchange ([4 slice sums] + the rim of z) runs inside k_pcg_dir3, the
// operator's ([p.q | camera block]) inside k_pcg_step3; the residual-reset passes (one in ten) push the rim of u = S x with a launch of their own (k_rim_u).
static int pcg_solve_fused(i3d_context* c, const i3d_optimizer_config& cfg, const OptParams& p, const PcgState** final_state) {
    hipStream_t s = c->stream;
    const Layout L = layout_of(c);
    const int K = c->K; const size_t to = L.tail_off; const Seg2 own = L.own;
    RowView r = c->row_view(); TilePlan tp = c->tile_plan();
    const bool sh = sharded(c);
    ShardArgs sa; std::memset(&sa, 0, sizeof(sa));
    if (sh) {
        if (!c->comm->fused_exchange(&sa.pd) || L.NS + 2 > sa.pd.L.red_cap) return ctx_fail(c, I3D_ERR_STATE, "pcg_solve_fused: the in-kernel exchanges are not available");
        const HaloPlan& h = c->halo;
        sa.rim = RimLists{h.n_send, h.n_recv, c->halo_send_idx.p, c->halo_send_peer.p, c->halo_recv_idx.p, c->halo_recv_peer.p, c->halo_offs.p, c->halo_offs.p + P2P_MAX_RANKS};
        sa.n_rim_wg = (h.n_send + h.n_recv) > 0 ? std::max(1, std::min(4, (std::max(h.n_send, h.n_recv) + 1023) / 1024)) : 1;
        sa.zb = c->v_z.p; sa.pb = c->v_p.p; sa.ub = c->v_u.p; sa.cmb = c->v_cm.p; sa.chunk = c->chunk;
    }
    const int wg_cap = sh ? sa.pd.wg_cap : 0;
    PcgState* const st2 = c->d_pcg2.p;
    const int NSP = (L.NS + 3) & ~3;
    double* const step_part = c->d_partials.p; double* const pq_part = c->d_partials.p + 4 * 2048; double* const d2_part = pq_part + 2048;
    { TimedScope t(c, I3D_K_VECTOR); launch_fill(s, (int)L.NP, c->v_x.p, 0.0f); launch_pcg_init3(s, st2, cfg.pcg_fixed_iterations, 500, c->d_lm.p); }
    CTX_HIP(c, hipMemcpyAsync(c->v_r.p, c->v_b.p, sizeof(float) * L.NP, hipMemcpyDeviceToDevice, s));
    Step3Args a; std::memset(&a, 0, sizeof(a));
    auto c4 = [&](const float* v) { return reinterpret_cast<const float4*>(v + own.off0); };
    auto m4 = [&](float* v) { return reinterpret_cast<float4*>(v + own.off0); };
    a.nq = own.n >> 2; a.chunk4 = (int)((own.off1 - own.off0) >> 2);
    a.p = c4(c->v_p.p); a.qacc = c4(c->v_qacc.p); a.x = m4(c->v_x.p); a.r = m4(c->v_r.p); a.b = c4(c->v_b.p); a.z = m4(c->v_z.p); a.cm = c4(c->v_cm.p); a.lm = c->d_lm.p;
    a.ext_off = tp.ext_off; a.ext_pos = tp.ext_pos; a.qh = reinterpret_cast<const float2*>(tp.qh); a.e0 = (int)own.off0;
    a.pq_partials = pq_part; a.d2_partials = d2_part; a.n_pq = 0; a.n_d2 = 0;
    a.n_slice_wg = pcg_step3_slice_wgs(own.n, wg_cap);
    a.sharded = sh ? 1 : 0;
    a.K = K; a.fix_poses = p.fix_poses; a.fix_intr = p.fix_intr; a.fix_dist = p.fix_dist; a.lad_sys = -1;
    a.cam_partials = c->cam_part.p; a.n_cam = 0; a.cam_stride = NSP; a.Mblk = c->Minv_blocks.p;
    a.tp = c->v_p.p + to; a.tx = c->v_x.p + to; a.tr = c->v_r.p + to; a.tb = c->v_b.p + to; a.tD2 = c->v_D2.p + to; a.tz = c->v_z.p + to; a.tS = c->v_S.p + to;
    a.step_partials = step_part;
    int n_step = 0;
    { TimedScope t(c, I3D_K_VECTOR); a.cur = st2; n_step = launch_pcg_step3(s, 0 /*init*/, a); }
    // pass numbers are the epochs of the in-kernel exchanges: identical on all ranks (every rank queues the same solves; how many passes a rank's HOST queued
    // behind the convergence flag may differ, so every solve takes a fixed block of numbers)
    const bool mr1 = c->mr1_serial && !sh && tp.T == 512 && tp.hp_off != nullptr && r.slots == 5;
    auto op = [&](double* pq, const PcgState* cur) -> int {      // q_acc = J^T W J u on the rows of this rank
        if (mr1) { const int sys0[3] = {0, 0, 0}; LadVec z; std::memset(&z, 0, sizeof(z));
                   return launch_eg_tile_mr(s, r, p, tp, 1, sys0, c->v_u.p, c->v_qacc.p, tp.qh, pq, c->cam_part.p, NSP, cur, z); }
        return launch_eg_tile(s, r, p, c->v_u.p, tp, nullptr, c->v_qacc.p, pq, cur, c->cam_part.p, NSP);
    };
    const int seq0 = c->pcg_seq;
    int it = 1;
    for (;; ++it) {
        PcgState* const prev = st2 + ((it + 1) & 1); PcgState* const cur = st2 + (it & 1);
        sa.seq = seq0 + it; sa.n_slice_partials = a.n_slice_wg; a.sh = sa;
        { TimedScope t(c, I3D_K_VECTOR);
          a.n_d2 = launch_pcg_dir3(s, it == 1, own, to, L.NS, c->v_z.p, c->v_p.p, c->v_S.p, c->v_u.p, c->v_D2.p, c->v_cm.p, c->d_lm.p, step_part, n_step, d2_part, prev, cur, c->d_flags, seq0 + it,
                                   sh ? &sa : nullptr); }
        if (sh) { c->comm->count_reduce(4); c->comm->count_halo(c->halo.n_send); c->comm->count_reduce((size_t)L.NS + 1); }      // (logged as exchanges of this pass: they have no launches of their own)
        { TimedScope t(c, I3D_K_EG_PASS); a.n_pq = op(pq_part, cur); a.n_cam = a.n_pq; }
        a.cur = cur;
        if (it % 10 != 0) { TimedScope t(c, I3D_K_VECTOR); n_step = launch_pcg_step3(s, 1, a); }
        else {                                                                   // residual_reset_period: r = b - A x instead of r -= alpha q
            { TimedScope t(c, I3D_K_VECTOR); launch_pcg_step3(s, 2, a);
              launch_mul2(s, own, c->v_S.p, c->v_x.p, c->v_u.p); launch_mul(s, L.NS, c->v_S.p + to, c->v_x.p + to, c->v_u.p + to); }
            if (sh) { TimedScope t(c, I3D_K_COMM); launch_rim_u(s, sa, cur); c->comm->count_halo(c->halo.n_send); }
            { TimedScope t(c, I3D_K_EG_PASS); a.n_cam = op(nullptr, cur); }
            { TimedScope t(c, I3D_K_VECTOR); n_step = launch_pcg_step3(s, 3, a); }
        }
        if (it >= 2) {                                                           // look at the boundary of pass it-1 while pass it runs
            const int want = seq0 + it - 1; volatile int* ring = c->h_flags + 2 * (want & 1);
            const double t_wait = now_s();
            while (__atomic_load_n((int*)&ring[0], __ATOMIC_ACQUIRE) != want) {
                if (now_s() - t_wait > 30.0) { CTX_HIP(c, sync_stream(c)); if (__atomic_load_n((int*)&ring[0], __ATOMIC_ACQUIRE) != want) return ctx_fail(c, I3D_ERR_HIP, "pcg_solve: the device stopped publishing its state"); }
            }
            if (__atomic_load_n((int*)&ring[1], __ATOMIC_ACQUIRE)) break;
        }
        if (it > 520) break;
    }
    c->pcg_seq = seq0 + PCG_SEQ_STRIDE;
    // boundary `it` copied the terminal state forward (kernels after `done` are no-ops), so st2[it & 1] is final (read by k_lm_decide on the stream)
    *final_state = st2 + (it & 1);
    { const int lrc = ctx_launch_check(c); if (lrc) return lrc; }
    return I3D_OK;
}

// ---- the damping ladder ------------------------------------------------------------------------------------------------------------------------------
// Slabs of the per-system arrays of a batch (LadVec): every system has its own x, r, p, z, u, operator accumulators, halo sums, camera partial rows, partial
// sums, block-Jacobi inverses, camera-tail diagonal and scalar states; b, the column norms and S are shared.
constexpr int LAD_PART_STEP = 4 * 2048, LAD_PART_PQ = 2048, LAD_PART_D2 = 2048;
static size_t lad_redop_stride(int NS) { return ((size_t)NS + 1 + 3) & ~(size_t)3; }
// Sized by what the CURRENT work list needs (round 6, advisor finding of round 5: the slabs were sized by the whole grid — 2.3 GB of vectors + 1.2 GB of halo sums at the bench
// size, 290 bytes per stored voxel on top of the 16 solver vectors, whatever the active share): the vectors by the list's layout (2 chunk + 6K + 9, rounded up to 64 K entries so
// that a list that grows a little does not reallocate), the halo sums by the list's tiles.  Grow-only.  Returns non-zero WITHOUT latching an error when the device cannot hold
// them: the caller then solves this outer iteration with the serial trust-region loop.
static int alloc_ladder(i3d_context* c) {
    const Layout L = layout_of(c);
    const int nsys = c->ladder_max;
    LadVec& lv = c->lad;
    const size_t vec = ((size_t)L.NP + 65535) & ~(size_t)65535;
    const size_t qh = 2 * (size_t)tile_plan_tiles_of((int)std::min((size_t)c->Acap, ((size_t)c->chunk + 65535) & ~(size_t)65535), 512) * (size_t)tile_plan_hmax_of(512);
    const size_t cam = (size_t)2048 * (((size_t)L.NS + 3) & ~(size_t)3);
    lv.vec = vec; lv.qh = qh; lv.cam = cam; lv.part = LAD_PART_STEP + LAD_PART_PQ + LAD_PART_D2; lv.mblk = ((size_t)36 * c->K + 41 + 3) & ~(size_t)3; lv.tail = ((size_t)L.NS + 3) & ~(size_t)3;
    const bool fresh = c->lad_vec.n < (size_t)6 * nsys * vec || !c->lad_vec.p;
    bool ok = c->lad_vec.alloc((size_t)6 * nsys * vec) == hipSuccess;
    if (ok && fresh) ok = hipMemsetAsync(c->lad_vec.p, 0, sizeof(float) * c->lad_vec.n, c->stream) == hipSuccess;      // padding entries stay finite
    ok = ok && c->lad_qh.alloc((size_t)nsys * qh) == hipSuccess && c->lad_cam.alloc((size_t)nsys * cam) == hipSuccess && c->lad_part.alloc((size_t)nsys * lv.part) == hipSuccess
            && c->lad_mblk.alloc((size_t)nsys * lv.mblk) == hipSuccess && c->lad_tail.alloc((size_t)nsys * lv.tail) == hipSuccess && c->lad_st.alloc((size_t)2 * LADDER_MAX) == hipSuccess;
    if (ok && sharded(c)) {      // what a pass all-reduces: [LADDER_MAX][4] slice sums of the step kernel, then [LADDER_MAX][NS + 1, padded] camera block + p.q of the operator pass
        const size_t n = (size_t)LADDER_MAX * (4 + lad_redop_stride(L.NS));
        const bool fresh_r = c->lad_red.n < n || !c->lad_red.p;
        ok = c->lad_red.alloc(n) == hipSuccess;
        if (ok && fresh_r) ok = hipMemsetAsync(c->lad_red.p, 0, sizeof(double) * c->lad_red.n, c->stream) == hipSuccess;      // slots of systems that are not live are summed too: keep them finite
    }
    if (!ok) {
        (void)hipGetLastError();
        c->lad_vec.release(); c->lad_qh.release(); c->lad_cam.release(); c->lad_part.release();
        std::fprintf(stderr, "[i3d] the slabs of the damping ladder (%d systems x %.1f MB) do not fit the device: this outer iteration runs the serial trust-region loop\n", nsys,
                     (double)(6 * vec + qh + cam) * 4.0 / 1e6);
        return 1;
    }
    return I3D_OK;
}
// the six vector kinds of the slab: kind * ladder_max * vec + system * vec
enum { LV_X = 0, LV_R, LV_P, LV_Z, LV_U, LV_Q };
static float* lad_vecp(const i3d_context* c, int kind, int sys = 0) { return c->lad_vec.p + ((size_t)kind * c->ladder_max + sys) * c->lad.vec; }

// B systems (J^T W J + D_j^2) y = b, j = 0 .. B-1 (the LM diagonals of a ladder batch, k_lm_begin_lad), iterated in LOCK STEP: pass `it` of the loop is iteration `it` of
// every system still running.  Per pass: ONE k_pcg_dir3 launch and ONE k_pcg_step3 launch over the live systems (blockIdx.y), and the operator — the rows are streamed
// once per GROUP of up to 3 live systems (k_eg_tile_mr) instead of once per system.  Every system keeps its own scalar state, stopping rule and host ring; a system
// that has stopped is dropped from the launches one pass after the host saw its flag.  The arithmetic of a system is that of pcg_solve_fused on it alone.
static int pcg_solve_ladder(i3d_context* c, const i3d_optimizer_config& cfg, const OptParams& p, int B, const PcgState** finals, int* passes_out) {
    hipStream_t s = c->stream;
    const Layout L = layout_of(c);
    const int K = c->K; const size_t to = L.tail_off; const Seg2 own = L.own;
    RowView r = c->row_view(); TilePlan tp = c->tile_plan();
    LadVec lv = c->lad;
    float* const X0 = lad_vecp(c, LV_X); float* const R0 = lad_vecp(c, LV_R); float* const P0 = lad_vecp(c, LV_P); float* const Z0 = lad_vecp(c, LV_Z);
    float* const U0 = lad_vecp(c, LV_U); float* const Q0 = lad_vecp(c, LV_Q);
    PcgState* const st2 = c->lad_st.p;
    const int NSP = (L.NS + 3) & ~3;
    double* const step_part0 = c->lad_part.p; double* const pq_part0 = step_part0 + LAD_PART_STEP; double* const d2_part0 = pq_part0 + LAD_PART_PQ;
    // (read per solve: tests and A/B runs switch them inside one process)
    const bool use_mr = [] { const char* e = std::getenv("I3D_LADDER_MR"); return !(e && e[0] == '0'); }();       // 0: the single-system operator once per system (A/B and parity runs)
    const bool mr1 = [] { const char* e = std::getenv("I3D_LADDER_MR1"); return e && e[0] == '1'; }();            // 1: a lone live system goes through k_eg_tile_mr<1> as well
    const int group_cap = [] { const char* e = std::getenv("I3D_LADDER_GROUP"); const int v = e ? std::atoi(e) : 3; return v < 1 ? 1 : (v > 3 ? 3 : v); }();
    const int mr_cap = std::min(group_cap, eg_tile_mr_max_systems(K));
    const bool mr_ok = use_mr && mr_cap >= 1 && tp.T == 512 && tp.hp_off != nullptr && r.slots == 5;
    // Sharded (exchanges as launches of the transport: RCCL, the rank simulation).  Per pass and BATCH, not per system:
    //   k_lad_reduce_step + ONE all-reduce of [LADDER_MAX][4] doubles (the slice sums of every live system) in front of k_pcg_dir3_lad,
    //   the rim of u = S p of every live system pushed after it (one exchange per system: the transport's lists are per vector),
    //   k_lad_reduce_op + ONE all-reduce of [LADDER_MAX][6K + 10] doubles (camera block + p.q of every live system) behind the operator.
    // The slots of systems that have stopped ride along unused: the message size does not depend on which systems are live, so the ranks' collectives always match.
    const bool sh = sharded(c);
    // (without the multi-system pass — I3D_LADDER_MR=0 — a sharded batch streams its rows once per system through k_eg_tile<..., GHOSTS>, like a single-rank one: the exchanges stay batched)
    double* const red4 = sh ? c->lad_red.p : nullptr;
    const int rstride = (int)lad_redop_stride(L.NS);
    double* const redop = sh ? c->lad_red.p + (size_t)LADDER_MAX * 4 : nullptr;
    { TimedScope t(c, I3D_K_VECTOR);
      for (int j = 0; j < B; ++j) { CTX_HIP(c, hipMemsetAsync(X0 + (size_t)j * lv.vec, 0, sizeof(float) * L.NP, s));
                                    CTX_HIP(c, hipMemcpyAsync(R0 + (size_t)j * lv.vec, c->v_b.p, sizeof(float) * L.NP, hipMemcpyDeviceToDevice, s)); }
      launch_pcg_init_lad(s, st2, B, cfg.pcg_fixed_iterations, 500, c->d_lm.p); }
    Step3Args a; std::memset(&a, 0, sizeof(a));
    auto c4 = [&](const float* v) { return reinterpret_cast<const float4*>(v + own.off0); };
    auto m4 = [&](float* v) { return reinterpret_cast<float4*>(v + own.off0); };
    a.nq = own.n >> 2; a.chunk4 = (int)((own.off1 - own.off0) >> 2);
    a.p = c4(P0); a.qacc = c4(Q0); a.x = m4(X0); a.r = m4(R0); a.b = c4(c->v_b.p); a.z = m4(Z0); a.cm = c4(c->v_cm.p); a.lm = c->d_lm.p;
    a.ext_off = tp.ext_off; a.ext_pos = tp.ext_pos; a.qh = reinterpret_cast<const float2*>(c->lad_qh.p); a.e0 = (int)own.off0;
    a.pq_partials = pq_part0; a.d2_partials = d2_part0; a.n_pq = 0; a.n_d2 = 0;
    a.n_slice_wg = pcg_step3_slice_wgs(own.n, 0);
    a.sharded = 0; a.lad_sys = 0; a.redop = redop; a.redop_stride = rstride;
    a.K = K; a.fix_poses = p.fix_poses; a.fix_intr = p.fix_intr; a.fix_dist = p.fix_dist;
    a.cam_partials = c->lad_cam.p; a.n_cam = 0; a.cam_stride = NSP; a.Mblk = c->lad_mblk.p;
    a.tp = P0 + to; a.tx = X0 + to; a.tr = R0 + to; a.tb = c->v_b.p + to; a.tD2 = c->lad_tail.p; a.tz = Z0 + to; a.tS = c->v_S.p + to;
    a.step_partials = step_part0;
    std::vector<int> live(B); for (int j = 0; j < B; ++j) live[j] = j;
    auto set_slots = [&]() { for (int q = 0; q < LADDER_MAX; ++q) lv.sysid[q] = live[q < (int)live.size() ? q : (int)live.size() - 1]; };
    // the operator on the live systems: groups of <= mr_cap systems share one stream of the rows; returns the workgroups per system (p.q partials / camera rows)
    auto rows_apply = [&](int parity, bool with_dot) -> int {
        int n = 0; const int nl = (int)live.size();
        if (sh) { TimedScope t(c, I3D_K_COMM);      // the rim of every live system's operator input: ONE message per peer, nl values per entry
                  if (c->comm->push_halo_multi(U0, lv.vec, live.data(), nl, c->halo, s)) return -1; }
        if (mr_ok && (nl > 1 || mr1 || sh)) {
            const int groups = (nl + mr_cap - 1) / mr_cap;
            int at = 0;
            for (int g = 0; g < groups; ++g) {
                const int ng = (nl - at + (groups - g) - 1) / (groups - g);        // balanced: 4 -> 2 + 2, 5 -> 3 + 2
                TimedScope t(c, ng == 1 ? I3D_K_EG_PASS : (ng == 2 ? I3D_K_EG_MR2 : I3D_K_EG_MR3));
                n = launch_eg_tile_mr(s, r, p, tp, ng, live.data() + at, U0, Q0, c->lad_qh.p, with_dot ? pq_part0 : nullptr, c->lad_cam.p, NSP, st2 + parity, lv);
                if (n <= 0) return -1;
                at += ng; ++c->lad_streams;
            }
        } else {
            for (int j : live) {
                TilePlan tj = tp; tj.qh = c->lad_qh.p + (size_t)j * lv.qh;
                TimedScope t(c, I3D_K_EG_PASS);
                n = launch_eg_tile(s, r, p, U0 + (size_t)j * lv.vec, tj, nullptr, Q0 + (size_t)j * lv.vec, with_dot ? pq_part0 + (size_t)j * lv.part : nullptr, st2 + 2 * j + parity,
                                   c->lad_cam.p + (size_t)j * lv.cam, NSP);
                ++c->lad_streams;
            }
        }
        c->lad_system_passes += nl;
        if (sh) {          // this rank's [camera block | p.q (+ D^2 p^2 of its slice)] of every live system -> summed over the ranks, one message
            { TimedScope t(c, I3D_K_VECTOR); launch_lad_reduce_op(s, nl, c->lad_cam.p, n, NSP, L.NS, pq_part0, with_dot ? n : 0, d2_part0, with_dot ? a.n_d2 : 0, redop, rstride, lv); }
            if (allreduce(c, redop, (size_t)LADDER_MAX * rstride)) return -1;
        }
        return n;
    };
    // sharded: the four slice sums of every live system, summed over the ranks in front of the boundary kernel
    auto step_sums = [&](int nl, int n_slice) -> int {
        if (!sh) return I3D_OK;
        { TimedScope t(c, I3D_K_VECTOR); launch_lad_reduce_step(s, nl, step_part0, n_slice, red4, lv); }
        return allreduce(c, red4, (size_t)LADDER_MAX * 4);
    };
    int n_step = 0;
    set_slots();
    { TimedScope t(c, I3D_K_VECTOR); a.cur = st2; n_step = launch_pcg_step3_lad(s, 0 /*init*/, B, a, lv); }
    const int seq0 = c->pcg_seq;
    for (int j = 0; j < B; ++j) finals[j] = st2 + 2 * j;
    int it = 1;
    for (;; ++it) {
        set_slots();
        const int nl = (int)live.size();
        { const int rc = step_sums(nl, a.n_slice_wg); if (rc) return rc; }
        { TimedScope t(c, I3D_K_VECTOR);
          a.n_d2 = launch_pcg_dir3_lad(s, it == 1, nl, own, to, L.NS, Z0, P0, c->v_S.p, U0, c->lad_tail.p, c->v_cm.p, c->d_lm.p, step_part0, n_step, d2_part0, st2, (it + 1) & 1, c->d_flags, seq0 + it, lv,
                                       a.n_slice_wg, red4); }
        { const int n = rows_apply(it & 1, true); if (n < 0) return ctx_fail(c, I3D_ERR_STATE, "pcg_solve_ladder: the multi-system operator pass could not be launched"); a.n_pq = n; a.n_cam = n; }
        a.cur = st2 + (it & 1);
        if (it % 10 != 0) { TimedScope t(c, I3D_K_VECTOR); n_step = launch_pcg_step3_lad(s, 1, nl, a, lv); }
        else {                                                                   // residual_reset_period: r = b - A x instead of r -= alpha q
            { TimedScope t(c, I3D_K_VECTOR); launch_pcg_step3_lad(s, 2, nl, a, lv);
              for (int j : live) { float* xj = X0 + (size_t)j * lv.vec; float* uj = U0 + (size_t)j * lv.vec;
                                   launch_mul2(s, own, c->v_S.p, xj, uj); launch_mul(s, L.NS, c->v_S.p + to, xj + to, uj + to); } }
            { const int n = rows_apply(it & 1, false); if (n < 0) return ctx_fail(c, I3D_ERR_STATE, "pcg_solve_ladder: the multi-system operator pass could not be launched"); a.n_cam = n; }
            { TimedScope t(c, I3D_K_VECTOR); n_step = launch_pcg_step3_lad(s, 3, nl, a, lv); }
        }
        {   // The boundary of THIS pass (k_pcg_dir3_lad of pass `it`, queued above), system by system.  Round 5 looked one pass further back (the boundary of pass it-1, after
            // queueing pass it): a system that had stopped was carried through TWO more passes — its slot in a stream of the rows (a dead system still stages its inputs, and
            // keeps the launch at its wider variant) — 103.6 system passes per iteration on the driver's protocol against the 83.5 the PCG iteration counts add up to.  Waiting
            // for this pass's own boundary costs nothing on the device: it is queued behind pass it-1, whose operator and step kernels are still running or queued when the host
            // gets here, and pass it+1 is queued while the operator of pass `it` runs.  A stopped system is carried through ONE pass (the rest of this one).
            const int want = seq0 + it;
            std::vector<int> still;
            for (int j : live) {
                volatile int* ring = c->h_flags + 4 * j + 2 * (want & 1);
                const double t_wait = now_s();
                while (__atomic_load_n((int*)&ring[0], __ATOMIC_ACQUIRE) != want) {
                    if (now_s() - t_wait > 30.0) { CTX_HIP(c, sync_stream(c)); if (__atomic_load_n((int*)&ring[0], __ATOMIC_ACQUIRE) != want) return ctx_fail(c, I3D_ERR_HIP, "pcg_solve_ladder: the device stopped publishing its state"); }
                }
                if (!__atomic_load_n((int*)&ring[1], __ATOMIC_ACQUIRE)) still.push_back(j);
                else finals[j] = st2 + 2 * j + (it & 1);       // boundary `it` wrote the terminal state into this pass's buffer; no later boundary copies it into the other
            }
            live.swap(still);
            if (live.empty()) break;
        }
        if (it > 520) break;
    }
    if (passes_out) *passes_out = it;
    c->pcg_seq = seq0 + PCG_SEQ_STRIDE;
    for (int j : live) finals[j] = st2 + 2 * j + (it & 1);       // (only after the pass limit: the state of the last pass queued)
    { const int lrc = ctx_launch_check(c); if (lrc) return lrc; }
    return I3D_OK;
}

// wait for record `idx` of the current solve (mapped host memory, written by the LM kernels with a release store of its sequence number)
static int wait_record(i3d_context* c, int idx, int seq, LmRecord& out) {
    LmRecord* const r = c->h_lmrec + idx;
    const double t0 = now_s();
    while (__atomic_load_n(&r->seq, __ATOMIC_ACQUIRE) != seq) {
        if (now_s() - t0 > 60.0) {
            CTX_HIP(c, sync_stream(c));
            if (__atomic_load_n(&r->seq, __ATOMIC_ACQUIRE) != seq) return ctx_fail(c, I3D_ERR_HIP, "lm_solve: the device stopped publishing the state of the trust-region loop");
        }
    }
    out = *r;
    return I3D_OK;
}

// NLSSolver::solve on the assembled rows.  Updates the device unknowns and the host camera when a step is accepted.
// The trust-region loop itself runs on the device (lm_kernels.hip); the host queues, per attempt,
//     k_lm_begin | PCG passes (polling the pass flags) | k_candidate | k_cand_frames | k_build<false> | k_lm_decide | k_accept
// and reads ONE record per attempt from mapped host memory — the record of attempt k-1 while the solve of attempt k is already queued (its kernels
// return at once when the loop has ended).  No stream synchronisation inside the loop; one at the end (accepted camera back to the host).
static int lm_solve(i3d_context* c, const i3d_optimizer_config& cfg, OptParams& p, i3d_iteration_stats* st, double initial_radius) {
    hipStream_t s = c->stream;
    const Layout L = layout_of(c);
    const int N = c->N, K = c->K, NP = (int)L.NP, NS = L.NS;
    GridView g = c->grid_view(); RowView r = c->row_view();
    if (cfg.lm_steps > LM_REC_SLOTS - 2) return ctx_fail(c, I3D_ERR_CAPACITY, "optimize: more than 62 LM steps per outer iteration");
    { TimedScope t(c, I3D_K_VECTOR); launch_freemask(s, r, p, c->v_mask.p); }
    // candidate arrays mirror x outside the work list (fixed parameters are read through them by the cost kernel)
    CTX_HIP(c, hipMemcpyAsync(c->xc_sdf.p, c->x_sdf.p, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice, s));
    CTX_HIP(c, hipMemcpyAsync(c->xc_alb.p, c->x_alb.p, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice, s));
    // column norms -> Jacobi scaling (computed once, TrustRegionMinimizer::Init); the camera blocks stay on the device for k_lm_begin
    // (one stream of the rows serves both on one rank: gradcol.hip; I3D_GRADCOL=0 keeps the two passes)
    const bool one_stream = one_stream_gradcol(c);
    auto after_colnorm = [&]() -> int {
        CTX_HIP(c, hipMemcpyAsync(c->d_cam_c.p, c->d_shared.p, sizeof(double) * (size_t)NS, hipMemcpyDeviceToDevice, s));
        CTX_HIP(c, hipMemcpyAsync(c->d_cam_H.p, c->d_blocks.p, sizeof(double) * ((size_t)21 * K + 25), hipMemcpyDeviceToDevice, s));
        { int rc2 = allgather(c, c->v_c.p); if (rc2) return rc2; }     // the candidate point needs S everywhere
        { TimedScope t(c, I3D_K_VECTOR); launch_scale_from_colnorm(s, NP, c->v_c.p, c->v_mask.p, c->v_S.p, c->v_cm.p); }
        return I3D_OK;
    };
    int rc = I3D_OK;
    if (one_stream) { rc = run_gradcol(c, p, after_colnorm); if (rc) return rc; }
    else {
        rc = run_pass(c, PASS_COLNORM, p, nullptr, c->v_c.p); if (rc) return rc;
        rc = after_colnorm(); if (rc) return rc;
        // gradient b = S J^T W r and initial cost
        rc = run_pass(c, PASS_GRAD, p, nullptr, c->v_acc.p); if (rc) return rc;
    }
    { TimedScope t(c, I3D_K_VECTOR); launch_mul(s, NP, c->v_S.p, c->v_acc.p, c->v_b.p); }
    // initial cost -> d_scal[16]: from the residuals the assembly already holds (I3D_COST0=0: the residual-only pass at the unchanged point, as up to round 4)
    if (env_off("I3D_COST0")) { rc = eval_cost_launch(c, p, false, c->d_frames.p); if (rc) return rc; }
    else if (!one_stream) { TimedScope t(c, I3D_K_VECTOR); launch_set_double(s, c->d_scal.p + 16, c->cost_at_build); }      // (one stream: k_eg_gradcol left it there)
    rc = count_above_dev(c, c->v_acc.p, c->v_mask.p, 1e-10f, c->d_scal.p + 8); if (rc) return rc;   // gradient_tolerance: free entries of g = J^T W r above 1e-10 (max-norm test)
    rc = dot_dev(c, c->v_mask.p, c->v_mask.p, c->d_scal.p + 9); if (rc) return rc;           // free parameters
    {   // camera unknowns of the current point for the candidate kernel (staged in pinned memory: the copy is asynchronous)
        double* xs = c->h_pinned;
        for (int i = 0; i < 6 * K; ++i) xs[i] = c->poses[i];
        for (int i = 0; i < 4; ++i) xs[6 * K + i] = c->intr[i];
        for (int i = 0; i < 5; ++i) xs[6 * K + 4 + i] = c->dist[i];
        CTX_HIP(c, hipMemcpyAsync(c->d_xshared.p, xs, sizeof(double) * NS, hipMemcpyHostToDevice, s));
    }
    const int seq0 = c->lm_seq; c->lm_seq += LM_REC_SLOTS;
    LmState* const lm = c->d_lm.p;
    launch_lm_init(s, lm, c->d_scal.p + 16, c->d_scal.p + 8, c->d_scal.p + 9, initial_radius, c->d_lmrec, seq0);

    static const bool legacy = [] { const char* e = std::getenv("I3D_PCG_LEGACY"); return e && e[0] == '1'; }();
    // three launches per pass on one rank with the tiled operator; the six-launch sequence when sharded, untiled, or asked for (A/B runs)
    bool fused = c->tile_ok && !legacy;
    if (fused && sharded(c)) { P2PDev probe; fused = c->comm->fused_exchange(&probe) && L.NS + 2 <= probe.L.red_cap; }      // else: the six-launch pass with separate exchange launches (RCCL, or a rim that exceeds a mailbox)
    int attempts = 0; bool ended = false, accepted = false;
    // one record consumed: statistics + whether the solve is over
    auto consume = [&](const LmRecord& rec) {
        if (rec.kind == 0) {
            if (st) { st->cost_initial = rec.cost; st->cost_final = rec.cost; st->free_parameters = (int64_t)(rec.nfree + 0.5); st->termination = rec.final_ ? 1 : 0; st->final_radius = initial_radius; }
            c->last_sizes[5] = (long long)(rec.nfree + 0.5);
        } else if (rec.kind == 2) { if (st) st->termination = 1; }                            // the radius ran out before the attempt (LevenbergMarquardtStrategy)
        else {
            if (st) {
                st->lm_iterations = attempts + 1;
                if (attempts < 50) { st->pcg_iterations[attempts] = rec.pcg_it; st->step_accepted[attempts] = rec.accepted; }
                st->final_radius = rec.radius_after; st->termination = rec.termination;
                if (rec.accepted) { st->cost_final = rec.cost; st->successful_steps += 1; }
            }
            if (cfg.verbose) std::printf("  [i3d LM] it %d cand %.9e model %.3e rho %.4f radius -> %.3e cg %d%s\n", attempts + 1, rec.cand_cost, rec.model_change, rec.rel, rec.radius_after, rec.pcg_it,
                                         rec.accepted ? " accepted" : "");
            accepted = rec.accepted != 0;
            ++attempts;
        }
        if (rec.final_) ended = true;
    };
    const int dbg_invalid = [] { const char* e = std::getenv("I3D_DEBUG_INVALID_ATTEMPT"); return e ? std::atoi(e) : -1; }();      // tests: this attempt's step counts as invalid (read per solve)
    // The damping ladder: batches of consecutive attempts are SOLVED together (k_lm_begin_lad: the radii a run of rejections leads to; pcg_solve_ladder: lock step, the
    // rows streamed once per group of systems) and then DECIDED one after the other by the same k_lm_decide — results, attempts, accept sequence and PCG counts are
    // those of the serial loop (bit for bit in the bit-reproducible mode).  Batch depth: what the previous outer iteration needed (the reference restarts at radius 1e4
    // every time, optimizer.cpp:138, so the count barely moves), doubling while everything is rejected.  An invalid step (radius halved instead of divided) puts a
    // batch out of step: the attempt behind it is solved again on its own (LmRecord kind 3).
    // (sharded: with the exchanges as launches of the transport — `fused` there means the in-kernel mailbox exchanges, which keep the serial loop)
    bool ladder = c->ladder_max > 1 && c->tile_ok && !legacy && (sharded(c) ? !fused : fused) && c->plan_T() == 512 && c->tile_plan().hp_off != nullptr && c->slots == 5;
    if (ladder && alloc_ladder(c) != I3D_OK) {      // (sharded: every rank holds the same list and takes the same decision unless its device alone is short of memory — then the collectives of the ranks no longer match and the run ends in the transport's time-out)
        ladder = false;
    }
    if (ladder) {
        const LadVec& lv = c->lad;
        { LmRecord rec; rc = wait_record(c, 0, seq0, rec); if (rc) return rc; consume(rec); }      // the initial tests
        // Batch depth.  First batch: what the last two outer iterations needed (the larger: one system too many costs its few PCG passes at the marginal price of a
        // shared stream, one too few costs a second batch); without history 2, doubling while everything is rejected.  With history, a batch that ends without an
        // accepted step is followed by batches of 2: the accepting attempt is near (a second full-depth batch wasted 5 of its 6 systems, profiles/r05_ladder_policy.json).
        const bool warm = c->ladder_hint > 0;
        int k = 0, prevB = 0; bool after_resync = false;
        while (k < cfg.lm_steps && !ended) {
            int B = after_resync ? 1 : (prevB == 0 ? (warm ? std::max(c->ladder_hint, c->ladder_hint_prev) : 2) : (warm ? 2 : 2 * prevB));
            B = std::max(1, std::min(B, std::min(c->ladder_max, cfg.lm_steps - k)));
            after_resync = false; prevB = B;
            { TimedScope t(c, I3D_K_VECTOR);
              launch_lm_begin_lad(s, lm, B, K, p.fix_poses, p.fix_intr, p.fix_dist, c->d_cam_c.p, c->d_cam_H.p, c->lad_mblk.p, lv.mblk, c->v_c.p + L.tail_off, c->v_S.p + L.tail_off, c->lad_tail.p, lv.tail,
                                  c->d_lmrec + 1 + k, seq0 + 1 + k); }
            const PcgState* fin[LADDER_MAX] = {nullptr};
            rc = pcg_solve_ladder(c, cfg, p, B, fin, nullptr); if (rc) return rc;
            ++c->lad_batches;
            for (int j = 0; j < B && sharded(c); ++j) { rc = allgather(c, lad_vecp(c, LV_X, j)); if (rc) return rc; }      // the candidate points are replicated: every rank needs the whole step of every system
            for (int j = 0; j < B; ++j) {      // the decision chain of every system, in ladder order; everything behind the deciding attempt returns at once
                { TimedScope t(c, I3D_K_VECTOR);
                  launch_candidate(s, g, r, K, -1.0f, lad_vecp(c, LV_X, j), c->v_S.p, c->d_xshared.p, c->xc_sdf.p, c->xc_alb.p, c->d_xcshared.p, c->d_scal.p + 4, c->v_mask.p, c->d_partials.p, lm);
                  launch_cand_frames(s, K, c->d_xcshared.p, c->d_frames.p, c->d_frames_cand.p, lm); }
                rc = eval_cost_launch(c, p, true, c->d_frames_cand.p, c->d_xcshared.p + 6 * K, lm); if (rc) return rc;
                { TimedScope t(c, I3D_K_VECTOR);
                  launch_lm_decide(s, lm, fin[j], c->d_scal.p + 4, c->d_scal.p + 16, k + j, cfg.lm_steps, c->d_lmrec + 1 + k + j, seq0 + 1 + k + j, j + 1, dbg_invalid == k + j ? 1 : 0);
                  launch_accept(s, g, r, c->xc_sdf.p, c->xc_alb.p, lm); }
            }
            int decided = 0;
            for (int j = 0; j < B && !ended; ++j) {
                LmRecord rec; rc = wait_record(c, 1 + k + j, seq0 + 1 + k + j, rec); if (rc) return rc;
                if (rec.kind == 3) {            // out of step: attempt k + j was solved with a radius the trust region did not reach — solve it again, alone
                    __atomic_store_n(&c->h_lmrec[1 + k + j].seq, 0, __ATOMIC_RELEASE);
                    ++c->lad_resyncs; after_resync = true; break;
                }
                consume(rec); ++decided;
            }
            if (ended && decided < B) c->lad_wasted += B - decided;
            k += decided;
        }
        if (attempts > 0) { c->ladder_hint_prev = c->ladder_hint; c->ladder_hint = attempts; }
        ended = true;        // (every record of the solve has been consumed)
    }
    int k = 0;
    for (; !ladder && k < cfg.lm_steps; ++k) {
        // attempt k, queued behind whatever attempt k-1 still has in flight.  k_lm_begin: radius test, 1/radius, LM diagonal of the camera tail, block-Jacobi
        // inverses of the damped camera blocks
        { TimedScope t(c, I3D_K_VECTOR);
          launch_lm_begin(s, lm, K, p.fix_poses, p.fix_intr, p.fix_dist, c->d_cam_c.p, c->d_cam_H.p, c->Minv_blocks.p, c->v_c.p + L.tail_off, c->v_S.p + L.tail_off, c->v_D2.p + L.tail_off, c->v_Minv.p + L.tail_off,
                          c->d_lmrec + 1 + k, seq0 + 1 + k);
          if (!fused) launch_lm_diag_dev(s, (int)L.tail_off, c->v_c.p, c->v_S.p, lm, c->v_D2.p, c->v_Minv.p); }      // (the three-launch pass recomputes S, D^2, M^-1 of the voxel unknowns from the column norms)
        const PcgState* ps = nullptr;
        rc = fused ? pcg_solve_fused(c, cfg, p, &ps) : pcg_solve(c, cfg, p, &ps); if (rc) return rc;
        // the pass flags this solve waited for were written after everything queued before it: record k (initial tests for k = 0, else attempt k-1) is there
        { LmRecord rec; rc = wait_record(c, k, seq0 + k, rec); if (rc) return rc; consume(rec); }
        if (ended) break;
        // candidate point (replicated: every rank needs the whole step), its keyframe constants and its cost, then the decision — all on the stream
        { int rc2 = allgather(c, c->v_x.p); if (rc2) return rc2; }
        { TimedScope t(c, I3D_K_VECTOR);
          launch_candidate(s, g, r, K, -1.0f, c->v_x.p, c->v_S.p, c->d_xshared.p, c->xc_sdf.p, c->xc_alb.p, c->d_xcshared.p, c->d_scal.p + 4, c->v_mask.p, c->d_partials.p, lm);
          launch_cand_frames(s, K, c->d_xcshared.p, c->d_frames.p, c->d_frames_cand.p, lm); }
        rc = eval_cost_launch(c, p, true, c->d_frames_cand.p, c->d_xcshared.p + 6 * K, lm); if (rc) return rc;
        { TimedScope t(c, I3D_K_VECTOR);
          launch_lm_decide(s, lm, ps, c->d_scal.p + 4, c->d_scal.p + 16, k, cfg.lm_steps, c->d_lmrec + 1 + k, seq0 + 1 + k, -1, dbg_invalid == k ? 1 : 0);
          launch_accept(s, g, r, c->xc_sdf.p, c->xc_alb.p, lm); }
    }
    if (!ended) { LmRecord rec; rc = wait_record(c, k, seq0 + k, rec); if (rc) return rc; consume(rec); }      // the last attempt's record (step limit)
    if (st) st->num_attempts = std::min(attempts, 50);
    // the accepted camera back to the host (the next outer iteration rebuilds its keyframe constants from it)
    CTX_HIP(c, hipMemcpyAsync(c->h_pinned, c->d_xcshared.p, sizeof(double) * NS, hipMemcpyDeviceToHost, s));
    CTX_HIP(c, sync_stream(c));
    if (sharded(c) && c->comm->health(s)) return ctx_fail(c, I3D_ERR_COMM, "lm_solve: a peer-to-peer exchange timed out (a rank stopped taking part)");
    if (accepted) {
        for (int i = 0; i < 6 * K; ++i) c->poses[i] = c->h_pinned[i];
        for (int i = 0; i < 4; ++i) c->intr[i] = c->h_pinned[6 * K + i];
        for (int i = 0; i < 5; ++i) c->dist[i] = c->h_pinned[6 * K + 4 + i];
    }
    { const int lrc = ctx_launch_check(c); if (lrc) return lrc; }
    return I3D_OK;
}

int optimize(i3d_context* c, const i3d_optimizer_config& cfg, i3d_iteration_stats* stats) {
    if (!c->have_grid || cfg.iterations < 1) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "optimize: no grid or iterations < 1");   // optimizer.cpp:113-114
    CTX_HIP(c, hipSetDevice(c->device));
    double carried_radius = 1e4;
    for (int itr = 0; itr < cfg.iterations; ++itr) {
        i3d_iteration_stats local; std::memset(&local, 0, sizeof(local));
        i3d_iteration_stats* st = stats ? &stats[itr] : &local;
        std::memset(st, 0, sizeof(*st));
        OptParams p;
        const double t0 = now_s();
        int rc = assemble(c, cfg, itr, p, st); if (rc) return rc;
        const double t1 = now_s();
        // t_add_end: device time from the start of the assembly to the end of the residual collection (events on the stream)
        st->time_add = (c->t_add_end >= 0.0 && c->t_add_end <= t1 - t0) ? c->t_add_end : t1 - t0; st->time_build = (t1 - t0) - st->time_add;
        if (c->n_active > 0) { rc = lm_solve(c, cfg, p, st, cfg.carry_trust_radius ? carried_radius : 1e4); if (rc) return rc;
                               if (st->final_radius > 0.0) carried_radius = st->final_radius; }
        const double t2 = now_s();
        st->time_solve = t2 - t1;
        if (cfg.verbose) std::printf("[i3d] itr %d rows %lld/%lld/%lld/%lld valid %lld cost %.9e -> %.9e (add %.3f ms, solve %.3f ms)\n", itr,
                                     (long long)st->rows[0], (long long)st->rows[1], (long long)st->rows[2], (long long)st->rows[3], (long long)st->valid_voxels,
                                     st->cost_initial, st->cost_final, st->time_add * 1e3, st->time_solve * 1e3);
        timing_flush(c);
    }
    c->assembled = false;
#ifdef I3D_MR_PHASES
    mr_phase_report_now();      // (variant build only: tools/build_variant.sh)
#endif
#ifdef I3D_MR_BLOCKTIME
    mr_blocktime_report_now();
#endif
    return I3D_OK;
}

// ---- parity probes ------------------------------------------------------------------------------------------------
static int list_maps(i3d_context* c, std::vector<int>& rank, std::vector<int>& alist) {
    rank.resize(c->N); alist.resize(c->A > 0 ? c->A : 1);
    CTX_HIP(c, hipMemcpy(rank.data(), c->rank.p, sizeof(int) * (size_t)c->N, hipMemcpyDeviceToHost));
    if (c->A > 0) CTX_HIP(c, hipMemcpy(alist.data(), c->alist.p, sizeof(int) * (size_t)c->A, hipMemcpyDeviceToHost));
    return I3D_OK;
}
static int to_visit_order(i3d_context* c, const float* dev_vec, double* out) {
    const int N = c->N, A = c->A, K = c->K, NS = 6 * K + 9, ch = c->chunk, NP = 2 * ch + NS;      // single-rank layout [sdf ch | alb ch | camera]
    std::vector<float> h(NP); std::vector<int> rank, alist;
    CTX_HIP(c, hipMemcpy(h.data(), dev_vec, sizeof(float) * (size_t)NP, hipMemcpyDeviceToHost));
    int rc = list_maps(c, rank, alist); if (rc) return rc;
    for (int i = 0; i < 2 * N + NS; ++i) out[i] = 0.0;
    for (int a = 0; a < A; ++a) { const int v = rank[alist[a]]; out[v] = h[a]; out[N + v] = h[ch + a]; }
    for (int i = 0; i < NS; ++i) out[2 * N + i] = h[2 * ch + i];
    return I3D_OK;
}

int normal_eq_debug(i3d_context* c, double* gradient, double* jtj_diag, double* cost) {
    if (!c->assembled) return ctx_fail(c, I3D_ERR_STATE, "debug: call i3d_debug_assemble first");
    CTX_HIP(c, hipSetDevice(c->device));
    OptParams p = c->last_params;
    launch_freemask(c->stream, c->row_view(), p, c->v_mask.p);
    int rc;
    if (jtj_diag) { rc = run_pass(c, PASS_COLNORM, p, nullptr, c->v_c.p); if (rc) return rc; CTX_HIP(c, sync_stream(c)); rc = to_visit_order(c, c->v_c.p, jtj_diag); if (rc) return rc; }
    if (gradient) { rc = run_pass(c, PASS_GRAD, p, nullptr, c->v_acc.p); if (rc) return rc; CTX_HIP(c, sync_stream(c)); rc = to_visit_order(c, c->v_acc.p, gradient); if (rc) return rc; }
    if (cost) { rc = eval_cost(c, p, false, c->d_frames.p, cost); if (rc) return rc; }
    return I3D_OK;
}

int jtj_apply_debug(i3d_context* c, const double* x, double* y) {
    if (!c->assembled) return ctx_fail(c, I3D_ERR_STATE, "debug: call i3d_debug_assemble first");
    CTX_HIP(c, hipSetDevice(c->device));
    const int N = c->N, A = c->A, K = c->K, NS = 6 * K + 9, ch = c->chunk, NP = 2 * ch + NS;
    OptParams p = c->last_params;
    std::vector<int> rank, alist; std::vector<float> h(NP, 0.0f), m(NP);
    int rc = list_maps(c, rank, alist); if (rc) return rc;
    launch_freemask(c->stream, c->row_view(), p, c->v_mask.p);
    CTX_HIP(c, sync_stream(c));
    CTX_HIP(c, hipMemcpy(m.data(), c->v_mask.p, sizeof(float) * (size_t)NP, hipMemcpyDeviceToHost));
    for (int a = 0; a < A; ++a) { const int v = rank[alist[a]]; h[a] = (float)x[v] * m[a]; h[ch + a] = (float)x[N + v] * m[ch + a]; }
    for (int i = 0; i < NS; ++i) h[2 * ch + i] = (float)x[2 * N + i] * m[2 * ch + i];
    CTX_HIP(c, hipMemcpy(c->v_u.p, h.data(), sizeof(float) * (size_t)NP, hipMemcpyHostToDevice));
    rc = run_pass(c, PASS_JTJP, p, c->v_u.p, c->v_acc.p); if (rc) return rc;
    CTX_HIP(c, sync_stream(c));
    return to_visit_order(c, c->v_acc.p, y);
}

}  // namespace i3d
