// Launch wrappers of the gfx950 kernels (definitions in *.hip).  All launches go to the given stream.
#pragma once
#include "common.hpp"
#include "p2p_device.hpp"

namespace i3d {

// ---- grid_kernels.hip -------------------------------------------------------------------------------------
struct HashTable { unsigned long long* keys; int* vals; unsigned int mask; };
void launch_sort_keys(hipStream_t st, int N, const int* keys_xyz /*[N][3] visit order*/, unsigned long long* sort_keys, int* iota);
void launch_permute_grid(hipStream_t st, int N, const int* perm, const int* keys_xyz, const double* sdf, const double* sdf_ref,
                         const double* alb, const float* w, const uint8_t* rgb,
                         int* cx, int* cy, int* cz, int* rank, double* sdf0, double* x_sdf, double* x_alb, float* f_sdf, float* f_alb,
                         float* weight, uchar4* color);
void launch_hash_build(hipStream_t st, int N, const int* cx, const int* cy, const int* cz, HashTable t);
void launch_nbr_build(hipStream_t st, int N, const int* cx, const int* cy, const int* cz, HashTable t, int* nbr);
void launch_classify(hipStream_t st, GridView g, OptParams p, int* active_flag);
void launch_compact(hipStream_t st, int N, const int* active_flag, const int* active_scan, const uint8_t* flags, int* aidx, int* alist, uint8_t* aflags);
// stable partition of every 512-entry block of the work list: entries that can own Eg rows (active, whole forward stencil stored) first.  Entries without rows then
// fill whole waves of the operator pass, which skip their row stream (k_group_rows); tiles (512 / 1024 entries) and rank slices keep their members
void launch_partition_blocks(hipStream_t st, GridView g, int A, int* alist, uint8_t* aflags, int* aidx);
void launch_group_rows(hipStream_t st, int A, const uint8_t* nrows, int* gmax);       // gmax[g] = max nrows over entries [64 g, 64 g + 64)
void launch_anbr(hipStream_t st, int N, int A, int Acap, const int* alist, const int* nbr, const int* aidx, int* anbr);
void launch_scatter_sh(hipStream_t st, int N, const int* rank, const double* sh_visit /*[N][9]*/, float* sh /*[9][N]*/);
void launch_gather_visit(hipStream_t st, int N, const int* rank, const double* x_sdf, const double* x_alb, double* out_sdf, double* out_alb);
void launch_update_fields(hipStream_t st, int N, const int* rank, const double* sdf_ref, const double* alb, const uint8_t* rgb,
                          double* x_sdf, double* x_alb, float* f_sdf, float* f_alb, uchar4* color);

// ---- observe.hip ------------------------------------------------------------------------------------------
void launch_observe(hipStream_t st, GridView g, RowView r, OptParams p, const FrameConst* frames, const unsigned* cull_mask = nullptr, bool prefilter = false);
// cull_kernels.hip: per-keyframe 8x8-block depth ranges; bounding spheres of the 64-entry groups of the compute list and their keyframe cull masks ([group][ceil(K/32)])
constexpr int CULL_BLOCK = 8, CULL_MAX_LEVELS = 12;          // depth-range pyramid: 8 x 8 pixel blocks at level 0, 2 x 2 reductions above
struct CullPyramid { int levels, cells; int bw[CULL_MAX_LEVELS], bh[CULL_MAX_LEVELS], off[CULL_MAX_LEVELS]; };     // cells = float2 entries per keyframe (all levels)
CullPyramid cull_pyramid(int w, int h);
void launch_depth_blocks(hipStream_t st, const FrameConst* frames, int K, int w, int h, float2* out);
void launch_group_cull(hipStream_t st, GridView g, RowView r, OptParams p, const FrameConst* frames, const float2* dblocks, float4* bounds, unsigned* mask);

// ---- build.hip --------------------------------------------------------------------------------------------
// with_jacobian: fills res/J/roww/rowfree + regulariser flags (assembly).  Otherwise evaluates the cost of the rows
// already assembled at the state (g.x_sdf, g.x_alb, frames, p) into cost_out (double, accumulated).
// cam9 (cost evaluation only, or null): intrinsics (4) + distortion (5) of the evaluated point in DEVICE memory, overriding p.intr / p.dist — the candidate
// camera of an LM attempt never visits the host (lm_kernels.hip).  lm (or null): skip the launch's work when the solve is already over.
void launch_build(hipStream_t st, GridView g, RowView r, OptParams p, const FrameConst* frames, bool with_jacobian, double* cost_out /* accumulated */, double* scratch,
                  const double* cam9 = nullptr, const LmState* lm = nullptr);
void launch_weight_sums(hipStream_t st, RowView r, GridView g, bool with_cost, double* sums13 /* accumulated: [0..3] weight sums, [4] Eg rows, [7] Ea rows ([5],[6] = [1],[2]), [8] active voxels,
                        [9..12] sum of w r^2 per row type over the rows with a free parameter */, double* scratch);

// ---- operator.hip -----------------------------------------------------------------------------------------
// All vectors are in work-list space: NP = 2A + 6K + 9.
enum PassMode { PASS_GRAD = 0, PASS_JTJP = 1, PASS_COLNORM = 2 };
// a rank's slice of a solver vector: [off0, off0 + n) of the sdf part and [off1, off1 + n) of the albedo part (n, offsets multiples of 4)
struct Seg2 { size_t off0, off1; int n; };
struct PassBuffers {
    float* C;            // [14][Acap] per-voxel-row-block column sums
    float* treg;         // [8][Acap]  tr, ts, ta[6]
    double* shared;      // [6K+9] pose/intr/dist accumulators (zeroed by the caller)
    double* blocks;      // COLNORM only: [21K + 10 + 15] upper triangles of the pose/intr/dist J^T W J blocks
    float* part; int part_stride;      // GRAD / COLNORM: one float row of camera totals per workgroup (summed in a fixed order by launch_sum_rows: no global atomics)
};
// gradient + column norms from one row stream (gradcol.hip): the staging planes and camera rows of BOTH passes
struct GradColBuffers {
    float* Cg; float* Cc;           // [14][Acap] column sums: J^T W r | diag(J^T W J)
    float* tregg; float* tregc;     // [8][Acap] regulariser terms of the two
    float* part; int part_stride;   // per workgroup: gradient row [6K | 9] at part, column-norm row [21K | 34] at part + col_off
    int col_off;
    double* cost_partials; double* cost_out;      // per-workgroup partials of the cost at this point (0.5 sum w r^2 over the rows with a free parameter); *cost_out += their sum
};
int  launch_eg_gradcol(hipStream_t st, GridView g, RowView r, OptParams p, GradColBuffers b);      // returns the rows written to b.part
int  launch_eg_pass(hipStream_t st, PassMode mode, GridView g, RowView r, OptParams p, const float* u /*[NP] or null*/, PassBuffers b, const PcgState* state);   // returns the rows written to b.part
void launch_sum_rows(hipStream_t st, PassMode mode, int K, const float* part, int nrows, int stride, double* shared, double* blocks);
void launch_gather(hipStream_t st, PassMode mode, RowView r, PassBuffers b, float* out /*[2A]*/);
int  launch_gather_tail(hipStream_t st, RowView r, PassBuffers b, float* out, const float* S, const float* D2, const float* v, double* dot_partials /* or, dot_atomic: the sum itself */,
                        bool dot_atomic, const PcgState* state);   // returns #partials
// dst[k] += sum_b partials[b*ncomp + k]: the second stage of every fp64 reduction (no same-address atomics from thousands of workgroups)
void launch_reduce_partials(hipStream_t st, const double* partials, int nblk, int ncomp, double* dst, const PcgState* state, bool assign = false);      // dst += sum (assign: dst = sum)
void launch_shared_finalize(hipStream_t st, size_t tail_off, int K, OptParams p, const double* shared, float* out /*[NP]*/, bool tail, const float* S, const float* D2,
                            const float* v, double* dot_out, const PcgState* state);

void launch_fill(hipStream_t st, int n, float* x, float v);
void launch_fill_d(hipStream_t st, int n, double* x, double v);
void launch_int_to_double(hipStream_t st, const int* src, double* dst);      // a device flag joins an fp64 all-reduce
void launch_mul(hipStream_t st, int n, const float* a, const float* b, float* out);                 // out = a*b
void launch_scale_from_colnorm(hipStream_t st, int n, const float* c, const float* freemask, float* S, float* cm);   // S = free ? 1/(1+sqrt(c)) : 0;  cm = free ? c : -1
void launch_lm_diag(hipStream_t st, int n, const float* c, const float* S, float inv_radius, float* D2, float* Minv_diag);  // D2 = clamp(c S^2)/radius, Minv = 1/(c S^2 + D2) (free) else 0
void launch_dot(hipStream_t st, int n, const float* a, const float* b, double* out /* accumulated */, double* scratch);
void launch_count_above2(hipStream_t st, Seg2 sg, const float* a, const float* m, float tol, double* out, double* scratch);
void launch_count_above(hipStream_t st, int n, const float* a, const float* m, float tol, double* out, double* scratch);
void launch_dot2(hipStream_t st, Seg2 sg, const float* a, const float* b, double* out /* accumulated */, double* scratch);
void launch_mul2(hipStream_t st, Seg2 sg, const float* a, const float* b, float* out);
void launch_freemask(hipStream_t st, RowView r, OptParams p, float* mask /*[NP]*/);

// fused PCG iteration, scalars resident in PcgState
void launch_pcg_init(hipStream_t st, PcgState* state, int fixed_iterations, int max_iterations, const LmState* lm = nullptr);      // lm->done: the solve starts finished
// mode: 0 init (z, r.z) | 1 x += a p, r -= a q, z, sums | 2 x only | 3 r = b - q(=A x), z, sums      (a rank's slice; off, n multiples of 4)
// S_for_inline_q != nullptr: `q` holds the raw accumulators of the tiled operator pass and q = S acc + D2 v is formed inside the kernel
int  launch_pcg_step(hipStream_t st, int mode, Seg2 sg, const float* p, const float* q, float* x, float* r, const float* b, const float* D2, const float* Minv,
                     float* z, const float* S_for_inline_q, double* partials /*[blocks][4]*/, PcgState* state);                               // returns #partials
void launch_pcg_tail_x(hipStream_t st, size_t tail_off, int K, const float* p, float* x, const PcgState* state);
void launch_pcg_tail_a(hipStream_t st, int mode, size_t tail_off, int K, const float* Minv_blocks, const float* p, const float* q, float* x, float* r, const float* b,
                       const float* D2, float* z, const double* partials, int nblk, PcgState* state,           // camera tail + Q-test + rho, beta;
                       double* shared_zero, int nzero, int* host_flags, int seq,                               // zeroes the camera accumulator, publishes (seq, done) to pinned memory
                       const P2PDev& pd);                                                                      // pd.on: sum the slice sums over the ranks in the kernel (p2p_device.hpp)
int  launch_pcg_direction(hipStream_t st, Seg2 sg, size_t tail_off, int ntail, const float* z, float* p, const float* S, float* u, const float* D2,
                          double* d2_partials /* or null */, const PcgState* state);   // p = z + beta p, u = S p (slice + ntail tail entries); returns the number of D^2 p^2 partials
void launch_pcg_tail_b(hipStream_t st, size_t tail_off, int K, OptParams p, double* shared, const double* pq_slice, const double* pq_partials, int nblk,
                       const double* pq_partials2, int nblk2, bool rowwise,
                       float* q, const float* S, const float* D2, const float* v, PcgState* state, const P2PDev& pd);                                        // camera tail of q, p.q, alpha (pd.on: [camera block | p.q] summed over the ranks in the kernel)
// ---- tile_pass.hip: the LDS-tiled operator pass of the PCG (single rank) ----------------------------------------------------------
struct TilePlan {                   // built once per outer iteration by launch_tile_plan
    unsigned* lnbr;                 // [9][Acap] local slots (uint16 pairs): 12 stencil neighbours read + 6 further in-tile sources
    float* eaw_sym;                 // [6][Acap] symmetric albedo-edge weights (pull form of the Ea rows)
    int* halo_idx; int* halo_cnt;   // [tiles][HMAX] foreign entries a tile's stencils reach (sorted, INT_MAX padded), [tiles] their number
    int* iota; int* ext_e; int* ext_pos;   // [tiles * HMAX] (entry, halo slot) pairs sorted by entry
    float* qh;                      // [tiles * HMAX][2] halo accumulators of one pass
    int* ext_off;                   // [chunk + 1] CSR offsets of the sorted pairs by entry (k_pcg_step3 folds the halo sums itself), or null
    int* overflow;                  // device flag: a halo did not fit -> use the untiled pass
    int T, hmax;                    // geometry of THIS plan: 512 / 1536 (two workgroups per CU) or 1024 / 2048 (the fallback when a tile's halo does not fit)
    int tile_first, ntiles_own;     // tiles this rank owns (all of them when not sharded)
    const int* ghost_tiles; int n_ghost;   // sharded: foreign tiles that hold ghost entries of this rank's compute list
    int det;                               // 1: fixed-order sums inside the workgroups of the operator pass as well (I3D_DETERMINISTIC=1, read per outer iteration)
    // halo PULL lists (k_tile_pull_plan; null = the halo sums of a tile are pushed with LDS atomics): for every halo slot of a tile the (column, lane) pairs of the tile
    // that contribute to it, CSR by halo slot, each segment sorted — the slot's owner thread adds them in that fixed order
    const unsigned short* hp_off;   // [tiles][hmax + 1]
    const unsigned short* hp_src;   // [tiles][4 * hmax]: (column << 10) | lane; column 0..8 sdf, 9..11 albedo, 12 the Er row value
    int pull;                       // 1: the single-system pass pulls its halo over the lists too (I3D_HALO_PULL=1 / the bit-reproducible mode); 0: the lists (if any) only serve the multi-system pass
};
inline int tile_plan_pull_cap(int hmax) { return 4 * hmax; }      // list entries per tile (= the LDS the pushed halo accumulators occupied)
int    tile_plan_T();                      // default geometry (I3D_EGT_TILE): entries per tile
int    tile_plan_tiles(int A);             // ... tiles of A entries, halo slots per tile
int    tile_plan_hmax();
inline int tile_plan_hmax_of(int T) { return T == 1024 ? 2048 : 1536; }
inline int tile_plan_tiles_of(int A, int T) { return (A + T - 1) / T; }
size_t tile_plan_temp_bytes(int ntiles);
hipError_t launch_tile_plan(hipStream_t st, RowView r, TilePlan t, void* temp, size_t temp_bytes);
void launch_eaw_sym(hipStream_t st, RowView r, TilePlan t, const int* cflag /* sharded: compute-list flags, else null */);   // after launch_build (reads the Ea weights it wrote)
// qacc[2 chunk] = J^T W J u on the voxel unknowns (raw), camera block added into `shared` (fp64), row-wise p.q partials; returns their number
// cam_partials != null: the camera block is NOT added into `shared`; workgroup w stores its float sums in cam_partials[w * cam_stride + 0..6K+9)
int  launch_eg_tile(hipStream_t st, RowView r, OptParams p, const float* u, TilePlan t, double* shared, float* qacc, double* pq_partials /* or null */,
                    const PcgState* state, float* cam_partials = nullptr, int cam_stride = 0);
void launch_ext_offsets(hipStream_t st, int n, const int* ext_e, int A, int* ext_off);
void eg_tile_launch_shape(const TilePlan& t, int K, int& blocks, int& tiles_per_block);      // workgroups / tiles per workgroup launch_eg_tile uses for this plan

// ---- tile_pass_mr.hip: the operator pass for up to 3 systems of a ladder batch in ONE stream of the rows (512-entry geometry with pull lists, 5 observation slots; sharded: own + ghost tiles) ----
struct LadVec;
int  eg_tile_mr_max_systems(int K);        // systems one launch can take at K keyframes (LDS): 3 at K = 200, 0 = the pass cannot run
int  launch_eg_tile_mr(hipStream_t st, RowView r, OptParams p, TilePlan t, int nsys, const int* sys, const float* u0, float* qacc0, float* qh0, double* pq0 /* or null */, float* cam0, int cam_stride,
                       const PcgState* st0 /* system 0's state of this pass's parity */, const LadVec& lv);      // returns the workgroups (= p.q partials / camera rows per system), 0 = not launched

// ---- pcg_fused.hip: the PCG iteration in three launches (single rank): k_pcg_dir3 | k_eg_tile | k_pcg_step3 -----------------------
// sharded three-launch pass over the peer-to-peer mailboxes: what k_pcg_dir3 / k_pcg_step3 need besides their single-rank arguments (the exchanges run INSIDE them)
struct ShardArgs {
    P2PDev pd; RimLists rim; int seq;      // seq: pass number of the host, identical on all ranks -> the epochs of the pass's exchanges (p2p_device.hpp)
    int n_slice_partials;                   // k_pcg_dir3: step partials [0, n) are sums over this rank's slice, the rest the (replicated) camera tail's
    int n_rim_wg;                           // k_pcg_dir3: workgroups that exchange the rim
    const float* zb; float* pb; float* ub; const float* cmb; int chunk;      // whole vectors (rim entries are addressed absolutely)
};
// ladder batch (common.hpp LADDER_MAX): system j's copy of every per-system array starts j * stride elements behind system 0's
struct LadVec {
    size_t vec;                             // solver vectors x, r, p, z, u, qacc (floats; a multiple of 4)
    size_t qh;                              // halo sums of the operator pass (floats; even)
    size_t cam;                             // camera partial rows of the operator pass (floats)
    size_t part;                            // fp64 partial sums: the step / p.q / D^2 p^2 regions of a system (doubles)
    size_t mblk;                            // block-Jacobi inverses of the camera blocks (floats)
    size_t tail;                            // LM diagonal of the camera tail (floats)
    int sysid[LADDER_MAX];                  // launch slot (blockIdx.y, or the slot of a multi-system operator pass) -> system
};
struct Step3Args {
    int nq; int chunk4;
    const float4* p; const float4* qacc; float4* x; float4* r; const float4* b; float4* z;
    const float4* cm; const LmState* lm;       // masked squared column norms (-1 = fixed) + the LM state holding 1 / trust-region radius of the attempt in flight: S, D^2 and M^-1 of a voxel unknown are recomputed from them (lm_from_colnorm)
    const int* ext_off; const int* ext_pos; const float2* qh; int e0;
    const double* pq_partials; int n_pq; const double* d2_partials; int n_d2;
    int n_slice_wg;
    int K; int fix_poses, fix_intr, fix_dist;
    const float* cam_partials; int n_cam; int cam_stride; const float* Mblk;
    const float* tp; float* tx; float* tr; const float* tb; const float* tD2; float* tz; const float* tS;
    double* step_partials;
    PcgState* cur;
    int sharded; ShardArgs sh;              // sharded: p.q and the camera block are summed over the ranks inside the kernel; d2_partials[n_d2] = the camera tail's D^2 p^2 (replicated, counted once)
    int lad_sys;                            // -1: 1 / radius from lm->inv_radius; >= 0: system of a ladder batch, lm->lad_inv_radius[lad_sys]
    const double* redop; int redop_stride;  // sharded ladder pass (exchanges as launches): [camera block 6K+9 | p.q] of system j, summed over the ranks, at redop + j * redop_stride; null otherwise
};
void launch_pcg_init3(hipStream_t st, PcgState* st2 /* [2] */, int fixed_iterations, int max_iterations, const LmState* lm = nullptr);
int  launch_pcg_dir3(hipStream_t st, bool init, Seg2 sg, size_t tail_off, int ntail, const float* z, float* p, const float* S, float* u, const float* D2, const float* cm, const LmState* lm,
                     const double* step_partials, int n_step, double* d2_partials, const PcgState* prev, PcgState* next, int* host_flags, int seq,
                     const ShardArgs* sa = nullptr);   // returns #d2 partials of the slice (sharded: the tail's follows at that index)
void launch_rim_u(hipStream_t st, const ShardArgs& sa, const PcgState* state);      // sharded residual-reset pass: the rim of u = S x
int  pcg_step3_slice_wgs(int n_entries, int cap = 0);      // cap > 0: at most this many (rank simulation on one device)
int  pcg_step3_tail_wgs(int K);
int  launch_pcg_step3(hipStream_t st, int mode /* 0 init | 1 normal | 2 x only | 3 reset */, Step3Args a);                                          // returns #[4]-partials
void launch_halo_fold(hipStream_t st, RowView r, TilePlan t, float* qacc, const PcgState* state);
// ladder batch: the same kernels over several systems at once (blockIdx.y; the arithmetic of a system does not change)
void launch_pcg_init_lad(hipStream_t st, PcgState* st2_all /* [LADDER_MAX][2] */, int B, int fixed_iterations, int max_iterations, const LmState* lm);
int  launch_pcg_dir3_lad(hipStream_t st, bool init, int nsys, Seg2 sg, size_t tail_off, int ntail, const float* z0, float* p0, const float* S, float* u0, const float* tD2_0, const float* cm, const LmState* lm,
                         const double* step_partials0, int n_step, double* d2_partials0, PcgState* st2_0, int prev_parity, int* host_flags0, int seq, const LadVec& lv,
                         int n_slice_partials = 0, const double* red4_0 = nullptr /* sharded: [LADDER_MAX][4] slice sums over all ranks; the tail then has a workgroup of its own */);
int  launch_pcg_step3_lad(hipStream_t st, int mode, int nsys, Step3Args a, const LadVec& lv);      // a.redop != null: sharded (see Step3Args)
// sharded ladder pass: a rank's contributions to the two all-reduces of a pass, all live systems in one launch each
void launch_lad_reduce_step(hipStream_t st, int nsys, const double* step_partials0, int n_slice, double* red4_0, const LadVec& lv);
void launch_lad_reduce_op(hipStream_t st, int nsys, const float* cam_partials0, int n_cam, int cam_stride, int NS, const double* pq_partials0, int n_pq, const double* d2_partials0, int n_d2,
                          double* redop0, int redop_stride, const LadVec& lv);

// ---- shard_kernels.hip: the sharding plan of one outer iteration ----------------------------------------------------------------
void launch_need_mask(hipStream_t st, RowView r, int slice, unsigned long long* need /* [A], zeroed */);
void launch_halo_items(hipStream_t st, int A, int slice, int me, const unsigned long long* need, unsigned long long* items, int* count /* zeroed */, int cap);
void launch_tile_flags(hipStream_t st, int A, int T, const int* cflag, int* tileflag /* zeroed */);
size_t halo_sort_temp_bytes(int cap);
hipError_t launch_halo_sort(hipStream_t st, void* temp, size_t temp_bytes, const unsigned long long* in, unsigned long long* out, int n);

void launch_candidate(hipStream_t st, GridView g, RowView r, int K, float sign, const float* step, const float* S, const double* x_shared, double* xc_sdf, double* xc_alb,
                      double* xc_shared, double* norms2 /* [0] += |delta|^2, [1] += |x|^2 over free */, const float* mask, double* scratch, const LmState* lm = nullptr);
void launch_accept(hipStream_t st, GridView g, RowView r, const double* xc_sdf, const double* xc_alb, const LmState* lm = nullptr);           // x <- candidate, refresh fp32 shadows (lm: only if it accepted)
// ---- lm_kernels.hip: the trust-region loop on the device -------------------------------------------------------------------------
void launch_set_double(hipStream_t st, double* dst, double v);
void launch_lm_init(hipStream_t st, LmState* lm, const double* cost, const double* ngrad, const double* nfree, double radius0, LmRecord* rec, int seq);
void launch_lm_begin(hipStream_t st, LmState* lm, int K, int fix_poses, int fix_intr, int fix_dist, const double* cdiag, const double* tri, float* Mblk,
                     const float* tc, const float* tS, float* tD2, float* tMinv, LmRecord* rec, int seq);
void launch_lm_diag_dev(hipStream_t st, int n, const float* c, const float* S, const LmState* lm, float* D2, float* Minv);
void launch_cand_frames(hipStream_t st, int K, const double* xc, const FrameConst* base, FrameConst* out, const LmState* lm);
void launch_lm_decide(hipStream_t st, LmState* lm, const PcgState* ps, const double* norms2, const double* cand_cost, int attempt, int lm_steps, LmRecord* rec, int seq,
                      int lad_next = -1 /* ladder batch: index of the next system (look-ahead), -1 = serial loop */, int debug_invalid = 0);
void launch_lm_begin_lad(hipStream_t st, LmState* lm, int B, int K, int fix_poses, int fix_intr, int fix_dist, const double* cdiag, const double* tri, float* Mblk, size_t mblk_stride,
                         const float* tc, const float* tS, float* tD2, size_t tail_stride, LmRecord* rec, int seq);
void launch_mark_compute(hipStream_t st, RowView r, int* flag);
void launch_compact_list(hipStream_t st, int A, const int* flag, const int* scan, int* list);

}  // namespace i3d
