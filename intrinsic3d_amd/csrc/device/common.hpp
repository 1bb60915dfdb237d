// Shared device/host definitions of the gfx950 shading-optimisation path.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

namespace i3d {

// ---- neighbour table -------------------------------------------------------------------------------------
// 18 voxel offsets per voxel.  0..5 is the reference's 1-ring order (sdf/algorithms.cpp:75-91: +x,-x,+y,-y,+z,-z);
// 6..11 complete the forward stencil of an Eg row (shading_cost.cpp:65-118); 12..17 are their mirrors, used by
// the gather formulation of J^T (a parameter pulls from the rows whose stencil contains it).
constexpr int NUM_NBR = 18;
enum Nbr { NB_PX = 0, NB_MX, NB_PY, NB_MY, NB_PZ, NB_MZ, NB_P2X, NB_P2Y, NB_P2Z, NB_PXY, NB_PXZ, NB_PYZ,
           NB_M2X, NB_M2Y, NB_M2Z, NB_MXY, NB_MXZ, NB_MYZ };
__host__ __device__ inline void nbr_offset(int i, int& dx, int& dy, int& dz) {
    constexpr int8_t O[NUM_NBR][3] = {{1,0,0},{-1,0,0},{0,1,0},{0,-1,0},{0,0,1},{0,0,-1},{2,0,0},{0,2,0},{0,0,2},{1,1,0},{1,0,1},{0,1,1},
                                      {-2,0,0},{0,-2,0},{0,0,-2},{-1,-1,0},{-1,0,-1},{0,-1,-1}};
    dx = O[i][0]; dy = O[i][1]; dz = O[i][2];
}

// ---- Eg row parameter slots (shading_cost.cpp:90-129): 10 sdf, 4 albedo, pose 6, intrinsics 4, distortion 5
constexpr int P_SDF = 0, P_ALB = 10, P_POSE = 14, P_INTR = 20, P_DIST = 24, P_TOTAL = 29, P_VOX = 14;
// forward neighbour (-1 = the voxel itself) holding the parameter of voxel-slot c (sdf 0..9, albedo 10..13)
__host__ __device__ inline int slot_fwd_nbr(int c) {
    constexpr int8_t F[P_VOX] = {-1, NB_PY, NB_P2Y, NB_PYZ, NB_PZ, NB_P2Z, NB_PX, NB_PXY, NB_PXZ, NB_P2X, -1, NB_PX, NB_PY, NB_PZ};
    return F[c];
}
// reverse neighbour: the voxel whose slot c is *this* voxel
__host__ __device__ inline int slot_rev_nbr(int c) {
    constexpr int8_t R[P_VOX] = {-1, NB_MY, NB_M2Y, NB_MYZ, NB_MZ, NB_M2Z, NB_MX, NB_MXY, NB_MXZ, NB_M2X, -1, NB_MX, NB_MY, NB_MZ};
    return R[c];
}

// ---- per-voxel flag bits (recomputed every outer iteration: the shell test reads sdf_refined, optimizer.cpp:187)
enum : uint8_t { F_VALID = 1, F_ACTIVE = 2, F_RING = 4, F_FREE_SDF = 8, F_FREE_ALB = 16 };

constexpr int MAX_SLOTS = 8;
constexpr int LNBR_WORDS = 7;             // tile plan: 18 local stencil slots x 12 bits per entry
constexpr int ROW_PLANES = 7;
constexpr int ROW_FREE_BIT = 1 << 30;     // set in a row's keyframe tag when the row has at least one free column
// One (64-entry group, slot) block of the row buffer = 7680 B: seven 1 KB planes of float4 (partials 0..27), then one 512 B plane of float2
// (partial 28, keyframe tag) — see RowView.  row_index: float4 units;  row_jt_index: float2 units of the same buffer
constexpr int ROW_BLOCK_F4 = 64 * ROW_PLANES + 32;        // 480 float4
__host__ __device__ inline size_t row_index(size_t a, int k, int plane, int slots) { return ((a >> 6) * (size_t)slots + (size_t)k) * ROW_BLOCK_F4 + ((size_t)plane << 6) + (a & 63); }
__host__ __device__ inline size_t row_jt_index(size_t a, int k, int slots) { return (((a >> 6) * (size_t)slots + (size_t)k) * ROW_BLOCK_F4 + 64 * ROW_PLANES) * 2 + (a & 63); }
__host__ __device__ inline size_t row_scalar_index(size_t a, int k, int slots) { return (((a >> 6) * (size_t)slots + (size_t)k) << 6) + (a & 63); }    // one value per row, same wave tiling

// ---- per-keyframe constants, rebuilt on the host (fp64) once per outer iteration ----------------------------
struct FrameHot {                  // what the build / cost kernels read per row: 144 B, staged in LDS for all keyframes
    double R[9], t[3];             // ceres::AngleAxisRotatePoint as a matrix (row-major) + translation (value path, fp64)
    float  Jr[9];                  // right Jacobian of the rotation: d(R P)/d omega_i = -R [P]x Jr e_i, i.e. Jr e_i = vee(R^T dR/d omega_i)
    int    pad;
    const float* lum;              // luminance image of the current pyramid level
};
struct FrameConst {
    FrameHot hot;
    float  Rf[9], tf[3];           // math::poseVecAAToMat(...).cast<float>() (math.cpp:151-163) for the observation pass
    const float* depth; const uint8_t* bgr;
    int w, h;
};

// ---- device views ------------------------------------------------------------------------------------------
struct GridView {
    int N;                          // stored voxels, in brick-sorted device order
    float voxel_size, truncation;
    const int* cx; const int* cy; const int* cz;     // voxel coordinates
    const int* rank;                // position in the caller's visit order
    const int* nbr;                 // [NUM_NBR][N] device indices, -1 = not stored
    const float* weight; const uchar4* color;
    const double* sdf0;             // VoxelSBR::sdf (fused value, constant)
    double* x_sdf; double* x_alb;   // master unknowns (fp64)
    float* f_sdf; float* f_alb;     // fp32 shadows read by the kernels
    const float* sh;                // [9][N] per-voxel SH coefficients
    uint8_t* flags;
    int* aidx;                      // device index -> active index or -1
};

// ---- sharding (one process per GPU) -------------------------------------------------------------------------
// The voxel state, flags and work list are REPLICATED on every rank (a few hundred bytes per voxel against 288 GB of HBM); what is sharded
// is the row work + row storage and the solver vectors.  The work list is brick-Morton ordered, so a contiguous range of it is a compact
// region of the surface; rank k OWNS the range [k*slice, (k+1)*slice), slice a multiple of SHARD_ALIGN entries (whole tiles of the
// operator pass, tile_pass.hip = whole groups of bricks).  It builds rows for its compute list = owned entries + the GHOST entries of other
// ranks whose rows touch an owned unknown (a 1-2 voxel rim, ~2 % at 8 ranks of 8 M voxels), so everything that lands on an owned unknown is
// computed locally and nothing has to be returned to an owner.  What a rank needs from others is the operator input on the rim its rows
// READ (ghosts and their forward stencils): pushed by the owners once per PCG pass (Comm::push_halo), peer to peer.
// Every solver vector has the SAME global layout on every rank:  [ sdf: chunk | albedo: chunk | poses 6K | intrinsics 4 | distortion 5 ],
// chunk = world * slice >= A; a rank keeps its two owned segments (and the halo values pushed to it) up to date, the camera tail is replicated.
constexpr int SHARD_ALIGN = 1024;
__host__ __device__ inline int vec_sdf(int a, int) { return a; }
__host__ __device__ inline int vec_alb(int a, int chunk) { return chunk + a; }

// does rank [own0, own1) need the rows of work-list entry a?  (owned, or an owned unknown is one of its ring / forward-stencil columns)
__host__ __device__ inline bool shard_needs_entry(int a, int own0, int own1, bool active, const int* anbr, size_t stride) {
    if (a >= own0 && a < own1) return true;
    if (!active) return false;
    bool need = false;
    for (int i = 0; i < 12; ++i) { const int la = anbr[(size_t)i * stride + a]; need |= (la >= own0 && la < own1); }   // ring 0..5, forward stencil 0,2,4,6..11
    return need;
}
// which ranks hold the rows of ACTIVE entry a (its owner and the owners of its 12 ring / forward-stencil columns); col[12] = those columns
__host__ __device__ inline unsigned long long shard_entry_ranks(int a, int slice, const int* anbr, size_t stride, int col[12], bool& interior) {
    const int o = a / slice;
    unsigned long long ranks = 1ull << o; interior = true;
    for (int i = 0; i < 12; ++i) { col[i] = anbr[(size_t)i * stride + a]; if (col[i] >= 0) { const int k = col[i] / slice; ranks |= 1ull << k; interior &= k == o; } }
    return ranks;
}
__host__ __device__ inline void shard_range(int A, int world, int rank, int& chunk, int& own0, int& own1) {
    const int tiles = (A + SHARD_ALIGN - 1) / SHARD_ALIGN, per_rank = tiles > 0 ? (tiles + world - 1) / world : 1;
    const int slice = per_rank * SHARD_ALIGN;
    chunk = world * slice;
    own0 = rank * slice < A ? rank * slice : A; own1 = (rank + 1) * slice < A ? (rank + 1) * slice : A;
}

struct RowView {                    // per work-list entry a in [0, A): voxels that are active (own rows) or free (own unknowns)
    int A; int Acap; int slots;
    int chunk;                      // length of the sdf / albedo part of every solver vector (>= A, see the layout above)
    int world;                      // number of ranks
    int own0, own1;                 // owned work-list range of this rank
    const int* clist; int nC;       // compute list (ascending work-list indices), nullptr = identity over [0, A)
    const int* alist;               // list index -> device voxel index (ascending)
    const uint8_t* aflags;          // voxel flags of the entry
    const int* anbr;                // [NUM_NBR][Acap] neighbour table in LIST space (-1 = neighbour not in the list => fixed, contributes 0)
    int* obs_frame; float* obs_w;   // [slots][Acap] observation pass output (ascending weight, 0 = none)
    // Eg rows of an entry are compacted into its first nrows slots.  What the operator streams per row is 120 B — the 29 partials (SURVEY.md 8(d)'s
    // 4 B per non-zero) + the row's keyframe id: the row weight is FOLDED into the partials (Js = sqrt(w) J, so J^T W J u = Js^T (Js u) and the
    // weight is never read); weight and residual sit in a side array only the once-per-iteration kernels read.
    //   rows   : wave-tiled (AoSoA) [group = a/64][slot] blocks of 7680 B, ONE contiguous block per wave and slot:
    //              [plane 0..6][lane = a%64] float4: sqrt(w) x columns 0..27 (sdf 0-9, albedo 10-13, pose 14-19, intrinsics 20-23, distortion 24-27), then
    //              [lane] float2: x = sqrt(w) x column 28 (p2), y = keyframe | ROW_FREE_BIT (int bits)
    //   row_wr : [group][slot][lane] float2: x = row weight obs.w*weight_sdf (> 0 for every stored row), y = raw residual
    float4* rows; float2* row_wr;
    __host__ __device__ float2* row_jt() const { return reinterpret_cast<float2*>(rows); }       // index with row_jt_index
    uint8_t* nrows;                 // [Acap]
    int* gmax;                      // [ceil(Acap / 64)] largest nrows of every group of 64 entries (k_group_rows, after the build): the operator pass bounds a wave's row
                                    // stream to the slots its group uses (buffer range check: the empty slots cost no memory traffic)
    uint8_t* regflags;              // [Acap] bit0 Er row, bit1 Es row, bit2 Es Jacobian is 1 (else 0), bit3 Er row has a free column, bit4 Es free
    float* ea_w;                    // [6][Acap] chroma weight of the Ea row towards 1-ring neighbour d, 0 = none
    uint8_t* ea_free;               // [Acap] bit d: Ea row d has a free column
};

// Jacobi scale S = 1 / (1 + |column|), LM diagonal D^2 = clamp(|column|^2 S^2) / radius and the 1x1 block-Jacobi inverse 1 / (|column|^2 S^2 + D^2) of a
// voxel unknown from its masked squared column norm cm (-1 = fixed parameter: everything 0).  ONE definition for the kernels that store them as vectors
// (k_scale, k_lm_diag: camera tail, sharded / untiled solve, candidate point) and for the fused PCG kernels, which recompute them from cm instead of
// reading three vectors (24 B less per entry and pass): both see the same values.  v_sqrt_f32 / v_rcp_f32 (1 ulp): what is rounded here is a
// preconditioner and a column scaling, and the correctly rounded forms cost ~40 instructions per unknown, which the vector kernels feel.
#ifdef __HIPCC__
static __device__ inline float lm_scale(float cm) { return cm >= 0.0f ? __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_sqrtf(cm)) : 0.0f; }
static __device__ inline void lm_diag(float cm, float s, float inv_radius, float& d2, float& minv) {
    if (s == 0.0f) { d2 = 0.0f; minv = 0.0f; return; }
    const float cs = (cm * s) * s;
    d2 = fminf(fmaxf(cs, 1e-6f), 1e32f) * inv_radius; minv = __builtin_amdgcn_rcpf(cs + d2);
}
#endif

// device-resident scalar state of one PCG solve (ConjugateGradientsSolver) — no host round trip inside an iteration
struct PcgState {
    double acc[4];                  // slice partial sums of the iteration in flight: r.z, x.(b+r), x.r, sum D^2 x^2 (one all-reduce when sharded)
    double rho, last_rho, pq, alpha, beta;
    double xbr, xr, d2xx;           // x.(b+r), x.r, sum D^2 x^2 at the last completed iteration
    double Q0, Q1;
    int it;                         // completed iterations
    int done;                       // 0 running, 1 converged (Q-test / fixed count / max), 2 breakdown before the update of this iteration
    int fixed_iterations;           // >= 0: stop exactly there
    int max_iterations;
};

// Device-resident state of one NLSSolver::solve (Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy as configured by nls_solver.cpp:296-337):
// the trust-region bookkeeping runs in one-thread kernels (lm_kernels.hip), the host only queues launches and polls one record per attempt.
// The damping ladder (solver.cpp lm_solve): the reference restarts the trust region at 1e4 in every outer iteration (optimizer.cpp:138), so most LM attempts are
// rejected, and after a rejection the next radius is known in advance (radius /= decrease_factor, decrease_factor *= 2: LevenbergMarquardtStrategy::StepRejected).
// Up to LADDER_MAX consecutive attempts are therefore SOLVED together — PCG systems that differ only in the LM diagonal, iterated in lock step so that one stream of
// J serves several of them (tile_pass_mr.hip) — and then DECIDED one after the other exactly as the serial loop would.
constexpr int LADDER_MAX = 6;
struct LmState {
    double cost;                    // cost at the current point
    double radius, decrease_factor; // trust-region radius, its reduction factor after a rejected step (doubles every time)
    double ngrad, nfree;            // free parameters whose gradient entry exceeds Ceres' gradient_tolerance (max-norm test: 0 = converged), free parameters at the start
    float  inv_radius; int pad0;    // (float)(1 / radius) of the attempt in flight: what the vector kernels read
    int done;                       // 0 running | 1 the solve is over: every later kernel of it returns at once | 2 the ladder is out of step (an invalid step halved the
                                    // radius instead): the rest of the batch is skipped and the host starts a new batch at the current radius (k_lm_begin_lad clears it)
    int termination;                // i3d_iteration_stats::termination: 0 step limit | 1 converged (tolerances, radius) | 2 successful step (the callback) | 3 invalid steps
    int accepted;                   // the deciding attempt accepted its candidate (k_accept applies it)
    int invalid;                    // consecutive invalid steps (max_num_consecutive_invalid_steps = 5)
    int attempts;                   // attempts decided
    int successful;
    // ladder batch in flight: the radius each of its systems was solved with (system j = the attempt after j rejections), what the vector kernels read of it
    double lad_radius[LADDER_MAX]; float lad_inv_radius[LADDER_MAX]; int lad_n; int pad1;
};
struct LmRecord {                   // one per attempt (index 0: the initial tests), in mapped host memory; `seq` is stored last with release semantics
    int seq; int final_;            // final_: the solve ended here
    int accepted; int pcg_it; int termination; int kind;      // kind: 0 init | 1 decided attempt | 2 ended before the attempt (radius underflow) | 3 ladder out of step: attempt NOT decided, solve it again
    double cost, cand_cost, model_change, rel, radius_after, ngrad, nfree;
};

struct OptParams {                  // scalar state of one outer iteration
    double thres_shell; double lambda_a;
    double type_w[4];               // lambda_t / sum_t * 1000
    float type_wf[4];               // the same in float: kernel arguments are scalar registers, (float)type_w[t] would be a VALU result = a VGPR
    int K; int level; double pyr_scale; float occlusion; int use_er, use_es, use_ea;
    double intr[4]; double dist[5];  // level-0 intrinsics, distortion
    float cam_f[4]; float dist_f[5]; int dist_zero; int w, h;   // float camera of the observation pass (scaled)
    int fix_poses, fix_intr, fix_dist, fix_sdf;
};

#define I3D_HIP_CHECK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    std::snprintf(i3d::g_errbuf, sizeof(i3d::g_errbuf), "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); return I3D_ERR_HIP_; } } while (0)
constexpr int I3D_ERR_HIP_ = 3;
extern thread_local char g_errbuf[512];

// Launch wrappers return counts, not status codes.  A launch CONFIGURATION that cannot work — a dynamic-LDS request over the device limit, which grows with the
// keyframe count K — is latched per thread with a message that names the kernel, the request and K; the host control picks it up where it checks
// hipGetLastError() (ctx_launch_check, context.cpp) and returns I3D_ERR_CAPACITY instead of a generic HIP launch failure.
constexpr size_t I3D_LDS_LIMIT = 160 * 1024;
bool set_dynamic_lds(const void* kernel, const char* name, size_t bytes, int K);       // false: latched, do not launch
bool take_launch_error(char* msg, size_t n);                                            // true: there was one (cleared)

}  // namespace i3d
