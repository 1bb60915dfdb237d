// K6, ladder form — the PCG operator pass q_b = J^T W J u_b for NB systems b at once: ONE stream of the stored rows serves NB right-hand sides.
//
// Why: the reference restarts its trust region at 1e4 in every outer iteration (optimizer.cpp:138 constructs a fresh NLSSolver, nls_solver.cpp:322-323 never
// takes effect), so 5 of 6 LM attempts are rejected, and the rejected attempts solve systems (J^T W J + D_k^2) y = b that share J, b and the Jacobi scaling and
// differ only in the damping D_k^2 = clamp(diag) / radius_k — whose radii are known in advance (common.hpp: LADDER_MAX).  The serial loop streams the same
// 1.46 GB of rows ~130 times per Gauss-Newton iteration on the bench workload; iterated in lock step the systems share each stream (solver.cpp pcg_solve_ladder).
//
// What is shared per row: the 120 B of the row, its unpacked plan slots, the keyframe grouping of the wave (which lanes hold which keyframe).  What is per system:
// the operator input u_b (staged in LDS per system), t_b = W (J u_b), the column sums.  Unlike k_eg_tile (4 waves per SIMD, 128 registers, column sums in LDS)
// this kernel runs 8 waves of 512 lanes per CU at 2 waves per SIMD: the 12 column sums of every system live in REGISTERS during the row loop and go through ONE
// LDS buffer system by system in the pull phase.  With NB systems the pass is VALU-bound from NB = 3 on (each row costs ~NB x the arithmetic for one load),
// which is why a batch of more than 3 systems is split into groups of <= 3 rather than widened (and why LDS holds 3 staged inputs and no more at K = 200).
//
// Every sum is taken in a fixed order: the halo is PULLED over the plan's lists (k_tile_pull_plan), the pose block goes through per-wave keyframe tables, the wave
// sums are DPP trees — a launch is bit-reproducible, and the result of a system does not depend on which other systems share its launch (NB is only a loop
// bound around per-system arithmetic): a system iterated in a batch goes through bit-identical states to the same system iterated alone through k_eg_tile_mr<1>
// (tests/test_gpu_ladder.py).  Against k_eg_tile the pose sums are associated differently (see mr_table_add), i.e. equal to fp32 round-off, not bit for bit.
//
// What the first version of this kernel taught (profiles/r05_ladder_mr_v1.json: 0.73 ms for 3 systems against 3 x 0.275 ms serial — no gain): the pass is bound by
// INSTRUCTION ISSUE, not by bandwidth, as soon as more than one system shares a row.  (a) The compiler does not batch LDS reads that feed one FMA chain: every
// ds_read was followed by its own s_waitcnt, 27 exposed LDS latencies per row and system at 2 waves per SIMD.  Here the inputs of all systems sit side by side in
// LDS (one 16-byte read per stencil slot serves every system) and a row's reads are issued together in front of a scheduling barrier.  (b) The wave sums of the pose
// block were readlane trees with an exec-masked LDS add each (25 instructions and 6 hazard stalls per value): here 4 + 2 DPP adds per value, all values of a round
// interleaved, ONE exec-masked block of LDS adds.
//
// What later A/Bs added (same session, builds interleaved): the row loop is NOT where an added system's 0.08 ms go.  The 29-term dot product as two packed chains (-6 % VALU
// instructions, profiles/r05_mr_packed_dot_ab.json) measures the same; an entry's 14 voxel inputs held in registers across its rows (-29 % LDS reads with two systems,
// profiles/r05_mr_hoist_inputs_ab.json; kept for one and two systems, the 3-system kernel has no registers left) buys 0.7 %.  What remains per system is its staging (2048 gathered
// inputs per tile) and its pull phase (column sums through LDS, two barriers, the halo list walk, the output stores) at ONE 512-thread workgroup per CU.
#include <cstring>
#include <algorithm>
#include <vector>
#include <cstdlib>
#include "kernels.hpp"
#include "reduce_device.hpp"
#include "wave_ops.hpp"
#include "tile_device.hpp"

namespace i3d {

constexpr int MR_T = 512, MR_HMAX = 1536, MR_NWV = MR_T / 64, MR_TC = 32;

// LDS layout in floats.  Everything whose size does not depend on K sits at COMPILE-TIME offsets (an offset that is a constant costs no scalar register across the
// row loop):
//   [NB] x { u_s, u_a [2 NSLOT] } | [NB] x Er row values [T + 4] | [flag | 3 pad] [NB][72] per-wave intrinsics / distortion sums | [NWV][TC] keyframe tags (shared by the systems: the rows are) |
//   [NWV][TC][6 NB] table sums (the values of a slot side by side) | the tile's pull list [2 HMAX] | column sums of ONE system [12][T] | pull-list offsets [(HMAX + 4) / 2, rounded]
//   | the lanes' running p.q [NB][T] fp64 (parked here: as register pairs they are live across the whole row loop)
// then, per system and K-dependent: dense camera accumulator [rs + 9] | pose part of u_b [6K] | its intrinsics / distortion part [9, padded to 12, 16-byte aligned].
// (Inputs of the systems interleaved per slot — one 16-byte read serving three systems — were built and dropped: the 40 registers one row's reads then occupy
// spilled the 3-system kernel; what matters is that a system's reads are issued TOGETHER, not that they are few.)
template <int NB> struct MrConst {
    static constexpr int T = MR_T, HMAX = MR_HMAX, NW = MR_NWV, TC = MR_TC, NSLOT = T + HMAX + 1;
    static constexpr int CAMW = (NW * 9 + 3) & ~3, VSTR = NW * TC * 6;          // (VSTR x NB floats of table sums: [NW][TC][6 NB])
    // The staged inputs come FIRST (round 6): the row loop reads them at lane-dependent slots, 14 reads per (row, system), and a ds_read takes an immediate offset of at most
    // 64 KB - with the inputs of all three systems below that line one shifted slot register serves every system (before, system 2's albedo half lay at 65.4 KB and every
    // read of it needed its own address register or add: the 3-system kernel sat at 248 registers).  The Er row values moved out of the per-system input blocks for the same reason.
    static constexpr int O_U = 0, UB = (2 * NSLOT + 3) & ~3;                     // system b: u_s at O_U + b UB, u_a behind it
    static constexpr int O_TR = O_U + NB * UB, TRB = (T + 4 + 3) & ~3;          // Er row values of system b at O_TR + b TRB
    static constexpr int D_FLAG = O_TR + NB * TRB, D_CAM9W = D_FLAG + 4, D_TAG = D_CAM9W + NB * CAMW, D_VAL = D_TAG + NW * TC;
    static constexpr int O_LIST = D_VAL + NB * VSTR, O_C = O_LIST + 2 * HMAX, O_OFFS = O_C + 12 * T, O_PQ = O_OFFS + (((HMAX + 4) / 2 + 3) & ~3), D0 = O_PQ + NB * 2 * T;      // O_PQ: the lanes' running p.q, fp64, [NB][T]
};
struct MrLayout { int o_upose, o_ui, SK; size_t bytes; };          // the K-dependent tail: system b at D0 + b SK: accumulator, then (at o_upose) the pose part of u_b, then (at o_ui, 16-byte aligned) its 9 intrinsics / distortion entries
static __host__ __device__ inline MrLayout mr_layout(int D0, int NB, int K) {
    const int rs = (6 * K) | 1;
    MrLayout L;
    L.o_upose = (rs + 9 + 3) & ~3; L.o_ui = (L.o_upose + 6 * K + 3) & ~3; L.SK = L.o_ui + 12;
    L.bytes = (size_t)(D0 + NB * L.SK) * sizeof(float);
    return L;
}
static size_t mr_lds_bytes(int NB, int K) { return NB == 1 ? mr_layout(MrConst<1>::D0, 1, K).bytes : (NB == 2 ? mr_layout(MrConst<2>::D0, 2, K).bytes : mr_layout(MrConst<3>::D0, 3, K).bytes); }

// The pose columns of one row slot across the wave for NB systems: wave_table_add (wave_ops.hpp) with the table look-up done once per round and the 6 NB wave sums of a
// round taken TOGETHER by wave_sum_quads (four values share one tree: 2.5 instructions per value instead of the 6 of a DPP tree, or the 25 of the first version's
// readlane trees).  The sums of value vi = 6 b + i end up in the lanes of one row of 16; the first lane of every row adds them to the wave's table — [slot][6 NB],
// the values of a slot side by side — in ONE exec-masked block of MQ LDS adds (4 lanes each, distinct addresses).
template <int NB>
static __device__ inline void mr_table_add(bool valid, int f, const float (&jp)[6], const float (&ts)[NB], float* lds, int o_tag, int o_val, int& count, int o_dense, int dense_stride) {
    constexpr int M = 6 * NB, MQ = (M + 3) / 4;
    bool pending = valid;
    unsigned long long todo = __ballot(pending);
    const int lane = (int)(threadIdx.x & 63u);
    const int qv = wave_quad_value(lane);                       // this lane's rows hold value 4 q + qv of quad q
    while (todo != 0ull) {
        const int leader = __ffsll((long long)todo) - 1;
        const int f0 = __builtin_amdgcn_readlane(f, leader);
        const bool mine = pending && f == f0;
        const int tg = __float_as_int(lds[o_tag + (lane & (MR_TC - 1))]);
        const unsigned long long hit = __ballot(tg == f0);
        int slot;
        if (hit != 0ull) slot = (__ffsll((long long)hit) - 1) & (MR_TC - 1);
        else if (count < MR_TC) { slot = count; if (lane == 0) lds[o_tag + slot] = __int_as_float(f0); count = count + 1; }
        else slot = -1;
        float v[4 * MQ], sum[MQ];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float tm = mine ? ts[b] : 0.0f;
#pragma unroll
            for (int i = 0; i < 6; ++i) v[b * 6 + i] = jp[i] * tm;
        }
#pragma unroll
        for (int j = M; j < 4 * MQ; ++j) v[j] = 0.0f;
        wave_sum_quads<MQ>(v, sum);
        if ((lane & 15) == 0) {
            if (slot >= 0) {
                const int base = o_val + slot * M + qv;
#pragma unroll
                for (int q = 0; q < MQ; ++q) { if (4 * q + 3 < M || 4 * q + qv < M) lds_add(&lds[base + 4 * q], sum[q]); }      // (no return value; the table is this wave's alone, its LDS operations execute in program order)
            } else {            // table full (not seen on the bench scenes): straight into the dense accumulators (order-dependent)
#pragma unroll
                for (int q = 0; q < MQ; ++q) { const int vi = 4 * q + qv; if (vi < M) { const int b = vi / 6; lds_add(&lds[o_dense + b * dense_stride + 6 * f0 + (vi - 6 * b)], sum[q]); } }
            }
        }
        pending = pending && !mine;
        todo = __ballot(pending);
    }
}
template <int NB>
static __device__ inline void mr_table_merge(float* lds, int o_tag, int o_val, int& count, int o_dense, int dense_stride) {
    constexpr int M = 6 * NB;
    const int lane = (int)(threadIdx.x & 63u);
    for (int e = lane; e < count * M; e += 64) {
        const int sl = e / M, vi = e - sl * M, b = vi / 6, i = vi - 6 * b;
        const int f = __float_as_int(lds[o_tag + sl]);
        lds[o_dense + b * dense_stride + 6 * f + i] += lds[o_val + e];
        lds[o_val + e] = 0.0f;
    }
    if (lane < MR_TC) lds[o_tag + lane] = __int_as_float(-1);
    count = 0;
}

// Phase timing (variant build only: -DI3D_MR_PHASES, tools/build_variant.sh; never in the shipped library).  Every wave accumulates the s_memtime ticks it spends in
// each phase of a tile in scalar registers and adds them to g_mr_phase[NB - 1][phase] at the end; mr_phase_report() (called by the variant's launch wrapper at
// exit) prints wave-averaged shares.  Phases: 0 prologue | 1 issue the tile's loads | 2 staging writes (waits for the gathers) | 3 staging barrier | 4 regulariser rows + hoisted inputs |
// 5 row loop | 6 reverse slots / Ea weights | 7 pull: column sums -> LDS | 8 pull: first barrier | 9 pull: sums, halo list walk, stores | 10 pull: second barrier | 11 epilogue
#ifdef I3D_MR_PHASES
constexpr int MR_NPH = 12;
__device__ unsigned long long g_mr_phase[3][MR_NPH + 2];
#define PH_DECL unsigned long long ph_[MR_NPH] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long ph_t_ = __builtin_amdgcn_s_memtime(); unsigned ph_tiles_ = 0
#define PH(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_[k] += t_ - ph_t_; ph_t_ = t_; } while (0)
#define PH_TILE ++ph_tiles_
#define PH_FLUSH(NBv) do { if ((threadIdx.x & 63u) == 0u) { for (int k_ = 0; k_ < MR_NPH; ++k_) atomicAdd(&g_mr_phase[(NBv) - 1][k_], ph_[k_]); atomicAdd(&g_mr_phase[(NBv) - 1][MR_NPH], (unsigned long long)ph_tiles_); atomicAdd(&g_mr_phase[(NBv) - 1][MR_NPH + 1], 1ull); } } while (0)
#else
#define PH_DECL
#define PH(k)
#define PH_TILE
#define PH_FLUSH(NBv)
#endif
// Workgroup balance (variant build only: -DI3D_MR_BLOCKTIME): two s_memtime reads per workgroup — start and end — summed per workgroup index over all launches; the report
// gives mean / min / max over the workgroups.  A launch lasts as long as its slowest workgroup: max / mean is what a balanced tile assignment could win.
#ifdef I3D_MR_BLOCKTIME
__device__ unsigned long long g_mr_blk[3][1024], g_mr_blk_n[3][1024];
#define BT_DECL const unsigned long long bt0_ = __builtin_amdgcn_s_memtime()
#define BT_FLUSH(NBv) do { if (threadIdx.x == 0u && blockIdx.x < 1024u) { atomicAdd(&g_mr_blk[(NBv) - 1][blockIdx.x], __builtin_amdgcn_s_memtime() - bt0_); atomicAdd(&g_mr_blk_n[(NBv) - 1][blockIdx.x], 1ull); } } while (0)
#else
#define BT_DECL
#define BT_FLUSH(NBv)
#endif

struct MrArgs {
    const float* u0; float* qacc0; float* qh0; double* pq0 /* or null: no p.q (residual-reset pass) */; float* cam0;
    const PcgState* st0;                   // system 0's state of this pass's parity; system j's is st0 + 2 j
    size_t vec, qh, cam, part;             // LadVec strides
    int sys[3];                            // the systems of this launch
};

// GHOSTS (sharded runs, round 6): as in k_eg_tile — the tile list is this rank's own tiles [tile_first, tile_first + n_own) followed by the foreign tiles that hold its ghost
// entries; p.q, the intrinsics / distortion sums and the pose block count a row ONCE, on the rank that owns its voxel (`owned`), while the voxel columns of every row this
// rank streams (owned and ghost) land in its accumulators.  A template parameter: the single-rank kernel keeps its registers.
template <int NB, bool GHOSTS>
__global__ void __launch_bounds__(MR_T, 2) k_eg_tile_mr(RowView r, OptParams p, MrArgs m, const unsigned* __restrict__ lnbr, const float* __restrict__ eaw_sym, const int* __restrict__ halo_idx,
                                                        const int* __restrict__ halo_cnt, int tiles_per_block, int ntl, int cam_stride, const int* __restrict__ gmaxv,
                                                        const unsigned short* __restrict__ hp_off, const unsigned short* __restrict__ hp_src,
                                                        int tile_first, int n_own, const int* __restrict__ ghost_list) {
    constexpr int T = MR_T, HMAX = MR_HMAX, NW = MR_NWV, TC = MR_TC, ZSLOT = T + HMAX, NCOL = 12, HPCAP = 4 * HMAX, NQH = HMAX / T;
    // A system that has stopped (the host drops it from the launches one pass after it saw the flag) is carried without arithmetic: its inputs are still staged — the
    // loads of a tile are unconditional — but its rows, its pull phase and its outputs are skipped (workgroup-uniform branches: the state is read once per launch).
    bool alive[NB];
    {
        bool any = false;
#pragma unroll
        for (int b = 0; b < NB; ++b) { alive[b] = m.st0[2 * m.sys[b]].done == 0; any = any || alive[b]; }
        if (!any) return;      // every system of the launch has stopped: nothing to do (launches queued behind the convergence flags)
    }
    extern __shared__ float lds[];
    const int K = p.K; const size_t Acap = r.Acap; const int A = r.A, chunk = r.chunk;
    const int nshared = 6 * K + 9, rs = (6 * K) | 1;
    using MC = MrConst<NB>;
    constexpr int D_FLAG = MC::D_FLAG, D_CAM9W = MC::D_CAM9W, CAMW = MC::CAMW, D_TAG = MC::D_TAG, D_VAL = MC::D_VAL, D0 = MC::D0;
    const MrLayout L = mr_layout(D0, NB, K);
    const int SB = L.SK, o_upose = L.o_upose, o_ui = L.o_ui;
#define SYS(b) (D0 + (b) * SB)                                   /* K-dependent block of system b: dense pose accumulator [0, 6K), intrinsics / distortion totals [rs, rs + 9), camera part of u_b at o_upose */
#define U_S(b) (MC::O_U + (b) * MC::UB)
#define U_A(b) (MC::O_U + (b) * MC::UB + MC::NSLOT)
#define TR_L(b) (MC::O_TR + (b) * MC::TRB)
#define C_L (MC::O_C)
#define hp_list reinterpret_cast<unsigned short*>(lds + MC::O_LIST)
#define hp_offs reinterpret_cast<unsigned short*>(lds + MC::O_OFFS)
    const int i = threadIdx.x, lane = (int)(threadIdx.x & 63u);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t tail = 2 * (size_t)chunk;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float* ub = m.u0 + (size_t)m.sys[b] * m.vec;
        for (int e = i; e < rs + 9; e += T) lds[SYS(b) + e] = 0.0f;
        for (int e = i; e < 6 * K; e += T) lds[SYS(b) + o_upose + e] = ub[tail + e];
        if (i < 12) lds[SYS(b) + o_ui + i] = i < 9 ? ub[tail + 6 * K + i] : 0.0f;
    }
    if (lane < TC) lds[D_TAG + wave * TC + lane] = __int_as_float(-1);
    const int o_tag = D_TAG + wave * TC, o_val = D_VAL + wave * (TC * 6 * NB);
    for (int e = lane; e < TC * 6 * NB; e += 64) lds[o_val + e] = 0.0f;
    int tcount = 0;
    float cam9[NB][9];
#define PQ_L(b) reinterpret_cast<double*>(lds + MC::O_PQ + (b) * 2 * T)
#pragma unroll
    for (int b = 0; b < NB; ++b) { PQ_L(b)[i] = 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q) cam9[b][q] = 0.0f; }
    const float tw0 = p.type_wf[0], tw1 = p.type_wf[1], tw2 = p.type_wf[2], tw3 = p.type_wf[3];
    const int tile0 = blockIdx.x * tiles_per_block;
    const int tk_end = min(tile0 + tiles_per_block, ntl);

    // the tile in flight (names as in k_eg_tile)
    int tile = 0, base = 0, a = 0, H = 0, nr_ld = 0; bool in = false, owned = false; size_t ac = 0;
    uint8_t fl = 0, rf_ld = 0; unsigned ln[5]; RowBlock rwA, rwB;
    float us[NB], ua[NB], hs[NB][NQH], ha[NB][NQH];
    auto tile_of = [&](int tk) { return (GHOSTS && tk >= n_own) ? ghost_list[tk - n_own] : (GHOSTS ? tile_first + tk : tk); };
    constexpr int NQL = (HPCAP / 8 + T - 1) / T, NQO = (HMAX + 1 + T - 1) / T;
    uint4 hpl[NQL]; unsigned short hpo[NQO];
    const char* wave_rows = reinterpret_cast<const char*>(r.rows);
    int gm = 0;
    const unsigned lane16 = (threadIdx.x & 63u) * 16u;
    auto load_block = [&](RowBlock& rw, int k) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wave_rows, 0, k < gm ? MAX_SLOTS * ROW_BLOCK_F4 * 16 : 0, 0x00020000);
#pragma unroll
        for (int q = 0; q < 7; ++q) { const v4u_b v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane16, k * (ROW_BLOCK_F4 * 16) + q * 1024, 2 /* nt */);
                                      rw.p[q] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)); }
        { const v2u_b t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane16 >> 1, k * (ROW_BLOCK_F4 * 16) + 64 * ROW_PLANES * 16, 2); rw.j28 = __uint_as_float(t.x); rw.tag = (int)t.y; }
    };
    auto group_rows = [&](int tk) -> int {
        const int wa0 = tile_of(tk) * T + (int)(threadIdx.x & ~63u);
        const int grp = __builtin_amdgcn_readfirstlane(wa0 < A ? (wa0 >> 6) : -1);
        return grp >= 0 ? gmaxv[grp] : 0;
    };
    // A tile's inputs arrive in TWO steps, both issued one tile ahead (round 6; s_memtime phase marks, profiles/r06_mr_phases_*.txt: with ONE workgroup per CU nothing hides
    // a tile's start-up — issuing its loads and waiting for the dependent gathers was 31 % of a tile's time, the row loop 30 %):
    //   issue_idx(tk)   : everything that depends on the tile number alone — halo indices, pull list and offsets, flags, plan slots — requested right behind the row loop of the
    //                     tile before, in front of its pull phase;
    //   issue_gather()  : the operator inputs of tile + halo (they need the halo indices) and the first two row blocks, requested behind the first barrier of that pull phase:
    //                     the indices have arrived by then, and the values have the rest of the pull phase to arrive in.
    // What the current tile still needs of its own (entry, flags, halo count, tile number) is copied out of the in-flight set at staging time (`_c`).
    int he[NQH];
    auto issue_idx = [&](int tk) {
        tile = tile_of(tk); base = tile * T; a = base + i; in = a < A; ac = in ? (size_t)a : 0;
        owned = GHOSTS ? (a >= r.own0 && a < r.own1) : in;          // p.q and the camera block count a row once: on the rank that owns its voxel
        { const int wa0 = base + (int)(threadIdx.x & ~63u);
          const unsigned grp = (unsigned)__builtin_amdgcn_readfirstlane(wa0 < A ? (wa0 >> 6) : 0);
          wave_rows = reinterpret_cast<const char*>(r.rows + (size_t)grp * (size_t)(r.slots * ROW_BLOCK_F4)); }
        H = halo_cnt[tile];
#pragma unroll
        for (int q = 0; q < NQH; ++q) he[q] = __builtin_nontemporal_load(&halo_idx[(size_t)tile * HMAX + (i + q * T < HMAX ? i + q * T : 0)]);
#pragma unroll
        for (int q = 0; q < NQL; ++q) { const int ch = i + q * T; hpl[q] = reinterpret_cast<const uint4*>(hp_src + (size_t)tile * HPCAP)[ch < HPCAP / 8 ? ch : 0]; }
#pragma unroll
        for (int q = 0; q < NQO; ++q) { const int o = i + q * T; hpo[q] = hp_off[(size_t)tile * (HMAX + 1) + (o <= HMAX ? o : HMAX)]; }
    };
    auto issue_gather = [&]() {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float* ub = m.u0 + (size_t)m.sys[b] * m.vec;
            us[b] = in ? ub[a] : 0.0f; ua[b] = in ? ub[chunk + a] : 0.0f;
#pragma unroll
            for (int q = 0; q < NQH; ++q) { const int e = (i + q * T < H) ? he[q] : 0; hs[b][q] = ub[e]; ha[b][q] = ub[chunk + e]; }
        }
    };
    auto issue_meta = [&]() {
        fl = in ? r.aflags[ac] : 0;
        nr_ld = r.nrows[ac];
        rf_ld = r.regflags[ac];
#pragma unroll
        for (int w = 0; w < 5; ++w) ln[w] = __builtin_nontemporal_load(&lnbr[(size_t)w * Acap + ac]);
    };
    PH_DECL; BT_DECL;
    int gm_next = 0;
    if (tile0 < tk_end) {          // (workgroup-uniform) the first tile's inputs: the only ones nothing overlaps
        gm = group_rows(tile0);
        issue_idx(tile0); issue_meta(); issue_gather();
        load_block(rwA, 0);
        load_block(rwB, 1);
        gm_next = tile0 + 1 < tk_end ? group_rows(tile0 + 1) : 0;
    }
    for (int tk = tile0; tk < tk_end; ++tk) {
        PH(tk == tile0 ? 0 : 10); PH_TILE;
        // the tile whose inputs are in flight becomes the current one
        const int a_c = a, H_c = H, tile_c = tile; const bool in_c = in, owned_c = owned;
        PH(1);
        // ---- stage the operator inputs of tile + halo ----
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            lds[U_S(b) + i] = us[b]; lds[U_A(b) + i] = ua[b];
#pragma unroll
            for (int q = 0; q < NQH; ++q) { const int hq = i + q * T; const bool hv = hq < H; lds[U_S(b) + T + hq] = hv ? hs[b][q] : 0.0f; lds[U_A(b) + T + hq] = hv ? ha[b][q] : 0.0f; }
            if (i == 0) { lds[U_S(b) + ZSLOT] = 0.0f; lds[U_A(b) + ZSLOT] = 0.0f; lds[TR_L(b) + T] = 0.0f; }
        }
#pragma unroll
        for (int q = 0; q < NQL; ++q) { const int ch = i + q * T; if (ch < HPCAP / 8) { reinterpret_cast<uint2*>(hp_list)[2 * ch] = make_uint2(hpl[q].x, hpl[q].y); reinterpret_cast<uint2*>(hp_list)[2 * ch + 1] = make_uint2(hpl[q].z, hpl[q].w); } }
#pragma unroll
        for (int q = 0; q < NQO; ++q) { const int o = i + q * T; if (o <= HMAX) hp_offs[o] = hpo[q]; }
        if (i == 0) lds[D_FLAG] = __int_as_float(0);
        const bool active = in_c && (fl & F_ACTIVE);
        const int nr = active ? nr_ld : 0;
        const uint8_t rf = active ? rf_ld : 0;
        if (!in_c) { constexpr AllZ<ZSLOT> az; for (int w = 0; w < 5; ++w) ln[w] = az.w[w]; }
        const unsigned ln4_c = ln[4];
        PH(2);
        __syncthreads();
        PH(3);
        const int sx = unpack12(ln, 5), sy = unpack12(ln, 0), sz = unpack12(ln, 3), mx = unpack12(ln, 9), my = unpack12(ln, 10), mz = unpack12(ln, 11);
        float self_s[NB], self_a[NB], pq_rows[NB], C[NB][NCOL];
        // ---- regulariser rows (constant coefficients), while the first two row blocks are in flight ----
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            self_s[b] = 0.0f; self_a[b] = 0.0f; pq_rows[b] = 0.0f;
#pragma unroll
            for (int c = 0; c < NCOL; ++c) C[b][c] = 0.0f;
            const int rg[6] = {sx, mx, sy, my, sz, mz};
            float tr = 0.0f; double pq_pre = 0.0;
            const float usb = lds[U_S(b) + i];
            if (rf & 1) {
                const float lap = ((((((-6.0f * usb) + lds[U_S(b) + rg[0]]) + lds[U_S(b) + rg[1]]) + lds[U_S(b) + rg[2]]) + lds[U_S(b) + rg[3]]) + lds[U_S(b) + rg[4]]) + lds[U_S(b) + rg[5]];
                tr = tw1 * lap; if (owned_c) pq_pre += (double)(tr * lap);
                self_s[b] += -6.0f * tr;
            }
            lds[TR_L(b) + i] = tr;
            if ((rf & 2) && (rf & 4)) { const float ts = tw2 * usb; if (owned_c) pq_pre += (double)(ts * usb); self_s[b] += ts; }
            if (rf & 7) PQ_L(b)[i] += pq_pre;
        }
        // the lane's 9 forward stencil slots (sdf slots 1..9 of a row): the same for every row of the entry and for every system
        int so[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) so[c] = unpack12(ln, c);
        // The 14 voxel inputs of a row (10 sdf, 4 albedo) are the ENTRY's: the same for its <= 5 rows.  With one or two systems in the launch there are registers to hold them
        // across the row loop (14 per system); the 3-system kernel (248 registers) reads them from LDS row by row.
        constexpr bool HOIST = NB <= 2;
        float hsv[HOIST ? NB : 1][10], hav[HOIST ? NB : 1][4];
        if constexpr (HOIST) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                hsv[b][0] = lds[U_S(b) + i];
#pragma unroll
                for (int c = 1; c < 10; ++c) hsv[b][c] = lds[U_S(b) + so[c - 1]];
                hav[b][0] = lds[U_A(b) + i]; hav[b][1] = lds[U_A(b) + sx]; hav[b][2] = lds[U_A(b) + sy]; hav[b][3] = lds[U_A(b) + sz];
            }
        }
        // one row: t_b = W (J u_b) for every system, J^T t_b into the lane's column sums (registers)
        // `reload` >= 0: the slot this buffer takes next, requested as soon as the row's partials have been used — BEFORE the wave sums of its pose block, which need
        // only the six pose partials (copied) and the t_b: the request is in flight for the whole of the reduction and of the next row
        auto consume = [&](RowBlock& rb, int k, int reload) {
            const float4 (&rw)[7] = rb.p;
            int fsel = 0; bool pvalid = false; float tsel[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) tsel[b] = 0.0f;
            if (k < nr) {
                const float rho = tw0;
                const int f = rb.tag & ~ROW_FREE_BIT;
                float J[P_TOTAL];
#pragma unroll
                for (int q = 0; q < 7; ++q) { J[4 * q] = rw[q].x; J[4 * q + 1] = rw[q].y; J[4 * q + 2] = rw[q].z; J[4 * q + 3] = rw[q].w; }
                J[28] = rb.j28;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if (!alive[b]) continue;
                    // every LDS read of (row, system) first, behind one another: left to itself the compiler puts an s_waitcnt behind each of the 29
                    float sv[10], av[4], pu[6];
                    if constexpr (HOIST) {      // (one or two systems: the 14 voxel inputs of the entry are held across its rows)
#pragma unroll
                        for (int c = 0; c < 10; ++c) sv[c] = hsv[b][c];
#pragma unroll
                        for (int c = 0; c < 4; ++c) av[c] = hav[b][c];
                    } else {
                    sv[0] = lds[U_S(b) + i];
#pragma unroll
                    for (int c = 1; c < 10; ++c) sv[c] = lds[U_S(b) + so[c - 1]];
                    av[0] = lds[U_A(b) + i]; av[1] = lds[U_A(b) + sx]; av[2] = lds[U_A(b) + sy]; av[3] = lds[U_A(b) + sz];
                    }
                    const int o_up = SYS(b) + o_upose + 6 * f;
#pragma unroll
                    for (int q = 0; q < 6; ++q) pu[q] = lds[o_up + q];
                    const float4 u0 = *reinterpret_cast<const float4*>(lds + SYS(b) + o_ui), u1 = *reinterpret_cast<const float4*>(lds + SYS(b) + o_ui + 4), u2 = *reinterpret_cast<const float4*>(lds + SYS(b) + o_ui + 8);
                    __builtin_amdgcn_sched_barrier(0);
                    // (this file is compiled without FMA contraction and every multiply-add is spelled out: what a system's arithmetic is must not depend on how
                    // the compiler packs the systems of a launch — the first version contracted differently for NB = 1 and NB = 3 and differed in the 11th digit)
                    float d = fmaf(J[0], sv[0], J[10] * av[0]);
#pragma unroll
                    for (int c = 1; c < 10; ++c) d = fmaf(J[c], sv[c], d);
                    d = fmaf(J[11], av[1], d); d = fmaf(J[12], av[2], d); d = fmaf(J[13], av[3], d);
#pragma unroll
                    for (int q = 0; q < 6; ++q) d = fmaf(J[P_POSE + q], pu[q], d);
                    d = fmaf(J[P_INTR], u0.x, d); d = fmaf(J[P_INTR + 1], u0.y, d); d = fmaf(J[P_INTR + 2], u0.z, d); d = fmaf(J[P_INTR + 3], u0.w, d);
                    d = fmaf(J[P_INTR + 4], u1.x, d); d = fmaf(J[P_INTR + 5], u1.y, d); d = fmaf(J[P_INTR + 6], u1.z, d); d = fmaf(J[P_INTR + 7], u1.w, d);
                    d = fmaf(J[P_INTR + 8], u2.x, d);
                    const float t = rho * d;
                    pq_rows[b] = fmaf(t, d, pq_rows[b]);
                    self_s[b] = fmaf(J[0], t, self_s[b]); self_a[b] = fmaf(J[10], t, self_a[b]);
#pragma unroll
                    for (int c = 1; c < 10; ++c) C[b][c - 1] = fmaf(J[c], t, C[b][c - 1]);
                    C[b][9] = fmaf(J[11], t, C[b][9]); C[b][10] = fmaf(J[12], t, C[b][10]); C[b][11] = fmaf(J[13], t, C[b][11]);
                    if (owned_c) {
#pragma unroll
                        for (int q = 0; q < 9; ++q) cam9[b][q] = fmaf(J[P_INTR + q], t, cam9[b][q]);
                    }
                    tsel[b] = t;
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!p.fix_poses && owned_c) { fsel = f; pvalid = true; }
            }
            const float jp[6] = {rw[3].z, rw[3].w, rw[4].x, rw[4].y, rw[4].z, rw[4].w};      // pose columns 14..19 of the row
            if (reload >= 0) load_block(rb, reload);
            mr_table_add<NB>(pvalid, fsel, jp, tsel, lds, o_tag, o_val, tcount, D0, SB);
        };
        PH(4);
        consume(rwA, 0, 2);
        consume(rwB, 1, 3);
        consume(rwA, 2, 4);
        consume(rwB, 3, -1);
        consume(rwA, 4, -1);
        PH(5);
        if (owned_c) {
#pragma unroll
            for (int b = 0; b < NB; ++b) PQ_L(b)[i] += (double)pq_rows[b];
        }
        const size_t ac_c = in_c ? (size_t)a_c : 0;          // (rebuilt here: as a 64-bit value it would be live across the row loop)
        unsigned lr[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) lr[w] = __builtin_nontemporal_load(&lnbr[(size_t)(5 + w) * Acap + ac_c]);
        float eaw[6];
#pragma unroll
        for (int d = 0; d < 6; ++d) eaw[d] = __builtin_nontemporal_load(&eaw_sym[(size_t)d * Acap + ac_c]);
        // the NEXT tile (the last tile of the workgroup asks for itself again: unconditional loads keep the compiler's wait counts exact; its row blocks are
        // requested through zero-sized descriptors and move nothing): everything that depends on the tile number alone, now
        const int tk_n = tk + 1 < tk_end ? tk + 1 : tk;
        gm = gm_next;
        issue_idx(tk_n);
        if (tcount > TC - 16 && lane == 0) lds[D_FLAG] = __int_as_float(1);      // a table is nearly full: behind the barrier ALL tables are merged, in wave order
        const unsigned lall[LNBR_WORDS] = {0u, 0u, 0u, 0u, ln4_c, lr[0], lr[1]};
        const int r2y = unpack12(lall, 12), ryz = unpack12(lall, 13), r2z = unpack12(lall, 14), rxy = unpack12(lall, 15), rxz = unpack12(lall, 16), r2x = unpack12(lall, 17);
        PH(6);
        // ---- pull, system by system through the one column-sum buffer ----
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b > 0) PH(10);
            if (alive[b]) {
#pragma unroll
                for (int c = 0; c < NCOL; ++c) lds[C_L + c * T + i] = C[b][c];
            }
            const float ua_c = lds[U_A(b) + i];
            PH(7);
            __syncthreads();
            PH(8);
            if (b == 0) {          // the next tile's operator inputs (its halo indices are here by now) and first two row blocks: in flight for the rest of the pull phase
                issue_meta(); issue_gather();
                load_block(rwA, 0);
                load_block(rwB, 1);
                gm_next = tk + 2 < tk_end ? group_rows(tk + 2) : 0;
            }
            if (b == 0 && __float_as_int(lds[D_FLAG]) != 0) {                      // (workgroup-uniform: written before the barrier, cleared by the next tile's staging behind the next one)
                for (int w = 0; w < NW; ++w) { if (wave == w) mr_table_merge<NB>(lds, o_tag, o_val, tcount, D0, SB); __syncthreads(); }
            }
            float* const qacc = m.qacc0 + (size_t)m.sys[b] * m.vec;
            float* const qh = m.qh0 + (size_t)m.sys[b] * m.qh;
            if (in_c && alive[b]) {
                auto pull = [&](int col, int slot) { return slot < T ? lds[C_L + col * T + slot] : 0.0f; };
                float qs = self_s[b], qa = self_a[b];
                qs += pull(0, my) + pull(1, r2y) + pull(2, ryz) + pull(3, mz) + pull(4, r2z) + pull(5, mx) + pull(6, rxy) + pull(7, rxz) + pull(8, r2x);
                qa += pull(9, mx) + pull(10, my) + pull(11, mz);
                const int rg[6] = {sx, mx, sy, my, sz, mz};
#pragma unroll
                for (int d = 0; d < 6; ++d) qs += lds[TR_L(b) + (rg[d] < T ? rg[d] : T)];
                float ea = 0.0f, eq = 0.0f;
#pragma unroll
                for (int d = 0; d < 6; ++d) { const float diff = ua_c - lds[U_A(b) + rg[d]]; const float t = eaw[d] * diff; ea += t; eq += (rg[d] == ZSLOT ? 1.0f : 0.5f) * t * diff; }
                qa += tw3 * ea; if (owned_c) PQ_L(b)[i] += (double)(tw3 * eq);      // an edge whose other voxel is a list entry is seen from both sides (by their owners)
                qacc[a_c] = qs; qacc[chunk + a_c] = qa;
            }
#pragma unroll
            for (int q = 0; q < NQH; ++q) {
                const int hq = i + q * T;
                if (hq < H_c && alive[b]) {
                    const size_t o = (size_t)tile_c * HMAX + hq;
                    float hsum = 0.0f, hal = 0.0f;
                    const int j1 = hp_offs[hq + 1];
                    for (int j = hp_offs[hq]; j < j1; ++j) {
                        const int e = hp_list[j], col = e >> 10, ln2 = e & 1023;
                        const float v = col == 12 ? lds[TR_L(b) + ln2] : lds[C_L + col * T + ln2];
                        if (col >= 9 && col != 12) hal += v; else hsum += v;
                    }
                    qh[2 * o] = hsum; qh[2 * o + 1] = hal;
                }
            }
            PH(9);
            __syncthreads();       // the next system's column sums / the next tile's staging overwrite what other lanes are still pulling from
        }
    }
    PH(10);
    // ---- the camera block of every system: per-wave sums and tables -> the dense accumulator, in wave order -> one float row per workgroup ----
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            float v = cam9[b][q];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if (lane == 0) lds[D_CAM9W + b * CAMW + wave * 9 + q] = v;
        }
    }
    __syncthreads();
    for (int w = 0; w < NW; ++w) { if (wave == w) mr_table_merge<NB>(lds, o_tag, o_val, tcount, D0, SB); __syncthreads(); }
    if (threadIdx.x < 9) {
#pragma unroll
        for (int b = 0; b < NB; ++b) { float v = 0.0f; for (int w = 0; w < NW; ++w) v += lds[D_CAM9W + b * CAMW + w * 9 + threadIdx.x]; lds[SYS(b) + rs + threadIdx.x] = v; }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (!alive[b]) continue;
        float* const cam = m.cam0 + (size_t)m.sys[b] * m.cam;
        for (int q = threadIdx.x; q < nshared; q += T) {
            float v;
            if (q < 6 * K) { v = 0.0f; v += lds[SYS(b) + q]; }
            else v = lds[SYS(b) + rs + q - 6 * K];
            cam[(size_t)blockIdx.x * cam_stride + q] = v;
        }
    }
    if (m.pq0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) { if (alive[b]) block_partial_d(PQ_L(b)[i], m.pq0 + (size_t)m.sys[b] * m.part, 1, 0); }
    }
    PH(11); PH_FLUSH(NB); BT_FLUSH(NB);
#undef SYS
#undef U_S
#undef U_A
#undef TR_L
#undef C_L
#undef hp_list
#undef hp_offs
#undef PQ_L
}

#ifdef I3D_MR_PHASES
static void mr_phase_report() {
    unsigned long long h[3][MR_NPH + 2];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mr_phase), sizeof(h)) != hipSuccess) return;
    static const char* name[MR_NPH] = {"prologue", "issue loads", "staging writes (gather wait)", "staging barrier", "regulariser rows + hoist", "row loop", "reverse slots / Ea weights",
                                       "pull: column sums -> LDS", "pull: barrier 1", "pull: sums + halo walk + stores", "pull: barrier 2", "epilogue"};
    for (int nb = 0; nb < 3; ++nb) {
        if (!h[nb][MR_NPH + 1]) continue;
        double tot = 0.0; for (int k = 0; k < MR_NPH; ++k) tot += (double)h[nb][k];
        std::fprintf(stderr, "[mr phases] k_eg_tile_mr<%d>: %llu waves, %.1f tiles per wave, %.0f ticks per wave (s_memtime), %.0f per tile\n", nb + 1, h[nb][MR_NPH + 1], (double)h[nb][MR_NPH] / (double)h[nb][MR_NPH + 1],
                     tot / (double)h[nb][MR_NPH + 1], tot / (double)h[nb][MR_NPH]);
        for (int k = 0; k < MR_NPH; ++k) std::fprintf(stderr, "[mr phases]   <%d> %-32s %6.2f %%  %9.0f ticks per tile\n", nb + 1, name[k], 100.0 * (double)h[nb][k] / tot, (double)h[nb][k] / (double)h[nb][MR_NPH]);
    }
}
void mr_phase_report_now() { (void)hipDeviceSynchronize(); mr_phase_report(); }
#endif
#ifdef I3D_MR_BLOCKTIME
void mr_blocktime_report_now() {
    (void)hipDeviceSynchronize();
    static unsigned long long h[3][1024], n[3][1024];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mr_blk), sizeof(h)) != hipSuccess || hipMemcpyFromSymbol(n, HIP_SYMBOL(g_mr_blk_n), sizeof(n)) != hipSuccess) return;
    for (int nb = 0; nb < 3; ++nb) {
        int cnt = 0; double sum = 0.0, mx = 0.0, mn = 1e300; int imx = -1;
        std::vector<double> v;
        for (int b = 0; b < 1024; ++b) if (n[nb][b]) { const double t = (double)h[nb][b] / (double)n[nb][b]; v.push_back(t); sum += t; if (t > mx) { mx = t; imx = b; } if (t < mn) mn = t; ++cnt; }
        if (!cnt) continue;
        std::sort(v.begin(), v.end());
        std::fprintf(stderr, "[mr blocktime] k_eg_tile_mr<%d>: %d workgroups, %llu launches; ticks per launch: mean %.0f  min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f (workgroup %d)  max/mean %.3f\n",
                     nb + 1, cnt, n[nb][0], sum / cnt, mn, v[cnt / 10], v[cnt / 2], v[(9 * cnt) / 10], mx, imx, mx / (sum / cnt));
    }
}
#endif

// the largest number of systems one launch can take at K keyframes (the staged inputs of every system must fit the 160 KB of LDS): 3 at the bench's K = 200, 0 = never
int eg_tile_mr_max_systems(int K) {
    for (int nb = 3; nb >= 1; --nb) if (mr_lds_bytes(nb, K) <= I3D_LDS_LIMIT) return nb;
    return 0;
}

// One stream of the rows for the nsys (<= 3) systems sys[0..nsys) of a ladder batch.  The plan must be the 512-entry geometry with its pull lists (t.hp_off); the launch
// shape (workgroups, tiles per workgroup) is the single-system pass's, so that the per-workgroup partial rows of a system are the ones it gets alone.
// Returns the number of workgroups = p.q partials / camera rows per system (0: not launched).
int launch_eg_tile_mr(hipStream_t st, RowView r, OptParams p, TilePlan t, int nsys, const int* sys, const float* u0, float* qacc0, float* qh0, double* pq0, float* cam0, int cam_stride,
                      const PcgState* st0, const LadVec& lv) {
    if (r.A <= 0 || nsys < 1 || nsys > 3 || t.T != MR_T || !t.hp_off || r.slots != 5) return 0;
    int blocks = 0, tiles_per_block = 0;
    eg_tile_launch_shape(t, p.K, blocks, tiles_per_block);
    if (blocks <= 0) return 0;
    const int ntl = t.ntiles_own + t.n_ghost;          // sharded: the rank's own tiles, then the foreign tiles that hold its ghost entries (one launch, as k_eg_tile)
    const bool gh = t.n_ghost > 0 || t.tile_first != 0 || r.own0 != 0 || r.own1 < r.A;
    MrArgs m; m.u0 = u0; m.qacc0 = qacc0; m.qh0 = qh0; m.pq0 = pq0; m.cam0 = cam0; m.st0 = st0; m.vec = lv.vec; m.qh = lv.qh; m.cam = lv.cam; m.part = lv.part;
    for (int b = 0; b < 3; ++b) m.sys[b] = sys[b < nsys ? b : nsys - 1];
    const size_t lds = mr_lds_bytes(nsys, p.K);
#define I3D_MR2(NB, GH) do { \
        if (!set_dynamic_lds((const void*)k_eg_tile_mr<NB, GH>, "k_eg_tile_mr", lds, p.K)) return 0; \
        k_eg_tile_mr<NB, GH><<<blocks, MR_T, lds, st>>>(r, p, m, t.lnbr, t.eaw_sym, t.halo_idx, t.halo_cnt, tiles_per_block, ntl, cam_stride, r.gmax, t.hp_off, t.hp_src, t.tile_first, t.ntiles_own, t.ghost_tiles); } while (0)
#define I3D_MR(NB) do { if (gh) I3D_MR2(NB, true); else I3D_MR2(NB, false); } while (0)
    if (nsys == 1) I3D_MR(1); else if (nsys == 2) I3D_MR(2); else I3D_MR(3);
#undef I3D_MR2
#undef I3D_MR
    return blocks;
}

}  // namespace i3d
