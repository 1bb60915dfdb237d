// K6 — the PCG operator pass q = J^T W J u, LDS-tiled (single-rank layout: list space == vector space).
//
// The work list is brick-Morton ordered, so 1024 consecutive entries (a TILE) are a compact patch of the surface shell: ~85-90 % of what
// a voxel's rows read from / contribute to lies in the same patch, the rest in a HALO of a few hundred entries of neighbouring patches.
// Once per outer iteration k_tile_plan gives every tile its halo (sorted list of the foreign entries its stencils reach) and every entry
// the LOCAL slots of its 12 stencil neighbours (9 forward slots of the Eg stencil + the -x,-y,-z ring entries; uint16: 0..1023 own tile,
// 1024.. halo, TP_ZSLOT = not in the list -> fixed parameter, reads 0 / contributions dropped).  The pass itself then never gathers or
// scatters through global memory:
//   * the operator input u of tile + halo is staged in LDS once per tile (coalesced for the tile, one gather per halo entry);
//   * a lane (= voxel, <= `slots` rows) reads its stencil values from LDS, forms t = W (J u) per row and PUSHES J^T t and the
//     regulariser terms into LDS accumulators (ds_add_f32; distinct lanes hit distinct voxels: conflict-free);
//   * the tile's totals leave with one coalesced store per unknown (qacc), the halo's with one coalesced store per halo slot (qh), and
//     k_halo_fold adds the halo sums to their owners (sorted (entry, slot) pairs, built once per outer iteration).
// All index / flag loads of an entry are issued BEFORE its first two row blocks and nothing in the entry's prologue touches global memory
// after them, so the row stream is never drained by an s_waitcnt on a younger gather (the in-order vmcnt was what held the first
// version of this pass, and k_eg_jtjp, at 0.40 ms against the 0.235 ms of the bare row stream).
// p.q needs no pass over q: p.(S J^T W J S p) = sum over rows of t (J u), accumulated here row by row (sum D^2 p^2: k_pcg_direction).
#include <cstring>
#include <cstdlib>
#include <climits>
#include "kernels.hpp"
#include "reduce_device.hpp"
#include "wave_ops.hpp"
#include "tile_device.hpp"
#include <rocprim/rocprim.hpp>

namespace i3d {

constexpr int TP_PLAN_LIST = 2048;              // capacity of a tile's halo list (the plan kernel sorts this many keys, two per thread)
constexpr int TP_NONE = 0x7f7f7f7f;             // padding of halo_idx (a byte pattern, so that a memset clears the tiles a rank does not run)
// the 12 stencil neighbours an entry READS (operator input): sdf slots 1..9 of the Eg row (shading_cost.cpp:90-129), then -x, -y, -z;
// the 6 further entries whose Eg stencil contains it (the others are -x,-y,-z again): pulled from when they are in the same tile
__device__ __host__ inline int tp_dir(int j) {
    constexpr int8_t D[18] = {NB_PY, NB_P2Y, NB_PYZ, NB_PZ, NB_P2Z, NB_PX, NB_PXY, NB_PXZ, NB_P2X, NB_MX, NB_MY, NB_MZ,
                              NB_M2Y, NB_MYZ, NB_M2Z, NB_MXY, NB_MXZ, NB_M2X};
    return D[j];
}

// ---- plan (once per outer iteration) ----------------------------------------------------------------------------------------------
// One workgroup per tile of T entries: hash set of the foreign entries its FORWARD stencils / rings reach -> compact -> bitonic sort
// (deterministic slot numbers, coalescing-friendly staging) -> local slots by binary search.  Sharded: a rank plans the tiles it runs
// (its own range + the foreign tiles that hold ghost entries); the others keep halo_cnt = 0 and padding.  lnbr: 18 local slots of 12 bits each, packed
// LSB first into 7 words per entry (28 B; zslot <= 3072 fits 12 bits): slots 0..11 = the 12 read slots (0..T-1 own tile, T.. halo, zslot = not a list
// entry), in words 0..4; slots 12..17 = the 6 extra reverse slots, own tile or zslot (foreign sources reach the entry through THEIR tile's halo accumulators).  halo_idx is padded with TP_NONE (those keys sort behind every real pair).
__global__ void __launch_bounds__(1024) k_tile_plan(RowView r, int T, int hmax, int hlimit /* <= hmax: more halo entries than this = overflow */, int tile_first, const int* __restrict__ tile_list, unsigned* __restrict__ lnbr,
                                                    int* __restrict__ halo_idx, int* __restrict__ halo_cnt, int* __restrict__ overflow) {
    __shared__ int hkeys[4096];
    __shared__ int hlist[TP_PLAN_LIST];
    __shared__ int cnt;
    const int tile = tile_list ? tile_list[blockIdx.x] : tile_first + (int)blockIdx.x, base = tile * T, a = base + threadIdx.x;
    const bool in = (int)threadIdx.x < T && a < r.A;
    const int zslot = T + hmax;
    for (int i = threadIdx.x; i < 4096; i += 1024) hkeys[i] = -1;
    hlist[threadIdx.x] = TP_NONE; hlist[threadIdx.x + 1024] = TP_NONE;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int la[18];
    // (all 18 neighbour slots requested before the first hash insert: interleaved with the insert loops every one of them was a round trip of its own)
#pragma unroll
    for (int j = 0; j < 18; ++j) la[j] = r.anbr[(size_t)tp_dir(j) * r.Acap + (in ? a : 0)];
#pragma unroll
    for (int j = 0; j < 18; ++j) {
        if (!in) la[j] = -1;
        if (j < 12 && la[j] >= 0 && (unsigned)(la[j] - base) >= (unsigned)T) {
            unsigned h = ((unsigned)la[j] * 2654435761u) >> 20;                 // 12 bits
            for (int probes = 0; probes < 4096; ++probes) {
                const int k = atomicCAS(&hkeys[h], -1, la[j]);
                if (k == -1 || k == la[j]) break;
                h = (h + 1) & 4095u;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 1024) if (hkeys[i] >= 0) { const int pos = atomicAdd(&cnt, 1); if (pos < TP_PLAN_LIST) hlist[pos] = hkeys[i]; }
    __syncthreads();
    const int H = cnt;
    if (H > hlimit) {                                                                             // the caller plans again with the other geometry / falls back to the untiled pass
        for (int i = threadIdx.x; i < hmax; i += 1024) halo_idx[(size_t)tile * hmax + i] = TP_NONE;    // (nothing stale or uninitialised reaches the sort of the pairs)
        if (threadIdx.x == 0) { *overflow = 1; halo_cnt[tile] = 0; }
        return;
    }
    // bitonic sort, ascending, of the first NS >= H list entries (everything behind the H keys is padding, which sorts last anyway): a 512-entry tile reaches a few
    // hundred foreign entries, not 2048 — sorting the whole list cost 0.55 ms per outer iteration on the bench workload (5.9 k tiles) against 0.29 ms for its 1024-entry tiles
    int NS = 64; while (NS < H) NS <<= 1;
    for (int k = 2; k <= NS; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int i = threadIdx.x + half * 1024, ixj = i ^ j;
                if (i < NS && ixj > i) { const int x = hlist[i], y = hlist[ixj]; const bool up = (i & k) == 0; if ((x > y) == up) { hlist[i] = y; hlist[ixj] = x; } }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < hmax; i += 1024) halo_idx[(size_t)tile * hmax + i] = hlist[i];
    if (threadIdx.x == 0) halo_cnt[tile] = H;
    if (!in) return;
    unsigned short ls[18];
#pragma unroll
    for (int j = 0; j < 18; ++j) {
        int slot = zslot;
        if (la[j] >= 0) {
            const unsigned off = (unsigned)(la[j] - base);
            if (off < (unsigned)T) slot = (int)off;
            else if (j < 12) { int lo = 0, hi = H - 1; while (lo < hi) { const int mid = (lo + hi) >> 1; if (hlist[mid] < la[j]) lo = mid + 1; else hi = mid; } slot = T + lo; }
        }
        ls[j] = (unsigned short)slot;
    }
    unsigned pw[LNBR_WORDS];
#pragma unroll
    for (int w = 0; w < LNBR_WORDS; ++w) pw[w] = 0u;
#pragma unroll
    for (int j = 0; j < 18; ++j) { const int bit = 12 * j, k = bit >> 5, sh = bit & 31; pw[k] |= (unsigned)ls[j] << sh; if (sh > 20) pw[k + 1] |= (unsigned)ls[j] >> (32 - sh); }
#pragma unroll
    for (int w = 0; w < LNBR_WORDS; ++w) lnbr[(size_t)w * r.Acap + a] = pw[w];
}

// symmetric albedo-edge weights: the Ea row of the edge (a, neighbour d) is created once, by whichever voxel is visited first
// (optimizer.cpp:259-279), so ea_w[d][a] is non-zero on one side only.  With w_sym[d][a] = ea_w[d][a] + ea_w[d^1][nb_d(a)] the Ea part of
// J^T W J u is a pure PULL: q_alb[a] = rho sum_d w_sym[d][a] (u_a - u_nb(d)) — no contribution has to be pushed to a neighbour.
__global__ void __launch_bounds__(256) k_eaw_sym(RowView r, const int* __restrict__ cflag /* sharded: 1 = rows built on this rank */, float* __restrict__ eaw_sym) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= r.A) return;
    const bool act = (r.aflags[a] & F_ACTIVE) != 0 && (!cflag || cflag[a]);
    // three batches of unconditional loads (a missing neighbour reads this entry's own slots) instead of a weight -> neighbour -> flags -> weight chain per direction
    float wo[6]; int nb[6];
#pragma unroll
    for (int d = 0; d < 6; ++d) { wo[d] = r.ea_w[(size_t)d * r.Acap + a]; nb[d] = r.anbr[(size_t)d * r.Acap + a]; }
    uint8_t nfl[6]; int ncf[6]; float wn[6];
#pragma unroll
    for (int d = 0; d < 6; ++d) { const int n = nb[d] >= 0 ? nb[d] : a; nfl[d] = r.aflags[n]; ncf[d] = cflag ? cflag[n] : 1; wn[d] = r.ea_w[(size_t)(d ^ 1) * r.Acap + n]; }
#pragma unroll
    for (int d = 0; d < 6; ++d) {
        float w = act ? wo[d] : 0.0f;
        if (nb[d] >= 0 && (nfl[d] & F_ACTIVE) && ncf[d]) w += wn[d];
        eaw_sym[(size_t)d * r.Acap + a] = w;
    }
}

// ---- the pass ---------------------------------------------------------------------------------------------------------------------
// Halo PULL lists (round 4, I3D_HALO_PULL=1 / the bit-reproducible mode).  What a tile's rows add to entries OUTSIDE the tile went through LDS float atomics into
// per-tile halo accumulators — in whatever order the waves arrive, the one order-dependent sum of the default operator pass besides the pose block, and ~70 cycles per
// wave-instruction.  The targets are known at plan time: lane L's stencil slot j points at local slot ls (>= T: halo slot ls - T).  This kernel inverts that map per
// tile: for every halo slot the (column, lane) pairs that feed it (CSR by halo slot; 9 sdf columns + 3 albedo columns of the lane's Eg rows, + the lane's Er row value
// towards its six ring neighbours = column 12), every segment sorted.  In the pass the halo slot's owner thread then adds C[column][lane] over its segment, after the
// barrier that completes the lane-private column sums: a pull like the in-tile one, fixed order, no atomics.
template <int T, int HMAX>
__global__ void __launch_bounds__(T) k_tile_pull_plan(RowView r, int tile_first, const int* __restrict__ tile_list, const unsigned* __restrict__ lnbr, const int* __restrict__ halo_cnt,
                                                      unsigned short* __restrict__ hp_off, unsigned short* __restrict__ hp_src, int* __restrict__ overflow) {
    constexpr int CAP = 4 * HMAX, ZSLOT = T + HMAX, PER = HMAX / T, NWV = T / 64;
    static_assert(HMAX % T == 0, "slots per thread");
    __shared__ int cnt[HMAX];
    __shared__ int off[HMAX + 1];
    __shared__ unsigned short list[CAP];
    __shared__ int wsum[NWV];
    const int tile = tile_list ? tile_list[blockIdx.x] : tile_first + (int)blockIdx.x, a = tile * T + (int)threadIdx.x;
    const bool in = a < r.A;
    for (int s2 = threadIdx.x; s2 < HMAX; s2 += T) cnt[s2] = 0;
    __syncthreads();
    unsigned ln[5];
#pragma unroll
    for (int w = 0; w < 5; ++w) ln[w] = in ? lnbr[(size_t)w * r.Acap + a] : 0u;
    // the lane's 18 outside references: (stencil slot j, column)
    constexpr int RJ[18] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 5, 0, 3, 5, 9, 0, 10, 3, 11};
    constexpr int RC[18] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 12, 12, 12, 12, 12};
    int hs[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) { const int sl = in ? unpack12(ln, RJ[k]) : ZSLOT; hs[k] = (sl >= T && sl != ZSLOT) ? sl - T : -1; if (hs[k] >= 0) atomicAdd(&cnt[hs[k]], 1); }
    __syncthreads();
    // exclusive scan of cnt -> off (PER consecutive slots per thread, wave scan, wave totals)
    int loc[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) { loc[q] = cnt[threadIdx.x * PER + q]; sum += loc[q]; }
    int inc = sum;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if ((int)(threadIdx.x & 63) >= o) inc += v; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < NWV; ++w) { if (w < (int)(threadIdx.x >> 6)) base += wsum[w]; total += wsum[w]; }
    int run = base + inc - sum;
#pragma unroll
    for (int q = 0; q < PER; ++q) { off[threadIdx.x * PER + q] = run; run += loc[q]; }
    if (threadIdx.x == T - 1) off[HMAX] = run;
    for (int s2 = threadIdx.x; s2 < HMAX; s2 += T) cnt[s2] = 0;
    __syncthreads();
    if (total > CAP) {                               // (wave-uniform) the caller plans again with the other geometry / falls back to the untiled pass, like a halo that does not fit
        if (threadIdx.x == 0) *overflow = 1;
        for (int s2 = threadIdx.x; s2 <= HMAX; s2 += T) hp_off[(size_t)tile * (HMAX + 1) + s2] = 0;
        return;
    }
#pragma unroll
    for (int k = 0; k < 18; ++k) if (hs[k] >= 0) { const int pos = off[hs[k]] + atomicAdd(&cnt[hs[k]], 1); list[pos] = (unsigned short)((RC[k] << 10) | (int)threadIdx.x); }
    __syncthreads();
    const int H = halo_cnt[tile];
    for (int s2 = threadIdx.x; s2 < H; s2 += T) {    // every segment ascending: the order the pass adds in
        const int j0 = off[s2], j1 = off[s2 + 1];
        for (int i2 = j0 + 1; i2 < j1; ++i2) { const unsigned short v = list[i2]; int j = i2 - 1; while (j >= j0 && list[j] > v) { list[j + 1] = list[j]; --j; } list[j + 1] = v; }
    }
    __syncthreads();
    for (int s2 = threadIdx.x; s2 <= HMAX; s2 += T) hp_off[(size_t)tile * (HMAX + 1) + s2] = (unsigned short)off[s2];
    for (int p2 = threadIdx.x; p2 < total; p2 += T) hp_src[(size_t)tile * CAP + p2] = list[p2];
}

// T lanes = T entries per tile.  SLOTS > 0: the row loop is unrolled for exactly that many observation slots and EVERY slot is requested
// (unused slots of a voxel are skipped per lane): straight-line code whose s_waitcnt the compiler can count exactly — with a run-time
// trip count and conditional refills it falls back to vmcnt(0) at the loop header, which drains the block meant to stay in flight.
// LDS float atomics cost ~70 cycles per wave instruction on gfx950 (measured: 14 ds_add_f32 per row doubled the kernel time), so the
// J^T accumulation inside the tile is a PULL from lane-private column sums staged in LDS; only what lands in the halo is pushed.
// wave_accumulate (wave_ops.hpp) on the kernel's own LDS array by integer offset.  wave_off is wave-uniform (a scalar register); the
// per-lane replica offset of the rare fallback is recomputed there, and lanes are tracked by a flag instead of a 64-bit lane mask: nothing
// of this helper lives in vector registers across the row loop.
// `val(i)` yields component i of this lane's contribution ON DEMAND and every wave sum is added to LDS at once: neither the NV values nor the
// NV sums are live together (the row loop has no register to spare: 128 per lane, and a spill reload there is a vmcnt(0) that drains the row stream)
template <int NV, class F>
static __device__ inline void wave_accumulate_lds(bool valid, int f, F val, float* lds, int reps, int rs, int wave_off, int stride, int base_off = 0) {
    bool pending = valid;
    unsigned long long todo = __ballot(pending);
    for (int round = 0; todo != 0ull; ++round) {
        if (round == 3) {                                   // > 3 distinct keyframes in this slot of the wave
            if (pending) {
                // (recomputed HERE on purpose: hoisted out of the row loop as a loop invariant it costs a register pair the loop does not have — in the
                // 1024-entry geometry the compiler spilled it and reloaded it behind every row block, each reload an s_waitcnt vmcnt(0))
                int lane_rep = (int)(threadIdx.x & (unsigned)(reps - 1));
                asm volatile("" : "+v"(lane_rep));
                const int lane_off = base_off + lane_rep * rs;
#pragma unroll
                for (int i = 0; i < NV; ++i) lds_add(&lds[lane_off + stride * f + i], val(i));
            }
            break;
        }
        const int leader = __ffsll((long long)todo) - 1;
        const int f0 = __builtin_amdgcn_readlane(f, leader);
        const bool mine = pending && f == f0;
        const bool lead = (int)(threadIdx.x & 63) == leader;
#pragma unroll
        for (int i = 0; i < NV; ++i) { const float sum = wave_sum(mine ? val(i) : 0.0f); if (lead) lds_add(&lds[wave_off + stride * f0 + i], sum); }
        pending = pending && !mine;
        todo = __ballot(pending);
    }
}

// GHOSTS: the tile list continues with foreign tiles that hold this rank's ghost entries (sharded runs).  A template parameter because the
// extra wave-uniform test costs the unrolled row loop registers: 12 instead of 4 spilled, 363 instead of 344 us on the single-GPU bench.
// (Software pipelining across the tiles of a workgroup — the next tile's input gathers / first row block issued before the pull phase of the current
// one — was built and measured in rounds 2 and 3, three sessions: never better than 1 %, slower whenever the values carried across the pull phase
// spilled; removed.  DESIGN.md section 9.)
// DET: fixed-order sums everywhere inside the workgroup — the halo pushes of a tile happen in an ORDERED SECTION (the waves take turns in wave order: a ticket in
// LDS; the LDS atomic unit serialises them anyway), the pose block goes through per-wave tables (above), the intrinsics / distortion sums through per-wave slots.
// With the fixed-order sums across workgroups (cam_part, p.q partials, the halo fold of k_pcg_step3) a PCG pass is then bit-reproducible from run to run.
// DETM (bit mask): 1 = ordered halo pushes, 2 = per-wave keyframe tables + per-wave camera slots.  Shipped: 0 (default) and 3 (I3D_DETERMINISTIC=1).
// Measured on the bench workload (profiles/r04_det_variants.json, one session): 0: 0.277-0.297 ms | 2: 0.296-0.302 (+4 %: the table look-up in the row loop) |
// 3: 0.370-0.384 (+30 %: sixteen waves taking turns per tile, with or without fences) — the price of bit-reproducibility, which is why it is opt-in.
template <int T, int HMAX, int SLOTS, bool GHOSTS, int DETM>
__global__ void __launch_bounds__(T, 4) k_eg_tile(RowView r, OptParams p, const float* __restrict__ u, const unsigned* __restrict__ lnbr,
                                                        const float* __restrict__ eaw_sym, const int* __restrict__ halo_idx, const int* __restrict__ halo_cnt,
                                                        double* __restrict__ shared, float* __restrict__ qacc, float* __restrict__ qh, double* __restrict__ pq_partials,
                                                        int reps, int tiles_per_block, int tile_first, int n_own /* tiles [tile_first, +n_own): this rank's own */,
                                                        const int* __restrict__ ghost_list /* then ntl - n_own foreign tiles holding ghost entries */, int ntl, const PcgState* __restrict__ state,
                                                        float* __restrict__ cam_partials /* or null: [gridDim.x][cam_stride] camera block of this workgroup (no atomics) */, int cam_stride,
                                                        const int* __restrict__ gmaxv /* = r.gmax as a restrict-qualified kernel argument: its wave-uniform loads become scalar loads (lgkmcnt), a
                                                                                         vector load here would put an s_waitcnt vmcnt(0) behind the row blocks just requested */,
                                                        const unsigned short* __restrict__ hp_off, const unsigned short* __restrict__ hp_src /* DETM & 4: the halo pull lists (k_tile_pull_plan) */) {
    if (state && state->done) return;
    constexpr bool DET = (DETM & 3) != 0, DORD = (DETM & 1) != 0, DTAB = (DETM & 2) != 0, HP = (DETM & 4) != 0;
    static_assert(!(DORD && HP), "a pulled halo needs no ordered pushes");
    constexpr int ZSLOT = T + HMAX, NSLOT = ZSLOT + 1;
    extern __shared__ float lds[];        // [reps][rs] pose acc | [9] | pad | camera part of u [6K+9] | pad | u_s,u_a [NSLOT] | qh_s,qh_a [HMAX] | tr [T+1] | C [12][T] | p.q [T] fp64
    const int K = p.K; const size_t Acap = r.Acap; const int A = r.A, chunk = r.chunk;
    const int nshared = 6 * K + 9;
    const int rs = (6 * K) | 1;
    // DET: the ticket, the per-wave intrinsics / distortion sums and the per-wave keyframe tables come FIRST, at compile-time offsets (everything behind them
    // depends on K; offsets that are constants or functions of the wave number cost no scalar register across the row loop)
    constexpr int NW = T / 64, TC = T == 1024 ? 64 : 32;
    constexpr int D_CAM9W = 4, D_TAG = D_CAM9W + ((NW * 9 + 3) & ~3), D_VAL = D_TAG + NW * TC, D0 = DET ? D_VAL + NW * TC * 6 : 0;      // (the region is laid out whenever any DETM bit is set)
    const int nacc = D0 + reps * rs + 9;                             // end of the dense camera accumulators [D0, nacc)
    const int o_upose = (nacc + 3) & ~3, o_u = (o_upose + nshared + 3) & ~3;
    for (int i = D0 + threadIdx.x; i < nacc; i += T) lds[i] = 0.0f;
    // every LDS access below is lds[<integer offset>]: pointers derived from `lds` and handed to helpers degrade to 64-bit generic pointers
    // (flat instructions, two registers each — they were what spilled inside the row loop)
#define upose (lds + o_upose)
#define u_s (lds + o_u)                                                        /* operator input: sdf / albedo unknowns by local slot */
#define u_a (lds + o_u + NSLOT)
#define qh_s (lds + o_u + 2 * NSLOT)                                           /* halo accumulators */
#define qh_a (lds + o_u + 2 * NSLOT + HMAX)
#define tr_l (lds + o_u + 2 * NSLOT + 2 * HMAX)                                /* Er row value of every tile entry (+ a zero at index T) */
#define C_l (lds + o_u + 2 * NSLOT + 2 * HMAX + T + 4)                         /* [12][T] column sums of the tile's Eg rows (slots 1..9, 11..13), lane-private until the pull */
    const size_t tail = 2 * (size_t)chunk;
    for (int i = threadIdx.x; i < nshared; i += T) upose[i] = u[tail + i];
    const int o_cam = D0 + reps * rs;
    const int o_wave_acc = __builtin_amdgcn_readfirstlane(D0 + ((threadIdx.x >> 6) & (reps - 1)) * rs);       // wave-uniform: a scalar
    float cam9[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) cam9[i] = 0.0f;
    // wave-uniform constants live in scalar registers (every VGPR counts: 128 per lane at 4 waves per SIMD, and a spill reload inside the row
    // loop is a vmcnt(0) that drains the row blocks in flight)
    const float tw0 = p.type_wf[0], tw1 = p.type_wf[1], tw2 = p.type_wf[2], tw3 = p.type_wf[3];
    const int tile0 = blockIdx.x * tiles_per_block;
#define ui (lds + o_upose + 6 * K)
    const int i = threadIdx.x;
    // the lane's running p.q (fp64) is parked in LDS: as a register pair it lived across the row loop, and in the 1024-entry geometry that pair was the
    // value the compiler spilled and reloaded behind every row block (each reload an s_waitcnt vmcnt(0) that drains the row stream)
    constexpr int NCOL = 12;
#define pq_l reinterpret_cast<double*>(lds + o_u + 2 * NSLOT + 2 * HMAX + T + 4 + NCOL * T)
    pq_l[i] = 0.0;
    // HP: the (column, lane) pull list of the tile in flight lives where the pushed halo accumulators were (2 HMAX floats = 4 HMAX list entries), its CSR offsets
    // behind the per-lane p.q
    constexpr int HPCAP = 4 * HMAX;
#define hp_list reinterpret_cast<unsigned short*>(lds + o_u + 2 * NSLOT)
#define hp_offs reinterpret_cast<unsigned short*>(lds + o_u + 2 * NSLOT + 2 * HMAX + T + 4 + NCOL * T + 2 * T)
    // DET: [ticket | 3 pad] [NW][9] per-wave intrinsics / distortion sums | [NW][TC] keyframe tags | [NW][TC][6] sums (at the front of the LDS, see above)
    constexpr int o_det = 0, o_cam9w = D_CAM9W;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#define o_tag (D_TAG + wave * TC)
#define o_val (D_VAL + wave * (TC * 6))
    int tcount = 0;                                                  // entries of this wave's table (wave-uniform)
    if (DTAB) {
        if ((threadIdx.x & 63u) < (unsigned)TC) lds[o_tag + (threadIdx.x & 63u)] = __int_as_float(-1);
        for (int e = threadIdx.x & 63u; e < TC * 6; e += 64) lds[o_val + e] = 0.0f;
    }

    // the tile in flight
    constexpr int NQH = (HMAX + T - 1) / T;
    int tile = 0, base = 0, a = 0, H = 0, nr_ld = 0; bool in = false, owned = false; size_t ac = 0;
    float us = 0.0f, ua = 0.0f; uint8_t fl = 0, rf_ld = 0; unsigned ln[5]; float hs[NQH], ha[NQH]; RowBlock rwA, rwB;
    constexpr int NQL = HP ? (HPCAP / 8 + T - 1) / T : 1, NQO = HP ? (HMAX + 1 + T - 1) / T : 1;      // 16-byte list chunks / offsets per thread
    uint4 hpl[NQL]; unsigned short hpo[NQO];
    const int tk_end = min(tile0 + tiles_per_block, ntl);
    // everything a tile needs besides its later rows is requested first (older than the row loads: waiting for it does not drain them)
    // one row = 120 B per lane: seven 16-byte planes + (column 28, keyframe id).  The 29 partials and the id, nothing else (the weight is folded in, RowView)
    // Addressing: the 64 entries of a wave share one 7680 B block per slot, so the rows are read through a BUFFER RESOURCE whose base is the wave's
    // slot-0 block (wave-uniform: scalar registers): a lane contributes one 32-bit offset register, the slot / plane offsets are scalar or immediate —
    // instead of a 64-bit per-lane pointer per stream (the row loop has no register to spare; a spilled pointer there is reloaded per row block, and
    // every reload is an s_waitcnt vmcnt(0) that drains the stream).  num_records bounds the wave to its own blocks.
    // The descriptor of slot k covers exactly the wave's block of that slot — or NOTHING when no entry of the wave's group has more than k rows (gm = r.gmax of
    // the group, k_group_rows): a buffer load outside its descriptor's range returns zeros without touching memory, so the unrolled, branch-free row loop below
    // keeps its exactly counted s_waitcnt while the empty slots of a group cost no bandwidth (round 3 streamed all five slots of every group: 36 % padding on
    // SURVEY.md 8(d)'s 4-voxel shell).  (One descriptor per slot, not one descriptor + a slot offset: the range check sees only voffset + the immediate.)
    const char* wave_rows = reinterpret_cast<const char*>(r.rows);      // the wave's slot-0 block (wave-uniform, set per tile)
    int gm = 0;                                                          // slots in use in the wave's group
    const unsigned lane16 = (threadIdx.x & 63u) * 16u;
    auto load_block = [&](RowBlock& rw, int k, int slots) {
        (void)slots;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wave_rows, 0, k < gm ? MAX_SLOTS * ROW_BLOCK_F4 * 16 : 0, 0x00020000);
#pragma unroll
        for (int q = 0; q < 7; ++q) { const v4u_b v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane16, k * (ROW_BLOCK_F4 * 16) + q * 1024, 2 /* nt */);
                                      rw.p[q] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)); }
        { const v2u_b t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane16 >> 1, k * (ROW_BLOCK_F4 * 16) + 64 * ROW_PLANES * 16, 2); rw.j28 = __uint_as_float(t.x); rw.tag = (int)t.y; }
    };
    // group row count of the wave in tile `tk` (scalar load); requested one tile ahead so that the descriptors of a tile never wait for it
    auto tile_of = [&](int tk) { return (GHOSTS && tk >= n_own) ? ghost_list[tk - n_own] : tile_first + tk; };
    auto group_rows = [&](int tk) -> int {
        const int wa0 = tile_of(tk) * T + (int)(threadIdx.x & ~63u);
        const int grp = __builtin_amdgcn_readfirstlane(wa0 < A ? (wa0 >> 6) : -1);
        return grp >= 0 ? gmaxv[grp] : 0;
    };
    auto issue_A = [&]() { load_block(rwA, 0, r.slots); };
    auto issue_B = [&]() { if (SLOTS > 1 || (SLOTS == 0 && r.slots > 1)) load_block(rwB, 1, r.slots); };
    // part 1: the operator input of the tile and of its halo — the only DEPENDENT loads of a tile (halo index -> gather).
    // part 2: flags, local slots (coalesced, issued with the row blocks).
    auto issue_in = [&](int tk) {
        const bool ghost = GHOSTS && tk >= n_own;
        tile = ghost ? ghost_list[tk - n_own] : tile_first + tk;
        base = tile * T; a = base + i;
        in = a < A;
        owned = a >= r.own0 && a < r.own1;          // p.q and the camera block count a row once: on the rank that owns its voxel
        ac = in ? (size_t)a : 0;
        { const int wa0 = base + (int)(threadIdx.x & ~63u);                                  // first entry of this wave: wave-uniform
          const unsigned grp = (unsigned)__builtin_amdgcn_readfirstlane(wa0 < A ? (wa0 >> 6) : 0);
          wave_rows = reinterpret_cast<const char*>(r.rows + (size_t)grp * (size_t)(r.slots * ROW_BLOCK_F4)); }
        H = halo_cnt[tile];
        us = in ? u[a] : 0.0f; ua = in ? u[chunk + a] : 0.0f;
        // branch-free (unconditional loads, padding slots gather entry 0 and are zeroed when staged): loads under divergent branches are waited for at
        // every join, which would serialise these dependent gathers
        int he[NQH];
#pragma unroll
        for (int q = 0; q < NQH; ++q) he[q] = __builtin_nontemporal_load(&halo_idx[(size_t)tile * HMAX + (i + q * T < HMAX ? i + q * T : 0)]);
#pragma unroll
        for (int q = 0; q < NQH; ++q) { const int e = (i + q * T < H) ? he[q] : 0; hs[q] = u[e]; ha[q] = u[chunk + e]; }
        if (HP) {          // the tile's pull list (one or two 16-byte chunks per thread) and its offsets: consumed by the staging below, long before the row loop
#pragma unroll
            for (int q = 0; q < NQL; ++q) { const int ch = i + q * T; hpl[q] = reinterpret_cast<const uint4*>(hp_src + (size_t)tile * HPCAP)[ch < HPCAP / 8 ? ch : 0]; }
#pragma unroll
            for (int q = 0; q < NQO; ++q) { const int o = i + q * T; hpo[q] = hp_off[(size_t)tile * (HMAX + 1) + (o <= HMAX ? o : HMAX)]; }
        }
    };
    auto issue_meta = [&]() {
        fl = in ? r.aflags[ac] : 0;
        nr_ld = r.nrows[ac];
        rf_ld = r.regflags[ac];
#pragma unroll
        for (int w = 0; w < 5; ++w) ln[w] = __builtin_nontemporal_load(&lnbr[(size_t)w * Acap + ac]);      // per-entry plan data: read once per pass, like the rows
    };
    int gm_next = tile0 < tk_end ? group_rows(tile0) : 0;
    for (int tk = tile0; tk < tk_end; ++tk) {
        gm = gm_next;
        issue_in(tk);
        issue_meta();
        issue_A();
        issue_B();
        gm_next = tk + 1 < tk_end ? group_rows(tk + 1) : 0;
        const bool ghost_tile = GHOSTS && tk >= n_own;     // a few ghost rows on the rim of a neighbour's tile: most of its waves have nothing to stream
        // ---- stage the operator input of tile + halo, clear the accumulators ----
        u_s[i] = us; u_a[i] = ua;
#pragma unroll
        for (int q = 0; q < NQH; ++q) { const int hq = i + q * T; if (hq < HMAX) { const bool hv = hq < H; u_s[T + hq] = hv ? hs[q] : 0.0f; u_a[T + hq] = hv ? ha[q] : 0.0f; if (!HP) { qh_s[hq] = 0.0f; qh_a[hq] = 0.0f; } } }
        if (HP) {
#pragma unroll
            for (int q = 0; q < NQL; ++q) { const int ch = i + q * T; if (ch < HPCAP / 8) { reinterpret_cast<uint2*>(hp_list)[2 * ch] = make_uint2(hpl[q].x, hpl[q].y); reinterpret_cast<uint2*>(hp_list)[2 * ch + 1] = make_uint2(hpl[q].z, hpl[q].w); } }      // (the list area is 8-byte aligned, not 16)
#pragma unroll
            for (int q = 0; q < NQO; ++q) { const int o = i + q * T; if (o <= HMAX) hp_offs[o] = hpo[q]; }
        }
        if (i == 0) { u_s[ZSLOT] = 0.0f; u_a[ZSLOT] = 0.0f; tr_l[T] = 0.0f; if (DET) lds[o_det] = __int_as_float(0); }      // (the ordered section of this tile starts at wave 0)
#pragma unroll
        for (int c = 0; c < NCOL; ++c) C_l[c * T + i] = 0.0f;
        const bool active = in && (fl & F_ACTIVE);
        const int nr = active ? nr_ld : 0;
        const uint8_t rf = active ? rf_ld : 0;
        if (!in) { constexpr AllZ<ZSLOT> az; for (int w = 0; w < 5; ++w) ln[w] = az.w[w]; }
        __syncthreads();
        // local slots: sdf stencil slot c (1..9) = unpack12(ln, c - 1); +x,+y,+z = 5,0,3; -x,-y,-z = 9,10,11
        const int sx = unpack12(ln, 5), sy = unpack12(ln, 0), sz = unpack12(ln, 3), mx = unpack12(ln, 9), my = unpack12(ln, 10), mz = unpack12(ln, 11);
        float self_s = 0.0f, self_a = 0.0f;
        // ---- regulariser rows (constant coefficients), while the first two row blocks are in flight ----
        {
            const int rg[6] = {sx, mx, sy, my, sz, mz};                    // ring order +x,-x,+y,-y,+z,-z
            float tr = 0.0f; double pq_pre = 0.0;
            if (rf & 1) {
                const float lap = ((((((-6.0f * us) + u_s[rg[0]]) + u_s[rg[1]]) + u_s[rg[2]]) + u_s[rg[3]]) + u_s[rg[4]]) + u_s[rg[5]];
                tr = tw1 * lap; if (owned) pq_pre += (double)(tr * lap);
                self_s += -6.0f * tr;
                if (!DORD && !HP) {
#pragma unroll
                    for (int d = 0; d < 6; ++d) if (rg[d] >= T && rg[d] != ZSLOT) lds_add(&qh_s[rg[d] - T], tr);      // ring neighbours of other tiles (ordered: in the ordered section; HP: pulled)
                }
            }
            tr_l[i] = tr;
            if ((rf & 2) && (rf & 4)) { const float ts = tw2 * us; if (owned) pq_pre += (double)(ts * us); self_s += ts; }
            if (rf & 7) pq_l[i] += pq_pre;
        }
#define Cme (C_l + i)
        float pq_rows = 0.0f;
        int nr_max = nr;
        if (SLOTS == 0) { for (int o = 32; o > 0; o >>= 1) nr_max = max(nr_max, __shfl_xor(nr_max, o, 64)); }
        // one row: t = W (J u), J^T t added to the lane's own column sums in LDS (plain read-modify-write: the slots are private to the lane)
        // (every register of a block is read below.  A loaded register that is never read gets handed to another value WHILE THE LOAD IS IN FLIGHT, and
        // the write-after-write hazard costs an s_waitcnt vmcnt(0) behind every refill: found in the ISA of the round-2 format, whose plane 7 carried the
        // residual the operator does not use)
        auto consume = [&](const RowBlock& rb, int k) {
            const float4 (&rw)[7] = rb.p;
            int fsel = 0; bool pvalid = false; float tsel = 0.0f;
            if (k < nr) {
                const float rho = tw0;                                 // the row weight is folded into the stored partials (Js = sqrt(w) J)
                const int f = rb.tag & ~ROW_FREE_BIT;
                float J[P_TOTAL];
#pragma unroll
                for (int q = 0; q < 7; ++q) { J[4 * q] = rw[q].x; J[4 * q + 1] = rw[q].y; J[4 * q + 2] = rw[q].z; J[4 * q + 3] = rw[q].w; }
                J[28] = rb.j28;
                float d = DTAB ? J[0] * u_s[i] + J[10] * u_a[i] : J[0] * us + J[10] * ua;      // (with the tables the row loop has no register left for the entry's own u: re-read from LDS)
#pragma unroll
                for (int c = 1; c < 10; ++c) d += J[c] * u_s[unpack12(ln, c - 1)];
                d += J[11] * u_a[sx] + J[12] * u_a[sy] + J[13] * u_a[sz];
                const int o_up = o_upose + 6 * f;
#pragma unroll
                for (int q = 0; q < 6; ++q) d += J[P_POSE + q] * lds[o_up + q];
#pragma unroll
                for (int q = 0; q < 9; ++q) d += J[P_INTR + q] * ui[q];
                const float t = rho * d;
                pq_rows += t * d;
                self_s += J[0] * t; self_a += J[10] * t;
#pragma unroll
                for (int c = 1; c < 10; ++c) Cme[(c - 1) * T] += J[c] * t;
                Cme[9 * T] += J[11] * t; Cme[10 * T] += J[12] * t; Cme[11 * T] += J[13] * t;
                if (!p.fix_poses && owned) { fsel = f; pvalid = true; tsel = t; }
                if (owned) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) cam9[q] += J[P_INTR + q] * t;
                }
            }
            // pose columns 14..19 of the row: plane 3 (.z, .w) and plane 4
            if (DTAB) wave_table_add<6, TC>(pvalid, fsel, [&](int q) { const float j = q == 0 ? rw[3].z : q == 1 ? rw[3].w : q == 2 ? rw[4].x : q == 3 ? rw[4].y : q == 4 ? rw[4].z : rw[4].w; return j * tsel; },
                                           lds, o_tag, o_val, tcount, D0, 6);
            else wave_accumulate_lds<6>(pvalid, fsel, [&](int q) { const float j = q == 0 ? rw[3].z : q == 1 ? rw[3].w : q == 2 ? rw[4].x : q == 3 ? rw[4].y : q == 4 ? rw[4].z : rw[4].w; return j * tsel; },
                                        lds, reps, rs, o_wave_acc, 6, D0);
        };
        if (SLOTS > 0 && ghost_tile && __ballot(nr > 0) == 0ull) {
            // (wave-uniform, a scalar compare) no row in this wave of a ghost tile: nothing to stream
        } else if (SLOTS > 0) {
#pragma unroll
            for (int k = 0; k < SLOTS; k += 2) {               // slot k is in rwA, slot k+1 (if any) in rwB; a buffer is refilled as soon as it is consumed
                consume(rwA, k);
                if (k + 2 < SLOTS) load_block(rwA, k + 2, SLOTS);
                if (k + 1 < SLOTS) {
                    consume(rwB, k + 1);
                    if (k + 3 < SLOTS) load_block(rwB, k + 3, SLOTS);
                }
            }
        } else {
            for (int k = 0; k < nr_max; k += 2) {
                consume(rwA, k);
                if (k + 2 < nr_max) load_block(rwA, k + 2, r.slots);
                if (k + 1 < nr_max) {
                    consume(rwB, k + 1);
                    if (k + 3 < nr_max) load_block(rwB, k + 3, r.slots);
                }
            }
        }
        if (owned) pq_l[i] += (double)pq_rows;
        // what the pull phase needs in addition (requested now, used behind the barrier): the 6 further reverse slots, the symmetric Ea weights
        unsigned lr[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) lr[w] = __builtin_nontemporal_load(&lnbr[(size_t)(5 + w) * Acap + ac]);
        float eaw[6];
#pragma unroll
        for (int d = 0; d < 6; ++d) eaw[d] = __builtin_nontemporal_load(&eaw_sym[(size_t)d * Acap + ac]);
        // ---- what lands in the halo is pushed (few lanes: the tile's outer shell) ----
        if (DORD) {      // ordered section: wave w enters when waves 0 .. w-1 have left (their LDS operations are older than the ticket they wrote)
            ordered_enter(&lds[o_det], wave);
            if (rf & 1) {
                const float tr = tr_l[i]; const int rg[6] = {sx, mx, sy, my, sz, mz};
#pragma unroll
                for (int d = 0; d < 6; ++d) if (rg[d] >= T && rg[d] != ZSLOT) lds_add(&qh_s[rg[d] - T], tr);      // ring neighbours of other tiles
            }
        }
        if (!HP && nr > 0) {
#pragma unroll
            for (int c = 1; c < 10; ++c) { const int sl = unpack12(ln, c - 1); if (sl >= T && sl != ZSLOT) lds_add(&qh_s[sl - T], Cme[(c - 1) * T]); }
            if (sx >= T && sx != ZSLOT) lds_add(&qh_a[sx - T], Cme[9 * T]);
            if (sy >= T && sy != ZSLOT) lds_add(&qh_a[sy - T], Cme[10 * T]);
            if (sz >= T && sz != ZSLOT) lds_add(&qh_a[sz - T], Cme[11 * T]);
        }
        if (DTAB && DORD) { if (tcount > TC - 16) wave_table_merge<6, TC>(lds, o_tag, o_val, tcount, D0, 6); }      // (wave-uniform) room for the next tile's keyframes
        if (DORD) ordered_leave(&lds[o_det], wave);
        // tables without an ordered section (pulled halo): a wave whose table is nearly full raises a flag, and behind the barrier ALL tables are merged in wave order
        if (DTAB && !DORD) { if (tcount > TC - 16 && (threadIdx.x & 63u) == 0u) lds[o_det] = __int_as_float(1); }
        const int a_c = a, H_c = H, tile_c = tile; const bool in_c = in, owned_c = owned; const float ua_c = DTAB ? u_a[i] : ua;
        __syncthreads();
        if (DTAB && !DORD) {
            if (__float_as_int(lds[o_det]) != 0) {                  // (workgroup-uniform: written before the barrier, cleared by the next tile's staging behind the next one)
                for (int w = 0; w < NW; ++w) { if (wave == w) wave_table_merge<6, TC>(lds, o_tag, o_val, tcount, D0, 6); __syncthreads(); }
            }
        }
        // ---- pull: every entry collects the column sums of the tile entries whose stencil contains it ----
        if (in_c) {
            // reverse entries: slot c of entry e is this entry <=> e = this entry's neighbour in the mirrored direction
            // c: 1 -y, 2 -2y, 3 -y-z, 4 -z, 5 -2z, 6 -x, 7 -x-y, 8 -x-z, 9 -2x; albedo 11 -x, 12 -y, 13 -z
            const unsigned lall[LNBR_WORDS] = {0u, 0u, 0u, 0u, ln[4], lr[0], lr[1]};       // slots 12..17 live in words 4..6
            const int r2y = unpack12(lall, 12), ryz = unpack12(lall, 13), r2z = unpack12(lall, 14), rxy = unpack12(lall, 15), rxz = unpack12(lall, 16), r2x = unpack12(lall, 17);
            auto pull = [&](int col, int slot) { return slot < T ? C_l[col * T + slot] : 0.0f; };
            float qs = self_s, qa = self_a;
            qs += pull(0, my) + pull(1, r2y) + pull(2, ryz) + pull(3, mz) + pull(4, r2z) + pull(5, mx) + pull(6, rxy) + pull(7, rxz) + pull(8, r2x);
            qa += pull(9, mx) + pull(10, my) + pull(11, mz);
            const int rg[6] = {sx, mx, sy, my, sz, mz};
#pragma unroll
            for (int d = 0; d < 6; ++d) qs += tr_l[rg[d] < T ? rg[d] : T];
            // Ea rows in pull form (k_eaw_sym): rho sum_d w_sym[d] (u_a - u_nb(d))
            float ea = 0.0f, eq = 0.0f;
#pragma unroll
            for (int d = 0; d < 6; ++d) { const float diff = ua_c - u_a[rg[d]]; const float t = eaw[d] * diff; ea += t; eq += (rg[d] == ZSLOT ? 1.0f : 0.5f) * t * diff; }
            qa += tw3 * ea; if (owned_c) pq_l[i] += (double)(tw3 * eq);      // an edge whose other voxel is a list entry is seen from both sides (by their owners)
            qacc[a_c] = qs; qacc[chunk + a_c] = qa;
        }
#pragma unroll
        for (int q = 0; q < NQH; ++q) {
            const int hq = i + q * T;
            if (hq < H_c) {
                const size_t o = (size_t)tile_c * HMAX + hq;
                if (HP) {        // this halo slot's sums: its (column, lane) segment, in list order
                    float hsum = 0.0f, hal = 0.0f;
                    const int j1 = hp_offs[hq + 1];
                    for (int j = hp_offs[hq]; j < j1; ++j) {
                        const int e = hp_list[j], col = e >> 10, lane = e & 1023;
                        const float v = col == 12 ? tr_l[lane] : C_l[col * T + lane];
                        if (col >= 9 && col != 12) hal += v; else hsum += v;
                    }
                    qh[2 * o] = hsum; qh[2 * o + 1] = hal;
                } else { qh[2 * o] = qh_s[hq]; qh[2 * o + 1] = qh_a[hq]; }
            }
        }
        __syncthreads();       // the staging of the next tile rewrites u / C / tr slots other lanes are still pulling from
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        float v = cam9[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) { if (DTAB) lds[o_cam9w + wave * 9 + q] = v; else if (v != 0.0f) lds_add(&lds[o_cam + q], v); }
    }
    __syncthreads();
    if (DTAB) {          // the waves' tables -> the dense accumulator, and their intrinsics / distortion sums, both in wave order
        for (int w = 0; w < NW; ++w) { if (wave == w) wave_table_merge<6, TC>(lds, o_tag, o_val, tcount, D0, 6); __syncthreads(); }
        if (threadIdx.x < 9) { float v = 0.0f; for (int w = 0; w < NW; ++w) v += lds[o_cam9w + w * 9 + threadIdx.x]; lds[o_cam + threadIdx.x] = v; }
        __syncthreads();
    }
    for (int q = threadIdx.x; q < nshared; q += T) {
        float v;
        if (q < 6 * K) { v = 0.0f; for (int rp = 0; rp < reps; ++rp) v += lds[D0 + rp * rs + q]; }
        else v = lds[o_cam + q - 6 * K];
        if (cam_partials) cam_partials[(size_t)blockIdx.x * cam_stride + q] = v;      // summed in a fixed order by k_pcg_step3's camera workgroups
        else if (v != 0.0f) atomicAdd(&shared[q], (double)v);
    }
    if (pq_partials) block_partial_d(pq_l[i], pq_partials, 1, 0);
#undef pq_l
#undef hp_list
#undef hp_offs
#undef o_tag
#undef o_val
#undef upose
#undef u_s
#undef u_a
#undef qh_s
#undef qh_a
#undef tr_l
#undef C_l
#undef ui
#undef Cme
}

// halo sums -> owners.  (ext_e, ext_pos): the (entry, halo slot) pairs of all tiles sorted by entry (padding last); the lane at the
// head of an entry's run adds the run (a handful of slots: the tiles around the entry) in its fixed order.
__global__ void __launch_bounds__(256) k_halo_fold(int n, const int* __restrict__ ext_e, const int* __restrict__ ext_pos, const float* __restrict__ qh,
                                                   float* __restrict__ qacc, int chunk, const PcgState* __restrict__ state) {
    if (state && state->done) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = ext_e[i];                 // (not non-temporal: the neighbouring lane re-reads it; measured 22 -> 26 us with nt)
    if (e >= TP_NONE || (i > 0 && ext_e[i - 1] == e)) return;
    float s = 0.0f, al = 0.0f;
    for (int j = i; j < n && ext_e[j] == e; ++j) { const float2 v = reinterpret_cast<const float2*>(qh)[ext_pos[j]]; s += v.x; al += v.y; }
    qacc[e] += s; qacc[chunk + e] += al;
}

__global__ void k_iota(int n, int* __restrict__ x) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = i; }

// tile geometry: T entries per tile / workgroup, HMAX halo slots.  Default 1024 / 2048 (one workgroup of 16 waves per CU, 122 KB of LDS): a third fewer
// tile boundaries than 512 / 1536 (two workgroups per CU) — same operator time once both row loops were spill-free, cheaper plan and halo fold: 40.1 vs
// 41.1 ms per iteration in the same-box A/B (profiles/r03_ab_variants.json).  I3D_EGT_TILE=512 selects the other; a plan that overflows falls back to it.
static int tp_T() { const char* e = std::getenv("I3D_EGT_TILE"); return (e && std::atoi(e) == 512) ? 512 : 1024; }      // (read per call: tests switch it inside one process)
static int tp_H() { return tp_T() == 1024 ? 2048 : 1536; }
int tile_plan_tiles(int A) { return (A + tp_T() - 1) / tp_T(); }
int tile_plan_hmax() { return tp_H(); }
size_t tile_plan_temp_bytes(int ntiles) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int*)nullptr, (int*)nullptr, (const int*)nullptr, (int*)nullptr, (size_t)ntiles * tp_H(), 0, 32, (hipStream_t)0);
    return bytes;
}

// returns hipSuccess or the first error; *overflow (device int, zeroed here) = 1 when a tile's halo does not fit.
// Plans the tiles [tile_first, tile_first + ntiles_own) and the n_ghost tiles of t.ghost_tiles; every other tile keeps an empty halo.
hipError_t launch_tile_plan(hipStream_t st, RowView r, TilePlan t, void* temp, size_t temp_bytes) {
    const int ntiles = tile_plan_tiles_of(r.A, t.T);
    if (ntiles <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(t.overflow, 0, sizeof(int), st); if (e != hipSuccess) return e;
    const int n = ntiles * t.hmax;
    // tests of the overflow handling: pretend the 512-entry geometry has fewer halo slots than it has (the 1024-entry fallback is not limited)
    const int limit512 = [] { const char* s = std::getenv("I3D_EGT_HMAX_LIMIT"); return s ? std::atoi(s) : 0; }();      // (read per plan: tests set it for one run)
    const int limit1024 = [] { const char* s = std::getenv("I3D_EGT_HMAX_LIMIT_1024"); return s ? std::atoi(s) : 0; }();
    const int lim = t.T == 512 ? limit512 : limit1024;
    const int hlimit = (lim > 0 && lim < t.hmax) ? lim : t.hmax;
    const bool all = t.tile_first == 0 && t.ntiles_own >= ntiles;
    if (!all) { e = hipMemsetAsync(t.halo_idx, 0x7f, sizeof(int) * (size_t)n, st); if (e != hipSuccess) return e;
                e = hipMemsetAsync(t.halo_cnt, 0, sizeof(int) * (size_t)ntiles, st); if (e != hipSuccess) return e; }
    if (t.ntiles_own > 0) k_tile_plan<<<t.ntiles_own, 1024, 0, st>>>(r, t.T, t.hmax, hlimit, t.tile_first, nullptr, t.lnbr, t.halo_idx, t.halo_cnt, t.overflow);
    if (t.n_ghost > 0) k_tile_plan<<<t.n_ghost, 1024, 0, st>>>(r, t.T, t.hmax, hlimit, 0, t.ghost_tiles, t.lnbr, t.halo_idx, t.halo_cnt, t.overflow);
    if (t.hp_off) {          // the halo pull lists of the tiles just planned (needs their lnbr words and halo counts)
        unsigned short* off = const_cast<unsigned short*>(t.hp_off); unsigned short* src = const_cast<unsigned short*>(t.hp_src);
        if (t.T == 1024) { if (t.ntiles_own > 0) k_tile_pull_plan<1024, 2048><<<t.ntiles_own, 1024, 0, st>>>(r, t.tile_first, nullptr, t.lnbr, t.halo_cnt, off, src, t.overflow);
                           if (t.n_ghost > 0) k_tile_pull_plan<1024, 2048><<<t.n_ghost, 1024, 0, st>>>(r, 0, t.ghost_tiles, t.lnbr, t.halo_cnt, off, src, t.overflow); }
        else { if (t.ntiles_own > 0) k_tile_pull_plan<512, 1536><<<t.ntiles_own, 512, 0, st>>>(r, t.tile_first, nullptr, t.lnbr, t.halo_cnt, off, src, t.overflow);
               if (t.n_ghost > 0) k_tile_pull_plan<512, 1536><<<t.n_ghost, 512, 0, st>>>(r, 0, t.ghost_tiles, t.lnbr, t.halo_cnt, off, src, t.overflow); }
    }
    k_iota<<<(n + 255) / 256, 256, 0, st>>>(n, t.iota);
    // keys are list entries < A or the padding TP_NONE: sort on as many low bits as separate them (the padding must compare above every entry) — 22 or 23 bits on the bench
    // workloads, three 8-bit passes instead of four; the stable sort leaves equal keys (and all padding) in slot order either way
    int key_bits = 32;
    for (int b = 1; b < 32; ++b) { const unsigned mask = (1u << b) - 1u; if ((1u << b) >= (unsigned)r.A && ((unsigned)TP_NONE & mask) >= (unsigned)r.A) { key_bits = b; break; } }
    e = rocprim::radix_sort_pairs(temp, temp_bytes, (const int*)t.halo_idx, t.ext_e, (const int*)t.iota, t.ext_pos, (size_t)n, 0, key_bits, st);
    if (e != hipSuccess) return e;
    if (t.ext_off) launch_ext_offsets(st, n, t.ext_e, r.chunk, t.ext_off);       // CSR offsets of the sorted pairs: k_pcg_step3 folds the halo sums itself (pcg_fused.hip)
    return hipGetLastError();
}

// after the build kernel has written the Ea weights of this outer iteration
void launch_eaw_sym(hipStream_t st, RowView r, TilePlan t, const int* cflag) { if (r.A > 0) k_eaw_sym<<<(r.A + 255) / 256, 256, 0, st>>>(r, cflag, t.eaw_sym); }

// LDS request and launch shape of the pass for a plan: shared with the multi-system pass (tile_pass_mr.hip), whose per-workgroup partial rows must be the ones a
// system gets from this kernel
struct EgtShape { int detm, reps, per_cu, blocks, tiles_per_block; size_t lds; };
template <int T, int HMAX>
static EgtShape egt_shape(const TilePlan& t, int K, int num_cu) {
    const int nshared = 6 * K + 9, rs = (6 * K) | 1;
    static const bool ordered = [] { const char* e = std::getenv("I3D_EGT_ORDERED"); return e && e[0] == '1'; }();
    const bool pull = t.pull && t.hp_off != nullptr && !(t.det && ordered);
    EgtShape s; s.detm = t.det ? (pull ? 6 : 3) : (pull ? 4 : 0);
    const bool det = (s.detm & 3) != 0;
    constexpr int NW = T / 64, TC = T == 1024 ? 64 : 32;
    const int det_words = det ? 4 + ((NW * 9 + 3) & ~3) + NW * TC * 7 : 0;      // ticket, per-wave camera sums, per-wave keyframe tables (at the front of the LDS)
    auto lds_bytes = [&](int reps) { const int nacc = det_words + reps * rs + 9; const int o_upose = (nacc + 3) & ~3, o_u = (o_upose + nshared + 3) & ~3;
                                     return (size_t)(o_u + 2 * (T + HMAX + 1) + 2 * HMAX + T + 4 + 12 * T + 2 * T /* p.q per lane, fp64 */ + (pull ? (HMAX + 4) / 2 : 0) /* pull-list offsets */) * sizeof(float); };
    const size_t budget = (T == 512 ? 79 : 158) * 1024;
    s.reps = (s.detm & 2) ? 1 : 4;                                  // replicas only serve the rare > 3-keyframe fallback of wave_accumulate (DET: one dense accumulator, the waves own tables)
    while (s.reps > 1 && lds_bytes(s.reps) > budget) s.reps >>= 1;
    s.lds = lds_bytes(s.reps);
    s.per_cu = (T == 512 && s.lds <= budget) ? 2 : 1;
    { static int knob = -1; if (knob < 0) { const char* e = std::getenv("I3D_EGT_WG_PER_CU"); knob = e ? std::atoi(e) : 0; } if (knob > 0) s.per_cu = knob; }
    // ONE launch over the rank's own tiles followed by the foreign tiles that hold its ghost entries (the short ghost tiles fill the idle tail of
    // the last round of own tiles instead of paying a launch and a round of their own)
    const int ntl = t.ntiles_own + t.n_ghost;
    s.blocks = 0; s.tiles_per_block = 0;
    if (ntl > 0) {
        s.blocks = ntl < s.per_cu * num_cu ? ntl : s.per_cu * num_cu;
        s.tiles_per_block = (ntl + s.blocks - 1) / s.blocks;
        // between one and two rounds of resident workgroups (a rank's share at 8 GPUs): one tile per workgroup — the short second round runs
        // on a nearly empty chip (measured at 557 tiles: 61.4 vs 64.0 us for 279 workgroups of two tiles)
        if (ntl > s.blocks && ntl < 2 * s.blocks) { s.tiles_per_block = 1; s.blocks = ntl; }
    }
    return s;
}
static int egt_num_cu() {
    static int num_cu = 0;
    if (!num_cu) { int dev = 0; (void)hipGetDevice(&dev); hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, dev); num_cu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
    return num_cu;
}
void eg_tile_launch_shape(const TilePlan& t, int K, int& blocks, int& tiles_per_block) {
    const EgtShape s = t.T == 1024 ? egt_shape<1024, 2048>(t, K, egt_num_cu()) : egt_shape<512, 1536>(t, K, egt_num_cu());
    blocks = s.blocks; tiles_per_block = s.tiles_per_block;
}

template <int T, int HMAX>
static int launch_eg_tile_t(hipStream_t st, RowView r, OptParams p, const float* u, TilePlan t, double* shared, float* qacc, double* pq_partials, const PcgState* state, int num_cu,
                            float* cam_partials, int cam_stride) {
    // I3D_DETERMINISTIC=1: fixed-order sums inside the workgroup as well (ordered halo pushes, per-wave keyframe tables) — every kernel of an outer iteration is then
    // bit-reproducible from run to run (tools/flake_hunt.py: 0.0 on every field) at ~20 % lower throughput; the default keeps the LDS atomics of round 3 HERE and
    // only here (gradient, column norms, SH Gram blocks, camera block and halo fold across workgroups are fixed-order in both modes)
    // halo sums: pushed with LDS atomics (default), or PULLED over the plan's lists (t.pull: I3D_HALO_PULL=1, and always in the bit-reproducible mode, where the pull
    // replaces the ordered pushes of the first version — I3D_EGT_ORDERED=1 brings those back for A/B runs)
    const EgtShape sh = egt_shape<T, HMAX>(t, p.K, num_cu);
    const int detm = sh.detm, reps = sh.reps; const size_t lds = sh.lds; const bool pull = (detm & 4) != 0;
    // tiles in units of T: the plan counts tiles of tp_T() == T
    int written = 0;
    // ONE launch over the rank's own tiles followed by the foreign tiles that hold its ghost entries (the short ghost tiles fill the idle tail of
    // the last round of own tiles instead of paying a launch and a round of their own)
    {
        const int ntl = t.ntiles_own + t.n_ghost;
        if (ntl > 0) {
            const int blocks = sh.blocks, tiles_per_block = sh.tiles_per_block;
#define I3D_EGT(SL, GH) do { if (detm == 3) I3D_EGT2(SL, GH, 3); else if (detm == 6) I3D_EGT2(SL, GH, 6); else if (detm == 4) I3D_EGT2(SL, GH, 4); else I3D_EGT2(SL, GH, 0); } while (0)
#define I3D_EGT2(SL, GH, DT) do { \
        if (!set_dynamic_lds((const void*)k_eg_tile<T, HMAX, SL, GH, DT>, "k_eg_tile", lds, p.K)) break; \
        k_eg_tile<T, HMAX, SL, GH, DT><<<blocks, T, lds, st>>>(r, p, u, t.lnbr, t.eaw_sym, t.halo_idx, t.halo_cnt, shared, qacc, t.qh, pq_partials, reps, tiles_per_block, t.tile_first, t.ntiles_own, \
                                                   t.ghost_tiles, ntl, state, cam_partials, cam_stride, r.gmax, pull ? t.hp_off : nullptr, pull ? t.hp_src : nullptr); } while (0)
            const bool gh = t.n_ghost > 0;
            // the shipped num_observations (data/intrinsic3d.yml): unrolled row loop.  (The run-time loop, which skips the slots no lane of a wave uses, is
            // slower even where 36 % of the slots are empty: 0.577 vs 0.483 ms on --band 2, 0.336 vs 0.272 on the default workload.)
            if (r.slots == 5) { if (gh) I3D_EGT(5, true); else I3D_EGT(5, false); }
            else { if (gh) I3D_EGT(0, true); else I3D_EGT(0, false); }
#undef I3D_EGT
#undef I3D_EGT2
            written = blocks;
        }
    }
    return written;                                          // number of p.q partials written
}

int launch_eg_tile(hipStream_t st, RowView r, OptParams p, const float* u, TilePlan t, double* shared, float* qacc, double* pq_partials, const PcgState* state,
                   float* cam_partials, int cam_stride) {
    if (r.A <= 0) return 0;
    const int num_cu = egt_num_cu();
    if (t.T == 1024) return launch_eg_tile_t<1024, 2048>(st, r, p, u, t, shared, qacc, pq_partials, state, num_cu, cam_partials, cam_stride);
    return launch_eg_tile_t<512, 1536>(st, r, p, u, t, shared, qacc, pq_partials, state, num_cu, cam_partials, cam_stride);
}
// halo accumulators of all tiles -> the per-entry accumulators (sorted by target entry at plan time; timed as its own category)
void launch_halo_fold(hipStream_t st, RowView r, TilePlan t, float* qacc, const PcgState* state) {
    if (r.A <= 0) return;
    const int n = tile_plan_tiles_of(r.A, t.T) * t.hmax;
    k_halo_fold<<<(n + 255) / 256, 256, 0, st>>>(n, t.ext_e, t.ext_pos, t.qh, qacc, r.chunk, state);
}
int tile_plan_T() { return tp_T(); }

}  // namespace i3d
