// Wave-level (64 lanes) reductions shared by the row passes.
#pragma once
#include <hip/hip_runtime.h>

namespace i3d {

// One 16-byte piece of a stored Eg row.  The row set (1.46 GB on the bench workload) is streamed once per pass: NON-TEMPORAL, so that it does not
// displace the solver vectors and the plan data from the L2 / last-level cache (measured on the operator pass: 316 -> 299 us, and the vector
// kernels behind it get faster too).
typedef float v4f_nt __attribute__((ext_vector_type(4)));
static __device__ inline float4 ld_row(const float4* p) { const v4f_nt v = __builtin_nontemporal_load(reinterpret_cast<const v4f_nt*>(p)); return make_float4(v.x, v.y, v.z, v.w); }

// sum of v over the 64 lanes of the wave, returned to every lane: 4 DPP steps inside each row of 16 lanes (pure VALU, no LDS
// crossbar), then the four row totals through scalar registers
static __device__ inline float wave_sum(float v) {
    int x;
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true);  v += __int_as_float(x);    // quad_perm [1,0,3,2]
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true);  v += __int_as_float(x);    // quad_perm [2,3,0,1]
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true); v += __int_as_float(x);    // row_half_mirror
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true); v += __int_as_float(x);    // row_mirror
    const int b = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16)) +
           __int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48));
}

// Camera-block accumulation of one row slot across the wave.  The observation slots are stored in keyframe order, so the lanes of
// a wave (64 neighbouring voxels) mostly hold the SAME keyframe in a slot: per distinct keyframe the NV values are summed across the
// wave in registers and one lane issues the LDS atomics (an LDS float atomic costs ~2 cycles PER ACTIVE LANE, measured).  Waves with
// many distinct keyframes fall back to per-lane atomics into the lane's replica.
template <int NV>
static __device__ inline void wave_accumulate(bool valid, int f, const float (&val)[NV], float* lane_acc, float* wave_acc, int stride) {
    unsigned long long todo = __ballot(valid);
    const int lane = threadIdx.x & 63;
    for (int round = 0; todo != 0ull; ++round) {
        if (round == 3) {                                   // > 3 distinct keyframes in this slot of the wave
            if (valid && ((todo >> lane) & 1ull)) {
#pragma unroll
                for (int i = 0; i < NV; ++i) atomicAdd(&lane_acc[stride * f + i], val[i]);
            }
            break;
        }
        const int leader = __ffsll((long long)todo) - 1;
        const int f0 = __builtin_amdgcn_readlane(f, leader);
        const bool mine = valid && f == f0;
        float sum[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) sum[i] = wave_sum(mine ? val[i] : 0.0f);
        if (lane == leader) {
#pragma unroll
            for (int i = 0; i < NV; ++i) atomicAdd(&wave_acc[stride * f0 + i], sum[i]);
        }
        todo &= ~__ballot(mine);
    }
}


static __device__ inline void lds_add(float* p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// DETERMINISTIC camera block (DET): the pose columns of a wave's rows go into a table PRIVATE to the wave — one (keyframe tag, NV sums) entry per distinct keyframe
// the wave has met, found with one LDS read of the tags (lane = table slot) and a ballot — instead of LDS accumulators that several waves add to in whatever
// order they arrive.  The tables are merged into the workgroup's dense accumulator in wave order (inside the ordered section of a tile when a table fills up, and
// once at the end of the workgroup), so the fp32 sums of a pass do not depend on timing.  TC entries per wave (<= 64: one lane per slot); a wave that meets more
// distinct keyframes between two merges falls back to atomics on the dense accumulator (order-dependent; not seen on the bench scenes: slots are keyframe-ordered
// and neighbouring voxels choose the same keyframes).
template <int NV, int TC, class F>
static __device__ inline void wave_table_add(bool valid, int f, F val, float* lds, int o_tag, int o_val, int& count /* wave-uniform */, int o_dense, int dense_stride) {
    bool pending = valid;
    unsigned long long todo = __ballot(pending);
    const int lane = (int)(threadIdx.x & 63u);
    while (todo != 0ull) {
        const int leader = __ffsll((long long)todo) - 1;
        const int f0 = __builtin_amdgcn_readlane(f, leader);
        const bool mine = pending && f == f0;
        const int tg = __float_as_int(lds[o_tag + (lane & (TC - 1))]);                 // tags of unused slots are -1
        const unsigned long long hit = __ballot(tg == f0);
        int slot;
        if (hit != 0ull) slot = (__ffsll((long long)hit) - 1) & (TC - 1);
        else if (count < TC) { slot = count; if (lane == 0) lds[o_tag + slot] = __int_as_float(f0); count = count + 1; }
        else slot = -1;
        const bool lead = lane == leader;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float sum = wave_sum(mine ? val(i) : 0.0f);
            if (lead) lds_add(slot >= 0 ? &lds[o_val + slot * NV + i] : &lds[o_dense + dense_stride * f0 + i], sum);      // (no return value: fire and forget; the table is this wave's alone, its LDS operations execute in program order)
        }
        pending = pending && !mine;
        todo = __ballot(pending);
    }
}
// one wave's table -> the dense accumulator (distinct keyframes per entry: no two lanes meet), table cleared.  Called in wave order.
template <int NV, int TC>
static __device__ inline void wave_table_merge(float* lds, int o_tag, int o_val, int& count, int o_dense, int dense_stride) {
    const int lane = (int)(threadIdx.x & 63u);
    for (int e = lane; e < count * NV; e += 64) {
        const int sl = e / NV, i = e - sl * NV;
        const int f = __float_as_int(lds[o_tag + sl]);
        lds[o_dense + dense_stride * f + i] += lds[o_val + e];
        lds[o_val + e] = 0.0f;
    }
    if (lane < TC) lds[o_tag + lane] = __int_as_float(-1);
    count = 0;
}

// Wave sums of MANY values at once (gfx950: v_permlane32_swap / v_permlane16_swap).  A DPP tree costs 6 instructions per value (4 inside the rows of 16 lanes, 2 across
// them); here four values share the tree.  v[4q .. 4q+3] = A, B, C, D:
//     permlane32_swap(A, B) + add:  rows 0,1 hold A[i] + A[i+32], rows 2,3 hold B[i] + B[i+32]          (same for C, D)
//     permlane16_swap(AB, CD) + add: row 0 = A, row 1 = C, row 2 = B, row 3 = D, each lane (x[i] + x[i+32]) + (x[i+16] + x[i+48])
//     xor butterfly inside the rows (4 DPP adds): every lane of a row holds the total of the row's value
// = 10 instructions per four values.  out[q]: lane l holds the wave sum of value 4q + wave_quad_value(l).  The association is fixed and the same for every value
// whatever shares its registers (what tile_pass_mr.hip needs: a system's sums must not depend on the systems it is batched with).
typedef unsigned v2u_swap __attribute__((ext_vector_type(2)));
static __device__ inline int wave_quad_value(int lane) { const int row = lane >> 4; return ((row & 1) << 1) | (row >> 1); }      // rows 0,1,2,3 -> 0,2,1,3
template <int MQ>
static __device__ inline void wave_sum_quads(const float (&v)[4 * MQ], float (&out)[MQ]) {
#pragma unroll
    for (int q = 0; q < MQ; ++q) {
        const v2u_swap ab = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[4 * q]), __float_as_uint(v[4 * q + 1]), false, false);
        const v2u_swap cd = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[4 * q + 2]), __float_as_uint(v[4 * q + 3]), false, false);
        const float sab = __uint_as_float(ab.x) + __uint_as_float(ab.y), scd = __uint_as_float(cd.x) + __uint_as_float(cd.y);
        const v2u_swap r = __builtin_amdgcn_permlane16_swap(__float_as_uint(sab), __float_as_uint(scd), false, false);
        out[q] = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
#pragma unroll
    for (int q = 0; q < MQ; ++q) out[q] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(out[q]), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
#pragma unroll
    for (int q = 0; q < MQ; ++q) out[q] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(out[q]), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
#pragma unroll
    for (int q = 0; q < MQ; ++q) out[q] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(out[q]), 0x141, 0xF, 0xF, true));    // row_half_mirror
#pragma unroll
    for (int q = 0; q < MQ; ++q) out[q] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(out[q]), 0x140, 0xF, 0xF, true));    // row_mirror
}

// wave_table_add with the NV wave sums of a round taken together by wave_sum_quads (2.5 instructions per value instead of the 15 of wave_sum: 4 DPP adds, 4 readlanes,
// 3 adds) and ONE exec-masked block of LDS adds: the first lane of every row of 16 adds the values it holds (value 4 q + wave_quad_value(lane) of quad q).  Same table
// layout ([slot][NV]) and merge as wave_table_add; a different — still fixed — association of the wave sums.
template <int NV, int QC, int Q0, class F>
static __device__ inline void wave_table_chunks(bool mine, F& val, float* dst, int lane, int qv) {
    constexpr int MQ = (NV + 3) / 4, NQ = (MQ - Q0) < QC ? (MQ - Q0) : QC;
    if constexpr (NQ > 0) {
        float v[4 * NQ], sum[NQ];
#pragma unroll
        for (int i = 0; i < 4 * NQ; ++i) v[i] = (4 * Q0 + i < NV && mine) ? val(4 * Q0 + (4 * Q0 + i < NV ? i : 0)) : 0.0f;
        wave_sum_quads<NQ>(v, sum);
        if ((lane & 15) == 0) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) { if (4 * (Q0 + q) + 3 < NV || 4 * (Q0 + q) + qv < NV) lds_add(dst + 4 * (Q0 + q), sum[q]); }
        }
        wave_table_chunks<NV, QC, Q0 + QC>(mine, val, dst, lane, qv);
    }
}
template <int NV, int TC, int QC = 4, class F>
static __device__ inline void wave_table_add_quads(bool valid, int f, F val, float* lds, int o_tag, int o_val, int& count /* wave-uniform */, int o_dense, int dense_stride) {
    bool pending = valid;
    unsigned long long todo = __ballot(pending);
    const int lane = (int)(threadIdx.x & 63u);
    const int qv = wave_quad_value(lane);
    while (todo != 0ull) {
        const int leader = __ffsll((long long)todo) - 1;
        const int f0 = __builtin_amdgcn_readlane(f, leader);
        const bool mine = pending && f == f0;
        const int tg = __float_as_int(lds[o_tag + (lane & (TC - 1))]);                 // tags of unused slots are -1
        const unsigned long long hit = __ballot(tg == f0);
        int slot;
        if (hit != 0ull) slot = (__ffsll((long long)hit) - 1) & (TC - 1);
        else if (count < TC) { slot = count; if (lane == 0) lds[o_tag + slot] = __int_as_float(f0); count = count + 1; }
        else slot = -1;
        // (in chunks of QC quads: the values of a chunk are live together, 4 QC + QC registers; a value's tree does not depend on what shares its chunk)
        float* const dst = slot >= 0 ? &lds[o_val + slot * NV + qv] : &lds[o_dense + dense_stride * f0 + qv];      // (table full, not seen on the bench scenes: the dense accumulator, order-dependent)
        wave_table_chunks<NV, QC, 0>(mine, val, dst, lane, qv);
        pending = pending && !mine;
        todo = __ballot(pending);
    }
}

// Ordered section of a workgroup: wave w enters when waves 0 .. w-1 have left (a ticket in LDS; the LDS operations of a wave are older than the ticket it wrote).
// The ticket word must be 0 when the first wave arrives (reset it behind a barrier).
// No fence on either side: the LDS executes the DS instructions of a compute unit in the order they were issued, so the atomics a wave issued before its ticket
// store are performed before any DS instruction another wave issues after having READ that ticket value (a workgroup-scope fence would also wait for the wave's
// outstanding global loads — the pull phase's operands are in flight here).  s_sleep in the spin: a busy spin of up to 15 waves costs the working waves their LDS
// and issue slots (measured: 0.41 ms without it, 0.37 with).
static __device__ inline void ordered_enter(float* ticket, int wave) {
    while (__float_as_int(__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != wave) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
static __device__ inline void ordered_leave(float* ticket, int wave) {
    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63u) == 0) __hip_atomic_store(ticket, __int_as_float(wave + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace i3d
