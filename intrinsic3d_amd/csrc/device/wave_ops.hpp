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


}  // namespace i3d
