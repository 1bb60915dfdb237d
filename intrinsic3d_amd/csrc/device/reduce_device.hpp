// Block-level fp64 reductions WITHOUT same-address global atomics.  Thousands of workgroups adding into one double serialise at the
// L2 (~11 ns per atomic on one address, MI355X_MICROARCH.md "fanin"): 9000 workgroups cost ~100 us, more than the kernels they end.
// Every workgroup stores its partial sum; a one-workgroup pass (k_reduce_partials, or the consumer itself) adds them up.
#pragma once
#include <hip/hip_runtime.h>

namespace i3d {

// sum of v over the workgroup (<= 1024 threads), valid in thread 0
static __device__ inline double block_sum_d(double v) {
    __shared__ double sm_red[16];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sm_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) for (int i = 0; i < (int)((blockDim.x + 63) >> 6); ++i) t += sm_red[i];
    __syncthreads();
    return t;
}
// partials[blockIdx.x * ncomp + comp] = workgroup sum of v
static __device__ inline void block_partial_d(double v, double* partials, int ncomp, int comp) {
    const double t = block_sum_d(v);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.x * ncomp + comp] = t;
}

}  // namespace i3d
