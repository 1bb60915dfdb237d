// GPU-box self-test of wave_sum_quads (device/wave_ops.hpp): the v_permlane32_swap / v_permlane16_swap + DPP tree must put the wave sum of value 4 q + wave_quad_value(lane)
// into every lane of out[q].  Integer-valued inputs: every partial sum is exact in fp32, so any lane mix-up shows as a mismatch, not as round-off.
//   hipcc -O2 --offload-arch=gfx950 -I intrinsic3d_amd/csrc/device tools/experiments/wave_reduce_test.hip -o /tmp/wave_reduce_test && /tmp/wave_reduce_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "wave_ops.hpp"
using namespace i3d;

template <int MQ>
__global__ void k_test(float* out /* [MQ][64] */) {
    const int lane = threadIdx.x;
    float v[4 * MQ], s[MQ];
    for (int j = 0; j < 4 * MQ; ++j) v[j] = (float)((lane * 7 + j * 13 + (lane >> 3) * j) % 31);
    wave_sum_quads<MQ>(v, s);
    for (int q = 0; q < MQ; ++q) out[q * 64 + lane] = s[q];
}
template <int MQ> int run() {
    float* d; hipMalloc(&d, sizeof(float) * MQ * 64);
    k_test<MQ><<<1, 64>>>(d);
    std::vector<float> h(MQ * 64); hipMemcpy(h.data(), d, sizeof(float) * MQ * 64, hipMemcpyDeviceToHost); hipFree(d);
    int bad = 0;
    for (int q = 0; q < MQ; ++q) for (int lane = 0; lane < 64; ++lane) {
        const int row = lane >> 4, qv = ((row & 1) << 1) | (row >> 1), j = 4 * q + qv;
        float want = 0.0f; for (int l = 0; l < 64; ++l) want += (float)((l * 7 + j * 13 + (l >> 3) * j) % 31);
        if (h[q * 64 + lane] != want) { if (bad < 5) std::printf("MQ %d q %d lane %d: got %g want %g\n", MQ, q, lane, h[q * 64 + lane], want); ++bad; }
    }
    std::printf("wave_sum_quads<%d>: %s\n", MQ, bad ? "MISMATCH" : "ok");
    return bad;
}
int main() { const int bad = run<2>() + run<3>() + run<5>(); return bad ? 1 : 0; }
