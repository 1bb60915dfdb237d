// LightingSVSH::estimate + computeVoxelShCoeffs on the device (lighting_svsh.cpp:93-110,166-346) — see sh_kernels.hip.
#include "context.hpp"

namespace i3d {

int estimate_sh(i3d_context* c, float, double, double, int*, double*, int32_t*, int, i3d_sh_stats*) {
    return ctx_fail(c, I3D_ERR_STATE, "i3d_estimate_sh: not built yet");
}

}  // namespace i3d

extern "C" int i3d_estimate_sh(i3d_context* c, float subvolume_size, double lambda_reg, double thres_shell, int32_t* num_subvolumes,
                               double* sh, int32_t* sub_index, int32_t cap, i3d_sh_stats* stats) {
    if (!c) return I3D_ERR_INVALID_ARGUMENT;
    return i3d::estimate_sh(c, subvolume_size, lambda_reg, thres_shell, num_subvolumes, sh, sub_index, cap, stats);
}
