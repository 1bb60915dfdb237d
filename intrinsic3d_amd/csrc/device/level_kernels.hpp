// Launch wrappers of level_kernels.hip.
#pragma once
#include "kernels.hpp"

namespace i3d {
void launch_recolor(hipStream_t st, GridView g, OptParams p, const FrameConst* frames, int nobs, uchar4* color_out);
void launch_shell_mark(hipStream_t st, GridView g, double thres, int* keep);
void launch_shell_crossing(hipStream_t st, GridView g, HashTable t, int* keep);
void launch_inv_rank(hipStream_t st, int N, const int* rank, int* inv);
void launch_keep_visit(hipStream_t st, int N, const int* inv, const int* keep_dev, int* keep_visit);
void launch_export_visit(hipStream_t st, GridView g, const int* inv_rank, const int* keep_dev, const int* scan_visit, int* kxyz, double* sdf, double* sdf_ref,
                         double* alb, float* w, uint8_t* rgb);
void launch_upsample(hipStream_t st, GridView g, HashTable t, const int* inv_rank, int* kxyz, double* sdf, double* sdf_ref, double* alb, float* w, uint8_t* rgb);
void launch_permute_staging(hipStream_t st, long long n, const int* perm, const int* kin, const double* s0, const double* s1, const double* al, const float* w,
                            const uint8_t* rgb, int* kout, double* o0, double* o1, double* oal, float* ow, uint8_t* orgb);
}  // namespace i3d

namespace i3d {
// mesh_kernels.hip — marching cubes on the resident grid (one lane per voxel in visit order)
void launch_mc_count(hipStream_t st, GridView g, HashTable t, const int* inv_rank, int refined, const unsigned char* ntri, int* counts);
void launch_mc_emit(hipStream_t st, GridView g, HashTable t, const int* inv_rank, int refined, int color_mode, const unsigned char* ntri, const signed char* tri,
                    int tri_stride, const int* offsets, float* pos, unsigned char* col, const uchar4* mode_color /* or null */);
// debug colour modes of SDFVisualization (vis_colors.hpp) painted per stored voxel, by device index
void launch_vis_colors(hipStream_t st, GridView g, int mode, float subvolume_size, const unsigned long long* sub_keys, int S, const double* sub_sh, uchar4* out);
}  // namespace i3d

namespace i3d {
// keyframe pyramids (Pyramid::create, rgbd/pyramid.cpp:59-166)
void launch_lum_from_bgr(hipStream_t st, int n, const uint8_t* bgr, float* lum);
void launch_pyr_down(hipStream_t st, int w, int h, const float* src, int ow, int oh, float* dst);
void launch_depth_down(hipStream_t st, int w, const float* src, int ow, int oh, float* dst);
}  // namespace i3d
namespace i3d {
void launch_resize_depth(hipStream_t st, int iw, int ih, const float* din, const float in_intr[4], int ow, int oh, const float out_intr[4], float* dout);
}
