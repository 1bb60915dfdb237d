// Launch wrappers of the SH lighting kernels (sh_kernels.hip).
#pragma once
#include "common.hpp"

namespace i3d {

struct ShParams { float size; double thres_shell; int single; };

void launch_sh_keys(hipStream_t st, GridView g, ShParams sp, unsigned long long* keys, int* iota);
void launch_sh_all_keys(hipStream_t st, GridView g, ShParams sp, unsigned long long* keys);
void launch_sh_assign(hipStream_t st, int M, const unsigned long long* sorted_keys, const unsigned long long* uniq, int S, int* sorted_sub);
int  sh_gram_chunks(long long longest_run);      // chunks a subvolume's run of that many voxels is cut into (one wave each)
int  sh_gram_slab_chunks(int S, int nchunk);     // chunks of every subvolume one launch takes (bounds the scratch and gridDim.y); the scratch below holds that many
// slice [m0, m1) of the subvolume-sorted list; one wave per (subvolume, chunk of its run), chunk blocks in part / wpart ([S * slab * 100] / [S * slab] scratch), summed in chunk order
hipError_t launch_sh_gram(hipStream_t st, GridView g, int m0, int m1, int S, int nchunk, const int* sorted_vox, const int* sorted_sub, double* part, double* wpart, double* gram /*[S][100]*/, double* wsub /*[S]*/);
void launch_sh_interpolate(hipStream_t st, GridView g, ShParams sp, const unsigned long long* uniq, int S, const double* sh, float* out);

}  // namespace i3d
