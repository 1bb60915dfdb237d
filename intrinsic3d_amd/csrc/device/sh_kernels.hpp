// Launch wrappers of the SH lighting kernels (sh_kernels.hip).
#pragma once
#include "common.hpp"

namespace i3d {

struct ShParams { float size; double thres_shell; int single; };

void launch_sh_keys(hipStream_t st, GridView g, ShParams sp, unsigned long long* keys, int* iota);
void launch_sh_all_keys(hipStream_t st, GridView g, ShParams sp, unsigned long long* keys);
void launch_sh_assign(hipStream_t st, int M, const unsigned long long* sorted_keys, const unsigned long long* uniq, int S, int* sorted_sub);
void launch_sh_gram(hipStream_t st, GridView g, int m0, int m1, int S, const int* sorted_vox, const int* sorted_sub, double* gram /*[S][100]*/, double* wsub /*[S]*/);      // slice [m0, m1) of the sorted list; one wave per subvolume
void launch_sh_interpolate(hipStream_t st, GridView g, ShParams sp, const unsigned long long* uniq, int S, const double* sh, float* out);

}  // namespace i3d
