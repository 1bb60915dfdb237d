/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * C ABI of the CPU restatement of NVlabs/intrinsic3d's shading-optimisation hot path
 * (fp64, hand-rolled forward-mode duals, Ceres-2.1.0-equivalent LM + CGNR).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * library (include/intrinsic3d_hip.h) never links or calls it.
 *
 * PARITY PARTLY PINNED: the reference ships no tests, fixtures or golden vectors and its library cannot
 * be built in this environment (Ceres 2.1.0 / Eigen / OpenCV / Boost absent).  What CAN be executed of it
 * is: oracle/extract_ref.py compiles the reference's own residual functors, camera / shading / SDF
 * templates, observation weights, hash / rounding, grid container and marching cubes (cut out of
 * /root/reference by file:line at build time) into oracle/_ref/libref_i3d.so over stand-ins for the
 * absent Eigen / Ceres / OpenCV names, and tests/test_oracle_vs_ref.py holds this restatement against
 * it.  Still unpinned: everything Ceres itself does (LM, CGNR, block-Jacobi, Jets, bicubic spline —
 * restated from the published 2.1.0 algorithm on both sides), Eigen's float reduction order and
 * OpenCV's pyrDown / cvtColor.
 */
#ifndef I3D_ORACLE_H
#define I3D_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t iterations, lm_steps;
    double lambda_g, lambda_r0, lambda_r1, lambda_s0, lambda_s1, lambda_a;
    int32_t fix_poses, fix_intrinsics, fix_distortion;
    float occlusion_distance; int32_t num_observations;
    double thres_shell; int32_t grid_level, rgbd_level;
    int32_t cg_fixed_iterations;   /* -1 = Ceres' quadratic-model stopping rule */
    int32_t verbose;
    int32_t fix_sdf;               /* extension mirrored from the product config: all sdf blocks constant */
    int32_t carry_trust_radius;    /* extension mirrored from the product config */
} orc_opt_config;

typedef struct {
    int32_t rows[4]; double weight_sum[4]; double type_weight[4];
    int32_t valid_voxels, num_params, num_rows_reduced;
    double cost_initial, cost_final; int32_t lm_iterations, successful;
    int32_t cg_iters[50]; int32_t accepted[50]; int32_t n_attempts; double final_radius; int32_t termination;
} orc_iter_stats;

typedef struct { int32_t data_rows, reg_rows, subvolumes, lm_iterations, termination; double cost_initial, cost_final; } orc_sh_stats;

/* grid: built the way the reference does it — insert in file order into a Voxel grid, then convert() */
void*   orc_grid_from_voxels(float voxel_size, int64_t n, const int32_t* keys, const float* sdf, const float* weight, const uint8_t* color);
int64_t orc_grid_size(void* g);
float   orc_grid_voxel_size(void* g);
void    orc_grid_export(void* g, int32_t* keys, double* sdf, double* sdf_refined, double* albedo, float* weight, uint8_t* color);
void    orc_grid_import(void* g, const double* sdf_refined, const double* albedo, const uint8_t* color);  /* visit order; NULL = keep */
void    orc_grid_clear_outside_shell(void* g, double thres_shell);
void*   orc_grid_upsample(void* g);
void    orc_grid_free(void* g);

void*   orc_frames_create(int32_t K, int32_t levels);
void    orc_frames_set(void* fr, int32_t f, int32_t lvl, int32_t w, int32_t h, const float* lum, const float* depth, const uint8_t* bgr);
void    orc_frames_free(void* fr);

int32_t orc_optimize(void* g, void* fr, const orc_opt_config* cfg, double* intr, double* dist, double* poses,
                     const double* voxel_sh, orc_iter_stats* stats);

/* one residual collection at the current state (no solve); returns an opaque problem handle */
void*   orc_collect(void* g, void* fr, const orc_opt_config* cfg, const double* intr, const double* dist, const double* poses,
                    const double* voxel_sh, int32_t iteration);
void    orc_problem_counts(void* p, int32_t rows[4], double weight_sum[4], double type_weight[4]);
void    orc_problem_flags(void* p, uint8_t* active, uint8_t* ring_ok, uint8_t* fix_sdf, uint8_t* fix_alb);
/* Eg rows: centre voxel (visit index), frame, normalised weight, raw residual, 29 raw partials */
void    orc_problem_eg(void* p, int32_t* v, int32_t* f, double* weight, double* residual, double* J);
void    orc_problem_reg(void* p, int32_t type, int32_t* v, int32_t* dir, double* weight, double* residual);
/* gradient J^T r and diag(J^T J) of the scaled problem over GLOBAL parameter ids (2N + 6K + 9), fixed ones = 0; cost */
double  orc_problem_normal_eq(void* p, const orc_opt_config* cfg, double* gradient, double* jtj_diag, int32_t* is_free);
/* y = (J^T J) x over global ids with fixed columns removed */
void    orc_problem_jtj_apply(void* p, const orc_opt_config* cfg, const double* x, double* y);
void    orc_problem_free(void* p);
/* per voxel (visit order): relative gap between the n-th and (n+1)-th best observation weight at the current state (1 = no cut), -1 = no rows */
void    orc_observation_margins(void* g, void* fr, const orc_opt_config* cfg, const double* intr, const double* dist, const double* poses, double* margin);

int32_t orc_estimate_sh(void* g, float subvolume_size, double lambda_reg, double thres_shell, int32_t cg_fixed_iterations,
                        int32_t* num_subvolumes, double* sh /* cap*9 */, int32_t* sub_index /* cap*3 */, int32_t cap,
                        double* voxel_sh /* N*9 or NULL */, uint8_t* voxel_has_sh /* N or NULL */, orc_sh_stats* st);

/* keyframe pyramids (rgbd/pyramid.cpp:59-166) */
void    orc_lum_from_bgr(int32_t n, const uint8_t* bgr, float* lum);
void    orc_pyr_down(int32_t w, int32_t h, const float* src, float* dst /* (w/2)*(h/2) */);
void    orc_resize_depth(int32_t iw, int32_t ih, const float* din, const float* in_intr4, int32_t ow, int32_t oh, const float* out_intr4, float* dout);
void    orc_depth_down(int32_t w, int32_t h, const float* src, float* dst);
int32_t orc_recompute_colors(void* g, void* fr, const double* intr, const double* dist, const double* poses,
                             float occlusion_distance, int32_t num_observations);
/* Intrinsic3D::refine (intrinsic3d.cpp:206-290); *grid_io is replaced by the upsampled grids */
int32_t orc_refine(void** grid_io, void* fr, const orc_opt_config* cfg, int32_t num_grid_levels, int32_t num_rgbd_levels,
                   double thres_shell_factor, double thres_shell_factor_final, int32_t clear_distant_voxels,
                   float subvolume_size_sh, double sh_lambda_reg, double* intr, double* dist, double* poses, int32_t* levels_done);

/* TSDF fusion, the stage in front of the path (sparse_voxel_grid.cpp:301-467, sdf/algorithms.cpp:260-366, app_fusion.cpp:107-200).
 * cam = {fx, fy, cx, cy}; pose = camera-to-world 4x4 row-major; depth is Sensor::depth (thresholded, not yet eroded). */
void*   orc_fusion_create(float voxel_size, float depth_min, float depth_max, const float* clip6 /* may be NULL */);
void    orc_fusion_integrate(void* f, int32_t dw, int32_t dh, const float* dcam4, int32_t cw, int32_t ch, const float* ccam4,
                             const float* depth, const uint8_t* bgr, const float* pose16, int32_t erode_window);
void    orc_fusion_finish(void* f, int32_t correct_iterations);            /* correctSDF + clearInvalidVoxels */
int64_t orc_fusion_size(void* f);
void    orc_fusion_export(void* f, int32_t* keys, float* sdf, float* weight, uint8_t* color);   /* iteration (= file) order */
void    orc_fusion_free(void* f);
void    orc_erode_discontinuities(int32_t w, int32_t h, const float* in, int32_t window, float max_diff, float* out);
void    orc_compute_normals(int32_t w, int32_t h, const float* cam4, const float* depth, float depth_threshold, float* normals);

/* known-answer probes */
double  orc_shading_row(int32_t vx, int32_t vy, int32_t vz, const double* sh9, double pyr_scale, double voxel_size,
                        int32_t w, int32_t h, const float* lum, const double* params29, double* J29 /* may be NULL */);
void    orc_bicubic(const float* img, int32_t w, int32_t h, double r, double c, double* f, double* dfdr, double* dfdc);
void    orc_pose_to_mat(const double* pose6, float* R9, float* t3);
uint64_t orc_hash(int32_t x, int32_t y, int32_t z);
/* Ceres-equivalent LM + CGNR on the dense linear least-squares problem min ||A x - b||^2 (A row-major m x n, column blocks given);
 * returns the number of LM iterations; cg_iters[<=50] receives the PCG iteration count of every attempt */
int32_t orc_test_lm_dense(int32_t m, int32_t n, int32_t nblocks, const int32_t* block_sizes, const double* A, const double* b, double* x_io,
                          int32_t max_iterations, int32_t stop_after_first_success, int32_t cg_fixed_iterations, int32_t* cg_iters, double* costs2);
/* one CGNR solve of (A^T A + diag(D)^2) x = A^T b with the block-Jacobi preconditioner; returns the iteration count */
int32_t orc_test_cgnr(int32_t m, int32_t n, int32_t nblocks, const int32_t* block_sizes, const double* A, const double* b, const double* D,
                      int32_t cg_fixed_iterations, double* x_out);
int32_t orc_round_trunc(float v);

/* primitive probes (same helpers the pipeline above runs), held against oracle/_ref by tests/test_oracle_vs_ref.py */
double  orc_sdf_to_weight(double sdf, double truncation);
double  orc_varying_lambda(int32_t it, int32_t n, double l0, double l1);
int32_t orc_project_f(const float* fxfycxcy, const float* dist5, int32_t w, int32_t h, const float* p3, float* p2f, int32_t* p2i);
int32_t orc_voxel_visible(float max_occlusion_distance, const float* pt3, int32_t w, int32_t h, const float* depth, int32_t x, int32_t y);
float   orc_observation_weight(int32_t w, int32_t h, const float* depth, const float* n3, int32_t x, int32_t y, const float* v3);
void    orc_compute_color(int32_t n, const uint8_t* rgb, const float* weights, float* out3);
void    orc_filter(int32_t count, float* weights, int32_t keep, int32_t* order);
double  orc_chroma_weight(const uint8_t* c3, const uint8_t* cn3);
/* regulariser rows: type 1 = Er (x[7]: centre, +x,-x,+y,-y,+z,-z), 2 = Es (x[0], sdf0), 3 = Ea (x[2]) */
double  orc_reg_row(int32_t type, const double* x, double sdf0, double* J);
double  orc_sh_data_row(double luminance, const float* normal3, double albedo, const double* sh9, double* J9);
void    orc_world_to_voxel(float voxel_size, const float* p3, int32_t* out3);
/* marching cubes of the grid (sdf or sdf_refined): merged, cleaned mesh as MarchingCubes<VoxelSBR>::extractSurface returns it */
void*   orc_mc_extract(void* g, int32_t use_refined);
void    orc_mesh_counts(void* mesh, int64_t* nv, int64_t* nf);
void    orc_mesh_get(void* mesh, float* verts, uint8_t* colors, int32_t* faces);
void    orc_mesh_free(void* mesh);

#ifdef __cplusplus
}
#endif
#endif
