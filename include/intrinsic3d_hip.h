/* intrinsic3d_hip.h — C ABI of the MI355X-native (gfx950, HIP) implementation of NVlabs/intrinsic3d's
 * voxel-SDF shading-optimisation hot path.
 *
 * The reference has no FFI layer; the boundary this library replaces is the C++ entry point
 *
 *     bool nv::Optimizer::optimize(SDFColorization&, Optimizer::Data&, Optimizer::ImageFormationModel&)
 *                                   libintrinsic3d/include/nv/refinement/optimizer.h:123-125
 *                                   libintrinsic3d/src/refinement/optimizer.cpp:109-173
 *
 * and its sibling  nv::LightingSVSH::estimate() + computeVoxelShCoeffs()
 *                                   libintrinsic3d/include/nv/lighting/lighting_svsh.h:52,60
 *                                   libintrinsic3d/src/lighting/lighting_svsh.cpp:93-110,166-346
 *
 * i.e. everything the reference hands to Ceres (nls_solver.cpp:190-367).  Plain pointers and sizes only; no
 * C++/torch types cross this boundary.  All functions return 0 on success and a non-zero i3d_status otherwise
 * (the reference's convention is bool + std::cerr, optimizer.cpp:113-114); nothing throws.  One host thread
 * drives a context.  INTEGRATION.md shows the reference-side shim that binds these entry points.
 */
#ifndef INTRINSIC3D_HIP_H
#define INTRINSIC3D_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct i3d_context i3d_context;

typedef enum {
    I3D_OK = 0,
    I3D_ERR_INVALID_ARGUMENT = 1,   /* null grid / iterations < 1 (optimizer.cpp:113-114 returns false) */
    I3D_ERR_NO_DEVICE = 2,          /* no HIP device: the product path never falls back to the CPU */
    I3D_ERR_HIP = 3,
    I3D_ERR_STATE = 4,              /* grid / frames / camera not set */
    I3D_ERR_CAPACITY = 5,
    I3D_ERR_COMM = 6,
    I3D_ERR_IO = 7                  /* file missing / truncated (SparseVoxelGrid::load, Camera::load return false) */
} i3d_status;

/* ---- lifetime ------------------------------------------------------------------------------------------ */
int  i3d_create(int32_t device_ordinal, i3d_context** out);
void i3d_destroy(i3d_context* ctx);
/* last error text of this context (ctx may be NULL for creation errors) */
const char* i3d_last_error(const i3d_context* ctx);
const char* i3d_version(void);

/* ---- voxel grid: SparseVoxelGrid<VoxelSBR> (sparse_voxel_grid.h:69-161) as flat arrays ---------------------
 * Arrays are in the caller's grid->begin()..end() iteration order.  That "visit order" is part of the
 * reference's result (the albedo-regulariser edge set depends on it, optimizer.cpp:264-279) and is kept as
 * a per-voxel rank on the device. */
typedef struct {
    int64_t        num_voxels;
    float          voxel_size;     /* SparseVoxelGrid::voxelSize() */
    float          truncation;     /* SparseVoxelGrid::truncation() = 5*voxel_size (sparse_voxel_grid.cpp:48) */
    const int32_t* keys;           /* [N][3] voxel coordinates */
    const double*  sdf;            /* VoxelSBR::sdf          (read-only for the path) */
    const double*  sdf_refined;    /* VoxelSBR::sdf_refined  (optimised in place)     */
    const double*  albedo;         /* VoxelSBR::albedo       (optimised in place)     */
    const float*   weight;         /* VoxelSBR::weight */
    const uint8_t* color;          /* [N][3] VoxelSBR::color (r,g,b) */
} i3d_grid_view;

int i3d_set_grid(i3d_context* ctx, const i3d_grid_view* grid);
/* write-back of the only per-voxel fields the path mutates; either pointer may be NULL */
int i3d_get_grid(i3d_context* ctx, double* sdf_refined, double* albedo);
/* overwrite the unknowns / colours of the resident grid (visit order); NULL = keep */
int i3d_update_grid(i3d_context* ctx, const double* sdf_refined, const double* albedo, const uint8_t* color);

/* ---- keyframes: ImageFormationModel::rgbd_pyr + ShadingCostData (optimizer.h:107-115, shading_cost.h:52-73)
 * lum/depth/bgr[f*levels + lvl]; float luminance in [0,1] (pyramid.cpp:66-74), float depth in metres (0 = invalid),
 * optional 8-bit BGR (only needed by i3d_recompute_colors).  Images are copied to the device. */
int i3d_set_frames(i3d_context* ctx, int32_t num_frames, int32_t levels, const int32_t* widths, const int32_t* heights,
                   const float* const* lum, const float* const* depth, const uint8_t* const* bgr);

/* ---- camera: intrinsics Vec4 (fx,fy,cx,cy at level 0), distortion Vec5 (k1,k2,k3,p1,p2), poses Vec6[K]
 * (angle-axis, translation; world->camera) — optimizer.h:109-114 */
int i3d_set_camera(i3d_context* ctx, const double* intrinsics4, const double* distortion5, const double* poses6k);
int i3d_get_camera(i3d_context* ctx, double* intrinsics4, double* distortion5, double* poses6k);

/* ---- Optimizer::Data::voxel_sh_coeffs (optimizer.h:96): 9 doubles per voxel, visit order ------------------ */
int i3d_set_voxel_sh(i3d_context* ctx, const double* voxel_sh);
int i3d_get_voxel_sh(i3d_context* ctx, double* voxel_sh);

/* Intrinsic3D::init's Pyramid(num_rgbd_levels, color, depth) per keyframe, built on the device from level-0 colour + depth (already in colour
 * geometry): float luminance (convertTo 1/255 + BGR2GRAY), cv::pyrDown levels, valid-mean depth levels (rgbd/pyramid.cpp:59-166).
 * Replaces i3d_set_frames for callers that do not want to build the pyramids with OpenCV. */
int i3d_set_frames_rgbd(i3d_context* ctx, int32_t num_frames, int32_t levels, int32_t width, int32_t height, const uint8_t* const* bgr, const float* const* depth);
int i3d_get_frame_image(i3d_context* ctx, int32_t frame, int32_t level, float* lum /* may be NULL */, float* depth /* may be NULL */);
/* resizeDepth (rgbd/processing.cpp:129-181): a depth image resampled into the colour camera's geometry; intrinsics = {fx, fy, cx, cy} */
int i3d_resize_depth(int32_t device_ordinal, int32_t in_w, int32_t in_h, const float* depth_in, const float* in_intr4, int32_t out_w, int32_t out_h,
                     const float* out_intr4, float* depth_out);

/* ---- Optimizer::Config (optimizer.h:67-84) + the fields of Intrinsic3D::Config / Optimizer::Data the path reads */
typedef struct {
    int32_t iterations;            /* outer Gauss-Newton iterations (optimizer.cpp:119) */
    int32_t lm_steps;              /* max LM attempts per iteration (nls_solver.cpp:300) */
    double  lambda_g, lambda_r0, lambda_r1, lambda_s0, lambda_s1, lambda_a;
    int32_t fix_poses, fix_intrinsics, fix_distortion;
    float   occlusion_distance;    /* SDFColorization::Config::max_occlusion_distance (intrinsic3d.cpp:165) */
    int32_t num_observations;      /* ...::max_num_observations (intrinsic3d.cpp:166) */
    double  thres_shell;           /* Optimizer::Data::thres_shell */
    int32_t grid_level, rgbd_level;
    /* parity / measurement controls (not in the reference) */
    int32_t pcg_fixed_iterations;  /* >=0: run exactly this many PCG iterations per LM attempt; -1: Ceres' Q-test */
    int32_t verbose;
    int32_t carry_trust_radius;    /* extension, default 0 = the reference's ACTUAL behaviour.  1: start every outer iteration at the trust-region radius the
                                      previous one ended with — what nls_solver.cpp:322-323 is written to do but never does (a fresh NLSSolver is
                                      constructed per iteration, optimizer.cpp:138, so solver_info_ is always empty).  Saves the ~5 rejected LM attempts
                                      that re-discover the radius every iteration; results then differ from the reference's. */
    int32_t fix_sdf;               /* extension: every sdf_refined block constant (BASELINE.json configs[0], "albedo-only"); the reference has
                                      no such switch — Optimizer::fixVoxelParams (optimizer.cpp:312-361) fixes per voxel only */
} i3d_optimizer_config;

void i3d_optimizer_config_default(i3d_optimizer_config* cfg);   /* the reference's struct defaults */

/* per outer iteration: the quantities NLSSolver prints (nls_solver.cpp:57-103) */
typedef struct {
    int64_t rows[4];               /* Eg, Er, Es, Ea residual blocks */
    double  weight_sum[4];         /* per-type sum of row weights before normalisation */
    double  type_weight[4];        /* lambda_t / weight_sum_t * 1000 (nls_solver.cpp:379-394) */
    int64_t valid_voxels;          /* "voxels (valid n)" of optimizer.cpp:158 */
    int64_t free_parameters;
    double  cost_initial, cost_final;
    int32_t lm_iterations, successful_steps, termination;   /* termination: 0 no-conv, 1 convergence, 2 first successful step, 3 failure */
    int32_t pcg_iterations[50];    /* one per LM attempt */
    int32_t step_accepted[50];
    int32_t num_attempts;
    double  final_radius;
    double  time_add, time_build, time_solve;               /* seconds, the reference's split (nls_solver.cpp:66-67,101) */
} i3d_iteration_stats;

/* Optimizer::optimize on the resident grid / frames / camera / per-voxel SH.  stats: [cfg->iterations] or NULL. */
int i3d_optimize(i3d_context* ctx, const i3d_optimizer_config* cfg, i3d_iteration_stats* stats);

/* One-shot drop-in with host buffers (upload, optimize, write back sdf_refined/albedo/camera in place). */
int i3d_optimize_host(int32_t device_ordinal, const i3d_optimizer_config* cfg, const i3d_grid_view* grid,
                      double* sdf_refined_io, double* albedo_io,
                      int32_t num_frames, int32_t levels, const int32_t* widths, const int32_t* heights,
                      const float* const* lum, const float* const* depth,
                      double* intrinsics4_io, double* distortion5_io, double* poses6k_io,
                      const double* voxel_sh, i3d_iteration_stats* stats);

/* ---- LightingSVSH(grid, subvolume_size, lambda_reg, thres_shell, weighted=true)::estimate() followed by
 * computeVoxelShCoeffs() (intrinsic3d.cpp:255-264).  sh: [cap][9], sub_index: [cap][3] or NULL.  The per-voxel
 * coefficients stay resident for i3d_optimize (fetch with i3d_get_voxel_sh). */
typedef struct { int64_t data_rows, reg_rows; int32_t subvolumes, lm_iterations, termination; double cost_initial, cost_final; } i3d_sh_stats;
int i3d_estimate_sh(i3d_context* ctx, float subvolume_size, double lambda_reg, double thres_shell,
                    int32_t* num_subvolumes, double* sh, int32_t* sub_index, int32_t cap, i3d_sh_stats* stats);

/* ---- level transitions and the refine schedule (Intrinsic3D::refine, intrinsic3d.cpp:206-409) ------------------------------
 * i3d_set_grid_from_tsdf_records: SparseVoxelGrid<Voxel>::load (records in file order) + SDFAlgorithms::convert (algorithms.cpp:47-72)
 * i3d_recompute_colors          : Intrinsic3D::recomputeColors (SDFColorization::add/compute, colorization.cpp:113-189,318-354)
 * i3d_clear_outside_thin_shell  : SDFAlgorithms::clearVoxelsOutsideThinShell (algorithms.cpp:368-458)
 * i3d_upsample                  : SDFAlgorithms::upsample (algorithms.cpp:202-235)
 * The grid stays resident; i3d_grid_info / i3d_export_grid return it in visit order (e.g. inside the refine callback). */
int i3d_set_grid_from_tsdf_records(i3d_context* ctx, float voxel_size, int64_t n, const int32_t* keys, const float* sdf, const float* weight, const uint8_t* color);
int i3d_recompute_colors(i3d_context* ctx, float occlusion_distance, int32_t num_observations);
int i3d_clear_outside_thin_shell(i3d_context* ctx, double thres_shell, int64_t* new_count);
int i3d_upsample(i3d_context* ctx, int64_t* new_count);
int i3d_grid_info(i3d_context* ctx, int64_t* num_voxels, float* voxel_size, float* truncation);
int i3d_export_grid(i3d_context* ctx, int32_t* keys, double* sdf, double* sdf_refined, double* albedo, float* weight, uint8_t* color);

typedef struct {                   /* Intrinsic3D::Config (intrinsic3d.h:67-84), keys of data/intrinsic3d.yml */
    int32_t num_grid_levels, num_rgbd_levels;
    double  thin_shell_factor, thin_shell_factor_final;
    int32_t clear_distant_voxels;
    float   occlusion_distance; int32_t num_observations;
    float   subvolume_size_sh; double sh_lambda_reg;
} i3d_refine_config;
/* RefinementCallback::onSDFRefined(RefinementInfo) (intrinsic3d.h:94-114) */
typedef void (*i3d_refine_callback)(void* user, int32_t grid_level, int32_t num_grid_levels, int32_t pyramid_level, int32_t num_pyramid_levels);
int i3d_refine(i3d_context* ctx, const i3d_refine_config* rcfg, const i3d_optimizer_config* ocfg, i3d_refine_callback cb, void* user);

/* ---- on-disk formats either side of the path (host-only; no device needed) -------------------------------------------------
 * .tsdf: header {f32 voxel_size, f32 truncation, f32 integration_weight_sample, u64 count, f32 max_load_factor} then count records
 *        {i32 x,y,z; f32 sdf; f32 weight; u8 r,g,b; u8 pad}  (SparseVoxelGrid<Voxel>::save/load, sparse_voxel_grid.cpp:484-569).
 * VoxelSBR dump: same header, records {i32 x,y,z; f64 sdf; f32 weight; u8 r,g,b,pad; f64 albedo; f64 sdf_refined} (44 bytes).
 * poses: TUM trajectory lines (Sensor::savePoses, rgbd/sensor.cpp:315-347); intrinsics: Camera::save/load (camera.cpp:202-274).
 * i3d_config_load_yaml reads the flat `key: "value"` map of data/intrinsic3d.yml into the two config structs. */
int i3d_tsdf_read_header(const char* path, float* voxel_size, float* truncation, float* integration_weight_sample, uint64_t* count, float* max_load_factor);
int i3d_tsdf_read_records(const char* path, uint64_t capacity, int32_t* keys, float* sdf, float* weight, uint8_t* color);
int i3d_tsdf_write(const char* path, float voxel_size, float truncation, float integration_weight_sample, float max_load_factor, uint64_t count,
                   const int32_t* keys, const float* sdf, const float* weight, const uint8_t* color);
int i3d_sbr_write(const char* path, float voxel_size, float truncation, float integration_weight_sample, float max_load_factor, uint64_t count,
                  const int32_t* keys, const double* sdf, const double* sdf_refined, const double* albedo, const float* weight, const uint8_t* color);
int i3d_sbr_read(const char* path, uint64_t capacity, int32_t* keys, double* sdf, double* sdf_refined, double* albedo, float* weight, uint8_t* color);
int i3d_write_poses(const char* path, int32_t num_frames, const double* timestamps, const double* poses_world_to_cam /* [K][6] */);
int i3d_write_intrinsics(const char* path, int32_t width, int32_t height, const double* intr4, const double* dist5);
int i3d_read_intrinsics(const char* path, int32_t* width, int32_t* height, double* intr4, double* dist5);
int i3d_config_load_yaml(const char* path, i3d_refine_config* rcfg, i3d_optimizer_config* ocfg);
int i3d_yaml_get(const char* path, const char* key, char* value, uint64_t capacity);     /* Settings::get<std::string>: any key of a flat yml */

/* ---- mesh export of the resident grid (MarchingCubes<VoxelSBR>::extractSurface, MeshUtil, Mesh::save; SDFVisualization::exportMesh) -----
 * use_refined_sdf: SDFAlgorithms::applyRefinedSdf before extraction (app_intrinsic3d.cpp:170-172).  color_mode: I3D_COLOR_* below — what
 * SDFVisualization::colorize paints on the voxels before the mesh of a mode is extracted (visualization.cpp:101-164, 228-373); the shading modes need the
 * lighting estimate of i3d_estimate_sh / i3d_refine.  largest_component_only: MeshUtil::removeLooseComponents.  PLY: binary_little_endian, float xyz, uchar rgb,
 * "uchar int" face lists (mesh.cpp:41-100). */
enum {
    I3D_COLOR_VOXEL = 0,                 /* ""                 the voxel colours */
    I3D_COLOR_ALBEDO = 1,                /* "albedo"           output_mesh_albedo            applyColorAlbedo            :308-315 */
    I3D_COLOR_NORMALS = 2,               /* "normals"          output_mesh_normals           applyColorNormals           :228-240 */
    I3D_COLOR_LAPLACIAN = 3,             /* "lap"              output_mesh_laplacian         applyColorLaplacian         :243-259 */
    I3D_COLOR_INTENSITY = 4,             /* "lum"              output_mesh_intensity         applyColorIntensity         :262-270 */
    I3D_COLOR_INTENSITY_GRAD = 5,        /* "lum_grad"         output_mesh_intensity_grad    applyColorIntensityGradient :273-305 */
    I3D_COLOR_SHADING = 6,               /* "shading_sv"       output_mesh_shading_sv        applyColorShading(false)    :318-359 */
    I3D_COLOR_SHADING_CONST_ALBEDO = 7,  /* "shading_sv_const" output_mesh_shading_sv_const  applyColorShading(true) */
    I3D_COLOR_CHROMACITY = 8             /* "chroma"           output_mesh_chromacity        applyColorChromacity        :362-373 */
    /* "subvol" / "subvol_interp" paint Subvolumes::color(), drawn from rand() in the reference (subvolumes.cpp:87-91): nothing to reproduce, not offered */
};
int i3d_extract_mesh(i3d_context* ctx, int32_t use_refined_sdf, int32_t color_mode, int32_t largest_component_only, int64_t* num_vertices, int64_t* num_faces);
int i3d_get_mesh(i3d_context* ctx, float* vertices /*[nv][3]*/, uint8_t* colors /*[nv][3]*/, int32_t* faces /*[nf][3]*/);
int i3d_export_mesh_ply(i3d_context* ctx, const char* path, int32_t use_refined_sdf, int32_t color_mode, int32_t largest_component_only);
int i3d_write_ply(const char* path, int64_t num_vertices, const float* vertices, const uint8_t* colors /* may be NULL */, int64_t num_faces, const int32_t* faces);
/* MeshUtil::removeLooseComponents + removeUnusedVertices (mesh/util.cpp:47-171) on caller arrays, in place (what largest_component_only applies): keeps the
 * largest connected component (first one among equals, components numbered by their first face), drops the vertices no face uses; counts updated.  Host only. */
int i3d_mesh_remove_loose_components(int64_t* num_vertices, float* vertices, uint8_t* colors /* may be NULL */, int64_t* num_faces, int32_t* faces);
/* SDFVisualization::applyColor* on caller arrays (voxels in any order; subvolumes as i3d_estimate_sh returns them, only read by the shading modes): the colour
 * every voxel gets in a colour mode.  The same function the export kernel runs, instantiated for the host.  visit_rank: the position of every voxel in the
 * reference's walk over its grid (NULL: the array order) — "lum_grad" is painted in place there, a voxel reads its +x neighbour repainted if the walk passed
 * it earlier (visualization.cpp:273-305), and the export reproduces that from the resident grid's visit order.  Host only. */
int i3d_visualization_colors(int32_t color_mode, float voxel_size, int64_t num_voxels, const int32_t* keys, const double* sdf_refined, const double* albedo, const float* weight,
                             const uint8_t* color, const int64_t* visit_rank /* or NULL */, float subvolume_size, int32_t num_subvolumes, const int32_t* subvolume_index /* [S][3] or NULL */,
                             const double* subvolume_sh /* [S][9] or NULL */, uint8_t* color_out /* [n][3] */);
int i3d_mc_tables(uint8_t* ntri /*[256]*/, int8_t* tri /*[256][16]*/);      /* the triangulation table (Bourke's, as in marching_cubes.cpp:330-623); returns max triangles per cell */

/* ---- dataset loader in front of the path.  Host code except i3d_init_frames_from_sensor.
 * PNG: the layout cv::imdecode(IMREAD_UNCHANGED) returns — interleaved, B,G,R[,A] order, 8-bit or native-endian 16-bit, palette and
 * 1/2/4-bit images expanded (rgbd/sensor_i3d.cpp:307-327). */
int i3d_png_info(const uint8_t* data, uint64_t size, int32_t* width, int32_t* height, int32_t* channels, int32_t* bit_depth);
int i3d_png_decode(const uint8_t* data, uint64_t size, void* pixels, uint64_t capacity_bytes);
/* Intrinsic3D::init's pose conversion (intrinsic3d.cpp:189-192): camera-to-world Mat4f (row-major) -> world-to-camera Vec6 (angle-axis, t) */
int i3d_pose_mat_to_vec6(const float* cam_to_world16, double* pose6);
/* Sensor::create + SensorI3d::init (sensor.cpp:63-96, sensor_i3d.cpp:60-144): `frame-%06d.{color,depth}.png`, `.pose.txt`,
 * `{depth,color}Intrinsics.txt` of `folder`; max_frames / min_depth / max_depth as in sensor.yml (0 = off) */
typedef struct i3d_sensor i3d_sensor;
int  i3d_sensor_open(const char* folder, int32_t max_frames, float min_depth, float max_depth, i3d_sensor** out);
/* Sensor::create(Settings&) (rgbd/sensor.cpp:64-118) over a sensor.yml: keys dataset, max_frames, min_depth, max_depth, converted like Settings::get<T> (settings.cpp:86-109);
 * the depth range read from the file is handed back (AppFusion sizes its volume with it, app_fusion.cpp:131-133). */
int  i3d_sensor_open_yaml(const char* sensor_yml, i3d_sensor** out, float* min_depth /* may be NULL */, float* max_depth /* may be NULL */);
void i3d_sensor_close(i3d_sensor* s);
int  i3d_sensor_info(const i3d_sensor* s, int32_t* num_frames, int32_t* num_loaded, int32_t* color_wh /*[2]*/, int32_t* depth_wh /*[2]*/,
                     float* color_intr4 /* fx fy cx cy */, float* depth_intr4);
int  i3d_sensor_color(const i3d_sensor* s, int32_t id, uint8_t* bgr /*[h][w][3]*/);          /* Sensor::color */
int  i3d_sensor_depth(const i3d_sensor* s, int32_t id, float* depth /*[h][w] metres*/);      /* Sensor::depth: decode, 1/1000, thresholdDepth */
int  i3d_sensor_pose(const i3d_sensor* s, int32_t id, float* cam_to_world16);
int  i3d_sensor_set_pose(i3d_sensor* s, int32_t id, const float* cam_to_world16);
int  i3d_sensor_set_pose_vec6(i3d_sensor* s, int32_t id, const double* pose_world_to_cam6);  /* finishRgbdLevel write-back, intrinsic3d.cpp:362-368 */
int  i3d_sensor_save_poses(const i3d_sensor* s, const char* path);                           /* Sensor::savePoses, sensor.cpp:315-347 */
/* KeyframeSelection::load / save / selectKeyframes (keyframe_selection.cpp:73-106,139-207); count = lines in the file even if > capacity */
int i3d_keyframes_load(const char* path, int32_t* window_size, uint64_t capacity, double* scores, uint8_t* is_keyframe, uint64_t* count);
int i3d_keyframes_save(const char* path, int32_t window_size, uint64_t count, const double* scores, const uint8_t* is_keyframe);
int i3d_keyframes_select(int32_t window_size, uint64_t count, const double* scores, uint8_t* is_keyframe);
/* KeyframeSelection::estimateBlur (keyframe_selection.cpp:219-311): blur metric of one colour (B,G,R) or grey 8-bit image; 1 = sharp */
int i3d_blur_score(const uint8_t* image, int32_t width, int32_t height, int32_t channels, double* score);
/* Intrinsic3D::init's keyframe loop (intrinsic3d.cpp:156-193): for every keyframe decode colour + depth, resample the depth into the colour
 * geometry and build the pyramids on the device, convert the pose; sets the frames and the camera (colour intrinsics, zero distortion) of ctx */
int i3d_init_frames_from_sensor(i3d_context* ctx, int32_t device_ordinal, const i3d_sensor* s, uint64_t num_flags, const uint8_t* is_keyframe,
                                int32_t num_rgbd_levels, int32_t frame_capacity, int32_t* frame_ids, int32_t* num_keyframes);

/* ---- TSDF fusion, the stage in front of the path: AppFusion::fuseSDF's volume on the device.
 * i3d_fusion_create    SparseVoxelGrid<Voxel>::create(voxel_size, depth_min, depth_max) + setClipBounds (app_fusion.cpp:121-139); clip6 = {x0,x1,y0,y1,z0,z1},
 *                      all zero / NULL = no clipping; initial_capacity = expected number of allocated voxels (the table grows when needed)
 * i3d_fusion_integrate erodeDiscontinuities(depth, erode_window) + computeNormals + SparseVoxelGrid::integrate (alloc + update) of one frame
 *                      (app_fusion.cpp:152-166, sparse_voxel_grid.cpp:301-467); depth = Sensor::depth (metres, 0 = invalid), pose = camera-to-world
 * i3d_fusion_finish    SDFAlgorithms::correctSDF(grid, correct_iterations) + clearInvalidVoxels (sdf/algorithms.cpp:260-366); count = saved voxels
 * i3d_fusion_get/save  the records in the order SparseVoxelGrid::save writes them (the reference's unordered_map iteration order) */
typedef struct i3d_fusion i3d_fusion;
int  i3d_fusion_create(int32_t device_ordinal, float voxel_size, float depth_min, float depth_max, const float* clip6, uint64_t initial_capacity, i3d_fusion** out);
void i3d_fusion_destroy(i3d_fusion* f);
const char* i3d_fusion_last_error(const i3d_fusion* f);
int  i3d_fusion_integrate(i3d_fusion* f, int32_t depth_w, int32_t depth_h, const float* depth_intr4, int32_t color_w, int32_t color_h, const float* color_intr4,
                          const float* depth, const uint8_t* bgr, const float* pose_cam_to_world16, int32_t erode_window);
int  i3d_fusion_finish(i3d_fusion* f, int32_t correct_iterations, uint64_t* count);
int  i3d_fusion_info(const i3d_fusion* f, uint64_t* frames, uint64_t* allocated, uint64_t* capacity, int32_t* correct_launches);
int  i3d_fusion_get(const i3d_fusion* f, int32_t* keys, float* sdf, float* weight, uint8_t* color);
int  i3d_fusion_save(const i3d_fusion* f, const char* path);

/* ---- one process per GPU: the voxel state is replicated; row work / row storage / solver vectors are sharded by contiguous, tile-aligned
 * ranges of the brick-ordered work list (compact regions of the surface).  A rank builds rows for its range + a thin rim of ghost entries;
 * per PCG pass it pushes the operator input of the rim to its neighbours and joins ONE small all-reduce [camera block | p.q] plus the 4 iteration
 * scalars — over peer-to-peer xGMI mailboxes (self-tested at start-up), RCCL as the fallback and for the rare large collectives.  Call after i3d_create on every rank with the same unique id (i3d_comm_unique_id on rank 0,
 * broadcast by the launcher, e.g. torch.distributed). */
int i3d_comm_unique_id(void* out128, int32_t* bytes);
int i3d_comm_init(i3d_context* ctx, int32_t rank, int32_t world, const void* unique_id, int32_t id_bytes);
/* single-GPU simulation of W ranks (W host threads, one context each, same device) — test vehicle for the SPMD control flow */
void* i3d_comm_sim_create(int32_t world);
void  i3d_comm_sim_destroy(void* shared);
int   i3d_comm_init_sim(i3d_context* ctx, void* shared, int32_t rank);
/* host-side view of the sharding plan (no device needed): owned range and vector layout of `rank`, and which work-list entries
 * it must compute rows for.  anbr: [18][A] neighbour table in work-list space (-1 = none), active: [A]. */
int   i3d_shard_plan(int32_t A, int32_t world, int32_t rank, const int32_t* anbr, const uint8_t* active,
                     int32_t* chunk, int32_t* own0, int32_t* own1, uint8_t* in_compute_list /*[A]*/);
int32_t i3d_shard_vec_index(int32_t a, int32_t chunk, int32_t albedo);
/* need[e] bit k: rank k's rows read the unknowns of work-list entry e, which it does not own (what Comm::push_halo moves once per PCG pass) */
int   i3d_shard_need(int32_t A, int32_t world, const int32_t* anbr /*[18][A]*/, const uint8_t* active /*[A]*/, uint64_t* need /*[A]*/);
/* traffic log of the sharded path since i3d_comm_init: halo exchanges (calls, bytes this rank sent), all-reduces (calls, bytes), and the
 * plan of the last outer iteration (rim entries sent / received per pass, foreign tiles with ghost entries, compute-list length) */
int   i3d_comm_stats(i3d_context* ctx, int64_t* halo_calls, int64_t* halo_bytes_sent, int64_t* reduce_calls, int64_t* reduce_bytes,
                     int32_t* halo_entries_send, int32_t* halo_entries_recv, int32_t* ghost_tiles, int32_t* compute_list);
/* what carries the per-pass exchanges: "p2p-mailbox" (peer-to-peer stores over xGMI), "rccl" (fallback), "sim-*" (1-GPU rank simulation), "" without a communicator */
const char* i3d_comm_transport(i3d_context* ctx);

/* ---- measurement: HIP-event time (ms) and launch count accumulated per kernel family on the context's stream
 * since the last reset.  names: see i3d_kernel_name(). */
enum { I3D_K_CLASSIFY = 0, I3D_K_OBSERVE, I3D_K_BUILD, I3D_K_EG_PASS /* J^T W J p passes of the PCG */, I3D_K_GATHER, I3D_K_COST, I3D_K_VECTOR, I3D_K_SH,
       I3D_K_EG_AUX /* gradient and column-norm passes over the rows */,
       I3D_K_COMM /* sharded runs: halo push, all-reduce, all-gather launches (GPU time incl. waiting for the peers) */,
       I3D_K_EG_MR2, I3D_K_EG_MR3 /* operator passes of a ladder batch that serve 2 / 3 systems with one stream of the rows (I3D_K_EG_PASS: one system) */, I3D_K_COUNT };
int i3d_timing_enable(i3d_context* ctx, int32_t on);
/* restrict the per-launch HIP events to the categories of the mask (bit = 1 << I3D_K_*; default: all).  An event pair around EVERY launch of a
 * Gauss-Newton iteration (~900 launches) costs ~8 % of its wall clock; bench.py times only what its roofline needs. */
int i3d_timing_select(i3d_context* ctx, uint32_t category_mask);
int i3d_timing_get(i3d_context* ctx, double* ms /*[I3D_K_COUNT]*/, int64_t* launches /*[I3D_K_COUNT]*/, int32_t reset);
/* the same restricted to launches that did work: PCG launches queued behind the device-side convergence flag return at once (~4 us) and
 * would flatter an average; a launch counts when it lasted >= 25 % of the longest launch of its category */
int i3d_timing_get_work(i3d_context* ctx, double* ms /*[I3D_K_COUNT]*/, int64_t* launches /*[I3D_K_COUNT]*/);
/* the same, plus what the upper cut-off removed: launches that lasted more than 4x the 90th percentile of their category (a launch that straddles a
 * hiccup of the device) — their number and total time, so that a caller can quote them beside the average.  The exchange category is never cut. */
int i3d_timing_get_work_ex(i3d_context* ctx, double* ms, int64_t* launches, double* slow_ms /*[I3D_K_COUNT]*/, int64_t* slow_launches /*[I3D_K_COUNT]*/);
const char* i3d_kernel_name(int32_t k);
/* sizes of the last assembled problem: active voxels, Eg/Er/Es/Ea rows, free parameters */
int i3d_problem_sizes(i3d_context* ctx, int64_t out[6]);

/* ---- parity probes (tests only): assemble the rows of outer iteration `iteration` without solving and export them.
 * Arrays are indexed by visit order; slot k in [0, slots).  Any pointer may be NULL. */
int i3d_debug_assemble(i3d_context* ctx, const i3d_optimizer_config* cfg, int32_t iteration, int32_t* slots_out);
/* iteration order of the reference's unordered_map<Vec3i,...> after `map[key_i] = i`, i = 0..n-1: mode 0 = the host replay (repeated keys
 * allowed), 1 = the replay for distinct keys, 2 = a real std::unordered_map, 3 = the per-rehash-epoch closed form on the host, 4 = the same on
 * the current device (what the level transitions and the fusion volume use; distinct keys); returns the number of elements, -1 on error */
int64_t i3d_debug_map_order(const int32_t* keys, int64_t n, int32_t mode, int32_t* order);
int i3d_debug_flags(i3d_context* ctx, uint8_t* flags /*[N]: bit0 valid,1 active,2 ring_ok,3 free_sdf,4 free_albedo*/);
int i3d_debug_eg_rows(i3d_context* ctx, int32_t* frame /*[N][slots], -1 = none*/, float* weight /*[N][slots] normalised*/,
                      float* residual /*[N][slots]*/, float* jac /*[N][slots][29]*/);
int i3d_debug_reg_rows(i3d_context* ctx, uint8_t* has_er /*[N]*/, uint8_t* has_es /*[N]*/, float* ea_weight /*[N][6] normalised, 0 = none*/);
int i3d_debug_neighbors(i3d_context* ctx, int32_t* nbr /*[N][18] visit indices, -1 = missing*/);
/* gradient S^-1-free: g = J^T W r, diag(J^T W J) and y = J^T W J x over parameter ids [sdf N | albedo N | poses 6K | intr 4 | dist 5], visit order */
int i3d_debug_normal_eq(i3d_context* ctx, double* gradient, double* jtj_diag, double* cost);
int i3d_debug_jtj_apply(i3d_context* ctx, const double* x, double* y);
/* counters of the context since its creation: stream synchronisations of the solver path (assemble + the LM loop).  The trust-region loop of
 * NLSSolver::solve (nls_solver.cpp:296-337) runs on the device; a Gauss-Newton iteration costs a handful of them, not two per LM attempt. */
int i3d_debug_counters(i3d_context* ctx, int64_t* stream_syncs);
/* the damping ladder of the trust-region loop (consecutive LM attempts of NLSSolver::solve, nls_solver.cpp:296-337, whose radii are known in advance are solved
 * together, I3D_LADDER): since the context was created — [0] batches, [1] streams of the stored rows (operator launches of ladder solves), [2] system passes (what
 * the serial loop would have streamed), [3] batches that went out of step (invalid step) and were re-solved, [4] systems solved but never decided (an earlier
 * attempt of their batch was accepted), [5] the batch depth in force (1 = serial loop). */
int i3d_debug_ladder_stats(i3d_context* ctx, int64_t* out6);
/* the conservative culling in front of the observation pass (SDFColorization::computeObservation is evaluated per (voxel, keyframe), colorization.cpp:215-315;
 * the device skips (group of 64 voxels, keyframe) pairs no voxel of which can be observed): pairs of the last assemble and how many were skipped.  culled = -1 when
 * culling is off (I3D_NO_CULL=1). */
int i3d_debug_cull_stats(i3d_context* ctx, int64_t* pairs, int64_t* culled);

#ifdef __cplusplus
}
#endif
#endif
