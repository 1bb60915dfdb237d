// AppIntrinsic3D on the MI355X library: the caller of the drop-in boundary (apps/src/app_intrinsic3d.cpp:72-210 of the reference),
// written against include/intrinsic3d_hip.h only.
//
//   app_intrinsic3d -s <path>/sensor.yml -i <path>/intrinsic3d.yml [--device N]
//
// sensor.yml / intrinsic3d.yml are the reference's files (data/*.yml).  As in the reference the working directory becomes the directory of
// sensor.yml, ./intrinsic3d is created, and after every (grid level, rgbd level) the meshes, poses and intrinsics are written with the
// `_g{L}_p{P}` postfix.  Mesh colour modes: every `output_mesh_*` switch of intrinsic3d.yml except the two subvolume views (random colours in the reference).
#include "../include/intrinsic3d_hip.h"
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <sys/stat.h>
#include <unistd.h>

namespace {

std::string yaml(const std::string& file, const char* key, const char* fallback = "") {
    char buf[4096];
    return i3d_yaml_get(file.c_str(), key, buf, sizeof(buf)) == I3D_OK ? std::string(buf) : std::string(fallback);
}
std::string absolute(const std::string& p) { char buf[PATH_MAX]; return realpath(p.c_str(), buf) ? std::string(buf) : p; }

struct App {
    i3d_context* ctx = nullptr;
    i3d_sensor* sensor = nullptr;
    std::string cfg_file;
    std::vector<int32_t> frame_ids;
    int color_w = 0, color_h = 0;
};

// AppIntrinsic3D::onSDFRefined (app_intrinsic3d.cpp:159-210) + the write-back of Intrinsic3D::finishRgbdLevel (intrinsic3d.cpp:362-372)
void on_refined(void* user, int32_t grid_level, int32_t, int32_t pyramid_level, int32_t) {
    App& a = *static_cast<App*>(user);
    const std::string post = "_g" + std::to_string(grid_level) + "_p" + std::to_string(pyramid_level);
    const std::string mesh_prefix = yaml(a.cfg_file, "output_mesh_prefix");
    if (!mesh_prefix.empty()) {
        const int largest = std::atoi(yaml(a.cfg_file, "output_mesh_largest_comp_only", "0").c_str());
        // SDFVisualization::getOutputModes(settings, true) (sdf/visualization.cpp:71-89): the voxel colours, then every enabled mode in this order
        static const struct { const char* key; const char* name; int mode; } MODES[] = {
            {nullptr, "", I3D_COLOR_VOXEL}, {"output_mesh_normals", "normals", I3D_COLOR_NORMALS}, {"output_mesh_laplacian", "lap", I3D_COLOR_LAPLACIAN},
            {"output_mesh_intensity", "lum", I3D_COLOR_INTENSITY}, {"output_mesh_intensity_grad", "lum_grad", I3D_COLOR_INTENSITY_GRAD}, {"output_mesh_albedo", "albedo", I3D_COLOR_ALBEDO},
            {"output_mesh_shading_sv", "shading_sv", I3D_COLOR_SHADING}, {"output_mesh_shading_sv_const", "shading_sv_const", I3D_COLOR_SHADING_CONST_ALBEDO},
            {"output_mesh_chromacity", "chroma", I3D_COLOR_CHROMACITY}};
        for (const auto& m : MODES) {
            if (m.key && !std::atoi(yaml(a.cfg_file, m.key, "0").c_str())) continue;
            std::printf("SDF visualization and export: %s\n", m.name);
            const std::string file = mesh_prefix + post + (m.name[0] ? "_" + std::string(m.name) : std::string()) + ".ply";
            if (i3d_export_mesh_ply(a.ctx, file.c_str(), 1, m.mode, largest) != I3D_OK) std::fprintf(stderr, "Could not save mesh: %s\n", i3d_last_error(a.ctx));
        }
        for (const char* key : {"output_mesh_subvolumes", "output_mesh_subvolumes_interpolated"})           // painted with rand() colours in the reference: nothing to reproduce
            if (std::atoi(yaml(a.cfg_file, key, "0").c_str())) std::fprintf(stderr, "%s: this view shows random subvolume colours in the reference and is not produced here\n", key);
    }
    double intr[4], dist[5]; std::vector<double> poses(6 * a.frame_ids.size());
    if (i3d_get_camera(a.ctx, intr, dist, poses.data()) != I3D_OK) { std::fprintf(stderr, "Could not read the camera: %s\n", i3d_last_error(a.ctx)); return; }
    for (size_t k = 0; k < a.frame_ids.size(); ++k) i3d_sensor_set_pose_vec6(a.sensor, a.frame_ids[k], &poses[6 * k]);
    const std::string poses_prefix = yaml(a.cfg_file, "output_poses_prefix");
    if (!poses_prefix.empty()) {
        const std::string file = poses_prefix + post + ".txt";
        std::printf("Saving camera poses to file %s\n", file.c_str());
        if (i3d_sensor_save_poses(a.sensor, file.c_str()) != I3D_OK) std::fprintf(stderr, "Could not save poses...\n");
    }
    const std::string intr_prefix = yaml(a.cfg_file, "output_intrinsics_prefix");
    if (!intr_prefix.empty()) {
        const std::string file = intr_prefix + post + ".txt";
        std::printf("Saving camera intrinsics to file %s\n", file.c_str());
        if (i3d_write_intrinsics(file.c_str(), a.color_w, a.color_h, intr, dist) != I3D_OK) std::fprintf(stderr, "Could not save color camera intrinsics!\n");
    }
    std::fflush(stdout);
}

}  // namespace

int main(int argc, char* argv[]) {
    std::string sensor_cfg, i3d_cfg; int device = 0;
    for (int i = 1; i < argc; ++i) {                            // cv::CommandLineParser accepts -s=<v> / --sensor=<v>; `-s <v>` is accepted as well
        std::string arg = argv[i], val; const size_t eq = arg.find('=');
        if (eq != std::string::npos) { val = arg.substr(eq + 1); arg = arg.substr(0, eq); } else if (i + 1 < argc) val = argv[++i];
        if (arg == "-s" || arg == "--sensor") sensor_cfg = val;
        else if (arg == "-i" || arg == "--intrinsic3d") i3d_cfg = val;
        else if (arg == "--device") device = std::atoi(val.c_str());
        else { std::fprintf(stderr, "usage: %s -s <sensor.yml> -i <intrinsic3d.yml> [--device N]\n", argv[0]); return 2; }
    }
    if (sensor_cfg.empty() || i3d_cfg.empty()) { std::fprintf(stderr, "usage: %s -s <sensor.yml> -i <intrinsic3d.yml> [--device N]\n", argv[0]); return 2; }
    sensor_cfg = absolute(sensor_cfg); i3d_cfg = absolute(i3d_cfg);
    const std::string dir = sensor_cfg.substr(0, sensor_cfg.find_last_of('/'));
    if (chdir(dir.c_str()) != 0) { std::fprintf(stderr, "cannot change the working directory to %s\n", dir.c_str()); return 1; }
    mkdir("./intrinsic3d", 0755);

    App app; app.cfg_file = i3d_cfg;
    int rc = i3d_sensor_open_yaml(sensor_cfg.c_str(), &app.sensor, nullptr, nullptr);                      // Sensor::create(sensor_cfg)
    int32_t num_frames = 0, num_loaded = 0, cwh[2] = {0, 0};
    if (rc == I3D_OK) i3d_sensor_info(app.sensor, &num_frames, &num_loaded, cwh, nullptr, nullptr, nullptr);
    if (rc != I3D_OK || num_loaded == 0) { std::fprintf(stderr, "RGB-D sensor could not be initialized!\n"); return 1; }
    app.color_w = cwh[0]; app.color_h = cwh[1];
    std::printf("%d filenames loaded.\n", num_frames);

    i3d_refine_config rcfg; i3d_optimizer_config ocfg;
    std::memset(&rcfg, 0, sizeof(rcfg)); i3d_optimizer_config_default(&ocfg);
    rcfg.num_grid_levels = 3; rcfg.num_rgbd_levels = 3; rcfg.thin_shell_factor = 2.0; rcfg.thin_shell_factor_final = 1.0; rcfg.clear_distant_voxels = 1;   // Intrinsic3D::Config
    rcfg.occlusion_distance = 0.02f; rcfg.num_observations = 5; rcfg.subvolume_size_sh = 0.2f; rcfg.sh_lambda_reg = 10.0;                                   // (intrinsic3d.h:67-83)
    if (i3d_config_load_yaml(i3d_cfg.c_str(), &rcfg, &ocfg) != I3D_OK) { std::fprintf(stderr, "Could not load %s\n", i3d_cfg.c_str()); return 1; }

    std::printf("Loading Keyframes...\n");
    int32_t window = 0; uint64_t nkf_lines = 0;
    std::vector<uint8_t> is_kf;
    const std::string kf_file = yaml(i3d_cfg, "keyframes");
    if (i3d_keyframes_load(kf_file.c_str(), &window, 0, nullptr, nullptr, &nkf_lines) == I3D_OK) {
        is_kf.resize(nkf_lines);
        i3d_keyframes_load(kf_file.c_str(), &window, nkf_lines, nullptr, is_kf.data(), &nkf_lines);
    } else std::fprintf(stderr, "Could not load keyframes ...\n");
    size_t nkf = 0; for (uint8_t k : is_kf) nkf += k != 0;
    std::printf("%zu keyframes loaded.\n", nkf);

    std::printf("Loading SDF volume...\n");
    const std::string sdf_file = yaml(i3d_cfg, "input_sdf");
    float voxel_size = 0, truncation = 0, iws = 0, mlf = 0; uint64_t count = 0;
    if (i3d_tsdf_read_header(sdf_file.c_str(), &voxel_size, &truncation, &iws, &count, &mlf) != I3D_OK) { std::fprintf(stderr, "Could not load voxel grid!\n"); return 1; }
    std::vector<int32_t> keys(3 * count); std::vector<float> sdf(count), weight(count); std::vector<uint8_t> color(3 * count);
    if (i3d_tsdf_read_records(sdf_file.c_str(), count, keys.data(), sdf.data(), weight.data(), color.data()) != I3D_OK) { std::fprintf(stderr, "Could not load voxel grid!\n"); return 1; }
    std::printf("   %llu voxels, voxel size %g\n", (unsigned long long)count, (double)voxel_size);

    if (i3d_create(device, &app.ctx) != I3D_OK) { std::fprintf(stderr, "no HIP device %d\n", device); return 1; }
    auto fail = [&](const char* what) { std::fprintf(stderr, "%s: %s\n", what, i3d_last_error(app.ctx)); i3d_destroy(app.ctx); i3d_sensor_close(app.sensor); return 1; };
    if (i3d_set_grid_from_tsdf_records(app.ctx, voxel_size, count, keys.data(), sdf.data(), weight.data(), color.data()) != I3D_OK) return fail("Intrinsic3D failed (grid)");
    std::printf("   convert and store input frames ...\n");
    app.frame_ids.resize(nkf ? nkf : 1); int32_t nk = 0;
    if (i3d_init_frames_from_sensor(app.ctx, device, app.sensor, is_kf.size(), is_kf.data(), rcfg.num_rgbd_levels, (int32_t)app.frame_ids.size(), app.frame_ids.data(), &nk) != I3D_OK)
        return fail("Intrinsic3D failed (frames)");
    app.frame_ids.resize(nk);
    if (i3d_refine(app.ctx, &rcfg, &ocfg, on_refined, &app) != I3D_OK) return fail("Intrinsic3D failed!");
    i3d_destroy(app.ctx); i3d_sensor_close(app.sensor);
    return 0;
}
