// AppFusion on the MI355X library (apps/src/app_fusion.cpp:64-200 of the reference), written against include/intrinsic3d_hip.h only.
//
//   app_fusion -s <path>/sensor.yml -f <path>/fusion.yml [--device N]
//
// As in the reference the working directory becomes the directory of sensor.yml and ./fusion is created; the frames (keyframes only when
// fusion.yml names a keyframes file) are fused into a TSDF volume on the device, corrected, cleaned, saved as `output_sdf`, and the
// marching-cubes mesh of the volume is saved as `output_mesh`.
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
float yamlf(const std::string& file, const char* key) { return (float)std::atof(yaml(file, key, "0").c_str()); }
std::string absolute(const std::string& p) { char buf[PATH_MAX]; return realpath(p.c_str(), buf) ? std::string(buf) : p; }
}  // namespace

int main(int argc, char* argv[]) {
    std::string sensor_cfg, fusion_cfg; int device = 0;
    for (int i = 1; i < argc; ++i) {
        std::string arg = argv[i], val; const size_t eq = arg.find('=');
        if (eq != std::string::npos) { val = arg.substr(eq + 1); arg = arg.substr(0, eq); } else if (i + 1 < argc) val = argv[++i];
        if (arg == "-s" || arg == "--sensor") sensor_cfg = val;
        else if (arg == "-f" || arg == "--fusion") fusion_cfg = val;
        else if (arg == "--device") device = std::atoi(val.c_str());
        else { std::fprintf(stderr, "usage: %s -s <sensor.yml> -f <fusion.yml> [--device N]\n", argv[0]); return 2; }
    }
    if (sensor_cfg.empty() || fusion_cfg.empty()) { std::fprintf(stderr, "usage: %s -s <sensor.yml> -f <fusion.yml> [--device N]\n", argv[0]); return 2; }
    sensor_cfg = absolute(sensor_cfg); fusion_cfg = absolute(fusion_cfg);
    const std::string dir = sensor_cfg.substr(0, sensor_cfg.find_last_of('/'));
    if (chdir(dir.c_str()) != 0) { std::fprintf(stderr, "cannot change the working directory to %s\n", dir.c_str()); return 1; }
    mkdir("./fusion", 0755);

    i3d_sensor* sensor = nullptr;
    float depth_min = 0.0f, depth_max = 0.0f;
    int rc = i3d_sensor_open_yaml(sensor_cfg.c_str(), &sensor, &depth_min, &depth_max);                    // Sensor::create(sensor_cfg)
    int32_t num_frames = 0, num_loaded = 0, cwh[2] = {0, 0}, dwh[2] = {0, 0}; float ci[4], di[4];
    if (rc == I3D_OK) i3d_sensor_info(sensor, &num_frames, &num_loaded, cwh, dwh, ci, di);
    if (rc != I3D_OK || num_loaded == 0) { std::fprintf(stderr, "RGB-D sensor could not be initialized!\n"); return 1; }
    std::printf("%d filenames loaded.\n", num_frames);

    // keyframes (optional): fuse only the selected frames (app_fusion.cpp:112-120,145-149)
    std::vector<uint8_t> is_kf; bool use_kf = false;
    const std::string kf_file = yaml(fusion_cfg, "keyframes");
    if (!kf_file.empty()) {
        use_kf = true; int32_t window = 0; uint64_t n = 0;
        if (i3d_keyframes_load(kf_file.c_str(), &window, 0, nullptr, nullptr, &n) == I3D_OK) { is_kf.resize(n); i3d_keyframes_load(kf_file.c_str(), &window, n, nullptr, is_kf.data(), &n); }
        else std::fprintf(stderr, "Could not load keyframes ...\n");
    }

    const float voxel_size = yamlf(fusion_cfg, "voxel_size");
    const float clip[6] = {yamlf(fusion_cfg, "clip_x0"), yamlf(fusion_cfg, "clip_x1"), yamlf(fusion_cfg, "clip_y0"), yamlf(fusion_cfg, "clip_y1"), yamlf(fusion_cfg, "clip_z0"),
                           yamlf(fusion_cfg, "clip_z1")};
    i3d_fusion* vol = nullptr;
    rc = i3d_fusion_create(device, voxel_size, depth_min, depth_max, clip, 1u << 22, &vol);
    if (rc != I3D_OK) { std::fprintf(stderr, rc == I3D_ERR_NO_DEVICE ? "no HIP device %d\n" : "Could not create voxel grid! (%d)\n", rc == I3D_ERR_NO_DEVICE ? device : rc); return 1; }
    std::printf("SDF volume info:\n   voxel size: %g\n   truncation: %g\n   integration depth min: %g\n   integration depth max: %g\n", (double)voxel_size, (double)(voxel_size * 5.0f),
                (double)depth_min, (double)depth_max);

    const int erode = std::atoi(yaml(fusion_cfg, "discont_window_size", "0").c_str());
    std::vector<float> depth((size_t)dwh[0] * dwh[1]), pose(16); std::vector<uint8_t> bgr((size_t)cwh[0] * cwh[1] * 3);
    std::printf("Fusion...\n");
    for (int i = 0; i < num_frames; ++i) {
        if (use_kf && !((size_t)i < is_kf.size() && is_kf[i])) continue;
        std::printf("   integrating frame %d... \n", i);
        if (i3d_sensor_depth(sensor, i, depth.data()) != I3D_OK || i3d_sensor_color(sensor, i, bgr.data()) != I3D_OK) continue;       // a frame that was not loaded: empty cv::Mat in the reference
        i3d_sensor_pose(sensor, i, pose.data());
        if (i3d_fusion_integrate(vol, dwh[0], dwh[1], di, cwh[0], cwh[1], ci, depth.data(), bgr.data(), pose.data(), erode) != I3D_OK) {
            std::fprintf(stderr, "SDF fusion failed! %s\n", i3d_fusion_last_error(vol)); return 1;
        }
    }
    std::printf("correct SDF ...\nclear invalid voxels ...\n");
    uint64_t count = 0;
    if (i3d_fusion_finish(vol, 10, &count) != I3D_OK) { std::fprintf(stderr, "SDF fusion failed! %s\n", i3d_fusion_last_error(vol)); return 1; }
    std::printf("Saving SDF (%llu voxels) ...\n", (unsigned long long)count);
    const std::string sdf_file = yaml(fusion_cfg, "output_sdf");
    if (!sdf_file.empty() && i3d_fusion_save(vol, sdf_file.c_str()) != I3D_OK) std::fprintf(stderr, "Could not save SDF volume to file ...\n");

    std::printf("Saving mesh ...\n");
    const std::string mesh_file = yaml(fusion_cfg, "output_mesh");
    if (!mesh_file.empty() && count > 0) {
        std::vector<int32_t> keys(3 * count); std::vector<float> sdf(count), weight(count); std::vector<uint8_t> color(3 * count);
        i3d_fusion_get(vol, keys.data(), sdf.data(), weight.data(), color.data());
        // MarchingCubes<Voxel>::extractSurface(*grid) walks the FUSION grid itself (app_fusion.cpp:186-193): its visit order is the record order of the
        // volume — not the order a load + convert of the saved file would give (what i3d_set_grid_from_tsdf_records restates for the refinement app)
        std::vector<double> sdf_d(sdf.begin(), sdf.end()), albedo(count, 0.0);
        i3d_grid_view gv; gv.num_voxels = (int64_t)count; gv.voxel_size = voxel_size; gv.truncation = voxel_size * 5.0f; gv.keys = keys.data();
        gv.sdf = sdf_d.data(); gv.sdf_refined = sdf_d.data(); gv.albedo = albedo.data(); gv.weight = weight.data(); gv.color = color.data();
        i3d_context* ctx = nullptr;
        if (i3d_create(device, &ctx) != I3D_OK || i3d_set_grid(ctx, &gv) != I3D_OK)
            std::fprintf(stderr, "Mesh could not be generated!\n");
        else if (i3d_export_mesh_ply(ctx, mesh_file.c_str(), 0, 0, 0) != I3D_OK) std::fprintf(stderr, "Mesh could not be saved!\n");
        if (ctx) i3d_destroy(ctx);
    }
    i3d_fusion_destroy(vol); i3d_sensor_close(sensor);
    return 0;
}
