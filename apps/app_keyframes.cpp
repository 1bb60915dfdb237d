// AppKeyframes on the MI355X library (apps/src/app_keyframes.cpp:58-144 of the reference), written against include/intrinsic3d_hip.h only.
// Host-only: the blur score of every colour frame (KeyframeSelection::estimateBlur), the sharpest frame of every window becomes a keyframe.
//
//   app_keyframes -s <path>/sensor.yml -k <path>/keyframes.yml
#include "../include/intrinsic3d_hip.h"
#include <climits>
#include <cstdio>
#include <cstdlib>
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
}  // namespace

int main(int argc, char* argv[]) {
    std::string sensor_cfg, kf_cfg;
    for (int i = 1; i < argc; ++i) {
        std::string arg = argv[i], val; const size_t eq = arg.find('=');
        if (eq != std::string::npos) { val = arg.substr(eq + 1); arg = arg.substr(0, eq); } else if (i + 1 < argc) val = argv[++i];
        if (arg == "-s" || arg == "--sensor") sensor_cfg = val;
        else if (arg == "-k" || arg == "--keyframes") kf_cfg = val;
        else { std::fprintf(stderr, "usage: %s -s <sensor.yml> -k <keyframes.yml>\n", argv[0]); return 2; }
    }
    if (sensor_cfg.empty() || kf_cfg.empty()) { std::fprintf(stderr, "usage: %s -s <sensor.yml> -k <keyframes.yml>\n", argv[0]); return 2; }
    sensor_cfg = absolute(sensor_cfg); kf_cfg = absolute(kf_cfg);
    const std::string dir = sensor_cfg.substr(0, sensor_cfg.find_last_of('/'));
    if (chdir(dir.c_str()) != 0) { std::fprintf(stderr, "cannot change the working directory to %s\n", dir.c_str()); return 1; }
    mkdir("./fusion", 0755);

    i3d_sensor* sensor = nullptr;
    int rc = i3d_sensor_open_yaml(sensor_cfg.c_str(), &sensor, nullptr, nullptr);                          // Sensor::create(sensor_cfg)
    int32_t num_frames = 0, num_loaded = 0, cwh[2] = {0, 0};
    if (rc == I3D_OK) i3d_sensor_info(sensor, &num_frames, &num_loaded, cwh, nullptr, nullptr, nullptr);
    if (rc != I3D_OK || num_loaded == 0) { std::fprintf(stderr, "RGB-D sensor could not be initialized!\n"); return 1; }

    const std::string out_file = yaml(kf_cfg, "filename"); const int window = std::atoi(yaml(kf_cfg, "window_size", "0").c_str());
    std::printf("keyframes_file %s\nkeyframe_selection_window %d\n", out_file.c_str(), window);
    if (out_file.empty() || window == 0) { std::fprintf(stderr, "Keyframe selection failed!\n"); return 1; }
    std::vector<double> scores(num_frames, 0.0); std::vector<uint8_t> is_kf(num_frames, 0), bgr((size_t)cwh[0] * cwh[1] * 3);
    for (int i = 0; i < num_frames; ++i) {
        if (i % 50 == 0) std::printf("Keyframe selection frame %d... \n", i);
        if (i3d_sensor_color(sensor, i, bgr.data()) == I3D_OK) i3d_blur_score(bgr.data(), cwh[0], cwh[1], 3, &scores[i]);     // a frame that was not loaded scores 0 (empty cv::Mat)
    }
    if (i3d_keyframes_select(window, (uint64_t)num_frames, scores.data(), is_kf.data()) != I3D_OK ||
        i3d_keyframes_save(out_file.c_str(), window, (uint64_t)num_frames, scores.data(), is_kf.data()) != I3D_OK) { std::fprintf(stderr, "Keyframe selection failed!\n"); return 1; }
    i3d_sensor_close(sensor);
    return 0;
}
