// On-disk formats either side of the path.  Host-only: no device work, usable without a GPU.
//   .tsdf volume            SparseVoxelGrid<Voxel>::save / load        (sdf/sparse_voxel_grid.cpp:484-569)
//   VoxelSBR level dump     SparseVoxelGrid<VoxelSBR>::save / load     (same template; record = key + the 32-byte VoxelSBR, sparse_voxel_grid.h:69-77)
//   poses_*.txt             Sensor::savePoses (rgbd/sensor.cpp:315-347): TUM trajectory "t tx ty tz qx qy qz qw", fixed 6 decimals
//   intrinsics_*.txt        Camera::save / Camera::load (camera.cpp:202-274)
//   intrinsic3d.yml         the `key: "value"` map read through cv::FileStorage by the apps (data/intrinsic3d.yml)
#include "../../../include/intrinsic3d_hip.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace {

#pragma pack(push, 1)
struct TsdfHeader { float voxel_size, truncation, integration_weight_sample; uint64_t count; float max_load_factor; };     // 24 bytes, written field by field
struct VoxelRec { int32_t x, y, z; float sdf, weight; uint8_t r, g, b, pad; };                                              // Vec3i + Voxel (sparse_voxel_grid.h:56-62)
struct VoxelSbrRec { int32_t x, y, z; double sdf; float weight; uint8_t r, g, b, pad; double albedo, sdf_refined; };       // Vec3i + VoxelSBR (offsets 0/8/12/16/24)
#pragma pack(pop)
static_assert(sizeof(TsdfHeader) == 24 && sizeof(VoxelRec) == 24 && sizeof(VoxelSbrRec) == 44, "on-disk layouts");

bool read_header(std::ifstream& f, TsdfHeader& h) { f.read((char*)&h, sizeof(h)); return f.good(); }

// Eigen::Quaternionf(Matrix3f) (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<Other,3,3>) restated on floats
void rot_to_quat(const float m[9], float q[4] /*x y z w*/) {
    float t = m[0] + m[4] + m[8];
    if (t > 0.0f) {
        t = std::sqrt(t + 1.0f); q[3] = 0.5f * t; t = 0.5f / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0; if (m[4] > m[0]) i = 1; if (m[8] > m[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0f); q[i] = 0.5f * t; t = 0.5f / t;
        q[3] = (m[3 * k + j] - m[3 * j + k]) * t; q[j] = (m[3 * j + i] + m[3 * i + j]) * t; q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    }
}
// math::poseVecAAToMat (math.cpp:151-163): angle-axis + translation -> rigid transform (fp64)
void pose_to_mat(const double* p, double R[9], double t[3]) {
    const double th = std::sqrt(p[0] * p[0] + (p[1] * p[1] + p[2] * p[2]));
    double k[3] = {0, 0, 0};
    if (th > 0.0) { k[0] = p[0] / th; k[1] = p[1] / th; k[2] = p[2] / th; }
    const double c = std::cos(th), s = std::sin(th), v = 1.0 - c;
    R[0] = c + k[0] * k[0] * v;        R[1] = k[0] * k[1] * v - k[2] * s; R[2] = k[0] * k[2] * v + k[1] * s;
    R[3] = k[1] * k[0] * v + k[2] * s; R[4] = c + k[1] * k[1] * v;        R[5] = k[1] * k[2] * v - k[0] * s;
    R[6] = k[2] * k[0] * v - k[1] * s; R[7] = k[2] * k[1] * v + k[0] * s; R[8] = c + k[2] * k[2] * v;
    t[0] = p[3]; t[1] = p[4]; t[2] = p[5];
}
// the flat `key: "value"` map the apps read through cv::FileStorage (data/*.yml): `%YAML:1.0` directive, # comments, quoted scalars
bool read_flat_yaml(const char* path, std::map<std::string, std::string>& kv) {
    std::ifstream f(path); if (!f.is_open()) return false;
    std::string line;
    auto trim = [](const std::string& s) { const char* ws = " \t\r\n\""; const size_t a = s.find_first_not_of(ws); if (a == std::string::npos) return std::string(); return s.substr(a, s.find_last_not_of(ws) - a + 1); };
    while (std::getline(f, line)) {
        bool quoted = false; size_t hash = std::string::npos;          // '#' starts a comment only outside a quoted value (paths may contain it)
        for (size_t i = 0; i < line.size(); ++i) { if (line[i] == '"') quoted = !quoted; else if (line[i] == '#' && !quoted) { hash = i; break; } }
        if (hash != std::string::npos) line.erase(hash);
        const size_t colon = line.find(':'); if (colon == std::string::npos || line.empty() || line[0] == '%') continue;
        const std::string k = trim(line.substr(0, colon)), v = trim(line.substr(colon + 1));
        if (!k.empty()) kv[k] = v;
    }
    return true;
}
std::string fmt_default(float v) { char b[64]; std::snprintf(b, sizeof(b), "%g", (double)v); return b; }      // operator<<(float): precision 6, %g

}  // namespace

extern "C" {

int i3d_tsdf_read_header(const char* path, float* voxel_size, float* truncation, float* integration_weight_sample, uint64_t* count, float* max_load_factor) {
    if (!path) return I3D_ERR_INVALID_ARGUMENT;
    std::ifstream f(path, std::ios::binary); if (!f.is_open()) return I3D_ERR_IO;
    TsdfHeader h; if (!read_header(f, h)) return I3D_ERR_IO;
    if (voxel_size) *voxel_size = h.voxel_size; if (truncation) *truncation = h.truncation;
    if (integration_weight_sample) *integration_weight_sample = h.integration_weight_sample;
    if (count) *count = h.count; if (max_load_factor) *max_load_factor = h.max_load_factor;
    return I3D_OK;
}

// records in FILE order (feed them to i3d_set_grid_from_tsdf_records, which restates the map insertion order of load + convert)
int i3d_tsdf_read_records(const char* path, uint64_t capacity, int32_t* keys, float* sdf, float* weight, uint8_t* color) {
    if (!path || !keys || !sdf || !weight || !color) return I3D_ERR_INVALID_ARGUMENT;
    std::ifstream f(path, std::ios::binary); if (!f.is_open()) return I3D_ERR_IO;
    TsdfHeader h; if (!read_header(f, h)) return I3D_ERR_IO;
    if (h.count > capacity) return I3D_ERR_CAPACITY;
    std::vector<VoxelRec> buf(1 << 16);
    for (uint64_t done = 0; done < h.count;) {
        const uint64_t n = std::min<uint64_t>(buf.size(), h.count - done);
        f.read((char*)buf.data(), (std::streamsize)(n * sizeof(VoxelRec)));
        if (!f.good()) return I3D_ERR_IO;                                  // truncated file (the reference asserts)
        for (uint64_t i = 0; i < n; ++i) {
            const VoxelRec& r = buf[i]; const uint64_t o = done + i;
            keys[3 * o] = r.x; keys[3 * o + 1] = r.y; keys[3 * o + 2] = r.z; sdf[o] = r.sdf; weight[o] = r.weight;
            color[3 * o] = r.r; color[3 * o + 1] = r.g; color[3 * o + 2] = r.b;
        }
        done += n;
    }
    return I3D_OK;
}

int i3d_tsdf_write(const char* path, float voxel_size, float truncation, float integration_weight_sample, float max_load_factor, uint64_t count,
                   const int32_t* keys, const float* sdf, const float* weight, const uint8_t* color) {
    if (!path || (count && (!keys || !sdf || !weight || !color))) return I3D_ERR_INVALID_ARGUMENT;
    std::ofstream f(path, std::ios::binary); if (!f.is_open()) return I3D_ERR_IO;
    TsdfHeader h{voxel_size, truncation, integration_weight_sample, count, max_load_factor};
    f.write((const char*)&h, sizeof(h));
    std::vector<VoxelRec> buf; buf.reserve(1 << 16);
    for (uint64_t i = 0; i < count; ++i) {
        buf.push_back(VoxelRec{keys[3 * i], keys[3 * i + 1], keys[3 * i + 2], sdf[i], weight[i], color[3 * i], color[3 * i + 1], color[3 * i + 2], 0});   // pad byte: uninitialised in the reference
        if (buf.size() == (1 << 16) || i + 1 == count) { f.write((const char*)buf.data(), (std::streamsize)(buf.size() * sizeof(VoxelRec))); buf.clear(); }
    }
    return f.good() ? I3D_OK : I3D_ERR_IO;
}

// VoxelSBR level dumps (what SparseVoxelGrid<VoxelSBR>::save writes): arrays as returned by i3d_export_grid (visit order)
int i3d_sbr_write(const char* path, float voxel_size, float truncation, float integration_weight_sample, float max_load_factor, uint64_t count,
                  const int32_t* keys, const double* sdf, const double* sdf_refined, const double* albedo, const float* weight, const uint8_t* color) {
    if (!path || (count && (!keys || !sdf || !sdf_refined || !albedo || !weight || !color))) return I3D_ERR_INVALID_ARGUMENT;
    std::ofstream f(path, std::ios::binary); if (!f.is_open()) return I3D_ERR_IO;
    TsdfHeader h{voxel_size, truncation, integration_weight_sample, count, max_load_factor};
    f.write((const char*)&h, sizeof(h));
    for (uint64_t i = 0; i < count; ++i) {
        const VoxelSbrRec r{keys[3 * i], keys[3 * i + 1], keys[3 * i + 2], sdf[i], weight[i], color[3 * i], color[3 * i + 1], color[3 * i + 2], 0, albedo[i], sdf_refined[i]};
        f.write((const char*)&r, sizeof(r));
    }
    return f.good() ? I3D_OK : I3D_ERR_IO;
}
int i3d_sbr_read(const char* path, uint64_t capacity, int32_t* keys, double* sdf, double* sdf_refined, double* albedo, float* weight, uint8_t* color) {
    if (!path || !keys || !sdf || !sdf_refined || !albedo || !weight || !color) return I3D_ERR_INVALID_ARGUMENT;
    std::ifstream f(path, std::ios::binary); if (!f.is_open()) return I3D_ERR_IO;
    TsdfHeader h; if (!read_header(f, h)) return I3D_ERR_IO;
    if (h.count > capacity) return I3D_ERR_CAPACITY;
    for (uint64_t i = 0; i < h.count; ++i) {
        VoxelSbrRec r; f.read((char*)&r, sizeof(r)); if (!f.good()) return I3D_ERR_IO;
        keys[3 * i] = r.x; keys[3 * i + 1] = r.y; keys[3 * i + 2] = r.z; sdf[i] = r.sdf; sdf_refined[i] = r.sdf_refined; albedo[i] = r.albedo; weight[i] = r.weight;
        color[3 * i] = r.r; color[3 * i + 1] = r.g; color[3 * i + 2] = r.b;
    }
    return I3D_OK;
}

// Sensor::savePoses after Intrinsic3D::finishRgbdLevel's write-back (intrinsic3d.cpp:362-368): world->cam vectors are inverted in fp64,
// cast to float, and written as camera-to-world translation + quaternion
int i3d_write_poses(const char* path, int32_t num_frames, const double* timestamps, const double* poses_world_to_cam) {
    if (!path || num_frames < 0 || (num_frames && (!timestamps || !poses_world_to_cam))) return I3D_ERR_INVALID_ARGUMENT;
    FILE* f = std::fopen(path, "w"); if (!f) return I3D_ERR_IO;
    for (int i = 0; i < num_frames; ++i) {
        double R[9], t[3]; pose_to_mat(poses_world_to_cam + 6 * i, R, t);
        // inverse of a rigid transform: R^T, -R^T t
        float Rf[9], tf[3];
        for (int a = 0; a < 3; ++a) { for (int b = 0; b < 3; ++b) Rf[3 * a + b] = (float)R[3 * b + a]; tf[a] = (float)(-(R[a] * t[0] + R[3 + a] * t[1] + R[6 + a] * t[2])); }
        float q[4]; rot_to_quat(Rf, q);
        std::fprintf(f, "%.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f\n", timestamps[i], (double)tf[0], (double)tf[1], (double)tf[2], (double)q[0], (double)q[1], (double)q[2], (double)q[3]);
    }
    return std::fclose(f) == 0 ? I3D_OK : I3D_ERR_IO;
}

// Camera::save after setIntrinsics / setDistortion (float storage, default stream formatting)
int i3d_write_intrinsics(const char* path, int32_t width, int32_t height, const double* intr, const double* dist) {
    if (!path || !intr || !dist) return I3D_ERR_INVALID_ARGUMENT;
    std::ofstream f(path); if (!f.is_open()) return I3D_ERR_IO;
    f << width << " " << height << "\n";
    f << fmt_default((float)intr[0]) << " 0 " << fmt_default((float)intr[2]) << "\n";
    f << "0 " << fmt_default((float)intr[1]) << " " << fmt_default((float)intr[3]) << "\n";
    f << "0 0 1\n";
    for (int i = 0; i < 5; ++i) f << fmt_default((float)dist[i]) << (i < 4 ? " " : "\n");
    return f.good() ? I3D_OK : I3D_ERR_IO;
}
// Camera::load (camera.cpp:202-243): returns I3D_ERR_IO when the file cannot be read; the outputs then hold what a default-constructed reference Camera holds
// after its failed load: 640 x 480 (camera.cpp:41-47), fx=fy=525, cx=319.5, cy=239.5, zero distortion.  Stricter than the reference in one respect: a file that
// opens but ends early is an error here, where the reference's unchecked stream reads report success with stale values in the missing entries.
int i3d_read_intrinsics(const char* path, int32_t* width, int32_t* height, double* intr, double* dist) {
    if (!path || !intr || !dist) return I3D_ERR_INVALID_ARGUMENT;
    std::ifstream f(path);
    int w = 0, h = 0; float K[9], d[5];
    bool ok = f.is_open() && (bool)(f >> w >> h);
    for (int i = 0; ok && i < 9; ++i) ok = (bool)(f >> K[i]);
    for (int i = 0; ok && i < 5; ++i) ok = (bool)(f >> d[i]);
    if (!ok) { intr[0] = 525.0; intr[1] = 525.0; intr[2] = 319.5; intr[3] = 239.5; for (int i = 0; i < 5; ++i) dist[i] = 0.0; if (width) *width = 640; if (height) *height = 480; return I3D_ERR_IO; }
    if (width) *width = w; if (height) *height = h;
    intr[0] = K[0]; intr[1] = K[4]; intr[2] = K[2]; intr[3] = K[5];
    for (int i = 0; i < 5; ++i) dist[i] = d[i];
    return I3D_OK;
}

// data/intrinsic3d.yml: flat `key: "value"` map (cv::FileStorage, values are quoted strings converted on access).  Unknown keys are ignored,
// missing keys keep the value already in the structs (call i3d_optimizer_config_default first).
int i3d_config_load_yaml(const char* path, i3d_refine_config* rc, i3d_optimizer_config* oc) {
    if (!path || !rc || !oc) return I3D_ERR_INVALID_ARGUMENT;
    std::map<std::string, std::string> kv;
    if (!read_flat_yaml(path, kv)) return I3D_ERR_IO;
    auto num = [&](const char* k, double& dst) { auto it = kv.find(k); if (it != kv.end() && !it->second.empty()) dst = std::atof(it->second.c_str()); };
    auto geti = [&](const char* k, int32_t& dst) { double t = dst; num(k, t); dst = (int32_t)t; };
    auto getf = [&](const char* k, float& dst) { double t = dst; num(k, t); dst = (float)t; };
    geti("num_grid_levels", rc->num_grid_levels); geti("num_rgbd_levels", rc->num_rgbd_levels);
    num("thin_shell_factor", rc->thin_shell_factor); num("thin_shell_factor_final", rc->thin_shell_factor_final);
    getf("subvolume_size_sh", rc->subvolume_size_sh); num("subvolume_sh_lamda_reg", rc->sh_lambda_reg);        // (sic) the shipped key is spelled "lamda"
    geti("clear_distant_voxels", rc->clear_distant_voxels);
    getf("occlusion_distance", rc->occlusion_distance); geti("num_observations", rc->num_observations);
    oc->occlusion_distance = rc->occlusion_distance; oc->num_observations = rc->num_observations;
    num("lambda_g", oc->lambda_g); num("lambda_r0", oc->lambda_r0); num("lambda_r1", oc->lambda_r1); num("lambda_s0", oc->lambda_s0); num("lambda_s1", oc->lambda_s1);
    num("lambda_a", oc->lambda_a); geti("iterations", oc->iterations); geti("lm_steps", oc->lm_steps);
    geti("fix_poses", oc->fix_poses); geti("fix_intrinsics", oc->fix_intrinsics); geti("fix_distortion", oc->fix_distortion);
    return I3D_OK;
}


// Sensor::create(Settings&) (rgbd/sensor.cpp:64-118) over a sensor.yml: `dataset` names the folder, `max_frames` / `min_depth` / `max_depth` configure the sensor before
// it is initialised.  Values are converted the way Settings::get<T> does it (settings.cpp:86-109): a missing key reads as "0" ("" for the folder), numbers through the
// stream extraction rules (leading number, direct rounding to float), the folder up to its first white space.
int i3d_sensor_open_yaml(const char* sensor_yml, i3d_sensor** out, float* min_depth, float* max_depth) {
    if (!sensor_yml || !out) return I3D_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    std::map<std::string, std::string> kv;
    if (!read_flat_yaml(sensor_yml, kv) || kv.empty()) return I3D_ERR_IO;                 // Settings::empty(): no sensor
    auto text = [&](const char* key, const char* missing) { const auto it = kv.find(key); return it == kv.end() ? std::string(missing) : it->second; };
    std::string folder; { std::stringstream ss(text("dataset", "")); ss >> folder; }
    int max_frames = 0; { std::stringstream ss(text("max_frames", "0")); ss >> max_frames; }
    float dmin = 0.0f, dmax = 0.0f; { std::stringstream ss(text("min_depth", "0")); ss >> dmin; } { std::stringstream ss(text("max_depth", "0")); ss >> dmax; }
    if (min_depth) *min_depth = dmin;
    if (max_depth) *max_depth = dmax;
    return i3d_sensor_open(folder.c_str(), max_frames, dmin, dmax, out);
}

// Settings::get<std::string>(key) (the apps' accessor over cv::FileStorage): the value as text; I3D_ERR_INVALID_ARGUMENT when the key is absent
int i3d_yaml_get(const char* path, const char* key, char* value, uint64_t capacity) {
    if (!path || !key || !value || capacity == 0) return I3D_ERR_INVALID_ARGUMENT;
    std::map<std::string, std::string> kv;
    if (!read_flat_yaml(path, kv)) return I3D_ERR_IO;
    const auto it = kv.find(key);
    if (it == kv.end() || it->second.size() + 1 > capacity) return I3D_ERR_INVALID_ARGUMENT;
    std::memcpy(value, it->second.c_str(), it->second.size() + 1);
    return I3D_OK;
}

}  // extern "C"
