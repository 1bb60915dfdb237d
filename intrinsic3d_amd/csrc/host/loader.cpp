// Dataset loader in front of the path.  Host code; only i3d_init_frames_from_sensor touches the device (depth
// resampling + pyramids, level_kernels.hip).
//   SensorI3d::init / listFiles / loadIntrinsics / loadPose / loadDepth / loadColor    rgbd/sensor_i3d.cpp:60-327
//   Sensor::depth / thresholdDepth / savePoses                                          rgbd/sensor.cpp:196-228,315-347
//   KeyframeSelection::load / save / selectKeyframes                                    keyframe_selection.cpp:73-106,139-207
//   Intrinsic3D::init (keyframe loop)                                                   refinement/intrinsic3d.cpp:151-203
//   math::poseMatToVecAA                                                                math.cpp:166-180
// The reference decodes PNGs with cv::imdecode(IMREAD_UNCHANGED) (libpng); OpenCV is not part of this image, so the PNG container
// (chunks, zlib stream, scanline filters, Adam7) is decoded here against the PNG specification with zlib's inflate and the channel
// layout imdecode produces (BGR[A] order, 16-bit kept, palette/low-bit-depth expanded).
#include "../../../include/intrinsic3d_hip.h"
#include <zlib.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <string>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------------------------- PNG
struct PngHeader { uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0; bool trns = false; };
struct PngImage { int w = 0, h = 0, channels = 0, depth = 0; std::vector<uint8_t> pix; };   // depth 8 or 16 (native-endian u16), interleaved

uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]); }
int channels_in(int ctype) { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4; }
// channel count of the cv::Mat imdecode(IMREAD_UNCHANGED) returns: colour types with alpha, and RGB / palette images that carry a tRNS
// chunk, come back with 4 channels; grey stays single-channel even with tRNS
int channels_out(const PngHeader& h) {
    if (h.ctype == 4 || h.ctype == 6) return 4;
    if (h.ctype == 2 || h.ctype == 3) return h.trns ? 4 : 3;
    return 1;
}
int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
// undo the scanline filters of one (sub)image in place; `raw` holds rows of 1 filter byte + rowbytes
bool unfilter(uint8_t* raw, size_t rows, size_t rowbytes, size_t bpp) {
    std::vector<uint8_t> zero(rowbytes, 0);
    const uint8_t* prev = zero.data();
    for (size_t y = 0; y < rows; ++y) {
        uint8_t* line = raw + y * (rowbytes + 1);
        const int ft = line[0];
        uint8_t* cur = line + 1;
        switch (ft) {
            case 0: break;
            case 1: for (size_t i = bpp; i < rowbytes; ++i) cur[i] = uint8_t(cur[i] + cur[i - bpp]); break;
            case 2: for (size_t i = 0; i < rowbytes; ++i) cur[i] = uint8_t(cur[i] + prev[i]); break;
            case 3: for (size_t i = 0; i < rowbytes; ++i) cur[i] = uint8_t(cur[i] + (((i >= bpp ? cur[i - bpp] : 0) + prev[i]) >> 1)); break;
            case 4: for (size_t i = 0; i < rowbytes; ++i) cur[i] = uint8_t(cur[i] + paeth(i >= bpp ? cur[i - bpp] : 0, prev[i], i >= bpp ? prev[i - bpp] : 0)); break;
            default: return false;
        }
        prev = cur;
    }
    return true;
}

static int png_decode_impl(const uint8_t* d, size_t n, PngImage& out, bool header_only) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (!d || n < 8 + 25 || std::memcmp(d, sig, 8) != 0) return I3D_ERR_IO;
    PngHeader hd; bool have_ihdr = false, have_end = false;
    std::vector<uint8_t> idat, plte, trns;
    size_t pos = 8;
    while (pos + 12 <= n && !have_end) {
        const uint32_t len = be32(d + pos);
        if (len > n - pos - 12) return I3D_ERR_IO;
        const uint8_t* type = d + pos + 4; const uint8_t* data = d + pos + 8;
        if (be32(data + len) != (uint32_t)crc32(crc32(0L, Z_NULL, 0), type, len + 4)) { if (!(type[0] & 0x20)) return I3D_ERR_IO; pos += 12 + len; continue; }
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) return I3D_ERR_IO;
            hd.w = be32(data); hd.h = be32(data + 4); hd.depth = data[8]; hd.ctype = data[9]; hd.interlace = data[12];
            if (data[10] != 0 || data[11] != 0 || hd.interlace > 1 || hd.w == 0 || hd.h == 0 || hd.w > (1u << 20) || hd.h > (1u << 20)) return I3D_ERR_IO;
            const int dd = hd.depth; bool ok = false;
            switch (hd.ctype) {
                case 0: ok = dd == 1 || dd == 2 || dd == 4 || dd == 8 || dd == 16; break;
                case 3: ok = dd == 1 || dd == 2 || dd == 4 || dd == 8; break;
                case 2: case 4: case 6: ok = dd == 8 || dd == 16; break;
            }
            if (!ok) return I3D_ERR_IO;
            have_ihdr = true;
        } else if (!have_ihdr) return I3D_ERR_IO;
        else if (!std::memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!std::memcmp(type, "tRNS", 4)) { trns.assign(data, data + len); hd.trns = true; }
        else if (!std::memcmp(type, "IDAT", 4)) { if (!header_only) idat.insert(idat.end(), data, data + len); }
        else if (!std::memcmp(type, "IEND", 4)) have_end = true;
        pos += 12 + len;
        if (header_only && have_ihdr && (!std::memcmp(type, "IDAT", 4))) break;     // tRNS precedes IDAT: the channel count is known
    }
    if (!have_ihdr) return I3D_ERR_IO;
    out.w = (int)hd.w; out.h = (int)hd.h; out.channels = channels_out(hd); out.depth = hd.depth == 16 ? 16 : 8;
    if (header_only) return I3D_OK;
    if (!have_end || idat.empty() || (hd.ctype == 3 && plte.size() < 3)) return I3D_ERR_IO;

    const int cin = channels_in(hd.ctype);
    const size_t bits = (size_t)cin * hd.depth, bpp = bits >= 8 ? bits / 8 : 1;
    struct Pass { int xs, ys, dx, dy; };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole = {0, 0, 1, 1};
    const Pass* passes = hd.interlace ? adam7 : &whole; const int npass = hd.interlace ? 7 : 1;
    auto pass_dim = [](uint32_t full, int start, int step) { return full > (uint32_t)start ? (full - start + step - 1) / step : 0u; };
    size_t total = 0;
    for (int p = 0; p < npass; ++p) {
        const size_t pw = pass_dim(hd.w, passes[p].xs, passes[p].dx), ph = pass_dim(hd.h, passes[p].ys, passes[p].dy);
        if (pw && ph) total += ph * (1 + (pw * bits + 7) / 8);
    }
    // a deflate stream expands at most ~1032:1: an IHDR that promises more pixels than the IDAT data can hold is corrupt (and would
    // otherwise size the buffers below from attacker-controlled dimensions)
    // (zlib's avail_in / avail_out are 32-bit: larger streams are refused rather than truncated)
    if (total >= ((size_t)1 << 31) || idat.size() >= ((size_t)1 << 31) || total / 1100 > idat.size() + 64) return I3D_ERR_IO;
    std::vector<uint8_t> raw(total);
    z_stream zs; std::memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) return I3D_ERR_IO;
    zs.next_in = idat.data(); zs.avail_in = (uInt)idat.size(); zs.next_out = raw.data(); zs.avail_out = (uInt)raw.size();
    const int zr = inflate(&zs, Z_FINISH);
    const size_t got = zs.total_out; inflateEnd(&zs);
    if ((zr != Z_STREAM_END && zr != Z_OK && zr != Z_BUF_ERROR) || got != total) return I3D_ERR_IO;

    // samples of every pixel at file depth
    std::vector<uint16_t> smp((size_t)hd.w * hd.h * cin);
    size_t off = 0;
    for (int p = 0; p < npass; ++p) {
        const size_t pw = pass_dim(hd.w, passes[p].xs, passes[p].dx), ph = pass_dim(hd.h, passes[p].ys, passes[p].dy);
        if (!pw || !ph) continue;
        const size_t rowbytes = (pw * bits + 7) / 8;
        if (!unfilter(raw.data() + off, ph, rowbytes, bpp)) return I3D_ERR_IO;
        for (size_t y = 0; y < ph; ++y) {
            const uint8_t* row = raw.data() + off + y * (rowbytes + 1) + 1;
            const size_t oy = passes[p].ys + y * passes[p].dy;
            for (size_t x = 0; x < pw; ++x) {
                uint16_t* dst = &smp[(oy * hd.w + passes[p].xs + x * passes[p].dx) * cin];
                for (int c = 0; c < cin; ++c) {
                    const size_t s = x * cin + c;
                    if (hd.depth == 16) dst[c] = uint16_t((row[2 * s] << 8) | row[2 * s + 1]);
                    else if (hd.depth == 8) dst[c] = row[s];
                    else { const size_t bit = s * hd.depth; dst[c] = uint16_t((row[bit >> 3] >> (8 - hd.depth - (bit & 7))) & ((1 << hd.depth) - 1)); }
                }
            }
        }
        off += ph * (rowbytes + 1);
    }

    const int cout = out.channels; const size_t npix = (size_t)hd.w * hd.h;
    const bool wide = out.depth == 16;
    out.pix.assign(npix * cout * (wide ? 2 : 1), 0);
    auto put = [&](size_t i, int c, uint16_t v) { if (wide) reinterpret_cast<uint16_t*>(out.pix.data())[i * cout + c] = v; else out.pix[i * cout + c] = (uint8_t)v; };
    const uint16_t opaque = wide ? 0xFFFF : 0xFF;
    uint16_t key[3] = {0, 0, 0};
    if (hd.trns && hd.ctype == 2 && trns.size() >= 6) for (int c = 0; c < 3; ++c) key[c] = uint16_t((trns[2 * c] << 8) | trns[2 * c + 1]);
    for (size_t i = 0; i < npix; ++i) {
        const uint16_t* s = &smp[i * cin];
        switch (hd.ctype) {
            case 0: put(i, 0, hd.depth < 8 ? uint16_t(s[0] * (255 / ((1 << hd.depth) - 1))) : s[0]); break;       // png_set_expand_gray_1_2_4_to_8
            case 2:
                put(i, 0, s[2]); put(i, 1, s[1]); put(i, 2, s[0]);
                if (cout == 4) put(i, 3, (trns.size() >= 6 && s[0] == key[0] && s[1] == key[1] && s[2] == key[2]) ? 0 : opaque);
                break;
            case 3: {
                const size_t idx = s[0];
                if (3 * idx + 2 >= plte.size()) return I3D_ERR_IO;
                put(i, 0, plte[3 * idx + 2]); put(i, 1, plte[3 * idx + 1]); put(i, 2, plte[3 * idx]);
                if (cout == 4) put(i, 3, idx < trns.size() ? trns[idx] : 0xFF);
                break;
            }
            case 4: put(i, 0, s[0]); put(i, 1, s[0]); put(i, 2, s[0]); put(i, 3, s[1]); break;                     // png_set_gray_to_rgb
            case 6: put(i, 0, s[2]); put(i, 1, s[1]); put(i, 2, s[0]); put(i, 3, s[3]); break;
        }
    }
    return I3D_OK;
}

bool read_file(const std::string& path, std::vector<uint8_t>& data) {             // SensorI3d::loadFile: false for missing or empty files
    std::ifstream f(path.c_str(), std::ios::in | std::ios::binary | std::ios::ate);
    if (!f.is_open()) return false;
    const std::streamoff size = f.tellg();
    if (size <= 0) return false;
    data.resize((size_t)size); f.seekg(0, std::ios::beg); f.read((char*)data.data(), size);
    return f.good();
}
bool read_mat4(const std::string& path, float m[16]) {                            // loadPose / loadIntrinsics: 16 floats through operator>>
    std::ifstream f(path.c_str());
    if (!f.is_open()) return false;
    float val = 0.0f;                                                              // a failed extraction leaves the previous value, like the reference
    for (int i = 0; i < 16; ++i) { f >> val; m[i] = val; }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------- poses
// Eigen's Matrix4d::inverse(): adjugate / determinant, formed from the 2x2 minors of the two row pairs
void inverse4(const double* m, double* inv) {
    const double s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2], s2 = m[0] * m[7] - m[4] * m[3];
    const double s3 = m[1] * m[6] - m[5] * m[2], s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
    const double c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11], c3 = m[9] * m[14] - m[13] * m[10];
    const double c2 = m[8] * m[15] - m[12] * m[11], c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
    const double id = 1.0 / (s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0);
    inv[0] = (m[5] * c5 - m[6] * c4 + m[7] * c3) * id;   inv[1] = (-m[1] * c5 + m[2] * c4 - m[3] * c3) * id;
    inv[2] = (m[13] * s5 - m[14] * s4 + m[15] * s3) * id; inv[3] = (-m[9] * s5 + m[10] * s4 - m[11] * s3) * id;
    inv[4] = (-m[4] * c5 + m[6] * c2 - m[7] * c1) * id;  inv[5] = (m[0] * c5 - m[2] * c2 + m[3] * c1) * id;
    inv[6] = (-m[12] * s5 + m[14] * s2 - m[15] * s1) * id; inv[7] = (m[8] * s5 - m[10] * s2 + m[11] * s1) * id;
    inv[8] = (m[4] * c4 - m[5] * c2 + m[7] * c0) * id;   inv[9] = (-m[0] * c4 + m[1] * c2 - m[3] * c0) * id;
    inv[10] = (m[12] * s4 - m[13] * s2 + m[15] * s0) * id; inv[11] = (-m[8] * s4 + m[9] * s2 - m[11] * s0) * id;
    inv[12] = (-m[4] * c3 + m[5] * c1 - m[6] * c0) * id; inv[13] = (m[0] * c3 - m[1] * c1 + m[2] * c0) * id;
    inv[14] = (-m[12] * s3 + m[13] * s1 - m[14] * s0) * id; inv[15] = (m[8] * s3 - m[9] * s1 + m[10] * s0) * id;
}
// Eigen::AngleAxisd(Matrix3d): rotation -> quaternion (Quaternion.h, the 3x3 assign) -> angle-axis (AngleAxis.h operator=(Quaternion))
void rot_to_angle_axis(const double* R /*row-major 3x3*/, double aa[3]) {
    double q[4];   // x y z w
    double t = R[0] + R[4] + R[8];
    if (t > 0.0) {
        t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0; if (R[4] > R[0]) i = 1; if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0); q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * t; q[j] = (R[3 * j + i] + R[3 * i + j]) * t; q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    }
    double n = std::sqrt(q[0] * q[0] + (q[1] * q[1] + q[2] * q[2]));          // q.vec().norm(): Eigen's fixed-size reduction a0 + (a1 + a2)
    if (n < 2.220446049250313e-16) {                        // stableNorm of a tiny vector
        const double mx = std::fmax(std::fabs(q[0]), std::fmax(std::fabs(q[1]), std::fabs(q[2])));
        n = mx > 0.0 ? mx * std::sqrt((q[0] / mx) * (q[0] / mx) + (q[1] / mx) * (q[1] / mx) + (q[2] / mx) * (q[2] / mx)) : 0.0;
    }
    if (n != 0.0) {
        const double angle = 2.0 * std::atan2(n, std::fabs(q[3]));
        if (q[3] < 0.0) n = -n;
        for (int a = 0; a < 3; ++a) aa[a] = (q[a] / n) * angle;
    } else { aa[0] = aa[1] = aa[2] = 0.0; }                // angle 0, axis (1,0,0)
}
void pose_to_vec6(const float* cam_to_world, double* p6) {       // intrinsic3d.cpp:189-192
    double m[16], inv[16]; for (int i = 0; i < 16; ++i) m[i] = (double)cam_to_world[i];
    inverse4(m, inv);
    const double R[9] = {inv[0], inv[1], inv[2], inv[4], inv[5], inv[6], inv[8], inv[9], inv[10]};
    rot_to_angle_axis(R, p6);
    p6[3] = inv[3]; p6[4] = inv[7]; p6[5] = inv[11];
}
// Eigen::Quaternionf(Matrix3f), as Sensor::savePoses uses it
void rot_to_quat_f(const float* m /*row-major 3x3*/, float q[4]) {
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

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- sensor
struct i3d_sensor {
    std::string folder;
    int max_frames = 0; float min_depth = 0.0f, max_depth = 0.0f;
    int num_frames = 0;                                    // number of frame files listed (Sensor::numFrames)
    float color_k[16], depth_k[16];
    int color_w = 0, color_h = 0, depth_w = 0, depth_h = 0;
    std::vector<std::vector<uint8_t>> color_png, depth_png;
    std::vector<double> timestamps;
    std::vector<float> poses;                              // 16 per stored frame, camera-to-world, row-major
    bool stored(int id) const { return id >= 0 && id < num_frames && (size_t)id < depth_png.size(); }
};

// no exception may cross the C ABI: allocation failures on hostile dimensions become an I/O error
int png_decode(const uint8_t* d, size_t n, PngImage& out, bool header_only) {
    try { return png_decode_impl(d, n, out, header_only); } catch (const std::exception&) { return I3D_ERR_IO; }
}

extern "C" {

int i3d_png_info(const uint8_t* data, uint64_t size, int32_t* width, int32_t* height, int32_t* channels, int32_t* bit_depth) {
    PngImage im; const int rc = png_decode(data, (size_t)size, im, true);
    if (rc != I3D_OK) return rc;
    if (width) *width = im.w; if (height) *height = im.h; if (channels) *channels = im.channels; if (bit_depth) *bit_depth = im.depth;
    return I3D_OK;
}
int i3d_png_decode(const uint8_t* data, uint64_t size, void* pixels, uint64_t capacity_bytes) {
    if (!pixels) return I3D_ERR_INVALID_ARGUMENT;
    PngImage im; const int rc = png_decode(data, (size_t)size, im, false);
    if (rc != I3D_OK) return rc;
    if (im.pix.size() > capacity_bytes) return I3D_ERR_INVALID_ARGUMENT;
    std::memcpy(pixels, im.pix.data(), im.pix.size());
    return I3D_OK;
}

int i3d_pose_mat_to_vec6(const float* cam_to_world16, double* pose6) {
    if (!cam_to_world16 || !pose6) return I3D_ERR_INVALID_ARGUMENT;
    pose_to_vec6(cam_to_world16, pose6);
    return I3D_OK;
}

int i3d_sensor_open(const char* folder, int32_t max_frames, float min_depth, float max_depth, i3d_sensor** out) {
    if (!folder || !out) return I3D_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!*folder) return I3D_ERR_IO;
    i3d_sensor* s = new i3d_sensor();
    s->folder = folder; s->max_frames = max_frames; s->min_depth = min_depth; s->max_depth = max_depth;
    for (int i = 0; i < 16; ++i) s->color_k[i] = s->depth_k[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    // a missing intrinsics file is reported and ignored (sensor_i3d.cpp:75-77,175-178)
    if (!read_mat4(s->folder + "/depthIntrinsics.txt", s->depth_k)) std::fprintf(stderr, "Intrinsics file ('%s/depthIntrinsics.txt') could not be loaded!\n", folder);
    if (!read_mat4(s->folder + "/colorIntrinsics.txt", s->color_k)) std::fprintf(stderr, "Intrinsics file ('%s/colorIntrinsics.txt') could not be loaded!\n", folder);
    std::vector<std::string> bases;
    for (size_t i = 0; i < 999999; ++i) {                                               // listFiles: stops at the first missing depth map
        char name[32]; std::snprintf(name, sizeof(name), "/frame-%06zu", i);
        const std::string base = s->folder + name;
        if (!std::ifstream((base + ".depth.png").c_str()).is_open()) break;
        bases.push_back(base);
    }
    s->num_frames = (int)bases.size();
    for (size_t i = 0; i < bases.size(); ++i) {
        std::vector<uint8_t> depth, color;
        if (!read_file(bases[i] + ".depth.png", depth) || !read_file(bases[i] + ".color.png", color)) break;
        float pose[16];
        if (!read_mat4(bases[i] + ".pose.txt", pose)) { std::fprintf(stderr, "Poses file ('%s.pose.txt') could not be loaded!\n", bases[i].c_str()); break; }
        if (i == 0) {
            PngImage c, d;
            if (png_decode(color.data(), color.size(), c, true) == I3D_OK) { s->color_w = c.w; s->color_h = c.h; }
            if (png_decode(depth.data(), depth.size(), d, true) == I3D_OK) { s->depth_w = d.w; s->depth_h = d.h; }
        }
        s->depth_png.push_back(std::move(depth)); s->color_png.push_back(std::move(color));
        s->timestamps.push_back((double)i);
        s->poses.insert(s->poses.end(), pose, pose + 16);
        if (s->max_frames > 0 && (int)s->depth_png.size() >= s->max_frames) break;
    }
    *out = s;
    return I3D_OK;
}
void i3d_sensor_close(i3d_sensor* s) { delete s; }

int i3d_sensor_info(const i3d_sensor* s, int32_t* num_frames, int32_t* num_loaded, int32_t* color_wh, int32_t* depth_wh, float* color_intr4, float* depth_intr4) {
    if (!s) return I3D_ERR_INVALID_ARGUMENT;
    if (num_frames) *num_frames = s->num_frames;
    if (num_loaded) *num_loaded = (int32_t)s->depth_png.size();
    if (color_wh) { color_wh[0] = s->color_w; color_wh[1] = s->color_h; }
    if (depth_wh) { depth_wh[0] = s->depth_w; depth_wh[1] = s->depth_h; }
    if (color_intr4) { color_intr4[0] = s->color_k[0]; color_intr4[1] = s->color_k[5]; color_intr4[2] = s->color_k[2]; color_intr4[3] = s->color_k[6]; }
    if (depth_intr4) { depth_intr4[0] = s->depth_k[0]; depth_intr4[1] = s->depth_k[5]; depth_intr4[2] = s->depth_k[2]; depth_intr4[3] = s->depth_k[6]; }
    return I3D_OK;
}

// Sensor::color: the decoded image as imdecode returns it; an alpha channel is dropped here (only B,G,R are read downstream)
int i3d_sensor_color(const i3d_sensor* s, int32_t id, uint8_t* bgr) {
    if (!s || !bgr || !s->stored(id)) return I3D_ERR_INVALID_ARGUMENT;
    PngImage im; const int rc = png_decode(s->color_png[id].data(), s->color_png[id].size(), im, false);
    if (rc != I3D_OK) return rc;
    if (im.w != s->color_w || im.h != s->color_h) return I3D_ERR_IO;
    const size_t n = (size_t)im.w * im.h;
    for (size_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) {
            const int sc = im.channels == 1 ? 0 : c;
            bgr[3 * i + c] = im.depth == 16 ? (uint8_t)(reinterpret_cast<const uint16_t*>(im.pix.data())[i * im.channels + sc] >> 8) : im.pix[i * im.channels + sc];
        }
    return I3D_OK;
}

// Sensor::depth = loadDepth (u16 millimetres -> float metres, cv::Mat::convertTo with scale 1/1000 evaluated in float) + thresholdDepth
int i3d_sensor_depth(const i3d_sensor* s, int32_t id, float* depth) {
    if (!s || !depth || !s->stored(id)) return I3D_ERR_INVALID_ARGUMENT;
    PngImage im; const int rc = png_decode(s->depth_png[id].data(), s->depth_png[id].size(), im, false);
    if (rc != I3D_OK) return rc;
    if (im.w != s->depth_w || im.h != s->depth_h || im.channels != 1) return I3D_ERR_IO;
    const size_t n = (size_t)im.w * im.h;
    const float scale = (float)(1.0 / 1000.0);
    for (size_t i = 0; i < n; ++i) {
        const float raw = im.depth == 16 ? (float)reinterpret_cast<const uint16_t*>(im.pix.data())[i] : (float)im.pix[i];
        float d = raw * scale;
        if (s->min_depth > 0.0f && !(d > s->min_depth)) d = 0.0f;      // THRESH_TOZERO
        if (s->max_depth > 0.0f && d > s->max_depth) d = 0.0f;         // THRESH_TOZERO_INV
        depth[i] = d;
    }
    return I3D_OK;
}

int i3d_sensor_pose(const i3d_sensor* s, int32_t id, float* cam_to_world16) {
    if (!s || !cam_to_world16) return I3D_ERR_INVALID_ARGUMENT;
    if (!s->stored(id)) { for (int i = 0; i < 16; ++i) cam_to_world16[i] = (i % 5 == 0) ? 1.0f : 0.0f; return I3D_OK; }   // SensorI3d::pose: identity
    std::memcpy(cam_to_world16, &s->poses[16 * (size_t)id], 16 * sizeof(float));
    return I3D_OK;
}
int i3d_sensor_set_pose(i3d_sensor* s, int32_t id, const float* cam_to_world16) {
    if (!s || !cam_to_world16) return I3D_ERR_INVALID_ARGUMENT;
    if (s->stored(id)) std::memcpy(&s->poses[16 * (size_t)id], cam_to_world16, 16 * sizeof(float));
    return I3D_OK;
}
// the write-back of Intrinsic3D::finishRgbdLevel (intrinsic3d.cpp:362-368): world->camera vector -> camera-to-world Mat4f
int i3d_sensor_set_pose_vec6(i3d_sensor* s, int32_t id, const double* p) {
    if (!s || !p) return I3D_ERR_INVALID_ARGUMENT;
    const double th = std::sqrt(p[0] * p[0] + (p[1] * p[1] + p[2] * p[2]));
    double k[3] = {0, 0, 0}; if (th > 0.0) { k[0] = p[0] / th; k[1] = p[1] / th; k[2] = p[2] / th; }
    const double c = std::cos(th), sn = std::sin(th), v = 1.0 - c;
    const double m[16] = {c + k[0] * k[0] * v, k[0] * k[1] * v - k[2] * sn, k[0] * k[2] * v + k[1] * sn, p[3],
                          k[1] * k[0] * v + k[2] * sn, c + k[1] * k[1] * v, k[1] * k[2] * v - k[0] * sn, p[4],
                          k[2] * k[0] * v - k[1] * sn, k[2] * k[1] * v + k[0] * sn, c + k[2] * k[2] * v, p[5], 0, 0, 0, 1};
    double inv[16]; inverse4(m, inv);
    float f[16]; for (int i = 0; i < 16; ++i) f[i] = (float)inv[i];
    return i3d_sensor_set_pose(s, id, f);
}

int i3d_sensor_save_poses(const i3d_sensor* s, const char* path) {
    if (!s || !path || !*path) return I3D_ERR_INVALID_ARGUMENT;
    FILE* f = std::fopen(path, "w"); if (!f) return I3D_ERR_IO;
    for (int i = 0; i < s->num_frames; ++i) {
        float m[16]; i3d_sensor_pose(s, i, m);
        const float R[9] = {m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]};
        float q[4]; rot_to_quat_f(R, q);
        const double ts = s->stored(i) ? s->timestamps[i] : 0.0;
        std::fprintf(f, "%.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f\n", ts, (double)m[3], (double)m[7], (double)m[11], (double)q[0], (double)q[1], (double)q[2], (double)q[3]);
    }
    return std::fclose(f) == 0 ? I3D_OK : I3D_ERR_IO;
}

// ---------------------------------------------------------------------------------------------------------------- keyframes
int i3d_keyframes_load(const char* path, int32_t* window_size, uint64_t capacity, double* scores, uint8_t* is_keyframe, uint64_t* count) {
    if (!path || !count) return I3D_ERR_INVALID_ARGUMENT;
    std::ifstream file(path); if (!file.is_open()) return I3D_ERR_IO;
    std::string line; uint64_t n = 0;
    if (std::getline(file, line)) {
        if (line.empty()) return I3D_ERR_IO;
        std::istringstream iss(line); int w;
        if (!(iss >> w)) return I3D_ERR_IO;
        if (window_size) *window_size = w;
    }
    while (std::getline(file, line)) {
        if (line.empty()) continue;
        std::istringstream iss(line); double score; bool kf;
        if (!(iss >> score >> kf)) break;
        if (n < capacity) { if (scores) scores[n] = score; if (is_keyframe) is_keyframe[n] = kf ? 1 : 0; }
        ++n;
    }
    *count = n;
    return I3D_OK;
}
int i3d_keyframes_save(const char* path, int32_t window_size, uint64_t count, const double* scores, const uint8_t* is_keyframe) {
    if (!path || !count || !scores || !is_keyframe) return I3D_ERR_INVALID_ARGUMENT;
    FILE* f = std::fopen(path, "w"); if (!f) return I3D_ERR_IO;
    std::fprintf(f, "%d\n", window_size);
    for (uint64_t i = 0; i < count; ++i) std::fprintf(f, "%.6f %d\n", scores[i], is_keyframe[i] ? 1 : 0);
    return std::fclose(f) == 0 ? I3D_OK : I3D_ERR_IO;
}
int i3d_keyframes_select(int32_t window_size, uint64_t count, const double* scores, uint8_t* is_keyframe) {
    if (window_size <= 0 || (count && (!scores || !is_keyframe))) return I3D_ERR_INVALID_ARGUMENT;
    for (uint64_t beg = 0; beg < count; beg += (uint64_t)window_size) {
        const uint64_t end = beg + (uint64_t)window_size < count ? beg + (uint64_t)window_size : count;
        double best = 0.0; uint64_t arg = beg;
        for (uint64_t i = beg; i < end; ++i) if (scores[i] > best) { best = scores[i]; arg = i; }
        for (uint64_t i = beg; i < end; ++i) is_keyframe[i] = (i == arg);
    }
    return I3D_OK;
}

// KeyframeSelection::estimateBlur / estimateBlurCrete (keyframe_selection.cpp:219-311): the no-reference perceptual blur metric of Crete et
// al. 2007 on the grey image in [0,1]; 1.0 = sharp, 0.0 = blurred.  8-bit BGR -> grey with cv::cvtColor's fixed-point weights
// (R 4899, G 9617, B 1868, >> 14 with rounding); the 9-tap box filters use BORDER_REFLECT_101 like cv::filter2D's default.
int i3d_blur_score(const uint8_t* image, int32_t width, int32_t height, int32_t channels, double* score) {
    if (!image || !score || width <= 0 || height <= 0) return I3D_ERR_INVALID_ARGUMENT;
    *score = 0.0;
    if (channels != 1 && channels != 3) return I3D_OK;                              // estimateBlur returns 0.0 for other layouts
    const int w = width, h = height; const size_t n = (size_t)w * h;
    std::vector<float> g(n), bv(n), bh(n);
    const float s255 = (float)(1.0 / 255.0);
    for (size_t i = 0; i < n; ++i) {
        const int v = channels == 1 ? image[i] : (image[3 * i] * 1868 + image[3 * i + 1] * 9617 + image[3 * i + 2] * 4899 + (1 << 13)) >> 14;
        g[i] = (float)v * s255;
    }
    auto refl = [](int p, int len) { if (len == 1) return 0; while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p; return p; };
    const float k9 = (float)(1.0 / 9.0);
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        float sv = 0.0f, sh = 0.0f;
        for (int k = -4; k <= 4; ++k) { sv += k9 * g[(size_t)refl(y + k, h) * w + x]; sh += k9 * g[(size_t)y * w + refl(x + k, w)]; }
        bv[(size_t)y * w + x] = sv; bh[(size_t)y * w + x] = sh;
    }
    double s_f_ver = 0, s_v_ver = 0, s_f_hor = 0, s_v_hor = 0;
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        const size_t i = (size_t)y * w + x;
        if (y >= 1) { const float df = std::fabs(g[i] - g[i - w]), db = std::fabs(bv[i] - bv[i - w]); s_f_ver += df; s_v_ver += std::max(0.0f, df - db); }
        if (x >= 1) { const float df = std::fabs(g[i] - g[i - 1]), db = std::fabs(bh[i] - bh[i - 1]); s_f_hor += df; s_v_hor += std::max(0.0f, df - db); }
    }
    const double b_ver = (s_f_ver - s_v_ver) / s_f_ver, b_hor = (s_f_hor - s_v_hor) / s_f_hor;
    *score = 1.0 - std::max(b_ver, b_hor);
    return I3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------- Intrinsic3D::init
int i3d_init_frames_from_sensor(i3d_context* ctx, int32_t device_ordinal, const i3d_sensor* s, uint64_t num_flags, const uint8_t* is_keyframe, int32_t num_rgbd_levels,
                                int32_t frame_capacity, int32_t* frame_ids, int32_t* num_keyframes) {
    if (!ctx || !s || !num_keyframes || num_rgbd_levels < 1 || (num_flags && !is_keyframe)) return I3D_ERR_INVALID_ARGUMENT;
    std::vector<int> ids;
    for (int i = 0; i < s->num_frames; ++i) if ((uint64_t)i < num_flags && is_keyframe[i]) ids.push_back(i);
    *num_keyframes = (int32_t)ids.size();
    if (ids.empty() || s->color_w <= 0 || s->depth_w <= 0) return I3D_ERR_INVALID_ARGUMENT;
    const size_t cn = (size_t)s->color_w * s->color_h, dn = (size_t)s->depth_w * s->depth_h;
    std::vector<std::vector<uint8_t>> bgr(ids.size()); std::vector<std::vector<float>> depth(ids.size());
    std::vector<const uint8_t*> bgr_p(ids.size()); std::vector<const float*> depth_p(ids.size());
    std::vector<double> poses(6 * ids.size());
    std::vector<float> raw(dn);
    float ci[4], di[4]; i3d_sensor_info(s, nullptr, nullptr, nullptr, nullptr, ci, di);
    for (size_t k = 0; k < ids.size(); ++k) {
        if (!s->stored(ids[k])) return I3D_ERR_IO;                                       // the reference would hand an empty cv::Mat to the pyramid
        bgr[k].resize(3 * cn); depth[k].resize(cn);
        int rc = i3d_sensor_color(s, ids[k], bgr[k].data()); if (rc != I3D_OK) return rc;
        rc = i3d_sensor_depth(s, ids[k], raw.data()); if (rc != I3D_OK) return rc;
        rc = i3d_resize_depth(device_ordinal, s->depth_w, s->depth_h, raw.data(), di, s->color_w, s->color_h, ci, depth[k].data()); if (rc != I3D_OK) return rc;
        bgr_p[k] = bgr[k].data(); depth_p[k] = depth[k].data();
        pose_to_vec6(&s->poses[16 * (size_t)ids[k]], &poses[6 * k]);
        if (frame_ids && (int32_t)k < frame_capacity) frame_ids[k] = ids[k];
    }
    int rc = i3d_set_frames_rgbd(ctx, (int32_t)ids.size(), num_rgbd_levels, s->color_w, s->color_h, bgr_p.data(), depth_p.data());
    if (rc != I3D_OK) return rc;
    const double intr[4] = {ci[0], ci[1], ci[2], ci[3]}, dist[5] = {0, 0, 0, 0, 0};      // intrinsic3d.cpp:157-159
    return i3d_set_camera(ctx, intr, dist, poses.data());
}

}  // extern "C"
