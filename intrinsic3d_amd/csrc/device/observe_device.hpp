// Float arithmetic of one voxel observation, shared by the observation pass (K1) and the recolourisation kernel.
// SDFColorization::computeObservation / isVoxelVisible / computeWeight (sdf/colorization.cpp:215-315) + Camera::project
// (camera.cpp:124-154), operation for operation; translation units including this header are compiled with -ffp-contract=off.
#pragma once
#include "common.hpp"

namespace i3d {

// (px,py,pz): iso-projected voxel position (world, float); (nx,ny,nz): unit normal.  Returns the observation weight (0 = not observed)
// and the float pixel coordinates.
static __device__ inline float observation_weight(const FrameConst& fc, const OptParams& p, float px, float py, float pz,
                                                  float nx, float ny, float nz, const float* __restrict__ depth, float& u, float& v) {
    const float qx = ((fc.Rf[0] * px + fc.Rf[1] * py) + fc.Rf[2] * pz) + fc.tf[0];
    const float qy = ((fc.Rf[3] * px + fc.Rf[4] * py) + fc.Rf[5] * pz) + fc.tf[1];
    const float qz = ((fc.Rf[6] * px + fc.Rf[7] * py) + fc.Rf[8] * pz) + fc.tf[2];
    float x = qx / qz, y = qy / qz;
    if (!p.dist_zero) {
        const float r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        const float dc = 1.0f + p.dist_f[0] * r2 + p.dist_f[1] * r4 + p.dist_f[2] * r6;
        x = x * dc + 2.0f * p.dist_f[3] * x * y + p.dist_f[4] * (r2 + 2.0f * x * x);
        y = y * dc + 2.0f * p.dist_f[4] * x * y + p.dist_f[3] * (r2 + 2.0f * y * y);
    }
    u = p.cam_f[0] * x + p.cam_f[2]; v = p.cam_f[1] * y + p.cam_f[3];
    const int ui = (int)(u + 0.5f), vi = (int)(v + 0.5f);
    float w = 0.0f;
    if (!(ui < 0 || ui >= p.w || vi < 0 || vi >= p.h)) {
        const float d = depth[(size_t)vi * fc.w + ui];
        bool vis = true;
        if (p.occlusion > 0.0f) vis = (d > 0.0f) && (fabsf(d - qz) <= p.occlusion);
        if (vis && d > 0.0f) {
            const float cnx = (fc.Rf[0] * nx + fc.Rf[1] * ny) + fc.Rf[2] * nz;
            const float cny = (fc.Rf[3] * nx + fc.Rf[4] * ny) + fc.Rf[5] * nz;
            const float cnz = (fc.Rf[6] * nx + fc.Rf[7] * ny) + fc.Rf[8] * nz;
            float wn = 0.0f;
            if (!(fabsf(cnx) <= 1e-5f && fabsf(cny) <= 1e-5f && fabsf(cnz) <= 1e-5f)) {
                const float vsq = qx * qx + (qy * qy + qz * qz);          // fixed-size Eigen reductions: a0 + (a1 + a2)
                float vx = qx, vy = qy, vz = qz;
                if (vsq > 0.0f) { const float l = sqrtf(vsq); vx /= l; vy /= l; vz /= l; }
                wn = 1.0f - fabsf(vx * cnx + (vy * cny + vz * cnz));
                wn = fmaxf(fminf(wn, 1.0f), 0.0f);
                const float div = 1.0f + 2.0f * wn;
                wn = fmaxf(1.0f / (div * div * div), 0.001f);
            }
            const float dw = fmaxf(fminf(5.0f, d), 0.01f);
            const float dn = (dw - 0.01f) / (5.0f - 0.01f);
            float wd = fmaxf(1.0f - dn, 1.0f);
            wd = fmaxf(fminf(wd, 5.0f), 0.001f);
            w = wn * wd;
        }
    }
    return w;
}

// interpolate<unsigned char>(img, x, y, channel) on an interleaved 3-channel image (rgbd/processing.cpp:238-287)
static __device__ inline unsigned char bilinear_u8(const uint8_t* __restrict__ img, int w, int h, float x, float y, int ch) {
    int x0 = (int)floorf(x), y0 = (int)floorf(y); const int x1 = x0 + 1, y1 = y0 + 1;
    float x1w = x - (float)x0, y1w = y - (float)y0, x0w = 1.0f - x1w, y0w = 1.0f - y1w;
    if (x0 < 0 || x0 >= w) x0w = 0.0f;
    if (x1 < 0 || x1 >= w) x1w = 0.0f;
    if (y0 < 0 || y0 >= h) y0w = 0.0f;
    if (y1 < 0 || y1 >= h) y1w = 0.0f;
    const float w00 = x0w * y0w, w10 = x1w * y0w, w01 = x0w * y1w, w11 = x1w * y1w;
    const float sw = w00 + w10 + w01 + w11;
    float sum = 0.0f;
    if (w00 > 0.0f) sum += (float)img[((size_t)y0 * w + x0) * 3 + ch] * w00;
    if (w01 > 0.0f) sum += (float)img[((size_t)y1 * w + x0) * 3 + ch] * w01;
    if (w10 > 0.0f) sum += (float)img[((size_t)y0 * w + x1) * 3 + ch] * w10;
    if (w11 > 0.0f) sum += (float)img[((size_t)y1 * w + x1) * 3 + ch] * w11;
    unsigned char out = 0;
    if (sw > 0.0f) out = (unsigned char)(sum / sw);
    return out;
}

}  // namespace i3d
