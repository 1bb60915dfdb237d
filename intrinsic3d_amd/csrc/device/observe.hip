// K1 — observation pass.  One lane per active voxel loops over the K keyframes, reproduces the float
// arithmetic of SDFColorization::computeObservation / isVoxelVisible / computeWeight (sdf/colorization.cpp:215-315)
// and Camera::project (camera.cpp:124-154) operation for operation, and keeps the best `slots` observations by
// weight in registers (SDFColorization::filter, colorization.cpp:357-370).
//
// The pass takes DISCRETE decisions (pixel rounding, visibility, top-n membership), so this file is compiled with
// -ffp-contract=off: no FMA may be formed that the reference's -O3 x86-64 build does not form.  The per-keyframe
// float rotation/translation (math::poseVecAAToMat(...).cast<float>(), math.cpp:151-163) is built on the host in
// fp64 with the same libm the reference uses and arrives in FrameConst::Rf/tf; every lane of a wave reads the same
// keyframe at the same time, so those loads are wave-uniform (scalar) loads.
#include "kernels.hpp"

namespace i3d {

template <int SLOTS, bool KEEP_ALL>
__global__ void __launch_bounds__(256) k_observe(GridView g, RowView r, OptParams p, const FrameConst* __restrict__ frames) {
    const int ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= r.nC) return;
    const int a = r.clist ? r.clist[ci] : ci;           // compute list of this rank (identity when not sharded)
    const int N = g.N;
    const int s = r.alist[a];
    if (!(r.aflags[a] & F_ACTIVE)) {          // free-only entry of the work list: owns unknowns but no rows
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) { r.obs_frame[(size_t)i * r.Acap + a] = -1; r.obs_w[(size_t)i * r.Acap + a] = 0.0f; }
        return;
    }
    // surface normal exactly as operators.cpp:58-77
    const float s0 = g.f_sdf[s];
    float nx = g.f_sdf[g.nbr[(size_t)NB_PX * N + s]] - s0;
    float ny = g.f_sdf[g.nbr[(size_t)NB_PY * N + s]] - s0;
    float nz = g.f_sdf[g.nbr[(size_t)NB_PZ * N + s]] - s0;
    {
        const float len = sqrtf(nx * nx + ny * ny + nz * nz);
        if (len != 0.0f) { nx /= len; ny /= len; nz /= len; }
    }
    // voxelCenterToIso (operators.cpp:44-55): voxelToWorld(v) - n * (float)sdf_refined
    const float px = (float)g.cx[s] * g.voxel_size - nx * s0;
    const float py = (float)g.cy[s] * g.voxel_size - ny * s0;
    const float pz = (float)g.cz[s] * g.voxel_size - nz * s0;

    float bw[SLOTS]; int bf[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) { bw[i] = 0.0f; bf[i] = -1; }

    const int K = p.K;
    for (int f = 0; f < K; ++f) {
        const FrameConst& fc = frames[f];
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
        const float u = p.cam_f[0] * x + p.cam_f[2], v = p.cam_f[1] * y + p.cam_f[3];
        const int ui = (int)(u + 0.5f), vi = (int)(v + 0.5f);
        float w = 0.0f;
        if (!(ui < 0 || ui >= p.w || vi < 0 || vi >= p.h)) {
            const float d = fc.depth[(size_t)vi * fc.w + ui];
            bool vis = true;
            if (p.occlusion > 0.0f) vis = (d > 0.0f) && (fabsf(d - qz) <= p.occlusion);
            if (vis && d > 0.0f) {
                const float cnx = (fc.Rf[0] * nx + fc.Rf[1] * ny) + fc.Rf[2] * nz;
                const float cny = (fc.Rf[3] * nx + fc.Rf[4] * ny) + fc.Rf[5] * nz;
                const float cnz = (fc.Rf[6] * nx + fc.Rf[7] * ny) + fc.Rf[8] * nz;
                float wn = 0.0f;
                if (!(fabsf(cnx) <= 1e-5f && fabsf(cny) <= 1e-5f && fabsf(cnz) <= 1e-5f)) {
                    const float vsq = qx * qx + qy * qy + qz * qz;
                    float vx = qx, vy = qy, vz = qz;
                    if (vsq > 0.0f) { const float l = sqrtf(vsq); vx /= l; vy /= l; vz /= l; }
                    wn = 1.0f - fabsf((vx * cnx + vy * cny) + vz * cnz);
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
        if (KEEP_ALL) {                 // n >= #frames: filter() returns before sorting, rows stay in frame order
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) if (i == f) { bw[i] = (w > 0.0f) ? w : 0.0f; bf[i] = (w > 0.0f) ? f : -1; }
        } else if (w > bw[0]) {         // ascending list of the `slots` largest weights
            bw[0] = w; bf[0] = f;
#pragma unroll
            for (int i = 0; i + 1 < SLOTS; ++i)
                if (bw[i] > bw[i + 1]) { const float tw = bw[i]; bw[i] = bw[i + 1]; bw[i + 1] = tw; const int tf = bf[i]; bf[i] = bf[i + 1]; bf[i + 1] = tf; }
        }
    }
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        r.obs_frame[(size_t)i * r.Acap + a] = bf[i];
        r.obs_w[(size_t)i * r.Acap + a] = bw[i];
    }
}

template <int S> static void launch_s(hipStream_t st, GridView g, RowView r, OptParams p, const FrameConst* fr, bool keep_all) {
    const int blocks = (r.nC + 255) / 256;
    if (keep_all) k_observe<S, true><<<blocks, 256, 0, st>>>(g, r, p, fr);
    else k_observe<S, false><<<blocks, 256, 0, st>>>(g, r, p, fr);
}

void launch_observe(hipStream_t st, GridView g, RowView r, OptParams p, const FrameConst* frames) {
    if (r.nC <= 0) return;
    const bool keep_all = r.slots >= p.K;
    switch (r.slots) {
        case 1: launch_s<1>(st, g, r, p, frames, keep_all); break;
        case 2: launch_s<2>(st, g, r, p, frames, keep_all); break;
        case 3: launch_s<3>(st, g, r, p, frames, keep_all); break;
        case 4: launch_s<4>(st, g, r, p, frames, keep_all); break;
        case 5: launch_s<5>(st, g, r, p, frames, keep_all); break;
        case 6: launch_s<6>(st, g, r, p, frames, keep_all); break;
        case 7: launch_s<7>(st, g, r, p, frames, keep_all); break;
        default: launch_s<8>(st, g, r, p, frames, keep_all); break;
    }
}

}  // namespace i3d
