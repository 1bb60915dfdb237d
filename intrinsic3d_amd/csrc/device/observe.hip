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
#include "observe_device.hpp"

namespace i3d {

template <int SLOTS, bool KEEP_ALL>
__global__ void __launch_bounds__(256) k_observe(GridView g, RowView r, OptParams p, const FrameConst* __restrict__ frames, const unsigned* __restrict__ cull, int ncw, int prefilter) {
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
        const float len = sqrtf(nx * nx + (ny * ny + nz * nz));
        if (len != 0.0f) { nx /= len; ny /= len; nz /= len; }
    }
    // voxelCenterToIso (operators.cpp:44-55): voxelToWorld(v) - n * (float)sdf_refined
    const float px = (float)g.cx[s] * g.voxel_size - nx * s0;
    const float py = (float)g.cy[s] * g.voxel_size - ny * s0;
    const float pz = (float)g.cz[s] * g.voxel_size - nz * s0;

    float bw[SLOTS]; int bf[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) { bw[i] = 0.0f; bf[i] = -1; }

    // keyframes that cannot observe ANY entry of this wave's group of 64 (cull_kernels.hip, conservative) are skipped: a wave-uniform mask word per 32 keyframes
    const unsigned* __restrict__ wmask = cull ? cull + (size_t)__builtin_amdgcn_readfirstlane(ci >> 6) * ncw : nullptr;
    const int K = p.K;
    unsigned skip = 0;
    for (int f = 0; f < K; ++f) {
        if ((f & 31) == 0) skip = wmask ? wmask[f >> 5] : 0u;
        if ((skip >> (f & 31)) & 1u) continue;
        const FrameConst& fc = frames[f];
        if (!KEEP_ALL && prefilter) {
            // Upper bound of the weight this keyframe can give the voxel, BEFORE the projection, the depth fetch and the correctly rounded divisions: the weight is
            // computeWeight's normal term alone (its depth term max(1 - dn, 1) is exactly 1 for every valid depth, colorization.cpp:296-312), a decreasing function of
            // 1 - |cos(view, normal)|.  The same q and R n as the exact evaluation (the compiler shares them), a 1-ulp reciprocal square root instead of sqrt + three
            // divisions, 2e-5 absolute slack on the cosine and 1e-4 relative on the result — orders above the round-off of either form.  A keyframe whose bound does
            // not beat the weakest of the kept observations cannot enter the list (insertion needs w > bw[0]): skipped without touching its depth image.
            const float qx = ((fc.Rf[0] * px + fc.Rf[1] * py) + fc.Rf[2] * pz) + fc.tf[0];
            const float qy = ((fc.Rf[3] * px + fc.Rf[4] * py) + fc.Rf[5] * pz) + fc.tf[1];
            const float qz = ((fc.Rf[6] * px + fc.Rf[7] * py) + fc.Rf[8] * pz) + fc.tf[2];
            const float cnx = (fc.Rf[0] * nx + fc.Rf[1] * ny) + fc.Rf[2] * nz;
            const float cny = (fc.Rf[3] * nx + fc.Rf[4] * ny) + fc.Rf[5] * nz;
            const float cnz = (fc.Rf[6] * nx + fc.Rf[7] * ny) + fc.Rf[8] * nz;
            const float vsq = qx * qx + (qy * qy + qz * qz);
            const float cosv = fabsf(qx * cnx + (qy * cny + qz * cnz)) * __builtin_amdgcn_rsqf(vsq);
            const float t = fmaxf((1.0f - cosv) - 2e-5f, 0.0f);           // NaN (q = 0) -> 0: the bound becomes 1, nothing is skipped
            const float div = 1.0f + 2.0f * t;
            const float ub = fmaxf(__builtin_amdgcn_rcpf(div * div * div), 0.001f) * 1.0001f;
            if (!(ub > bw[0])) continue;
        }
        float uf, vf;
        const float w = observation_weight(fc, p, px, py, pz, nx, ny, nz, fc.depth, uf, vf);
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
    // Slots are stored in ascending KEYFRAME order (empty slots last).  The reference creates a voxel's rows in ascending weight
    // order, which only permutes terms of sums; with frame order, neighbouring voxels — which mostly pick the same keyframes —
    // line the same keyframe up in the same slot, so a wave samples one image region and accumulates into one camera block.
    if (!KEEP_ALL) {
#pragma unroll
        for (int pass = 0; pass < SLOTS - 1; ++pass)
#pragma unroll
            for (int i = 0; i + 1 < SLOTS - pass; ++i)
                if ((unsigned)bf[i] > (unsigned)bf[i + 1]) { const float tw = bw[i]; bw[i] = bw[i + 1]; bw[i + 1] = tw; const int tf = bf[i]; bf[i] = bf[i + 1]; bf[i + 1] = tf; }
    }
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        r.obs_frame[(size_t)i * r.Acap + a] = bf[i];
        r.obs_w[(size_t)i * r.Acap + a] = bw[i];
    }
}

template <int S> static void launch_s(hipStream_t st, GridView g, RowView r, OptParams p, const FrameConst* fr, bool keep_all, const unsigned* cull, int prefilter) {
    const int blocks = (r.nC + 255) / 256, ncw = (p.K + 31) / 32;
    if (keep_all) k_observe<S, true><<<blocks, 256, 0, st>>>(g, r, p, fr, cull, ncw, prefilter);
    else k_observe<S, false><<<blocks, 256, 0, st>>>(g, r, p, fr, cull, ncw, prefilter);
}

void launch_observe(hipStream_t st, GridView g, RowView r, OptParams p, const FrameConst* frames, const unsigned* cull_mask, bool prefilter) {
    if (r.nC <= 0) return;
    const bool keep_all = r.slots >= p.K;
    switch (r.slots) {
        case 1: launch_s<1>(st, g, r, p, frames, keep_all, cull_mask, prefilter ? 1 : 0); break;
        case 2: launch_s<2>(st, g, r, p, frames, keep_all, cull_mask, prefilter ? 1 : 0); break;
        case 3: launch_s<3>(st, g, r, p, frames, keep_all, cull_mask, prefilter ? 1 : 0); break;
        case 4: launch_s<4>(st, g, r, p, frames, keep_all, cull_mask, prefilter ? 1 : 0); break;
        case 5: launch_s<5>(st, g, r, p, frames, keep_all, cull_mask, prefilter ? 1 : 0); break;
        case 6: launch_s<6>(st, g, r, p, frames, keep_all, cull_mask, prefilter ? 1 : 0); break;
        case 7: launch_s<7>(st, g, r, p, frames, keep_all, cull_mask, prefilter ? 1 : 0); break;
        default: launch_s<8>(st, g, r, p, frames, keep_all, cull_mask, prefilter ? 1 : 0); break;
    }
}

}  // namespace i3d
