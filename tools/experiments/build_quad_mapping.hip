// EXPERIMENT, not part of the product (not compiled by the Makefile): the quad-mapped variant of k_build, kept as evidence for the bound
// stated in DESIGN.md / profiles/README.md.  It was dropped into device/build.hip (same helpers: shared_point, project_point, bicubic_taps,
// bicubic_eval, chroma_weight, sdf_to_weight) and passed the GPU parity tests (tests/test_gpu_parity.py, tests/test_gpu_bench_parity.py).
// Measured on the bench workload (rocprofv3 kernel trace, MI355X):
//     lane per voxel (product)      k_build<true>  1444 us  (247 VGPRs, 2 waves/SIMD)      cost evaluation 587 us (4 waves/SIMD)
//     quad per voxel, 3 waves/SIMD  k_build_quad   2172 us  (168 VGPRs)                    cost evaluation 805 us (62 VGPRs, 5-6 waves/SIMD)
//     quad per voxel, 2 / 4 waves   2541 / 2530 us
// More waves did NOT help: the per-row work that cannot be split over the four lanes (keyframe constants, residual, sqrt, division, row
// bookkeeping, the DPP exchange and the per-lane selection of the eight outputs) is executed four times, the wave-instruction count per
// voxel rises 1.4x (cost variant) and the run time rises by the same factor — the kernels are VALU-issue-bound (about 5.8 cycles per VALU
// wave-instruction at 52 % fp64 in BOTH mappings, at 3 and at 6 waves per SIMD), neither latency- nor HBM-bound.
// ---- quad mapping: FOUR lanes per voxel, one per stencil point (000, 100, 010, 001) ----------------------------------------------------------
// The lane-per-voxel kernel above keeps the state of 4 points + 29 partials per lane: ~245 VGPRs, 2 waves per SIMD, and its long dependent
// fp64 chains leave the VALU idle 2/3 of the time (profiles/README.md: removing every gather and every store takes only 19 % off it).  Here a
// lane owns ONE point of its voxel: normal / iso-point / SH shading once per voxel, then per row its projection, bicubic taps and the partials
// of that point; the residual, the row validity and the 15 camera partials are combined across the quad with DPP (no LDS, no barrier).  Four
// times the lanes at a third of the registers: 4+ waves per SIMD hide the same chains.  A wave holds 16 consecutive work-list entries, so the
// row planes are still written in 256-byte runs; the per-keyframe constants of ALL keyframes sit in LDS (29 KB for 200 keyframes).
template <int CTRL> static __device__ inline int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL> static __device__ inline float dpp_f(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }
template <int CTRL> static __device__ inline double dpp_d(double v) { return __hiloint2double(dpp_i<CTRL>(__double2hiint(v)), dpp_i<CTRL>(__double2loint(v))); }
constexpr int Q_XOR1 = 0xB1, Q_XOR2 = 0x4E;                         // quad_perm [1,0,3,2], [2,3,0,1]
constexpr int Q_B0 = 0x00, Q_B1 = 0x55, Q_B2 = 0xAA, Q_B3 = 0xFF;   // broadcast lane i of the quad
static __device__ inline float quad_sum(float v) { v += dpp_f<Q_XOR1>(v); v += dpp_f<Q_XOR2>(v); return v; }      // same bits in all four lanes
static __device__ inline int quad_or(int v) { v |= dpp_i<Q_XOR1>(v); v |= dpp_i<Q_XOR2>(v); return v; }
static __device__ inline int quad_and(int v) { v &= dpp_i<Q_XOR1>(v); v &= dpp_i<Q_XOR2>(v); return v; }

constexpr int BUILD_WG = 256;          // 64 voxels per workgroup = one tile of the row storage

template <bool WITH_J, bool FR_LDS, int OCC>
__global__ void __launch_bounds__(BUILD_WG, OCC) k_build_quad(GridView g, RowView r, OptParams p, const FrameConst* __restrict__ frames, double* cost_out) {
    extern __shared__ double frame_lds_raw[];
    FrameHot* const flds = reinterpret_cast<FrameHot*>(frame_lds_raw);
    if (FR_LDS) {
        constexpr int WORDS = sizeof(FrameHot) / 8;
        for (int i = threadIdx.x; i < p.K * WORDS; i += blockDim.x) {
            const int f = i / WORDS, w = i - f * WORDS;
            frame_lds_raw[(size_t)f * WORDS + w] = reinterpret_cast<const double*>(&frames[f].hot)[w];
        }
        __syncthreads();
    }
    const int j = threadIdx.x & 3;                                    // stencil point of this lane
    const int ci = blockIdx.x * (BUILD_WG / 4) + (threadIdx.x >> 2);
    const int a = ci < r.nC ? (r.clist ? r.clist[ci] : ci) : -1;      // compute list of this rank (identity when not sharded)
    const bool owned = a >= r.own0 && a < r.own1;                     // cost / weight sums count every row once: on its owner
    double cost = 0.0;
    // every condition below that guards a DPP exchange is the same in the four lanes of a quad
    if (a >= 0 && (WITH_J || owned)) {
        const int N = g.N; const size_t Acap = r.Acap;
        const int s = r.alist[a];
        const uint8_t fl = r.aflags[a];
        if (!(fl & F_ACTIVE)) {                 // free-only entry: unknowns but no rows
            if (WITH_J && j == 0) {
                r.regflags[a] = 0; r.ea_free[a] = 0; r.nrows[a] = 0;
                for (int d = 0; d < 6; ++d) r.ea_w[(size_t)d * Acap + a] = 0.0f;
                for (int k = 0; k < r.slots; ++k) r.rows[row_index(a, k, 7, r.slots)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        } else {
        const double xs = g.x_sdf[s];
        // ---- regulariser rows (optimizer.cpp:238-276): lane 0 of the quad -----------------------------------
        if (j == 0) {
            int ring[6];
#pragma unroll
            for (int d = 0; d < 6; ++d) ring[d] = g.nbr[(size_t)d * N + s];
            const bool ring_ok = (fl & F_RING) != 0;
            if (WITH_J) {
                uint8_t rf = 0;
                if (p.use_er && ring_ok) {
                    rf |= 1;
                    bool fr = (fl & F_FREE_SDF) != 0;
#pragma unroll
                    for (int d = 0; d < 6; ++d) fr |= (g.flags[ring[d]] & F_FREE_SDF) != 0;
                    if (fr) rf |= 8;
                }
                if (p.use_es) { rf |= 2; if ((xs - g.sdf0[s]) != 0.0) rf |= 4; if (fl & F_FREE_SDF) rf |= 16; }
                uint8_t eafree = 0;
                const uchar4 col = g.color[s];
                const int myrank = g.rank[s];
#pragma unroll
                for (int d = 0; d < 6; ++d) {
                    float w = 0.0f;
                    if (p.use_ea && ring_ok) {
                        const int nb = ring[d];
                        const bool added_before = (g.flags[nb] & F_ACTIVE) && g.rank[nb] < myrank;     // voxels_added, optimizer.cpp:267-279
                        if (!added_before) {
                            w = chroma_weight(col, g.color[nb]);
                            if (!(w == w) || isinf(w)) w = 0.0f;
                            if (w != 0.0f && ((fl & F_FREE_ALB) || (g.flags[nb] & F_FREE_ALB))) eafree |= (uint8_t)(1 << d);
                        }
                    }
                    r.ea_w[(size_t)d * Acap + a] = w;
                }
                r.regflags[a] = rf; r.ea_free[a] = eafree;
            } else {
                const uint8_t rf = r.regflags[a];
                // cost of the regulariser rows at this state (rows without a free parameter are not part of the reduced program)
                if ((rf & 1) && (rf & 8)) {
                    const double dxx = g.x_sdf[ring[0]] + g.x_sdf[ring[1]] - 2.0 * xs, dyy = g.x_sdf[ring[2]] + g.x_sdf[ring[3]] - 2.0 * xs,
                                 dzz = g.x_sdf[ring[4]] + g.x_sdf[ring[5]] - 2.0 * xs;
                    const double lap = dxx + dyy + dzz; cost += 0.5 * p.type_w[1] * lap * lap;
                }
                if ((rf & 2) && (rf & 16)) { double e = xs - g.sdf0[s]; if (e == 0.0) e = 0.0000001; cost += 0.5 * p.type_w[2] * e * e; }
                const uint8_t eafree = r.ea_free[a];
                const double xa = g.x_alb[s];
#pragma unroll
                for (int d = 0; d < 6; ++d) if (eafree & (1 << d)) {
                    const double e = xa - g.x_alb[ring[d]];
                    cost += 0.5 * (double)r.ea_w[(size_t)d * Acap + a] * p.type_w[3] * e * e;
                }
            }
        }

        // ---- Eg rows ---------------------------------------------------------------------------------------
        // the point's own stencil (s, +x, +y, +z of the point): forward-neighbour codes of (sdf slots 0 6 1 4 | 6 9 7 8 | 1 7 2 3 | 4 8 3 5), 0xFF = the voxel itself
        const unsigned codes = j == 0 ? ((unsigned)NB_PZ << 24 | (unsigned)NB_PY << 16 | (unsigned)NB_PX << 8 | 0xFFu)
                             : j == 1 ? ((unsigned)NB_PXZ << 24 | (unsigned)NB_PXY << 16 | (unsigned)NB_P2X << 8 | (unsigned)NB_PX)
                             : j == 2 ? ((unsigned)NB_PYZ << 24 | (unsigned)NB_P2Y << 16 | (unsigned)NB_PXY << 8 | (unsigned)NB_PY)
                                      : ((unsigned)NB_P2Z << 24 | (unsigned)NB_PYZ << 16 | (unsigned)NB_PXZ << 8 | (unsigned)NB_PZ);
        int idx[4];
        bool eligible = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const unsigned nb = (codes >> (8 * i)) & 0xFFu; idx[i] = nb == 0xFFu ? s : g.nbr[(size_t)nb * N + s]; eligible &= idx[i] >= 0; }
        eligible = quad_and(eligible ? 1 : 0) != 0;
        const int nin = WITH_J ? r.slots : (int)r.nrows[a];       // candidates: observation slots (assembly) or stored rows (cost)
        bool any_row = false;
        if (WITH_J) { for (int k = 0; k < r.slots; ++k) any_row |= r.obs_w[(size_t)k * Acap + a] > 0.0f; any_row &= eligible; }
        else any_row = nin > 0;
        int nout = 0;
        if (any_row) {
            float sh[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) sh[i] = g.sh[(size_t)i * N + s];
            PointShared q;
            shared_point(q, g.x_sdf[idx[0]], g.x_sdf[idx[1]], g.x_sdf[idx[2]], g.x_sdf[idx[3]], g.x_alb[idx[0]], sh,
                         g.cx[s] + (j == 1 ? 1 : 0), g.cy[s] + (j == 2 ? 1 : 0), g.cz[s] + (j == 3 ? 1 : 0), (double)g.voxel_size);
            bool vox_free = false;
            if (WITH_J) {
                int fr = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) fr |= (g.flags[idx[i]] & F_FREE_SDF);
                fr |= (g.flags[idx[0]] & F_FREE_ALB);
                vox_free = quad_or(fr) != 0 || !p.fix_poses || !p.fix_intr || !p.fix_dist;
            }
            const double weight_sdf = sdf_to_weight(xs, (double)g.truncation);
            const float psf = (float)p.pyr_scale;
            const float fxs = (float)(p.intr[0] * p.pyr_scale), fys = (float)(p.intr[1] * p.pyr_scale);
            const float k0 = (float)p.dist[0], k1 = (float)p.dist[1], k2 = (float)p.dist[2], k3 = (float)p.dist[3], k4 = (float)p.dist[4];

            for (int k = 0; k < nin; ++k) {
                const size_t ka = (size_t)k * Acap + a;
                float roww; int f;
                if (WITH_J) { const float ow = r.obs_w[ka]; f = r.obs_frame[ka]; roww = (ow > 0.0f) ? (float)((double)ow * weight_sdf) : 0.0f; }
                else { const float4 m = r.rows[row_index(a, k, 7, r.slots)]; const int fb = __float_as_int(m.z); roww = (fb & ROW_FREE_BIT) ? m.x : 0.0f; f = fb & ~ROW_FREE_BIT; }
                if (roww == 0.0f) continue;
                const FrameHot& fc = FR_LDS ? flds[f] : frames[f].hot;
                // ---- phase 1: the value of this lane's point (fp64) ----
                double pu, pw, lum = 0.0; PointVal pv;
                bool ok = quad_and(project_point<WITH_J>(q.P, fc.R, fc.t, p, pu, pw, pv) ? 1 : 0) != 0;      // a row with a point outside the image is dropped (cost.h:100-105)
                double res = 0.0; float c = 0.0f;
                if (ok) {
                    Taps tp;
                    bicubic_taps(fc.lum, p.w, p.h, pw, pu, tp);
                    bicubic_eval<WITH_J>(tp, lum, pv.dfdr, pv.dfdc);
                    // d_j = (B_j - B_0) - (lum_j - lum_0); every lane forms the residual from the same three numbers
                    const double d = (q.B - dpp_d<Q_B0>(q.B)) - (lum - dpp_d<Q_B0>(lum));
                    const double d1 = dpp_d<Q_B1>(d), d2 = dpp_d<Q_B2>(d), d3 = dpp_d<Q_B3>(d);
                    res = sqrt(d1 * d1 + d2 * d2 + d3 * d3);
                    if (!(res > 0.0) || isinf(res)) { ok = false; res = 0.0; }    // 0, NaN, inf -> NV_INVALID_RESIDUAL (shading_cost.h:186-195)
                    else { const double ir = 1.0 / res; c = j == 0 ? -(float)((d1 + d2 + d3) * ir) : (float)(d * ir); }
                }
                if (!WITH_J) { if (ok && j == 0) cost += 0.5 * (double)roww * p.type_w[0] * res * res; continue; }
                if (!ok) continue;                                                 // dropped at creation (shading_cost.cpp:136-145)
                // ---- phase 2: the partials through this lane's point (fp32) ----
                const float R0 = (float)fc.R[0], R1 = (float)fc.R[1], R2 = (float)fc.R[2], R3 = (float)fc.R[3], R4 = (float)fc.R[4], R5 = (float)fc.R[5], R6 = (float)fc.R[6], R7 = (float)fc.R[7], R8 = (float)fc.R[8];
                const float x0 = pv.x0, y0 = pv.y0, iz = pv.iz;
                const float r2 = x0 * x0 + y0 * y0, r4 = r2 * r2, r6 = r4 * r2;
                const float dc = 1.0f + k0 * r2 + k1 * r4 + k2 * r6;
                const float dcr = k0 + 2.0f * k1 * r2 + 3.0f * k2 * r4;          // d dc / d r2
                const float xd = x0 * dc + 2.0f * k3 * x0 * y0 + k4 * (r2 + 2.0f * x0 * x0);
                const float yd = y0 * dc + 2.0f * k4 * xd * y0 + k3 * (r2 + 2.0f * y0 * y0);
                const float dxd_dx0 = dc + 2.0f * x0 * x0 * dcr + 2.0f * k3 * y0 + 6.0f * k4 * x0;
                const float dxd_dy0 = 2.0f * x0 * y0 * dcr + 2.0f * k3 * x0 + 2.0f * k4 * y0;
                const float dyd_dx0 = 2.0f * x0 * y0 * dcr + 2.0f * k4 * y0 * dxd_dx0 + 2.0f * k3 * x0;
                const float dyd_dy0 = dc + 2.0f * y0 * y0 * dcr + 2.0f * k4 * (xd + y0 * dxd_dy0) + 6.0f * k3 * y0;
                const float au = pv.dfdc * fxs, av = pv.dfdr * fys;
                const float lx = au * dxd_dx0 + av * dyd_dx0, ly = au * dxd_dy0 + av * dyd_dy0;
                const float L0 = lx * iz, L1 = ly * iz, L2 = -(lx * x0 + ly * y0) * iz;      // d lum / d Q
                const float M0 = L0 * R0 + L1 * R3 + L2 * R6, M1 = L0 * R1 + L1 * R4 + L2 * R7, M2 = L0 * R2 + L1 * R5 + L2 * R8;   // d lum / d P
                // E_j = B_j - lum_j;  dE/dg = alb * N dLs + s * N M,  dE/ds (direct) = M . n
                const float v[3] = {q.alb * q.dLs[0] + q.s * M0, q.alb * q.dLs[1] + q.s * M1, q.alb * q.dLs[2] + q.s * M2};
                float G[3]; apply_normal_jac(q, v, G);
                const float direct = M0 * q.n[0] + M1 * q.n[1] + M2 * q.n[2];
                // the point's four sdf columns (its own voxel, +x, +y, +z) and its albedo column
                const float e0 = c * (direct - (G[0] + G[1] + G[2])), e1 = c * G[0], e2 = c * G[1], e3 = c * G[2];
                const float ealb = c * q.Ls;
                // camera columns: sums over the four points
                const float Px = (float)q.P[0], Py = (float)q.P[1], Pz = (float)q.P[2];
                const float Wx = quad_sum(c * (Py * M2 - Pz * M1)), Wy = quad_sum(c * (Pz * M0 - Px * M2)), Wz = quad_sum(c * (Px * M1 - Py * M0));   // rotation part before Jr
                const float T0 = quad_sum(-(c * L0)), T1 = quad_sum(-(c * L1)), T2 = quad_sum(-(c * L2));                    // d lum / d t = L
                const float I0 = quad_sum(-(c * pv.dfdc * psf * xd)), I1 = quad_sum(-(c * pv.dfdr * psf * yd)), I2 = quad_sum(-(c * pv.dfdc * psf)), I3 = quad_sum(-(c * pv.dfdr * psf));
                const float dxk0 = x0 * r2, dxk1 = x0 * r4, dxk2 = x0 * r6, dxk3 = 2.0f * x0 * y0, dxk4 = r2 + 2.0f * x0 * x0;
                const float c2 = 2.0f * k4 * y0;
                const float D0 = quad_sum(-(c * (au * dxk0 + av * (y0 * r2 + c2 * dxk0))));
                const float D1 = quad_sum(-(c * (au * dxk1 + av * (y0 * r4 + c2 * dxk1))));
                const float D2 = quad_sum(-(c * (au * dxk2 + av * (y0 * r6 + c2 * dxk2))));
                const float D3 = quad_sum(-(c * (au * dxk3 + av * (c2 * dxk3 + (r2 + 2.0f * y0 * y0)))));
                const float D4 = quad_sum(-(c * (au * dxk4 + av * (2.0f * xd * y0 + c2 * dxk4))));
                // ---- the 32 floats of the row, 8 per lane: lane j writes planes 2j and 2j+1 ----
                // sdf column of slot c = sum over the points whose stencil holds it, in ascending point order:
                //   0: p0.e0 | 1: p0.e2 + p2.e0 | 2: p2.e2 | 3: p2.e3 + p3.e2 | 4: p0.e3 + p3.e0 | 5: p3.e3 | 6: p0.e1 + p1.e0 | 7: p1.e2 + p2.e1 | 8: p1.e3 + p3.e1 | 9: p1.e1
                // (the exchanges are executed by the whole quad; each lane then keeps its eight)
                float o[8];
                const float b00 = dpp_f<Q_B0>(e0), b01 = dpp_f<Q_B0>(e1), b02 = dpp_f<Q_B0>(e2), b03 = dpp_f<Q_B0>(e3);
                const float b10 = dpp_f<Q_B1>(e0), b11 = dpp_f<Q_B1>(e1), b12 = dpp_f<Q_B1>(e2), b13 = dpp_f<Q_B1>(e3);
                const float b20 = dpp_f<Q_B2>(e0), b21 = dpp_f<Q_B2>(e1), b22 = dpp_f<Q_B2>(e2), b23 = dpp_f<Q_B2>(e3);
                const float b30 = dpp_f<Q_B3>(e0), b31 = dpp_f<Q_B3>(e1), b32 = dpp_f<Q_B3>(e2), b33 = dpp_f<Q_B3>(e3);
                const float a0 = dpp_f<Q_B0>(ealb), a1 = dpp_f<Q_B1>(ealb), a2 = dpp_f<Q_B2>(ealb), a3 = dpp_f<Q_B3>(ealb);
                const float W0 = -(Wx * fc.Jr[0] + Wy * fc.Jr[3] + Wz * fc.Jr[6]), W1 = -(Wx * fc.Jr[1] + Wy * fc.Jr[4] + Wz * fc.Jr[7]), W2 = -(Wx * fc.Jr[2] + Wy * fc.Jr[5] + Wz * fc.Jr[8]);
                const int fbits = f | (vox_free ? ROW_FREE_BIT : 0);
                if (j == 0)      { o[0] = b00; o[1] = b02 + b20; o[2] = b22; o[3] = b23 + b32; o[4] = b03 + b30; o[5] = b33; o[6] = b01 + b10; o[7] = b12 + b21; }
                else if (j == 1) { o[0] = b13 + b31; o[1] = b11; o[2] = a0; o[3] = a1; o[4] = a2; o[5] = a3; o[6] = W0; o[7] = W1; }
                else if (j == 2) { o[0] = W2; o[1] = T0; o[2] = T1; o[3] = T2; o[4] = I0; o[5] = I1; o[6] = I2; o[7] = I3; }
                else             { o[0] = D0; o[1] = D1; o[2] = D2; o[3] = D3; o[4] = roww; o[5] = (float)res; o[6] = __int_as_float(fbits); o[7] = D4; }
                bool fin = true;                  // every partial finite (lane 3: o[4..6] are the row record, not partials)
#pragma unroll
                for (int i = 0; i < 8; ++i) fin = fin && ((j == 3 && i >= 4 && i <= 6) || !(isnan(o[i]) || isinf(o[i])));
                if (!quad_and(fin ? 1 : 0)) continue;
                // rows of a voxel are compacted into its first slots (creation order = ascending observation weight)
                r.rows[row_index(a, nout, 2 * j, r.slots)] = make_float4(o[0], o[1], o[2], o[3]);
                r.rows[row_index(a, nout, 2 * j + 1, r.slots)] = make_float4(o[4], o[5], o[6], o[7]);
                ++nout;
            }
        }
        if (WITH_J && j == 0) {
            r.nrows[a] = (uint8_t)nout;
            for (int k = nout; k < r.slots; ++k) r.rows[row_index(a, k, 7, r.slots)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        }
    }
    if (!WITH_J) block_partial_d(cost, cost_out, 1, 0);          // per-workgroup partial (no same-address atomics), summed by k_reduce_partials
}


