// extern "C" entry points that are not pure context plumbing: optimize, one-shot host drop-in, parity probes.
#include <cstdlib>
#include "context.hpp"

using namespace i3d;

extern "C" {

int i3d_optimize(i3d_context* c, const i3d_optimizer_config* cfg, i3d_iteration_stats* stats) {
    if (!c || !cfg) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_optimize: null argument");
    return optimize(c, *cfg, stats);
}

int i3d_optimize_host(int32_t device_ordinal, const i3d_optimizer_config* cfg, const i3d_grid_view* grid,
                      double* sdf_refined_io, double* albedo_io, int32_t num_frames, int32_t levels, const int32_t* widths,
                      const int32_t* heights, const float* const* lum, const float* const* depth,
                      double* intr_io, double* dist_io, double* poses_io, const double* voxel_sh, i3d_iteration_stats* stats) {
    if (!cfg || !grid || !sdf_refined_io || !albedo_io || !intr_io || !dist_io || !poses_io || !voxel_sh) return I3D_ERR_INVALID_ARGUMENT;
    i3d_context* c = nullptr;
    int rc = i3d_create(device_ordinal, &c); if (rc) return rc;
    i3d_grid_view gv = *grid; gv.sdf_refined = sdf_refined_io; gv.albedo = albedo_io;
    rc = i3d_set_grid(c, &gv);
    if (!rc) rc = i3d_set_frames(c, num_frames, levels, widths, heights, lum, depth, nullptr);
    if (!rc) rc = i3d_set_camera(c, intr_io, dist_io, poses_io);
    if (!rc) rc = i3d_set_voxel_sh(c, voxel_sh);
    if (!rc) rc = i3d_optimize(c, cfg, stats);
    if (!rc) rc = i3d_get_grid(c, sdf_refined_io, albedo_io);
    if (!rc) rc = i3d_get_camera(c, intr_io, dist_io, poses_io);
    if (rc) std::fprintf(stderr, "i3d_optimize_host: %s\n", i3d_last_error(c));
    i3d_destroy(c);
    return rc;
}

int i3d_comm_unique_id(void* out128, int32_t* bytes) {
    if (!out128 || !bytes) return I3D_ERR_INVALID_ARGUMENT;
    size_t n = 0; if (rccl_unique_id(out128, &n)) return I3D_ERR_COMM;
    *bytes = (int32_t)n; return I3D_OK;
}
int i3d_comm_init(i3d_context* c, int32_t rank, int32_t world, const void* unique_id, int32_t id_bytes) {
    if (!c || !unique_id || world < 1 || rank < 0 || rank >= world) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_comm_init: bad arguments");
    CTX_HIP(c, hipSetDevice(c->device));
    char err[256] = {0};
    Comm* cm = make_rccl_comm(rank, world, unique_id, (size_t)id_bytes, c->stream, err, sizeof(err));
    if (!cm) return ctx_fail(c, I3D_ERR_COMM, err);
    { const char* e = std::getenv("I3D_FORCE_COLLECTIVES"); cm->force = e && e[0] == '1'; }      // test hook: sharded path with a 1-rank communicator
    delete c->comm; c->comm = cm; c->assembled = false; c->slots = 0;      // solver vectors are sized for the world: re-derive storage
    return I3D_OK;
}
void* i3d_comm_sim_create(int32_t world) { return world >= 1 ? sim_create(world) : nullptr; }
void i3d_comm_sim_destroy(void* shared) { if (shared) sim_destroy((SimShared*)shared); }
int i3d_comm_init_sim(i3d_context* c, void* shared, int32_t rank) {
    if (!c || !shared) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_comm_init_sim: bad arguments");
    delete c->comm; c->comm = make_sim_comm((SimShared*)shared, rank); c->assembled = false; c->slots = 0;
    return I3D_OK;
}

int i3d_shard_plan(int32_t A, int32_t world, int32_t rank, const int32_t* anbr, const uint8_t* active, int32_t* chunk, int32_t* own0,
                   int32_t* own1, uint8_t* in_compute_list) {
    if (A < 0 || world < 1 || rank < 0 || rank >= world || !chunk || !own0 || !own1) return I3D_ERR_INVALID_ARGUMENT;
    int ch, o0, o1; shard_range(A, world, rank, ch, o0, o1);
    *chunk = ch; *own0 = o0; *own1 = o1;
    if (in_compute_list && anbr && active)
        for (int a = 0; a < A; ++a) in_compute_list[a] = shard_needs_entry(a, o0, o1, active[a] != 0, anbr, (size_t)A) ? 1 : 0;
    return I3D_OK;
}
int32_t i3d_shard_vec_index(int32_t a, int32_t chunk, int32_t albedo) { return albedo ? vec_alb(a, chunk) : vec_sdf(a, chunk); }
// need[e] bit k: rank k's rows read the unknowns of entry e and it does not own e — the host statement of k_need_mask (shard_kernels.hip)
int i3d_shard_need(int32_t A, int32_t world, const int32_t* anbr, const uint8_t* active, uint64_t* need) {
    if (A < 0 || world < 1 || world > 64 || !anbr || !active || !need) return I3D_ERR_INVALID_ARGUMENT;
    int chunk, o0, o1; shard_range(A, world, 0, chunk, o0, o1);
    const int slice = chunk / world;
    for (int a = 0; a < A; ++a) need[a] = 0;
    for (int a = 0; a < A; ++a) {
        if (!active[a]) continue;
        int col[12]; bool interior;
        const unsigned long long ranks = shard_entry_ranks(a, slice, anbr, (size_t)A, col, interior);
        if (interior) continue;
        for (int k = 0; k < world; ++k) if ((ranks >> k) & 1ull) {
            if (a / slice != k) need[a] |= 1ull << k;
            for (int i = 0; i < 12; ++i) if (col[i] >= 0 && col[i] / slice != k) need[col[i]] |= 1ull << k;
        }
    }
    return I3D_OK;
}
int i3d_comm_stats(i3d_context* c, int64_t* halo_calls, int64_t* halo_bytes_sent, int64_t* reduce_calls, int64_t* reduce_bytes, int32_t* halo_entries_send, int32_t* halo_entries_recv,
                   int32_t* ghost_tiles, int32_t* compute_list) {
    if (!c) return I3D_ERR_INVALID_ARGUMENT;
    if (halo_calls) *halo_calls = c->comm ? c->comm->halo_calls : 0; if (halo_bytes_sent) *halo_bytes_sent = c->comm ? c->comm->halo_bytes_sent : 0;
    if (reduce_calls) *reduce_calls = c->comm ? c->comm->reduce_calls : 0; if (reduce_bytes) *reduce_bytes = c->comm ? c->comm->reduce_bytes : 0;
    if (halo_entries_send) *halo_entries_send = c->halo.n_send; if (halo_entries_recv) *halo_entries_recv = c->halo.n_recv;
    if (ghost_tiles) *ghost_tiles = c->n_ghost_tiles; if (compute_list) *compute_list = c->nC;
    return I3D_OK;
}

const char* i3d_comm_transport(i3d_context* c) { return (c && c->comm) ? c->comm->transport : ""; }

int i3d_debug_assemble(i3d_context* c, const i3d_optimizer_config* cfg, int32_t iteration, int32_t* slots_out) {
    if (!c || !cfg) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_debug_assemble: null argument");
    CTX_HIP(c, hipSetDevice(c->device));
    OptParams p; i3d_iteration_stats st; std::memset(&st, 0, sizeof(st));
    int rc = assemble(c, *cfg, iteration, p, &st);
    if (!rc && slots_out) *slots_out = c->slots;
    return rc;
}

int i3d_debug_flags(i3d_context* c, uint8_t* flags) {
    if (!c || !flags || !c->have_grid) return ctx_fail(c, I3D_ERR_STATE, "i3d_debug_flags: no grid");
    const int N = c->N; std::vector<uint8_t> f(N); std::vector<int> rank(N);
    CTX_HIP(c, hipMemcpy(f.data(), c->flags.p, N, hipMemcpyDeviceToHost));
    CTX_HIP(c, hipMemcpy(rank.data(), c->rank.p, sizeof(int) * (size_t)N, hipMemcpyDeviceToHost));
    for (int s = 0; s < N; ++s) flags[rank[s]] = f[s];
    return I3D_OK;
}

int i3d_debug_eg_rows(i3d_context* c, int32_t* frame, float* weight, float* residual, float* jac) {
    if (!c || !c->assembled) return ctx_fail(c, I3D_ERR_STATE, "i3d_debug_eg_rows: not assembled");
    const int N = c->N, A = c->A, S = c->slots; const size_t Acap = c->Acap;
    const size_t nrow = ((Acap + 63) / 64) * 64 * (size_t)S;
    std::vector<int> rank(N), alist(A > 0 ? A : 1); std::vector<float4> rows(nrow / 64 * ROW_BLOCK_F4); std::vector<float2> wr(nrow); std::vector<uint8_t> nr(Acap);
    CTX_HIP(c, hipMemcpy(rank.data(), c->rank.p, sizeof(int) * (size_t)N, hipMemcpyDeviceToHost));
    if (A > 0) CTX_HIP(c, hipMemcpy(alist.data(), c->alist.p, sizeof(int) * (size_t)A, hipMemcpyDeviceToHost));
    CTX_HIP(c, hipMemcpy(rows.data(), c->rows.p, sizeof(float4) * rows.size(), hipMemcpyDeviceToHost));
    const float2* const jt = reinterpret_cast<const float2*>(rows.data());
    CTX_HIP(c, hipMemcpy(wr.data(), c->row_wr.p, sizeof(float2) * nrow, hipMemcpyDeviceToHost));
    CTX_HIP(c, hipMemcpy(nr.data(), c->nrows.p, Acap, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < (size_t)N * S; ++i) { if (frame) frame[i] = -1; if (weight) weight[i] = 0.0f; if (residual) residual[i] = 0.0f; }
    if (jac) std::memset(jac, 0, sizeof(float) * (size_t)N * S * P_TOTAL);
    const float tw = (float)c->last_params.type_w[0];
    std::vector<uint8_t> afl(Acap); CTX_HIP(c, hipMemcpy(afl.data(), c->aflags.p, Acap, hipMemcpyDeviceToHost));
    for (int a = 0; a < A; ++a) {
        if (!(afl[a] & F_ACTIVE)) continue;
        const int v = rank[alist[a]];
        for (int k = 0; k < (int)nr[a]; ++k) {
            const float2 m = wr[row_scalar_index(a, k, S)];
            if (m.x == 0.0f) continue;
            const size_t o = (size_t)v * S + k;
            if (frame) { int f; std::memcpy(&f, &jt[row_jt_index(a, k, S)].y, sizeof(int)); frame[o] = f & ~ROW_FREE_BIT; }
            if (weight) weight[o] = m.x * tw;
            if (residual) residual[o] = m.y;
            if (jac) {      // stored with the row weight folded in (Js = sqrt(w) J): handed out as the raw partials
                const float isw = 1.0f / std::sqrt(m.x);
                for (int q = 0; q < 7; ++q) { const float4 t = rows[row_index(a, k, q, S)]; jac[o * P_TOTAL + 4 * q] = t.x * isw; jac[o * P_TOTAL + 4 * q + 1] = t.y * isw; jac[o * P_TOTAL + 4 * q + 2] = t.z * isw; jac[o * P_TOTAL + 4 * q + 3] = t.w * isw; }
                jac[o * P_TOTAL + 28] = jt[row_jt_index(a, k, S)].x * isw;
            }
        }
    }
    return I3D_OK;
}

int i3d_debug_reg_rows(i3d_context* c, uint8_t* has_er, uint8_t* has_es, float* ea_weight) {
    if (!c || !c->assembled) return ctx_fail(c, I3D_ERR_STATE, "i3d_debug_reg_rows: not assembled");
    const int N = c->N, A = c->A; const size_t Acap = c->Acap;
    std::vector<int> rank(N), alist(A > 0 ? A : 1); std::vector<uint8_t> rf(Acap), afl(Acap); std::vector<float> ew(Acap * 6);
    CTX_HIP(c, hipMemcpy(rank.data(), c->rank.p, sizeof(int) * (size_t)N, hipMemcpyDeviceToHost));
    if (A > 0) CTX_HIP(c, hipMemcpy(alist.data(), c->alist.p, sizeof(int) * (size_t)A, hipMemcpyDeviceToHost));
    CTX_HIP(c, hipMemcpy(rf.data(), c->regflags.p, Acap, hipMemcpyDeviceToHost));
    CTX_HIP(c, hipMemcpy(afl.data(), c->aflags.p, Acap, hipMemcpyDeviceToHost));
    CTX_HIP(c, hipMemcpy(ew.data(), c->ea_w.p, sizeof(float) * Acap * 6, hipMemcpyDeviceToHost));
    for (int v = 0; v < N; ++v) { if (has_er) has_er[v] = 0; if (has_es) has_es[v] = 0; if (ea_weight) for (int d = 0; d < 6; ++d) ea_weight[(size_t)v * 6 + d] = 0.0f; }
    const float tw = (float)c->last_params.type_w[3];
    for (int a = 0; a < A; ++a) {
        if (!(afl[a] & F_ACTIVE)) continue;
        const int v = rank[alist[a]];
        if (has_er) has_er[v] = rf[a] & 1;
        if (has_es) has_es[v] = (rf[a] >> 1) & 1;
        if (ea_weight) for (int d = 0; d < 6; ++d) ea_weight[(size_t)v * 6 + d] = ew[(size_t)d * Acap + a] * tw;
    }
    return I3D_OK;
}

int i3d_debug_neighbors(i3d_context* c, int32_t* nbr) {
    if (!c || !nbr || !c->have_grid) return ctx_fail(c, I3D_ERR_STATE, "i3d_debug_neighbors: no grid");
    const int N = c->N; std::vector<int> rank(N), t((size_t)NUM_NBR * N);
    CTX_HIP(c, hipMemcpy(rank.data(), c->rank.p, sizeof(int) * (size_t)N, hipMemcpyDeviceToHost));
    CTX_HIP(c, hipMemcpy(t.data(), c->nbr.p, sizeof(int) * (size_t)NUM_NBR * N, hipMemcpyDeviceToHost));
    for (int s = 0; s < N; ++s) for (int i = 0; i < NUM_NBR; ++i) { const int n = t[(size_t)i * N + s]; nbr[(size_t)rank[s] * NUM_NBR + i] = n < 0 ? -1 : rank[n]; }
    return I3D_OK;
}

int i3d_debug_normal_eq(i3d_context* c, double* gradient, double* jtj_diag, double* cost) {
    if (!c) return I3D_ERR_INVALID_ARGUMENT;
    return normal_eq_debug(c, gradient, jtj_diag, cost);
}
int i3d_debug_jtj_apply(i3d_context* c, const double* x, double* y) {
    if (!c || !x || !y) return ctx_fail(c, I3D_ERR_INVALID_ARGUMENT, "i3d_debug_jtj_apply: null pointer");
    return jtj_apply_debug(c, x, y);
}
int i3d_debug_counters(i3d_context* c, int64_t* stream_syncs) {
    if (!c) return I3D_ERR_INVALID_ARGUMENT;
    if (stream_syncs) *stream_syncs = (int64_t)c->n_syncs;
    return I3D_OK;
}

int i3d_debug_ladder_stats(i3d_context* c, int64_t* out6) {
    if (!c || !out6) return I3D_ERR_INVALID_ARGUMENT;
    out6[0] = c->lad_batches; out6[1] = c->lad_streams; out6[2] = c->lad_system_passes; out6[3] = c->lad_resyncs; out6[4] = c->lad_wasted; out6[5] = c->ladder_max;
    return I3D_OK;
}

int i3d_debug_cull_stats(i3d_context* c, int64_t* pairs, int64_t* culled) {
    if (!c || !pairs || !culled) return I3D_ERR_INVALID_ARGUMENT;
    if (!c->cull_mask.p || c->nC <= 0) return ctx_fail(c, I3D_ERR_STATE, "i3d_debug_cull_stats: nothing assembled");
    const size_t ngroups = ((size_t)c->nC + 63) / 64, ncw = ((size_t)c->K + 31) / 32;
    *pairs = (int64_t)(ngroups * (size_t)c->K);
    if (!c->cull_on) { *culled = -1; return I3D_OK; }
    std::vector<unsigned> m(ngroups * ncw);
    CTX_HIP(c, hipSetDevice(c->device));
    CTX_HIP(c, hipMemcpy(m.data(), c->cull_mask.p, m.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    int64_t n = 0; for (unsigned w : m) n += __builtin_popcount(w);
    *culled = n;
    return I3D_OK;
}

}  // extern "C"
