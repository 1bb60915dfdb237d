// GPU box: the peer-to-peer mailbox transport (host/p2p.cpp) across PROCESSES, without RCCL: `world` child processes (device = rank mod
// device count, so on a one-GPU box they share the device and on a multi-GPU node every rank has its own), IPC handles exchanged through
// pipes, then `rounds` all-reduces (4 doubles and the full camera-block size) and rim pushes between every pair, every result checked.
//   tools/p2p_ipc_selftest [world = 2] [rounds = 200]        exit code 0 = every rank passed
// What it covers that the in-process rank simulation cannot: hipIpcGetMemHandle / hipIpcOpenMemHandle on the fine-grained mailbox, and
// stores into another process's mapping becoming visible to its polling kernel.
#include "../intrinsic3d_amd/csrc/host/p2p.hpp"
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace i3d;

static bool read_all(int fd, void* p, size_t n) { char* c = (char*)p; while (n) { const ssize_t r = read(fd, c, n); if (r <= 0) return false; c += r; n -= (size_t)r; } return true; }
static bool write_all(int fd, const void* p, size_t n) { const char* c = (const char*)p; while (n) { const ssize_t r = write(fd, c, n); if (r <= 0) return false; c += r; n -= (size_t)r; } return true; }

static int child(int rank, int world, int rounds, int to_parent, int from_parent) {
    int ndev = 0; if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::fprintf(stderr, "rank %d: no device\n", rank); return 2; }
    if (hipSetDevice(rank % ndev) != hipSuccess) return 2;
    hipStream_t st; if (hipStreamCreate(&st) != hipSuccess) return 2;
    const int RED = 6 * 200 + 9 + 1, B = 777;
    P2PEngine e;
    if (e.create(rank, world, RED, 1 << 12)) { std::fprintf(stderr, "rank %d: create failed\n", rank); return 3; }
    unsigned char mine[64] = {0};
    if (e.export_handle(mine)) { std::fprintf(stderr, "rank %d: hipIpcGetMemHandle failed on the fine-grained mailbox\n", rank); return 4; }
    std::vector<unsigned char> all(64 * (size_t)world);
    if (!write_all(to_parent, mine, 64) || !read_all(from_parent, all.data(), all.size())) return 5;
    if (e.attach_ipc(all.data())) { std::fprintf(stderr, "rank %d: hipIpcOpenMemHandle failed\n", rank); return 6; }
    char tok = 1; if (!write_all(to_parent, &tok, 1) || !read_all(from_parent, &tok, 1)) return 5;      // everyone has mapped everyone
    // rim lists: every rank owns entries [0, B) and receives the B entries of peer k at [(k + 1) B, (k + 2) B)
    const int chunk = B * (world + 1);
    HaloPlan h; h.world = world; h.chunk = chunk; h.send_cnt.assign(world, 0); h.send_off.assign(world, 0); h.recv_cnt.assign(world, 0); h.recv_off.assign(world, 0);
    std::vector<int> sidx, ridx;
    for (int k = 0; k < world; ++k) if (k != rank) {
        h.send_off[k] = (int)sidx.size(); h.send_cnt[k] = B; for (int i = 0; i < B; ++i) sidx.push_back(i);
        h.recv_off[k] = (int)ridx.size(); h.recv_cnt[k] = B; for (int i = 0; i < B; ++i) ridx.push_back((k + 1) * B + i);
    }
    h.n_send = (int)sidx.size(); h.n_recv = (int)ridx.size();
    int* d_idx = nullptr; float* d_vec = nullptr; double* d_red = nullptr;
    if (hipMalloc((void**)&d_idx, sizeof(int) * (sidx.size() + ridx.size() + 1)) != hipSuccess || hipMalloc((void**)&d_vec, sizeof(float) * 2 * (size_t)chunk) != hipSuccess
        || hipMalloc((void**)&d_red, sizeof(double) * RED) != hipSuccess) return 7;
    (void)hipMemcpy(d_idx, sidx.data(), sizeof(int) * sidx.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_idx + sidx.size(), ridx.data(), sizeof(int) * ridx.size(), hipMemcpyHostToDevice);
    h.d_send_idx = d_idx; h.d_recv_idx = d_idx + sidx.size();
    if (e.set_halo_lists(h, st)) return 8;
    std::vector<double> hr(RED); std::vector<float> hv(2 * (size_t)chunk);
    const double tri = 0.5 * world * (world + 1);
    int bad = 0;
    for (int rep = 0; rep < rounds && !bad; ++rep) {
        const int n = (rep & 1) ? RED : 4;
        for (int i = 0; i < n; ++i) hr[i] = (double)(rank + 1) * (i + 1) + rep;
        (void)hipMemcpyAsync(d_red, hr.data(), sizeof(double) * n, hipMemcpyHostToDevice, st);
        if (e.allreduce(d_red, (size_t)n, st)) { bad = 10; break; }
        (void)hipMemcpyAsync(hr.data(), d_red, sizeof(double) * n, hipMemcpyDeviceToHost, st);
        std::fill(hv.begin(), hv.end(), -1.0f);
        for (int i = 0; i < B; ++i) { hv[i] = (float)(1000 * rank + i + rep); hv[(size_t)chunk + i] = -(float)(1000 * rank + i + rep); }
        (void)hipMemcpyAsync(d_vec, hv.data(), sizeof(float) * hv.size(), hipMemcpyHostToDevice, st);
        if (e.push_halo(d_vec, h, st)) { bad = 11; break; }
        (void)hipMemcpyAsync(hv.data(), d_vec, sizeof(float) * hv.size(), hipMemcpyDeviceToHost, st);
        if (hipStreamSynchronize(st) != hipSuccess) { bad = 12; break; }
        for (int i = 0; i < n && !bad; ++i) if (hr[i] != tri * (i + 1) + (double)rep * world) bad = 13;
        for (int k = 0; k < world && !bad; ++k) if (k != rank) for (int i = 0; i < B && !bad; ++i)
            if (hv[(size_t)(k + 1) * B + i] != (float)(1000 * k + i + rep) || hv[(size_t)chunk + (size_t)(k + 1) * B + i] != -(float)(1000 * k + i + rep)) bad = 14;
        if (e.check(st)) bad = 15;
        if (bad) std::fprintf(stderr, "rank %d: failure %d in round %d\n", rank, bad, rep);
    }
    // the exchanges that run INSIDE the multi-workgroup PCG kernels of the three-launch sharded pass (one writer workgroup, every workgroup reads; rim items pushed and
    // consumed in the same launch), across processes
    if (!bad && e.selftest_fused(st)) { bad = 16; std::fprintf(stderr, "rank %d: the in-kernel multi-workgroup exchange failed\n", rank); }
    // nobody unmaps while a peer may still be storing into it
    tok = bad ? 0 : 1; (void)write_all(to_parent, &tok, 1); (void)read_all(from_parent, &tok, 1);
    e.destroy();
    return bad;
}

int main(int argc, char** argv) {
    const int world = argc > 1 ? std::atoi(argv[1]) : 2, rounds = argc > 2 ? std::atoi(argv[2]) : 200;
    if (world < 1 || world > 8) { std::fprintf(stderr, "world must be 1..8\n"); return 1; }
    std::vector<int> up(2 * world), down(2 * world); std::vector<pid_t> pid(world);
    for (int r = 0; r < world; ++r) { if (pipe(&up[2 * r]) || pipe(&down[2 * r])) return 1; }
    for (int r = 0; r < world; ++r) {           // the parent never touches HIP (the runtime does not survive a fork)
        pid[r] = fork();
        if (pid[r] == 0) { alarm(120); _exit(child(r, world, rounds, up[2 * r + 1], down[2 * r])); }
    }
    std::vector<unsigned char> all(64 * (size_t)world);
    bool ok = true;
    for (int r = 0; r < world; ++r) ok = read_all(up[2 * r], &all[64 * (size_t)r], 64) && ok;
    for (int r = 0; r < world; ++r) ok = write_all(down[2 * r + 1], all.data(), all.size()) && ok;
    for (int phase = 0; phase < 2; ++phase) {   // barrier after mapping, barrier before unmapping
        char tok = 0; for (int r = 0; r < world; ++r) ok = read_all(up[2 * r], &tok, 1) && ok;
        tok = 1; for (int r = 0; r < world; ++r) ok = write_all(down[2 * r + 1], &tok, 1) && ok;
    }
    int fails = ok ? 0 : 1;
    for (int r = 0; r < world; ++r) { int st = 0; waitpid(pid[r], &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { ++fails; std::fprintf(stderr, "rank %d exited with %d\n", r, WIFEXITED(st) ? WEXITSTATUS(st) : -1); } }
    std::printf("p2p_ipc_selftest: world %d, %d rounds: %s\n", world, rounds, fails ? "FAILED" : "ok");
    return fails ? 1 : 0;
}
