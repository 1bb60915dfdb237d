// Microbenchmark: cost of one more DEPENDENT dispatch in a stream (development aid).
// hipcc --offload-arch=gfx950 -O3 tools/microbench/dispatch_gap.hip -o /tmp/dispatch_gap && /tmp/dispatch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 12345) p[0] = 1; }
__global__ void k_touch(float* v, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) v[i] += 1.0f; }
int main() {
    float* v; hipMalloc(&v, 1 << 24); hipMemset(v, 0, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, int n, auto launch) {
        for (int i = 0; i < 50; ++i) launch();
        hipDeviceSynchronize(); hipEventRecord(e0);
        for (int i = 0; i < n; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-46s %.2f us per dispatch\n", name, ms * 1e3 / n);
    };
    run("empty kernel, 1 block x 64", 2000, [&] { k_empty<<<1, 64>>>(nullptr); });
    run("empty kernel, 1024 blocks x 256", 2000, [&] { k_empty<<<1024, 256>>>(nullptr); });
    run("touch 4 MB (1M floats), 4096 blocks x 256", 2000, [&] { k_touch<<<4096, 256>>>(v, 1 << 20); });
    run("touch 64 KB, 64 blocks x 256", 2000, [&] { k_touch<<<64, 256>>>(v, 1 << 14); });
    return 0;
}
