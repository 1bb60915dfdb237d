// Microbenchmark: what does the ACCESS PATTERN of k_eg_pass cost by itself?  (development aid; not part of the library)
//   A: persistent 1024-thread workgroups, one per CU, walking 1024-entry tiles; per lane 5 slots x 8 float4 planes (the AoSoA row store)
//   B: the same bytes as a plain grid-stride float4 read
//   C: pattern A with two slots in flight per lane
// hipcc --offload-arch=gfx950 -O3 tools/microbench/rows_stream.hip -o /tmp/rows_stream && /tmp/rows_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int SLOTS = 5;
__host__ __device__ inline size_t row_index(size_t a, int slot, int plane) { return (((a >> 6) * SLOTS + slot) * 8 + plane) * 64 + (a & 63); }

__global__ void __launch_bounds__(1024) kA(const float4* __restrict__ rows, int A, int tiles_per_block, float* out) {
    const int ntiles = (A + 1023) / 1024; float s = 0.0f;
    for (int tile = blockIdx.x * tiles_per_block; tile < (blockIdx.x + 1) * tiles_per_block && tile < ntiles; ++tile) {
        const size_t a = (size_t)tile * 1024 + threadIdx.x; if (a >= (size_t)A) continue;
        for (int k = 0; k < SLOTS; ++k) {
            const float4* row = rows + row_index(a, k, 0); float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = row[q * 64];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q].x * v[q].y + v[q].z * v[q].w;
        }
    }
    if (s == 12345.678f) out[0] = s;
}
__global__ void __launch_bounds__(1024) kC(const float4* __restrict__ rows, int A, int tiles_per_block, float* out) {
    const int ntiles = (A + 1023) / 1024; float s = 0.0f;
    for (int tile = blockIdx.x * tiles_per_block; tile < (blockIdx.x + 1) * tiles_per_block && tile < ntiles; ++tile) {
        const size_t a = (size_t)tile * 1024 + threadIdx.x; if (a >= (size_t)A) continue;
        float4 va[8], vb[8];
        { const float4* row = rows + row_index(a, 0, 0);
#pragma unroll
          for (int q = 0; q < 8; ++q) va[q] = row[q * 64]; }
        for (int k = 0; k < SLOTS; k += 2) {
            if (k + 1 < SLOTS) { const float4* row = rows + row_index(a, k + 1, 0);
#pragma unroll
                for (int q = 0; q < 8; ++q) vb[q] = row[q * 64]; }
#pragma unroll
            for (int q = 0; q < 8; ++q) s += va[q].x * va[q].y + va[q].z * va[q].w;
            if (k + 2 < SLOTS) { const float4* row = rows + row_index(a, k + 2, 0);
#pragma unroll
                for (int q = 0; q < 8; ++q) va[q] = row[q * 64]; }
            if (k + 1 < SLOTS) {
#pragma unroll
                for (int q = 0; q < 8; ++q) s += vb[q].x * vb[q].y + vb[q].z * vb[q].w; }
        }
    }
    if (s == 12345.678f) out[0] = s;
}
__global__ void __launch_bounds__(256) kB(const float4* __restrict__ rows, size_t n4, float* out) {
    float s = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { const float4 v = rows[i]; s += v.x * v.y + v.z * v.w; }
    if (s == 12345.678f) out[0] = s;
}
int main() {
    const int A = 2286345; const size_t n4 = ((size_t)(A + 63) / 64) * 64 * SLOTS * 8;
    float4* rows; float* out; CK(hipMalloc(&rows, n4 * 16)); CK(hipMalloc(&out, 4)); CK(hipMemset(rows, 0, n4 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int ntiles = (A + 1023) / 1024, blocks = 256, tpb = (ntiles + blocks - 1) / blocks;
    auto time = [&](const char* name, auto launch) { float best = 1e9f; for (int it = 0; it < 6; ++it) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms; }
        printf("%-40s %.4f ms  %.2f TB/s\n", name, best, (double)n4 * 16 / best / 1e9); };
    time("A: eg_pass pattern, 1 slot in flight", [&] { kA<<<blocks, 1024>>>(rows, A, tpb, out); });
    time("C: eg_pass pattern, 2 slots in flight", [&] { kC<<<blocks, 1024>>>(rows, A, tpb, out); });
    time("B: grid-stride float4 read, 2048 blocks", [&] { kB<<<2048, 256>>>(rows, n4, out); });
    time("B: grid-stride float4 read, 8192 blocks", [&] { kB<<<8192, 256>>>(rows, n4, out); });
    return 0;
}
