// EXPERIMENT (not part of the product): issue cost of the VALU instructions the build / cost kernels are made of, on one MI355X.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_rate tools/experiments/valu_rate.hip && /tmp/valu_rate
// Every wave runs N iterations of 8 independent dependency chains of one instruction kind; cycles per wave-instruction per SIMD =
// elapsed * clock * SIMDs / (waves * N * 8).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND> __global__ void __launch_bounds__(256) k(double* out, int n, double seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    double a[8]; float b[8]; f2 b2[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x * 1e-3; b[i] = (float)a[i]; b2[i] = f2{b[i], b[i] + 1.0f}; }
    const double m = 1.0000001, c = 1e-9; const float mf = 1.0000001f, cf = 1e-9f;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) a[i] = __builtin_fma(a[i], m, c);
            if (KIND == 1) a[i] = a[i] * m;
            if (KIND == 2) a[i] = a[i] + c;
            if (KIND == 3) b[i] = __builtin_fmaf(b[i], mf, cf);
            if (KIND == 4) b[i] = b[i] * mf;
            if (KIND == 5) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(b[i])); b[i] += 1.0f; }      // 1 cvt + 1 f32 add
            if (KIND == 6) { asm volatile("v_rcp_f64 %0, %1" : "=v"(a[i]) : "v"(a[i])); }
            if (KIND == 7) { asm volatile("v_sqrt_f64 %0, %1" : "=v"(a[i]) : "v"(a[i])); }
            if (KIND == 8) a[i] = 1.0 / a[i];                                                                           // full IEEE division sequence
            if (KIND == 9) { int x = __double2loint(a[i]); asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(x)); a[i] = __hiloint2double(__double2hiint(a[i]), x); }
            if (KIND == 11) { const f2 mm = {mf, mf}, cc = {cf, cf}; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(b2[i]) : "v"(b2[i]), "v"(mm), "v"(cc)); }   // 2 fp32 fma per lane
            if (KIND == 10) { asm volatile("v_floor_f64 %0, %1" : "=v"(a[i]) : "v"(a[i])); }
        }
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + b[i] + b2[i].x + b2[i].y;
    if (s == 12345.678) out[0] = s;
}

template <int KIND> static void run(const char* name, int waves_per_simd, double clock_hz, int cus) {
    double* out; (void)hipMalloc(&out, 8);
    const int n = 4096, blocks = cus * waves_per_simd;      // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(out, n, 1.0); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<KIND><<<blocks, 256>>>(out, n, 1.0); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)waves_per_simd * n * 8;
    std::printf("%-28s waves/SIMD %d  %8.3f ms  %6.2f cycles per wave-instruction (at %.2f GHz)\n", name, waves_per_simd, ms, ms * 1e-3 * clock_hz / instr_per_simd, clock_hz * 1e-9);
    (void)hipFree(out);
}

int main() {
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const double clock = pr.clockRate * 1e3; const int cus = pr.multiProcessorCount;
    std::printf("%s: %d CUs, %.2f GHz\n", pr.name, cus, clock * 1e-9);
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64", w, clock, cus); run<1>("v_mul_f64", w, clock, cus); run<2>("v_add_f64", w, clock, cus);
        run<3>("v_fma_f32", w, clock, cus); run<4>("v_mul_f32", w, clock, cus); run<5>("v_cvt_f64_f32 + v_add_f32", w, clock, cus);
        run<6>("v_rcp_f64", w, clock, cus); run<7>("v_sqrt_f64", w, clock, cus); run<8>("1.0 / x (fp64 division)", w, clock, cus);
        run<9>("v_mov_b32", w, clock, cus); run<10>("v_floor_f64", w, clock, cus); run<11>("v_pk_fma_f32 (2 fma per lane)", w, clock, cus);
    }
    return 0;
}
