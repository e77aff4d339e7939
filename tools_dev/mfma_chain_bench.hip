// Dev microbenchmark: cycles per v_mfma_f32_32x32x2_f32 when NACC independent accumulator chains are interleaved
// (NACC = 1: every MFMA depends on the previous one).  One wave per SIMD (256-thread block), one block per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void chain(float* out, long long* cyc, int iters) {
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32 / NACC; ++u)
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int k = 0; k < NACC; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC> void run(float* out, long long* cyc) {
    const int iters = 200;
    hipLaunchKernelGGL(chain<NACC>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(chain<NACC>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
    printf("NACC=%d: %.1f cycles per MFMA (mean over 256 blocks)\n", NACC, s / 256 / (iters * 32.0));
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<1>(out, cyc); run<2>(out, cyc); run<4>(out, cyc); run<8>(out, cyc);
    return 0;
}
