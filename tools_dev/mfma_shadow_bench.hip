// Dev microbenchmark: how much independent VALU / transcendental work fits in the shadow of one
// v_mfma_f32_32x32x2_f32 (one wave per SIMD)?  Prints cycles per MFMA for NV v_fma + NT v_exp per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int NT, int NACC, int KIND>
__global__ __launch_bounds__(256) void shadow(float* out, long long* cyc, int iters) {
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    __shared__ float sm[256];
    sm[threadIdx.x] = a;
    int sreg = 0;
    const int ldsaddr = (threadIdx.x & 63) * 4;
    float x[8];
    for (int k = 0; k < 8; ++k) x[k] = a + k;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % NACC], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[k % 8]) : "v"(b));
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                if (KIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(k + 3) % 8]));
                if (KIND == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg));
                if (KIND == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(x[(k + 3) % 8]) : "v"(ldsaddr));
            }
            if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int k = 0; k < NACC; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
    for (int k = 0; k < 8; ++k) s += x[k];
    s += sreg + sm[(threadIdx.x + 1) & 255];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NV, int NT, int NACC = 1, int KIND = 0> void run(float* out, long long* cyc) {
    const int iters = 100;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((shadow<NV, NT, NACC, KIND>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
    printf("NACC=%d NV=%2d N2=%d kind=%s: %.1f cycles per MFMA\n", NACC, NV, NT, KIND == 0 ? "v_exp" : KIND == 1 ? "salu" : "ds_read", s / 256 / (iters * 32.0));
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<0, 0>(out, cyc); run<4, 0>(out, cyc); run<8, 0>(out, cyc); run<16, 0>(out, cyc);
    run<0, 0, 2>(out, cyc); run<4, 0, 2>(out, cyc); run<8, 0, 2>(out, cyc); run<16, 0, 2>(out, cyc);
    run<0, 0, 4>(out, cyc); run<4, 0, 4>(out, cyc); run<8, 0, 4>(out, cyc); run<16, 0, 4>(out, cyc); run<8, 2, 4>(out, cyc);
    run<0, 8, 4, 1>(out, cyc); run<0, 16, 4, 1>(out, cyc); run<0, 2, 4, 2>(out, cyc); run<0, 4, 4, 2>(out, cyc); run<0, 8, 4, 2>(out, cyc);
    return 0;
}
