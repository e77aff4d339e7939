// Dev microbenchmark: does VALU work execute in the shadow of v_mfma_f32_32x32x16_bf16 (it does NOT for the fp32 MFMA,
// see mfma_shadow_bench.hip)?  Decides whether a split-bf16 ("bf16x3") emulation of the fp32 convolutions could convert
// its operands in the loader for free.   hipcc --offload-arch=gfx950 -O3 tools_dev/mfma_bf16_shadow_bench.hip -o /tmp/b && /tmp/b
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NV, int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void shadow(float* out, long long* cyc, int iters) {
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 1e-3f + e); b[e] = (__bf16)(1.0f + e * 0.25f); }
    float bb = 1.0f + threadIdx.x * 1e-4f;
    float x[8];
    for (int k = 0; k < 8; ++k) x[k] = threadIdx.x + k;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u % NACC], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[k % 8]) : "v"(bb));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int k = 0; k < NACC; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
    for (int k = 0; k < 8; ++k) s += x[k];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NV, int NACC, int WAVES> void run(float* out, long long* cyc) {
    const int iters = 100;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((shadow<NV, NACC, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
    printf("waves/CU=%d chains=%d VALU/MFMA=%2d: %.1f cycles per bf16 MFMA (32x32x16)\n", WAVES, NACC, NV, s / 256 / (iters * 32.0));
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    run<0, 1, 4>(out, cyc); run<0, 2, 4>(out, cyc); run<0, 4, 4>(out, cyc);
    run<2, 4, 4>(out, cyc); run<4, 4, 4>(out, cyc); run<8, 4, 4>(out, cyc); run<16, 4, 4>(out, cyc);
    run<0, 4, 8>(out, cyc); run<4, 4, 8>(out, cyc); run<8, 4, 8>(out, cyc);
    return 0;
}
