// Dev probe: the SUSTAINED shader clock under pure matrix load, measured inside the kernel.
// Every wave issues back-to-back MFMAs (fp32 32x32x2 or bf16 32x32x16, two accumulator chains) for `ms` milliseconds of
// s_memrealtime (constant 100 MHz) and counts shader cycles with s_memtime; wave 0 of workgroup 0 stores one
// (realtime ticks, shader cycles, MFMAs issued) sample per chunk.  clock = d(cycles) / d(realtime); achieved FLOP/s =
// all waves' MFMAs x FLOPs per MFMA / elapsed.  Built as a shared library so tools_dev/clocks.py can run it inside the
// process that has just executed training steps (same power / thermal state):
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools_dev/clock_probe.hip -o tools_dev/clock_probe.bin
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void burn_kernel(unsigned long long budget, unsigned long long* samples, int maxs,
                                                   unsigned long long* per_wave, float* sink) {
    f32x16 acc0, acc1;
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    bf16x8 pa, pb;
    for (int e = 0; e < 8; ++e) { pa[e] = (__bf16)(threadIdx.x * 1e-3f + e); pb[e] = (__bf16)(1.0f + e * 0.25f); }
    const unsigned long long r0 = wall_clock64();
    unsigned long long n = 0;
    int s = 0;
    const bool rec = blockIdx.x == 0 && threadIdx.x == 0;
    const unsigned long long max_n = budget / 100000ull * 50ull * 2048ull;   // safety: ends even if the realtime counter stalls
    // KIND 2 / 3: the same with RANDOM operand bits that change from MFMA to MFMA (eight operand registers per side,
    // xorshift-filled per lane): constant operands toggle almost nothing in the multiplier arrays, real data does --
    // the power (and with it the sustained clock) of a real kernel is that of random operands
    float ra[8], rb[8];
    bf16x8 qa[8], qb[8];
    {
        unsigned x = 0x9E3779B9u * (threadIdx.x + 1) + blockIdx.x * 0x85EBCA6Bu;
        auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
        for (int k = 0; k < 8; ++k) {
            ra[k] = __builtin_bit_cast(float, (rnd() & 0x807fffffu) | 0x3f000000u);       // random sign / mantissa, |v| in [0.5, 1)
            rb[k] = __builtin_bit_cast(float, (rnd() & 0x807fffffu) | 0x3f000000u);
            for (int e = 0; e < 8; ++e) {
                qa[k][e] = __builtin_bit_cast(__bf16, (unsigned short)((rnd() & 0x807fu) | 0x3f00u));
                qb[k][e] = __builtin_bit_cast(__bf16, (unsigned short)((rnd() & 0x807fu) | 0x3f00u));
            }
        }
    }
    for (;;) {
        for (int it = 0; it < 64; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (KIND == 0) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
                } else if (KIND == 1) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pb, pa, acc1, 0, 0, 0);
                } else if (KIND == 2) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[u & 7], rb[(u + 3) & 7], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(rb[u & 7], ra[(u + 5) & 7], acc1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[u & 7], qb[(u + 3) & 7], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qb[u & 7], qa[(u + 5) & 7], acc1, 0, 0, 0);
                }
            }
            if (KIND >= 2 && (it & 7) == 7) {            // keep the accumulators finite (random signs: a random walk, but cheap insurance)
                for (int e = 0; e < 16; ++e) { acc0[e] *= 0.5f; acc1[e] *= 0.5f; }
            }
        }
        n += 64 * 32;
        const unsigned long long r = wall_clock64();
        if (rec && s < maxs) {
            samples[3 * s] = r - r0;
            samples[3 * s + 1] = (unsigned long long)__builtin_readcyclecounter();
            samples[3 * s + 2] = n;
            ++s;
        }
        if (r - r0 >= budget || n >= max_n) break;
    }
    float t = 0.f;
    for (int e = 0; e < 16; ++e) t += acc0[e] + acc1[e];
    sink[blockIdx.x * 256 + threadIdx.x] = t;
    if ((threadIdx.x & 63) == 0) per_wave[blockIdx.x * 4 + (threadIdx.x >> 6)] = n;
    if (rec) samples[3 * maxs] = (unsigned long long)s;
}

// kind 0: fp32 MFMA, 1: bf16 MFMA (constant operands); 2 / 3: the same with random, changing operands.  samples: device buffer of 3 * maxs + 1 u64; per_wave: blocks * 4 u64; sink: blocks * 256 f32.
extern "C" int clock_probe_launch(int kind, int ms, int blocks, void* samples, int maxs, void* per_wave, void* sink,
                                  void* stream) {
    const unsigned long long budget = (unsigned long long)ms * 100000ull;      // 100 MHz ticks
#define BURN(K_)                                                                                       \
    hipLaunchKernelGGL(burn_kernel<K_>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, budget,         \
                       (unsigned long long*)samples, maxs, (unsigned long long*)per_wave, (float*)sink)
    if (kind == 0) BURN(0);
    else if (kind == 1) BURN(1);
    else if (kind == 2) BURN(2);
    else BURN(3);
#undef BURN
    return (int)hipGetLastError();
}
