// Helpers shared by the small-S xSlot kernels (xslot_small_fwd.h, xslot_small_bwd.h): v_mfma_f32_16x16x4_f32 tiles in the
// slot-per-lane-16 layout -- lane (i = lane & 15, g = lane >> 4), register r of tile t <-> M^T[c = 16 t + 4 g + r][i].
#pragma once
#include "xslot_common.h"

#ifndef XS_LOG2E
#define XS_LOG2E 1.4426950408889634f
#endif
#define XS16_UX_FLOATS (4 * 16 * XS_LD)     // one LDS hand-off buffer: [4 waves][16 slots][68]

// dev builds (tools_dev/xs_small_timing.hip): cycle stamps of wave 0 of every workgroup, [B][64]
#ifdef XS16_TIMING
__device__ long long* g_xs16_stamps = nullptr;
#define XS16_STAMP() do { if (threadIdx.x == 0 && g_xs16_stamps) g_xs16_stamps[blockIdx.x * 64 + nstamp] = (long long)__builtin_amdgcn_s_memtime(); ++nstamp; } while (0)
#define XS16_STAMP_INIT() int nstamp = 0
#else
#define XS16_STAMP() do { } while (0)
#define XS16_STAMP_INIT() do { } while (0)
#endif

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    // v_mfma_f32_16x16x4_f32: A[m = l & 15][k = l >> 4], B[k = l >> 4][n = l & 15]; D col = l & 15, row = 4 (l >> 4) + r
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// x(lane) summed over the four 16-lane rows (g = 0..3), the same bits in every lane: ((g0 + g1) + (g2 + g3))
__device__ __forceinline__ float xs16_gsum(float v) {
    const unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned us = __float_as_uint(s);
    const auto b = __builtin_amdgcn_permlane32_swap(us, us, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ double xs16_gsum_f64(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto al = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto ah = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double s = __hiloint2double((int)ah[0], (int)al[0]) + __hiloint2double((int)ah[1], (int)al[1]);
    return xs_halfsum_f64(s);
}
// sum over the 16 lanes of a row (the 16 slots); every lane of the row holds the result
__device__ __forceinline__ double xs16_rowsum_f64(double v) {
    v += xs_dpp_f64<0xB1>(v);
    v += xs_dpp_f64<0x4E>(v);
    v += xs_dpp_f64<0x141>(v);
    v += xs_dpp_f64<0x140>(v);
    return v;
}
__device__ __forceinline__ float xs16_rowsum(float v) {
    v += xs_dpp_f32<0xB1>(v);
    v += xs_dpp_f32<0x4E>(v);
    v += xs_dpp_f32<0x141>(v);
    v += xs_dpp_f32<0x140>(v);
    return v;
}
__device__ __forceinline__ f32x4 xs16_pick(const f32x4 (&p)[4], int w) {      // p[w], w wave-uniform (no dynamic indexing)
    return w == 0 ? p[0] : w == 1 ? p[1] : w == 2 ? p[2] : p[3];
}
