// Shared device/host helpers for the scouter_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define SC_OK 0
#define SC_ERR_ARG (-1)
#define SC_ERR_LAUNCH (-2)
#define SC_ERR_UNSUPPORTED (-3)
#define SC_ERR_WORKSPACE (-4)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// thread-local error text: the ABI never throws and never exits (include/scouter_hip.h)
void sc_set_error(const char* fmt, ...);
int sc_check_launch(const char* what);

#define SC_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            sc_set_error(__VA_ARGS__);        \
            return SC_ERR_ARG;                \
        }                                     \
    } while (0)

#define SC_UNSUPPORTED(cond, ...)             \
    do {                                      \
        if (!(cond)) {                        \
            sc_set_error(__VA_ARGS__);        \
            return SC_ERR_UNSUPPORTED;        \
        }                                     \
    } while (0)

// ---- optional per-kernel timing (bench.py's roofline leg): hipEvents recorded on the launch stream around ONE kernel
// launch, aggregated by kernel-instance name (static strings, e.g. "igemm_fwd<128x128>")
struct ScProfScope {
    hipStream_t stream;
    int slot;
    ScProfScope(const char* name, hipStream_t stream, double flops, double bytes);
    ~ScProfScope();
};

static inline int sc_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- device helpers
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    // v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// exact split of an fp32 value into three bf16 terms x = hi + mid + lo (RNE at each step; the residuals are exactly
// representable): the operand format of the plane convolutions (conv_planes.hip)
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split3_bf16(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    const __bf16 bh = (__bf16)x;
    const float r1 = x - (float)bh;
    const __bf16 bm = (__bf16)r1;
    const float r2 = r1 - (float)bm;
    const __bf16 bl = (__bf16)r2;
    h = __builtin_bit_cast(unsigned short, bh);
    m = __builtin_bit_cast(unsigned short, bm);
    l = __builtin_bit_cast(unsigned short, bl);
}
// writes the planes of four consecutive elements (element index i4 * 4) of a tensor with `plane_elems` elements
__device__ __forceinline__ void store_planes4(unsigned short* __restrict__ planes, long plane_elems, int nplanes, long i4,
                                              f32x4 v) {
    u16x4 h, m, l;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned short a, b, c;
        split3_bf16(v[k], a, b, c);
        h[k] = a; m[k] = b; l[k] = c;
    }
    *(u16x4*)(planes + i4 * 4) = h;
    if (nplanes == 3) {
        *(u16x4*)(planes + plane_elems + i4 * 4) = m;
        *(u16x4*)(planes + 2 * plane_elems + i4 * 4) = l;
    }
}

// ---- activation storage type (round 4: --precision bf16 keeps the 4g-wide tensors of a bottleneck -- conv3 / downsample
// outputs and the block outputs -- as bf16 in HBM).  The kernels that touch them take an `io` bit set next to untyped
// pointers; values are widened / RNE-rounded (v_cvt_pk_bf16_f32) at the load / store, all arithmetic stays fp32.
#define SC_IO_X_BF16 1       // the main input (x / src)
#define SC_IO_Y_BF16 2       // the output (y / dst)
#define SC_IO_R_BF16 4       // the residual / second input
typedef __bf16 sc_bf16x4 __attribute__((ext_vector_type(4)));
template <bool BF>
__device__ __forceinline__ f32x4 sc_load4(const void* __restrict__ p, long elem) {
    if constexpr (BF) {
        const sc_bf16x4 h = *(const sc_bf16x4*)((const __bf16*)p + elem);
        return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    } else {
        return *(const f32x4*)((const float*)p + elem);
    }
}
template <bool BF>
__device__ __forceinline__ f32x4 sc_load4_nt(const void* __restrict__ p, long elem) {       // streamed once: non-temporal
    if constexpr (BF) {
        const sc_bf16x4 h = __builtin_nontemporal_load((const sc_bf16x4*)((const __bf16*)p + elem));
        return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    } else {
        return __builtin_nontemporal_load((const f32x4*)((const float*)p + elem));
    }
}
template <bool BF>
__device__ __forceinline__ void sc_store4(void* __restrict__ p, long elem, f32x4 v) {
    if constexpr (BF) {
        sc_bf16x4 h;
#pragma unroll
        for (int k = 0; k < 4; ++k) h[k] = (__bf16)v[k];
        *(sc_bf16x4*)((__bf16*)p + elem) = h;
    } else {
        *(f32x4*)((float*)p + elem) = v;
    }
}
// eight consecutive bf16-stored elements as ONE 16-byte access (the all-bf16 apply passes, bn_elem.hip)
typedef __bf16 sc_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void sc_load8_bf16(const void* __restrict__ p, long elem, f32x4& lo, f32x4& hi) {
    const sc_bf16x8 h = __builtin_nontemporal_load((const sc_bf16x8*)((const __bf16*)p + elem));
    lo = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    hi = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
}
__device__ __forceinline__ void sc_store8_bf16(void* __restrict__ p, long elem, f32x4 lo, f32x4 hi) {
    sc_bf16x8 h;
#pragma unroll
    for (int k = 0; k < 4; ++k) { h[k] = (__bf16)lo[k]; h[4 + k] = (__bf16)hi[k]; }
    *(sc_bf16x8*)((__bf16*)p + elem) = h;
}
// bits of a (even) and b (odd) interleaved: result bit 2l = a bit l, bit 2l+1 = b bit l (l < 32); scalar ALU on ballots
__device__ __forceinline__ unsigned long long sc_spread32(unsigned long long v) {
    v &= 0xffffffffull;
    v = (v | (v << 16)) & 0x0000ffff0000ffffull;
    v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
    v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}
__device__ __forceinline__ unsigned long long sc_interleave32(unsigned long long a, unsigned long long b) {
    return sc_spread32(a) | (sc_spread32(b) << 1);
}

// eight consecutive elements, storage type as a wave-uniform run-time flag (typed convolution epilogue)
__device__ __forceinline__ void sc_load8_rt(const void* __restrict__ p, long elem, bool bf, f32x4& lo, f32x4& hi) {
    if (bf) {
        const sc_bf16x8 h = *(const sc_bf16x8*)((const __bf16*)p + elem);
        lo = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        hi = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
    } else {
        lo = *(const f32x4*)((const float*)p + elem);
        hi = *(const f32x4*)((const float*)p + elem + 4);
    }
}
__device__ __forceinline__ void sc_store8_rt(void* __restrict__ p, long elem, f32x4 lo, f32x4 hi, bool bf) {
    if (bf) {
        sc_store8_bf16(p, elem, lo, hi);
    } else {
        *(f32x4*)((float*)p + elem) = lo;
        *(f32x4*)((float*)p + elem + 4) = hi;
    }
}

// run-time flag variants for epilogues (a wave-uniform branch per access)
__device__ __forceinline__ f32x4 sc_load4_rt(const void* __restrict__ p, long elem, bool bf) {
    return bf ? sc_load4<true>(p, elem) : sc_load4<false>(p, elem);
}
__device__ __forceinline__ void sc_store4_rt(void* __restrict__ p, long elem, f32x4 v, bool bf) {
    if (bf) sc_store4<true>(p, elem, v); else sc_store4<false>(p, elem, v);
}

// BatchNorm apply, one expression for EVERY kernel that evaluates it (apply pass, fused split-attention passes and
// their backward): (x - mean) * scale + shift as an explicit fma, so the ReLU sign of an element is the same bit
// wherever it is recomputed.
__device__ __forceinline__ f32x4 bn_affine(f32x4 x, f32x4 mu, f32x4 sc, f32x4 sh) {
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = fmaf(x[k] - mu[k], sc[k], sh[k]);
    return v;
}

// bijective XCD-aware remap of a linear workgroup id (8 XCDs; consecutive logical ids share an XCD's L2)
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ReLU sign bitmask (1 bit per element instead of re-reading the 4-byte activation in the backward): float4 index i
// of the flat tensor owns bit (i & 63) of the four words mask[(i >> 6) * 4 + k], k = component.  Written with one
// wave ballot per component by the BatchNorm apply pass, whose waves cover 64 consecutive float4s.
__device__ __forceinline__ void relu_mask_apply(f32x4& g, const unsigned long long* __restrict__ mbits, long i4) {
    const unsigned long long* w = mbits + (i4 >> 6) * 4;
    const int bit = (int)(i4 & 63);
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = ((w[k] >> bit) & 1ull) ? g[k] : 0.f;
}
