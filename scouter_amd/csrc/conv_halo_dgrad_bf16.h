// Input gradient of the 3x3 / stride 1 / pad 1 layers with 32 INPUT channels per group when dy is STORED as bf16 (BASELINE
// configs[4], --precision bf16: the deep stem's 32 -> 32 and 32 -> 64 convolutions, resnet.py:471-489; layer1's radix
// convolutions, 64 -> 128 in two groups, split_attn.py:54-60) -- the GEMM is 32 columns wide per group, K = 9 taps x the 32 / 64
// dy channels of the group.  Included by conv_bf16.hip; reached through tile 3 (128 x 32) of
// scouter_conv2d_dgrad_bnbwd_bf16_io, whose workgroup <-> tile mapping, partial rows and epilogue it shares (SCOUTER_BHALO=0: off).
//
// igemm_bf16_kernel<128, 32> walks these layers as nine K-tiles of ONE or TWO k-steps each: per K-tile it re-fetches the 128 dy
// rows at the tap's shift from L2, converts a weight tile in flight and meets at a barrier -- 4-8 MFMAs per wave between two
// barriers and a load round trip (5 launches x 336 us per step at batch 256: 2.5 TB/s of algorithmic bytes).  Here, as in the
// resident-rows kernels (conv_xhalo.hip, phalo_kernel): in the flat NHWC pixel index the taps are the row offsets (1 - r) W +
// (1 - q), so the dy rows a 128-pixel tile needs for ALL nine taps are one contiguous range of 130 + 2 W pixels.  Per 32-channel
// half of the group's dy channels that range is fetched ONCE by LDS-DMA ([row][32 ch] bf16, 64-byte rows, the four 16-byte pieces
// of a row permuted by bits 2-3 of the row so that 16 consecutive rows at any tap shift touch 16 different bank groups -- the
// permutation is applied to the SOURCE piece a lane fetches, the DMA's LDS side is lane-linear), the half's weights [9 taps][32
// ci][32 co] are converted once (fp32 HWIO -> bf16, RNE: what the tile kernel's loader rounds to) into LDS with the same
// permutation, and 9 taps x 2 k-steps of v_mfma_f32_32x32x16_bf16 run without a barrier; border taps read a zero row (one select
// on the address).  K order: 32-channel half outer, tap inner (the tile kernel: tap outer) -- another fp32 summation order, same
// operands.
// The matrix work is 18 / 36 MFMAs per wave and tile; what these launches cost is their EPILOGUE (typed stores of 128 x 32 outputs,
// the ReLU mask, the fused BatchNorm-backward sums in fp64: measured 260 of 370 us with one tile per workgroup, and 58 us of the
// rest was the launch of 25 088 workgroups).  So the kernel is PERSISTENT -- 768 / 512 workgroups (44 / 61 KB of LDS: three / two per
// CU) walk the tile list, the weights are converted once per workgroup, not per tile -- and the epilogue's operands (the BatchNorm
// input, the mask words) are requested at the START of a tile, in front of the image DMA, so a tile pays one memory round trip, not
// two in a row (the tile kernels could not afford those registers: conv_bf16.hip; here LDS, not registers, bounds the residency).
// Arithmetic and partial-row layout of the epilogue are igemm_epilogue_typed's (one row per 128-pixel tile).
#pragma once
#include "conv_common.h"

#define BH_IMG_BYTES (23 * 1024)          // 130 + 2 * 112 = 354 rows of 64 bytes, rounded up to whole 1 KB DMA instructions
#define BH_ZERO_OFF BH_IMG_BYTES
#define BH_PS_OFF (BH_IMG_BYTES + 256)    // [4 waves][32 columns][2] fp64: the cross-wave sum of the BatchNorm-backward partials
#define BH_WGT_OFF (BH_PS_OFF + 2048)
#define BH_WGT_HALF (9 * 32 * 64)
static size_t bhalo_lds_bytes(int halves) { return (size_t)BH_WGT_OFF + (size_t)halves * BH_WGT_HALF; }

typedef __bf16 bh_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bh_bf16x4 __attribute__((ext_vector_type(4)));

// (a NON-template helper on purpose: the builtin inside a kernel template makes hipcc's host pass drop the kernel's launch stub)
__device__ __forceinline__ void bh_dma16(__amdgpu_buffer_rsrc_t rs, char* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
}

static bool bhalo_dgrad_ok(const ConvGeom& g, int io) {      // g: the input gradient's geometry (C / Cg = dy channels, N / Ng = dx channels)
    return g.R == 3 && g.S == 3 && g.stride == 1 && g.pad == 1 && g.Ng == 32 && (g.Cg == 32 || g.Cg == 64) && (io & SC_IO_X_BF16) &&
           g.W <= 112 && g.H >= 2 && g.Ho == g.H && g.Wo == g.W && (g.M + 512) * (long)g.C * 2 < (1L << 31) &&
           (g.M + 128) * (long)g.N < (1L << 31);
}

__global__ __launch_bounds__(256, 3) void bhalo_dgrad_kernel(const unsigned short* __restrict__ src, const float* __restrict__ wgt,
                                                             const float* __restrict__ addend, float* __restrict__ dst,
                                                             ConvGeom g, int mtiles, BnBwdFuse fz, int dst_bf16) {
    extern __shared__ __attribute__((aligned(1024))) char bh_lds[];
    char* img = bh_lds;
    char* wl = bh_lds + BH_WGT_OFF;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = mtiles * g.groups;
    const int grp = blockIdx.x % g.groups;                       // (the grid is a multiple of the groups: one group per workgroup)
    const int W = g.W, H = g.H, hw = H * W;
    const int halves = g.Cg / 32;

    if (tid < 16) *(float*)(bh_lds + BH_ZERO_OFF + tid * 4) = 0.f;

    // ---- the group's weights, once per workgroup: [half][tap][ci][co] fp32 (HWIO, co contiguous) -> bf16, nine float4 per thread
    // and half; 16-byte pieces of a [tap][ci] row permuted by bits 2-3 of ci (conflict-free B fragments)
    for (int hf = 0; hf < halves; ++hf) {
        const float* wsrc = wgt + (long)grp * g.Cg + hf * 32;
        f32x4 wv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int idx = tid + 256 * k, tap = idx >> 8, ci = (idx & 255) >> 3, q = idx & 7;
            wv[k] = *(const f32x4*)(wsrc + (long)tap * g.wtap + (long)ci * g.wrow + 4 * q);
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int idx = tid + 256 * k, tap = idx >> 8, ci = (idx & 255) >> 3, q = idx & 7;
            const bh_bf16x4 b = {(__bf16)wv[k][0], (__bf16)wv[k][1], (__bf16)wv[k][2], (__bf16)wv[k][3]};
            *(bh_bf16x4*)(wl + hf * BH_WGT_HALF + (tap * 32 + ci) * 64 + (((q >> 1) ^ ((ci >> 2) & 3)) * 16) + (q & 1) * 8) = b;
        }
    }

    // ---- DMA geometry: instruction i covers LDS rows 16 i .. 16 i + 15 (lane -> row 16 i + (lane >> 2), slot lane & 3)
    const int rows = 130 + 2 * W, ninst = (rows + 15) >> 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (long)grp * g.Cg), 0, 0x7fffffff, 0x00020000);

    // ---- epilogue geometry (igemm_epilogue_typed<128, 32, 32, 32>): 4 lanes per row (8 columns each), 16 rows per pass, 2 passes
    const int qcol = (lane & 3) * 8, qrow = lane >> 2;
    const int ncol = grp * 32 + qcol;
    const bool bwd = fz.part1 != nullptr;                        // (a second BatchNorm on the same gradient: the tile kernel)
    const int io = fz.io;
    const bool out_bf16 = dst_bf16 != 0 || (io & 8) != 0;
    f32x4 mu1[2], rs1[2];
    if (bwd) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            mu1[q] = *(const f32x4*)(fz.sv1 + ncol + 4 * q);
            rs1[q] = *(const f32x4*)(fz.sv1 + g.N + ncol + 4 * q);
        }
    }
    constexpr int LDE = 36;                                       // staging row: 32 columns + 4 (igemm_epilogue_typed)
    float* Es = (float*)bh_lds + wave * (32 * LDE);               // aliases the image: used behind the tile's last MFMA only
    double* Ps = (double*)(bh_lds + BH_PS_OFF);

    for (int bid = blockIdx.x; bid < nblk; bid += gridDim.x) {
        const int mt_id = bid / g.groups;
        const long m0 = (long)mt_id * 128;
        // ---- the epilogue's operands first: they fly with the image DMA (one round trip per tile)
        f32x4 xa[2][2];
        unsigned long long mw[2][4];
        if (bwd) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                long mr = m0 + wave * 32 + rr * 16 + qrow;
                mr = mr < g.M ? mr : g.M - 1;
                const long off = mr * g.N + ncol;
                sc_load8_rt(fz.x1, off, (io & 1) != 0, xa[rr][0], xa[rr][1]);
                if (fz.mask) {
                    const unsigned long long* w = fz.mask + ((off >> 2) >> 6) * 4;
#pragma unroll
                    for (int k = 0; k < 4; ++k) mw[rr][k] = w[k];
                }
            }
        }
        // ---- this lane's output pixel (A-fragment row l31 of the wave's 32-pixel block): tap validity and LDS rows
        const long m = m0 + 32 * wave + l31;
        unsigned vmask = 0;
        if (m < g.M) {
            const int rem = (int)(m % hw), y = rem / W, x = rem - y * W;
            const unsigned cb = (x + 1 < W ? 1u : 0u) | 2u | (x >= 1 ? 4u : 0u);           // q = 0: x + 1, q = 2: x - 1
            vmask = (y + 1 < H ? cb : 0u) | (cb << 3) | (y >= 1 ? (cb << 6) : 0u);         // r = 0: y + 1, r = 2: y - 1
        }
        const int jbase = 32 * wave + l31 + (W + 1);              // LDS row of the pixel itself (tap r = q = 1)

        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        for (int hf = 0; hf < halves; ++hf) {
            __syncthreads();                                      // every wave is done with the image (previous half / tile's staging)
            for (int i = wave; i < ninst; i += 4) {
                const int j = 16 * i + (lane >> 2), s = lane & 3, c = s ^ ((j >> 2) & 3);
                long p = m0 - (W + 1) + j;
                p = p < 0 ? 0 : (p >= g.M ? g.M - 1 : p);         // (rows outside the tensor: only masked taps point at them)
                bh_dma16(rs, img + i * 1024, (unsigned)((p * g.C + hf * 32 + 8 * c) * 2));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const char* wh = wl + hf * BH_WGT_HALF;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int r = t / 3, q = t % 3;
                const int j = jbase + (1 - r) * W + (1 - q);
                const bool ok = (vmask >> t) & 1u;
                const char* arow = ok ? img + j * 64 : bh_lds + BH_ZERO_OFF;
                const int sw = ok ? (j >> 2) & 3 : 0;
                const char* brow = wh + (t * 32 + l31) * 64;
                const int swb = (l31 >> 2) & 3;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bh_bf16x8 a = *(const bh_bf16x8*)(arow + (((2 * ks + h) ^ sw) * 16));
                    const bh_bf16x8 b = *(const bh_bf16x8*)(brow + (((2 * ks + h) ^ swb) * 16));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                }
            }
        }

        // ---- epilogue (the arithmetic of igemm_epilogue_typed<128, 32, 32, 32, true>): stage the wave's 32 x 32 tile through LDS,
        // 16-byte row segments
        __syncthreads();                                          // the image is dead
#pragma unroll
        for (int e = 0; e < 16; ++e) Es[mfma32_row(e, lane) * LDE + l31] = acc[e];
        __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): each wave reads back its own tile only
        double cs[8], cq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { cs[e] = 0.0; cq[e] = 0.0; }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = rr * 16 + qrow;
            const long mr = m0 + wave * 32 + row;
            if (mr >= g.M) continue;
            const long off = mr * g.N + ncol;
            f32x4 v[2];
            v[0] = *(const f32x4*)(Es + row * LDE + qcol);
            v[1] = *(const f32x4*)(Es + row * LDE + qcol + 4);
            if (addend) {
                f32x4 a0, a1;
                sc_load8_rt(addend, off, (io & 4) != 0, a0, a1);
                v[0] += a0; v[1] += a1;
            }
            if (bwd) {
                if (fz.mask) {
                    const int bit = (int)((off >> 2) & 63);
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[q][k] = ((mw[rr][k] >> (bit + q)) & 1ull) ? v[q][k] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        cs[4 * q + e] += v[q][e];
                        cq[4 * q + e] += (double)v[q][e] * ((xa[rr][q][e] - mu1[q][e]) * rs1[q][e]);
                    }
            }
            sc_store8_rt(dst, off, v[0], v[1], out_bf16);
        }
        if (bwd) {
#pragma unroll
            for (int o = 4; o < 64; o <<= 1)
#pragma unroll
                for (int e = 0; e < 8; ++e) { cs[e] += __shfl_xor(cs[e], o, 64); cq[e] += __shfl_xor(cq[e], o, 64); }
            if (qrow == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { Ps[(wave * 32 + qcol + e) * 2] = cs[e]; Ps[(wave * 32 + qcol + e) * 2 + 1] = cq[e]; }
            }
            __syncthreads();
            if (tid < 32) {
                double a0 = 0, a1 = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) { a0 += Ps[(w * 32 + tid) * 2]; a1 += Ps[(w * 32 + tid) * 2 + 1]; }
                double* o = fz.part1 + ((long)mt_id * g.N + grp * 32 + tid) * 2;
                o[0] = a0;
                o[1] = a1;
            }
        }
    }
}

// ---- FORWARD of the same layers (x stored as bf16; 32 input channels per group, 32 / 64 output channels per group): the same
// resident rows -- 130 + 2 W x rows per 128-pixel tile, fetched once for all nine taps --, the group's bf16 W^T rows ([tap][co][32
// ci], the forward's pre-transposed weights) copied once per persistent workgroup, K order tap outer / k-step inner: the MFMA
// sequence of igemm_bf16_kernel per accumulator, so the result -- and the fused BatchNorm statistics, one partial row per 128-pixel
// tile -- is BIT-IDENTICAL to the tile kernels' (tiles 1 / 3 of scouter_conv2d_fwd_bf16_io, through which it is reached).
static bool bhalo_fwd_ok(const ConvGeom& g, int io) {
    return g.R == 3 && g.S == 3 && g.stride == 1 && g.pad == 1 && g.Cg == 32 && (g.Ng == 32 || g.Ng == 64) && (io & SC_IO_X_BF16) &&
           g.W <= 112 && g.H >= 2 && g.Ho == g.H && g.Wo == g.W && (g.M + 512) * (long)g.C * 2 < (1L << 31);
}
static size_t bhalo_fwd_lds_bytes(int nb) {
    const size_t stage = (size_t)4 * (32 * nb) * (32 + 4) * 4;
    return (stage > BH_IMG_BYTES ? stage : (size_t)BH_IMG_BYTES) + 256 + (size_t)9 * 32 * nb * 64;
}

template <int NB>       // 32-column blocks per group: 1 | 2
__global__ __launch_bounds__(256, NB == 1 ? 3 : 2) void bhalo_fwd_kernel(const unsigned short* __restrict__ src,
                                                                         const unsigned short* __restrict__ wt,
                                                                         const float* __restrict__ bias,
                                                                         const float* __restrict__ addend, float* __restrict__ dst,
                                                                         double* __restrict__ bn_part, ConvGeom g, int relu,
                                                                         int mtiles, int dst_bf16) {
    constexpr int BN = 32 * NB;
    constexpr int STAGE = 4 * (32 * NB) * (32 + 4) * 4;           // igemm_epilogue_typed's staging, on top of the (dead) image
    constexpr int IMGR = STAGE > BH_IMG_BYTES ? STAGE : BH_IMG_BYTES;
    extern __shared__ __attribute__((aligned(1024))) char bh_lds[];
    char* img = bh_lds;
    char* zero = bh_lds + IMGR;
    char* wl = bh_lds + IMGR + 256;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = mtiles * g.groups;
    const int grp = blockIdx.x % g.groups;                       // (the grid is a multiple of the groups: one group per workgroup)
    const int W = g.W, H = g.H, hw = H * W;
    const BnBwdFuse nofz{};

    if (tid < 16) *(float*)(zero + tid * 4) = 0.f;
    // ---- the group's weights, once per workgroup: rows (tap, co) of 32 ci = 64 bytes, 16-byte pieces permuted by bits 2-3 of co
    for (int idx = tid; idx < 9 * BN * 4; idx += 256) {
        const int row = idx >> 2, p = idx & 3, tap = row / BN, co = row - tap * BN;
        const bh_bf16x8 v = *(const bh_bf16x8*)(wt + ((long)tap * g.N + grp * BN + co) * 32 + 8 * p);
        *(bh_bf16x8*)(wl + row * 64 + ((p ^ ((co >> 2) & 3)) * 16)) = v;
    }
    const int rows = 130 + 2 * W, ninst = (rows + 15) >> 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (long)grp * 32), 0, 0x7fffffff, 0x00020000);

    // wave layout of the tile kernels (so that the fp64 statistics are summed in the same order): NB = 1: four waves along the
    // pixels, 32 x 32 each (tile 3); NB = 2: two along the pixels x two along the columns, 64 x 32 each (tile 1)
    constexpr int MT = NB;
    const int wm = wave / NB, wn = wave % NB;
    for (int bid = blockIdx.x; bid < nblk; bid += gridDim.x) {
        const int mt_id = bid / g.groups;
        const long m0 = (long)mt_id * 128;
        unsigned vmask[MT];
        int jbase[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int prow = wm * (32 * MT) + 32 * i + l31;
            const long m = m0 + prow;
            vmask[i] = 0;
            if (m < g.M) {
                const int rem = (int)(m % hw), y = rem / W, x = rem - y * W;
                const unsigned cb = (x >= 1 ? 1u : 0u) | 2u | (x + 1 < W ? 4u : 0u);       // q = 0: x - 1, q = 2: x + 1
                vmask[i] = (y >= 1 ? cb : 0u) | (cb << 3) | (y + 1 < H ? (cb << 6) : 0u); // r = 0: y - 1, r = 2: y + 1
            }
            jbase[i] = prow + (W + 1);
        }
        __syncthreads();                                          // every wave is done with the image / the previous tile's staging
        for (int i = wave; i < ninst; i += 4) {
            const int j = 16 * i + (lane >> 2), s = lane & 3, c = s ^ ((j >> 2) & 3);
            long p = m0 - (W + 1) + j;
            p = p < 0 ? 0 : (p >= g.M ? g.M - 1 : p);
            bh_dma16(rs, img + i * 1024, (unsigned)((p * g.C + 8 * c) * 2));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x16 acc[MT][1];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][0][e] = 0.f;
        const int swb = (l31 >> 2) & 3;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const char* brow = wl + (t * BN + wn * 32 + l31) * 64;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bh_bf16x8 b = *(const bh_bf16x8*)(brow + (((2 * ks + h) ^ swb) * 16));
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int j = jbase[i] + (t / 3 - 1) * W + (t % 3 - 1);
                    const bool ok = (vmask[i] >> t) & 1u;
                    const char* arow = ok ? img + j * 64 : zero;
                    const int sw = ok ? (j >> 2) & 3 : 0;
                    const bh_bf16x8 a = *(const bh_bf16x8*)(arow + (((2 * ks + h) ^ sw) * 16));
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i][0], 0, 0, 0);
                }
            }
        }
        igemm_epilogue_typed<128, BN, 32 * MT, 32, false>(acc, (float*)bh_lds, g, bias, addend, dst, bn_part, relu, m0, 0, grp, mt_id,
                                                     &nofz, dst_bf16 != 0);
    }
}
