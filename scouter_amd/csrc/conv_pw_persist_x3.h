// Persistent pointwise INPUT GRADIENT with the fused BatchNorm-backward epilogue on the bf16 matrix cores, fp32 tensors
// (round 5) -- tile 5 of scouter_conv2d_dgrad_bnbwd_f32.  Call sites: conv1 of the ResNeSt bottlenecks of layer1-3
// (/root/reference/timm/models/resnest.py:111-118; GEMM-K = Cout = 64 / 128 / 256), whose input gradient is the block-input
// gradient that the previous block's bn3 / downsample BatchNorm backward consumes.
//
// These launches are HBM streams with a small GEMM attached (13-18 bytes per output element against 2 K flops): the fp32
// kernels run them at 2.8 TB/s (igemm_kernel<64, 64, .., FUSE>) / 4.6 TB/s on one layer (pwp_fused_kernel: one workgroup
// per CU, 4-8-byte accesses per lane, the exact-fp32 MFMA -- K = 256 alone is 60 % of the HBM time of its tile).  The
// bf16-storage twin of round 5 (conv_pw_persist_bf16.h) reached 5.2 TB/s with three changes, all taken over here:
//   * v_mfma_f32_16x16x32_bf16 with PERMUTED columns -- MFMA column l of 16-column block j is actual column 4 l + j, so a
//     lane's four accumulators of a row are four CONSECUTIVE fp32 columns: one 16-byte access per row and operand, 16 lanes
//     = 256 contiguous bytes of a row, no transposition, no LDS staging;
//   * a 16 x 64 wave tile (4 rows x 4 columns per lane): accumulators + the prefetched epilogue operands of a tile fit 256
//     registers, so TWO workgroups per CU are resident for K <= 128 -- one streams while the other multiplies;
//   * one partial row per WORKGROUP (lane-local fp64 sums over all its tiles).
// fp32 accuracy comes from the three-way bf16 split of conv_planes.hip / conv_x3.hip: dY stays fp32 in HBM, streams through
// a four-slot LDS ring of [64 rows][32 k] fp32 stages (LDS-DMA, three stages ahead) and is split hi + mid + lo IN
// REGISTERS when a lane reads its fragment (8 values, ~52 VALU, in the shadow of 24 MFMAs); W^T is split once per
// workgroup into three resident LDS planes; six products, the five small ones first into their own accumulator set.  A third
// of the exact-fp32 MFMA kernel's rounding error at 3/8 of its matrix time.  The summation order over K differs from every
// other kernel's: equal to fp32 rounding, which tile runs is the static table's choice.
#pragma once
#include "conv_common.h"

#include <type_traits>

typedef __bf16 xpw_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short xpw_u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned xpw_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void xpw_dma16(__amdgpu_buffer_rsrc_t rs, char* lds, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ f32x4 xpw_mfma(xpw_bf16x8 a, xpw_bf16x8 b, f32x4 c) {
    // A[i = l & 15][k = 8 (l >> 4) + 0..7], B[k = 8 (l >> 4) + 0..7][j = l & 15]; D[row 4 (l >> 4) + e][col l & 15]
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// KS = K / 64 (K = Cout of the layer: 64 / 128 / 256); grid = wg_per_col * (N / 64) workgroups, workgroup (p, cg) owns the
// 64-row tiles p, p + wg_per_col, ... of the 64-column group cg.  part1 / part2: [wg_per_col][N][2] fp64.
template <int KS, bool TWO>
__global__ __launch_bounds__(256, (KS <= 2 ? 2 : 1)) void xpw_fused_kernel(
    const float* __restrict__ src, const float* __restrict__ wgt, const float* __restrict__ addend, float* __restrict__ dst,
    long M, int N, int mtiles, int wg_per_col, BnBwdFuse fz, long mask_words) {
    constexpr int K = 64 * KS, BN = 64, RB = 2 * K;              // bytes per W^T plane row
    constexpr int PLANE = BN * RB, WBYTES = 3 * PLANE, SLOT = 64 * 128, NST = 2 * KS;   // stages (32 k) per tile
    constexpr int CHM = (K / 8 < 16 ? K / 8 : 16) - 1;           // W^T swizzle: chunk c of row n at c ^ ((n >> 2) & CHM)
    extern __shared__ __attribute__((aligned(1024))) char xpw_lds[];
    char* Wl = xpw_lds;
    char* ring = xpw_lds + WBYTES;

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int colgroups = N / BN;
    // workgroups of one M range (same p) share an XCD (dispatch is round-robin over the eight XCDs): dY is read from HBM once
    const int L = blockIdx.x;
    const int cg = (L >> 3) % colgroups, p = (L & 7) + 8 * ((L >> 3) / colgroups);
    const int n0 = cg * BN;
    const int n_my = p < mtiles ? (mtiles - p + wg_per_col - 1) / wg_per_col : 0;
    const int nstage4 = (n_my * NST + 3) / 4;

    const unsigned obytes = (unsigned)(M * N * 4);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)(M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)(addend ? addend : dst), 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc((void*)fz.x1, 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc((void*)(TWO ? fz.x2 : fz.x1), 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(fz.mask ? (const void*)fz.mask : (const void*)dst), 0, (unsigned)(fz.mask ? mask_words * 8 : 0), 0x00020000);
    const bool has_add = addend != nullptr, has_mask = fz.mask != nullptr;

    // ---- dY ring: [64 rows][32 k] fp32 = 8 KB per stage, two DMA instructions per thread (8 pieces of 8 rows x 128 bytes);
    // 16-byte chunk c of row r at chunk c ^ (r & 7)
    unsigned a_voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 8 * (wave + 4 * j) + (lane >> 3), c = (lane & 7) ^ (r & 7);
        a_voff[j] = (unsigned)((r * K + c * 4) * 4);
    }
    auto issue = [&](int G, int slot) __attribute__((always_inline)) {
        const int ti = G / NST, kc = G % NST;
        const long m0 = (long)__builtin_amdgcn_readfirstlane(p + ti * wg_per_col) * 64;
        const long off = (m0 * K + kc * 32) * 4;
        const int soff = off > 0x7fffffffL ? 0x7fffffff : (int)off;            // (beyond the tensor: reads zeros)
#pragma unroll
        for (int j = 0; j < 2; ++j) xpw_dma16(rs_a, ring + slot * SLOT + (wave + 4 * j) * 1024, a_voff[j], soff);
    };
    issue(0, 0); issue(1, 1); issue(2, 2); issue(3, 3);
    // ---- W^T rows n0 .. n0 + 63 (n = ci, k = co contiguous: the HWIO layout of a 1x1 weight itself), split three-way
    for (int e = tid; e < BN * (K / 8); e += 256) {
        const int n = e / (K / 8), c = e % (K / 8);
        const float* wp = wgt + (long)(n0 + n) * K + 8 * c;
        const f32x4 lo = *(const f32x4*)wp, hi = *(const f32x4*)(wp + 4);
        xpw_u16x8 ph, pm, pl;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned short a, b, cc;
            split3_bf16(i < 4 ? lo[i & 3] : hi[i & 3], a, b, cc);
            ph[i] = a; pm[i] = b; pl[i] = cc;
        }
        char* d = Wl + n * RB + ((c ^ ((n >> 2) & CHM)) << 4);
        *(xpw_u16x8*)(d) = ph;
        *(xpw_u16x8*)(d + PLANE) = pm;
        *(xpw_u16x8*)(d + 2 * PLANE) = pl;
    }

    // ---- fragment geometry: A row = 16 wave + l15 (fp32 chunks 2 q, 2 q + 1 of the stage); B row (block j) = 4 l15 + j
    const int arow = 16 * wave + l15;
    const char* a_base = ring + arow * 128;
    const int a_sw = arow & 7;
    const char* b_base = Wl + 4 * l15 * RB;
    const int b_t = q ^ (l15 & CHM);                            // (chunk base | q) ^ swizzle = chunk base ^ b_t
    // ---- epilogue geometry: rows 16 wave + 4 q + e, columns col0 .. col0 + 3 (one float4 group)
    const int col0 = n0 + 4 * l15;
    const int row_in = 16 * wave + 4 * q;
    const unsigned o_voff = (unsigned)(((long)row_in * N + col0) * 4);
    const f32x4 mu1 = *(const f32x4*)(fz.sv1 + col0), rs1 = *(const f32x4*)(fz.sv1 + N + col0);
    f32x4 mu2 = {0.f, 0.f, 0.f, 0.f}, rs2 = {0.f, 0.f, 0.f, 0.f};
    if (TWO) { mu2 = *(const f32x4*)(fz.sv2 + col0); rs2 = *(const f32x4*)(fz.sv2 + N + col0); }

    // two accumulator sets (conv_planes.hip): hi * hi in `acc`, the five correction products in `accl`
    f32x4 acc[4], accl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; accl[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    double sg[4], sx1[4], sx2[TWO ? 4 : 1];
#pragma unroll
    for (int k = 0; k < 4; ++k) { sg[k] = 0.0; sx1[k] = 0.0; if (TWO) sx2[TWO ? k : 0] = 0.0; }
    // epilogue operands of the tile whose last K stage is running
    f32x4 pa[4], px1[4], px2[TWO ? 4 : 1];
    xpw_u32x4 pm[4][2];

#define SB() __builtin_amdgcn_sched_barrier(0)
    auto prefetch = [&](int ti) __attribute__((always_inline)) {
        const long m0 = (long)__builtin_amdgcn_readfirstlane(p + ti * wg_per_col) * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long off = (m0 + e) * (long)N * 4;
            const int soff = off > 0x7fffffffL ? 0x7fffffff : (int)off;
            if (has_add) pa[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, o_voff, soff, 0));
            px1[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x1, o_voff, soff, 0));
            if (TWO) px2[TWO ? e : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x2, o_voff, soff, 0));
            if (has_mask) {
                // ReLU bits: float4 group i4 = (m N + c) / 4 owns bit (i4 & 63) of the words mask[(i4 >> 6) * 4 + (c & 3)]
                const long i4 = ((m0 + row_in + e) * (long)N + col0) >> 2;
                const unsigned mo = (unsigned)((i4 >> 6) * 32);                    // (rows beyond M: beyond the words: zeros)
                pm[e][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_m, mo, 0, 0);
                pm[e][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_m, mo + 16u, 0, 0);
            }
        }
    };
    auto epilogue = [&](int ti) __attribute__((always_inline)) {
        const long m0 = (long)__builtin_amdgcn_readfirstlane(p + ti * wg_per_col) * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int bit = (int)((((m0 + row_in + e) * (long)N + col0) >> 2) & 63);
            f32x4 out;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v = acc[k][e] + accl[k][e];
                if (has_add) v += pa[e][k];
                if (has_mask) {
                    const unsigned long long w = ((unsigned long long)pm[e][k >> 1][2 * (k & 1) + 1] << 32) | pm[e][k >> 1][2 * (k & 1)];
                    v = ((w >> bit) & 1ull) ? v : 0.f;
                }
                out[k] = v;
                const double d = (double)v;
                sg[k] += d;
                sx1[k] += d * (double)((px1[e][k] - mu1[k]) * rs1[k]);
                if (TWO) sx2[TWO ? k : 0] += d * (double)((px2[TWO ? e : 0][k] - mu2[k]) * rs2[k]);
                acc[k][e] = 0.f;
                accl[k][e] = 0.f;
            }
            // (the row offset in the PER-LANE offset, an immediate 0 as scalar offset: see conv_pw_persist_bf16.h -- with a
            //  register there the compiler leaves no wait state before the data registers are overwritten, and two waves per
            //  SIMD then store the next row's address arithmetic.)  Rows beyond M: beyond num_records, dropped.
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(xpw_u32x4, out), rs_o,
                                                   o_voff + (unsigned)((m0 + e) * (long)N * 4), 0, 0);
        }
    };
    auto stage_body = [&](int G, int slot, int kc) __attribute__((always_inline)) {
        const bool last = kc == NST - 1 && G / NST < n_my;
        // this lane's eight k-values of its row (fp32) and the twelve weight fragments of the stage
        const char* As = a_base + slot * SLOT;
        const f32x4 a0 = *(const f32x4*)(As + (((2 * q) ^ a_sw) << 4)), a1 = *(const f32x4*)(As + (((2 * q + 1) ^ a_sw) << 4));
        xpw_bf16x8 fb[3][4];
        const char* bp = b_base + (((kc * 4) ^ b_t) << 4);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[pl][j] = *(const xpw_bf16x8*)(bp + pl * PLANE + j * RB);
        if (last) prefetch(G / NST);                             // lands under this stage's MFMAs (and the other workgroup's work)
        xpw_u16x8 h, m, l;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned short a, b, c;
            split3_bf16(i < 4 ? a0[i & 3] : a1[i & 3], a, b, c);
            h[i] = a; m[i] = b; l[i] = c;
        }
        const xpw_bf16x8 fa[3] = {__builtin_bit_cast(xpw_bf16x8, h), __builtin_bit_cast(xpw_bf16x8, m), __builtin_bit_cast(xpw_bf16x8, l)};
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};      // smallest terms first (as pconv_kernel)
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (pr < 5) accl[j] = xpw_mfma(fa[PA[pr]], fb[PB[pr]][j], accl[j]);
                else acc[j] = xpw_mfma(fa[PA[pr]], fb[PB[pr]][j], acc[j]);
            }
        SB();
        // the next stage's DMA has landed (two younger stages = 4 instructions may stay in flight; the epilogue's operands,
        // younger still, are waited for by the compiler's own counting where they are used)
        if (!last) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (last) epilogue(G / NST);
        __builtin_amdgcn_s_barrier();                            // slot is free, slot + 1 is visible to every wave
        SB();
        issue(G + 4, slot);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // stage 0 .. 3 and the weight rows have landed
    __syncthreads();
    for (int base = 0; base < nstage4; ++base) {
#pragma unroll
        for (int S = 0; S < 4; ++S) stage_body(4 * base + S, S, NST <= 4 ? S % NST : (4 * base + S) % NST);
    }
#undef SB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- one partial row per workgroup: the four row quads of a wave by shuffles, the four waves through LDS
    __syncthreads();
    double* Ps = (double*)xpw_lds;                               // [4 waves][64 columns][3]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double tg = sg[k], t1 = sx1[k], t2 = TWO ? sx2[TWO ? k : 0] : 0.0;
        tg += __shfl_xor(tg, 16, 64); t1 += __shfl_xor(t1, 16, 64); t2 += __shfl_xor(t2, 16, 64);
        tg += __shfl_xor(tg, 32, 64); t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
        if (q == 0) {
            double* o = Ps + ((wave * BN) + 4 * l15 + k) * 3;
            o[0] = tg; o[1] = t1; o[2] = t2;
        }
    }
    __syncthreads();
    if (tid < BN) {
        double tg = 0.0, t1 = 0.0, t2 = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { tg += Ps[(w * BN + tid) * 3]; t1 += Ps[(w * BN + tid) * 3 + 1]; t2 += Ps[(w * BN + tid) * 3 + 2]; }
        const long o = ((long)p * N + n0 + tid) * 2;
        fz.part1[o] = tg;
        fz.part1[o + 1] = t1;
        if (TWO) { fz.part2[o] = tg; fz.part2[o + 1] = t2; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same skeleton for the FORWARD of the short-K pointwise layers (K = Cin = 64 / 128 / 256: conv1 / conv3 / downsample of
// layer1 and layer2, /root/reference/timm/models/resnest.py:111-143, resnet.py:292-306) with the fused BatchNorm statistics:
// y = x W, per column sum y and sum y^2 in fp64.  These are output streams (64 -> 256 at 56 x 56, batch 70: 225 MB written,
// 56 MB read, 7.2 GFLOP) that pwp_kernel (exact-fp32 MFMA, one workgroup per CU, 4-byte stores per lane) runs at 2.75 TB/s.
// W (HWIO [k][n]) is transposed and split three-way into the resident [n][k] planes once per workgroup; no epilogue operands,
// so two workgroups per CU fit for every K.  part: [wg_per_col][N][2] fp64.
template <int KS, bool STATS>
__global__ __launch_bounds__(256, (KS <= 2 ? 2 : 1)) void xpw_fwd_kernel(const float* __restrict__ src, const float* __restrict__ wgt,
                                                                         float* __restrict__ dst, double* __restrict__ part, long M,
                                                                         int N, int mtiles, int wg_per_col) {
    constexpr int K = 64 * KS, BN = 64, RB = 2 * K;
    constexpr int PLANE = BN * RB, WBYTES = 3 * PLANE, SLOT = 64 * 128, NST = 2 * KS;
    constexpr int CHM = (K / 8 < 16 ? K / 8 : 16) - 1;
    extern __shared__ __attribute__((aligned(1024))) char xpw_lds[];
    char* Wl = xpw_lds;
    char* ring = xpw_lds + WBYTES;

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int colgroups = N / BN;
    const int L = blockIdx.x;
    const int cg = (L >> 3) % colgroups, p = (L & 7) + 8 * ((L >> 3) / colgroups);
    const int n0 = cg * BN;
    const int n_my = p < mtiles ? (mtiles - p + wg_per_col - 1) / wg_per_col : 0;
    const int nstage4 = (n_my * NST + 3) / 4;

    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)(M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, (unsigned)(M * N * 4), 0x00020000);
    unsigned a_voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 8 * (wave + 4 * j) + (lane >> 3), c = (lane & 7) ^ (r & 7);
        a_voff[j] = (unsigned)((r * K + c * 4) * 4);
    }
    auto issue = [&](int G, int slot) __attribute__((always_inline)) {
        const int ti = G / NST, kc = G % NST;
        const long m0 = (long)__builtin_amdgcn_readfirstlane(p + ti * wg_per_col) * 64;
        const long off = (m0 * K + kc * 32) * 4;
        const int soff = off > 0x7fffffffL ? 0x7fffffff : (int)off;
#pragma unroll
        for (int j = 0; j < 2; ++j) xpw_dma16(rs_a, ring + slot * SLOT + (wave + 4 * j) * 1024, a_voff[j], soff);
    };
    issue(0, 0); issue(1, 1); issue(2, 2); issue(3, 3);
    // ---- W[k][n0 .. n0 + 63] -> planes [n][k]: a thread gathers eight k-values of one column (lanes = consecutive columns)
    for (int e = tid; e < BN * (K / 8); e += 256) {
        const int n = e % BN, c = e / BN;
        xpw_u16x8 ph, pm, pl;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned short a, b, cc;
            split3_bf16(wgt[(long)(8 * c + i) * N + n0 + n], a, b, cc);
            ph[i] = a; pm[i] = b; pl[i] = cc;
        }
        char* d = Wl + n * RB + ((c ^ ((n >> 2) & CHM)) << 4);
        *(xpw_u16x8*)(d) = ph;
        *(xpw_u16x8*)(d + PLANE) = pm;
        *(xpw_u16x8*)(d + 2 * PLANE) = pl;
    }

    const int arow = 16 * wave + l15;
    const char* a_base = ring + arow * 128;
    const int a_sw = arow & 7;
    const char* b_base = Wl + 4 * l15 * RB;
    const int b_t = q ^ (l15 & CHM);
    const int col0 = n0 + 4 * l15;
    const int row_in = 16 * wave + 4 * q;
    const unsigned o_voff = (unsigned)(((long)row_in * N + col0) * 4);

    f32x4 acc[4], accl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; accl[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};

#define SB() __builtin_amdgcn_sched_barrier(0)
    auto epilogue = [&](int ti) __attribute__((always_inline)) {
        const long m0 = (long)__builtin_amdgcn_readfirstlane(p + ti * wg_per_col) * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f32x4 out;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float v = acc[k][e] + accl[k][e];
                out[k] = v;
                if (STATS) {                                     // (rows beyond M: their A rows read zeros, v = 0)
                    const double d = (double)v;
                    s1[k] += d;
                    s2[k] = __builtin_fma(d, d, s2[k]);
                }
                acc[k][e] = 0.f;
                accl[k][e] = 0.f;
            }
            // (row offset in the per-lane offset, scalar offset an immediate 0: see the input-gradient kernel above)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(xpw_u32x4, out), rs_o,
                                                   o_voff + (unsigned)((m0 + e) * (long)N * 4), 0, 0);
        }
    };
    auto stage_body = [&](int G, int slot, int kc) __attribute__((always_inline)) {
        const bool last = kc == NST - 1 && G / NST < n_my;
        const char* As = a_base + slot * SLOT;
        const f32x4 a0 = *(const f32x4*)(As + (((2 * q) ^ a_sw) << 4)), a1 = *(const f32x4*)(As + (((2 * q + 1) ^ a_sw) << 4));
        xpw_bf16x8 fb[3][4];
        const char* bp = b_base + (((kc * 4) ^ b_t) << 4);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[pl][j] = *(const xpw_bf16x8*)(bp + pl * PLANE + j * RB);
        xpw_u16x8 h, m, l;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned short a, b, c;
            split3_bf16(i < 4 ? a0[i & 3] : a1[i & 3], a, b, c);
            h[i] = a; m[i] = b; l[i] = c;
        }
        const xpw_bf16x8 fa[3] = {__builtin_bit_cast(xpw_bf16x8, h), __builtin_bit_cast(xpw_bf16x8, m), __builtin_bit_cast(xpw_bf16x8, l)};
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (pr < 5) accl[j] = xpw_mfma(fa[PA[pr]], fb[PB[pr]][j], accl[j]);
                else acc[j] = xpw_mfma(fa[PA[pr]], fb[PB[pr]][j], acc[j]);
            }
        SB();
        // the next stage's DMA has landed: two younger stages (4 instructions) -- and, behind an epilogue, its four stores, which
        // are younger than all three -- may stay in flight
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (last) epilogue(G / NST);
        __builtin_amdgcn_s_barrier();
        SB();
        issue(G + 4, slot);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int base = 0; base < nstage4; ++base) {
#pragma unroll
        for (int S = 0; S < 4; ++S) stage_body(4 * base + S, S, NST <= 4 ? S % NST : (4 * base + S) % NST);
    }
#undef SB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (STATS) {
        __syncthreads();
        double* Ps = (double*)xpw_lds;                           // [4 waves][64 columns][2]
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double t1 = s1[k], t2 = s2[k];
            t1 += __shfl_xor(t1, 16, 64); t2 += __shfl_xor(t2, 16, 64);
            t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
            if (q == 0) { double* o = Ps + ((wave * BN) + 4 * l15 + k) * 2; o[0] = t1; o[1] = t2; }
        }
        __syncthreads();
        if (tid < BN) {
            double t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { t1 += Ps[(w * BN + tid) * 2]; t2 += Ps[(w * BN + tid) * 2 + 1]; }
            const long o = ((long)p * N + n0 + tid) * 2;
            part[o] = t1;
            part[o + 1] = t2;
        }
    }
}
