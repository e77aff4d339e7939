// Tap-fused fp32 weight gradient for 3x3, stride-1, pad-1 convolutions with 32 input channels per group (the ResNeSt
// stem and its first grouped layer).  The generic kernel gives every filter tap its own blocks, so the activation and
// gradient rows of a pixel range are fetched nine times (PMC: 1.9 GB per launch for 32->32 @112x112, batch 70 -- 8x the
// algorithmic bytes, i.e. fabric-bandwidth bound at 60 TFLOP/s).  Here ONE block owns a pixel range and accumulates
// all nine taps: dY is loaded once per 32-pixel chunk, the nine shifted activation tiles are fetched back to back by
// the same block (L1/L2 hits) and both stay in LDS for 9 x fewer passes.
//   grid (co-tiles x groups, pixel splits);  block = 4 waves = (BN/32 column waves) x (WK k-split waves)
//   per chunk: 9 buffer loads of X (tap offsets in SGPRs, validity from four ballots: row/column edge bits),
//   1-2 loads of dY, one LDS stage, 9 x (16/WK) x (MFMA, 1-2 LDS reads)
#pragma once
#include "conv_common.h"

template <int BN>       // output-channel tile; input-channel tile is the whole group (32)
__global__ __launch_bounds__(256, 2) void wgrad_taps_kernel(const float* __restrict__ act, const float* __restrict__ dy,
                                                            float* __restrict__ out, ConvGeom g, int co_tiles,
                                                            long pix_per_split, long slab, float* __restrict__ dw,
                                                            unsigned* __restrict__ arrival) {
    constexpr int BM = 32, CH = 32;                    // chunk = 32 pixels = 16 MFMA k-steps of 2
    constexpr int NWN = BN / 32, WK = 4 / NWN, SPW = 16 / WK, BI = BN / 32;
    constexpr int A_T = CH * BM;                       // floats per tap tile
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float ldst[];     // As[9][32][32] | Bs[32][BN]
    float* As = ldst;
    float* Bs = ldst + 9 * A_T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int wn = wave % NWN, wk = wave / NWN;
    int bid, split_id;
    wgrad_block_coords(bid, split_id);
    const int tile_id = bid;
    const int co_t = bid % co_tiles, grp = bid / co_tiles;
    const int co0 = co_t * BN;
    const long mbeg = (long)split_id * pix_per_split;
    long mend = mbeg + pix_per_split;
    if (mend > g.M) mend = g.M;
    const int KT = (int)((mend - mbeg + CH - 1) / CH);
    const int hw = g.Ho * g.Wo;

    // pixel walk of chunk pixel l31 (edge bits only; addresses are affine in the pixel index)
    const int blk_b = (int)(((double)(unsigned long)mbeg + 0.5) * g.inv_hw);
    const int blk_rem = (int)(mbeg - (long)blk_b * hw);
    const int blk_y = fast_div(blk_rem, g.inv_wo), blk_x = blk_rem - blk_y * g.Wo;
    const int tx0 = blk_x + l31, qx0 = fast_div(tx0, g.inv_wo);
    const int ty0 = blk_y + qx0, qy0 = fast_div(ty0, g.inv_ho);
    int qx1 = tx0 - qx0 * g.Wo, qy1 = ty0 - qy0 * g.Ho;
    const int adv_x = CH % g.Wo, adv_y = CH / g.Wo;

    const int prow = tid >> 3;                                     // A: thread = (pixel row, channel quad)
    const unsigned a_voff = (unsigned)(((long)prow * g.C + grp * g.Cg + (tid & 7) * 4) * 4);
    const unsigned a_bit = 1u << prow;
    unsigned b_voff[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int c = tid + 256 * i;
        b_voff[i] = (unsigned)(((long)(c / (BN / 4)) * g.N + grp * g.Ng + co0 + (c % (BN / 4)) * 4) * 4);
    }
    auto records = [&](long m_chunk, int row_elems) {
        long n = (mend - m_chunk) * (long)row_elems * 4;
        return (unsigned)(n < 0 ? 0 : (n > 0x7fffffffL ? 0x7fffffffL : n));
    };
    f32x4 ra[9], rb[BI];
    long cur_m = mbeg;
    auto load = [&]() {
        // descriptor base = pixel (m - W - 1): tap (r, q) is then the non-negative scalar offset (r W + q) pixels
        // (the hardware range check adds the scalar offset, so the end of the pixel range cannot be expressed through
        // num_records here: rows past `mend` are masked with the tap bits instead)
        const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(act + (cur_m - g.W - 1) * g.C), 0, 0x7fffffff, 0x00020000);
        const long left_rows = mend - cur_m;
        const unsigned rows = left_rows >= 32 ? 0xffffffffu : ((1u << (int)left_rows) - 1u);
        const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + cur_m * g.N), 0,
                                                                             records(cur_m, g.N), 0x00020000);
        const unsigned up = (unsigned)__ballot(qy1 >= 1), down = (unsigned)__ballot(qy1 + 1 < g.H);
        const unsigned left = (unsigned)__ballot(qx1 >= 1), right = (unsigned)__ballot(qx1 + 1 < g.W);
        qx1 += adv_x;
        const int wrap = qx1 >= g.Wo ? 1 : 0;
        qx1 -= wrap ? g.Wo : 0;
        qy1 += adv_y + wrap;
        qy1 -= qy1 >= g.Ho ? g.Ho : 0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const unsigned ok = rows & (r == 0 ? up : r == 2 ? down : 0xffffffffu) & (q == 0 ? left : q == 2 ? right : 0xffffffffu);
                const unsigned vo = (ok & a_bit) ? a_voff : OOB;
                ra[3 * r + q] = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa, vo, (r * g.W + q) * g.C * 4, 0));
            }
#pragma unroll
        for (int i = 0; i < BI; ++i)
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsb, b_voff[i], 0, 0));
        cur_m += CH;
    };
    auto store = [&]() {
#pragma unroll
        for (int t = 0; t < 9; ++t) *(f32x4*)(As + t * A_T + tid * 4) = ra[t];
#pragma unroll
        for (int i = 0; i < BI; ++i) *(f32x4*)(Bs + (tid + 256 * i) * 4) = rb[i];
    };
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    auto compute = [&]() {
#pragma unroll
        for (int ss = 0; ss < SPW; ++ss) {
            const int k = h * 16 + wk * SPW + ss;
            const float fb = Bs[k * BN + wn * 32 + l31];
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = mfma32(As[t * A_T + k * BM + l31], fb, acc[t]);
        }
    };
    // single LDS stage, registers hold the next chunk: [store] sync [issue next loads] compute sync
    if (KT > 0) load();
    for (int kt = 0; kt < KT; ++kt) {
        store();
        __syncthreads();
        if (kt + 1 < KT) load();
        compute();
        __syncthreads();
    }
    // cross-wave (k-split) reduction tap by tap through LDS, fixed order; As is free now
    float* o = out + (long)split_id * slab;
    float* red = ldst;                                   // [WK][32][BN]
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wk * BM * BN + mfma32_row(e, lane) * BN + wn * 32 + l31] = acc[t][e];
        __syncthreads();
        for (int e = tid; e < BM * BN; e += 256) {
            float v = red[e];
#pragma unroll
            for (int k = 1; k < WK; ++k) v += red[k * BM * BN + e];
            slab_store(o + ((long)t * g.Cg + e / BN) * g.N + grp * g.Ng + co0 + e % BN, v, arrival != nullptr);
        }
        __syncthreads();
    }
    if (arrival)        // the last workgroup of this tile sums the slabs itself (conv_common.h slab_tile_finish)
        slab_tile_finish(out, dw, slab, gridDim.y, arrival + tile_id, 9, (long)g.Cg * g.N, 0, BM, g.N, grp * g.Ng + co0, BN);
}
