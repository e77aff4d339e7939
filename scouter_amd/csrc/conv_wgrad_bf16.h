// bf16-input weight-gradient kernel (included by conv_igemm.hip, which owns the split-K plan and the slab reduction).
//   dW[tap][ci][co] = sum_m X[m + tapoff][ci] * dY[m][co]        ("same" stride-1 convolutions only; others use fp32)
// Both operands have the contraction index (pixel) as their slow axis in memory, while v_mfma_f32_32x32x16_bf16 wants 8
// consecutive k per lane: each thread loads a 4-pixel x 4-channel patch (four coalesced 16-byte loads), transposes it
// in registers, rounds to bf16 and writes four 8-byte rows of the TRANSPOSED LDS tile [channel][pixel]; fragments are
// then plain 16-byte reads.  Chunk = 64 pixels (one validity bit per lane for padded taps, made scalar by a ballot).
#pragma once
#include "conv_common.h"

typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x4 __attribute__((ext_vector_type(4)));

// A_BF16: the activation X is STORED as bf16 (round 4): the same 4-pixel x 4-channel patches as four 8-byte loads, no
// conversion -- the values are the ones the fp32 path rounds to.
// B_BF16: the same for dy.
template <int BM, int BN, int WM, int WN, int MODE, bool A_BF16 = false, bool B_BF16 = false>     // MODE 1: padded taps, 2: 1x1
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const void* __restrict__ act, const void* __restrict__ dy,
                                                            float* __restrict__ out, ConvGeom g, int ci_tiles,
                                                            int co_tiles, long pix_per_split, long slab, float* __restrict__ dw,
                                                            unsigned* __restrict__ arrival) {
    constexpr int KP = 64;                                 // pixels per chunk
    constexpr int LDP = KP + 8;                            // LDS row stride (bf16): 16-byte aligned rows
    constexpr int MT = WM / 32, NT = WN / 32, WAVES_N = BN / WN;
    constexpr int A_H = BM * LDP, B_H = BN * LDP, STAGE_H = A_H + B_H;
    constexpr int AP = BM / 64, BP = BN / 64;              // (channel-quad, pixel-quad) patches per thread
    constexpr int AES = A_BF16 ? 2 : 4, BES = B_BF16 ? 2 : 4;    // bytes per stored activation / gradient element
    static_assert((BM / WM) * (BN / WN) == 4 && BM % 64 == 0 && BN % 64 == 0, "tile config");
    extern __shared__ __attribute__((aligned(16))) __bf16 ldsw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int bid, split_id;
    wgrad_block_coords(bid, split_id);
    const int tile_id = bid;
    const int co_t = bid % co_tiles; bid /= co_tiles;
    const int ci_t = bid % ci_tiles; bid /= ci_tiles;
    const int grp = bid % g.groups;
    const int tap = bid / g.groups;
    const int r = tap / g.S, q = tap - r * g.S;
    const int ci0 = ci_t * BM, co0 = co_t * BN;
    const long mbeg = (long)split_id * pix_per_split;
    long mend = mbeg + pix_per_split;
    if (mend > g.M) mend = g.M;
    const int KT = (int)((mend - mbeg + KP - 1) / KP);
    const int hw = g.Ho * g.Wo;

    // validity walk: lane = chunk pixel
    int qx1 = 0, qy1 = 0;
    if (MODE == 1) {
        const int blk_b = (int)(((double)(unsigned long)mbeg + 0.5) * g.inv_hw);
        const int blk_rem = (int)(mbeg - (long)blk_b * hw);
        const int blk_y = fast_div(blk_rem, g.inv_wo), blk_x = blk_rem - blk_y * g.Wo;
        const int tx = blk_x + lane, qx = fast_div(tx, g.inv_wo);
        const int ty = blk_y + qx, qy = fast_div(ty, g.inv_ho);
        qx1 = tx - qx * g.Wo;
        qy1 = ty - qy * g.Ho;
    }
    const int adv_x = KP % g.Wo, adv_y = KP / g.Wo;
    const long tapoff = (long)(r - g.pad) * g.W + (q - g.pad);
    constexpr unsigned OOB = 0x80000000u;
    // patch p = tid + 256 i:  cq8 = p % 8 (8 channel quads = 128 contiguous bytes of a pixel), pq = (p / 8) % 16,
    // cblk = p / 128
    unsigned a_voff[AP], b_voff[BP];
    int a_row[AP], b_row[BP], a_pq[AP], b_pq[BP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int p = tid + 256 * i, cq = (p & 7) + 8 * (p >> 7), pq = (p >> 3) & 15;
        a_row[i] = 4 * cq; a_pq[i] = pq;
        a_voff[i] = (unsigned)((4 * pq * (long)g.C + grp * g.Cg + ci0 + 4 * cq) * AES);
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
        const int p = tid + 256 * i, cq = (p & 7) + 8 * (p >> 7), pq = (p >> 3) & 15;
        b_row[i] = 4 * cq; b_pq[i] = pq;
        b_voff[i] = (unsigned)((4 * pq * (long)g.N + grp * g.Ng + co0 + 4 * cq) * BES);
    }
    // bytes the descriptor may touch from pixel `first`: up to this split's last pixel and (ragged tiles read one pixel
    // past a valid one) never beyond the tensor's last pixel
    auto records = [&](long m_chunk, long first, int row_elems, int es) {
        long px = mend - m_chunk;
        if (px > g.M - first) px = g.M - first;
        const long n = px * (long)row_elems * es;
        return (unsigned)(n < 0 ? 0 : (n > 0x7fffffffL ? 0x7fffffffL : n));
    };
    f32x4 ra[AP][4], rb[BP][4];
    wg_bf16x4 rah[AP][4], rbh[BP][4];                      // (A_BF16 / B_BF16: the patches arrive as bf16)
    long a_m = mbeg, b_m = mbeg;
    auto load_a = [&]() {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((const char*)act + (a_m + tapoff) * g.C * AES), 0, records(a_m, a_m + tapoff, g.C, AES), 0x00020000);
        unsigned long long vmask = ~0ull;
        if (MODE == 1) {
            const int iy = qy1 - g.pad + r, ix = qx1 - g.pad + q;
            vmask = __ballot((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W);
            qx1 += adv_x;
            const int wrap = qx1 >= g.Wo ? 1 : 0;
            qx1 -= wrap ? g.Wo : 0;
            qy1 += adv_y + wrap;
            qy1 -= qy1 >= g.Ho ? g.Ho : 0;
        }
#pragma unroll
        for (int i = 0; i < AP; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                unsigned vo = a_voff[i] + (unsigned)(rr * g.C * AES);
                if (MODE == 1) vo = ((vmask >> (4 * a_pq[i] + rr)) & 1ull) ? vo : OOB;
                if constexpr (A_BF16) rah[i][rr] = __builtin_bit_cast(wg_bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rs, vo, 0, 0));
                else ra[i][rr] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
            }
        a_m += KP;
    };
    auto load_b = [&]() {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((const char*)dy + b_m * g.N * BES), 0, records(b_m, b_m, g.N, BES), 0x00020000);
#pragma unroll
        for (int i = 0; i < BP; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const unsigned vo = b_voff[i] + (unsigned)(rr * g.N * BES);
                if constexpr (B_BF16) rbh[i][rr] = __builtin_bit_cast(wg_bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rs, vo, 0, 0));
                else rb[i][rr] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
            }
        b_m += KP;
    };
    auto store_t = [&](__bf16* T, const f32x4 (&v)[4], int row, int pq) {      // 4x4 transpose, four 8-byte row writes
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            wg_bf16x4 o;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) o[rr] = (__bf16)v[rr][e];
            *(wg_bf16x4*)(T + (row + e) * LDP + 4 * pq) = o;
        }
    };
    auto store_th = [&](__bf16* T, const wg_bf16x4 (&v)[4], int row, int pq) {  // the same transpose of bf16 patches
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            wg_bf16x4 o;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) o[rr] = v[rr][e];
            *(wg_bf16x4*)(T + (row + e) * LDP + 4 * pq) = o;
        }
    };
    auto store_ab = [&](int buf) {
        __bf16* As = ldsw + buf * STAGE_H;
        __bf16* Bs = As + A_H;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            if constexpr (A_BF16) store_th(As, rah[i], a_row[i], a_pq[i]);
            else store_t(As, ra[i], a_row[i], a_pq[i]);
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            if constexpr (B_BF16) store_th(Bs, rbh[i], b_row[i], b_pq[i]);
            else store_t(Bs, rb[i], b_row[i], b_pq[i]);
        }
    };
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    auto compute = [&](int buf) {
        const __bf16* As = ldsw + buf * STAGE_H + (wm * WM + l31) * LDP + 8 * h;
        const __bf16* Bs = ldsw + buf * STAGE_H + A_H + (wn * WN + l31) * LDP + 8 * h;
        wg_bf16x8 fa[2][MT], fb[2][NT];                  // fragments of k-step s+1 are read while step s multiplies
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[0][i] = *(const wg_bf16x8*)(As + i * 32 * LDP);
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[0][j] = *(const wg_bf16x8*)(Bs + j * 32 * LDP);
#pragma unroll
        for (int s = 0; s < KP / 16; ++s) {
            if (s + 1 < KP / 16) {
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[(s + 1) & 1][i] = *(const wg_bf16x8*)(As + i * 32 * LDP + 16 * (s + 1));
#pragma unroll
                for (int j = 0; j < NT; ++j) fb[(s + 1) & 1][j] = *(const wg_bf16x8*)(Bs + j * 32 * LDP + 16 * (s + 1));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s & 1][i], fb[s & 1][j], acc[i][j], 0, 0, 0);
        }
    };
    if (KT > 0) {
        load_a(); load_b();
        store_ab(0);
        if (KT > 1) { load_a(); load_b(); }
    }
    __syncthreads();
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    auto chunk = [&](int kt, auto CUR) {
        constexpr int cur = decltype(CUR)::value, nxt = cur ^ 1;
        if (kt + 1 < KT) store_ab(nxt);
        if (kt + 2 < KT) { load_a(); load_b(); }
        compute(cur);
        __syncthreads();
    };
    for (int kt = 0; kt < KT; kt += 2) {
        chunk(kt, P0{});
        if (kt + 1 < KT) chunk(kt + 1, P1{});
    }
    float* o = out + (long)split_id * slab + (long)tap * g.Cg * g.N;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ci = ci0 + wm * WM + i * 32 + mfma32_row(e, lane);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int cl = co0 + wn * WN + j * 32 + l31, co = grp * g.Ng + cl;
                if (ci < g.Cg && cl < g.Ng)             // (ragged tiles over 32-channel groups: the rest is not this group's)
                    slab_store(o + (long)ci * g.N + co, acc[i][j][e], arrival != nullptr);
            }
        }
    if (arrival)
        slab_tile_finish(out + (long)tap * g.Cg * g.N, dw + (long)tap * g.Cg * g.N, slab, gridDim.y, arrival + tile_id, 1, 0,
                         (long)ci0 * g.N, BM, g.N, grp * g.Ng + co0, BN);
}
