// Tap-fused weight gradient of the 3x3 / stride 1 / pad 1 layers with 32 input channels per group (the deep stem's 32 -> 32
// and 32 -> 64 convolutions, resnet.py:471-489; layer1's radix convolutions, 64 -> 128 in two groups, split_attn.py:54-60) on
// the bf16 matrix cores with the exact three-way split done in REGISTERS (round 6; the library's own plan -- plan_hint < 0 -- of
// scouter_conv2d_wgrad_f32 for these shapes; SCOUTER_XWT=0: wgrad_taps_kernel).
//
//   dW[tap][ci][co] = sum_m X[m + tapoff(tap)][ci] * dY[m][co]                    (pixels m in flat NHWC order)
//
// These were the last large block on the exact-fp32 MFMA pipe (wgrad_taps_kernel: 0.8 ms per step at 95-110 TFLOP/s) and the
// END of the step -- the stem's weight gradients are what still runs when the compute stream is done.  The plane kernel's
// scheme (conv_planes_wgrad_taps.h) with fp32 operands instead of pre-split planes:
//   * X rows live in an LDS RING of 512 pixel rows per plane ([row][32 ch] bf16, 64-byte rows): in the flat pixel index the
//     taps are the row offsets (r - 1) W + (q - 1), so a 64-pixel chunk needs rows m0 - W - 1 .. m0 + 64 + W (W <= 112: five
//     64-row units), consecutive chunks share all but one unit -- every X row is fetched, split (hi + mid + lo, exact) and
//     stored ONCE per workgroup; dY rows once per chunk (two stages); both as plain 16-byte global loads requested before the
//     chunk's MFMAs and split / stored behind them: one barrier per chunk;
//   * fragments are transposing reads (ds_read_b64_tr_b16: each lane addresses one pixel row and receives four pixels of one
//     channel -- the k-contiguous operand v_mfma_f32_32x32x16_bf16 wants); a lane whose (pixel, tap) falls outside the image
//     points at a ZERO ROW instead (no masks in the data);
//   * the 64 pixels of a chunk are four k-steps: the 4 waves split them (BN = 32: one k-step each; BN = 64: two, by output
//     column half), every wave carries all nine taps -- nine 32 x 32 accumulators, the six products of a tap smallest first
//     on ONE accumulator (a second set for the corrections does not fit the 256 AGPRs) -- and the waves' sums meet in LDS at
//     the end (fixed order);
//   * one wave per SIMD, so what is not issued BETWEEN MFMAs leaves the matrix pipe idle: a tap's fragments are requested one
//     tap ahead, and sched_group_barrier puts one fragment read, one LDS store of the next chunk's rows and a share of the
//     split / address arithmetic behind each of the six MFMAs of a tap; no branch inside the chunk (a branch ends the
//     scheduling region); ONE round of 256 long pixel ranges (147 KB of LDS = one workgroup per CU, five-unit prologue each).
// Measured (tools_dev/xwt_check.py, batch 70): stem 32 -> 32 191 -> 118 us, 32 -> 64 300 -> 194, layer1's radix convolution
// 174 -> 100 (137 / 167 / 162 TFLOP/s; first version, reads in front of their MFMAs and 1 024 short ranges: 163 / 283).
// Same slab / split-K output format as wgrad_taps_kernel (deterministic; slab_reduce_kernel or the arrival counters).
#pragma once
#include "conv_common.h"

typedef short xw_v4i16 __attribute__((ext_vector_type(4)));
typedef short xw_v8i16 __attribute__((ext_vector_type(8)));
typedef __bf16 xw_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short xw_u16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ xw_bf16x8 xw_frag(const char* p0, const char* p1) {       // two transposing reads = 8 k of one row
    const xw_v4i16 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) xw_v4i16*)p0);
    const xw_v4i16 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) xw_v4i16*)p1);
    const xw_v8i16 ab = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(xw_bf16x8, ab);
}
__device__ __forceinline__ void xw_split_store(char* base, long plane_bytes, const f32x4& v) {   // 4 elements -> 3 x 8 bytes
    xw_u16x4 hi, mid, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned short h, m, l;
        split3_bf16(v[e], h, m, l);
        hi[e] = h; mid[e] = m; lo[e] = l;
    }
    *(xw_u16x4*)base = hi;
    *(xw_u16x4*)(base + plane_bytes) = mid;
    *(xw_u16x4*)(base + 2 * plane_bytes) = lo;
}

#ifndef XWT_NACC
#define XWT_NACC 1
#endif
#define XWT_CH 64
#define XWT_RING 512
#define XWT_HALO 128
#define XWT_XROW 64
#define XWT_XPL ((XWT_RING + 1) * XWT_XROW)
static size_t xwgrad_taps_lds_bytes(int bn) { return (size_t)3 * XWT_XPL + (size_t)2 * 3 * XWT_CH * bn * 2; }

template <int BN>       // output-channel tile (32 | 64); the input-channel tile is the whole group (32)
__global__ __launch_bounds__(256, 1) void xwgrad_taps_kernel(const float* __restrict__ act, const float* __restrict__ dy,
                                                             float* __restrict__ out, ConvGeom g, int co_tiles,
                                                             long pix_per_split, long slab, float* __restrict__ dw,
                                                             unsigned* __restrict__ arrival) {
    constexpr int CH = XWT_CH, RING = XWT_RING, HALO = XWT_HALO, XROW = XWT_XROW, XPL = XWT_XPL;
    constexpr int DYROW = BN * 2, DYPL = CH * DYROW, DYST = 3 * DYPL;
    constexpr int NWN = BN / 32, WK = 4 / NWN, KSW = 4 / WK;      // waves along co / along k; k-steps per wave and chunk
    constexpr int DQ = BN / 16;                                  // dY float4 per thread and chunk
    extern __shared__ __attribute__((aligned(1024))) char lds_raw[];
    char* Xl = lds_raw;
    char* Dl = lds_raw + 3 * XPL;

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % NWN, wk = wave / NWN;
    int bid, split_id;
    wgrad_block_coords(bid, split_id);
    const int tile_id = bid;
    const int co_t = bid % co_tiles, grp = bid / co_tiles;
    const int co0 = co_t * BN;
    const long mbeg = (long)split_id * pix_per_split;            // multiple of 64
    long mend = mbeg + pix_per_split;
    if (mend > g.M) mend = g.M;
    const int KT = (int)((mend - mbeg + CH - 1) / CH);
    const int W = g.W, H = g.H;

    // zero rows (one per plane, behind its ring)
    if (tid < 3 * 16) *(float*)(Xl + (tid >> 4) * XPL + RING * XROW + (tid & 15) * 4) = 0.f;

    // ---- loads: X unit u = pixel rows mbeg - 128 + 64 u .. + 63 (two float4 per thread), dY chunk c (DQ float4 per thread).
    // Rows outside the tensor / the split load a valid row and are dropped (no predicated loads).
    const float* xsrc = act + grp * g.Cg;
    const float* ysrc = dy + grp * g.Ng + co0;
    // Two register sets (A / B) hold the NEXT chunk's rows (requested a whole chunk ahead: an HBM round trip is longer than the
    // first taps of a chunk) -- the set loaded during chunk c - 1 is split and stored piece by piece BETWEEN the taps of chunk c.
    constexpr int NPIECE = 2 + DQ;                                // pieces of a set: two X float4, DQ dY float4
    struct RowSet { f32x4 x[2]; f32x4 d[DQ]; };
    auto load_x_to = [&](int u, f32x4 (&r)[2]) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = tid + 256 * k, row = idx >> 3, q = idx & 7;
            long p = mbeg - HALO + 64L * u + row;
            p = p < 0 ? 0 : (p >= g.M ? g.M - 1 : p);
            r[k] = *(const f32x4*)(xsrc + p * g.C + 4 * q);
        }
    };
    auto store_x_piece = [&](int u, int k, const f32x4& r) {
        const int idx = tid + 256 * k, row = idx >> 3, q = idx & 7;
        const long p = mbeg - HALO + 64L * u + row;
        const f32x4 v = (p >= 0 && p < g.M) ? r : f32x4{0.f, 0.f, 0.f, 0.f};
        xw_split_store(Xl + ((64 * u + row) & (RING - 1)) * XROW + q * 8, XPL, v);
    };
    auto load_dy_to = [&](int c, f32x4 (&r)[DQ]) {
#pragma unroll
        for (int k = 0; k < DQ; ++k) {
            const int idx = tid + 256 * k, row = idx / (BN / 4), q = idx % (BN / 4);
            long p = mbeg + 64L * c + row;
            p = p >= g.M ? g.M - 1 : p;
            r[k] = *(const f32x4*)(ysrc + p * g.N + 4 * q);
        }
    };
    auto store_dy_piece = [&](int c, int k, const f32x4& r) {
        const int idx = tid + 256 * k, row = idx / (BN / 4), q = idx % (BN / 4);
        const long p = mbeg + 64L * c + row;
        const f32x4 v = p < mend ? r : f32x4{0.f, 0.f, 0.f, 0.f};
        xw_split_store(Dl + (c & 1) * DYST + row * DYROW + q * 8, DYPL, v);
    };
    // piece i of the set that holds (X unit c + 5, dY chunk c + 1)
    auto store_piece = [&](int c, int i, const RowSet& rs) __attribute__((always_inline)) {
        if (i < 2) store_x_piece(c + 5, i, rs.x[i]);
        else store_dy_piece(c + 1, i - 2, rs.d[i - 2]);
    };

    // ---- fragment geometry of a transposing read: lane -> pixel row r2 of a 4-row group, channel columns 16 g1 + 4 (l & 3) ..
    const int r2 = (lane >> 2) & 3, g1 = (lane >> 4) & 1, c4 = 4 * (lane & 3);
    const char* xlane = Xl + (16 * g1 + c4) * 2;
    const char* zrow = xlane + RING * XROW;
    const char* ylane = Dl + (wn * 32 + 16 * g1 + c4) * 2;
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) toff[t] = (t / 3 - 1) * W + (t % 3 - 1);
    // (y, x) of the lane's pixels of a chunk: p(j, u) = m0 + 16 (wk KSW + j) + 8 h + 4 u + r2, advanced by 64 pixels per chunk
    const int hw = H * W;
    int px[2 * KSW], py[2 * KSW], prel[2 * KSW];
#pragma unroll
    for (int k = 0; k < 2 * KSW; ++k) {
        prel[k] = 16 * (wk * KSW + (k >> 1)) + 8 * h + 4 * (k & 1) + r2;
        const long p = mbeg + prel[k];
        const int rem = (int)(p % hw);
        py[k] = rem / W;
        px[k] = rem - py[k] * W;
    }
    const int adv_x = CH % W, adv_y = (CH / W) % H;

    // (XWT_NACC = 2: the five correction products on their own accumulators -- 288 accumulator registers do not fit the 256
    //  AGPRs, and the compiler then moves 5-6 registers per MFMA between the files: 850 VALU per 54 MFMAs, VALU-bound)
    f32x16 acc[9], accl[XWT_NACC == 2 ? 9 : 1];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[t][e] = 0.f; if (XWT_NACC == 2) accl[XWT_NACC == 2 ? t : 0][e] = 0.f; }

    // ---- prologue: units 0 .. 4 and dY chunk 0 into LDS, the rows of chunk 1 into set A
    RowSet ra_, rb_;
    if (KT > 0) {               // (all loads in flight together: one round trip, not six)
        f32x4 x5[5][2], d0[DQ];
#pragma unroll
        for (int u = 0; u < 5; ++u) load_x_to(u, x5[u]);
        load_dy_to(0, d0);
        load_x_to(5, ra_.x); load_dy_to(1, ra_.d);
#pragma unroll
        for (int u = 0; u < 5; ++u) { store_x_piece(u, 0, x5[u][0]); store_x_piece(u, 1, x5[u][1]); }
#pragma unroll
        for (int k = 0; k < DQ; ++k) store_dy_piece(0, k, d0[k]);
    }
    __syncthreads();

    // one chunk: `cur` holds the rows of chunk c + 1 (stored between this chunk's taps), `nxt` receives those of chunk c + 2
    auto chunk = [&](int c, const RowSet& cur, RowSet& nxt) __attribute__((always_inline)) {
        // (no branches around the loads / the stores of the last chunks: a branch ends the scheduling region and the split
        //  arithmetic then sits in a block of its own, not between the MFMAs.  Loads are clamped to the tensor; what the last
        //  two chunks store lands in ring units / a dY stage nobody reads any more)
        load_x_to(c + 6, nxt.x);
        load_dy_to(c + 2, nxt.d);
        const char* dst = ylane + (c & 1) * DYST;
        const int rbase = 64 * c + HALO;
        // dY fragments of the wave's k-steps, tap validity of the lane's pixels (bit 3 r + q)
        xw_bf16x8 FB[KSW][3];
        unsigned vm[2 * KSW];
#pragma unroll
        for (int j = 0; j < KSW; ++j) {
            const char* yb = dst + (16 * (wk * KSW + j) + 8 * h + r2) * DYROW;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) FB[j][pl] = xw_frag(yb + pl * DYPL, yb + pl * DYPL + 4 * DYROW);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = 2 * j + u;
                const unsigned cb = (px[k] >= 1 ? 1u : 0u) | 2u | (px[k] <= W - 2 ? 4u : 0u);
                vm[k] = (py[k] >= 1 ? cb : 0u) | (cb << 3) | (py[k] <= H - 2 ? (cb << 6) : 0u);
            }
        }
        // X fragments of (k-step j, tap t): requested ONE TAP AHEAD of the six MFMAs that use them, the order pinned -- left to
        // itself the compiler puts the six transposing reads right in front of their MFMAs and waits lgkmcnt(0): an exposed LDS
        // round trip per tap, 2 k of a chunk's 4.6 k cycles when first measured
        constexpr int NSLOT = 9 * KSW;
        xw_bf16x8 FA[2][3];
        auto read_a = [&](int n, xw_bf16x8 (&F)[3]) __attribute__((always_inline)) {
            const int j = n / 9, t = n % 9;
            const unsigned ra = (unsigned)(rbase + prel[2 * j] + toff[t]) & (RING - 1);
            const unsigned rb = (unsigned)(rbase + prel[2 * j + 1] + toff[t]) & (RING - 1);
            const char* a0 = ((vm[2 * j] >> t) & 1u) ? xlane + ra * XROW : zrow;
            const char* a1 = ((vm[2 * j + 1] >> t) & 1u) ? xlane + rb * XROW : zrow;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) F[pl] = xw_frag(a0 + pl * XPL, a1 + pl * XPL);
        };
        read_a(0, FA[0]);
#pragma unroll
        for (int n = 0; n < NSLOT; ++n) {
            const int j = n / 9, t = n % 9;
            if (n + 1 < NSLOT) read_a(n + 1, FA[(n + 1) & 1]);
            const xw_bf16x8 (&F)[3] = FA[n & 1];
            // smallest products first; hi * hi on its own accumulator
            f32x16& al = XWT_NACC == 2 ? accl[XWT_NACC == 2 ? t : 0] : acc[t];
            al = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[0], FB[j][2], al, 0, 0, 0);
            al = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[2], FB[j][0], al, 0, 0, 0);
            al = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[1], FB[j][1], al, 0, 0, 0);
            al = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[0], FB[j][1], al, 0, 0, 0);
            al = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[1], FB[j][0], al, 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[0], FB[j][0], acc[t], 0, 0, 0);
            // one piece of the next chunk's rows per tap slot (split + three LDS stores under the MFMAs just issued)
            if (n >= 1 && (n - 1) % ((NSLOT - 1) / NPIECE) == 0 && (n - 1) / ((NSLOT - 1) / NPIECE) < NPIECE)
                store_piece(c, (n - 1) / ((NSLOT - 1) / NPIECE), cur);
            // one wave per SIMD: whatever is not issued BETWEEN the MFMAs leaves the matrix pipe idle -- each of the six MFMAs
            // of this tap is followed by one of the next tap's fragment reads, an LDS store of the piece and a share of the
            // address / split arithmetic
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the lane's pixels, one chunk on
#pragma unroll
        for (int k = 0; k < 2 * KSW; ++k) {
            px[k] += adv_x;
            const int wrap = px[k] >= W ? 1 : 0;
            px[k] -= wrap ? W : 0;
            py[k] += adv_y + wrap;
            py[k] -= py[k] >= H ? H : 0;
        }
        __syncthreads();
    };
    for (int c = 0; c < KT; ++c) {        // (one body: the set moves by sixteen / twenty-four register copies per chunk -- two
        chunk(c, ra_, rb_);               //  inlined bodies with the sets swapped spilled 299 registers)
        ra_ = rb_;
    }

    // ---- the k-split waves' sums meet in LDS tap by tap (fixed order); the operand images are dead
    float* o = out + (long)split_id * slab;
    float* red = (float*)lds_raw;                                   // [WK][32][BN]
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wk * 32 * BN + mfma32_row(e, lane) * BN + wn * 32 + l31] = XWT_NACC == 2 ? acc[t][e] + accl[XWT_NACC == 2 ? t : 0][e] : acc[t][e];
        __syncthreads();
        for (int e = tid; e < 32 * BN; e += 256) {
            float v = red[e];
#pragma unroll
            for (int k = 1; k < WK; ++k) v += red[k * 32 * BN + e];
            slab_store(o + ((long)t * g.Cg + e / BN) * g.N + grp * g.Ng + co0 + e % BN, v, arrival != nullptr);
        }
        __syncthreads();
    }
    if (arrival)        // the last workgroup of this tile sums the slabs itself (conv_common.h slab_tile_finish)
        slab_tile_finish(out, dw, slab, gridDim.y, arrival + tile_id, 9, (long)g.Cg * g.N, 0, 32, g.N, grp * g.Ng + co0, BN);
}
