// Tap-fused weight gradient of the 3x3 / stride 1 / pad 1 layers with 32 input channels per group when BOTH operands are STORED as
// bf16 (BASELINE configs[4]: --precision bf16 with bf16 activation / gradient storage; the deep stem's 32 -> 32 and 32 -> 64
// convolutions, resnet.py:471-489, and layer1's radix convolutions, split_attn.py:54-60).  The library's own plan (plan_hint < 0)
// of scouter_conv2d_wgrad_bf16_io for these shapes; SCOUTER_BWT=0: wgrad_bf16_kernel.
//
//   dW[tap][ci][co] = sum_m X[m + tapoff(tap)][ci] * dY[m][co]                    (pixels m in flat NHWC order)
//
// wgrad_bf16_kernel runs one workgroup per TAP on ragged 64-wide tiles: X and dY are read nine times (5 x 416 us per step at batch
// 256, 2.2 TB/s of algorithmic bytes).  This is conv_wgrad_taps_x3.h without the split -- one bf16 plane, one MFMA per (tap, k-step):
//   * X rows live in an LDS RING of 512 pixel rows ([row][32 ch] bf16, 64-byte rows): a 64-pixel chunk needs rows m0 - W - 1 ..
//     m0 + 64 + W (five 64-row units, W <= 112), consecutive chunks share all but one unit -- every X row is fetched ONCE per
//     workgroup, dY rows once per chunk (four stages); both by LDS-DMA requested THREE chunks ahead (a chunk is 18 MFMAs per
//     wave -- a sixth of the x3 kernel's -- so one chunk of distance does not cover an HBM round trip, and DMA holds no registers);
//   * fragments are transposing reads (ds_read_b64_tr_b16); a lane whose (pixel, tap) falls outside the image points at a zero row;
//   * the 64 pixels of a chunk are four k-steps, split over the 4 waves as in the x3 kernel (BN = 32: one k-step each; BN = 64:
//     two, by output-column half); every wave carries all nine taps (nine 32 x 32 accumulators), the waves' sums meet in LDS at
//     the end (fixed order);
//   * 65 KB of LDS and 256 registers: TWO workgroups per CU (the second one's MFMAs cover the first one's fragment reads and
//     barrier), 512 pixel ranges.
// Same slab / split-K output format as wgrad_taps_kernel (deterministic; slab_reduce_kernel or the arrival counters).
#pragma once
#include "conv_common.h"
#include "conv_wgrad_taps_x3.h"

#define BWT_CH 64
#define BWT_RING 512
#define BWT_HALO 128
#define BWT_XROW 64
#define BWT_XPL ((BWT_RING + 1) * BWT_XROW)
static size_t bwgrad_taps_lds_bytes(int bn) { return (size_t)BWT_XPL + (size_t)4 * BWT_CH * bn * 2; }

// (a NON-template helper on purpose: the builtin inside a kernel template makes hipcc's host pass drop the kernel's launch stub)
__device__ __forceinline__ void bwt_dma16(__amdgpu_buffer_rsrc_t rs, char* lds, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// Transposing LDS reads as inline asm, waited for by hand (lgkmcnt): hipcc's counter tracking puts `s_waitcnt vmcnt(0)` in front of
// every LDS read it can see behind an LDS-DMA whose target it cannot tell apart (here: all of them, one dynamic LDS block) -- with
// the builtin reads every chunk waited for the DMA it had just requested (first version: 2.2 us per 64-pixel chunk, the HBM round
// trip).  The wait takes the fragment as an in / out operand, so the MFMA that consumes it cannot move above it.
__device__ __forceinline__ xw_bf16x8 bwt_frag(unsigned a0, unsigned a1) {
    xw_v4i16 a, b;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a) : "v"(a0) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b) : "v"(a1) : "memory");
    const xw_v8i16 ab = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(xw_bf16x8, ab);
}
template <int N>
__device__ __forceinline__ void bwt_wait(xw_bf16x8& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N) : "memory"); }

template <int BN>       // output-channel tile (32 | 64); the input-channel tile is the whole group (32)
__global__ __launch_bounds__(256, 2) void bwgrad_taps_kernel(const unsigned short* __restrict__ act,
                                                             const unsigned short* __restrict__ dy, float* __restrict__ out,
                                                             ConvGeom g, int co_tiles, long pix_per_split, long slab,
                                                             float* __restrict__ dw, unsigned* __restrict__ arrival) {
    constexpr int CH = BWT_CH, RING = BWT_RING, HALO = BWT_HALO, XROW = BWT_XROW, XPL = BWT_XPL;
    constexpr int DYROW = BN * 2, DYST = CH * DYROW;
    constexpr int NWN = BN / 32, WK = 4 / NWN, KSW = 4 / WK;      // waves along co / along k; k-steps per wave and chunk
    constexpr int DQ = BN / 32;                                  // dY 16-byte pieces per thread and chunk
    extern __shared__ __attribute__((aligned(1024))) char lds_raw[];
    char* Xl = lds_raw;
    char* Dl = lds_raw + XPL;

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

    if (tid < 16) *(float*)(Xl + RING * XROW + tid * 4) = 0.f;   // the zero row behind the ring

    // ---- LDS-DMA (buffer_load ... lds, 16 bytes per lane, 1 KB per wave and instruction): X unit u = pixel rows mbeg - 128 + 64 u
    // .. + 63 -> ring slot u & 7 (one instruction per wave: 16 rows), dY chunk c -> stage c & 3 (DQ instructions per wave).  X rows
    // outside the tensor are clamped to a valid row (only masked taps / zero dY rows ever meet them); dY rows beyond the split's last
    // pixel are beyond the descriptor and read zeros.
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(act + grp * g.Cg), 0, 0x7fffffff, 0x00020000);
    const long dy_col = (long)grp * g.Ng + co0;
    const __amdgpu_buffer_rsrc_t rsy =
        __builtin_amdgcn_make_buffer_rsrc((void*)(dy + dy_col), 0, (unsigned)((mend * g.N - dy_col) * 2), 0x00020000);
    const int xrow = 16 * wave + (lane >> 2), xq = lane & 3;
    auto dma_x = [&](int u) {
        long p = mbeg - HALO + 64L * u + xrow;
        p = p < 0 ? 0 : (p >= g.M ? g.M - 1 : p);
        bwt_dma16(rsx, Xl + ((64 * u) & (RING - 1)) * XROW + wave * 1024, (unsigned)(p * g.C * 2 + xq * 16), 0);
    };
    auto dma_dy = [&](int c) {
#pragma unroll
        for (int k = 0; k < DQ; ++k) {
            const int inst = wave * DQ + k;
            const int row = inst * (1024 / DYROW) + lane / (DYROW / 16), q = lane % (DYROW / 16);
            const long p = mbeg + 64L * c + row;
            bwt_dma16(rsy, Dl + (c & 3) * DYST + inst * 1024, (unsigned)(p * g.N * 2 + q * 16), 0);
        }
    };

    // ---- fragment geometry of a transposing read: lane -> pixel row r2 of a 4-row group, channel columns 16 g1 + 4 (l & 3) ..
    const int r2 = (lane >> 2) & 3, g1 = (lane >> 4) & 1, c4 = 4 * (lane & 3);
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)lds_raw;
    const unsigned xlane = lds0 + (16 * g1 + c4) * 2;
    const unsigned zrow = xlane + RING * XROW;
    const unsigned ylane = lds0 + XPL + (wn * 32 + 16 * g1 + c4) * 2;
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

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // ---- prologue: units 0 .. 4 and dY chunk 0, then the two groups that fly ahead (unit 5 + chunk 1, unit 6 + chunk 2)
#pragma unroll
    for (int u = 0; u < 5; ++u) dma_x(u);
    dma_dy(0);
    dma_x(5); dma_dy(1);
    dma_x(6); dma_dy(2);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * (1 + DQ)) : "memory");      // (lgkmcnt: the zero row's store)
    __builtin_amdgcn_s_barrier();              // (not __syncthreads(): its fence makes hipcc drain vmcnt to 0)

    // one chunk: the DMA of unit c + 7 / dY chunk c + 3 goes out first (three chunks of distance: an HBM round trip is longer than
    // a chunk's 18 MFMAs); what the last chunks request lands in ring units / dY stages nobody reads any more
    auto chunk = [&](int c) __attribute__((always_inline)) {
        dma_x(c + 7);
        dma_dy(c + 3);
        const unsigned dst = ylane + (c & 3) * DYST;
        const int rbase = 64 * c + HALO;
        xw_bf16x8 FB[KSW];
        unsigned vm[2 * KSW];
#pragma unroll
        for (int j = 0; j < KSW; ++j) {
            const unsigned yb = dst + (16 * (wk * KSW + j) + 8 * h + r2) * DYROW;
            FB[j] = bwt_frag(yb, yb + 4 * DYROW);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = 2 * j + u;
                const unsigned cb = (px[k] >= 1 ? 1u : 0u) | 2u | (px[k] <= W - 2 ? 4u : 0u);
                vm[k] = (py[k] >= 1 ? cb : 0u) | (cb << 3) | (py[k] <= H - 2 ? (cb << 6) : 0u);
            }
        }
        constexpr int NSLOT = 9 * KSW, AHEAD = 3;                 // fragments requested three taps ahead of their MFMA
        xw_bf16x8 FA[AHEAD + 1];
        auto read_a = [&](int n, xw_bf16x8& F) __attribute__((always_inline)) {
            const int j = n / 9, t = n % 9;
            const unsigned ra = (unsigned)(rbase + prel[2 * j] + toff[t]) & (RING - 1);
            const unsigned rb = (unsigned)(rbase + prel[2 * j + 1] + toff[t]) & (RING - 1);
            const unsigned a0 = ((vm[2 * j] >> t) & 1u) ? xlane + ra * XROW : zrow;
            const unsigned a1 = ((vm[2 * j + 1] >> t) & 1u) ? xlane + rb * XROW : zrow;
            F = bwt_frag(a0, a1);
        };
#pragma unroll
        for (int n = 0; n < AHEAD; ++n) read_a(n, FA[n]);
#pragma unroll
        for (int n = 0; n < NSLOT; ++n) {
            const int j = n / 9, t = n % 9;
            if (n + AHEAD < NSLOT) read_a(n + AHEAD, FA[(n + AHEAD) % (AHEAD + 1)]);
            xw_bf16x8& F = FA[n % (AHEAD + 1)];
            // the reads of slots n + 1 .. n + AHEAD (two each) may stay in flight; LDS operations retire in order
            if (NSLOT - 1 - n >= 3) bwt_wait<6>(F);
            else if (NSLOT - 1 - n == 2) bwt_wait<4>(F);
            else if (NSLOT - 1 - n == 1) bwt_wait<2>(F);
            else bwt_wait<0>(F);
            if (n == 0) {                                         // the dY fragments were requested before slot 0's
#pragma unroll
                for (int jj = 0; jj < KSW; ++jj) bwt_wait<6>(FB[jj]);
            }
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F, FB[j], acc[t], 0, 0, 0);
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
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (1 + DQ)) : "memory");     // unit c + 5 and dY chunk c + 1 have landed
        __builtin_amdgcn_s_barrier();
    };
    for (int c = 0; c < KT; ++c) chunk(c);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // no DMA may land in the reduction buffer
    __syncthreads();

    // ---- the k-split waves' sums meet in LDS tap by tap (fixed order); the operand images are dead
    float* o = out + (long)split_id * slab;
    float* red = (float*)lds_raw;                                   // [WK][32][BN]
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wk * 32 * BN + mfma32_row(e, lane) * BN + wn * 32 + l31] = acc[t][e];
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
