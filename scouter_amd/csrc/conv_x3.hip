// fp32-grade POINTWISE (1x1, stride 1) convolutions on the bf16 matrix cores with the three-way operand split done IN
// REGISTERS (round 5).  Call sites: the deep 1x1 layers of the ResNeSt bottlenecks -- conv1 / conv3
// (/root/reference/timm/models/resnest.py:111-143) and the downsample convolution (resnet.py:292-306) -- forward and
// input gradient.
//
// The bf16x3 scheme of conv_planes.hip (x = hi + mid + lo exactly, six bf16 products per fp32 product, two accumulator
// sets) reads PRE-SPLIT operand planes: 6 bytes per activation element written by the producing pass.  Round 3 routed the
// deep 1x1 layers there and lost: the producers of their inputs (block outputs, attention-weighted sums, pooled maps,
// BatchNorm-backward gradients) had to write planes next to -- or instead of -- the fp32 tensor every other consumer reads.
// Here the ACTIVATION stays a plain fp32 tensor in HBM (4 bytes per element, no producer changes): a thread loads eight
// consecutive k-values of a row (two 16-byte loads), splits them with ~5 VALU instructions per element -- which execute in
// the shadow of the bf16 MFMAs (tools_dev/mfma_bf16_shadow_bench.hip: up to 4 VALU per MFMA are free, unlike the fp32
// MFMA) -- and writes one 16-byte chunk per plane into the SAME swizzled LDS image the LDS-DMA of conv_planes.hip
// produces.  The WEIGHT planes are split once per step by the model's one split launch (scouter_planes_split_weights_multi)
// and arrive by LDS-DMA.  K order, product order and the two accumulator sets are those of pconv_kernel, so the result is
// BIT-IDENTICAL to scouter_conv2d_fwd_planes / _dgrad_planes on pre-split planes (tests/test_x3_gpu.py) -- a third of
// the exact-fp32 MFMA kernel's rounding error -- at 2/3 of the operand bytes.
//
//   C[M][N] = A[M][K] * Bt[N][K]^T     A fp32 row-major (row stride = K), Bt = weight planes [3][N][K] bf16 (k contiguous):
//   forward: A = x, planes = the forward layout [3][1][Cout][Cin]; input gradient: A = dy, planes = [3][1][Cin][Cout].
//
// Workgroup = 2*NWM waves, tile BM x BN x 32, two LDS stages; pipeline per K-tile kt (one barrier):
//   frag reads (kt, step 1) | MFMAs step 0 | wait: own LDS traffic + the DMA of B(kt+1) | barrier |
//   A(kt+2): registers -> split -> LDS stage(kt);  DMA B(kt+2) -> stage(kt);  global loads A(kt+3) -> registers |
//   frag reads (kt+1, step 0) | MFMAs step 1
// so a global load has one whole K-tile of MFMAs (>= 1.5 k cycles) to land, and the split's VALU sits between MFMAs.
#include "conv_common.h"

#include <stdlib.h>
#include <type_traits>

typedef __bf16 xbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short xu16x8 __attribute__((ext_vector_type(8)));

#define X3_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// Development builds only (tools_dev/x3_ablate.sh): X3_ABLATE bits remove one ingredient of xgemm_kernel's K loop each --
// 1 the split's arithmetic (the planes get the raw upper halves), 2 the global loads of A, 4 the LDS-DMA of B, 8 the fragment
// reads, 16 the barrier, 32 the MFMAs, 64 the epilogue.  Results are WRONG; only the timing means something.
#ifndef X3_ABLATE
#define X3_ABLATE 0
#endif

// (a NON-template helper on purpose -- see conv_planes.hip: the builtin inside a kernel template makes hipcc's host pass drop
//  the kernel's launch stub)
__device__ __forceinline__ void x3_dma16(__amdgpu_buffer_rsrc_t rs, char* lds, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, X3_LDS_PTR(lds), 16, voff, soff, 0, 0);
}
__device__ __forceinline__ f32x16 x3_mfma(xbf16x8 a, xbf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// CONV = false: pointwise GEMM (groups == 1; A rows bounded by the descriptor).  CONV = true: stride-1 convolution with
// R x S taps, padding and groups -- implicit GEMM with the row state of conv_igemm.hip (block-relative offsets, separable
// tap masks, out-of-image taps read zeros through an out-of-range offset); K order as pconv_kernel's (forward: tap outer,
// 32-channel chunk inner; input gradient: chunk outer, tap inner), so results stay bit-identical to the plane kernels.
// NWM x NWN waves; wave tile (BM / NWM) x (BN / NWN).
template <int BM, int BN, int NWM, int NWN, int MINB, bool DGRAD, bool CONV>
__global__ __launch_bounds__(NWM * NWN * 64, MINB) void xgemm_kernel(const float* __restrict__ a_f32,
                                                                       const unsigned short* __restrict__ w_planes, long w_plane_elems,
                                                                       const float* __restrict__ bias, const float* __restrict__ addend,
                                                                       float* __restrict__ dst, double* __restrict__ bn_part, ConvGeom g,
                                                                       int relu, int mtiles, int ntiles, BnBwdFuse fz) {
    constexpr int BK = 32, NW = NWM * NWN, WM = BM / NWM, WN = BN / NWN, MT = WM / 32, NT = WN / 32;
    constexpr int A_BYTES = 3 * BM * 64, B_BYTES = 3 * BN * 64, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int ARG = BM / 16 / NW;                          // 16-row groups of A per wave
    constexpr int BRT = BN / 16, BRG = (BRT + NW - 1) / NW;    // 16-row groups of B: in all / per wave (the first BRT waves-slots)
    constexpr int GF = (ARG + 1) / 2, GB = ARG - GF;           // row groups split in section F / in section B (staggered)
    constexpr int NA = 2 * ARG;                                // global loads of A per thread and K-tile
    static_assert(BM % (16 * NW) == 0 && WM % 32 == 0 && WN % 32 == 0 && ARG >= 1, "tile");
    extern __shared__ __attribute__((aligned(1024))) char lds_raw[];

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: LDS-DMA destinations are wave-uniform (M0)
    const int wm = wave / NWN, wn = wave % NWN;
    const int nblk = mtiles * ntiles * g.groups;
    const int bid = xcd_remap(blockIdx.x, nblk);
    const int grp = bid % g.groups;
    const int nt_id = (bid / g.groups) % ntiles;
    const int mt_id = bid / (g.groups * ntiles);
    const long m0 = (long)mt_id * BM;
    const int n0 = nt_id * BN;
    const int cpt = g.Cg / BK;                 // K chunks per filter tap
    const int ntaps = g.R * g.S;
    const int KT = ntaps * cpt;

    // ---- addressing.  Lane = (row-in-group = lane >> 2, LDS slot = lane & 3); it handles the 16-byte chunk (8 k-values)
    // that the swizzle maps to its slot: chunk c of row r lives at slot c ^ ((r >> 2) & 3), so a wave's 16 rows x 4 slots
    // are lane-linear in LDS (what the DMA needs for B, and conflict-free 16-byte stores for A).
    constexpr unsigned OOB = 0x80000000u;
    const int slot = lane & 3, rin = lane >> 2;
    const int chunk = slot ^ ((lane >> 4) & 3);
    const int rows_valid = (int)(g.M - m0 < BM ? g.M - m0 : BM);
    unsigned a_voff[ARG], a_mask[CONV ? ARG : 1];
    __amdgpu_buffer_rsrc_t rs_a;
    long shift = 0;
    if constexpr (CONV) {
        const int hw = g.Ho * g.Wo;
        const int blk_b = (int)(((double)(unsigned)m0 + 0.5) * g.inv_hw);
        const int blk_rem = (int)((unsigned)m0 - (unsigned)blk_b * (unsigned)hw);
        const int blk_y = fast_div(blk_rem, g.inv_wo), blk_x = blk_rem - blk_y * g.Wo;
#pragma unroll
        for (int t = 0; t < ARG; ++t) {
            const int rowoff = 16 * (wave + NW * t) + rin;
            const bool okm = rowoff < rows_valid;
            const int tx = blk_x + rowoff, qx = fast_div(tx, g.inv_wo), x = tx - qx * g.Wo;
            const int ty = blk_y + qx, qy = fast_div(ty, g.inv_ho), y = ty - qy * g.Ho;
            const int ay = DGRAD ? y + g.pad : y - g.pad, ax = DGRAD ? x + g.pad : x - g.pad;
            unsigned colbits = 0, mask = 0;
            for (int q = 0; q < g.S; ++q) colbits |= ((unsigned)(DGRAD ? ax - q : ax + q) < (unsigned)g.W ? 1u : 0u) << q;
            for (int r = 0; r < g.R; ++r) mask |= ((unsigned)(DGRAD ? ay - r : ay + r) < (unsigned)g.H ? colbits : 0u) << (r * g.S);
            a_mask[t] = okm ? mask : 0u;
            const int rel = okm ? qy * g.H * g.W : 0;
            const int e = DGRAD ? (rel + ay * g.W + ax) * g.C : (rel + (ay + g.pad) * g.W + (ax + g.pad)) * g.C;
            a_voff[t] = (unsigned)(e + grp * g.Cg + chunk * 8) * 4u;
        }
        const long img_elems = (long)g.H * g.W * g.C;
        shift = DGRAD ? ((long)(g.R - 1) * g.W + (g.S - 1)) * g.C : ((long)g.pad * g.W + g.pad) * g.C;
        rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(a_f32 + (long)blk_b * img_elems - shift), 0, 0x7fffffff, 0x00020000);
    } else {
        // block-relative byte offsets; the descriptor ends with the block's last valid row, rows beyond read zeros
        rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(a_f32 + m0 * g.Cg), 0, (unsigned)rows_valid * (unsigned)g.Cg * 4u, 0x00020000);
#pragma unroll
        for (int t = 0; t < ARG; ++t) a_voff[t] = ((unsigned)(16 * (wave + NW * t) + rin) * (unsigned)g.Cg + chunk * 8u) * 4u;
    }
    // B rows: n = n0 + row; weight planes [tap][N_total][Cg(k)] with this group's rows at grp * Ng
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(w_planes + (long)grp * g.Ng * g.Cg), 0, 0x7fffffff, 0x00020000);
    unsigned b_voff[BRG];
#pragma unroll
    for (int t = 0; t < BRG; ++t) b_voff[t] = (unsigned)((n0 + 16 * (wave + NW * t) + rin) * g.Cg + chunk * 8) * 2u;
    const long w_plane_bytes = w_plane_elems * 2;
    const long wtap_bytes = (long)g.N * g.Cg * 2;

    // K-tile state, advanced incrementally (no divisions in the loop): filter tap (r, q) and first channel of the chunk
    struct XK { int tap, r, q, c0; };
    auto k_next = [&](XK k) -> XK {
        if constexpr (!CONV) {
            k.c0 += BK;
        } else if constexpr (DGRAD) {               // chunk outer, tap inner
            ++k.tap; ++k.q;
            if (k.q == g.S) { k.q = 0; ++k.r; }
            if (k.tap == ntaps) { k.tap = 0; k.r = 0; k.q = 0; k.c0 += BK; }
        } else {                                    // tap outer, chunk inner
            k.c0 += BK;
            if (k.c0 == g.Cg) { k.c0 = 0; ++k.tap; ++k.q; if (k.q == g.S) { k.q = 0; ++k.r; } }
        }
        return k;
    };
    f32x4 ra[ARG][2];                                               // A of this thread: fp32, waiting for the split
    auto load_a_group = [&](const XK& k, int t) {
        unsigned vo = a_voff[t];
        int soff = k.c0 * 4;
        if constexpr (CONV) {
            const long toff = (DGRAD ? -((long)k.r * g.W + k.q) : ((long)k.r * g.W + k.q)) * g.C + k.c0;
            soff = (int)((DGRAD ? shift + toff : toff) * 4);
            vo = ((a_mask[t] >> k.tap) & 1u) ? vo : OOB;
        }
        ra[t][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, vo, soff, 0));
        ra[t][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, vo + 16u, soff, 0));
    };
    auto store_a_group = [&](int stage, int t) {                    // exact split x = hi + mid + lo, one 16-byte chunk per plane
        xu16x8 ph, pm, pl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            unsigned short a, b, c;
            if (X3_ABLATE & 1) { a = b = c = (unsigned short)(__float_as_uint(ra[t][e >> 2][e & 3]) >> 16); }
            else split3_bf16(ra[t][e >> 2][e & 3], a, b, c);
            ph[e] = a; pm[e] = b; pl[e] = c;
        }
        char* d = lds_raw + stage * STAGE_BYTES + lane * 16 + (wave + NW * t) * 1024;
        *(xu16x8*)(d) = ph;
        *(xu16x8*)(d + BM * 64) = pm;
        *(xu16x8*)(d + 2 * BM * 64) = pl;
    };
    auto dma_b = [&](const XK& k, int stage) {
        char* st = lds_raw + stage * STAGE_BYTES + A_BYTES;
        const long sb = k.tap * wtap_bytes + (long)k.c0 * 2;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int t = 0; t < BRG; ++t)
                if (BRT % NW == 0 || wave + NW * t < BRT)           // (narrow tiles: fewer 16-row groups than waves)
                    x3_dma16(rs_b, st + pl * (BN * 64) + (wave + NW * t) * 1024, b_voff[t], (int)(sb + pl * w_plane_bytes));
    };

    // Two accumulator sets (conv_planes.hip): hi*hi in `acc`, the five correction products in `accl`, added once at the end.
    f32x16 acc[MT][NT], accl[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; accl[i][j][e] = 0.f; }

    const int sw = (l31 >> 2) & 3;
    auto frag = [&](const char* base, int row0, int step) -> xbf16x8 {
        const int c = (2 * step + h) ^ sw;
        return *(const xbf16x8*)(base + (row0 + l31) * 64 + c * 16);
    };
    xbf16x8 F[2][MT + NT][3];
    auto load_frags = [&](int buf, int stage, int step) {
        const char* As = lds_raw + stage * STAGE_BYTES;
        const char* Bs = As + A_BYTES;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < MT; ++i) F[buf][i][pl] = frag(As + pl * (BM * 64), wm * WM + i * 32, step);
#pragma unroll
            for (int j = 0; j < NT; ++j) F[buf][MT + j][pl] = frag(Bs + pl * (BN * 64), wn * WN + j * 32, step);
        }
    };
    auto mma = [&](int buf) {
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};      // smallest terms first (as pconv_kernel)
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (pr < 5) accl[i][j] = x3_mfma(F[buf][i][PA[pr]], F[buf][MT + j][PB[pr]], accl[i][j]);
                    else acc[i][j] = x3_mfma(F[buf][i][PA[pr]], F[buf][MT + j][PB[pr]], acc[i][j]);
                }
    };
#define SBAR() __builtin_amdgcn_sched_barrier(0)
#define SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
    constexpr int NMMA = MT * NT * 6;                     // MFMAs per half K-tile
    constexpr int NFR = (MT + NT) * 3;                    // fragment reads per half K-tile
    constexpr int VGRP = CONV ? 58 : 52;                  // VALU instructions of one row group's split (+ its masked addresses)

    // ---- STAGGERED row groups.  Row groups [0, GF) of A(kt+2) are split in section F(kt), groups [GF, ARG) in section
    // B(kt+1) (same target stage: it is free from barrier D(kt) to barrier D(kt+1)); each group's registers are re-loaded
    // right behind its split, so every load has a whole iteration (two sections) to come back from HBM -- with ARG * 8
    // staging registers -- and the split's VALU is spread over both sections.
    //   B(kt):  frag reads (kt, step 1) | split groups [GF, ARG) of A(kt+1) -> stage(kt+1) | reload them from A(kt+2) | MFMAs step 0
    //   C/D:    lgkmcnt(0); the DMA of B(kt+1) has landed (NA younger loads may be in flight: counted wait, the DMA is
    //           kept in front of them by a scheduling fence); barrier
    //   E:      DMA B(kt+2) -> stage(kt)
    //   F(kt):  frag reads (kt+1, step 0) | split groups [0, GF) of A(kt+2) -> stage(kt) | reload them from A(kt+3) | MFMAs step 1
    // Program order inside a section = the order LDS accesses must keep (the compiler cannot tell the two stages apart):
    // fragment READS first, then the split's stores, then the global loads.
    const XK k0{0, 0, 0, 0};
    const XK k1 = KT > 1 ? k_next(k0) : k0;                        // (past the end: the last tile again, never used)
    XK k2 = KT > 2 ? k_next(k1) : k1;
    XK k3 = KT > 3 ? k_next(k2) : k2;
#pragma unroll
    for (int t = 0; t < ARG; ++t) load_a_group(k0, t);
    dma_b(k0, 0);
    dma_b(k1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < ARG; ++t) store_a_group(0, t);
#pragma unroll
    for (int t = 0; t < ARG; ++t) load_a_group(k1, t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < GF; ++t) store_a_group(1, t);              // (groups [GF, ARG) of A(1) stay in registers for B(0))
#pragma unroll
    for (int t = 0; t < GF; ++t) load_a_group(k2, t);
    __builtin_amdgcn_s_waitcnt(0xc07f);                            // lgkmcnt(0): this wave's LDS stores
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0xc07f);
#ifdef X3_SETPRIO
    // (MI355X_MICROARCH.md "Two waves per SIMD", item 4: the second-dispatched half of an 8-wave workgroup loses every
    //  arbitration; ONE static priority bump for it)
    if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    for (int kt = 0; kt < KT; ++kt) {
        const int stage = kt & 1, nstage = stage ^ 1;
        SBAR();
        if (!(X3_ABLATE & 8)) load_frags(1, stage, 1);           // A
#pragma unroll
        for (int t = GF; t < ARG; ++t) { store_a_group(nstage, t); if (!(X3_ABLATE & 2)) load_a_group(k2, t); }
        if (!(X3_ABLATE & 32)) mma(0);                            // B
        {
            constexpr int VPS = GB ? (GB * VGRP + NMMA - 1) / NMMA + 1 : 2;
#pragma unroll
            for (int q = 0; q < NMMA; ++q) {
                SG(0x008, 1);
                if (q < NFR) SG(0x100, 1);
                SG(GB ? 0x002 : 0x006, VPS);
#pragma unroll
                for (int t = 0; t < GB; ++t) {
                    constexpr int dummy = 0; (void)dummy;
                    const int wq = ((t + 1) * VGRP + VPS - 1) / VPS < NMMA - 2 ? ((t + 1) * VGRP + VPS - 1) / VPS : NMMA - 2;
                    if (q == wq) { SG(0x200, 3); SG(0x020, 2); }
                }
            }
        }
        SBAR();
        __builtin_amdgcn_s_waitcnt(0xc07f);                       // C
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA) : "memory");  // the DMA of B(kt+1) has landed; A loads may fly on
        if (!(X3_ABLATE & 16)) __builtin_amdgcn_s_barrier();      // D: tile kt+1 complete in LDS, stage(kt) free
        SBAR();
        if (!(X3_ABLATE & 4)) dma_b(k2, stage);                   // E (fenced: stays in front of the loads of section F)
        SBAR();
        if (!(X3_ABLATE & 8)) load_frags(0, nstage, 0);
#pragma unroll
        for (int t = 0; t < GF; ++t) { store_a_group(stage, t); if (!(X3_ABLATE & 2)) load_a_group(k3, t); }
        if (!(X3_ABLATE & 32)) mma(1);                            // F
        {
            constexpr int VPS = (GF * VGRP + NMMA - 1) / NMMA + 1;
#pragma unroll
            for (int q = 0; q < NMMA; ++q) {
                SG(0x008, 1);
                if (q < NFR) SG(0x100, 1);
                SG(0x002, VPS);
#pragma unroll
                for (int t = 0; t < GF; ++t) {
                    const int wq = ((t + 1) * VGRP + VPS - 1) / VPS < NMMA - 2 ? ((t + 1) * VGRP + VPS - 1) / VPS : NMMA - 2;
                    if (q == wq) { SG(0x200, 3); SG(0x020, 2); }
                }
            }
        }
        SBAR();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        k2 = k3;
        if (kt + 4 < KT) k3 = k_next(k3);
    }
#undef SBAR
#undef SG
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] += accl[i][j];
    if (X3_ABLATE & 64) {                       // (every accumulator stays live: nothing of the K loop may be eliminated)
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
        if (sum == 12345.678f) dst[tid] = sum;
        return;
    }
    igemm_epilogue<BM, BN, WM, WN, DGRAD>(acc, (float*)lds_raw, g, bias, addend, dst, bn_part, relu, m0, n0, grp, mt_id, &fz);
}

template <int BM, int BN, int NWM, int NWN, int MINB, bool DGRAD, bool CONV>
static void launch_x3(const float* a, const void* w, long w_pe, const float* bias, const float* addend, float* dst,
                      double* bn_part, const ConvGeom& g, int relu, hipStream_t st, const BnBwdFuse& fz) {
    const int mtiles = sc_cdiv(g.M, BM), ntiles = g.Ng / BN;
    ConvGeom gg = g;
    gg.inv_hw = 1.0 / ((double)g.Ho * g.Wo);
    gg.inv_wo = 1.0f / (float)g.Wo;
    gg.inv_ho = 1.0f / (float)g.Ho;
    constexpr int STAGE_BYTES = 3 * (BM + BN) * 64;
    constexpr int EPI_BYTES = NWM * NWN * (BM / NWM) * (BN / NWN + 4) * 4;
    const size_t lds = (size_t)(2 * STAGE_BYTES > EPI_BYTES ? 2 * STAGE_BYTES : EPI_BYTES);
    auto kern = xgemm_kernel<BM, BN, NWM, NWN, MINB, DGRAD, CONV>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(mtiles * ntiles * g.groups), dim3(NWM * NWN * 64), lds, st, a, (const unsigned short*)w, w_pe,
                       bias, addend, dst, bn_part, gg, relu, mtiles, ntiles, fz);
}

// block tiles: 0 = 256x128 (eight waves), 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 256x64 (eight waves), 5 = 256x32 (four
// waves along M), 6 = 128x32 (two waves, two workgroups per CU: the 32-channel layers with a heavy epilogue)
static int x3_tile_rows(int tile) { return (tile == 0 || tile == 4 || tile == 5) ? 256 : (tile == 3 ? 64 : 128); }
static bool x3_tile_ok(int tile, int N) {
    switch (tile) {
        case 0: case 1: return N % 128 == 0;
        case 2: case 3: case 4: return N % 64 == 0;
        case 5: case 6: return N % 32 == 0;
        default: return false;
    }
}
static int x3_tile(long M, int N, int hint) {
    if (hint >= 0 && hint <= 6 && x3_tile_ok(hint, N)) return hint;
    if (N % 64 != 0) return 6;
    // enough 256 x 128 tiles for two rounds of the chip, else the 128-row tiles (two workgroups per CU)
    if (N % 128 == 0 && (long)sc_cdiv(M, 256) * (N / 128) >= 448) return 0;
    return 2;
}
template <bool DGRAD, bool CONV>
static int dispatch_x3(const float* a, const void* w, long w_pe, const float* bias, const float* addend, float* dst,
                       double* bn_part, const ConvGeom& g, int relu, int tile, hipStream_t st, const BnBwdFuse& fz) {
    switch (tile) {
        case 0: launch_x3<256, 128, 4, 2, 1, DGRAD, CONV>(a, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz); break;
        case 1: launch_x3<128, 128, 2, 2, 1, DGRAD, CONV>(a, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz); break;
        case 2: launch_x3<128, 64, 2, 2, 2, DGRAD, CONV>(a, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz); break;
        case 3: launch_x3<64, 64, 2, 2, 3, DGRAD, CONV>(a, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz); break;
        case 4: launch_x3<256, 64, 4, 2, 1, DGRAD, CONV>(a, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz); break;
        case 5: launch_x3<256, 32, 4, 1, 1, DGRAD, CONV>(a, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz); break;
        default: launch_x3<128, 32, 2, 1, 2, DGRAD, CONV>(a, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz); break;
    }
    return sc_check_launch(DGRAD ? "conv2d_dgrad_x3" : "conv2d_fwd_x3");
}

extern "C" int scouter_conv2d_x3_tile(long M, int N, int tile_hint) { return tile_hint == 7 && N == 32 ? 7 : x3_tile(M, N, tile_hint); }
extern "C" int scouter_conv2d_x3_partial_rows(long M, int N, int tile_hint) {
    return sc_cdiv(M, x3_tile_rows(x3_tile(M, N, tile_hint)));
}
// tile 7 (conv_xhalo.hip, persistent): ONE partial row per workgroup of a group, whatever M
extern "C" int scouter_conv2d_x3_halo_partial_rows(int groups) { return groups > 0 ? sc_xhalo_partial_rows(groups) : 0; }

// kh x kw: 1x1 (pad 0) or an odd "same" filter (pad = (k - 1) / 2), stride 1; groups with 32-multiples of channels per group
extern "C" int scouter_conv2d_fwd_x3(const float* x, const void* w_planes_fwd, const float* bias, const float* addend, float* y,
                                     double* bn_partial, int B, int H, int W, int Cin, int Cout, int kh, int kw, int pad,
                                     int groups, int relu, int tile_hint, void* stream) {
    SC_REQUIRE(x && w_planes_fwd && y && B > 0 && H > 0 && W > 0, "conv2d_fwd_x3: null pointer or empty shape");
    SC_REQUIRE(!(bn_partial && relu), "conv2d_fwd_x3: fused BatchNorm statistics are taken before any activation");
    SC_REQUIRE(groups > 0 && Cin % groups == 0 && Cout % groups == 0, "conv2d_fwd_x3: channels not divisible by groups");
    const int Cg = Cin / groups, Ng = Cout / groups;
    SC_UNSUPPORTED(kh == kw && (kh & 1) && 2 * pad == kh - 1 && kh <= 5, "conv2d_fwd_x3: 1x1 or odd same-size filters only");
    SC_UNSUPPORTED(Cg % 32 == 0 && Ng % 32 == 0 && kh * kw * (Cg / 32) >= 2, "conv2d_fwd_x3: needs 32-multiples of channels per group and two K-tiles");
    ConvGeom g{B, H, W, Cin, H, W, Cout, kh, kw, 1, pad, groups, Cg, Ng, 0, Cout, Cg * Cout};
    g.M = (long)B * H * W;
    SC_UNSUPPORTED(g.M < (1L << 31) && (long)kh * kw * Cg * Cout * 2 * 3 < (1L << 31) && (long)H * W * Cin < (1L << 28) &&
                   256L * Cin * 4 < (1L << 31), "conv2d_fwd_x3: tensor too large for 32-bit offsets");
    const long w_pe = (long)kh * kw * Cg * Cout;
    if (tile_hint == 7) {           // resident rows, split once per element, persistent (conv_xhalo.hip): 32 columns per group
        SC_UNSUPPORTED(sc_xhalo_ok(g) && !bias && !addend, "conv2d_fwd_x3: tile 7 covers 3x3 / pad 1 layers with 32 output channels per "
                       "group, 16-multiples of input channels per group, maps up to 126 wide, no bias / addend");
        ScProfScope prof("xhalo_fwd<bf16x3>", (hipStream_t)stream, 2.0 * g.M * Cout * Cg * kh * kw, 4.0 * g.M * Cin + 4.0 * g.M * Cout);
        sc_launch_xhalo(false, x, w_planes_fwd, w_pe, nullptr, y, bn_partial, g, relu, (hipStream_t)stream, BnBwdFuse{});
        return sc_check_launch("conv2d_fwd_x3<halo>");
    }
    const int tile = x3_tile(g.M, Ng, tile_hint);
    const bool conv = kh > 1 || groups > 1;
    ScProfScope prof(conv ? "xconv3_fwd<bf16x3>" : "xconv_fwd<bf16x3>", (hipStream_t)stream, 2.0 * g.M * Cout * Cg * kh * kw,
                     4.0 * g.M * Cin + 4.0 * g.M * Cout);
    if (conv) return dispatch_x3<false, true>(x, w_planes_fwd, w_pe, bias, addend, y, bn_partial, g, relu, tile, (hipStream_t)stream, BnBwdFuse{});
    return dispatch_x3<false, false>(x, w_planes_fwd, w_pe, bias, addend, y, bn_partial, g, relu, tile, (hipStream_t)stream, BnBwdFuse{});
}

// dx = dy (*) W^T (+ addend), optionally with the BatchNorm-backward epilogue of conv_common.h
extern "C" int scouter_conv2d_dgrad_x3_bnbwd(const float* dy, const void* w_planes_dgrad, const float* addend, float* dx, int B,
                                             int H, int W, int Cin, int Cout, int kh, int kw, int pad, int groups, int tile_hint,
                                             const void* relu_mask, const float* x1, const float* saved1, double* part1,
                                             const float* x2, const float* saved2, double* part2, void* stream) {
    SC_REQUIRE(dy && w_planes_dgrad && dx && B > 0 && H > 0 && W > 0, "conv2d_dgrad_x3: null pointer or empty shape");
    SC_REQUIRE(!part1 || (x1 && saved1), "conv2d_dgrad_x3: fused BatchNorm backward needs x1 and saved1");
    SC_REQUIRE(!part2 || (part1 && x2 && saved2), "conv2d_dgrad_x3: second fused BatchNorm needs the first, x2 and saved2");
    SC_REQUIRE(groups > 0 && Cin % groups == 0 && Cout % groups == 0, "conv2d_dgrad_x3: channels not divisible by groups");
    const int Cig = Cin / groups, Cog = Cout / groups;
    SC_UNSUPPORTED(kh == kw && (kh & 1) && 2 * pad == kh - 1 && kh <= 5, "conv2d_dgrad_x3: 1x1 or odd same-size filters only");
    SC_UNSUPPORTED(Cog % 32 == 0 && Cig % 32 == 0 && kh * kw * (Cog / 32) >= 2, "conv2d_dgrad_x3: needs 32-multiples of channels per group and two K-tiles");
    ConvGeom g{B, H, W, Cout, H, W, Cin, kh, kw, 1, pad, groups, Cog, Cig, 0, Cout, Cig * Cout};
    g.M = (long)B * H * W;
    SC_UNSUPPORTED(g.M < (1L << 31) && (long)kh * kw * Cig * Cout * 2 * 3 < (1L << 31) && (long)H * W * Cout < (1L << 28) &&
                   256L * Cout * 4 < (1L << 31), "conv2d_dgrad_x3: tensor too large for 32-bit offsets");
    const BnBwdFuse fz{part1 ? (const unsigned long long*)relu_mask : nullptr, x1, saved1, part1, x2, saved2, part2, 0};
    const long w_pe = (long)kh * kw * Cig * Cout;
    if (tile_hint == 7) {           // (conv_xhalo.hip: 32 input channels per group = 32 GEMM columns)
        SC_UNSUPPORTED(sc_xhalo_ok(g), "conv2d_dgrad_x3: tile 7 covers 3x3 / pad 1 layers with 32 input channels per group, "
                       "16-multiples of output channels per group, maps up to 126 wide");
        ScProfScope prof(part1 ? "xhalo_dgrad+bn_bwd<bf16x3>" : "xhalo_dgrad<bf16x3>", (hipStream_t)stream,
                         2.0 * g.M * Cin * Cog * kh * kw, 4.0 * g.M * Cout + 4.0 * g.M * Cin);
        sc_launch_xhalo(true, dy, w_planes_dgrad, w_pe, addend, dx, nullptr, g, 0, (hipStream_t)stream, fz);
        return sc_check_launch("conv2d_dgrad_x3<halo>");
    }
    const int tile = x3_tile(g.M, Cig, tile_hint);
    const bool conv = kh > 1 || groups > 1;
    ScProfScope prof(part1 ? (conv ? "xconv3_dgrad+bn_bwd<bf16x3>" : "xconv_dgrad+bn_bwd<bf16x3>")
                           : (conv ? "xconv3_dgrad<bf16x3>" : "xconv_dgrad<bf16x3>"),
                     (hipStream_t)stream, 2.0 * g.M * Cin * Cog * kh * kw, 4.0 * g.M * Cout + 4.0 * g.M * Cin);
    if (conv) return dispatch_x3<true, true>(dy, w_planes_dgrad, w_pe, nullptr, addend, dx, nullptr, g, 0, tile, (hipStream_t)stream, fz);
    return dispatch_x3<true, false>(dy, w_planes_dgrad, w_pe, nullptr, addend, dx, nullptr, g, 0, tile, (hipStream_t)stream, fz);
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient of the same layers:  dW[ci][co] = sum_m X[m][ci] * dY[m][co]    (1x1: a plain X^T dY over the pixels)
// ---------------------------------------------------------------------------------------------------------------
// Both operands have the contraction index (the pixel) as their slow axis, while a bf16 MFMA lane wants 8 consecutive k of
// one row: as in conv_wgrad_bf16.h every thread loads 4-pixel x 4-channel patches (four coalesced 16-byte loads per
// patch), transposes them in registers and writes 8-byte rows of the TRANSPOSED LDS tile [channel][pixel] -- here three of
// them per operand (hi / mid / lo of the exact split), so fragments stay plain 16-byte reads and the six products run as
// in xgemm_kernel (two accumulator sets).  128 x 128 output tile, four waves (one per SIMD: 64 x 64 wave tiles keep the
// LDS at 0.5 fragment reads per MFMA), chunk = 32 pixels, two LDS stages, split-K over the pixels with the deterministic
// slab sum of conv_igemm.hip.  One wave per SIMD means every latency is hidden inside the wave:
//   * the patches of chunk c+2 are requested into the SECOND register set when chunk c starts (a whole chunk of MFMAs,
//     ~1.6 k cycles, before they are needed: the loop is unrolled by two, set = chunk parity);
//   * the split + transposed stores of chunk c+1 are issued between the MFMAs of chunk c (program order: fragment reads,
//     then the stores -- the compiler cannot tell the two stages apart and keeps LDS accesses in order).
// development builds (tools_dev/x3_ablate.sh xw): XW_ABLATE bits 1 no split arithmetic, 2 no global loads in the loop, 8 no
// fragment reads, 16 no barrier, 32 no MFMAs, 128 no LDS stores.  Results are WRONG; only the timing means something.
#ifndef XW_ABLATE
#define XW_ABLATE 0
#endif
template <int DUMMY>
__global__ __launch_bounds__(256, 1) void xwgrad_kernel(const float* __restrict__ act, const float* __restrict__ dy,
                                                        float* __restrict__ out, ConvGeom g, int ci_tiles, int co_tiles,
                                                        long pix_per_split, long slab) {
    constexpr int BM = 128, BN = 128, KP = 32, LDP = KP + 8;       // LDS row stride (bf16): 16-byte aligned, conflict-free reads
    constexpr int PL_A = BM * LDP, PL_B = BN * LDP;                // bf16 elements per operand plane
    constexpr int STAGE_H = 3 * (PL_A + PL_B);
    constexpr int MT = 2, NT = 2;
    extern __shared__ __attribute__((aligned(16))) __bf16 ldsx[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    int bid, split_id;
    wgrad_block_coords(bid, split_id);
    const int co_t = bid % co_tiles, ci_t = bid / co_tiles;
    const int ci0 = ci_t * BM, co0 = co_t * BN;
    const long mbeg = (long)split_id * pix_per_split;
    long mend = mbeg + pix_per_split;
    if (mend > g.M) mend = g.M;
    const int KT = (int)((mend - mbeg + KP - 1) / KP);

    // patch of this thread: channel quad cq (0..31), pixel quad pq (0..7) -- 8 lanes cover 128 contiguous bytes of a pixel
    const int cq = (tid & 7) + 8 * (tid >> 6), pq = (tid >> 3) & 7;
    const unsigned a_voff = (unsigned)((4 * pq * (long)g.C + ci0 + 4 * cq) * 4);
    const unsigned b_voff = (unsigned)((4 * pq * (long)g.N + co0 + 4 * cq) * 4);
    auto records = [&](long m_chunk, int row_elems) {             // bytes up to this split's last pixel: rows beyond read zeros
        const long n = (mend - m_chunk) * (long)row_elems * 4;
        return (unsigned)(n < 0 ? 0 : (n > 0x7fffffffL ? 0x7fffffffL : n));
    };
    f32x4 ra[2][4], rb[2][4];                                      // two register sets (chunk parity)
    auto load_ab = [&](int kt, auto SET) {
        constexpr int s = decltype(SET)::value;
        const long m = mbeg + (long)kt * KP;
        const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(act + m * g.C), 0, records(m, g.C), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + m * g.N), 0, records(m, g.N), 0x00020000);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            ra[s][rr] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa, a_voff + (unsigned)(rr * g.C * 4), 0, 0));
            rb[s][rr] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsb, b_voff + (unsigned)(rr * g.N * 4), 0, 0));
        }
    };
    typedef unsigned short xu16x4 __attribute__((ext_vector_type(4)));
    auto store_t3 = [&](__bf16* T, int plane_elems, const f32x4 (&v)[4]) {    // 4x4 transpose + exact three-way split
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xu16x4 oh, om, ol;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                unsigned short a, b, c;
                if (XW_ABLATE & 1) { a = b = c = (unsigned short)(__float_as_uint(v[rr][e]) >> 16); }
                else split3_bf16(v[rr][e], a, b, c);
                oh[rr] = a; om[rr] = b; ol[rr] = c;
            }
            __bf16* d = T + (4 * cq + e) * LDP + 4 * pq;
            if (XW_ABLATE & 128) { if (oh[0] == 0x7fc1 && om[1] == 0x7fc1 && ol[2] == 0x7fc1) *(xu16x4*)(d) = oh; continue; }
            *(xu16x4*)(d) = oh;
            *(xu16x4*)(d + plane_elems) = om;
            *(xu16x4*)(d + 2 * plane_elems) = ol;
        }
    };
    auto store_ab = [&](int stage, auto SET) {
        constexpr int s = decltype(SET)::value;
        __bf16* As = ldsx + stage * STAGE_H;
        store_t3(As, PL_A, ra[s]);
        store_t3(As + 3 * PL_A, PL_B, rb[s]);
    };
    f32x16 acc[MT][NT], accl[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; accl[i][j][e] = 0.f; }
    xbf16x8 F[2][MT + NT][3];
    auto load_frags = [&](int buf, int stage, int step) {
        const __bf16* As = ldsx + stage * STAGE_H + (wm * 64 + l31) * LDP + 8 * h + 16 * step;
        const __bf16* Bs = ldsx + stage * STAGE_H + 3 * PL_A + (wn * 64 + l31) * LDP + 8 * h + 16 * step;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < MT; ++i) F[buf][i][pl] = *(const xbf16x8*)(As + pl * PL_A + i * 32 * LDP);
#pragma unroll
            for (int j = 0; j < NT; ++j) F[buf][MT + j][pl] = *(const xbf16x8*)(Bs + pl * PL_B + j * 32 * LDP);
        }
    };
    auto mma = [&](int buf) {
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (pr < 5) accl[i][j] = x3_mfma(F[buf][i][PA[pr]], F[buf][MT + j][PB[pr]], accl[i][j]);
                    else acc[i][j] = x3_mfma(F[buf][i][PA[pr]], F[buf][MT + j][PB[pr]], acc[i][j]);
                }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
#define SBAR() __builtin_amdgcn_sched_barrier(0)
#define SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
    constexpr int NMMA = MT * NT * 6, NFR = (MT + NT) * 3;
    // Chunk c lives in register set c & 1 (ra: the X patch, rb: the dY patch) and in LDS stage c & 1.  Steady state of
    // chunk kt (cur = kt & 1), one barrier:
    //   A/B  fragment reads (kt, step 1) | 24 MFMAs of step 0 | between them: split + transposed stores of chunk kt+1's dY
    //        patch (rb[cur ^ 1]) into stage cur ^ 1
    //   C/D  lgkmcnt(0), barrier: stage cur ^ 1 is complete (its X half was written in F of chunk kt-1), stage cur is free
    //   E/F  request chunk kt+3 into set cur ^ 1 (ra consumed in F(kt-1), rb just now) | fragment reads (kt+1, step 0) |
    //        24 MFMAs of step 1 | between them: split + stores of chunk kt+2's X patch (ra[cur], requested a chunk ago)
    //        into stage cur
    // Program order inside a section = the order LDS accesses must keep (reads of one stage, then stores into the other).
    // (round 5: the build without the loop's global loads is 16 us of 102 faster, but requesting the patches FOUR chunks ahead
    //  (four register sets, loop unrolled by four) changed nothing, 106 vs 103 us, and neither did spreading the eight loads
    //  over both sections (one per six MFMAs, 107 us): neither latency nor issue pressure.  Two sets, one burst stay.)
    auto chunk = [&](int kt, auto CUR) {
        constexpr int cur = decltype(CUR)::value, nxt = cur ^ 1;
        SBAR();
        if (!(XW_ABLATE & 8)) load_frags(1, cur, 1);                            // A
        store_t3(ldsx + nxt * STAGE_H + 3 * PL_A, PL_B, rb[nxt]);               // (chunk kt+1's dY patch)
        if (!(XW_ABLATE & 32)) mma(0);                                          // B
#pragma unroll
        for (int q = 0; q < NMMA; ++q) {
            SG(0x008, 1);
            if (q < NFR) SG(0x100, 1);
            SG(0x002, 5);
            if (q >= 5 && (q - 5) % 6 == 0) SG(0x200, 3);
        }
        SBAR();
        __builtin_amdgcn_s_waitcnt(0xc07f);                                     // C
        if (!(XW_ABLATE & 16)) __builtin_amdgcn_s_barrier();                    // D
        SBAR();
        if (!(XW_ABLATE & 2)) load_ab(kt + 3, std::integral_constant<int, nxt>{});    // E (beyond the split: zeros, never used)
        if (!(XW_ABLATE & 8)) load_frags(0, nxt, 0);
        store_t3(ldsx + cur * STAGE_H, PL_A, ra[cur]);                          // (chunk kt+2's X patch)
        if (!(XW_ABLATE & 32)) mma(1);                                          // F
#pragma unroll
        for (int q = 0; q < NMMA; ++q) {
            SG(0x008, 1);
            if (q < 8) SG(0x020, 1);
            if (q < NFR) SG(0x100, 1);
            SG(0x002, 5);
            if (q >= 5 && (q - 5) % 6 == 0) SG(0x200, 3);
        }
        SBAR();
        __builtin_amdgcn_s_waitcnt(0xc07f);
    };
    // ---- prologue: chunk 0 complete in stage 0, chunk 1's X half in stage 1 (its dY half follows in B of chunk 0),
    // chunk 2 on its way into set 0
    load_ab(0, S0{});
    store_ab(0, S0{});
    load_ab(1, S1{});
    store_t3(ldsx + STAGE_H, PL_A, ra[1]);
    load_ab(2, S0{});
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int kt = 0; kt < KT; kt += 2) {
        chunk(kt, S0{});
        if (kt + 1 < KT) chunk(kt + 1, S1{});
    }
#undef SBAR
#undef SG
    __builtin_amdgcn_s_barrier();
    float* o = out + (long)split_id * slab;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc[i][j] += accl[i][j];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ci = ci0 + wm * 64 + i * 32 + mfma32_row(e, lane);
                const int co = co0 + wn * 64 + j * 32 + l31;
                o[(long)ci * g.N + co] = acc[i][j][e];
            }
        }
}

struct XwPlan { int ci_tiles, co_tiles, splits; long pix_per_split, tiles; };
static XwPlan xw_plan(long M, int Cin, int Cout, int plan_hint) {
    XwPlan p;
    p.ci_tiles = Cin / 128; p.co_tiles = Cout / 128;
    p.tiles = (long)p.ci_tiles * p.co_tiles;
    // one workgroup per CU: `budget` workgroups in all (plan_hint bits 0-1: 256 << hint; default 256 = one round)
    const long budget = 256L << ((plan_hint >= 0 ? plan_hint : 0) & 3);
    long want = budget / p.tiles;
    if (want < 1) want = 1;
    const long chunks = (M + 31) / 32;
    long cps = (chunks + want - 1) / want;
    if (cps < 8) cps = 8;                                 // at least 256 pixels per workgroup
    p.pix_per_split = cps * 32;
    p.splits = (int)((M + p.pix_per_split - 1) / p.pix_per_split);
    return p;
}
extern "C" size_t scouter_conv2d_wgrad_x3_workspace_bytes(int B, int H, int W, int Cin, int Cout, int plan_hint) {
    if (Cin % 128 || Cout % 128 || B <= 0) return 0;
    const XwPlan p = xw_plan((long)B * H * W, Cin, Cout, plan_hint);
    return p.splits > 1 ? (size_t)p.splits * Cin * Cout * sizeof(float) : 0;
}
// dw: [Cin][Cout] fp32 (the HWIO weight gradient of a 1x1 layer).  Deterministic: fixed split boundaries, slabs summed in
// slab order.  plan_hint: bits 0-1 = workgroup budget 256 << hint (-1: 256); different plans sum the pixels in another order.
extern "C" int scouter_conv2d_wgrad_x3(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                                       int plan_hint, void* ws, size_t ws_bytes, void* stream) {
    SC_REQUIRE(x && dy && dw && B > 0 && H > 0 && W > 0, "conv2d_wgrad_x3: null pointer or empty shape");
    SC_UNSUPPORTED(Cin % 128 == 0 && Cout % 128 == 0, "conv2d_wgrad_x3: needs 128-multiples of Cin and Cout");
    ConvGeom g{B, H, W, Cin, H, W, Cout, 1, 1, 1, 0, 1, Cin, Cout, 0, Cout, Cin * Cout};
    g.M = (long)B * H * W;
    SC_UNSUPPORTED(g.M < (1L << 31), "conv2d_wgrad_x3: more than 2^31 pixels");
    const XwPlan p = xw_plan(g.M, Cin, Cout, plan_hint);
    const long slab = (long)Cin * Cout;
    const size_t need = p.splits > 1 ? (size_t)p.splits * slab * sizeof(float) : 0;
    if (need > ws_bytes || (need && !ws)) {
        sc_set_error("conv2d_wgrad_x3: workspace too small (%zu < %zu bytes)", ws_bytes, need);
        return SC_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* out = p.splits > 1 ? (float*)ws : dw;
    int rc;
    {
        ScProfScope prof("xwgrad<bf16x3>", st, 2.0 * g.M * Cout * Cin, 4.0 * g.M * (Cin + Cout));
        constexpr int LDS = 2 * 3 * (128 + 128) * 40 * 2;
        auto kern = xwgrad_kernel<0>;
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)p.tiles, (unsigned)p.splits), dim3(256), LDS, st, x, dy, out, g, p.ci_tiles,
                           p.co_tiles, p.pix_per_split, slab);
        rc = sc_check_launch("conv2d_wgrad_x3");
    }
    if (rc) return rc;
    if (p.splits > 1) {
        ScProfScope prof2("slab_reduce", st, 0, 4.0 * (double)slab * (p.splits + 1));
        sc_launch_slab_reduce((const float*)ws, dw, slab, p.splits, slab, st);
        rc = sc_check_launch("conv2d_wgrad_x3_reduce");
    }
    return rc;
}
