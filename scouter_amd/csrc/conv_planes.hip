// Implicit-GEMM convolution on the bf16 matrix cores over PRE-SPLIT operand planes, fed by LDS-DMA.
//
// Why.  v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate, and the large 3x3 layers of the backbones already
// sit at what the clock allows on it (~130 TFLOP/s).  An fp32 value splits EXACTLY into three bf16 terms
//      x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)          (8 + 8 + 8 mantissa bits)
// and a product a*b is recovered to fp32 accuracy from the six bf16 products of weight <= 2^-16
//      a*b ~= ah*bh + (ah*bm + am*bh) + (ah*bl + al*bh + am*bm)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16: 6 MFMAs at 16x the rate = 2.7x the fp32-MFMA throughput, with the
// error of an fp32 product (relative 2^-24 per term; measured against fp64 like the fp32 kernel, tests/test_planes_gpu.py).
// Round 1 tried this with the split done INSIDE the convolution (VALU + ds_write per loaded element, repeated for each
// of the nine taps): 120-165 TFLOP/s-equivalent, LDS-store-bound.  Here the split is done ONCE by the producer of the
// tensor (the BatchNorm apply pass writes planes instead of fp32, 6 instead of 4 bytes per element) and the
// convolution moves planes global -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR round trip, no ds_write, no VALU):
// per 128x128x32 block tile 48 KB of DMA and 24 ds_read_b128 per wave feed 48 MFMAs (1536 matrix cycles) per wave.
// NP = 1 is the plain bf16 mode (BASELINE configs[4]) with bf16 activation storage: same kernel, one product.
//
// Layouts.  activation planes [NP][M pixels][C] bf16 (each plane an NHWC tensor); weight planes, k contiguous:
//   forward [NP][tap][Cout][Cin/groups]      dgrad [NP][tap][Cin][Cout/groups]     (scouter_planes_split_weight_*)
// LDS image of one (operand, plane) K-tile: rows of 32 bf16 = 64 B = four 16-byte chunks, chunk c of row r stored at
// slot c ^ ((r >> 2) & 3): the DMA writes lane-linearly (1 KB per wave instruction = 16 rows), so the swizzle is applied
// on the SOURCE address; fragment reads (ds_read_b128, lane = row) are then bank-conflict free.
#include "conv_common.h"

#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// LDS-DMA of 16 bytes per lane (buffer_load_dwordx4 ... lds): LDS destination = wave-uniform `lds` + lane * 16.
// (A NON-template helper on purpose: with the builtin inside a kernel template, hipcc's host pass silently drops the
// kernel's launch stub -- the builtin does not exist for the host target -- and the library fails to link at load time.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* lds, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds), 16, voff, soff, 0, 0);
}

__device__ __forceinline__ f32x16 mfma_bf16p(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

#define split3 split3_bf16

// ---------------------------------------------------------------------------------------------------------------
// producers of planes
// ---------------------------------------------------------------------------------------------------------------
// fp32 [n] -> planes [NP][n]   (stand-alone split: weights, tests; activations get their planes from the BatchNorm pass)
template <int NP>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, unsigned short* __restrict__ p,
                                                           long n4, long plane_elems) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *(const f32x4*)(x + i * 4);
        u16x4 h, m, l;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned short a, b, c;
            split3(v[k], a, b, c);
            h[k] = a; m[k] = b; l[k] = c;
        }
        *(u16x4*)(p + i * 4) = h;
        if (NP == 3) { *(u16x4*)(p + plane_elems + i * 4) = m; *(u16x4*)(p + 2 * plane_elems + i * 4) = l; }
    }
}

// weights HWIO fp32 [tap][Cin/g][Cout] -> forward planes [NP][tap][Cout][Cg] and / or dgrad planes [NP][tap][Cin][Ng]
// (Cin index = grp*Cg + ci ; Ng = Cout/groups).  One thread per (tap, ci, co) element; tiny tensors.
template <int NP>
__global__ __launch_bounds__(256) void split_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ wf,
                                                           unsigned short* __restrict__ wd, int taps, int Cg, int Cout,
                                                           int groups) {
    const long n = (long)taps * Cg * Cout;
    const int Ng = Cout / groups;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        const int ci = (int)((i / Cout) % Cg);
        const int tap = (int)(i / ((long)Cout * Cg));
        unsigned short s[3];
        split3(w[i], s[0], s[1], s[2]);
        const int grp = co / Ng;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            if (wf) wf[(long)pl * n + ((long)tap * Cout + co) * Cg + ci] = s[pl];
            if (wd) wd[(long)pl * n + ((long)tap * (Cg * groups) + grp * Cg + ci) * Ng + (co - grp * Ng)] = s[pl];
        }
    }
}

// The same for EVERY plane convolution of a model in one launch (their weights all change with every optimizer step;
// one 16-us launch per layer was 0.13 ms per step for eight layers): row t of the table describes tensor t, `first` is
// the running element count.
struct SplitWeightRow {
    const float* w; unsigned short* wf; unsigned short* wd;
    int taps, Cg, Cout, groups;
    long first;
};
// Round 5: TILED.  The first version ran one thread per element: the input-gradient layout was written coalesced, the
// forward layout -- its transpose -- as 2-byte stores a whole row apart (22 M scattered stores once the deep 1x1 layers of
// conv_x3.hip joined the table).  Now a workgroup takes [32 ci][32 co] tiles of one filter tap (every tensor's Cg and
// Cout are 32-multiples, so `first` / 1024 is the tensor's first tile): coalesced 128-byte reads, the input-gradient planes
// written straight away (co contiguous), the forward planes through an LDS transpose (ci contiguous, 64-byte segments).
template <int NP>
__global__ __launch_bounds__(256) void split_weight_multi_kernel(const SplitWeightRow* __restrict__ rows, int nrows, long total) {
    __shared__ unsigned short tile[3][32][34];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 8 rows of 32 lanes
    const long ntiles = total >> 10;
    for (long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        int t = 0;
        while (t + 1 < nrows && (rows[t + 1].first >> 10) <= tl) ++t;
        const SplitWeightRow R = rows[t];
        const long n = (long)R.taps * R.Cg * R.Cout;
        const int Ng = R.Cout / R.groups;
        const int cot = R.Cout >> 5, cit = R.Cg >> 5;
        long j = tl - (R.first >> 10);
        const int co0 = (int)(j % cot) << 5;
        j /= cot;
        const int ci0 = (int)(j % cit) << 5;
        const int tap = (int)(j / cit);
        const int co = co0 + tx, grp = co / Ng;
#pragma unroll
        for (int r = ty; r < 32; r += 8) {
            const int ci = ci0 + r;
            unsigned short s[3];
            split3(R.w[((long)tap * R.Cg + ci) * R.Cout + co], s[0], s[1], s[2]);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                if (R.wd) R.wd[(long)pl * n + ((long)tap * (R.Cg * R.groups) + grp * R.Cg + ci) * Ng + (co - grp * Ng)] = s[pl];
                tile[pl][r][tx] = s[pl];
            }
        }
        __syncthreads();
        if (R.wf) {
#pragma unroll
            for (int r = ty; r < 32; r += 8)          // row r of the transposed tile: output channel co0 + r, 32 ci contiguous
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    R.wf[(long)pl * n + ((long)tap * R.Cout + co0 + r) * R.Cg + ci0 + tx] = tile[pl][tx][r];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// forward / dgrad on planes
// ---------------------------------------------------------------------------------------------------------------
// g describes the GEMM exactly as in conv_igemm.hip (for DGRAD the A tensor is dY and the columns are Cin).
// NWM waves along M x 2 along N (4 or 8 waves: 256 or 512 threads); every wave owns a 64 x (BN/2) or (BM/NWM) x (BN/2) tile.
template <int BM, int BN, int NP, int NSTAGE, bool DGRAD, int NWM = 2>
__global__ __launch_bounds__(NWM * 128, 1) void pconv_kernel(const unsigned short* __restrict__ a_planes, long a_plane_elems,
                                                       const unsigned short* __restrict__ w_planes, long w_plane_elems,
                                                       const float* __restrict__ bias, const float* __restrict__ addend,
                                                       float* __restrict__ dst, double* __restrict__ bn_part, ConvGeom g,
                                                       int relu, int mtiles, int ntiles, BnBwdFuse fz) {
    constexpr int BK = 32, NW = 2 * NWM, WM = BM / NWM, WN = BN / 2, MT = WM / 32, NT = WN / 32;
    constexpr int A_BYTES = NP * BM * 64, B_BYTES = NP * BN * 64, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int ARG = BM / 16 / NW, BRG = BN / 16 / NW;      // 16-row groups per wave and operand
    constexpr int DMA_PER_TILE = (ARG + BRG) * NP;             // DMA instructions one wave issues per K-tile
    constexpr int NPROD = NP == 3 ? 6 : 1;
    static_assert(BM % 64 == 0 && BN % 64 == 0, "tile");
    extern __shared__ __attribute__((aligned(1024))) char lds_raw[];

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: LDS-DMA destinations are wave-uniform (M0)
    const int wm = wave >> 1, wn = wave & 1;
    const int nblk = mtiles * ntiles * g.groups;
    const int bid = xcd_remap(blockIdx.x, nblk);
    const int grp = bid % g.groups;
    const int nt_id = (bid / g.groups) % ntiles;
    const int mt_id = bid / (g.groups * ntiles);
    const long m0 = (long)mt_id * BM;
    const int n0 = nt_id * BN;
    const int cpt = g.Cg / BK;
    const int KT = g.R * g.S * cpt;

    // ---- DMA source addressing.  Lane = (row-in-group = lane >> 2, LDS slot = lane & 3); it fetches the chunk that the
    // swizzle maps to its slot.  Row state as in conv_igemm.hip (block-relative offsets, separable tap masks, OOB rows
    // read zeros -- also through the LDS path).
    constexpr unsigned OOB = 0x80000000u;
    const int slot = lane & 3, rin = lane >> 2;
    const int chunk = slot ^ ((lane >> 4) & 3);                // (row >> 2) & 3 == (lane >> 4) & 3 for every row group
    const int hw = g.Ho * g.Wo;
    const int blk_b = (int)(((double)(unsigned)m0 + 0.5) * g.inv_hw);
    const int blk_rem = (int)((unsigned)m0 - (unsigned)blk_b * (unsigned)hw);
    const int blk_y = fast_div(blk_rem, g.inv_wo), blk_x = blk_rem - blk_y * g.Wo;
    const int rows_valid = (int)(g.M - m0 < BM ? g.M - m0 : BM);
    unsigned a_mask[ARG], a_voff[ARG], a_veff[ARG];
#pragma unroll
    for (int t = 0; t < ARG; ++t) {
        const int rowoff = 16 * (wave + NW * t) + rin;
        const bool okm = rowoff < rows_valid;
        const int tx = blk_x + rowoff, qx = fast_div(tx, g.inv_wo), x = tx - qx * g.Wo;
        const int ty = blk_y + qx, qy = fast_div(ty, g.inv_ho), y = ty - qy * g.Ho;
        const int ay = DGRAD ? y + g.pad : y * g.stride - g.pad, ax = DGRAD ? x + g.pad : x * g.stride - g.pad;
        unsigned colbits = 0, mask = 0;
        for (int q = 0; q < g.S; ++q) colbits |= ((unsigned)(DGRAD ? ax - q : ax + q) < (unsigned)g.W ? 1u : 0u) << q;
        for (int r = 0; r < g.R; ++r) mask |= ((unsigned)(DGRAD ? ay - r : ay + r) < (unsigned)g.H ? colbits : 0u) << (r * g.S);
        a_mask[t] = okm ? mask : 0u;
        const int rel = okm ? qy * g.H * g.W : 0;
        const int e = DGRAD ? (rel + ay * g.W + ax) * g.C : (rel + (ay + g.pad) * g.W + (ax + g.pad)) * g.C;
        a_voff[t] = (unsigned)(e + grp * g.Cg + chunk * 8) * 2u;
        a_veff[t] = OOB;
    }
    const long img_elems = (long)g.H * g.W * g.C;
    const long shift = DGRAD ? ((long)(g.R - 1) * g.W + (g.S - 1)) * g.C : ((long)g.pad * g.W + g.pad) * g.C;
    // one descriptor per plane would cost SGPRs; the plane offset goes into the scalar offset instead (< 2^31 bytes per
    // tensor is checked by the host)
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a_planes + (long)blk_b * img_elems - shift), 0, 0x7fffffff, 0x00020000);
    // B rows: n = n0 + row; weight planes [tap][N_total][Cg(k)] with this group's rows at grp*Ng
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(w_planes + (long)grp * g.Ng * g.Cg), 0, 0x7fffffff, 0x00020000);
    unsigned b_voff[BRG];
#pragma unroll
    for (int t = 0; t < BRG; ++t) b_voff[t] = (unsigned)((n0 + 16 * (wave + NW * t) + rin) * g.Cg + chunk * 8) * 2u;
    const long wtap_bytes = (long)g.N * g.Cg * 2;
    const long a_plane_bytes = a_plane_elems * 2, w_plane_bytes = w_plane_elems * 2;

    // K order of the input-gradient kernel: channel chunk outer, filter tap INNER.  The nine taps of one 32-channel
    // chunk re-read (shifted) the same pixel rows right away -- 64 bytes per row stay cache-resident -- whereas tap-major
    // order comes back to a pixel row only after sweeping all Cg channels of every row of the tile: 128 KB per workgroup
    // x 32 workgroups per XCD = the whole 4 MB L2 for the wide dY rows (PMC: 1.2 GB fetched per launch against 168 MB
    // of operands; 118 -> 152 TFLOP/s-equivalent on the 56x56 layer).  The forward kernel (narrower rows, and weight
    // tiles whose two 64-byte halves are then used back to back) measured 5 % faster tap-major and keeps that order.
    constexpr bool TAP_INNER = DGRAD;
    const int ntaps = g.R * g.S;
    auto issue = [&](int kt, int stage) {          // DMA of K-tile kt into LDS stage `stage`
        const int chunk_i = TAP_INNER ? kt / ntaps : kt % cpt, tap = TAP_INNER ? kt - chunk_i * ntaps : kt / cpt;
        const int c0 = chunk_i * BK;
        const int r = tap / g.S, q = tap - r * g.S;
        const long toff = (DGRAD ? -((long)r * g.W + q) : ((long)r * g.W + q)) * g.C + c0;
#pragma unroll
        for (int t = 0; t < ARG; ++t) a_veff[t] = ((a_mask[t] >> tap) & 1u) ? a_voff[t] : OOB;
        const long sa = (DGRAD ? shift + toff : toff) * 2;
        const long sb = tap * wtap_bytes + (long)c0 * 2;
        char* st = lds_raw + stage * STAGE_BYTES;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int t = 0; t < ARG; ++t)
                dma16(rs_a, st + pl * (BM * 64) + (wave + NW * t) * 1024, a_veff[t], (int)(sa + pl * a_plane_bytes));
#pragma unroll
            for (int t = 0; t < BRG; ++t)
                dma16(rs_b, st + A_BYTES + pl * (BN * 64) + (wave + NW * t) * 1024, b_voff[t], (int)(sb + pl * w_plane_bytes));
        }
    };

    // Two accumulator sets for NP = 3: the hi*hi products in `acc`, the five correction products (2^-8 ... 2^-16 of it) in
    // `accl`, added once at the end.  With ONE accumulator the small products are aligned against a large running sum
    // inside the matrix unit and lose their low bits by truncation: measured a NEGATIVE bias of 7e-8 x mean|y| per layer
    // (K = 1152; the exact-fp32 MFMA kernel: 1e-9) at an unchanged rms error -- harmless per layer, but a coherent error,
    // and end to end (26 layers) it tripled the deviation from the fp64 reference.
    constexpr int NACC = NP == 3 ? 2 : 1;
    f32x16 acc[MT][NT], accl[NACC == 2 ? MT : 1][NACC == 2 ? NT : 1];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][j][e] = 0.f;
                if (NACC == 2) accl[i][j][e] = 0.f;
            }

    // fragment read offsets: lane = row l31 of a 32-row block, k-chunk 2*step + h, swizzled slot
    const int sw = (l31 >> 2) & 3;
    auto frag = [&](const char* base, int row0, int step) -> bf16x8 {
        const int c = (2 * step + h) ^ sw;
        return *(const bf16x8*)(base + (row0 + l31) * 64 + c * 16);
    };

    // ---- pipeline.  One wave per SIMD (the LDS image leaves room for one workgroup per CU), so latency must be hidden
    // inside the wave: fragments are double-buffered per half K-tile (step), and the DMA runs NSTAGE - 1 K-tiles ahead.
    //   A  ds_read  F[1] <- (kt, step 1)
    //   B  24 MFMA on F[0]                                  (covers A)
    //   C  lgkmcnt(0): F[1] there, this wave is done with stage(kt);  vmcnt: this wave's part of tile kt+1 has landed
    //   D  ONE barrier per K-tile: tile kt+1 complete in LDS, stage(kt) free
    //   E  DMA tile kt+NSTAGE -> stage(kt);  ds_read F[0] <- (kt+1, step 0)
    //   F  24 MFMA on F[1]                                  (covers E)
    // A first version (fragments read at the top of each step, two barriers per K-tile) ran the matrix pipe at 50 % with
    // the DMA ablated: every step exposed its LDS round trip.
    bf16x8 F[2][MT + NT][NP];
    auto load_frags = [&](int buf, int stage, int step) {
        const char* As = lds_raw + stage * STAGE_BYTES;
        const char* Bs = As + A_BYTES;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int i = 0; i < MT; ++i) F[buf][i][pl] = frag(As + pl * (BM * 64), wm * WM + i * 32, step);
#pragma unroll
            for (int j = 0; j < NT; ++j) F[buf][MT + j][pl] = frag(Bs + pl * (BN * 64), wn * WN + j * 32, step);
        }
    };
    auto mma = [&](int buf) {
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};      // smallest terms first
#pragma unroll
        for (int pr = (NP == 3 ? 0 : 5); pr < 6; ++pr)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (NACC == 2 && pr < 5)
                        accl[i][j] = mfma_bf16p(F[buf][i][PA[pr]], F[buf][MT + j][PB[pr]], accl[i][j]);
                    else
                        acc[i][j] = mfma_bf16p(F[buf][i][NP == 3 ? PA[pr] : 0], F[buf][MT + j][NP == 3 ? PB[pr] : 0], acc[i][j]);
                }
    };
#define SBAR() __builtin_amdgcn_sched_barrier(0)
// scheduling groups: one MFMA followed by a few of the other instruction kinds -- a wave issues in order, so everything
// that is not placed BETWEEN two MFMAs (32 cycles apart) runs with the matrix pipe idle; PMC on the first version: 240
// non-MFMA instructions per K-tile issued in clusters = 1100 of 3500 cycles per K-tile with the pipe empty
#define SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
    constexpr int NMMA = MT * NT * NPROD;                 // MFMAs per half K-tile
    constexpr int NFR = (MT + NT) * NP;                   // fragment reads per half K-tile
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s) issue(s < KT ? s : KT - 1, s);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 1) * DMA_PER_TILE) : "memory");     // tile 0 has landed
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0xc07f);       // (so that the loop header is entered with no LDS read pending on any path)
    for (int kt = 0; kt < KT; ++kt) {
        const int stage = kt % NSTAGE, nstage = (kt + 1) % NSTAGE;
        SBAR();
        load_frags(1, stage, 1);                                  // A
        mma(0);                                                   // B
#pragma unroll
        for (int q = 0; q < NMMA; ++q) { SG(0x008, 1); if (q < NFR) { SG(0x100, 1); SG(0x006, 2); } }
        SBAR();
        __builtin_amdgcn_s_waitcnt(0xc07f);                       // C: lgkmcnt(0) (the builtin: hipcc's counter tracking
                                                                  //    sees it, an asm wait would be invisible to it)
        // this wave's part of tile kt+1 has landed (NSTAGE - 2 younger tiles may stay in flight; past the end the
        // pipeline re-fetches the last tile instead of branching, so the count is the same in every iteration)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * DMA_PER_TILE) : "memory");
        __builtin_amdgcn_s_barrier();                             // D
        SBAR();
        issue(kt + NSTAGE < KT ? kt + NSTAGE : KT - 1, stage);    // E
        load_frags(0, nstage, 0);
        mma(1);                                                   // F
#pragma unroll
        for (int q = 0; q < NMMA; ++q) {
            SG(0x008, 1);
            if (q < DMA_PER_TILE) { SG(0x020, 1); SG(0x006, 4); }
            else if (q - DMA_PER_TILE < NFR) { SG(0x100, 1); SG(0x006, 2); }
        }
        // F[0]'s reads retired long ago; saying so stops hipcc from waiting lgkmcnt(0) behind the NEXT iteration's F[1]
        // reads before it lets the first MFMA of step 0 go (it cannot count across the back edge)
        SBAR();
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
#undef SBAR
#undef SG
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (NACC == 2) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] += accl[i][j];
    }
    // epilogue shared with the fp32 kernels (LDS-staged 16-byte stores, bias / addend / ReLU, BatchNorm statistics); the
    // one-plane instantiations (bf16 mode) are the TYPED ones: their forward may store bf16 (fz.io bit 3)
    if constexpr (NP == 1)
        igemm_epilogue_typed<BM, BN, WM, WN, DGRAD>(acc, (float*)lds_raw, g, bias, addend, dst, bn_part, relu, m0, n0, grp, mt_id,
                                                    &fz, false);
    else
        igemm_epilogue<BM, BN, WM, WN, DGRAD>(acc, (float*)lds_raw, g, bias, addend, dst, bn_part, relu, m0, n0, grp, mt_id, &fz);
}

#include "conv_planes_persist.h"

// ---------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 on planes with the input rows RESIDENT in LDS for all nine taps ("halo" kernel)
// ---------------------------------------------------------------------------------------------------------------
// pconv_kernel fetches the A tile once per filter tap: nine (shifted) copies of the same pixel rows go L2 -> LDS, and
// that delivery -- not the matrix pipe -- bounds it (48 of the 72 KB per K-tile of the 256x128 tile are A).  In the flat
// NHWC pixel index the nine taps of a same-size convolution are the row offsets dr*W + dq, so the rows a 256-pixel tile
// needs for ALL taps are one contiguous range of 256 + 2W + 2 pixels.  This kernel keeps that range resident:
//   * K order: 16-channel chunk outer, filter tap inner.  Per chunk ONE image [256 + 2W + 2 rows][16 ch] per plane
//     (32-byte rows, <= 12 KB) is DMA'd and serves 9 taps x 24 MFMAs per wave; only the weights (BN x 16 ch per tap)
//     stream per tap: 12 + 9 * 12 KB instead of 9 * 36 KB per 16 channels of K (BN = 128, three planes).
//   * taps that fall outside the image (borders, and the pixels of the neighbouring image that the flat range drags
//     in) are not masked in the data: the lane reads its fragment from a 32-byte ZERO row instead (one v_cndmask on the
//     ADDRESS per 32-row block and tap).
//   * pipeline unit = one filter row (3 taps, 72 MFMAs per wave) per workgroup barrier; weights double-buffered per
//     unit, the image double-buffered per chunk and fetched in three slices while the previous chunk computes.
// Same fragment / accumulator scheme as pconv_kernel (two accumulator sets, smallest products first); the summation
// order over K differs from pconv_kernel's, so results agree with the other tiles to fp32 rounding, not bit for bit.
#ifndef PHALO_ABLATE
#define PHALO_ABLATE 0      // dev builds only (tools_dev/phalo_ablate.sh): 1 = no DMA, 2 = no DMA wait / barrier, 4 = no LDS reads,
                            // 8 = cycle stamps (start, prologue done, K loop done, stores retired) instead of BatchNorm partials
#endif
template <int BN, int NP, bool DGRAD>
__global__ __launch_bounds__(512, 1) void phalo_kernel(const unsigned short* __restrict__ a_planes, long a_plane_elems,
                                                       const unsigned short* __restrict__ w_planes, long w_plane_elems,
                                                       const float* __restrict__ bias, const float* __restrict__ addend,
                                                       float* __restrict__ dst, double* __restrict__ bn_part, ConvGeom g,
                                                       int relu, int mtiles, int ntiles, int agroups, BnBwdFuse fz) {
    constexpr int BM = 256, NW = 8, WM = 64, WN = BN / 2, MT = 2, NT = WN / 32;
    constexpr int AG_MAX = 12;                                   // 32-row groups of the image (W <= 63)
    constexpr int APLANE = (AG_MAX + 1) * 1024;                  // + one zero block behind the image of every plane
    constexpr int ABUF = NP * APLANE;
    constexpr int BBLK = BN / 32;                                // 1 KB blocks of one (tap, plane) weight tile
    constexpr int BTAP = NP * BN * 32, BSTAGE = 3 * BTAP;
    constexpr int B_OFF = 2 * ABUF;
    constexpr int ZERO = AG_MAX * 1024;                          // zero block of plane 0 in image buffer 0 (+ pl * APLANE)
    constexpr int A_SLICE = 2;                                    // image DMAs per wave and unit (see issue_a)
    constexpr int B_PER_WAVE = (3 * NP * BBLK + NW - 1) / NW;
    constexpr int NPROD = NP == 3 ? 6 : 1;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) char lds_raw[];

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nblk = mtiles * ntiles * g.groups;
    const int bid = xcd_remap(blockIdx.x, nblk);
    const int grp = bid % g.groups;
    const int nt_id = (bid / g.groups) % ntiles;
    const int mt_id = bid / (g.groups * ntiles);
    const long m0 = (long)mt_id * BM;
    const int n0 = nt_id * BN;
    const int W = g.W, cpt16 = g.Cg / 16, KT = 3 * cpt16;

#if PHALO_ABLATE & 8
    long long stamp[4];
    stamp[0] = __builtin_readcyclecounter();
#endif
    // ---- zero blocks (one per plane, in image buffer 0)
    if (tid < NP * 64) *(f32x4*)(lds_raw + (tid >> 6) * APLANE + ZERO + (tid & 63) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_waitcnt(0xc07f);                          // (published by the barrier at the end of the prologue)

    // ---- tap masks of this lane's two fragment rows (pixels m0 + wm*64 + i*32 + l31) and their image rows
    const int hw = g.Ho * g.Wo;
    unsigned amask2 = 0;                                         // both rows' 9-bit masks in one register
    const int arow0 = (wm * WM + l31) * 32;                      // row i: + i * 1024
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const long m = m0 + wm * WM + i * 32 + l31;
        const int bimg = (int)(((double)(unsigned)m + 0.5) * g.inv_hw);
        const int rem = (int)((unsigned)m - (unsigned)bimg * (unsigned)hw);
        const int y = fast_div(rem, g.inv_wo), x = rem - y * g.Wo;
        unsigned mask = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int yy = DGRAD ? y + 1 - r : y + r - 1, xx = DGRAD ? x + 1 - q : x + q - 1;
                mask |= ((unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)W ? 1u : 0u) << (r * 3 + q);
            }
        amask2 |= (m < g.M ? mask : 0u) << (16 * i);
    }
    // image row j holds pixel m0 - (W + 1) + j; tap (r, q) of local row t reads image row t + (W + 1) + dr*W + dq
    int trow[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) trow[r] = ((W + 1) + (DGRAD ? 1 - r : r - 1) * W) * 32;

    // ---- DMA addressing.  Lane = (row in the 32-row group = lane >> 1, 16-byte slot = lane & 1).  LDS rows are 32 bytes;
    // the two 16-byte halves of row j are stored swapped when bit 3 of j is set, so that 16 consecutive rows read at one
    // k-half (a quarter of a ds_read_b128) touch all 64 banks once for ANY tap shift (unswizzled: PMC showed half of the
    // LDS cycles as bank conflicts).  The DMA writes lane-linearly, so the swap is applied on the source address.
    const int shalf = (lane & 1) ^ ((lane >> 4) & 1);
    const long p_first = m0 - (W + 1);
    const long pbase = p_first > 0 ? p_first : 0;                 // folded into the descriptor: offsets stay small
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a_planes + pbase * g.C + grp * g.Cg), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(w_planes + (long)grp * g.Ng * g.Cg), 0, 0x7fffffff, 0x00020000);
    const long a_plane_bytes = a_plane_elems * 2, w_plane_bytes = w_plane_elems * 2;
    const long wtap_bytes = (long)g.N * g.Cg * 2;
    // image: wave w fetches the 32-row groups w and w + 8 of every plane; slice sl of a chunk = plane sl (NP = 3) --
    // two DMAs per wave and slice, the source offset of a group is the same for every plane and chunk
    unsigned a_voff[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int gi = wave + NW * v;
        const long p = p_first + 32 * gi + (lane >> 1);
        a_voff[v] = (gi < agroups && p >= 0 && p < g.M) ? (unsigned)((p - pbase) * g.C * 2 + shalf * 16) : OOB;
    }
    const bool g1 = wave + NW < agroups;                          // (wave-uniform) this wave has a second group
    auto issue_a = [&](int c16, int sl) {
        const bool live = c16 < cpt16;
        static_assert(A_SLICE == 2 || NP == 1, "one plane per slice");
        if (NP == 3 || sl == 0) {
            const int pl = NP == 3 ? sl : 0;
            char* base = lds_raw + (c16 & 1) * ABUF + pl * APLANE;
            const int soff = (int)(pl * a_plane_bytes) + c16 * 32;
            dma16(rs_a, base + wave * 1024, live ? a_voff[0] : OOB, soff);
            dma16(rs_a, g1 ? base + (wave + NW) * 1024 : lds_raw + ZERO, live ? a_voff[1] : OOB, soff);
        } else {
            dma16(rs_a, lds_raw + ZERO, OOB, 0);
            dma16(rs_a, lds_raw + ZERO, OOB, 0);
        }
    };
    // weights of filter row r, chunk c16 into weight stage `stage`: DMA e = wave + 8 u covers (tap column, plane, 1 KB
    // block) = (e / (NP BBLK), (e / BBLK) % NP, e % BBLK); BBLK divides 8, so a wave always fetches the same block rows
    static_assert(NW % BBLK == 0, "block index of a wave is fixed");
    const unsigned b_voff = (unsigned)((n0 + (wave % BBLK) * 32 + (lane >> 1)) * g.Cg * 2 + shalf * 16);
    auto issue_b = [&](int c16, int r, int stage) {
        const bool live = c16 < cpt16;
#pragma unroll
        for (int u = 0; u < B_PER_WAVE; ++u) {
            const int e = wave + NW * u;
            const int q = e / (NP * BBLK), pl = (e / BBLK) % NP, blk = e % BBLK;
            const bool real = e < 3 * NP * BBLK;
            char* dstp = lds_raw + (real ? B_OFF + stage * BSTAGE + q * BTAP + pl * (BN * 32) + blk * 1024 : ZERO);
            dma16(rs_b, dstp, live && real ? b_voff : OOB, (int)((r * 3 + q) * wtap_bytes + c16 * 32 + pl * w_plane_bytes));
        }
    };

    constexpr int NACC = NP == 3 ? 2 : 1;
    f32x16 acc[MT][NT], accl[NACC == 2 ? MT : 1][NACC == 2 ? NT : 1];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][j][e] = 0.f;
                if (NACC == 2) accl[i][j][e] = 0.f;
            }

    bf16x8 F[2][MT + NT][NP];
    const int bfrag = B_OFF + (wn * WN + l31) * 32 + ((h ^ ((l31 >> 3) & 1)) << 4);
    // fragments of step (filter row r, tap column q) of the chunk in image buffer cp, weights in stage `stage`
    auto load_frags = [&](int buf, int cp, int r, int q, int stage) {
        const int tap = r * 3 + q;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            // (the empty asm keeps the 2 x 9 x 2 selected addresses from being hoisted out of the K loop as loop
            //  invariants: 36 registers this kernel does not have -- they were spilled and reloaded behind vmcnt(0))
            unsigned am = amask2;
            int ar = arow0;
            asm volatile("" : "+v"(am), "+v"(ar));
            const int row32 = ar + i * 1024 + cp * ABUF + trow[r] + (DGRAD ? 1 - q : q - 1) * 32;   // image row * 32 (+ buffer)
            const int live = row32 + ((((row32 >> 8) ^ h) & 1) << 4);                           // swizzled k-half
            const int addr = ((am >> (tap + 16 * i)) & 1u) ? live : ZERO + h * 16;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) F[buf][i][pl] = *(const bf16x8*)(lds_raw + addr + pl * APLANE);
        }
        const char* bs = lds_raw + bfrag + stage * BSTAGE + q * BTAP;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int j = 0; j < NT; ++j) F[buf][MT + j][pl] = *(const bf16x8*)(bs + pl * (BN * 32) + j * 1024);
    };
    auto mma = [&](int buf) {
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};      // smallest terms first
#pragma unroll
        for (int pr = (NP == 3 ? 0 : 5); pr < 6; ++pr)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (NACC == 2 && pr < 5)
                        accl[i][j] = mfma_bf16p(F[buf][i][PA[pr]], F[buf][MT + j][PB[pr]], accl[i][j]);
                    else
                        acc[i][j] = mfma_bf16p(F[buf][i][NP == 3 ? PA[pr] : 0], F[buf][MT + j][NP == 3 ? PB[pr] : 0], acc[i][j]);
                }
    };
#define SBAR() __builtin_amdgcn_sched_barrier(0)
#define SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
    constexpr int NMMA = MT * NT * NPROD;
    constexpr int NFR = (MT + NT) * NP;
    constexpr int DPT = A_SLICE + B_PER_WAVE;

    // ---- prologue: image of chunk 0, weights of units 0 and 1, first slice of chunk 1's image
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) issue_a(0, sl);
    issue_b(0, 0, 0);
    issue_a(1, 0);
    issue_b(0, 1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPT) : "memory");          // image 0 and unit 0 have landed
    __builtin_amdgcn_s_barrier();                                        // (also publishes the zero blocks)
    load_frags(0, 0, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0xc07f);
#if PHALO_ABLATE & 8
    stamp[1] = __builtin_readcyclecounter();
#endif

    // One unit = filter row R of the chunk in image buffer CP; FB = fragment buffer holding its first step on entry.
    // kt = 3 * c16 + R.  Steps q = 0, 1 prefetch the next step's fragments under 24 MFMAs each; step 2 synchronises:
    // this wave's DMAs for unit kt + 1 have landed, barrier (unit kt + 1 and any image slice issued one unit ago are
    // complete in LDS; weight stage kt & 1 and -- after R = 2 -- this chunk's image buffer are free), then the DMAs of
    // unit kt + 2 and one image slice go out and the first fragments of unit kt + 1 are read under the last 24 MFMAs.
    auto unit = [&](int c16, auto R_, auto CP_, auto FB_) {
        constexpr int R = decltype(R_)::value, CP = decltype(CP_)::value, FB = decltype(FB_)::value;
        constexpr int ST = (3 * CP + R) & 1;                             // weight stage = kt & 1 (c16 parity = CP)
        SBAR();
        if (!(PHALO_ABLATE & 4)) load_frags(FB ^ 1, CP, R, 1, ST);
        mma(FB);
#pragma unroll
        for (int k = 0; k < NMMA; ++k) { SG(0x008, 1); if (k < NFR) { SG(0x100, 1); SG(0x006, 2); } }
        SBAR();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        SBAR();
        if (!(PHALO_ABLATE & 4)) load_frags(FB, CP, R, 2, ST);
        mma(FB ^ 1);
#pragma unroll
        for (int k = 0; k < NMMA; ++k) { SG(0x008, 1); if (k < NFR) { SG(0x100, 1); SG(0x006, 2); } }
        SBAR();
        __builtin_amdgcn_s_waitcnt(0xc07f);                              // all LDS reads of this unit are done
        if (!(PHALO_ABLATE & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's part of unit kt + 1 (+ slice) landed
            __builtin_amdgcn_s_barrier();
        }
        SBAR();
        // DMAs: weights of unit kt + 2 into this unit's stage; one slice of an upcoming image
        constexpr int R2 = (R + 2) % 3, DC2 = (R + 2) / 3;
        if (!(PHALO_ABLATE & 1)) {
            issue_b(c16 + DC2, R2, ST);
            if constexpr (R == 2) issue_a(c16 + 2, 0);                   // this chunk's buffer is free now
            else issue_a(c16 + 1, R + 1);
        }
        // first fragments of unit kt + 1
        constexpr int R1 = (R + 1) % 3, CP1 = R == 2 ? CP ^ 1 : CP;
        if (!(PHALO_ABLATE & 4)) load_frags(FB ^ 1, CP1, R1, 0, ST ^ 1);
        mma(FB);
#pragma unroll
        for (int k = 0; k < NMMA; ++k) {
            SG(0x008, 1);
            if (k < DPT) { SG(0x020, 1); SG(0x006, 4); }
            else if (k - DPT < NFR) { SG(0x100, 1); SG(0x006, 2); }
        }
        SBAR();
        __builtin_amdgcn_s_waitcnt(0xc07f);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    for (int c = 0; c < cpt16; c += 2) {                                 // Cg % 32 == 0: an even number of chunks
        unit(c, I0{}, I0{}, I0{});
        unit(c, I1{}, I0{}, I1{});
        unit(c, I2{}, I0{}, I0{});
        unit(c + 1, I0{}, I1{}, I1{});
        unit(c + 1, I1{}, I1{}, I0{});
        unit(c + 1, I2{}, I1{}, I1{});
    }
#undef SBAR
#undef SG
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (NACC == 2) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] += accl[i][j];
    }
    (void)KT;
#if PHALO_ABLATE & 8
    stamp[2] = __builtin_readcyclecounter();
    igemm_epilogue<BM, BN, WM, WN, DGRAD>(acc, (float*)lds_raw, g, bias, addend, dst, nullptr, relu, m0, n0, grp, mt_id, &fz);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp[3] = __builtin_readcyclecounter();
    if (tid == 0 && bn_part) {           // dev: [block][4] cycle stamps + the CU-local start order
        long long* o = (long long*)bn_part + (long)blockIdx.x * 4;
        o[0] = stamp[0]; o[1] = stamp[1]; o[2] = stamp[2]; o[3] = stamp[3];
    }
#else
    if constexpr (NP == 1)
        igemm_epilogue_typed<BM, BN, WM, WN, DGRAD>(acc, (float*)lds_raw, g, bias, addend, dst, bn_part, relu, m0, n0, grp, mt_id,
                                                    &fz, false);
    else
        igemm_epilogue<BM, BN, WM, WN, DGRAD>(acc, (float*)lds_raw, g, bias, addend, dst, bn_part, relu, m0, n0, grp, mt_id, &fz);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static int conv_out_p(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }

extern "C" int scouter_planes_split_f32(const float* x, void* planes, long n, int nplanes, void* stream) {
    SC_REQUIRE(x && planes && n > 0 && n % 4 == 0 && (nplanes == 1 || nplanes == 3), "planes_split: bad arguments");
    const long n4 = n / 4;
    long nb = (n4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nplanes == 3)
        hipLaunchKernelGGL(split_planes_kernel<3>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x,
                           (unsigned short*)planes, n4, n);
    else
        hipLaunchKernelGGL(split_planes_kernel<1>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x,
                           (unsigned short*)planes, n4, n);
    return sc_check_launch("planes_split");
}

extern "C" int scouter_planes_split_weight_f32(const float* w_hwio, void* w_fwd, void* w_dgrad, int kh, int kw, int Cin,
                                               int Cout, int groups, int nplanes, void* stream) {
    SC_REQUIRE(w_hwio && (w_fwd || w_dgrad) && groups > 0 && Cin % groups == 0 && Cout % groups == 0 &&
               (nplanes == 1 || nplanes == 3), "planes_split_weight: bad arguments");
    const long n = (long)kh * kw * (Cin / groups) * Cout;
    long nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nplanes == 3)
        hipLaunchKernelGGL(split_weight_kernel<3>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w_hwio,
                           (unsigned short*)w_fwd, (unsigned short*)w_dgrad, kh * kw, Cin / groups, Cout, groups);
    else
        hipLaunchKernelGGL(split_weight_kernel<1>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w_hwio,
                           (unsigned short*)w_fwd, (unsigned short*)w_dgrad, kh * kw, Cin / groups, Cout, groups);
    return sc_check_launch("planes_split_weight");
}

extern "C" size_t scouter_planes_split_weights_row_bytes(void) { return sizeof(SplitWeightRow); }
// table: `nrows` device-resident rows {w_hwio, w_fwd, w_dgrad (either may be NULL), taps, Cin/groups, Cout, groups, first}
// (layout of SplitWeightRow: three pointers, four ints, one long = 48 bytes); total = sum of taps * Cin/groups * Cout
extern "C" int scouter_planes_split_weights_multi(const void* table, int nrows, long total, int nplanes, void* stream) {
    SC_REQUIRE(table && nrows > 0 && total > 0 && (nplanes == 1 || nplanes == 3), "planes_split_weights_multi: bad arguments");
    SC_UNSUPPORTED((total & 1023) == 0, "planes_split_weights_multi: every tensor needs 32-multiples of channels per group "
                   "and of output channels (total %ld)", total);
    long nb = total >> 10;                     // one [32 ci][32 co] tile per workgroup and round
    if (nb > 4096) nb = 4096;
    ScProfScope prof(nplanes == 3 ? "split_weights<bf16x3>" : "split_weights<bf16>", (hipStream_t)stream, 0.0,
                     (4.0 + 4.0 * nplanes) * (double)total);
    if (nplanes == 3)
        hipLaunchKernelGGL(split_weight_multi_kernel<3>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                           (const SplitWeightRow*)table, nrows, total);
    else
        hipLaunchKernelGGL(split_weight_multi_kernel<1>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                           (const SplitWeightRow*)table, nrows, total);
    return sc_check_launch("planes_split_weights_multi");
}

template <int BM, int BN, int NP, int NSTAGE, bool DGRAD, int NWM = 2>
static void launch_pconv(const void* a, long a_pe, const void* w, long w_pe, const float* bias, const float* addend,
                         float* dst, double* bn_part, const ConvGeom& g, int relu, hipStream_t st, const BnBwdFuse& fz) {
    const int mtiles = sc_cdiv(g.M, BM), ntiles = g.Ng / BN;
    ConvGeom gg = g;
    gg.inv_hw = 1.0 / ((double)g.Ho * g.Wo);
    gg.inv_wo = 1.0f / (float)g.Wo;
    gg.inv_ho = 1.0f / (float)g.Ho;
    constexpr int STAGE_BYTES = NP * (BM + BN) * 64;
    constexpr int EPI_BYTES = 2 * NWM * (BM / NWM) * (BN / 2 + 4) * 4;
    const size_t lds = (size_t)(NSTAGE * STAGE_BYTES > EPI_BYTES ? NSTAGE * STAGE_BYTES : EPI_BYTES);
    auto kern = pconv_kernel<BM, BN, NP, NSTAGE, DGRAD, NWM>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(mtiles * ntiles * g.groups), dim3(NWM * 128), lds, st, (const unsigned short*)a, a_pe,
                       (const unsigned short*)w, w_pe, bias, addend, dst, bn_part, gg, relu, mtiles, ntiles, fz);
}

// persistent 256 x BN kernel (conv_planes_persist.h): one workgroup per CU walks the tile list
static int sc_persistent_grid() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
            n = 256;
        cus = n / 8 * 8;
    }
    return cus;
}
template <int BN, int NP, bool DGRAD>
static void launch_ppersist(const void* a, long a_pe, const void* w, long w_pe, const float* bias, const float* addend,
                            float* dst, double* bn_part, const ConvGeom& g, int relu, hipStream_t st, const BnBwdFuse& fz) {
    constexpr int BM = 256;
    const int mtiles = sc_cdiv(g.M, BM), ntiles = g.Ng / BN;
    ConvGeom gg = g;
    gg.inv_hw = 1.0 / ((double)g.Ho * g.Wo);
    gg.inv_wo = 1.0f / (float)g.Wo;
    gg.inv_ho = 1.0f / (float)g.Ho;
    constexpr int LDS = 2 * NP * (BM + BN) * 64 + 1024;
    auto kern = ppersist_kernel<BN, NP, DGRAD>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    const int ntot = mtiles * ntiles * g.groups, cus = sc_persistent_grid();
    hipLaunchKernelGGL(kern, dim3(ntot <= cus ? ntot : cus), dim3(512), LDS, st, (const unsigned short*)a, a_pe,
                       (const unsigned short*)w, w_pe, bias, addend, dst, bn_part, gg, relu, mtiles, ntiles, fz);
}

template <int BN, int NP, bool DGRAD>
static void launch_phalo(const void* a, long a_pe, const void* w, long w_pe, const float* bias, const float* addend,
                         float* dst, double* bn_part, const ConvGeom& g, int relu, hipStream_t st, const BnBwdFuse& fz) {
    constexpr int BM = 256;
    const int mtiles = sc_cdiv(g.M, BM), ntiles = g.Ng / BN;
    ConvGeom gg = g;
    gg.inv_hw = 1.0 / ((double)g.Ho * g.Wo);
    gg.inv_wo = 1.0f / (float)g.Wo;
    gg.inv_ho = 1.0f / (float)g.Ho;
    constexpr int IMG = 2 * NP * 13 * 1024, WST = 2 * 3 * NP * BN * 32, EPI = 8 * 64 * (BN / 2 + 4) * 4;
    constexpr int LDS = IMG + WST > EPI ? IMG + WST : EPI;
    auto kern = phalo_kernel<BN, NP, DGRAD>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    const int agroups = (BM + 2 * g.W + 2 + 31) / 32;
    hipLaunchKernelGGL(kern, dim3(mtiles * ntiles * g.groups), dim3(512), LDS, st, (const unsigned short*)a, a_pe,
                       (const unsigned short*)w, w_pe, bias, addend, dst, bn_part, gg, relu, mtiles, ntiles, agroups, fz);
}
// the resident-rows kernel covers same-size 3x3 convolutions on maps up to 63 pixels wide
static bool phalo_ok(const ConvGeom& g) {
    return g.R == 3 && g.S == 3 && g.stride == 1 && g.pad == 1 && g.H == g.Ho && g.W == g.Wo && g.W <= 63 &&
           g.Cg % 32 == 0 && g.Ng % 64 == 0;
}

// tile: 0 = 128x128, 1 = 128x64 (N per group must be a multiple of the tile's N)
template <bool DGRAD>
static int dispatch_pconv(const void* a, long a_pe, const void* w, long w_pe, const float* bias, const float* addend,
                          float* dst, double* bn_part, const ConvGeom& g, int relu, int nplanes, int tile,
                          hipStream_t st, const BnBwdFuse& fz = BnBwdFuse{}) {
    const bool wide = g.Ng % 128 == 0 && tile != 1;
    // tile 6: persistent 256 x (128 | 64); its register epilogue takes BatchNorm statistics from the accumulators, so a
    // launch that wants statistics of bias / addend-shifted values keeps the LDS-staged epilogue of tile 4 / 2
    // -- that request is REFUSED, not silently re-routed: the caller sized bn_part / the fused partial rows for the tile
    // it named (tile 6: four rows per 256-row tile), another tile writes another row layout
    if (tile == 6 && ((bn_part && (bias || addend)) || g.R * g.S * (g.Cg / 32) < 2)) {
        sc_set_error("conv planes: tile 6 (persistent) cannot take BatchNorm statistics of bias / addend-shifted outputs "
                     "and needs at least two K-tiles of 32 channels x taps; name another tile");
        return SC_ERR_UNSUPPORTED;
    }
    if (tile == 6) {
        if (nplanes == 3) {
            if (g.Ng % 128 == 0) launch_ppersist<128, 3, DGRAD>(a, a_pe, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz);
            else launch_ppersist<64, 3, DGRAD>(a, a_pe, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz);
        } else {
            if (g.Ng % 128 == 0) launch_ppersist<128, 1, DGRAD>(a, a_pe, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz);
            else launch_ppersist<64, 1, DGRAD>(a, a_pe, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz);
        }
    } else if (nplanes == 3 && tile == 5 && phalo_ok(g)) {   // 256 x (128 | 64), input rows resident in LDS for all nine taps
        if (g.Ng % 128 == 0) launch_phalo<128, 3, DGRAD>(a, a_pe, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz);
        else launch_phalo<64, 3, DGRAD>(a, a_pe, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz);
    } else if (nplanes == 1 && tile == 5 && phalo_ok(g)) {
        if (g.Ng % 128 == 0) launch_phalo<128, 1, DGRAD>(a, a_pe, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz);
        else launch_phalo<64, 1, DGRAD>(a, a_pe, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz);
    } else if (nplanes == 3 && tile == 2) {        // two workgroups per CU (2 LDS stages of 36 KB)
        launch_pconv<128, 64, 3, 2, DGRAD>(a, a_pe, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz);
    } else if (nplanes == 3 && tile == 3) { // three workgroups per CU
        launch_pconv<64, 64, 3, 2, DGRAD>(a, a_pe, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz);
    } else if (nplanes == 3 && tile == 4 && g.Ng % 128 == 0) {   // 256 x 128, eight waves: 25 % fewer L2 -> LDS bytes
        launch_pconv<256, 128, 3, 2, DGRAD, 4>(a, a_pe, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz);
    } else if (nplanes == 3) {
        if (wide) launch_pconv<128, 128, 3, 3, DGRAD>(a, a_pe, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz);
        else launch_pconv<128, 64, 3, 3, DGRAD>(a, a_pe, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz);
    } else {
        if (wide) launch_pconv<128, 128, 1, 3, DGRAD>(a, a_pe, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz);
        else launch_pconv<128, 64, 1, 3, DGRAD>(a, a_pe, w, w_pe, bias, addend, dst, bn_part, g, relu, st, fz);
    }
    return sc_check_launch(DGRAD ? "conv2d_dgrad_planes" : "conv2d_fwd_planes");
}

extern "C" int scouter_conv2d_fwd_planes_bn_partial_rows(int B, int H, int W, int kh, int kw, int stride, int pad) {
    // upper bound over every tile: tile 6 writes one row per 64-row WAVE row of each 256-row tile, incl. the all-zero
    // ones beyond M (4 * ceil(M / 256) >= ceil(M / 64), the count of the 64-row tile 3)
    return 4 * sc_cdiv((long)B * conv_out_p(H, kh, stride, pad) * conv_out_p(W, kw, stride, pad), 256);
}

// x_planes: [nplanes][B*H*W][Cin] bf16; w_planes: scouter_planes_split_weight_f32's forward layout.
// `io` & SC_IO_Y_BF16: y is stored as bf16 (RNE of the fp32 result; bn_partial still sums the fp32 accumulators); tiles 0-5
extern "C" int scouter_conv2d_fwd_planes_io(const void* x_planes, const void* w_planes, const float* bias,
                                            const float* addend, void* y, double* bn_partial, int B, int H, int W, int Cin,
                                            int Cout, int kh, int kw, int stride, int pad, int groups, int relu, int nplanes,
                                            int tile, int io, void* stream) {
    SC_REQUIRE(x_planes && w_planes && y && B > 0 && H > 0 && W > 0 && (nplanes == 1 || nplanes == 3),
               "conv2d_fwd_planes: bad arguments");
    SC_REQUIRE((io & ~SC_IO_Y_BF16) == 0, "conv2d_fwd_planes: unsupported io bits %d (only y may be bf16)", io);
    SC_UNSUPPORTED(!(io & SC_IO_Y_BF16) || tile != 6, "conv2d_fwd_planes: the persistent tile 6 writes fp32 only");
    SC_REQUIRE(!(bn_partial && (relu & 1)), "conv2d_fwd_planes: fused BatchNorm statistics are taken before any activation");
    SC_REQUIRE(groups > 0 && Cin % groups == 0 && Cout % groups == 0, "conv2d_fwd_planes: channels not divisible by groups");
    const int Cg = Cin / groups, Ng = Cout / groups;
    SC_UNSUPPORTED(Cg % 32 == 0 && Ng % 64 == 0, "conv2d_fwd_planes: needs Cin/groups %% 32 == 0 and Cout/groups %% 64 == 0");
    ConvGeom g{B, H, W, Cin, conv_out_p(H, kh, stride, pad), conv_out_p(W, kw, stride, pad), Cout, kh, kw, stride, pad,
               groups, Cg, Ng, 0, Cout, Cg * Cout};
    g.M = (long)B * g.Ho * g.Wo;
    const long a_pe = (long)B * H * W * Cin, w_pe = (long)kh * kw * Cg * Cout;
    SC_UNSUPPORTED(g.M < (1L << 31) && (long)nplanes * a_pe * 2 < (1L << 31) + ((long)H * W * Cin * 2) &&
                   (long)H * W * Cin < (1L << 28), "conv2d_fwd_planes: tensor too large for 32-bit plane offsets");
    ScProfScope prof(nplanes == 3 ? "pconv_fwd<bf16x3>" : "pconv_fwd<bf16>", (hipStream_t)stream,
                     2.0 * g.M * Cout * Cg * kh * kw,
                     2.0 * nplanes * ((double)a_pe) + ((io & SC_IO_Y_BF16) ? 2.0 : 4.0) * (double)g.M * Cout);
    BnBwdFuse fz{};
    fz.io = (io & SC_IO_Y_BF16) ? 8 : 0;             // (the shared block epilogue reads the output's storage type here)
    return dispatch_pconv<false>(x_planes, a_pe, w_planes, w_pe, bias, addend, (float*)y, bn_partial, g, relu, nplanes, tile,
                                 (hipStream_t)stream, fz);
}
extern "C" int scouter_conv2d_fwd_planes(const void* x_planes, const void* w_planes, const float* bias,
                                         const float* addend, float* y, double* bn_partial, int B, int H, int W, int Cin,
                                         int Cout, int kh, int kw, int stride, int pad, int groups, int relu, int nplanes,
                                         int tile, void* stream) {
    return scouter_conv2d_fwd_planes_io(x_planes, w_planes, bias, addend, y, bn_partial, B, H, W, Cin, Cout, kh, kw, stride,
                                        pad, groups, relu, nplanes, tile, 0, stream);
}

// dy_planes: [nplanes][B*Ho*Wo][Cout]; w_planes: the dgrad layout.  Stride-1 convolutions only.
extern "C" int scouter_conv2d_dgrad_planes_bnbwd(const void* dy_planes, const void* w_planes, const float* addend,
                                                 float* dx, int B, int H, int W, int Cin, int Cout, int kh, int kw,
                                                 int stride, int pad, int groups, int nplanes, int tile,
                                                 const void* relu_mask, const float* x1, const float* saved1,
                                                 double* part1, const float* x2, const float* saved2, double* part2,
                                                 void* stream) {
    SC_REQUIRE(dy_planes && w_planes && dx && B > 0 && (nplanes == 1 || nplanes == 3), "conv2d_dgrad_planes: bad arguments");
    SC_REQUIRE(groups > 0 && Cin % groups == 0 && Cout % groups == 0, "conv2d_dgrad_planes: channels not divisible by groups");
    SC_REQUIRE(!part1 || (x1 && saved1), "conv2d_dgrad_planes: fused BatchNorm backward needs x1 and saved1");
    SC_REQUIRE(!part2 || (part1 && x2 && saved2), "conv2d_dgrad_planes: second fused BatchNorm needs the first, x2 and saved2");
    SC_UNSUPPORTED(stride == 1, "conv2d_dgrad_planes: stride-1 convolutions only");
    const int Cig = Cin / groups, Cog = Cout / groups;
    SC_UNSUPPORTED(Cog % 32 == 0 && Cig % 64 == 0, "conv2d_dgrad_planes: needs Cout/groups %% 32 == 0 and Cin/groups %% 64 == 0");
    const int Ho = conv_out_p(H, kh, stride, pad), Wo = conv_out_p(W, kw, stride, pad);
    ConvGeom g{B, Ho, Wo, Cout, H, W, Cin, kh, kw, stride, pad, groups, Cog, Cig, 0, Cout, Cig * Cout};
    g.M = (long)B * H * W;
    const long a_pe = (long)B * Ho * Wo * Cout, w_pe = (long)kh * kw * Cig * Cout;
    SC_UNSUPPORTED(g.M < (1L << 31) && (long)nplanes * a_pe * 2 < (1L << 31) + ((long)Ho * Wo * Cout * 2) &&
                   (long)Ho * Wo * Cout < (1L << 28), "conv2d_dgrad_planes: tensor too large for 32-bit plane offsets");
    ScProfScope prof(nplanes == 3 ? "pconv_dgrad<bf16x3>" : "pconv_dgrad<bf16>", (hipStream_t)stream,
                     2.0 * g.M * Cin * Cog * kh * kw, 2.0 * nplanes * ((double)a_pe) + 4.0 * (double)g.M * Cin);
    const BnBwdFuse fz{part1 ? (const unsigned long long*)relu_mask : nullptr, x1, saved1, part1, x2, saved2, part2};
    return dispatch_pconv<true>(dy_planes, a_pe, w_planes, w_pe, nullptr, addend, dx, nullptr, g, 0, nplanes, tile,
                                (hipStream_t)stream, fz);
}

extern "C" int scouter_conv2d_dgrad_planes(const void* dy_planes, const void* w_planes, const float* addend, float* dx,
                                           int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad,
                                           int groups, int nplanes, int tile, void* stream) {
    return scouter_conv2d_dgrad_planes_bnbwd(dy_planes, w_planes, addend, dx, B, H, W, Cin, Cout, kh, kw, stride, pad,
                                             groups, nplanes, tile, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                             nullptr, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient on planes:  dW[tap][ci][co] = sum_m X[m + tapoff][ci] * dY[m][co]   ("same" stride-1 convolutions)
// ---------------------------------------------------------------------------------------------------------------
// The contraction index is the PIXEL, but both operands are stored pixel-major / channel-contiguous, while a bf16 MFMA
// lane wants 8 consecutive k of one row.  ds_read_b64_tr_b16 does that transpose in the LDS read: the 16 lanes of a group
// each point at 4 contiguous bf16 of a [4 rows][16 columns] block (lane t: row t >> 2, columns 4 (t & 3) ..) and lane t
// RECEIVES column t of the four rows (measured with tools_dev/tr_b16_probe.hip).  So the LDS image stays the natural one
// -- [32 pixels][BM channels] per operand plane, filled by LDS-DMA exactly like the forward kernel -- and a fragment
// (row = channel l & 31, k = pixels 8 (l >> 5) .. + 7) is two transposing reads.  Bank conflicts: the four pixel rows a
// 32-lane half touches are BM * 2 bytes apart (a multiple of 256 for BM = 128): the 16-byte slot s of pixel row r is
// stored at slot s ^ 4 (r & 3) [BM = 128] resp. s ^ 4 ((r >> 1) & 1) [BM = 64], again applied on the DMA's source side.
template <int OFF>    // OFF: compile-time byte offset (the instruction's 16-bit offset field): no address VALU per read
__device__ __forceinline__ unsigned long long ds_read_tr16(unsigned addr) {
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

template <int CH>   // channels per LDS row (BM or BN): slot swizzle of pixel row r
__device__ __forceinline__ int px_swz(int r) { return CH == 128 ? 4 * (r & 3) : 4 * ((r >> 1) & 1); }

template <int BM, int BN, int NP, int NSTAGE>
__global__ __launch_bounds__(256, 1) void pwgrad_kernel(const unsigned short* __restrict__ x_planes, long x_pe,
                                                        const unsigned short* __restrict__ dy_planes, long dy_pe,
                                                        float* __restrict__ out, ConvGeom g, int ci_tiles, int co_tiles,
                                                        long pix_per_split, long slab, float* __restrict__ dw,
                                                        unsigned* __restrict__ arrival) {
    constexpr int BK = 32, WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32;
    constexpr int A_BYTES = NP * BK * BM * 2, B_BYTES = NP * BK * BN * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int APX = 512 / BM, BPX = 512 / BN;               // pixel rows one DMA instruction covers (1 KB)
    constexpr int AQ = BK / APX / 4, BQ = BK / BPX / 4;         // DMA instructions per wave, operand and plane
    constexpr int DMA_PER_TILE = (AQ + BQ) * NP;
    constexpr int NACC = NP == 3 ? 2 : 1;
    static_assert((BM == 64 || BM == 128) && (BN == 64 || BN == 128), "tile");
    extern __shared__ __attribute__((aligned(1024))) char lds_raw[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_raw;

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
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
    const int KT = (int)((mend - mbeg + BK - 1) / BK);
    const long tapoff = (long)(r - g.pad) * g.W + (q - g.pad);           // source pixel = m + tapoff

    // ---- tap validity of the chunk's pixels: lane l tests pixel (l & 31) of the chunk, a ballot makes it scalar
    // (branch-free walk over (y, x): at most one row wrap and one image wrap per 32-pixel chunk, checked by the host)
    const int hw = g.Ho * g.Wo;
    const int blk_b = (int)(((double)(unsigned long)mbeg + 0.5) * g.inv_hw);
    const int blk_rem = (int)(mbeg - (long)blk_b * hw);
    const int blk_y = fast_div(blk_rem, g.inv_wo), blk_x = blk_rem - blk_y * g.Wo;
    int qx1, qy1;
    {
        const int tx = blk_x + l31, qx = fast_div(tx, g.inv_wo);
        const int ty = blk_y + qx, qy = fast_div(ty, g.inv_ho);
        qx1 = tx - qx * g.Wo;
        qy1 = ty - qy * g.Ho;
    }
    const int adv_x = BK % g.Wo, adv_y = BK / g.Wo;

    // ---- DMA lane geometry: instruction iq of this wave covers pixel rows APX * (wave + 4 iq) .. of the chunk
    constexpr unsigned OOB = 0x80000000u;
    constexpr int ASL = BM / 8, BSL = BN / 8;                    // 16-byte slots per pixel row
    unsigned a_voff[AQ], a_bit[AQ], b_voff[BQ];
#pragma unroll
    for (int iq = 0; iq < AQ; ++iq) {
        const int rq = APX * (wave + 4 * iq) + lane / ASL, sl = lane % ASL;
        const int c = sl ^ px_swz<BM>(rq);                       // channel chunk this lane fetches for its slot
        a_voff[iq] = (unsigned)(((long)rq * g.C + grp * g.Cg + ci0 + 8 * c) * 2);
        a_bit[iq] = 1u << rq;
    }
#pragma unroll
    for (int iq = 0; iq < BQ; ++iq) {
        const int rq = BPX * (wave + 4 * iq) + lane / BSL, sl = lane % BSL;
        const int c = sl ^ px_swz<BN>(rq);
        b_voff[iq] = (unsigned)(((long)rq * g.N + grp * g.Ng + co0 + 8 * c) * 2);
    }
    auto records = [&](long m_chunk, int row_elems) {            // bytes of the rows m_chunk .. mend-1
        long n = (mend - m_chunk) * (long)row_elems * 2;
        return (unsigned)(n < 0 ? 0 : (n > 0x7fffffffL ? 0x7fffffffL : n));
    };
    long next_m = mbeg;                                           // first pixel of the chunk the next issue() fetches
    auto issue = [&](int stage) {
        const long m = next_m;
        const unsigned rec_a = records(m, g.C), rec_b = records(m, g.N);
        const int iy = qy1 - g.pad + r, ix = qx1 - g.pad + q;
        const unsigned vmask = (unsigned)__ballot((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W);
        qx1 += adv_x;
        const int wrap = qx1 >= g.Wo ? 1 : 0;
        qx1 -= wrap ? g.Wo : 0;
        qy1 += adv_y + wrap;
        qy1 -= qy1 >= g.Ho ? g.Ho : 0;
        char* st = lds_raw + stage * STAGE_BYTES;
        unsigned a_veff[AQ];
#pragma unroll
        for (int iq = 0; iq < AQ; ++iq) a_veff[iq] = (vmask & a_bit[iq]) ? a_voff[iq] : OOB;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            // one descriptor per plane (scalar work): num_records clips the chunk at the end of the pixel range, and the
            // range check covers the scalar offset too -- a plane offset passed there read zeros for planes 1 and 2
            const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(x_planes + pl * x_pe + (m + tapoff) * g.C), 0, rec_a, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(dy_planes + pl * dy_pe + m * g.N), 0, rec_b, 0x00020000);
#pragma unroll
            for (int iq = 0; iq < AQ; ++iq)
                dma16(rs_a, st + pl * (BK * BM * 2) + (wave + 4 * iq) * 1024, a_veff[iq], 0);
#pragma unroll
            for (int iq = 0; iq < BQ; ++iq)
                dma16(rs_b, st + A_BYTES + pl * (BK * BN * 2) + (wave + 4 * iq) * 1024, b_voff[iq], 0);
        }
        next_m += BK;
    };

    f32x16 acc[MT][NT], accl[NACC == 2 ? MT : 1][NACC == 2 ? NT : 1];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][j][e] = 0.f;
                if (NACC == 2) accl[i][j][e] = 0.f;
            }

    // ---- transposing fragment reads.  Lane l, read u (k sub-block of 4): pixel row 16 step + 8 h + 4 u + ((l >> 2) & 3),
    // columns blockbase + 16 ((l >> 4) & 1) + 4 (l & 3) .. + 3
    const int r2 = (lane >> 2) & 3, g1 = (lane >> 4) & 1, c4 = 4 * (lane & 3);
    unsigned fa_off[MT], fb_off[NT];                              // lane's byte offset inside an operand plane image
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int col = wm * WM + 32 * i + 16 * g1 + c4, row0 = 8 * h + r2;
        fa_off[i] = (unsigned)(row0 * (BM * 2) + (((col >> 3) ^ px_swz<BM>(row0)) << 4) + (col & 7) * 2);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = wn * WN + 32 * j + 16 * g1 + c4, row0 = 8 * h + r2;
        fb_off[j] = (unsigned)(row0 * (BN * 2) + (((col >> 3) ^ px_swz<BN>(row0)) << 4) + (col & 7) * 2);
    }
    // (rows 16 step + 4 u further down keep (row & 3) and, for 64-channel rows, flip ((row >> 1) & 1) only with u -> the
    // swizzle term is folded per u below)
    bf16x8 F[2][MT + NT][NP];
    // (4 u or 16 step rows further down change neither (row & 3) nor ((row >> 1) & 1): the swizzle term of a lane is fixed,
    // and plane / step / u offsets are instruction immediates -- one address VGPR per 32-channel block and stage)
    auto load_frags = [&](int buf, int stage, auto STEP) {
        constexpr int step = decltype(STEP)::value;
        const unsigned sbase = lds_base + stage * STAGE_BYTES;
        unsigned ab[MT], bb[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) ab[i] = sbase + fa_off[i];
#pragma unroll
        for (int j = 0; j < NT; ++j) bb[j] = sbase + fb_off[j];
        auto rd = [&](auto PL) {
            constexpr int pl = decltype(PL)::value;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                u64x2 v;
                v[0] = ds_read_tr16<pl * (BK * BM * 2) + (16 * step) * (BM * 2)>(ab[i]);
                v[1] = ds_read_tr16<pl * (BK * BM * 2) + (16 * step + 4) * (BM * 2)>(ab[i]);
                F[buf][i][pl] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                u64x2 v;
                v[0] = ds_read_tr16<A_BYTES + pl * (BK * BN * 2) + (16 * step) * (BN * 2)>(bb[j]);
                v[1] = ds_read_tr16<A_BYTES + pl * (BK * BN * 2) + (16 * step + 4) * (BN * 2)>(bb[j]);
                F[buf][MT + j][pl] = __builtin_bit_cast(bf16x8, v);
            }
        };
        rd(std::integral_constant<int, 0>{});
        if constexpr (NP == 3) { rd(std::integral_constant<int, 1>{}); rd(std::integral_constant<int, 2>{}); }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    auto mma = [&](int buf) {
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int pr = (NP == 3 ? 0 : 5); pr < 6; ++pr)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (NACC == 2 && pr < 5)
                        accl[i][j] = mfma_bf16p(F[buf][i][PA[pr]], F[buf][MT + j][PB[pr]], accl[i][j]);
                    else
                        acc[i][j] = mfma_bf16p(F[buf][i][NP == 3 ? PA[pr] : 0], F[buf][MT + j][NP == 3 ? PB[pr] : 0], acc[i][j]);
                }
    };
#define SBAR() __builtin_amdgcn_sched_barrier(0)
    // (the transposing reads are inline asm: hipcc does not count them, every wait is explicit)
#define LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
    if (KT > 0) {
#pragma unroll
        for (int s = 0; s < NSTAGE; ++s) issue(s);               // chunks past the end read zeros (num_records = 0)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 1) * DMA_PER_TILE) : "memory");
        __builtin_amdgcn_s_barrier();
        load_frags(0, 0, S0{});
        LGKM0();
        for (int kt = 0; kt < KT; ++kt) {
            const int stage = kt % NSTAGE, nstage = (kt + 1) % NSTAGE;
            SBAR();
            load_frags(1, stage, S1{});
            SBAR();
            mma(0);
            SBAR();
            LGKM0();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * DMA_PER_TILE) : "memory");
            __builtin_amdgcn_s_barrier();
            SBAR();
            issue(stage);
            load_frags(0, nstage, S0{});
            SBAR();
            mma(1);
            SBAR();
            LGKM0();
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
#undef SBAR
#undef LGKM0
    float* o = out + (long)split_id * slab + (long)tap * g.Cg * g.N;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ci = ci0 + wm * WM + i * 32 + mfma32_row(e, lane);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int co = grp * g.Ng + co0 + wn * WN + j * 32 + l31;
                slab_store(o + (long)ci * g.N + co, NACC == 2 ? acc[i][j][e] + accl[i][j][e] : acc[i][j][e], arrival != nullptr);
            }
        }
    if (arrival)        // the last workgroup of this tile sums the slabs itself (conv_common.h slab_tile_finish)
        slab_tile_finish(out + (long)tap * g.Cg * g.N, dw + (long)tap * g.Cg * g.N, slab, gridDim.y, arrival + tile_id, 1, 0,
                         (long)ci0 * g.N, BM, g.N, grp * g.Ng + co0, BN);
}

#include "conv_planes_wgrad_taps.h"

// (tile, split-K) plan of the plane weight gradient.  plan_hint < 0: static (largest tile, ~1024 workgroups); otherwise
// (autotuned by the caller) bits 0-1 = workgroup budget {512, 1024, 2048, 4096}, bit 4 / bit 5 = 64 instead of 128 input /
// output channels per tile.  Every plan is deterministic; different plans sum the pixels in a different order.
// bit 6 (3x3 only): TAP-FUSED kernel (conv_planes_wgrad_taps.h) -- 64 x 64 tiles that carry all nine taps, one workgroup per
// CU, budget {256, 512, 1024, 2048} workgroups.
struct PwgPlan { int bm, bn; long tiles, pps; int splits; bool fused; };
static PwgPlan pwg_plan(long M, int Cg, int Ng, int groups, int taps, int plan_hint) {
    PwgPlan p;
    p.bm = Cg % 128 == 0 ? 128 : 64;
    p.bn = Ng % 128 == 0 ? 128 : 64;
    p.fused = plan_hint >= 0 && (plan_hint & 64) && taps == 9;
    long budget = 1024;
    if (plan_hint >= 0) {
        budget = 512L << (plan_hint & 3);
        if (plan_hint & 16) p.bm = 64;
        if (plan_hint & 32) p.bn = 64;
    }
    if (p.fused) {
        p.bm = p.bn = 64;
        budget = 256L << (plan_hint & 3);
        taps = 1;                                    // (a tile carries all nine taps)
    }
    p.tiles = (long)(Cg / p.bm) * (Ng / p.bn) * groups * taps;
    long want = budget / p.tiles;
    if (want < 1) want = 1;
    const long chunks = (M + 31) / 32;
    long cps = (chunks + want - 1) / want;
    if (cps < 8) cps = 8;
    p.pps = cps * 32;
    p.splits = (int)((M + p.pps - 1) / p.pps);
    return p;
}

extern "C" size_t scouter_conv2d_wgrad_planes_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw,
                                                              int groups, int plan_hint) {
    if (groups <= 0 || Cin % groups || Cout % groups) return 0;
    const int Cg = Cin / groups, Ng = Cout / groups;
    const PwgPlan p = pwg_plan((long)B * H * W, Cg, Ng, groups, kh * kw, plan_hint);
    size_t need = p.splits > 1 ? (size_t)p.splits * kh * kw * Cg * Cout * sizeof(float) : 0;
#ifdef PWT_STAMPS
    need += (size_t)p.tiles * p.splits * 48;
#endif
    return need;
}

// x_planes [np][B*H*W][Cin], dy_planes [np][B*H*W][Cout] (stride 1, output size == input size); dw: HWIO fp32.
extern "C" int scouter_conv2d_wgrad_planes(const void* x_planes, const void* dy_planes, float* dw, int B, int H, int W,
                                           int Cin, int Cout, int kh, int kw, int pad, int groups, int nplanes,
                                           int plan_hint, void* ws, size_t ws_bytes, void* arrival, int arrival_slots,
                                           void* stream) {
    SC_REQUIRE(x_planes && dy_planes && dw && B > 0 && (nplanes == 1 || nplanes == 3), "conv2d_wgrad_planes: bad arguments");
    SC_REQUIRE(groups > 0 && Cin % groups == 0 && Cout % groups == 0, "conv2d_wgrad_planes: channels not divisible by groups");
    const int Cg = Cin / groups, Ng = Cout / groups;
    SC_UNSUPPORTED(kh == kw && 2 * pad == kh - 1 && Cg % 64 == 0 && Ng % 64 == 0 && 32 / W + 1 < H,
                   "conv2d_wgrad_planes: same-size convolutions with 64-multiples of channels per group only");
    ConvGeom g{B, H, W, Cin, H, W, Cout, kh, kw, 1, pad, groups, Cg, Ng, 0, Cout, Cg * Cout};
    g.M = (long)B * H * W;
    g.inv_hw = 1.0 / ((double)H * W);
    g.inv_wo = 1.0f / (float)W;
    g.inv_ho = 1.0f / (float)H;
    const long x_pe = g.M * Cin, dy_pe = g.M * Cout;
    SC_UNSUPPORTED(g.M < (1L << 31) && (long)nplanes * x_pe * 2 < (1L << 32) && (long)nplanes * dy_pe * 2 < (1L << 32),
                   "conv2d_wgrad_planes: tensor too large for 32-bit plane offsets");
    const PwgPlan plan = pwg_plan(g.M, Cg, Ng, groups, kh * kw, plan_hint);
    const int bm = plan.bm, bn = plan.bn;
    const int ci_tiles = Cg / bm, co_tiles = Ng / bn;
    const long tiles = plan.tiles;
    const long pps = plan.pps;
    const int splits = plan.splits;
    const long slab = (long)kh * kw * Cg * Cout;
    const size_t need = splits > 1 ? (size_t)splits * slab * sizeof(float) : 0;
    if (need > ws_bytes || (need && !ws)) {
        sc_set_error("conv2d_wgrad_planes: workspace too small (%zu < %zu bytes)", ws_bytes, need);
        return SC_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* out = splits > 1 ? (float*)ws : dw;
    unsigned* arr = splits > 1 && arrival && tiles <= arrival_slots ? (unsigned*)arrival : nullptr;
    dim3 grid((unsigned)tiles, (unsigned)splits);
    if (plan.fused) {
        SC_UNSUPPORTED(kh == 3 && kw == 3 && pad == 1 && W <= 63 && x_pe * 2 < (1L << 31) && dy_pe * 2 < (1L << 31),
                       "conv2d_wgrad_planes: the tap-fused plan covers 3x3 / pad 1 layers on maps up to 63 pixels wide");
        ScProfScope prof(nplanes == 3 ? "pwgrad<bf16x3>" : "pwgrad<bf16>", st, 2.0 * g.M * Cout * Cg * kh * kw,
                         2.0 * nplanes * ((double)x_pe + (double)dy_pe));
        const size_t lds = (size_t)nplanes * (256 * 128 + 128) + (size_t)3 * nplanes * 32 * 64 * 2;
        if (nplanes == 3) {
            auto kern = pwgrad_taps_kernel<3, 1>;
            static bool attr_set = false;
            if (!attr_set) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const unsigned short*)x_planes, x_pe,
                               (const unsigned short*)dy_planes, dy_pe, out, g, ci_tiles, co_tiles, pps, slab, dw, arr);
        } else {
            auto kern = pwgrad_taps_kernel<1, 1>;
            static bool attr_set = false;
            if (!attr_set) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const unsigned short*)x_planes, x_pe,
                               (const unsigned short*)dy_planes, dy_pe, out, g, ci_tiles, co_tiles, pps, slab, dw, arr);
        }
    } else {
        ScProfScope prof(nplanes == 3 ? "pwgrad<bf16x3>" : "pwgrad<bf16>", st, 2.0 * g.M * Cout * Cg * kh * kw,
                         2.0 * nplanes * ((double)x_pe + (double)dy_pe));
#define PWG(BM_, BN_, NP_)                                                                                          \
    do {                                                                                                            \
        constexpr int NST = 3;                                                                                      \
        const size_t lds = (size_t)NST * NP_ * 32 * (BM_ + BN_) * 2;                                                \
        auto kern = pwgrad_kernel<BM_, BN_, NP_, NST>;                                                              \
        static bool attr_set = false;                                                                               \
        if (!attr_set) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; } \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const unsigned short*)x_planes, x_pe,                   \
                           (const unsigned short*)dy_planes, dy_pe, out, g, ci_tiles, co_tiles, pps, slab, dw, arr);         \
    } while (0)
        if (nplanes == 3) {
            if (bm == 128 && bn == 128) PWG(128, 128, 3);
            else if (bm == 128) PWG(128, 64, 3);
            else if (bn == 128) PWG(64, 128, 3);
            else PWG(64, 64, 3);
        } else {
            if (bm == 128 && bn == 128) PWG(128, 128, 1);
            else if (bm == 128) PWG(128, 64, 1);
            else if (bn == 128) PWG(64, 128, 1);
            else PWG(64, 64, 1);
        }
#undef PWG
    }
    int rc = sc_check_launch("conv2d_wgrad_planes");
    if (rc) return rc;
    if (splits > 1 && !arr) {
        ScProfScope prof2("slab_reduce", st, 0, 4.0 * (double)slab * (splits + 1));
        sc_launch_slab_reduce((const float*)ws, dw, slab, splits, slab, st);
        rc = sc_check_launch("conv2d_wgrad_planes_reduce");
    }
    return rc;
}
