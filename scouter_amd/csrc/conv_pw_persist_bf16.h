// Persistent pointwise INPUT GRADIENT with the fused BatchNorm-backward epilogue for bf16-STORED tensors (round 5) -- the
// mixed-precision mode's counterpart of pwp_fused_kernel (conv_pw_persist.h).  Call sites: conv1 of every ResNeSt bottleneck
// (/root/reference/timm/models/resnest.py:111-118; its input gradient is the block-input gradient that the previous block's
// bn3 / downsample BatchNorm backward consumes) under BASELINE configs[4] ("bf16").
//
// Why: igemm_bf16_kernel<128, 64, .., dgrad, typed> runs these launches at 3.0 TB/s with the matrix pipe 4.6 % busy
// (profiles/r05_pmc_mfma_util_bench_config5.txt) -- per output element they move dY / N-th, the shortcut gradient, one or two
// BatchNorm inputs, mask bits and the result: 6-8 bytes against 2 K flops on a 2.5 PFLOP/s pipe, i.e. an HBM stream with a
// small GEMM attached, and a workgroup that waits for its operands, then for its epilogue operands, then for its stores
// leaves the memory system a third idle.  Skeleton of pwp_fused_kernel:
//   * W^T of the workgroup's 128 columns ([128 n][K] bf16, RNE of the fp32 master weights -- what igemm_bf16_kernel rounds
//     them to) is converted ONCE into LDS; dY streams through a four-slot ring of [64 rows][64 k] bf16 stages (8 KB) filled
//     by LDS-DMA three stages ahead, 16-byte chunk c of row r at chunk c ^ (r & 7);
//   * v_mfma_f32_16x16x32_bf16 with the COLUMNS PERMUTED: MFMA column l of 16-column block j is actual column 8 l + j, so a
//     lane's eight accumulators of a row are EIGHT CONSECUTIVE output columns -- one 16-byte access per row and operand in
//     the epilogue (16 lanes = 256 contiguous bytes of a row) with no transposition and no LDS staging; the GEMM does not
//     care about the column order (the B fragment of block j simply reads W rows 8 l + j);
//   * wave = 16 rows x 128 columns (4 rows x 8 columns per lane: 32 accumulator registers), so the epilogue's operands of a
//     tile -- requested when its last K stage starts -- fit the register file next to TWO resident workgroups per CU for
//     K <= 128 and one BatchNorm (a second one's operands would spill: one workgroup per CU then): one streams while the
//     other multiplies;
//   * the per-channel sums (sum g, sum g * xhat) are lane-local fp64 accumulators over ALL tiles of the workgroup, combined
//     across the four row quads and the four waves once at the end: one partial row per workgroup.
// Arithmetic per element as igemm_epilogue_typed (same products, fp32 accumulation over K in the MFMA's own order, the
// shortcut gradient added in fp32, ReLU bit, RNE to bf16); sums agree with the other tiles to fp64 rounding of another
// grouping, dx to one bf16 ulp where the accumulation order differs by an fp32 ulp.
#pragma once
#include "conv_common.h"

#include <type_traits>

#ifndef PWB_RESIDENT
#define PWB_RESIDENT 2                                            // workgroups per CU for K <= 128, one BatchNorm
#endif
typedef __bf16 pwb_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned pwb_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void pwb_dma16(__amdgpu_buffer_rsrc_t rs, char* lds, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ f32x4 pwb_mfma(pwb_bf16x8 a, pwb_bf16x8 b, f32x4 c) {
    // A[i = l & 15][k = 8 (l >> 4) + 0..7], B[k = 8 (l >> 4) + 0..7][j = l & 15]; D[row 4 (l >> 4) + e][col l & 15]
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float pwb_f32(unsigned w, int hi) {        // element 2 i + hi of a packed bf16 pair
    return __builtin_bit_cast(float, hi ? (w & 0xffff0000u) : (w << 16));
}

// KS = K / 64 (K = Cout of the layer: 64 ... 512); grid = wg_per_col * (N / 128) workgroups, workgroup (p, cg) owns the 64-row
// tiles p, p + wg_per_col, ... of column group cg.  part1 / part2: [wg_per_col][N][2] fp64.
template <int KS, bool TWO>
__global__ __launch_bounds__(256, (KS <= 2 && !TWO ? PWB_RESIDENT : 1)) void pwb_fused_kernel(
    const void* __restrict__ src, const float* __restrict__ wgt, const void* __restrict__ addend, void* __restrict__ dst, long M,
    int N, int mtiles, int wg_per_col, BnBwdFuse fz, long mask_words) {
    constexpr int K = 64 * KS, BN = 128, RB = 2 * K;             // bytes per W^T row
    constexpr int WBYTES = BN * RB, SLOT = 64 * 128;
    constexpr int CHM = (K / 8 < 16 ? K / 8 : 16) - 1;           // W^T swizzle: chunk c of row n at c ^ ((n >> 3) & CHM)
    extern __shared__ __attribute__((aligned(1024))) char pwb_lds[];
    char* Wl = pwb_lds;
    char* ring = pwb_lds + WBYTES;

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int colgroups = N / BN;
    // workgroups of one M range (same p) share an XCD (dispatch is round-robin over the eight XCDs): dY is read from HBM once
    const int L = blockIdx.x;
    const int cg = (L >> 3) % colgroups, p = (L & 7) + 8 * ((L >> 3) / colgroups);
    const int n0 = cg * BN;
    const int n_my = p < mtiles ? (mtiles - p + wg_per_col - 1) / wg_per_col : 0;
    const int nstage4 = (n_my * KS + 3) / 4;

    const unsigned obytes = (unsigned)(M * N * 2);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)(M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)(addend ? addend : dst), 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc((void*)fz.x1, 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc((void*)(TWO ? fz.x2 : fz.x1), 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(fz.mask ? (const void*)fz.mask : (const void*)dst), 0, (unsigned)(fz.mask ? mask_words * 8 : 0), 0x00020000);
    const bool has_add = addend != nullptr, has_mask = fz.mask != nullptr;

    // ---- dY ring: two DMA instructions per thread and stage (8 pieces of 8 rows x 128 bytes)
    unsigned a_voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 8 * (wave + 4 * j) + (lane >> 3), c = (lane & 7) ^ (r & 7);
        a_voff[j] = (unsigned)((r * K + c * 8) * 2);
    }
    auto issue = [&](int G, int slot) __attribute__((always_inline)) {
        const int ti = G / KS, kc = G % KS;
        const long m0 = (long)__builtin_amdgcn_readfirstlane(p + ti * wg_per_col) * 64;
        const long off = (m0 * K + kc * 64) * 2;
        const int soff = off > 0x7fffffffL ? 0x7fffffff : (int)off;            // (beyond the tensor: reads zeros)
#pragma unroll
        for (int j = 0; j < 2; ++j) pwb_dma16(rs_a, ring + slot * SLOT + (wave + 4 * j) * 1024, a_voff[j], soff);
    };
    issue(0, 0); issue(1, 1); issue(2, 2); issue(3, 3);
    // ---- W^T rows n0 .. n0 + 127 (n = ci, k = co contiguous: the HWIO layout of a 1x1 weight itself) -> bf16 in LDS
    for (int e = tid; e < BN * (K / 8); e += 256) {
        const int n = e / (K / 8), c = e % (K / 8);
        const float* wp = wgt + (long)(n0 + n) * K + 8 * c;
        const f32x4 lo = *(const f32x4*)wp, hi = *(const f32x4*)(wp + 4);
        pwb_bf16x8 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = (__bf16)lo[i]; v[4 + i] = (__bf16)hi[i]; }
        *(pwb_bf16x8*)(Wl + n * RB + ((c ^ ((n >> 3) & CHM)) << 4)) = v;
    }

    // ---- fragment geometry: A row = 16 wave + l15; B row (block j) = 8 l15 + j
    const int arow = 16 * wave + l15;
    const char* a_base = ring + arow * 128;
    const int a_sw = arow & 7;
    const char* b_base = Wl + 8 * l15 * RB;
    const int b_t = q ^ (l15 & CHM);                            // (chunk base | q) ^ swizzle = chunk base ^ b_t
    // ---- epilogue geometry: rows 16 wave + 4 q + e, columns col0 .. col0 + 7
    const int col0 = n0 + 8 * l15;
    const int row_in = 16 * wave + 4 * q;
    const unsigned o_voff = (unsigned)(((long)row_in * N + col0) * 2);
    float mu1[8], rs1[8], mu2[TWO ? 8 : 1], rs2[TWO ? 8 : 1];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mu1[k] = fz.sv1[col0 + k];
        rs1[k] = fz.sv1[N + col0 + k];
        if (TWO) { mu2[TWO ? k : 0] = fz.sv2[col0 + k]; rs2[TWO ? k : 0] = fz.sv2[N + col0 + k]; }
    }

    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    double sg[8], sx1[8], sx2[TWO ? 8 : 1];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sg[k] = 0.0; sx1[k] = 0.0; if (TWO) sx2[TWO ? k : 0] = 0.0; }
    // epilogue operands of the tile whose last K stage is running
    pwb_u32x4 pa[4], px1[4], px2[TWO ? 4 : 1], pm[4][2];

#define SB() __builtin_amdgcn_sched_barrier(0)
    auto row_soff = [&](long m0, int e) __attribute__((always_inline)) -> int {
        const long off = (m0 + e) * (long)N * 2;
        return off > 0x7fffffffL ? 0x7fffffff : (int)off;
    };
    auto prefetch = [&](int ti) __attribute__((always_inline)) {
        const long m0 = (long)__builtin_amdgcn_readfirstlane(p + ti * wg_per_col) * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int soff = row_soff(m0, e);
            if (has_add) pa[e] = __builtin_amdgcn_raw_buffer_load_b128(rs_d, o_voff, soff, 0);
            px1[e] = __builtin_amdgcn_raw_buffer_load_b128(rs_x1, o_voff, soff, 0);
            if (TWO) px2[TWO ? e : 0] = __builtin_amdgcn_raw_buffer_load_b128(rs_x2, o_voff, soff, 0);
            if (has_mask) {
                // ReLU bits: float4 group i4 = (m N + c) / 4 owns bit (i4 & 63) of the words mask[(i4 >> 6) * 4 + (c & 3)]; the
                // lane's eight columns are the groups i4, i4 + 1 (i4 even: same four words)
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
            const int soff = row_soff(m0, e);
            const int bit0 = (int)((((m0 + row_in + e) * (long)N + col0) >> 2) & 63);
            pwb_u32x4 out;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float v = acc[k][e];
                if (has_add) v += pwb_f32(pa[e][k >> 1], k & 1);
                if (has_mask) {
                    const unsigned long long w = ((unsigned long long)pm[e][(k & 3) >> 1][2 * (k & 1) + 1] << 32) |
                                                 pm[e][(k & 3) >> 1][2 * (k & 1)];
                    v = ((w >> (bit0 + (k >> 2))) & 1ull) ? v : 0.f;
                }
                const double d = (double)v;
                sg[k] += d;
                sx1[k] += d * (double)((pwb_f32(px1[e][k >> 1], k & 1) - mu1[k]) * rs1[k]);
                if (TWO) sx2[TWO ? k : 0] += d * (double)((pwb_f32(px2[TWO ? e : 0][k >> 1], k & 1) - mu2[TWO ? k : 0]) * rs2[TWO ? k : 0]);
                const unsigned short hb = __builtin_bit_cast(unsigned short, (__bf16)v);
                if (k & 1) out[k >> 1] |= (unsigned)hb << 16; else out[k >> 1] = hb;
                acc[k][e] = 0.f;
            }
            // The row offset goes into the PER-LANE offset, the scalar offset stays an immediate 0.  With a 16-byte store whose
            // scalar offset is a REGISTER the compiler assumes that no wait state is needed before the data registers are
            // overwritten (GCNHazardRecognizer: ">64-bit MUBUF store data hazard only without an soffset register") -- on gfx950
            // that is not enough once two waves share the SIMD: lanes 12-15 of every 16 stored the NEXT row's address arithmetic
            // (v_mad_u64_u32 into the first two data registers right behind the store) with two workgroups per CU, and never
            // with one (tools_dev/pwb_debug.py found it).  Rows beyond M: the offset is beyond num_records, the store is dropped.
            __builtin_amdgcn_raw_buffer_store_b128(out, rs_o, o_voff + (unsigned)((m0 + e) * (long)N * 2), 0, 0);
        }
    };
    auto stage_body = [&](int G, int slot, int kc) __attribute__((always_inline)) {
        const bool last = kc == KS - 1 && G / KS < n_my;
        const char* As = a_base + slot * SLOT;
        pwb_bf16x8 fa[2], fb[2][8];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            fa[s] = *(const pwb_bf16x8*)(As + (((4 * s + q) ^ a_sw) << 4));
            const char* bp = b_base + ((((kc * 8 + 4 * s) ^ b_t)) << 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) fb[s][j] = *(const pwb_bf16x8*)(bp + j * RB);
        }
        if (last) prefetch(G / KS);                              // lands under this stage's MFMAs (and the other workgroup's work)
        SB();
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = pwb_mfma(fa[s], fb[s][j], acc[j]);
        SB();
        // the next stage's DMA has landed (two younger stages = 4 instructions may stay in flight; the epilogue's operands,
        // younger still, are waited for by the compiler's own counting where they are used)
        if (!last) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (last) epilogue(G / KS);
        __builtin_amdgcn_s_barrier();                            // slot is free, slot + 1 is visible to every wave
        SB();
        issue(G + 4, slot);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // stage 0 .. 3 and the weight rows have landed
    __syncthreads();
    for (int base = 0; base < nstage4; ++base) {
#pragma unroll
        for (int S = 0; S < 4; ++S) stage_body(4 * base + S, S, KS <= 4 ? S % KS : (4 * base + S) % KS);
    }
#undef SB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- one partial row per workgroup: the four row quads of a wave by shuffles, the four waves through LDS
    __syncthreads();
    double* Ps = (double*)pwb_lds;                               // [4 waves][128 columns][3]
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double tg = sg[k], t1 = sx1[k], t2 = TWO ? sx2[TWO ? k : 0] : 0.0;
        tg += __shfl_xor(tg, 16, 64); t1 += __shfl_xor(t1, 16, 64); t2 += __shfl_xor(t2, 16, 64);
        tg += __shfl_xor(tg, 32, 64); t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
        if (q == 0) {
            double* o = Ps + ((wave * BN) + 8 * l15 + k) * 3;
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
// The same skeleton for the FORWARD of the pointwise layers whose input is stored as bf16 (tile 4 of
// scouter_conv2d_fwd_bf16_io): y = x W with the fused BatchNorm statistics (fp64 column sums of the fp32 accumulators), y
// stored as bf16 (Y_BF16: conv3 / downsample outputs) or fp32 (conv1 outputs).  Call sites: conv1 / downsample of the ResNeSt
// bottlenecks (/root/reference/timm/models/resnest.py:111-118, resnet.py:292-306) under BASELINE configs[4].  igemm_bf16_kernel
// runs these output streams at 2.5 TB/s of algorithmic bytes.  W^T arrives as bf16 [N][K] (scouter_conv2d_weight_bf16t) and is
// copied once into LDS; K = Cin <= 512.  part: [wg_per_col][N][2] fp64.
template <int KS, bool Y_BF16, bool STATS>
__global__ __launch_bounds__(256, (KS == 1 ? 3 : (KS == 2 ? 2 : 1))) void pwb_fwd_kernel(
    const void* __restrict__ src, const void* __restrict__ wt, void* __restrict__ dst, double* __restrict__ part, long M, int N,
    int mtiles, int wg_per_col) {
    constexpr int K = 64 * KS, BN = 128, RB = 2 * K;
    constexpr int WBYTES = BN * RB, SLOT = 64 * 128;
    constexpr int CHM = (K / 8 < 16 ? K / 8 : 16) - 1;
    constexpr int OB = Y_BF16 ? 2 : 4;                          // bytes per output element
    extern __shared__ __attribute__((aligned(1024))) char pwb_lds[];
    char* Wl = pwb_lds;
    char* ring = pwb_lds + WBYTES;

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int colgroups = N / BN;
    const int L = blockIdx.x;
    const int cg = (L >> 3) % colgroups, p = (L & 7) + 8 * ((L >> 3) / colgroups);
    const int n0 = cg * BN;
    const int n_my = p < mtiles ? (mtiles - p + wg_per_col - 1) / wg_per_col : 0;
    const int nstage4 = (n_my * KS + 3) / 4;

    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)(M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, (unsigned)(M * N * OB), 0x00020000);
    unsigned a_voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 8 * (wave + 4 * j) + (lane >> 3), c = (lane & 7) ^ (r & 7);
        a_voff[j] = (unsigned)((r * K + c * 8) * 2);
    }
    auto issue = [&](int G, int slot) __attribute__((always_inline)) {
        const int ti = G / KS, kc = G % KS;
        const long m0 = (long)__builtin_amdgcn_readfirstlane(p + ti * wg_per_col) * 64;
        const long off = (m0 * K + kc * 64) * 2;
        const int soff = off > 0x7fffffffL ? 0x7fffffff : (int)off;
#pragma unroll
        for (int j = 0; j < 2; ++j) pwb_dma16(rs_a, ring + slot * SLOT + (wave + 4 * j) * 1024, a_voff[j], soff);
    };
    issue(0, 0); issue(1, 1); issue(2, 2); issue(3, 3);
    for (int e = tid; e < BN * (K / 8); e += 256) {              // W^T rows n0 .. n0 + 127, 16-byte chunks
        const int n = e / (K / 8), c = e % (K / 8);
        const pwb_u32x4 v = *(const pwb_u32x4*)((const __bf16*)wt + (long)(n0 + n) * K + 8 * c);
        *(pwb_u32x4*)(Wl + n * RB + ((c ^ ((n >> 3) & CHM)) << 4)) = v;
    }

    const int arow = 16 * wave + l15;
    const char* a_base = ring + arow * 128;
    const int a_sw = arow & 7;
    const char* b_base = Wl + 8 * l15 * RB;
    const int b_t = q ^ (l15 & CHM);
    const int col0 = n0 + 8 * l15;
    const int row_in = 16 * wave + 4 * q;
    const unsigned o_voff = (unsigned)(((long)row_in * N + col0) * OB);

    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    double s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] = 0.0; s2[k] = 0.0; }

#define SB() __builtin_amdgcn_sched_barrier(0)
    auto epilogue = [&](int ti) __attribute__((always_inline)) {
        const long m0 = (long)__builtin_amdgcn_readfirstlane(p + ti * wg_per_col) * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // (row offset in the per-lane offset, scalar offset an immediate 0: see the input-gradient kernel above)
            const unsigned off = o_voff + (unsigned)((m0 + e) * (long)N * OB);
            pwb_u32x4 ob;
            f32x4 of[2];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float v = acc[k][e];
                if (STATS) {                                     // (rows beyond M: their A rows read zeros, v = 0)
                    const double d = (double)v;
                    s1[k] += d;
                    s2[k] = __builtin_fma(d, d, s2[k]);
                }
                if (Y_BF16) {
                    const unsigned short hb = __builtin_bit_cast(unsigned short, (__bf16)v);
                    if (k & 1) ob[k >> 1] |= (unsigned)hb << 16; else ob[k >> 1] = hb;
                } else {
                    of[k >> 2][k & 3] = v;
                }
                acc[k][e] = 0.f;
            }
            if (Y_BF16) {
                __builtin_amdgcn_raw_buffer_store_b128(ob, rs_o, off, 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pwb_u32x4, of[0]), rs_o, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pwb_u32x4, of[1]), rs_o, off + 16u, 0, 0);
            }
        }
    };
    auto stage_body = [&](int G, int slot, int kc) __attribute__((always_inline)) {
        const bool last = kc == KS - 1 && G / KS < n_my;
        const char* As = a_base + slot * SLOT;
        pwb_bf16x8 fa[2], fb[2][8];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            fa[s] = *(const pwb_bf16x8*)(As + (((4 * s + q) ^ a_sw) << 4));
            const char* bp = b_base + ((((kc * 8 + 4 * s) ^ b_t)) << 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) fb[s][j] = *(const pwb_bf16x8*)(bp + j * RB);
        }
        SB();
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = pwb_mfma(fa[s], fb[s][j], acc[j]);
        SB();
        // the next stage's DMA has landed: at most the two younger stages (4 instructions) stay in flight -- stores of an
        // epilogue in between are younger than the stage waited for and only make the wait stricter
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (last) epilogue(G / KS);
        __builtin_amdgcn_s_barrier();
        SB();
        issue(G + 4, slot);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int base = 0; base < nstage4; ++base) {
#pragma unroll
        for (int S = 0; S < 4; ++S) stage_body(4 * base + S, S, KS <= 4 ? S % KS : (4 * base + S) % KS);
    }
#undef SB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (STATS) {
        __syncthreads();
        double* Ps = (double*)pwb_lds;                           // [4 waves][128 columns][2]
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double t1 = s1[k], t2 = s2[k];
            t1 += __shfl_xor(t1, 16, 64); t2 += __shfl_xor(t2, 16, 64);
            t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
            if (q == 0) { double* o = Ps + ((wave * BN) + 8 * l15 + k) * 2; o[0] = t1; o[1] = t2; }
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

// ---------------------------------------------------------------------------------------------------------------------------
// The PLAIN input gradient of the pointwise layers whose output gradient is stored as bf16 (tile 4 of
// scouter_conv2d_dgrad_bnbwd_bf16_io without the fused epilogue): dx = dY W^T (+ addend), dx fp32.  Call sites: conv3 and the
// downsample convolution of the ResNeSt bottlenecks (/root/reference/timm/models/resnest.py:128-143, resnet.py:292-306) under
// BASELINE configs[4] -- dY is 4 g wide, dx g wide: input streams (64 <- 256 at 56 x 56, batch 256: 411 MB read, 205 MB
// written) that igemm_bf16_kernel<64, 64> runs at 1.9 TB/s of algorithmic bytes.  64 output columns per workgroup (wave =
// 16 rows x 64 columns, a lane's four accumulators of a row are four consecutive fp32 columns: 16-byte stores), W^T [64][K]
// converted once into LDS, K = Cout up to 1024.
template <int KS, bool HAS_ADD>
__global__ __launch_bounds__(256, (KS <= 1 ? 3 : (KS <= 4 ? 2 : 1))) void pwb_dgrad_kernel(
    const void* __restrict__ src, const float* __restrict__ wgt, const void* __restrict__ addend, float* __restrict__ dst, long M,
    int N, int mtiles, int wg_per_col, int add_bf16) {
    constexpr int K = 64 * KS, BN = 64, RB = 2 * K;
    constexpr int WBYTES = BN * RB, SLOT = 64 * 128;
    constexpr int CHM = (K / 8 < 16 ? K / 8 : 16) - 1;
    extern __shared__ __attribute__((aligned(1024))) char pwb_lds[];
    char* Wl = pwb_lds;
    char* ring = pwb_lds + WBYTES;

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int colgroups = N / BN;
    const int L = blockIdx.x;
    const int cg = (L >> 3) % colgroups, p = (L & 7) + 8 * ((L >> 3) / colgroups);
    const int n0 = cg * BN;
    const int n_my = p < mtiles ? (mtiles - p + wg_per_col - 1) / wg_per_col : 0;
    const int nstage4 = (n_my * KS + 3) / 4;

    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)(M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, (unsigned)(M * N * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)(HAS_ADD ? addend : (const void*)dst), 0,
                                                                          (unsigned)(M * N * (add_bf16 ? 2 : 4)), 0x00020000);
    unsigned a_voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 8 * (wave + 4 * j) + (lane >> 3), c = (lane & 7) ^ (r & 7);
        a_voff[j] = (unsigned)((r * K + c * 8) * 2);
    }
    auto issue = [&](int G, int slot) __attribute__((always_inline)) {
        const int ti = G / KS, kc = G % KS;
        const long m0 = (long)__builtin_amdgcn_readfirstlane(p + ti * wg_per_col) * 64;
        const long off = (m0 * K + kc * 64) * 2;
        const int soff = off > 0x7fffffffL ? 0x7fffffff : (int)off;
#pragma unroll
        for (int j = 0; j < 2; ++j) pwb_dma16(rs_a, ring + slot * SLOT + (wave + 4 * j) * 1024, a_voff[j], soff);
    };
    issue(0, 0); issue(1, 1); issue(2, 2); issue(3, 3);
    for (int e = tid; e < BN * (K / 8); e += 256) {              // W^T rows n0 .. n0 + 63 (n = ci, k = co contiguous) -> bf16
        const int n = e / (K / 8), c = e % (K / 8);
        const float* wp = wgt + (long)(n0 + n) * K + 8 * c;
        const f32x4 lo = *(const f32x4*)wp, hi = *(const f32x4*)(wp + 4);
        pwb_bf16x8 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = (__bf16)lo[i]; v[4 + i] = (__bf16)hi[i]; }
        *(pwb_bf16x8*)(Wl + n * RB + ((c ^ ((n >> 2) & CHM)) << 4)) = v;
    }

    const int arow = 16 * wave + l15;
    const char* a_base = ring + arow * 128;
    const int a_sw = arow & 7;
    const char* b_base = Wl + 4 * l15 * RB;
    const int b_t = q ^ (l15 & CHM);
    const int col0 = n0 + 4 * l15;
    const int row_in = 16 * wave + 4 * q;
    const long elem0 = (long)row_in * N + col0;

    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 pa[4];

#define SB() __builtin_amdgcn_sched_barrier(0)
    auto prefetch = [&](int ti) __attribute__((always_inline)) {
        const long m0 = (long)__builtin_amdgcn_readfirstlane(p + ti * wg_per_col) * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long el = elem0 + (m0 + e) * (long)N;
            if (add_bf16) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 w = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_d, (unsigned)(el * 2), 0, 0));
                pa[e] = f32x4{pwb_f32(w[0], 0), pwb_f32(w[0], 1), pwb_f32(w[1], 0), pwb_f32(w[1], 1)};
            } else {
                pa[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, (unsigned)(el * 4), 0, 0));
            }
        }
    };
    auto epilogue = [&](int ti) __attribute__((always_inline)) {
        const long m0 = (long)__builtin_amdgcn_readfirstlane(p + ti * wg_per_col) * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f32x4 out;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                out[k] = acc[k][e] + (HAS_ADD ? pa[e][k] : 0.f);
                acc[k][e] = 0.f;
            }
            // (row offset in the per-lane offset, scalar offset an immediate 0: see the fused kernel above)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pwb_u32x4, out), rs_o,
                                                   (unsigned)((elem0 + (m0 + e) * (long)N) * 4), 0, 0);
        }
    };
    auto stage_body = [&](int G, int slot, int kc) __attribute__((always_inline)) {
        const bool last = kc == KS - 1 && G / KS < n_my;
        const char* As = a_base + slot * SLOT;
        pwb_bf16x8 fa[2], fb[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            fa[s] = *(const pwb_bf16x8*)(As + (((4 * s + q) ^ a_sw) << 4));
            const char* bp = b_base + ((((kc * 8 + 4 * s) ^ b_t)) << 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[s][j] = *(const pwb_bf16x8*)(bp + j * RB);
        }
        if (HAS_ADD && last) prefetch(G / KS);
        SB();
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = pwb_mfma(fa[s], fb[s][j], acc[j]);
        SB();
        // the next stage's DMA has landed (at most the two younger stages = 4 instructions stay in flight; with an addend its
        // loads are younger still and waited for by the compiler where they are used)
        if (!(HAS_ADD && last)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (last) epilogue(G / KS);
        __builtin_amdgcn_s_barrier();
        SB();
        issue(G + 4, slot);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int base = 0; base < nstage4; ++base) {
#pragma unroll
        for (int S = 0; S < 4; ++S) stage_body(4 * base + S, S, KS <= 4 ? S % KS : (4 * base + S) % KS);
    }
#undef SB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
