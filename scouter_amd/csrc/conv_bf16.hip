// bf16-input / fp32-accumulate implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the
// fp32-MFMA rate) -- the mixed-precision mode of BASELINE configs[4] ("bf16").  By default tensors are fp32 in HBM (activations,
// gradients, master weights): operands are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) on their way from the global
// loads into LDS, so every other kernel of the step (BatchNorm, xSlot, optimizer) is shared with the fp32 path and
// only the matrix inputs lose precision -- what torch.autocast(bfloat16) does to nn.Conv2d.  The typed (`*_io`) entry
// points additionally read / write bf16-STORED tensors (A_BF16 loaders, the TYPED epilogue): same products, half the bytes.
//
//   forward : Y = X (*) W      A = X rows (pixel, k = (tap, ci) contiguous in ci), B = W^T as bf16 [tap][co][ci]
//                              (scouter_conv2d_weight_bf16t, one small launch per layer and step)
//   dgrad   : dX = dY (*) W^T  A = dY rows (k = co), B = W rows (n = ci, k = co contiguous: the HWIO layout itself)
//   wgrad   : dW = X^T dY      both operands have the contraction index (pixel) as the slow axis: they are written
//                              TRANSPOSED into LDS ([channel][pixel] bf16) so that fragments are 16-byte reads
//
// With bf16 MFMA the matrix pipe is no longer the bound: a 128x128x64 block tile needs 64 KB of fp32 operands for
// 2.1 MFLOP (32 FLOP per L2 byte), so these kernels run at the operand-delivery rate.  Same scalar addressing
// (buffer descriptors), block epilogue and BatchNorm-statistics fusion as conv_igemm.hip (conv_common.h).
#include "conv_common.h"
#include "conv_pw_persist_bf16.h"
#include "conv_halo_dgrad_bf16.h"

#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    // A[i = l&31][k = 8*(l>>5) + 0..7], B[k = 8*(l>>5) + 0..7][j = l&31]; D as the fp32 32x32 form
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x4 to_bf16x4(f32x4 v) {
    bf16x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (__bf16)v[e];
    return r;
}

// ----------------------------------------------------------------------------------------------------------------
// forward / dgrad.  Block = 4 waves, tile BM x BN x KT (KT = 64, or 32 for 32-channel groups), 2 LDS stages,
// LDS rows of KT+8 bf16 (16-byte aligned, conflict-free 16-byte fragment reads).
// ----------------------------------------------------------------------------------------------------------------
// A_BF16: the A operand (the forward's input activation, the input gradient's dy) is STORED as bf16 (round 4, activation
// storage): 16-byte loads carry eight k-values straight into LDS, no conversion.  dst_bf16 (run time): the output tensor is stored as bf16 -- the
// fused BatchNorm statistics still come from the fp32 accumulators.
template <int BM, int BN, int WM, int WN, int KT_, bool DGRAD, bool A_BF16 = false>
__global__ __launch_bounds__(256, 2) void igemm_bf16_kernel(const void* __restrict__ src, const void* __restrict__ wgt,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ addend, float* __restrict__ dst,
                                                            double* __restrict__ bn_part, ConvGeom g, int relu,
                                                            int mtiles, int ntiles, BnBwdFuse fz, int dst_bf16,
                                                            int one_stage) {
    constexpr int MT = WM / 32, NT = WN / 32, WAVES_N = BN / WN;
    constexpr int LDH = KT_ + 8;                          // LDS row stride in bf16
    constexpr int A_H = BM * LDH, B_H = BN * LDH, STAGE_H = A_H + B_H;
    constexpr int TPR = KT_ / 4;                          // threads per operand row when loading fp32 float4 along k
    constexpr int RPI = 256 / TPR;                        // rows per load instruction
    constexpr int AEL = A_BF16 ? 8 : 4, AES = A_BF16 ? 2 : 4;   // A: elements per 16-byte load, bytes per element
    constexpr int ATPR = KT_ / AEL, ARPI = 256 / ATPR;    // A: threads per row, rows per load instruction
    constexpr int AI = (BM + ARPI - 1) / ARPI, BI = BN / RPI;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
    extern __shared__ __attribute__((aligned(16))) __bf16 ldsh[];      // max(2 stages, epilogue staging) bytes

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int nblk = mtiles * ntiles * g.groups;
    const int bid = xcd_remap(blockIdx.x, nblk);
    const int grp = bid % g.groups;
    const int nt_id = (bid / g.groups) % ntiles;
    const int mt_id = bid / (g.groups * ntiles);
    const long m0 = (long)mt_id * BM;
    const int n0 = nt_id * BN;
    const int cpt = g.Cg / KT_;                // K chunks per filter tap
    const int KT = g.R * g.S * cpt;

    // ---- A rows: block-relative buffer offsets + separable tap masks (see conv_igemm.hip)
    constexpr unsigned OOB = 0x80000000u;
    const int hw = g.Ho * g.Wo;
    const int blk_b = (int)(((double)(unsigned)m0 + 0.5) * g.inv_hw);
    const int blk_rem = (int)((unsigned)m0 - (unsigned)blk_b * (unsigned)hw);
    const int blk_y = fast_div(blk_rem, g.inv_wo), blk_x = blk_rem - blk_y * g.Wo;
    const int rows_valid = (int)(g.M - m0 < BM ? g.M - m0 : BM);
    const int a_col = (tid % ATPR) * AEL;
    unsigned a_mask[AI], a_voff[AI], a_veff[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int rowoff = tid / ATPR + ARPI * i;
        const bool okm = rowoff < rows_valid && rowoff < BM;
        const int tx = blk_x + rowoff, qx = fast_div(tx, g.inv_wo), x = tx - qx * g.Wo;
        const int ty = blk_y + qx, qy = fast_div(ty, g.inv_ho), y = ty - qy * g.Ho;
        const int ay = DGRAD ? y + g.pad : y * g.stride - g.pad, ax = DGRAD ? x + g.pad : x * g.stride - g.pad;
        unsigned colbits = 0, mask = 0;
        for (int q = 0; q < g.S; ++q) colbits |= ((unsigned)(DGRAD ? ax - q : ax + q) < (unsigned)g.W ? 1u : 0u) << q;
        for (int r = 0; r < g.R; ++r) mask |= ((unsigned)(DGRAD ? ay - r : ay + r) < (unsigned)g.H ? colbits : 0u) << (r * g.S);
        a_mask[i] = okm ? mask : 0u;
        const int rel = okm ? qy * g.H * g.W : 0;
        const int e = DGRAD ? (rel + ay * g.W + ax) * g.C : (rel + (ay + g.pad) * g.W + (ax + g.pad)) * g.C;
        a_voff[i] = (unsigned)(e + grp * g.Cg + a_col) * (unsigned)AES;
        a_veff[i] = OOB;
    }
    const long img_elems = (long)g.H * g.W * g.C;
    const long shift = DGRAD ? ((long)(g.R - 1) * g.W + (g.S - 1)) * g.C : ((long)g.pad * g.W + g.pad) * g.C;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)src + ((long)blk_b * img_elems - shift) * AES), 0, 0x7fffffff, 0x00020000);
    // ---- B rows (n = output column of the GEMM, k contiguous)
    //   fwd  : bf16 W^T [tap][co][Cg]      : 8 bf16 (16 B) per load, KT/8 threads per row
    //   dgrad: fp32 W   [tap][ci][Cout]    : rows n = ci, k = co within the group, float4 per load
    constexpr int BTPR = DGRAD ? TPR : KT_ / 8, BRPI = 256 / BTPR, BBI = (BN + BRPI - 1) / BRPI;
    const int b_col = (tid % BTPR) * (DGRAD ? 4 : 8);
    const char* wbase = DGRAD ? (const char*)((const float*)wgt + (long)grp * g.Cg)
                              : (const char*)((const __bf16*)wgt + (long)grp * g.Ng * g.Cg);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, 0x7fffffff, 0x00020000);
    unsigned b_voff[BBI];
#pragma unroll
    for (int i = 0; i < BBI; ++i) {
        const int n = n0 + tid / BTPR + BRPI * i;
        b_voff[i] = DGRAD ? (unsigned)(((long)n * g.wrow + b_col) * 4) : (unsigned)(((long)n * g.Cg + b_col) * 2);
        if (tid / BTPR + BRPI * i >= BN) b_voff[i] = OOB;          // (tiles narrower than one load instruction)
    }
    const long wtap_bytes = DGRAD ? (long)g.wtap * 4 : (long)g.N * g.Cg * 2;     // fwd: [tap][Cout_total][Cg] bf16

    f32x4 ra[AI];
    f32x4 rbf[DGRAD ? BBI : 1];
    bf16x8 rbh[DGRAD ? 1 : BBI];
    auto load_a = [&](int kt) {
        const int tap = kt / cpt, c0 = (kt - tap * cpt) * KT_;
        const int r = tap / g.S, q = tap - r * g.S;
        const long toff = (DGRAD ? -((long)r * g.W + q) : ((long)r * g.W + q)) * g.C + c0;
        if (c0 == 0) {
#pragma unroll
            for (int i = 0; i < AI; ++i) a_veff[i] = ((a_mask[i] >> tap) & 1u) ? a_voff[i] : OOB;
        }
        const int soff = (int)((DGRAD ? shift + toff : toff) * AES);
#pragma unroll
        for (int i = 0; i < AI; ++i)
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_veff[i], soff, 0));
    };
    auto load_b = [&](int kt) {
        const int tap = kt / cpt, c0 = (kt - tap * cpt) * KT_;
        const int soff = (int)(tap * wtap_bytes + (long)c0 * (DGRAD ? 4 : 2));
#pragma unroll
        for (int i = 0; i < BBI; ++i) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_b, b_voff[i], soff, 0);
            if (DGRAD) rbf[i] = __builtin_bit_cast(f32x4, v);
            else rbh[i] = __builtin_bit_cast(bf16x8, v);
        }
    };
    auto store_a = [&](int buf) {
        __bf16* As = ldsh + buf * STAGE_H;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            if (BM % ARPI != 0 && tid / ATPR + ARPI * i >= BM) continue;      // (tiles shorter than one load instruction)
            if constexpr (A_BF16) *(bf16x8*)(As + (tid / ATPR + ARPI * i) * LDH + a_col) = __builtin_bit_cast(bf16x8, ra[i]);
            else *(bf16x4*)(As + (tid / ATPR + ARPI * i) * LDH + a_col) = to_bf16x4(ra[i]);
        }
    };
    auto store_b = [&](int buf) {
        __bf16* Bs = ldsh + buf * STAGE_H + A_H;
#pragma unroll
        for (int i = 0; i < BBI; ++i) {
            if (BN % BRPI != 0 && tid / BTPR + BRPI * i >= BN) continue;
            if (DGRAD) *(bf16x4*)(Bs + (tid / BTPR + BRPI * i) * LDH + b_col) = to_bf16x4(rbf[i]);
            else *(bf16x8*)(Bs + (tid / BTPR + BRPI * i) * LDH + b_col) = rbh[i];
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
        const __bf16* As = ldsh + buf * STAGE_H + (wm * WM + l31) * LDH + 8 * h;
        const __bf16* Bs = ldsh + buf * STAGE_H + A_H + (wn * WN + l31) * LDH + 8 * h;
        bf16x8 fa[2][MT], fb[2][NT];                     // fragments of k-step s+1 are read while step s multiplies
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[0][i] = *(const bf16x8*)(As + i * 32 * LDH);
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[0][j] = *(const bf16x8*)(Bs + j * 32 * LDH);
#pragma unroll
        for (int s = 0; s < KT_ / 16; ++s) {
            if (s + 1 < KT_ / 16) {
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[(s + 1) & 1][i] = *(const bf16x8*)(As + i * 32 * LDH + 16 * (s + 1));
#pragma unroll
                for (int j = 0; j < NT; ++j) fb[(s + 1) & 1][j] = *(const bf16x8*)(Bs + j * 32 * LDH + 16 * (s + 1));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma_bf16(fa[s & 1][i], fb[s & 1][j], acc[i][j]);
        }
    };

    // ---- pipeline: registers hold tile kt+1 while LDS[cur] feeds the MFMAs; two blocks per CU hide the rest
    load_a(0); load_b(0);
    store_a(0); store_b(0);
    if (KT > 1) { load_a(1); load_b(1); }
    __syncthreads();
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    auto tile = [&](int kt, auto CUR) {
        constexpr int cur = decltype(CUR)::value, nxt = cur ^ 1;
        if (kt + 1 < KT) { store_a(nxt); store_b(nxt); }
        if (kt + 2 < KT) { load_a(kt + 2); load_b(kt + 2); }
        compute(cur);
        __syncthreads();
    };
    if (one_stage) {
        // ONE LDS stage (launched with half the LDS): the next tile waits in registers until this one is consumed -- two
        // barriers per K-tile, no store / MFMA overlap inside the workgroup, but up to twice the resident workgroups.  For
        // the short-K launches with a heavy epilogue (fused input gradients) residency is what counts.
        for (int kt = 0; kt < KT; ++kt) {
            compute(0);
            __syncthreads();
            if (kt + 1 < KT) { store_a(0); store_b(0); }
            if (kt + 2 < KT) { load_a(kt + 2); load_b(kt + 2); }
            __syncthreads();
        }
    } else {
        for (int kt = 0; kt < KT; kt += 2) {
            tile(kt, P0{});
            if (kt + 1 < KT) tile(kt + 1, P1{});
        }
    }
    // (the fp32 kernels request the fused epilogue's operands at workgroup start; here both that -- 140 instead of 82
    //  registers, fewer resident workgroups: 6.3 -> 7.0 ms per step on BASELINE configs[4] -- and issuing them together
    //  right here in front of the staging -- 4.85 -> 5.03 ms -- were measured to LOSE)
    igemm_epilogue_typed<BM, BN, WM, WN, DGRAD>(acc, (float*)ldsh, g, bias, addend, dst, bn_part, relu, m0, n0, grp, mt_id, &fz,
                                                dst_bf16 != 0);
}

// W (HWIO fp32 [taps][Cin/groups][Cout]) -> bf16 W^T [taps][Cout][Cin/groups]
__global__ __launch_bounds__(256) void weight_bf16t_kernel(const float* __restrict__ w, __bf16* __restrict__ wt, int taps,
                                                           int Cg, int Cout) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z, k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) tile[r][tx] = w[((long)tap * Cg + k0 + r) * Cout + n0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) wt[((long)tap * Cout + n0 + r) * Cg + k0 + tx] = (__bf16)tile[tx][r];
}

template <int BM, int BN, int WM, int WN, bool DGRAD>
static void launch_bf16(const void* src, const void* w, const float* bias, const float* addend, float* dst,
                        double* bn_part, const ConvGeom& g, int relu, hipStream_t st, const BnBwdFuse& fz, int io) {
    const int mtiles = sc_cdiv(g.M, BM), ntiles = g.Ng / BN;
    dim3 grid(mtiles * ntiles * g.groups);
    ConvGeom gg = g;
    gg.inv_hw = 1.0 / ((double)g.Ho * g.Wo);
    gg.inv_wo = 1.0f / (float)g.Wo;
    gg.inv_ho = 1.0f / (float)g.Ho;
    const size_t epi = (size_t)4 * WM * (WN + 4) * sizeof(float);
    auto go = [&](auto kern, int kt) {
        // one LDS stage (half the LDS, up to twice the resident workgroups) for launches that live on loads in flight rather
        // than on the K loop: a single K-tile (the second stage would never be touched), and the input gradients with up to
        // ONE_STAGE_MAX K-tiles (the fused ones in front of their heavy epilogue gain most)
        const int ktiles = g.R * g.S * (g.Cg / kt);
        // (measured on BASELINE configs[4]: the fused launches 4.63 -> 4.20 ms per step, tools_dev/tune_fused_dgrad_bf16.py;
        //  all input-gradient launches with one stage up to 32 K-tiles 8.14 -> 7.84 ms of kernel time; the forward
        //  launches 4.70 -> 4.66: within noise, kept two-stage)
        constexpr int ONE_STAGE_MAX = 32;
        const int one = ktiles <= 1 || (DGRAD && ktiles <= ONE_STAGE_MAX);
        size_t lds = (size_t)(one ? 1 : 2) * (BM + BN) * (kt + 8) * 2;
        if (lds < epi) lds = epi;
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, src, w, bias, addend, dst, bn_part, gg, relu, mtiles, ntiles, fz,
                           (io & SC_IO_Y_BF16) ? 1 : 0, one);
    };
    if (io & SC_IO_X_BF16) {
        if (g.Cg % 64 == 0) go(igemm_bf16_kernel<BM, BN, WM, WN, 64, DGRAD, true>, 64);
        else go(igemm_bf16_kernel<BM, BN, WM, WN, 32, DGRAD, true>, 32);
        return;
    }
    if (g.Cg % 64 == 0) go(igemm_bf16_kernel<BM, BN, WM, WN, 64, DGRAD>, 64);
    else go(igemm_bf16_kernel<BM, BN, WM, WN, 32, DGRAD>, 32);
}
static bool bhalo_dgrad_on(const ConvGeom& g, int io) {        // SCOUTER_BHALO=0: the tile kernel
    const char* e = getenv("SCOUTER_BHALO");
    return !(e && e[0] == '0') && bhalo_dgrad_ok(g, io);
}
static bool bhalo_fwd_on(const ConvGeom& g, int io, int tile) {  // tiles 1 (128 x 64) / 3 (128 x 32): the same 128-pixel partial rows
    const char* e = getenv("SCOUTER_BHALO");
    return !(e && e[0] == '0') && bhalo_fwd_ok(g, io) && tile == (g.Ng == 64 ? 1 : 3);
}
template <int NB>
static void launch_bhalo_fwd(const void* src, const void* w, const float* bias, const float* addend, float* dst, double* bn_part,
                             const ConvGeom& g, int relu, hipStream_t st, int io) {
    const int mtiles = sc_cdiv(g.M, 128);
    const size_t blds = bhalo_fwd_lds_bytes(NB);
    const char* we = getenv("SCOUTER_BHALO_WGS");
    long wgs = we ? atoi(we) : (NB == 1 ? 768 : 512);            // three / two workgroups per CU
    if (wgs > (long)mtiles * g.groups) wgs = (long)mtiles * g.groups;
    wgs -= wgs % g.groups;                                       // one group per workgroup (weights resident)
    if (wgs < g.groups) wgs = g.groups;
    auto kern = bhalo_fwd_kernel<NB>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds);
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(256), blds, st, (const unsigned short*)src, (const unsigned short*)w, bias,
                       addend, dst, bn_part, g, relu, mtiles, (io & SC_IO_Y_BF16) ? 1 : 0);
}
template <bool DGRAD>
static int dispatch_bf16(const void* src, const void* w, const float* bias, const float* addend, float* dst,
                         double* bn_part, const ConvGeom& g, int relu, int tile, hipStream_t st,
                         const BnBwdFuse& fz = BnBwdFuse{}, int io = 0) {
    SC_UNSUPPORTED(g.M < (1L << 31) && (long)g.H * g.W * g.C < (1L << 28),
                   "conv2d_bf16: more than 2^31 output pixels or an image above 2^28 elements is not supported");
    if (!DGRAD && bhalo_fwd_on(g, io, tile)) {
        // resident x rows for all nine taps (conv_halo_dgrad_bf16.h): bit-identical to tiles 1 / 3, same partial rows
        if (g.Ng == 64) launch_bhalo_fwd<2>(src, w, bias, addend, dst, bn_part, g, relu, st, io);
        else launch_bhalo_fwd<1>(src, w, bias, addend, dst, bn_part, g, relu, st, io);
        return sc_check_launch("conv2d_fwd_bf16(resident rows)");
    }
    switch (tile) {
        case 0: launch_bf16<128, 128, 64, 64, DGRAD>(src, w, bias, addend, dst, bn_part, g, relu, st, fz, io); break;
        case 1: launch_bf16<128, 64, 64, 32, DGRAD>(src, w, bias, addend, dst, bn_part, g, relu, st, fz, io); break;
        case 2: launch_bf16<64, 64, 32, 32, DGRAD>(src, w, bias, addend, dst, bn_part, g, relu, st, fz, io); break;
        default:
            if (DGRAD && bhalo_dgrad_on(g, io) && !fz.part2 && !bias && !bn_part && !relu) {
                // resident dy rows for all nine taps (conv_halo_dgrad_bf16.h): same tiles, partial rows and epilogue as <128, 32>
                const int mtiles = sc_cdiv(g.M, 128), halves = g.Cg / 32;
                const size_t blds = bhalo_lds_bytes(halves);
                const char* we = getenv("SCOUTER_BHALO_WGS");
                long wgs = we ? atoi(we) : (halves == 1 ? 768 : 512);          // three / two workgroups per CU
                if (wgs > (long)mtiles * g.groups) wgs = (long)mtiles * g.groups;
                wgs -= wgs % g.groups;                                         // one group per workgroup (weights resident)
                if (wgs < g.groups) wgs = g.groups;
                hipFuncSetAttribute((const void*)bhalo_dgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds);
                hipLaunchKernelGGL(bhalo_dgrad_kernel, dim3((unsigned)wgs), dim3(256), blds, st, (const unsigned short*)src,
                                   (const float*)w, addend, dst, g, mtiles, fz, (io & SC_IO_Y_BF16) ? 1 : 0);
                break;
            }
            launch_bf16<128, 32, 32, 32, DGRAD>(src, w, bias, addend, dst, bn_part, g, relu, st, fz, io);
            break;
    }
    return sc_check_launch(DGRAD ? "conv2d_dgrad_bf16" : "conv2d_fwd_bf16");
}
static int bf16_tile(const ConvGeom& g, int hint) {       // 0: 128x128  1: 128x64  2: 64x64  3: 128x32
    auto ok = [&](int t) { return (t == 0 && g.Ng % 128 == 0) || ((t == 1 || t == 2) && g.Ng % 64 == 0) || t == 3; };
    if (hint >= 0 && hint <= 3 && ok(hint)) return hint;
    auto blocks = [&](int bm, int bn) { return (long)sc_cdiv(g.M, bm) * (g.Ng / bn) * g.groups; };
    if (g.Ng % 128 == 0 && blocks(128, 128) >= 768) return 0;
    if (g.Ng % 64 == 0 && blocks(128, 64) >= 768) return 1;
    if (g.Ng % 64 == 0) return 2;
    return 3;
}
static int conv_out_b(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }

// ---- tile 4: the persistent pointwise input gradient with the fused BatchNorm-backward epilogue, all tensors bf16-stored
// (conv_pw_persist_bf16.h)
struct PwbPlan { int ks, wg_per_col; };
static bool pwb_geom_ok(const ConvGeom& g, bool has_addend, const BnBwdFuse& fz, int io) {
    const bool pointwise = g.R == 1 && g.S == 1 && g.stride == 1 && g.pad == 0 && g.groups == 1;
    const bool k_ok = g.Cg == 64 || g.Cg == 128 || g.Cg == 256 || g.Cg == 512;
    const bool stored_bf16 = (io & SC_IO_X_BF16) && (io & SC_IO_Y_BF16) && (fz.io & 1) && (!fz.part2 || (fz.io & 2)) &&
                             (!has_addend || (fz.io & 4));
    return pointwise && k_ok && g.Ng % 128 == 0 && fz.part1 != nullptr && stored_bf16 &&
           (g.M + 128) * g.Ng < (1L << 30) &&      // (every bf16 tensor < 2 GiB: the kernel clamps scalar row offsets at 2^31 - 1)
           g.M * g.Cg < (1L << 30);
}
static PwbPlan pwb_plan(const ConvGeom& g, bool two) {
    PwbPlan p;
    p.ks = g.Cg / 64;
    const int resident = p.ks <= 2 && !two ? PWB_RESIDENT : 1;          // workgroups per CU (registers; LDS: 48 / 64 KB for K = 64 / 128)
    const int colgroups = g.Ng / 128;
    int w = ((256 * resident) / colgroups) & ~7;
    p.wg_per_col = w < 8 ? 8 : w;
    return p;
}
static void launch_pwb_fused(const void* src, const float* w, const void* addend, void* dst, const ConvGeom& g, hipStream_t st,
                             const BnBwdFuse& fz) {
    const PwbPlan p = pwb_plan(g, fz.part2 != nullptr);
    const int mtiles = sc_cdiv(g.M, 64), grid = p.wg_per_col * (g.Ng / 128);
    const size_t lds = (size_t)128 * g.Cg * 2 + 4 * 8192;
    const long mask_words = ((g.M * g.Ng / 4 + 63) / 64) * 4;
#define PWB(KS_, TWO_)                                                                                             \
    do {                                                                                                           \
        auto kern = pwb_fused_kernel<KS_, TWO_>;                                                                   \
        static bool attr_set = false;                                                                              \
        if (!attr_set) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 64 * KS_ * 2 + 4 * 8192); attr_set = true; } \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, src, w, addend, dst, g.M, g.Ng, mtiles, p.wg_per_col, fz, \
                           mask_words);                                                                            \
    } while (0)
#define PWB2(KS_) do { if (fz.part2) PWB(KS_, true); else PWB(KS_, false); } while (0)
    if (p.ks == 1) PWB2(1);
    else if (p.ks == 2) PWB2(2);
    else if (p.ks == 4) PWB2(4);
    else PWB2(8);
#undef PWB2
#undef PWB
}

// ---- tile 4 of the typed input gradient WITHOUT the fused epilogue (conv_pw_persist_bf16.h pwb_dgrad_kernel): dY bf16, dx fp32
static bool pwb_plain_geom_ok(const ConvGeom& g, int io_xy) {
    const bool pointwise = g.R == 1 && g.S == 1 && g.stride == 1 && g.pad == 0 && g.groups == 1;
    const bool k_ok = g.Cg == 64 || g.Cg == 128 || g.Cg == 256 || g.Cg == 512 || g.Cg == 1024;
    return pointwise && k_ok && g.Ng % 64 == 0 && (io_xy & SC_IO_X_BF16) && !(io_xy & SC_IO_Y_BF16) &&
           (g.M + 128) * g.Ng < (1L << 30) && g.M * g.Cg < (1L << 30);
}
static void launch_pwb_dgrad(const void* src, const float* w, const void* addend, float* dst, const ConvGeom& g, hipStream_t st,
                             int add_bf16) {
    const int ks = g.Cg / 64, resident = ks <= 1 ? 3 : (ks <= 4 ? 2 : 1);     // LDS: 40 ... 64 / 96 / 160 KB
    int wg = ((256 * resident) / (g.Ng / 64)) & ~7;
    if (wg < 8) wg = 8;
    const int mtiles = sc_cdiv(g.M, 64), grid = wg * (g.Ng / 64);
    const size_t lds = (size_t)64 * g.Cg * 2 + 4 * 8192;
#define PWD(KS_, ADD_)                                                                                             \
    do {                                                                                                           \
        auto kern = pwb_dgrad_kernel<KS_, ADD_>;                                                                   \
        static bool attr_set = false;                                                                              \
        if (!attr_set) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 64 * KS_ * 2 + 4 * 8192); attr_set = true; } \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, src, w, addend, dst, g.M, g.Ng, mtiles, wg, add_bf16); \
    } while (0)
#define PWD2(KS_) do { if (addend) PWD(KS_, true); else PWD(KS_, false); } while (0)
    switch (ks) {
        case 1: PWD2(1); break;
        case 2: PWD2(2); break;
        case 4: PWD2(4); break;
        case 8: PWD2(8); break;
        default: PWD2(16); break;
    }
#undef PWD2
#undef PWD
}

// ---- tile 4 of the typed forward: persistent pointwise kernel, input stored as bf16 (conv_pw_persist_bf16.h pwb_fwd_kernel)
static bool pwb_fwd_geom_ok(const ConvGeom& g, bool plain, int io) {
    const bool pointwise = g.R == 1 && g.S == 1 && g.stride == 1 && g.pad == 0 && g.groups == 1;
    const bool k_ok = g.Cg == 64 || g.Cg == 128 || g.Cg == 256 || g.Cg == 512;
    return pointwise && k_ok && g.Ng % 128 == 0 && plain && (io & SC_IO_X_BF16) &&
           (g.M + 128) * g.Ng * ((io & SC_IO_Y_BF16) ? 2 : 4) < (1L << 32) && g.M * g.Cg < (1L << 30);
}
static int pwb_fwd_wg_per_col(const ConvGeom& g) {
    const int ks = g.Cg / 64, resident = ks == 1 ? 3 : (ks == 2 ? 2 : 1);     // LDS: 48 / 64 / 96 / 160 KB
    int w = ((256 * resident) / (g.Ng / 128)) & ~7;
    return w < 8 ? 8 : w;
}
static void launch_pwb_fwd(const void* src, const void* wt, void* dst, double* bn_part, const ConvGeom& g, hipStream_t st, int io) {
    const int wg = pwb_fwd_wg_per_col(g), mtiles = sc_cdiv(g.M, 64), grid = wg * (g.Ng / 128);
    const size_t lds = (size_t)128 * g.Cg * 2 + 4 * 8192;
#define PWF(KS_, YB_, ST_)                                                                                         \
    do {                                                                                                           \
        auto kern = pwb_fwd_kernel<KS_, YB_, ST_>;                                                                 \
        static bool attr_set = false;                                                                              \
        if (!attr_set) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 64 * KS_ * 2 + 4 * 8192); attr_set = true; } \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, src, wt, dst, bn_part, g.M, g.Ng, mtiles, wg);    \
    } while (0)
#define PWF3(KS_)                                                                                                  \
    do {                                                                                                           \
        if (io & SC_IO_Y_BF16) { if (bn_part) PWF(KS_, true, true); else PWF(KS_, true, false); }                  \
        else { if (bn_part) PWF(KS_, false, true); else PWF(KS_, false, false); }                                  \
    } while (0)
    switch (g.Cg / 64) {
        case 1: PWF3(1); break;
        case 2: PWF3(2); break;
        case 4: PWF3(4); break;
        default: PWF3(8); break;
    }
#undef PWF3
#undef PWF
}

extern "C" int scouter_conv2d_weight_bf16t(const float* w, void* wt, int kh, int kw, int Cin, int Cout, int groups,
                                           void* stream) {
    SC_REQUIRE(w && wt && groups > 0 && Cin % groups == 0 && Cout % groups == 0, "conv2d_weight_bf16t: bad arguments");
    const int Cg = Cin / groups;
    SC_UNSUPPORTED(Cg % 32 == 0 && Cout % 32 == 0, "conv2d_weight_bf16t: channels must be multiples of 32");
    hipLaunchKernelGGL(weight_bf16t_kernel, dim3(Cout / 32, Cg / 32, kh * kw), dim3(256), 0, (hipStream_t)stream, w,
                       (__bf16*)wt, kh * kw, Cg, Cout);
    return sc_check_launch("conv2d_weight_bf16t");
}

// `io`: SC_IO_X_BF16 -- x is stored as bf16 (the values the kernel would round to anyway: same result, half the bytes);
// SC_IO_Y_BF16 -- y is stored as bf16 (RNE of the fp32 result; bn_partial still sums the fp32 accumulators)
extern "C" int scouter_conv2d_fwd_bf16_io(const void* x, const void* wt_bf16, const float* bias, const float* addend,
                                          void* y, double* bn_partial, int B, int H, int W, int Cin, int Cout, int kh,
                                          int kw, int stride, int pad, int groups, int relu, int tile_hint, int io,
                                          void* stream) {
    SC_REQUIRE(x && wt_bf16 && y && B > 0 && H > 0 && W > 0, "conv2d_fwd_bf16: null pointer or empty shape");
    SC_REQUIRE((io & ~(SC_IO_X_BF16 | SC_IO_Y_BF16)) == 0, "conv2d_fwd_bf16: unknown io bits %d", io);
    SC_REQUIRE(!(bn_partial && relu), "conv2d_fwd_bf16: fused BatchNorm statistics are taken before any activation");
    SC_REQUIRE(groups > 0 && Cin % groups == 0 && Cout % groups == 0, "conv2d_fwd_bf16: channels not divisible by groups");
    const int Cg = Cin / groups, Ng = Cout / groups;
    SC_UNSUPPORTED(Cg % 32 == 0 && Ng % 32 == 0, "conv2d_fwd_bf16: per-group channels must be multiples of 32");
    ConvGeom g{B, H, W, Cin, conv_out_b(H, kh, stride, pad), conv_out_b(W, kw, stride, pad), Cout, kh, kw, stride, pad,
               groups, Cg, Ng, 0, Cout, Cg * Cout};
    g.M = (long)B * g.Ho * g.Wo;
    static const char* names[5] = {"igemm_fwd_bf16<128x128>", "igemm_fwd_bf16<128x64>", "igemm_fwd_bf16<64x64>", "igemm_fwd_bf16<128x32>",
                                   "igemm_fwd_bf16<persistent>"};
    SC_UNSUPPORTED(tile_hint != 4 || pwb_fwd_geom_ok(g, !bias && !addend && !relu, io),
                   "conv2d_fwd_bf16: tile 4 (persistent) covers 1x1 / stride 1 / groups 1 layers with 64 / 128 / 256 / 512 input "
                   "channels stored as bf16, 128-multiples of output channels, no bias / addend / ReLU");
    const int tile = tile_hint == 4 ? 4 : bf16_tile(g, tile_hint);
    ScProfScope prof(bhalo_fwd_on(g, io, tile) ? "bhalo_fwd<bf16>" : names[tile], (hipStream_t)stream, 2.0 * g.M * Cout * g.Cg * kh * kw,
                     ((io & SC_IO_X_BF16) ? 2.0 : 4.0) * B * H * W * Cin + ((io & SC_IO_Y_BF16) ? 2.0 : 4.0) * g.M * Cout);
    if (tile == 4) {
        launch_pwb_fwd(x, wt_bf16, y, bn_partial, g, (hipStream_t)stream, io);
        return sc_check_launch("conv2d_fwd_bf16(persistent)");
    }
    return dispatch_bf16<false>(x, wt_bf16, bias, addend, (float*)y, bn_partial, g, relu, tile, (hipStream_t)stream,
                                BnBwdFuse{}, io);
}
// rows of bn_partial the typed forward writes: tiles 0-3 one per M tile, tile 4 one per workgroup row
extern "C" int scouter_conv2d_fwd_bn_partial_rows_bf16(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride,
                                                       int pad, int groups, int tile_hint) {
    if (!(groups > 0 && Cin % groups == 0 && Cout % groups == 0)) return 0;
    const int Cg = Cin / groups, Ng = Cout / groups;
    ConvGeom g{B, H, W, Cin, conv_out_b(H, kh, stride, pad), conv_out_b(W, kw, stride, pad), Cout, kh, kw, stride, pad,
               groups, Cg, Ng, 0, Cout, Cg * Cout};
    g.M = (long)B * g.Ho * g.Wo;
    if (tile_hint == 4) return pwb_fwd_wg_per_col(g);
    return sc_cdiv(g.M, bf16_tile(g, tile_hint) == 2 ? 64 : 128);
}
extern "C" int scouter_conv2d_fwd_bf16(const float* x, const void* wt_bf16, const float* bias, const float* addend,
                                       float* y, double* bn_partial, int B, int H, int W, int Cin, int Cout, int kh,
                                       int kw, int stride, int pad, int groups, int relu, int tile_hint, void* stream) {
    return scouter_conv2d_fwd_bf16_io(x, wt_bf16, bias, addend, y, bn_partial, B, H, W, Cin, Cout, kh, kw, stride, pad,
                                      groups, relu, tile_hint, 0, stream);
}

// `io` (SCOUTER_DGRAD_IO_*): 1 / 2 -- x1 / x2 (the BatchNorm inputs the epilogue reads) are stored as bf16, 4 -- the addend
// is, 8 -- dy is (the values the kernel rounds an fp32 dy to), 16 -- dx is
extern "C" int scouter_conv2d_dgrad_bnbwd_bf16_io(const void* dy, const float* w, const void* addend, void* dx, int B,
                                                  int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad,
                                                  int groups, int tile_hint, const void* relu_mask, const void* x1,
                                                  const float* saved1, double* part1, const void* x2,
                                                  const float* saved2, double* part2, int io, void* stream) {
    SC_REQUIRE(dy && w && dx && B > 0, "conv2d_dgrad_bf16: null pointer or empty shape");
    SC_REQUIRE((io & ~31) == 0, "conv2d_dgrad_bf16: unknown io bits %d", io);
    SC_REQUIRE(groups > 0 && Cin % groups == 0 && Cout % groups == 0, "conv2d_dgrad_bf16: channels not divisible by groups");
    SC_REQUIRE(!part1 || (x1 && saved1), "conv2d_dgrad_bf16: fused BatchNorm backward needs x1 and saved1");
    SC_REQUIRE(!part2 || (part1 && x2 && saved2), "conv2d_dgrad_bf16: second fused BatchNorm needs the first, x2 and saved2");
    SC_UNSUPPORTED(stride == 1, "conv2d_dgrad_bf16: strided input gradients use the fp32 kernel");
    const int Cig = Cin / groups, Cog = Cout / groups;
    SC_UNSUPPORTED(Cig % 32 == 0 && Cog % 32 == 0, "conv2d_dgrad_bf16: per-group channels must be multiples of 32");
    const int Ho = conv_out_b(H, kh, stride, pad), Wo = conv_out_b(W, kw, stride, pad);
    ConvGeom g{B, Ho, Wo, Cout, H, W, Cin, kh, kw, stride, pad, groups, Cog, Cig, 0, Cout, Cig * Cout};
    g.M = (long)B * H * W;
    static const char* names[5] = {"igemm_dgrad_bf16<128x128>", "igemm_dgrad_bf16<128x64>", "igemm_dgrad_bf16<64x64>", "igemm_dgrad_bf16<128x32>",
                                   "igemm_dgrad_bf16<persistent>"};
    const BnBwdFuse fz{part1 ? (const unsigned long long*)relu_mask : nullptr, (const float*)x1, saved1, part1,
                       (const float*)x2, saved2, part2, io & 7};
    const int io_xy = ((io & 8) ? SC_IO_X_BF16 : 0) | ((io & 16) ? SC_IO_Y_BF16 : 0);
    // (the caller named the persistent kernel: its partial-row layout differs, so an unsupported request is an error, never a
    // silent re-route)
    SC_UNSUPPORTED(tile_hint != 4 || (part1 ? pwb_geom_ok(g, addend != nullptr, fz, io_xy) : pwb_plain_geom_ok(g, io_xy)),
                   "conv2d_dgrad_bf16: tile 4 (persistent) covers 1x1 / stride 1 / groups 1 input gradients -- with the fused "
                   "BatchNorm-backward epilogue: Cout of 64 / 128 / 256 / 512, 128-multiples of Cin, every tensor stored as bf16; "
                   "without: Cout of 64 ... 1024, 64-multiples of Cin, dy stored as bf16, dx fp32");
    const int tile = tile_hint == 4 ? 4 : bf16_tile(g, tile_hint);
    // algorithmic bytes: dy, dx and -- the fused launch -- the shortcut gradient, the BatchNorm input(s) and the ReLU bits
    const double out_elems = (double)g.M * Cin;
    const bool bhalo = tile == 3 && !part2 && bhalo_dgrad_on(g, io_xy);
    ScProfScope prof(bhalo ? "bhalo_dgrad<bf16>" : names[tile], (hipStream_t)stream, 2.0 * g.M * Cin * Cog * kh * kw,
                     ((io & 8) ? 2.0 : 4.0) * B * Ho * Wo * Cout + ((io & 16) ? 2.0 : 4.0) * out_elems +
                         (addend ? ((io & 4) ? 2.0 : 4.0) * out_elems : 0.0) +
                         (part1 ? ((io & 1) ? 2.0 : 4.0) * out_elems + (part2 ? ((io & 2) ? 2.0 : 4.0) * out_elems : 0.0) +
                                      (relu_mask ? out_elems / 8 : 0.0)
                                : 0.0));
    if (tile == 4) {
        if (part1) launch_pwb_fused(dy, w, addend, dx, g, (hipStream_t)stream, fz);
        else launch_pwb_dgrad(dy, w, addend, (float*)dx, g, (hipStream_t)stream, (io & 4) ? 1 : 0);
        return sc_check_launch("conv2d_dgrad_bf16(persistent)");
    }
    return dispatch_bf16<true>(dy, w, nullptr, (const float*)addend, (float*)dx, nullptr, g, 0, tile, (hipStream_t)stream,
                               fz, io_xy);
}
// partial rows ([rows][Cin][2] fp64) the fused launch writes: tiles 0-3 one per M tile, tile 4 one per workgroup row
extern "C" int scouter_conv2d_dgrad_bn_partial_rows_bf16(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride,
                                                         int pad, int groups, int tile_hint, int two_batchnorms) {
    if (!(groups > 0 && Cin % groups == 0 && Cout % groups == 0) || stride != 1) return 0;
    const int Cig = Cin / groups, Cog = Cout / groups;
    const int Ho = conv_out_b(H, kh, stride, pad), Wo = conv_out_b(W, kw, stride, pad);
    ConvGeom g{B, Ho, Wo, Cout, H, W, Cin, kh, kw, stride, pad, groups, Cog, Cig, 0, Cout, Cig * Cout};
    g.M = (long)B * H * W;
    if (tile_hint == 4) return pwb_plan(g, two_batchnorms != 0).wg_per_col;
    return sc_cdiv(g.M, bf16_tile(g, tile_hint) == 2 ? 64 : 128);
}
extern "C" int scouter_conv2d_dgrad_bnbwd_bf16(const float* dy, const float* w, const float* addend, float* dx, int B,
                                               int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad,
                                               int groups, int tile_hint, const void* relu_mask, const float* x1,
                                               const float* saved1, double* part1, const float* x2,
                                               const float* saved2, double* part2, void* stream) {
    return scouter_conv2d_dgrad_bnbwd_bf16_io(dy, w, addend, dx, B, H, W, Cin, Cout, kh, kw, stride, pad, groups, tile_hint,
                                              relu_mask, x1, saved1, part1, x2, saved2, part2, 0, stream);
}

extern "C" int scouter_conv2d_dgrad_bf16(const float* dy, const float* w, const float* addend, float* dx, int B, int H,
                                         int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int groups,
                                         int tile_hint, void* stream) {
    return scouter_conv2d_dgrad_bnbwd_bf16(dy, w, addend, dx, B, H, W, Cin, Cout, kh, kw, stride, pad, groups, tile_hint,
                                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}
