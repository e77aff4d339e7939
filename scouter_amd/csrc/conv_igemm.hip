// fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), NHWC activations,
// HWIO weights ([kh][kw][Cin/groups][Cout]).  Replaces the cuDNN/cuBLAS calls behind nn.Conv2d on the
// reference's hot path (timm/models/resnet.py:491-501, resnest.py:111-143, split_attn.py:54-60,
// sloter/slot_model.py:108) and their autograd backward (engine.py:33).
//
//   forward : Y[m=(b,oy,ox)][co] = sum_{tap,ci} X[b, oy*s-p+r, ox*s-p+q, ci] * W[tap][ci][co]   (+bias,+addend,relu)
//   dgrad   : dX[m=(b,iy,ix)][ci] = sum_{tap,co} dY[b,(iy+p-r)/s,(ix+p-q)/s,co] * W[tap][ci][co] (+addend)
//   wgrad   : dW[tap][ci][co]     = sum_m X[pixel(m,tap)][ci] * dY[m][co]      (split over m, deterministic reduce)
//
// Block = 256 threads = 4 waves (one per SIMD), 2 blocks per CU; block tile BMxBNx32, double-buffered LDS.
// fp32 MFMA issues once per 64 cycles per SIMD, so the loop is matrix-pipe bound as long as the (cheap) LDS
// reads and the global prefetch of the next K-tile hide under it -- no reshaping tricks, exact fp32 (fmaf chain).
#include "conv_common.h"
#include "conv_wgrad_bf16.h"
#include "conv_wgrad_taps.h"
#include "conv_wgrad_taps_x3.h"
#include "conv_wgrad_taps_bf16.h"
#include "conv_pw_persist.h"
#include "conv_pw_persist_x3.h"

#include <stdlib.h>
#include <type_traits>

#ifdef IGEMM_STAMPS    // dev builds (tools_dev/igemm_stamps.sh): per workgroup {start, first fragments, K loop done, end, XCC / CU id}
__device__ long long* g_igemm_stamps = nullptr;
extern "C" int scouter_dev_set_igemm_stamps(void* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_igemm_stamps), &p, sizeof(p));
}
#define IG_STAMP(k) if (g_igemm_stamps && threadIdx.x == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_igemm_stamps[8L * blockIdx.x + (k)] = (long long)wall_clock64(); }
#else
#define IG_STAMP(k)
#endif

constexpr int BK = 32;
constexpr int LDK = BK + 4;   // k-contiguous tiles: 36-float rows -> conflict-free ds_read_b128, 16B aligned

template <int BM, int BN, bool B_KC>
struct TileCfg {
    static constexpr int A_ELEMS = BM * LDK;
    static constexpr int B_ELEMS = B_KC ? BN * LDK : BK * BN;
    static constexpr int STAGE = A_ELEMS + B_ELEMS;
};

// ----------------------------------------------------------------------------------------------------------------
// forward / dgrad
// ----------------------------------------------------------------------------------------------------------------
// PW ("pointwise"): 1x1 / stride 1 / pad 0 -- GEMM row m IS pixel m, so the block prologue needs no (b, y, x)
// decomposition, no tap masks and no per-tap offset switch: the A descriptor covers exactly the block's valid rows
// (rows beyond M read zeros through num_records).  36 of resnest26d's 48 convolutions are pointwise and they have the
// SHORTEST K loops (Cin/32 = 2..64 tiles), where the general prologue's ~300 VALU instructions per thread -- fp32 MFMA
// shares the vector lanes -- were a tenth to a third of the block's matrix time (tools_dev/isa_phases.py).
// FUSE: the epilogue carries a BatchNorm-backward reduction (conv_common.h BnBwdFuse) -- its own instantiation, so the
// plain input gradient keeps its leaner epilogue (the fused one prefetches up to three more operands: +30 registers)
template <int BM, int BN, int WM, int WN, bool DGRAD, bool STRIDED, bool PW, bool FUSE = false>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const float* __restrict__ src, const float* __restrict__ wgt,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ addend, float* __restrict__ dst,
                                                       double* __restrict__ bn_part, ConvGeom g, int relu, int mtiles,
                                                       int ntiles, BnBwdFuse fz) {
    constexpr bool B_KC = DGRAD;
    using T = TileCfg<BM, BN, B_KC>;
    constexpr int MT = WM / 32, NT = WN / 32, WAVES_N = BN / WN;
    constexpr int AI = BM / 32, BI = BN / 32;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
    __shared__ __attribute__((aligned(16))) float lds[2 * T::STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int nblk = mtiles * ntiles * g.groups;
    const int bid = xcd_remap(blockIdx.x, nblk);
    const int grp = bid % g.groups;
    const int nt_id = (bid / g.groups) % ntiles;
    const int mt_id = bid / (g.groups * ntiles);
    const long m0 = (long)mt_id * BM;
    const int n0 = nt_id * BN;
    const int cpt = g.Cg / BK;                 // K chunks per filter tap
    const int KT = g.R * g.S * cpt;
    IG_STAMP(0)
    EpiPre<(WM / (64 / (WN / 4)) <= 8 ? WM / (64 / (WN / 4)) : 1)> epre;
    epre.on = false;
    if constexpr (DGRAD && FUSE) igemm_epilogue_prefetch<BM, BN, WM, WN>(epre, g, addend, m0, n0, grp, fz);

    unsigned a_mask[AI];
    int a_y[AI], a_x[AI];
    long a_base[AI];
    const int a_col = (tid & 7) * 4;
    constexpr bool lin = !(DGRAD && STRIDED);          // source pixel is linear in the tap index
    const int rows_valid = (int)(g.M - m0 < BM ? g.M - m0 : BM);
    const float* wbase = wgt + (long)grp * (DGRAD ? g.Cg : g.Ng);   // group offset along the contiguous Cout axis
    constexpr unsigned OOB = 0x80000000u;
    unsigned a_voff[AI], a_veff[AI];
    long shift = 0;
    __amdgpu_buffer_rsrc_t rs_a;
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, 0x7fffffff, 0x00020000);
    if constexpr (PW) {
        // pointwise: row r of the block is pixel m0 + r; the descriptor spans the block's valid rows only, so rows
        // beyond M read zeros without any per-row state
        rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(src + m0 * g.C), 0,
                                                 (unsigned)rows_valid * (unsigned)g.C * 4u, 0x00020000);
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            a_voff[i] = (unsigned)(((tid >> 3) + 32 * i) * g.C + grp * g.Cg + a_col) * 4u;
            a_veff[i] = a_voff[i];
            a_mask[i] = 0; a_y[i] = a_x[i] = 0; a_base[i] = 0;
        }
    } else {
        // ---- per-thread A rows (fixed for the whole K loop): pointer of filter tap (0,0) + a validity bit per tap, so
        // the in-loop address work is one 64-bit add of a wave-uniform tap offset and a bit test (the general
        // bounds/stride arithmetic per load cost ~25 VALU issues each and measurably starved the MFMA issue)
        // pixel (b, y, x) of each row without integer division: the block's first pixel once (fp64 reciprocal, exact for
        // M < 2^31), then small carries per row; tap validity is separable (row bits x column bits).  The first version
        // spent ~780 VALU instructions per block here (64-bit div/mod per row, R*S compares) -- fp32 MFMA time.
        int a_rel[AI];
        const int hw = g.Ho * g.Wo;
        const int blk_b = (int)(((double)(unsigned)m0 + 0.5) * g.inv_hw);
        const int blk_rem = (int)((unsigned)m0 - (unsigned)blk_b * (unsigned)hw);
        const int blk_y = fast_div(blk_rem, g.inv_wo), blk_x = blk_rem - blk_y * g.Wo;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int rowoff = (tid >> 3) + 32 * i;
            const bool okm = rowoff < rows_valid;
            const int tx = blk_x + rowoff, qx = fast_div(tx, g.inv_wo), x = tx - qx * g.Wo;
            const int ty = blk_y + qx, qy = fast_div(ty, g.inv_ho), y = ty - qy * g.Ho;
            const int b = okm ? blk_b + qy : 0;
            a_base[i] = (long)b * g.H * g.W;
            a_rel[i] = okm ? qy * g.H * g.W : 0;                      // pixels from the block's base image
            if (DGRAD) { a_y[i] = y + g.pad; a_x[i] = x + g.pad; }
            else       { a_y[i] = y * g.stride - g.pad; a_x[i] = x * g.stride - g.pad; }
            unsigned mask = 0;
            if (DGRAD && STRIDED) {
                for (int r = 0; r < g.R; ++r)
                    for (int q = 0; q < g.S; ++q) {
                        const int ty2 = a_y[i] - r, tx2 = a_x[i] - q;
                        bool ok = okm && ty2 >= 0 && tx2 >= 0;
                        const int iy = ty2 / g.stride, ix = tx2 / g.stride;
                        ok = ok && (iy * g.stride == ty2) && (ix * g.stride == tx2) && iy < g.H && ix < g.W;
                        mask |= (ok ? 1u : 0u) << (r * g.S + q);
                    }
            } else {
                unsigned colbits = 0;
                for (int q = 0; q < g.S; ++q)
                    colbits |= ((unsigned)(DGRAD ? a_x[i] - q : a_x[i] + q) < (unsigned)g.W ? 1u : 0u) << q;
                for (int r = 0; r < g.R; ++r)
                    mask |= ((unsigned)(DGRAD ? a_y[i] - r : a_y[i] + r) < (unsigned)g.H ? colbits : 0u) << (r * g.S);
                mask = okm ? mask : 0u;
            }
            a_mask[i] = mask;
        }
        // ---- buffer addressing.  fp32 MFMA runs on the vector FP32 lanes (its peak IS the vector peak), so every VALU
        // instruction in the K loop is matrix throughput lost -- unlike SALU and memory-instruction issue, which are free.
        // Loads therefore go through buffer descriptors: address = base (SGPR) + per-row byte offset (VGPR, fixed for the
        // whole loop) + wave-uniform tap/channel offset (SGPR, scalar arithmetic); a padding row gets an offset beyond
        // num_records for that tap and the hardware returns zeros -- no exec masking, zero fills or 64-bit VALU adds.
        // The base is block-relative (image of the block's first pixel, shifted down so that every tap offset is >= 0; it
        // may precede the allocation, only valid taps are ever dereferenced), so offsets fit 31 bits for any tensor size.
        const long img_elems = (long)g.H * g.W * g.C;
        shift = DGRAD ? ((long)(g.R - 1) * g.W + (g.S - 1)) * g.C : ((long)g.pad * g.W + g.pad) * g.C;
        rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (long)blk_b * img_elems - shift), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int e = DGRAD ? (a_rel[i] + a_y[i] * g.W + a_x[i]) * g.C
                                : (a_rel[i] + (a_y[i] + g.pad) * g.W + (a_x[i] + g.pad)) * g.C;
            a_voff[i] = (unsigned)(e + grp * g.Cg + a_col) * 4u;
            a_veff[i] = OOB;
        }
    }

    // ---- software pipeline (per wave; the MFMA stream never waits for memory inside a K-tile):
    //   global -> registers G   two tiles ahead        (issued in the shadow of tile k's MFMAs, phase 1)
    //   G -> LDS[(k+1)&1]        one tile ahead         (phase 1)                       -> barrier
    //   LDS -> fragment registers, per HALF K-tile (16 of the 32 k's): F1(k) during phase 1, F0(k+1) during phase 2
    // Counters on the first version (fragments fetched at the top of every tile) showed the matrix pipe 64 % busy:
    // the two co-resident blocks convoy and sit in their LDS round trips together.
    f32x4 ra[AI], rb[BI];
    auto load_a = [&](int kt) {
        if constexpr (PW) {                            // one tap: K-tile kt is channels [32 kt, 32 kt + 32)
#pragma unroll
            for (int i = 0; i < AI; ++i)
                ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_voff[i], kt * (BK * 4), 0));
            return;
        }
        const int tap = kt / cpt, c0 = (kt - tap * cpt) * BK;
        const int r = tap / g.S, q = tap - r * g.S;
        // wave-uniform element offset of this (tap, channel chunk) relative to tap (0,0)
        const long toff = (DGRAD ? -((long)r * g.W + q) : ((long)r * g.W + q)) * g.C + c0;
        if (lin) {
            if (c0 == 0) {                             // new filter tap (wave-uniform): which rows are padding now
#pragma unroll
                for (int i = 0; i < AI; ++i) a_veff[i] = ((a_mask[i] >> tap) & 1u) ? a_voff[i] : OOB;
            }
            const int soff = (int)((DGRAD ? shift + toff : toff) * 4);
#pragma unroll
            for (int i = 0; i < AI; ++i)
                ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_veff[i], soff, 0));
        } else {
#pragma unroll
            for (int i = 0; i < AI; ++i) {             // strided dgrad (resnet18 only): general addressing
                const bool ok = (a_mask[i] >> tap) & 1u;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    const int iy = (a_y[i] - r) / g.stride, ix = (a_x[i] - q) / g.stride;
                    v = *(const f32x4*)(src + (a_base[i] + (long)iy * g.W + ix) * g.C + grp * g.Cg + c0 + a_col);
                }
                ra[i] = v;
            }
        }
    };
    const float* b_ptr[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        if (B_KC) b_ptr[i] = wbase + (long)(n0 + (tid >> 3) + 32 * i) * g.wrow + a_col;
        else { const int c = tid + 256 * i, krow = c / (BN / 4), col4 = c % (BN / 4); b_ptr[i] = wbase + (long)krow * g.wrow + n0 + col4 * 4; }
    }
    unsigned b_voff[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) b_voff[i] = (unsigned)((b_ptr[i] - wbase) * 4);
    auto load_b = [&](int kt) {
        const int tap = PW ? 0 : kt / cpt, c0 = (kt - tap * cpt) * BK;
        const long woff = (long)tap * g.wtap + (B_KC ? (long)c0 : (long)c0 * g.wrow);     // wave-uniform
#pragma unroll
        for (int i = 0; i < BI; ++i)
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, b_voff[i], (int)(woff * 4), 0));
    };
    auto store_a = [&](int buf) {
        float* As = lds + buf * T::STAGE;
#pragma unroll
        for (int i = 0; i < AI; ++i) *(f32x4*)(As + ((tid >> 3) + 32 * i) * LDK + a_col) = ra[i];
    };
    auto store_b = [&](int buf) {
        float* Bs = lds + buf * T::STAGE + T::A_ELEMS;
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            if (B_KC) *(f32x4*)(Bs + ((tid >> 3) + 32 * i) * LDK + a_col) = rb[i];
            else { const int c = tid + 256 * i; *(f32x4*)(Bs + c * 4) = rb[i]; }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // half-tile fragments: p = 0 -> k-steps 0..7, p = 1 -> k-steps 8..15 (lane half h covers k = 16 h + step)
    f32x4 fa[2][MT][2];
    f32x4 fbk[2][NT][2];
    float fbv[2][8][NT];
    auto frag_a = [&](int buf, auto P) {
        constexpr int pp = decltype(P)::value;
        const float* As = lds + buf * T::STAGE;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                fa[pp][i][q] = *(const f32x4*)(As + (wm * WM + i * 32 + l31) * LDK + h * 16 + (2 * pp + q) * 4);
    };
    auto frag_b = [&](int buf, auto P) {
        constexpr int pp = decltype(P)::value;
        const float* Bs = lds + buf * T::STAGE + T::A_ELEMS;
        if (B_KC) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    fbk[pp][j][q] = *(const f32x4*)(Bs + (wn * WN + j * 32 + l31) * LDK + h * 16 + (2 * pp + q) * 4);
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int j = 0; j < NT; ++j) fbv[pp][s][j] = Bs[(h * 16 + 8 * pp + s) * BN + wn * WN + j * 32 + l31];
        }
    };
    auto mma = [&](auto P, int s0, int s1) {          // k-steps [s0, s1) of half pp
        constexpr int pp = decltype(P)::value;
#pragma unroll
        for (int s = s0; s < s1; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = mfma32(fa[pp][i][s >> 2][s & 3], B_KC ? fbk[pp][j][s >> 2][s & 3] : fbv[pp][s][j],
                                       acc[i][j]);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
#define SB() __builtin_amdgcn_sched_barrier(0)

    // prologue: tile 0 -> LDS[0], tile 1 -> registers, F0(0) -> fragment registers
    load_a(0); load_b(0);
    store_a(0); store_b(0);
    if (KT > 1) { load_a(1); load_b(1); }
    __syncthreads();
    frag_a(0, P0{}); frag_b(0, P0{});
    IG_STAMP(1)
    // one K-tile; the LDS buffer index is a compile-time constant (the loop is unrolled by two) so that every LDS
    // address is a register + immediate -- no per-tile VALU address arithmetic
    auto tile = [&](int kt, auto CUR) {
        constexpr int cur = decltype(CUR)::value, nxt = cur ^ 1;
        const bool has1 = kt + 1 < KT, has2 = kt + 2 < KT;
        // ---- phase 1: MFMAs of F0(kt); in their shadow: F1(kt) <- LDS[cur], G(kt+1) -> LDS[nxt], G(kt+2) <- global
        SB();
        frag_a(cur, P1{});
        mma(P0{}, 0, 1); SB();
        frag_b(cur, P1{});
        mma(P0{}, 1, 2); SB();
        if (has1) store_a(nxt);
        mma(P0{}, 2, 4); SB();
        if (has1) store_b(nxt);
        mma(P0{}, 4, 5); SB();
        if (has2) load_a(kt + 2);
        mma(P0{}, 5, 7); SB();
        if (has2) load_b(kt + 2);
        mma(P0{}, 7, 8); SB();
        __syncthreads();
        // ---- phase 2: MFMAs of F1(kt); in their shadow: F0(kt+1) <- LDS[nxt]
        SB();
        if (has1) frag_a(nxt, P0{});
        mma(P1{}, 0, 1); SB();
        if (has1) frag_b(nxt, P0{});
        mma(P1{}, 1, 8); SB();
    };
    for (int kt = 0; kt < KT; kt += 2) {
        tile(kt, P0{});
        if (kt + 1 < KT) tile(kt + 1, P1{});
    }
#undef SB
    IG_STAMP(2)

    // ---- epilogue (conv_common.h): LDS-staged vector stores, fused bias / addend / ReLU / BatchNorm statistics
    static_assert(4 * WM * (WN + 4) <= 2 * T::STAGE, "epilogue staging fits in the K-loop LDS");
    igemm_epilogue<BM, BN, WM, WN, DGRAD && FUSE, true>(acc, lds, g, bias, addend, dst, bn_part, relu, m0, n0, grp, mt_id, &fz,
                                                        &epre);
#ifdef IGEMM_STAMPS
    if (g_igemm_stamps && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g_igemm_stamps[8L * blockIdx.x + 3] = (long long)wall_clock64();
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_igemm_stamps[8L * blockIdx.x + 4] = (long long)(((xcc & 0xf) << 16) | (hw & 0xffff));
    }
#endif
}

// ----------------------------------------------------------------------------------------------------------------
// wgrad: C[tap][ci][co] partial sums over a pixel range; both operands are "reduction-row, channel-contiguous"
// ----------------------------------------------------------------------------------------------------------------
// MODE 0: general addressing (strided / shrinking convolutions, tiny maps).
// MODE 1/2: "same" convolutions (stride 1, Ho = H, Wo = W): the source pixel of filter tap (r, q) is m + const, so both
// operands are affine in the pixel index m and go through buffer descriptors rebuilt per chunk with SCALAR arithmetic
// (base = first pixel of the chunk, num_records = bytes up to the end of the block's pixel range: rows beyond it
// read zeros); 2 = 1x1 (no padding, no per-thread state at all), 1 = padded taps (branch-free x/y walk for the
// validity bit only).  The general path spent 223 VALU instructions per 64 MFMAs in the loop; fp32 MFMA shares the
// vector FP32 lanes, so that was a quarter of the kernel.
template <int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const float* __restrict__ act, const float* __restrict__ dy,
                                                       float* __restrict__ out, ConvGeom g, int ci_tiles,
                                                       int co_tiles, long pix_per_split, long slab,
                                                       float* __restrict__ dw, unsigned* __restrict__ arrival) {
    constexpr int MT = WM / 32, NT = WN / 32, WAVES_N = BN / WN, WAVES_MN = (BM / WM) * (BN / WN);
    constexpr int WK = 4 / WAVES_MN;           // small tiles: the spare waves split the 16 k-steps of a chunk
    constexpr int SPW = 16 / WK;
    constexpr int AI = (BM + 31) / 32, BI = (BN + 31) / 32;
    constexpr int STAGE = BK * BM + BK * BN;
    static_assert(WAVES_MN * WK == 4 && WK * BM * BN <= 2 * STAGE, "tile config");
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int wmn = wave % WAVES_MN, wk = wave / WAVES_MN;
    const int wm = wmn / WAVES_N, wn = wmn % WAVES_N;
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
    const int hw = g.Ho * g.Wo;
#ifdef IGEMM_STAMPS
    if (g_igemm_stamps && threadIdx.x == 0) { g_igemm_stamps[8L * (blockIdx.x + (long)gridDim.x * blockIdx.y) + 0] = (long long)wall_clock64(); }
#undef IG_STAMP
#define IG_STAMP(k) if (g_igemm_stamps && threadIdx.x == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_igemm_stamps[8L * (blockIdx.x + (long)gridDim.x * blockIdx.y) + (k)] = (long long)wall_clock64(); }
#endif

    // per-thread pixel state of its A rows, advanced by BK pixels per chunk without divisions
    int pb[AI], py[AI], px[AI];
    int qy1 = 0, qx1 = 0;               // MODE 1: coordinates of chunk pixel (lane & 31), one per lane
    if (MODE != 2) {                    // (the 1x1 path needs no pixel coordinates at all)
        // block's first pixel once (fp64 reciprocal, exact below 2^31), then carries per row -- no integer division
        const int blk_b = (int)(((double)(unsigned long)mbeg + 0.5) * g.inv_hw);
        const int blk_rem = (int)(mbeg - (long)blk_b * hw);
        const int blk_y = fast_div(blk_rem, g.inv_wo), blk_x = blk_rem - blk_y * g.Wo;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int c = tid + 256 * i, prow = MODE == 1 ? l31 : c / (BM / 4);
            const int tx = blk_x + prow, qx = fast_div(tx, g.inv_wo);
            const int ty = blk_y + qx, qy = fast_div(ty, g.inv_ho);
            px[i] = tx - qx * g.Wo;
            py[i] = ty - qy * g.Ho;
            pb[i] = blk_b + qy;
        }
        qx1 = px[0]; qy1 = py[0];
    }
    const int a_c4 = ((tid % (BM / 4)) * 4), b_c4 = ((tid % (BN / 4)) * 4);
    const float* actg = act + grp * g.Cg + ci0 + a_c4;
    const float* dyg = dy + grp * g.Ng + co0 + b_c4;
    f32x4 ra[AI], rb[BI];
    long a_m = mbeg;                    // first pixel of the chunk the next load_a fetches
    // affine modes: fixed per-thread byte offsets inside a chunk
    constexpr unsigned OOB = 0x80000000u;
    unsigned a_voff[AI], b_voff[BI], a_bit[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) a_bit[i] = 1u << ((tid + 256 * i) / (BM / 4));
#pragma unroll
    for (int i = 0; i < AI; ++i) a_voff[i] = (unsigned)((((tid + 256 * i) / (BM / 4)) * (long)g.C + grp * g.Cg + ci0 + a_c4) * 4);
#pragma unroll
    for (int i = 0; i < BI; ++i) b_voff[i] = (unsigned)((((tid + 256 * i) / (BN / 4)) * (long)g.N + grp * g.Ng + co0 + b_c4) * 4);
    const long tapoff = (long)(r - g.pad) * g.W + (q - g.pad);           // source pixel = m + tapoff
    const int adv_x = BK % g.Wo, adv_y = BK / g.Wo;
    auto records = [&](long m_chunk, int row_elems) {                    // bytes of the rows m_chunk .. mend-1
        long n = (mend - m_chunk) * (long)row_elems * 4;
        return (unsigned)(n < 0 ? 0 : (n > 0x7fffffffL ? 0x7fffffffL : n));
    };
    auto load_a = [&]() {
        if (MODE != 0) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(act + (a_m + tapoff) * g.C), 0, records(a_m, g.C), 0x00020000);
            unsigned vmask = 0xffffffffu;              // bit p: the tap's source of chunk pixel p is inside the image
            if (MODE == 1) {
                // every lane tests ONE pixel of the chunk (p = lane & 31); the ballot makes the 32 validity bits a
                // scalar, so a load costs a bit test instead of its own coordinate walk (was 84 VALU per chunk)
                const int iy = qy1 - g.pad + r, ix = qx1 - g.pad + q;
                vmask = (unsigned)__ballot((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W);
                qx1 += adv_x;                                  // advance by one chunk of BK pixels, branch-free
                const int wrap = qx1 >= g.Wo ? 1 : 0;
                qx1 -= wrap ? g.Wo : 0;
                qy1 += adv_y + wrap;
                qy1 -= qy1 >= g.Ho ? g.Ho : 0;
            }
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                unsigned vo = a_voff[i];
                if (MODE == 1) vo = (vmask & a_bit[i]) ? vo : OOB;
                ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
            }
            a_m += BK;
            return;
        }
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int c = tid + 256 * i, prow = c / (BM / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const int iy = py[i] * g.stride - g.pad + r, ix = px[i] * g.stride - g.pad + q;
            if (a_m + prow < mend && iy >= 0 && ix >= 0 && iy < g.H && ix < g.W)
                v = *(const f32x4*)(actg + (((long)pb[i] * g.H + iy) * g.W + ix) * g.C);
            ra[i] = v;
            px[i] += BK;                                 // advance this row by one chunk
            while (px[i] >= g.Wo) { px[i] -= g.Wo; ++py[i]; }
            while (py[i] >= g.Ho) { py[i] -= g.Ho; ++pb[i]; }
        }
        a_m += BK;
    };
    long b_m = mbeg;
    auto load_b = [&]() {
        if (MODE != 0) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(dy + b_m * g.N), 0, records(b_m, g.N), 0x00020000);
#pragma unroll
            for (int i = 0; i < BI; ++i)
                rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, b_voff[i], 0, 0));
            b_m += BK;
            return;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int c = tid + 256 * i, prow = c / (BN / 4);
            const long m = b_m + prow;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < mend) v = *(const f32x4*)(dyg + m * g.N);
            rb[i] = v;
        }
        b_m += BK;
    };
    auto store_a = [&](int buf) {
        float* As = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < AI; ++i) *(f32x4*)(As + (tid + 256 * i) * 4) = ra[i];
    };
    auto store_b = [&](int buf) {
        float* Bs = lds + buf * STAGE + BK * BM;
#pragma unroll
        for (int i = 0; i < BI; ++i) *(f32x4*)(Bs + (tid + 256 * i) * 4) = rb[i];
    };
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // half-chunk fragments (same software pipeline as igemm_kernel): this wave's SPW k-steps split in two halves
    constexpr int HS = SPW / 2;
    float fa[2][HS][MT], fb[2][HS][NT];
    auto frag = [&](int buf, auto P) {
        constexpr int pp = decltype(P)::value;
        const float* As = lds + buf * STAGE;
        const float* Bs = As + BK * BM;
#pragma unroll
        for (int ss = 0; ss < HS; ++ss) {
            const int s = wk * SPW + pp * HS + ss;
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[pp][ss][i] = As[(h * 16 + s) * BM + wm * WM + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[pp][ss][j] = Bs[(h * 16 + s) * BN + wn * WN + j * 32 + l31];
        }
    };
    auto mma = [&](auto P, int s0, int s1) {
        constexpr int pp = decltype(P)::value;
#pragma unroll
        for (int ss = s0; ss < s1; ++ss)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma32(fa[pp][ss][i], fb[pp][ss][j], acc[i][j]);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
#define SB() __builtin_amdgcn_sched_barrier(0)
    if (KT > 0) {
        load_a(); load_b();
        store_a(0); store_b(0);
        if (KT > 1) { load_a(); load_b(); }
    }
    __syncthreads();
    if (KT > 0) frag(0, P0{});
    IG_STAMP(1)
    auto chunk = [&](int kt, auto CUR) {              // LDS buffer index is a compile-time constant (loop unrolled by 2)
        constexpr int cur = decltype(CUR)::value, nxt = cur ^ 1;
        const bool has1 = kt + 1 < KT, has2 = kt + 2 < KT;
        SB();
        frag(cur, P1{});
        mma(P0{}, 0, HS / 2); SB();
        if (has1) { store_a(nxt); store_b(nxt); }
        if (has2) load_a();
        mma(P0{}, HS / 2, HS); SB();
        if (has2) load_b();
        __syncthreads();
        SB();
        if (has1) frag(nxt, P0{});
        mma(P1{}, 0, HS); SB();
    };
    for (int kt = 0; kt < KT; kt += 2) {
        chunk(kt, P0{});
        if (kt + 1 < KT) chunk(kt + 1, P1{});
    }
#undef SB
    IG_STAMP(2)
    float* o = out + (long)split_id * slab + (long)tap * g.Cg * g.N;
    const bool coh = arrival != nullptr;               // partial tiles leave write-through: another workgroup sums them
    if (WK == 1) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ci = ci0 + wm * WM + i * 32 + mfma32_row(e, lane);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int co = grp * g.Ng + co0 + wn * WN + j * 32 + l31;
                    slab_store(o + (long)ci * g.N + co, acc[i][j][e], coh);
                }
            }
    } else {   // cross-wave (k-split) reduction through LDS, fixed order
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    lds[wk * BM * BN + (wm * WM + i * 32 + mfma32_row(e, lane)) * BN + wn * WN + j * 32 + l31] =
                        acc[i][j][e];
        __syncthreads();
        for (int e = tid; e < BM * BN; e += 256) {
            float v = lds[e];
#pragma unroll
            for (int k = 1; k < WK; ++k) v += lds[k * BM * BN + e];
            slab_store(o + (long)(ci0 + e / BN) * g.N + grp * g.Ng + co0 + e % BN, v, coh);
        }
    }
    if (coh)
        slab_tile_finish(out + (long)tap * g.Cg * g.N, dw + (long)tap * g.Cg * g.N, slab, gridDim.y, arrival + tile_id, 1, 0,
                         (long)ci0 * g.N, BM, g.N, grp * g.Ng + co0, BN);
#ifdef IGEMM_STAMPS
    if (g_igemm_stamps && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g_igemm_stamps[8L * (blockIdx.x + (long)gridDim.x * blockIdx.y) + 3] = (long long)wall_clock64();
        g_igemm_stamps[8L * (blockIdx.x + (long)gridDim.x * blockIdx.y) + 5] = KT;
    }
#endif
}

// Sums the split slabs in a fixed order (deterministic).  Block = 8 float4 columns x 32 split-lanes: each lane adds
// every 32nd slab with 4 loads in flight, then a fixed-order LDS combine -- a thread per element looping over
// hundreds of splits would serialise that many dependent HBM/L2 round trips.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ part, float* __restrict__ dst,
                                                          long n, int splits, long slab) {
    __shared__ f32x4 red[32][8];
    const int q = threadIdx.x & 7, lane = threadIdx.x >> 3;
    const long i = ((long)blockIdx.x * 8 + q) * 4;
    f32x4 s4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) s4[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < n) {
        int k = lane;
        for (; k + 96 < splits; k += 128) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s4[u] += *(const f32x4*)(part + (long)(k + 32 * u) * slab + i);
        }
        for (; k < splits; k += 32) s4[0] += *(const f32x4*)(part + (long)k * slab + i);
    }
    red[lane][q] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    __syncthreads();
    if (lane == 0 && i < n) {
        f32x4 s = red[0][q];
        for (int k = 1; k < 32; ++k) s += red[k][q];
        *(f32x4*)(dst + i) = s;
    }
}

// (also used by the plane weight-gradient kernel, conv_planes.hip)
void sc_launch_slab_reduce(const float* part, float* dst, long n, int splits, long slab, hipStream_t st) {
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(sc_cdiv(n / 4, 8)), dim3(256), 0, st, part, dst, n, splits, slab);
}

// ----------------------------------------------------------------------------------------------------------------
// host dispatch
// ----------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool DGRAD, bool STRIDED, bool PW>
static void launch_igemm_s(const float* src, const float* w, const float* bias, const float* addend, float* dst,
                           double* bn_part, const ConvGeom& g, int relu, hipStream_t st, const BnBwdFuse& fz) {
    const int mtiles = sc_cdiv(g.M, BM), ntiles = g.Ng / BN;
    dim3 grid(mtiles * ntiles * g.groups);
    ConvGeom gg = g;
    gg.inv_hw = 1.0 / ((double)g.Ho * g.Wo);
    gg.inv_wo = 1.0f / (float)g.Wo;
    gg.inv_ho = 1.0f / (float)g.Ho;
    if (DGRAD && fz.part1)
        hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, DGRAD, STRIDED, PW, DGRAD>), grid, dim3(256), 0, st, src, w, bias,
                           addend, dst, bn_part, gg, relu, mtiles, ntiles, fz);
    else
        hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, DGRAD, STRIDED, PW, false>), grid, dim3(256), 0, st, src, w, bias,
                           addend, dst, bn_part, gg, relu, mtiles, ntiles, fz);
}
template <int BM, int BN, int WM, int WN, bool DGRAD>
static void launch_igemm(const float* src, const float* w, const float* bias, const float* addend, float* dst,
                         double* bn_part, const ConvGeom& g, int relu, hipStream_t st, const BnBwdFuse& fz) {
    // pointwise fast path: 1x1 / stride 1 / pad 0 (row m == pixel m on both sides), block rows within 32-bit offsets
    const bool pw = g.R == 1 && g.S == 1 && g.stride == 1 && g.pad == 0 && g.H == g.Ho && g.W == g.Wo &&
                    (long)BM * g.C * 4 < (1L << 31);
    if (pw) launch_igemm_s<BM, BN, WM, WN, DGRAD, false, true>(src, w, bias, addend, dst, bn_part, g, relu, st, fz);
    else if (DGRAD && g.stride != 1) launch_igemm_s<BM, BN, WM, WN, DGRAD, true, false>(src, w, bias, addend, dst, bn_part, g, relu, st, fz);
    else launch_igemm_s<BM, BN, WM, WN, DGRAD, false, false>(src, w, bias, addend, dst, bn_part, g, relu, st, fz);
}

// ---- tile 4: persistent pointwise kernel with the weights resident in LDS (conv_pw_persist.h)
struct PwpPlan { int ks, bn, wg_per_col, waves_m; };
static bool pwp_geom_ok(const ConvGeom& g) {                 // (g as for the GEMM: Cg = K, Ng = N)
    return g.R == 1 && g.S == 1 && g.stride == 1 && g.pad == 0 && g.H == g.Ho && g.W == g.Wo && g.groups == 1 &&
           (g.Cg == 64 || g.Cg == 128 || g.Cg == 256) && g.Ng % 64 == 0 && g.M * g.Cg * 4 < (1L << 31) &&
           g.M * g.Ng * 4 < (1L << 31);
}
static PwpPlan pwp_plan(const ConvGeom& g) {
    PwpPlan p;
    p.ks = g.Cg / 64;
    const int cap = 256 / p.ks;                              // columns whose [K][BN] fp32 tile fits 64 KB
    // (128 columns even where 256 fit: 64 -> 256 at 56 x 56, batch 70: 110.6 -> 91.7 us, the same bits -- twice the workgroups
    //  per M range share the XCD's L2 copy of A; tools_dev/pwp_plan_bench.py.  256 stays reachable through the development override)
    p.bn = (cap >= 128 && g.Ng % 128 == 0) ? 128 : 64;
    int resident = 1;
    // development override (tools_dev/pwp_plan_bench.py): "bn,resident"
    if (const char* e = getenv("SCOUTER_PWP_PLAN_DEV")) {
        int bn = 0, r = 0;
        if (sscanf(e, "%d,%d", &bn, &r) == 2 && (bn == 64 || bn == 128 || bn == 256) && bn * p.ks <= 256 && g.Ng % bn == 0 && r >= 1 && r <= 4) {
            p.bn = bn;
            resident = r;
        }
    }
    const int colgroups = g.Ng / p.bn;
    int w = ((256 * resident) / colgroups) & ~7;
    p.wg_per_col = w < 8 ? 8 : w;
    p.waves_m = p.bn >= 256 ? 1 : 2;
    return p;
}
template <bool DGRAD>
static void launch_pwp(const float* src, const float* w, const float* addend, float* dst, double* bn_part, const ConvGeom& g,
                       hipStream_t st) {
    const PwpPlan p = pwp_plan(g);
    const int mtiles = sc_cdiv(g.M, 64), grid = p.wg_per_col * (g.Ng / p.bn);
    const size_t lds = (size_t)g.Cg * p.bn * 4 + 4 * 16384;
#define PWP(KS_, BN_)                                                                                              \
    do {                                                                                                           \
        if (DGRAD ? addend != nullptr : bn_part != nullptr) {                                                      \
            auto kern = pwp_kernel<KS_, BN_, DGRAD, true>;                                                         \
            static bool attr_set = false;                                                                          \
            if (!attr_set) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4 * 16384); attr_set = true; } \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, src, w, addend, dst, bn_part, g.M, g.Ng, mtiles, \
                               p.wg_per_col);                                                                      \
        } else {                                                                                                   \
            auto kern = pwp_kernel<KS_, BN_, DGRAD, false>;                                                        \
            static bool attr_set = false;                                                                          \
            if (!attr_set) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4 * 16384); attr_set = true; } \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, src, w, addend, dst, bn_part, g.M, g.Ng, mtiles, \
                               p.wg_per_col);                                                                      \
        }                                                                                                          \
    } while (0)
    if (p.ks == 1 && p.bn == 256) PWP(1, 256);
    else if (p.ks == 1 && p.bn == 128) PWP(1, 128);
    else if (p.ks == 1) PWP(1, 64);
    else if (p.ks == 2 && p.bn == 128) PWP(2, 128);
    else if (p.ks == 2) PWP(2, 64);
    else PWP(4, 64);
#undef PWP
}

// the input gradient with the fused BatchNorm-backward epilogue on the persistent skeleton (pwp_fused_kernel)
static PwpPlan pwp_fused_plan(const ConvGeom& g) {
    PwpPlan p;
    p.ks = g.Cg / 64;
    const int cap = p.ks == 4 ? 64 : 128;                    // 32 x 64 wave tiles at most (epilogue operands stay in registers)
    p.bn = (cap >= 128 && g.Ng % 128 == 0) ? 128 : 64;
    const int colgroups = g.Ng / p.bn;
    int w = (256 / colgroups) & ~7;
    p.wg_per_col = w < 8 ? 8 : w;
    p.waves_m = 2;
    return p;
}
static void launch_pwp_fused(const float* src, const float* w, const float* addend, float* dst, const ConvGeom& g,
                             hipStream_t st, const BnBwdFuse& fz) {
    const PwpPlan p = pwp_fused_plan(g);
    const int mtiles = sc_cdiv(g.M, 64), grid = p.wg_per_col * (g.Ng / p.bn);
    const size_t lds = (size_t)g.Cg * p.bn * 4 + 4 * 16384;
    const long mask_words = ((g.M * g.Ng / 4 + 63) / 64) * 4;
#define PWF(KS_, BN_, TWO_)                                                                                        \
    do {                                                                                                           \
        auto kern = pwp_fused_kernel<KS_, BN_, TWO_>;                                                              \
        static bool attr_set = false;                                                                              \
        if (!attr_set) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4 * 16384); attr_set = true; } \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, src, w, addend, dst, g.M, g.Ng, mtiles, p.wg_per_col, fz, \
                           mask_words);                                                                            \
    } while (0)
#define PWF2(KS_, BN_) do { if (fz.part2) PWF(KS_, BN_, true); else PWF(KS_, BN_, false); } while (0)
    if (p.ks == 1 && p.bn == 128) PWF2(1, 128);
    else if (p.ks == 1) PWF2(1, 64);
    else if (p.ks == 2 && p.bn == 128) PWF2(2, 128);
    else if (p.ks == 2) PWF2(2, 64);
    else PWF2(4, 64);
#undef PWF2
#undef PWF
}

// ---- tile 5: the persistent pointwise input gradient with the fused BatchNorm-backward epilogue on the bf16 matrix cores
// (three-way operand split in registers / in LDS; conv_pw_persist_x3.h)
struct XpwPlan { int ks, wg_per_col; };
static bool xpw_geom_ok(const ConvGeom& g) {                 // (g as for the GEMM: Cg = K, Ng = N)
    return g.R == 1 && g.S == 1 && g.stride == 1 && g.pad == 0 && g.H == g.Ho && g.W == g.Wo && g.groups == 1 &&
           (g.Cg == 64 || g.Cg == 128 || g.Cg == 256) && g.Ng % 64 == 0 && g.M * g.Cg < (1L << 29) &&
           (g.M + 128) * g.Ng < (1L << 29);          // (every streamed tensor < 2 GiB: the kernels clamp scalar row offsets at 2^31 - 1)
}
static XpwPlan xpw_plan(const ConvGeom& g) {
    XpwPlan p;
    p.ks = g.Cg / 64;
    const int resident = p.ks <= 2 ? 2 : 1;                  // workgroups per CU (LDS: 56 / 80 / 128 KB for K = 64 / 128 / 256)
    const int colgroups = g.Ng / 64;
    int w = ((256 * resident) / colgroups) & ~7;
    p.wg_per_col = w < 8 ? 8 : w;
    return p;
}
static void launch_xpw_fused(const float* src, const float* w, const float* addend, float* dst, const ConvGeom& g,
                             hipStream_t st, const BnBwdFuse& fz) {
    const XpwPlan p = xpw_plan(g);
    const int mtiles = sc_cdiv(g.M, 64), grid = p.wg_per_col * (g.Ng / 64);
    const size_t lds = (size_t)3 * 64 * g.Cg * 2 + 4 * 8192;
    const long mask_words = ((g.M * g.Ng / 4 + 63) / 64) * 4;
#define XPW(KS_, TWO_)                                                                                             \
    do {                                                                                                           \
        auto kern = xpw_fused_kernel<KS_, TWO_>;                                                                   \
        static bool attr_set = false;                                                                              \
        if (!attr_set) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 64 * 64 * KS_ * 2 + 4 * 8192); attr_set = true; } \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, src, w, addend, dst, g.M, g.Ng, mtiles, p.wg_per_col, fz, \
                           mask_words);                                                                            \
    } while (0)
#define XPW2(KS_) do { if (fz.part2) XPW(KS_, true); else XPW(KS_, false); } while (0)
    if (p.ks == 1) XPW2(1);
    else if (p.ks == 2) XPW2(2);
    else XPW2(4);
#undef XPW2
#undef XPW
}

static void launch_xpw_fwd(const float* src, const float* w, float* dst, double* bn_part, const ConvGeom& g, hipStream_t st) {
    const XpwPlan p = xpw_plan(g);
    const int mtiles = sc_cdiv(g.M, 64), grid = p.wg_per_col * (g.Ng / 64);
    const size_t lds = (size_t)3 * 64 * g.Cg * 2 + 4 * 8192;
#define XPF(KS_, ST_)                                                                                              \
    do {                                                                                                           \
        auto kern = xpw_fwd_kernel<KS_, ST_>;                                                                      \
        static bool attr_set = false;                                                                              \
        if (!attr_set) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 64 * 64 * KS_ * 2 + 4 * 8192); attr_set = true; } \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, src, w, dst, bn_part, g.M, g.Ng, mtiles, p.wg_per_col); \
    } while (0)
#define XPF2(KS_) do { if (bn_part) XPF(KS_, true); else XPF(KS_, false); } while (0)
    if (p.ks == 1) XPF2(1);
    else if (p.ks == 2) XPF2(2);
    else XPF2(4);
#undef XPF2
#undef XPF
}

// tile choice: largest tile that still fills the chip (>= ~1.5 rounds of 256 CUs x 2 blocks); Ng is a multiple of 32
static bool igemm_tile_ok(const ConvGeom& g, int t) {
    return (t == 0 && g.Ng % 128 == 0) || ((t == 1 || t == 2) && g.Ng % 64 == 0) || t == 3;
}
// hint >= 0: the caller's (autotuned) choice if legal for this shape; otherwise the static heuristic
static int igemm_tile(const ConvGeom& g, int hint = -1) {     // 0: 128x128  1: 128x64  2: 64x64  3: 128x32  4: persistent
    if (hint == 4 && pwp_geom_ok(g)) return 4;                //                                  5: persistent bf16x3 (fused dgrad)
    if (hint == 5 && xpw_geom_ok(g)) return 5;
    if (hint >= 0 && hint <= 3 && igemm_tile_ok(g, hint)) return hint;
    auto blocks = [&](int bm, int bn) { return (long)sc_cdiv(g.M, bm) * (g.Ng / bn) * g.groups; };
    const long want = 768;
    if (g.Ng % 128 == 0 && blocks(128, 128) >= want) return 0;
    if (g.Ng % 64 == 0 && blocks(128, 64) >= want) return 1;
    if (g.Ng % 64 == 0 && g.Ng >= 64) return 2;
    return 3;
}

template <bool DGRAD>
static int dispatch_igemm(const float* src, const float* w, const float* bias, const float* addend, float* dst,
                          double* bn_part, const ConvGeom& g, int relu, int tile, hipStream_t st,
                          const BnBwdFuse& fz = BnBwdFuse{}) {
    SC_UNSUPPORTED(g.M < (1L << 31) && (long)g.H * g.W * g.C < (1L << 28),
                   "conv2d: more than 2^31 output pixels or an image above 2^28 elements is not supported");
    if (tile == 4) {
        // (the caller named the persistent kernel: its partial-row layout differs, so an unsupported request is an error,
        // never a silent re-route)
        SC_UNSUPPORTED(pwp_geom_ok(g) && !bias && !relu && (DGRAD || (!addend && !fz.part1)),
                       "conv2d: tile 4 (persistent pointwise kernel) covers 1x1 / stride 1 / groups 1 convolutions with 64, 128 "
                       "or 256 GEMM-K channels, no bias / ReLU (forward: no addend)");
        if (DGRAD && fz.part1) launch_pwp_fused(src, w, addend, dst, g, st, fz);
        else launch_pwp<DGRAD>(src, w, addend, dst, bn_part, g, st);
        return sc_check_launch(DGRAD ? "conv2d_dgrad(persistent)" : "conv2d_fwd(persistent)");
    }
    if (tile == 5) {
        SC_UNSUPPORTED(xpw_geom_ok(g) && !bias && !relu && (DGRAD ? fz.part1 != nullptr : (!addend && !fz.part1)),
                       "conv2d: tile 5 (persistent bf16x3 kernel) covers 1x1 / stride 1 / groups 1 layers with GEMM-K of 64 / 128 / "
                       "256 and 64-multiples of output columns: the forward without bias / addend / ReLU, the input gradient with "
                       "the fused BatchNorm-backward epilogue");
        if constexpr (DGRAD) launch_xpw_fused(src, w, addend, dst, g, st, fz);
        else launch_xpw_fwd(src, w, dst, bn_part, g, st);
        return sc_check_launch(DGRAD ? "conv2d_dgrad(persistent bf16x3)" : "conv2d_fwd(persistent bf16x3)");
    }
    switch (tile) {
        case 0: launch_igemm<128, 128, 64, 64, DGRAD>(src, w, bias, addend, dst, bn_part, g, relu, st, fz); break;
        case 1: launch_igemm<128, 64, 64, 32, DGRAD>(src, w, bias, addend, dst, bn_part, g, relu, st, fz); break;
        case 2: launch_igemm<64, 64, 32, 32, DGRAD>(src, w, bias, addend, dst, bn_part, g, relu, st, fz); break;
        default: launch_igemm<128, 32, 32, 32, DGRAD>(src, w, bias, addend, dst, bn_part, g, relu, st, fz); break;
    }
    return sc_check_launch(DGRAD ? "conv2d_dgrad" : "conv2d_fwd");
}

static int conv_out(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }

static int conv_fwd_geom(ConvGeom& g, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad,
                         int groups) {
    if (!(groups > 0 && Cin % groups == 0 && Cout % groups == 0)) return SC_ERR_ARG;
    const int Cg = Cin / groups, Ng = Cout / groups;
    if (!(Cg % 32 == 0 && Ng % 32 == 0)) return SC_ERR_UNSUPPORTED;
    g = ConvGeom{B, H, W, Cin, conv_out(H, kh, stride, pad), conv_out(W, kw, stride, pad), Cout, kh, kw, stride, pad,
                 groups, Cg, Ng, 0, Cout, Cg * Cout};
    g.M = (long)B * g.Ho * g.Wo;
    return SC_OK;
}

// number of [Cout][2] fp64 rows the fused BatchNorm-statistics epilogue writes (= M tiles of the chosen kernel)
extern "C" int scouter_conv2d_fwd_bn_partial_rows(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride,
                                                  int pad, int groups, int tile_hint) {
    ConvGeom g;
    if (conv_fwd_geom(g, B, H, W, Cin, Cout, kh, kw, stride, pad, groups) != SC_OK) return 0;
    const int t = igemm_tile(g, tile_hint);
    if (t == 4) { const PwpPlan p = pwp_plan(g); return p.wg_per_col * p.waves_m; }    // one row per (workgroup, wave row)
    if (t == 5) return xpw_plan(g).wg_per_col;                                         // one row per workgroup row
    return sc_cdiv(g.M, t == 2 ? 64 : 128);
}

extern "C" int scouter_conv2d_fwd_f32(const float* x, const float* w, const float* bias, const float* addend,
                                      float* y, double* bn_partial, int B, int H, int W, int Cin, int Cout, int kh,
                                      int kw, int stride, int pad, int groups, int relu, int tile_hint, void* stream) {
    SC_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0, "conv2d_fwd: null pointer or empty shape");
    SC_REQUIRE(!(bn_partial && relu), "conv2d_fwd: fused BatchNorm statistics are taken before any activation");
    ConvGeom g;
    const int rc = conv_fwd_geom(g, B, H, W, Cin, Cout, kh, kw, stride, pad, groups);
    if (rc == SC_ERR_ARG) { sc_set_error("conv2d_fwd: channels not divisible by groups"); return rc; }
    if (rc == SC_ERR_UNSUPPORTED) {
        sc_set_error("conv2d_fwd: per-group channels must be multiples of 32 (got %d -> %d)", Cin / groups, Cout / groups);
        return rc;
    }
    static const char* names[6] = {"igemm_fwd<128x128>", "igemm_fwd<128x64>", "igemm_fwd<64x64>", "igemm_fwd<128x32>",
                                   "igemm_fwd<persistent>", "xpw_fwd<bf16x3>"};
    SC_UNSUPPORTED(tile_hint != 5 || xpw_geom_ok(g), "conv2d_fwd: tile 5 (persistent bf16x3 kernel) covers 1x1 / stride 1 / groups 1 "
                   "layers with 64 / 128 / 256 input channels and 64-multiples of output channels");
    const int tile = igemm_tile(g, tile_hint);
    ScProfScope prof(names[tile], (hipStream_t)stream, 2.0 * g.M * Cout * g.Cg * kh * kw,
                     4.0 * ((double)B * H * W * Cin + (double)g.M * Cout));
    return dispatch_igemm<false>(x, w, bias, addend, y, bn_partial, g, relu, tile, (hipStream_t)stream);
}

static ConvGeom dgrad_geom(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int groups) {
    const int Cig = Cin / groups, Cog = Cout / groups;
    const int Ho = conv_out(H, kh, stride, pad), Wo = conv_out(W, kw, stride, pad);
    // A operand = dY [B][Ho][Wo][Cout]; GEMM rows = input pixels (H x W); columns = Cin
    ConvGeom g{B, Ho, Wo, Cout, H, W, Cin, kh, kw, stride, pad, groups, Cog, Cig, 0, Cout, Cig * Cout};
    g.M = (long)B * H * W;
    return g;
}

// rows of the [rows][Cin][2] fp64 partials the fused BatchNorm-backward epilogue writes (= M tiles of the chosen kernel)
extern "C" int scouter_conv2d_dgrad_bn_partial_rows(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride,
                                                    int pad, int groups, int tile_hint) {
    if (!(groups > 0 && Cin % groups == 0 && Cout % groups == 0)) return 0;
    const ConvGeom g = dgrad_geom(B, H, W, Cin, Cout, kh, kw, stride, pad, groups);
    const int t = igemm_tile(g, tile_hint);
    if (t == 4) { const PwpPlan p = pwp_fused_plan(g); return p.wg_per_col * p.waves_m; }   // one row per (workgroup, wave row)
    if (t == 5) return xpw_plan(g).wg_per_col;                                              // one row per workgroup row
    return sc_cdiv(g.M, t == 2 ? 64 : 128);
}

extern "C" int scouter_conv2d_dgrad_bnbwd_f32(const float* dy, const float* w, const float* addend, float* dx, int B,
                                              int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad,
                                              int groups, int tile_hint, const void* relu_mask, const float* x1,
                                              const float* saved1, double* part1, const float* x2, const float* saved2,
                                              double* part2, void* stream) {
    SC_REQUIRE(dy && w && dx && B > 0, "conv2d_dgrad: null pointer or empty shape");
    SC_REQUIRE(groups > 0 && Cin % groups == 0 && Cout % groups == 0, "conv2d_dgrad: channels not divisible by groups");
    SC_REQUIRE(!part1 || (x1 && saved1), "conv2d_dgrad: fused BatchNorm backward needs x1 and saved1");
    SC_REQUIRE(!part2 || (part1 && x2 && saved2), "conv2d_dgrad: second fused BatchNorm needs the first, x2 and saved2");
    const int Cig = Cin / groups, Cog = Cout / groups;
    SC_UNSUPPORTED(Cig % 32 == 0 && Cog % 32 == 0, "conv2d_dgrad: per-group channels must be multiples of 32");
    const int Ho = conv_out(H, kh, stride, pad), Wo = conv_out(W, kw, stride, pad);
    const ConvGeom g = dgrad_geom(B, H, W, Cin, Cout, kh, kw, stride, pad, groups);
    static const char* names[5] = {"igemm_dgrad<128x128>", "igemm_dgrad<128x64>", "igemm_dgrad<64x64>",
                                   "igemm_dgrad<128x32>", "igemm_dgrad<persistent>"};
    // with the BatchNorm-backward reductions in the epilogue the kernel is a different piece of work (it also reads the
    // BatchNorm input(s), the addend and the ReLU bits -- co-bound by HBM on the short-K layers): its own profile row
    static const char* names_bn[6] = {"igemm_dgrad+bn_bwd<128x128>", "igemm_dgrad+bn_bwd<128x64>",
                                      "igemm_dgrad+bn_bwd<64x64>", "igemm_dgrad+bn_bwd<128x32>", "igemm_dgrad+bn_bwd<persistent>",
                                      "xpw_dgrad+bn_bwd<bf16x3>"};
    SC_UNSUPPORTED(tile_hint != 5 || (part1 && xpw_geom_ok(g)),
                   "conv2d_dgrad: tile 5 (persistent bf16x3 kernel) covers 1x1 / stride 1 / groups 1 input gradients with the fused "
                   "BatchNorm-backward epilogue, Cout of 64 / 128 / 256, 64-multiples of Cin");
    const int tile = igemm_tile(g, tile_hint);
    const double out_elems = (double)g.M * Cin;
    ScProfScope prof(part1 ? names_bn[tile] : names[tile], (hipStream_t)stream,
                     2.0 * g.M * Cin * Cog * kh * kw / (stride * stride),
                     4.0 * ((double)B * Ho * Wo * Cout + out_elems) + (addend ? 4.0 * out_elems : 0.0) +
                         (part1 ? (part2 ? 8.0 : 4.0) * out_elems + out_elems / 8 : 0.0));
    const BnBwdFuse fz{part1 ? (const unsigned long long*)relu_mask : nullptr, x1, saved1, part1, x2, saved2, part2};
    return dispatch_igemm<true>(dy, w, nullptr, addend, dx, nullptr, g, 0, tile, (hipStream_t)stream, fz);
}

extern "C" int scouter_conv2d_dgrad_f32(const float* dy, const float* w, const float* addend, float* dx, int B, int H,
                                        int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int groups,
                                        int tile_hint, void* stream) {
    return scouter_conv2d_dgrad_bnbwd_f32(dy, w, addend, dx, B, H, W, Cin, Cout, kh, kw, stride, pad, groups, tile_hint,
                                          nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

struct WgradPlan { int bm, bn, ci_tiles, co_tiles, splits; long pix_per_split, tiles; };

// plan_hint < 0: the static plan below.  Otherwise (autotuned by the caller, include/scouter_hip.h) bits 0-1 select the
// block budget {512, 1024, 2048, 4096} that sets the split count and bits 4-5 halve the ci / co tile edge -- smaller
// output tiles give more parallelism without split-K slabs, which pays for the short-K (7x7, 14x14) layers.
static WgradPlan wgrad_plan(const ConvGeom& g, int plan_hint, bool chunk64 = false, int min_tile = 32) {
    WgradPlan p;
    p.bm = (g.Cg % 128 == 0) ? 128 : (g.Cg % 64 == 0 ? 64 : 32);
    p.bn = (g.Ng % 128 == 0) ? 128 : (g.Ng % 64 == 0 ? 64 : 32);
    long budget = 1024;
    if (plan_hint >= 0) {
        budget = 512L << (plan_hint & 3);
        if ((plan_hint & 16) && p.bm > 32) p.bm >>= 1;
        if ((plan_hint & 32) && p.bn > 32) p.bn >>= 1;
    }
    if (p.bm < min_tile) p.bm = min_tile;                 // (bf16 kernel: ragged 64-wide tiles over 32-channel groups)
    if (p.bn < min_tile) p.bn = min_tile;
    p.ci_tiles = sc_cdiv(g.Cg, p.bm);
    p.co_tiles = sc_cdiv(g.Ng, p.bn);
    p.tiles = (long)p.ci_tiles * p.co_tiles * g.groups * g.R * g.S;
    // enough blocks to fill 256 CUs x 2 blocks twice over, but few splits for big weight tensors: every split costs a
    // full-size slab write + read in the reduction kernel
    long want = budget / p.tiles;
    if (want < 1) want = 1;
    long chunks = (g.M + BK - 1) / BK;
    long cps = (chunks + want - 1) / want;               // K-chunks (of 32 pixels) per split
    if (cps < 8) cps = 8;                                 // at least 256 pixels per block
    if (chunk64 && (cps & 1)) ++cps;                      // bf16 kernel: whole 64-pixel chunks
    p.pix_per_split = cps * BK;
    p.splits = (int)((g.M + p.pix_per_split - 1) / p.pix_per_split);
    return p;
}

static ConvGeom wgrad_geom(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int groups) {
    const int Cg = Cin / groups, Ng = Cout / groups;
    ConvGeom g{B, H, W, Cin, conv_out(H, kh, stride, pad), conv_out(W, kw, stride, pad), Cout, kh, kw, stride, pad,
               groups, Cg, Ng, 0, Cout, Cg * Cout};
    g.M = (long)B * g.Ho * g.Wo;
    g.inv_hw = 1.0 / ((double)g.Ho * g.Wo);
    g.inv_wo = 1.0f / (float)g.Wo;
    g.inv_ho = 1.0f / (float)g.Ho;
    return g;
}

extern "C" size_t scouter_conv2d_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw,
                                                       int stride, int pad, int groups, int plan_hint) {
    if (groups <= 0 || Cin % groups || Cout % groups) return 0;
    ConvGeom g = wgrad_geom(B, H, W, Cin, Cout, kh, kw, stride, pad, groups);
    WgradPlan p = wgrad_plan(g, plan_hint);
    size_t need = p.splits <= 1 ? 0 : (size_t)p.splits * kh * kw * g.Cg * Cout * sizeof(float);
    if (plan_hint < 0 && kh == 3 && kw == 3 && stride == 1 && pad == 1 && g.Cg == 32) {     // tap-fused plan (see wgrad_f32)
        const long tiles = (long)(g.Ng / (g.Ng % 64 == 0 ? 64 : 32)) * groups;
        long want = 1024 / tiles;
        if (want < 1) want = 1;
        long chunks = (g.M + BK - 1) / BK, cps = (chunks + want - 1) / want;
        if (cps < 8) cps = 8;
        const long splits = (g.M + cps * BK - 1) / (cps * BK);
        const size_t nt = splits > 1 ? (size_t)splits * 9 * g.Cg * Cout * sizeof(float) : 0;
        if (nt > need) need = nt;
    }
    return need;
}

extern "C" int scouter_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin,
                                        int Cout, int kh, int kw, int stride, int pad, int groups, int plan_hint,
                                        void* ws, size_t ws_bytes, void* arrival, int arrival_slots, void* stream) {
    SC_REQUIRE(x && dy && dw && B > 0, "conv2d_wgrad: null pointer or empty shape");
    SC_REQUIRE(groups > 0 && Cin % groups == 0 && Cout % groups == 0, "conv2d_wgrad: channels not divisible by groups");
    ConvGeom g = wgrad_geom(B, H, W, Cin, Cout, kh, kw, stride, pad, groups);
    SC_UNSUPPORTED(g.Cg % 32 == 0 && g.Ng % 32 == 0, "conv2d_wgrad: per-group channels must be multiples of 32");
    // 3x3 / stride 1 / pad 1 with 32 input channels per group: tap-fused kernel (conv_wgrad_taps.h)
    if (plan_hint < 0 && kh == 3 && kw == 3 && stride == 1 && pad == 1 && g.Cg == 32 && BK / g.Wo + 1 < g.Ho &&
        g.M < (1L << 31) && (long)(2 * g.W + 2) * g.C * 4 < (1L << 31)) {
        const int bn = g.Ng % 64 == 0 ? 64 : 32;
        const int co_tiles = g.Ng / bn;
        const long tiles = (long)co_tiles * groups;
        // default (SCOUTER_XWT=0: off): the register-split bf16x3 kernel (64-pixel chunks, rows up to 112 pixels wide).  One workgroup per CU
        // (147 KB of LDS) and a five-unit prologue per workgroup: ONE round of long pixel ranges, not four of short ones
        const char* xe = getenv("SCOUTER_XWT");
        const bool xwt = !(xe && xe[0] == '0') && g.W <= 112 && g.H >= 2 && BK == 32;
        const char* xw = getenv("SCOUTER_XWT_WGS");
        long want = (xwt ? (xw ? atoi(xw) : 256) : 1024) / tiles;
        if (want < 1) want = 1;
        long chunks = (g.M + BK - 1) / BK, cps = (chunks + want - 1) / want;
        if (cps < 8) cps = 8;
        if (xwt) cps += cps & 1;
        const long pps = cps * BK;
        const int splits = (int)((g.M + pps - 1) / pps);
        const long slab_t = (long)9 * g.Cg * Cout;
        const size_t need_t = splits > 1 ? (size_t)splits * slab_t * sizeof(float) : 0;
        if (need_t <= ws_bytes && (!need_t || ws)) {
            hipStream_t st = (hipStream_t)stream;
            float* out = splits > 1 ? (float*)ws : dw;
            unsigned* arr = splits > 1 && arrival && tiles <= arrival_slots ? (unsigned*)arrival : nullptr;
            const size_t lds = (size_t)(9 * 32 * 32 + 32 * bn) * sizeof(float);
            if (xwt && pps % XWT_CH == 0) {
                // the same tile on the bf16 matrix cores, operands split in registers (conv_wgrad_taps_x3.h)
                const size_t xlds = xwgrad_taps_lds_bytes(bn);
                ScProfScope prof("xwgrad_taps<bf16x3>", st, 2.0 * g.M * Cout * g.Cg * 9,
                                 4.0 * ((double)B * H * W * Cin + (double)g.M * Cout));
                if (bn == 64) {
                    auto kern = xwgrad_taps_kernel<64>;
                    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)xlds);
                    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)splits), dim3(256), xlds, st, x, dy, out, g,
                                       co_tiles, pps, slab_t, dw, arr);
                } else {
                    auto kern = xwgrad_taps_kernel<32>;
                    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)xlds);
                    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)splits), dim3(256), xlds, st, x, dy, out, g,
                                       co_tiles, pps, slab_t, dw, arr);
                }
            } else {
                ScProfScope prof("wgrad_taps", st, 2.0 * g.M * Cout * g.Cg * 9,
                                 4.0 * ((double)B * H * W * Cin + (double)g.M * Cout));
                if (bn == 64) {
                    auto kern = wgrad_taps_kernel<64>;
                    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)splits), dim3(256), lds, st, x, dy, out, g,
                                       co_tiles, pps, slab_t, dw, arr);
                } else {
                    auto kern = wgrad_taps_kernel<32>;
                    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)splits), dim3(256), lds, st, x, dy, out, g,
                                       co_tiles, pps, slab_t, dw, arr);
                }
            }
            int rc = sc_check_launch("conv2d_wgrad_taps");
            if (rc) return rc;
            if (splits > 1 && !arr) {
                ScProfScope prof2("slab_reduce", st, 0, 4.0 * (double)slab_t * (splits + 1));
                hipLaunchKernelGGL(slab_reduce_kernel, dim3(sc_cdiv(slab_t / 4, 8)), dim3(256), 0, st, (const float*)ws,
                                   dw, slab_t, splits, slab_t);
                rc = sc_check_launch("conv2d_wgrad_taps_reduce");
            }
            return rc;
        }
    }
    WgradPlan p = wgrad_plan(g, plan_hint);
    const long slab = (long)kh * kw * g.Cg * Cout;
    const size_t need = p.splits > 1 ? (size_t)p.splits * slab * sizeof(float) : 0;
    if (need > ws_bytes || (need && !ws)) {
        sc_set_error("conv2d_wgrad: workspace too small (%zu < %zu bytes)", ws_bytes, need);
        return SC_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* out = p.splits > 1 ? (float*)ws : dw;
    unsigned* arr = p.splits > 1 && arrival && p.tiles <= arrival_slots ? (unsigned*)arrival : nullptr;
    dim3 grid((unsigned)p.tiles, (unsigned)p.splits);
    static char wname[9][24];
    static const int wb[9][2] = {{128, 128}, {128, 64}, {128, 32}, {64, 128}, {64, 64}, {64, 32}, {32, 128}, {32, 64}, {32, 32}};
    int wi = 8;
    for (int k = 0; k < 9; ++k) if (wb[k][0] == p.bm && wb[k][1] == p.bn) wi = k;
    if (!wname[wi][0]) snprintf(wname[wi], sizeof(wname[wi]), "wgrad<%dx%d>", p.bm, p.bn);
    int rc;
    {
    ScProfScope prof(wname[wi], st, 2.0 * g.M * Cout * g.Cg * kh * kw,
                     4.0 * ((double)B * H * W * Cin + (double)g.M * Cout));
    // "same" stride-1 convolutions take the scalar-addressed paths (see wgrad_kernel); the branch-free pixel walk of
    // mode 1 needs at most one row wrap and one image wrap per 32-pixel chunk
    const bool same = stride == 1 && g.Ho == H && g.Wo == W;
    const int mode = !same ? 0 : (kh == 1 && kw == 1 && pad == 0) ? 2 : (BK / g.Wo + 1 < g.Ho ? 1 : 0);
#define WG(BM_, BN_, WM_, WN_)                                                                                      \
    do {                                                                                                            \
        if (mode == 2)                                                                                              \
            hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, WM_, WN_, 2>), grid, dim3(256), 0, st, x, dy, out, g,       \
                               p.ci_tiles, p.co_tiles, p.pix_per_split, slab, dw, arr);                            \
        else if (mode == 1)                                                                                         \
            hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, WM_, WN_, 1>), grid, dim3(256), 0, st, x, dy, out, g,       \
                               p.ci_tiles, p.co_tiles, p.pix_per_split, slab, dw, arr);                            \
        else                                                                                                        \
            hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, WM_, WN_, 0>), grid, dim3(256), 0, st, x, dy, out, g,       \
                               p.ci_tiles, p.co_tiles, p.pix_per_split, slab, dw, arr);                            \
    } while (0)
    if (p.bm == 128 && p.bn == 128) WG(128, 128, 64, 64);
    else if (p.bm == 128 && p.bn == 64) WG(128, 64, 64, 32);
    else if (p.bm == 128 && p.bn == 32) WG(128, 32, 32, 32);
    else if (p.bm == 64 && p.bn == 128) WG(64, 128, 32, 64);
    else if (p.bm == 64 && p.bn == 64) WG(64, 64, 32, 32);
    else if (p.bm == 64 && p.bn == 32) WG(64, 32, 32, 32);
    else if (p.bm == 32 && p.bn == 128) WG(32, 128, 32, 32);
    else if (p.bm == 32 && p.bn == 64) WG(32, 64, 32, 32);
    else WG(32, 32, 32, 32);
#undef WG
    }
    rc = sc_check_launch("conv2d_wgrad");
    if (rc) return rc;
    if (p.splits > 1 && !arr) {
        const long n = slab;
        ScProfScope prof2("slab_reduce", st, 0, 4.0 * (double)slab * (p.splits + 1));
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(sc_cdiv(n / 4, 8)), dim3(256), 0, st, (const float*)ws, dw, n,
                           p.splits, slab);
        rc = sc_check_launch("conv2d_wgrad_reduce");
    }
    return rc;
}

// bf16 matrix inputs (conv_wgrad_bf16.h).  Supported: "same" stride-1 convolutions whose per-group channel counts are
// multiples of 32; SC_ERR_UNSUPPORTED tells the caller to use the fp32 kernel for that layer.  32-channel groups (the
// ResNeSt stem and its first grouped layer) run on RAGGED 64-wide tiles: the tile reads 64 channels from the group's first
// one -- the upper 32 are the neighbouring group's / pixel's values -- and the rows / columns past the group are dropped at
// the store (at bf16 matrix rates the doubled tile is free, and 4x faster than the fp32 kernel these layers used).
// Same workspace as scouter_conv2d_wgrad_f32.
// `io`: SC_IO_X_BF16 -- the activation x is stored as bf16, SC_IO_R_BF16 -- dy is
extern "C" int scouter_conv2d_wgrad_bf16_io(const void* x, const void* dy, float* dw, int B, int H, int W, int Cin,
                                            int Cout, int kh, int kw, int stride, int pad, int groups, int plan_hint,
                                            void* ws, size_t ws_bytes, void* arrival, int arrival_slots, int io,
                                            void* stream) {
    SC_REQUIRE(x && dy && dw && B > 0, "conv2d_wgrad_bf16: null pointer or empty shape");
    SC_REQUIRE((io & ~(SC_IO_X_BF16 | SC_IO_R_BF16)) == 0, "conv2d_wgrad_bf16: unsupported io bits %d (x and dy may be bf16)", io);
    const bool xb = (io & SC_IO_X_BF16) != 0, db = (io & SC_IO_R_BF16) != 0;
    SC_REQUIRE(groups > 0 && Cin % groups == 0 && Cout % groups == 0, "conv2d_wgrad_bf16: channels not divisible by groups");
    ConvGeom g = wgrad_geom(B, H, W, Cin, Cout, kh, kw, stride, pad, groups);
    const bool same = stride == 1 && g.Ho == H && g.Wo == W;
    const int mode = !same ? 0 : (kh == 1 && kw == 1 && pad == 0) ? 2 : (64 / g.Wo + 1 < g.Ho ? 1 : 0);
    SC_UNSUPPORTED(mode != 0 && g.Cg % 32 == 0 && g.Ng % 32 == 0 && g.M < (1L << 31),
                   "conv2d_wgrad_bf16: shape not covered by the bf16 kernel");
    // 3x3 / stride 1 / pad 1 with 32 input channels per group, both operands stored as bf16: tap-fused kernel (conv_wgrad_taps_bf16.h;
    // the library's own plan, SCOUTER_BWT=0: off) -- every X row fetched once per workgroup instead of once per tap
    if (plan_hint < 0 && xb && db && kh == 3 && kw == 3 && stride == 1 && pad == 1 && g.Cg == 32 && g.W <= 112 && g.H >= 2 &&
        BWT_CH / g.Wo + 1 < g.Ho && (g.M + 512) * (long)(g.C > g.N ? g.C : g.N) * 2 < (1L << 31)) {
        const char* be = getenv("SCOUTER_BWT");
        const int bn = g.Ng % 64 == 0 ? 64 : 32;
        const int co_tiles = g.Ng / bn;
        const long tiles = (long)co_tiles * groups;
        const char* bw = getenv("SCOUTER_BWT_WGS");
        long want = (bw ? atoi(bw) : 512) / tiles;                 // two workgroups per CU
        if (want < 1) want = 1;
        long chunks = (g.M + BWT_CH - 1) / BWT_CH, cps = (chunks + want - 1) / want;
        if (cps < 4) cps = 4;
        const long pps = cps * BWT_CH;
        const int splits = (int)((g.M + pps - 1) / pps);
        const long slab_t = (long)9 * g.Cg * Cout;
        const size_t need_t = splits > 1 ? (size_t)splits * slab_t * sizeof(float) : 0;
        if (!(be && be[0] == '0') && need_t <= ws_bytes && (!need_t || ws)) {
            hipStream_t st = (hipStream_t)stream;
            float* out = splits > 1 ? (float*)ws : dw;
            unsigned* arr = splits > 1 && arrival && tiles <= arrival_slots ? (unsigned*)arrival : nullptr;
            const size_t blds = bwgrad_taps_lds_bytes(bn);
            {
            ScProfScope prof("bwgrad_taps<bf16>", st, 2.0 * g.M * Cout * g.Cg * 9, 2.0 * ((double)B * H * W * Cin + (double)g.M * Cout));
            if (bn == 64) {
                auto kern = bwgrad_taps_kernel<64>;
                hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds);
                hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)splits), dim3(256), blds, st, (const unsigned short*)x,
                                   (const unsigned short*)dy, out, g, co_tiles, pps, slab_t, dw, arr);
            } else {
                auto kern = bwgrad_taps_kernel<32>;
                hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds);
                hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)splits), dim3(256), blds, st, (const unsigned short*)x,
                                   (const unsigned short*)dy, out, g, co_tiles, pps, slab_t, dw, arr);
            }
            }
            int rc = sc_check_launch("conv2d_wgrad_taps_bf16");
            if (rc) return rc;
            if (splits > 1 && !arr) {
                ScProfScope prof2("slab_reduce", st, 0, 4.0 * (double)slab_t * (splits + 1));
                hipLaunchKernelGGL(slab_reduce_kernel, dim3(sc_cdiv(slab_t / 4, 8)), dim3(256), 0, st, (const float*)ws,
                                   dw, slab_t, splits, slab_t);
                rc = sc_check_launch("conv2d_wgrad_taps_bf16_reduce");
            }
            return rc;
        }
    }
    const bool ragged = g.Cg % 64 != 0 || g.Ng % 64 != 0;
    WgradPlan p = wgrad_plan(g, plan_hint, true, 64);                        // the bf16 kernel has no 32-wide tiles
    const long slab = (long)kh * kw * g.Cg * Cout;
    const size_t need = p.splits > 1 ? (size_t)p.splits * slab * sizeof(float) : 0;
    if (need > ws_bytes || (need && !ws)) {
        sc_set_error("conv2d_wgrad_bf16: workspace too small (%zu < %zu bytes)", ws_bytes, need);
        return SC_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* out = p.splits > 1 ? (float*)ws : dw;
    unsigned* arr = p.splits > 1 && arrival && p.tiles <= arrival_slots && !ragged ? (unsigned*)arrival : nullptr;
    dim3 grid((unsigned)p.tiles, (unsigned)p.splits);
    int rc;
    {
    ScProfScope prof("wgrad_bf16", st, 2.0 * g.M * Cout * g.Cg * kh * kw,
                     (xb ? 2.0 : 4.0) * B * H * W * Cin + (db ? 2.0 : 4.0) * g.M * Cout);
#define WGK(KERN)                                                                                                   \
    do {                                                                                                            \
        auto kern = KERN;                                                                                           \
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, dy, out, g, p.ci_tiles, p.co_tiles, p.pix_per_split,  \
                           slab, dw, arr);                                                                          \
    } while (0)
#define WGH(BM_, BN_, WM_, WN_)                                                                                     \
    do {                                                                                                            \
        const size_t lds = (size_t)2 * (BM_ + BN_) * 72 * 2;                                                        \
        if (mode == 2 && xb && db) WGK((wgrad_bf16_kernel<BM_, BN_, WM_, WN_, 2, true, true>));                     \
        else if (mode == 2 && xb) WGK((wgrad_bf16_kernel<BM_, BN_, WM_, WN_, 2, true, false>));                     \
        else if (mode == 2 && db) WGK((wgrad_bf16_kernel<BM_, BN_, WM_, WN_, 2, false, true>));                     \
        else if (mode == 2) WGK((wgrad_bf16_kernel<BM_, BN_, WM_, WN_, 2, false, false>));                          \
        else if (xb && db) WGK((wgrad_bf16_kernel<BM_, BN_, WM_, WN_, 1, true, true>));                             \
        else if (xb) WGK((wgrad_bf16_kernel<BM_, BN_, WM_, WN_, 1, true, false>));                                  \
        else if (db) WGK((wgrad_bf16_kernel<BM_, BN_, WM_, WN_, 1, false, true>));                                  \
        else WGK((wgrad_bf16_kernel<BM_, BN_, WM_, WN_, 1, false, false>));                                         \
    } while (0)
    if (p.bm == 128 && p.bn == 128) WGH(128, 128, 64, 64);
    else if (p.bm == 128 && p.bn == 64) WGH(128, 64, 64, 32);
    else if (p.bm == 64 && p.bn == 128) WGH(64, 128, 32, 64);
    else WGH(64, 64, 32, 32);
#undef WGH
#undef WGK
    }
    rc = sc_check_launch("conv2d_wgrad_bf16");
    if (rc) return rc;
    if (p.splits > 1 && !arr) {
        ScProfScope prof2("slab_reduce", st, 0, 4.0 * (double)slab * (p.splits + 1));
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(sc_cdiv(slab / 4, 8)), dim3(256), 0, st, (const float*)ws, dw, slab,
                           p.splits, slab);
        rc = sc_check_launch("conv2d_wgrad_bf16_reduce");
    }
    return rc;
}
extern "C" int scouter_conv2d_wgrad_bf16(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin,
                                         int Cout, int kh, int kw, int stride, int pad, int groups, int plan_hint,
                                         void* ws, size_t ws_bytes, void* arrival, int arrival_slots, void* stream) {
    return scouter_conv2d_wgrad_bf16_io(x, dy, dw, B, H, W, Cin, Cout, kh, kw, stride, pad, groups, plan_hint, ws, ws_bytes,
                                        arrival, arrival_slots, 0, stream);
}
