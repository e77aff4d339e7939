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
#include "common.h"

struct ConvGeom {
    int B, H, W, C;   // tensor feeding the A operand, NHWC, C = all channels
    int Ho, Wo;       // pixel grid of the GEMM rows
    int N;            // channels (all groups) of the GEMM-column tensor
    int R, S, stride, pad, groups;
    int Cg, Ng;       // per-group channels on the K side / the column side
    long M;           // B*Ho*Wo
    int wrow, wtap;   // weight strides (elements): fwd row=k -> N, tap -> Cg*N ; dgrad row=n -> C, tap -> Ng*C
};

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
template <int BM, int BN, int WM, int WN, bool DGRAD>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const float* __restrict__ src, const float* __restrict__ wgt,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ addend, float* __restrict__ dst,
                                                       ConvGeom g, int relu, int mtiles, int ntiles) {
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

    // ---- per-thread A rows (fixed for the whole K loop)
    int a_y[AI], a_x[AI];
    long a_base[AI];
    bool a_ok[AI];
    const int a_col = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const long m = m0 + (tid >> 3) + 32 * i;
        a_ok[i] = m < g.M;
        const long mm = a_ok[i] ? m : 0;
        const int hw = g.Ho * g.Wo;
        const int b = (int)(mm / hw), rem = (int)(mm % hw);
        const int y = rem / g.Wo, x = rem % g.Wo;
        a_base[i] = (long)b * g.H * g.W;
        if (DGRAD) { a_y[i] = y + g.pad; a_x[i] = x + g.pad; }
        else       { a_y[i] = y * g.stride - g.pad; a_x[i] = x * g.stride - g.pad; }
    }
    const float* wbase = wgt + (long)grp * (DGRAD ? g.Cg : g.Ng);   // group offset along the contiguous Cout axis

    f32x4 ra[AI], rb[BI];
    auto load_tile = [&](int kt) {
        const int tap = kt / cpt, c0 = (kt - tap * cpt) * BK;
        const int r = tap / g.S, q = tap - r * g.S;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            int iy, ix;
            bool ok = a_ok[i];
            if (DGRAD) {
                const int ty = a_y[i] - r, tx = a_x[i] - q;
                ok = ok && ty >= 0 && tx >= 0;
                if (g.stride == 1) { iy = ty; ix = tx; }
                else {
                    iy = ty / g.stride; ix = tx / g.stride;
                    ok = ok && (iy * g.stride == ty) && (ix * g.stride == tx);
                }
            } else { iy = a_y[i] + r; ix = a_x[i] + q; ok = ok && iy >= 0 && ix >= 0; }
            ok = ok && iy < g.H && ix < g.W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *(const f32x4*)(src + (a_base[i] + (long)iy * g.W + ix) * g.C + grp * g.Cg + c0 + a_col);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            if (B_KC) {   // rows = n (ci), contiguous k (co)
                const int n = n0 + (tid >> 3) + 32 * i;
                rb[i] = *(const f32x4*)(wbase + (long)tap * g.wtap + (long)n * g.wrow + c0 + a_col);
            } else {      // rows = k (ci), contiguous n (co)
                const int c = tid + 256 * i, krow = c / (BN / 4), col4 = c % (BN / 4);
                rb[i] = *(const f32x4*)(wbase + (long)tap * g.wtap + (long)(c0 + krow) * g.wrow + n0 + col4 * 4);
            }
        }
    };
    auto store_tile = [&](int buf) {
        float* As = lds + buf * T::STAGE;
        float* Bs = As + T::A_ELEMS;
#pragma unroll
        for (int i = 0; i < AI; ++i) *(f32x4*)(As + ((tid >> 3) + 32 * i) * LDK + a_col) = ra[i];
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

    auto compute = [&](int buf) {
        const float* As = lds + buf * T::STAGE;
        const float* Bs = As + T::A_ELEMS;
        f32x4 a[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) a[i][q] = *(const f32x4*)(As + (wm * WM + i * 32 + l31) * LDK + h * 16 + q * 4);
        f32x4 bk[NT][4];
        if (B_KC) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    bk[j][q] = *(const f32x4*)(Bs + (wn * WN + j * 32 + l31) * LDK + h * 16 + q * 4);
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float bv[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bv[j] = B_KC ? bk[j][s >> 2][s & 3] : Bs[(h * 16 + s) * BN + wn * WN + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma32(a[i][s >> 2][s & 3], bv[j], acc[i][j]);
        }
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if (more) load_tile(kt + 1);
        compute(kt & 1);
        if (more) store_tile((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: one 128-B row segment per (register, half-wave)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const long m = m0 + wm * WM + i * 32 + mfma32_row(e, lane);
            if (m >= g.M) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = grp * g.Ng + n0 + wn * WN + j * 32 + l31;
                float v = acc[i][j][e];
                if (bias) v += bias[n];
                if (addend) v += addend[m * g.N + n];
                if (relu) v = fmaxf(v, 0.f);
                dst[m * g.N + n] = v;
            }
        }
}

// ----------------------------------------------------------------------------------------------------------------
// wgrad: C[tap][ci][co] partial sums over a pixel range; both operands are "reduction-row, channel-contiguous"
// ----------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const float* __restrict__ act, const float* __restrict__ dy,
                                                       float* __restrict__ out, ConvGeom g, int ci_tiles,
                                                       int co_tiles, long pix_per_split, long slab) {
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
    int bid = blockIdx.x;
    const int co_t = bid % co_tiles; bid /= co_tiles;
    const int ci_t = bid % ci_tiles; bid /= ci_tiles;
    const int grp = bid % g.groups;
    const int tap = bid / g.groups;
    const int r = tap / g.S, q = tap - r * g.S;
    const int ci0 = ci_t * BM, co0 = co_t * BN;
    const long mbeg = (long)blockIdx.y * pix_per_split;
    long mend = mbeg + pix_per_split;
    if (mend > g.M) mend = g.M;
    const int KT = (int)((mend - mbeg + BK - 1) / BK);
    const int hw = g.Ho * g.Wo;

    f32x4 ra[AI], rb[BI];
    auto load_tile = [&](int kt) {
        const long mb = mbeg + (long)kt * BK;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int c = tid + 256 * i, prow = c / (BM / 4), col4 = c % (BM / 4);
            const long m = mb + prow;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < mend) {
                const int b = (int)(m / hw), rem = (int)(m % hw);
                const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
                const int iy = oy * g.stride - g.pad + r, ix = ox * g.stride - g.pad + q;
                if (iy >= 0 && ix >= 0 && iy < g.H && ix < g.W)
                    v = *(const f32x4*)(act + (((long)b * g.H + iy) * g.W + ix) * g.C + grp * g.Cg + ci0 + col4 * 4);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int c = tid + 256 * i, prow = c / (BN / 4), col4 = c % (BN / 4);
            const long m = mb + prow;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < mend) v = *(const f32x4*)(dy + m * g.N + grp * g.Ng + co0 + col4 * 4);
            rb[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
        float* As = lds + buf * STAGE;
        float* Bs = As + BK * BM;
#pragma unroll
        for (int i = 0; i < AI; ++i) *(f32x4*)(As + (tid + 256 * i) * 4) = ra[i];
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
    auto compute = [&](int buf) {
        const float* As = lds + buf * STAGE;
        const float* Bs = As + BK * BM;
#pragma unroll
        for (int ss = 0; ss < SPW; ++ss) {
            const int s = wk * SPW + ss;
            float av[MT], bv[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) av[i] = As[(h * 16 + s) * BM + wm * WM + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = Bs[(h * 16 + s) * BN + wn * WN + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma32(av[i], bv[j], acc[i][j]);
        }
    };
    if (KT > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if (more) load_tile(kt + 1);
        compute(kt & 1);
        if (more) store_tile((kt + 1) & 1);
        __syncthreads();
    }
    float* o = out + (long)blockIdx.y * slab + (long)tap * g.Cg * g.N;
    if (WK == 1) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ci = ci0 + wm * WM + i * 32 + mfma32_row(e, lane);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int co = grp * g.Ng + co0 + wn * WN + j * 32 + l31;
                    o[(long)ci * g.N + co] = acc[i][j][e];
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
            o[(long)(ci0 + e / BN) * g.N + grp * g.Ng + co0 + e % BN] = v;
        }
    }
}

// sums the split slabs in a fixed order (deterministic); optional accumulate into dst
__global__ void slab_reduce_kernel(const float* __restrict__ part, float* __restrict__ dst, long n, int splits,
                                   long slab) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 s = *(const f32x4*)(part + i);
    for (int k = 1; k < splits; ++k) s += *(const f32x4*)(part + (long)k * slab + i);
    *(f32x4*)(dst + i) = s;
}

// ----------------------------------------------------------------------------------------------------------------
// host dispatch
// ----------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool DGRAD>
static void launch_igemm(const float* src, const float* w, const float* bias, const float* addend, float* dst,
                         const ConvGeom& g, int relu, hipStream_t st) {
    const int mtiles = sc_cdiv(g.M, BM), ntiles = g.Ng / BN;
    dim3 grid(mtiles * ntiles * g.groups);
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, DGRAD>), grid, dim3(256), 0, st, src, w, bias, addend, dst, g,
                       relu, mtiles, ntiles);
}

template <bool DGRAD>
static int dispatch_igemm(const float* src, const float* w, const float* bias, const float* addend, float* dst,
                          const ConvGeom& g, int relu, hipStream_t st) {
    // largest tile that still fills the chip (>= ~2 waves of 256 CUs x 2 blocks); Ng is a multiple of 32
    auto blocks = [&](int bm, int bn) { return (long)sc_cdiv(g.M, bm) * (g.Ng / bn) * g.groups; };
    const long want = 768;
    if (g.Ng % 128 == 0 && blocks(128, 128) >= want) launch_igemm<128, 128, 64, 64, DGRAD>(src, w, bias, addend, dst, g, relu, st);
    else if (g.Ng % 64 == 0 && blocks(128, 64) >= want) launch_igemm<128, 64, 64, 32, DGRAD>(src, w, bias, addend, dst, g, relu, st);
    else if (g.Ng % 64 == 0 && g.Ng >= 64) launch_igemm<64, 64, 32, 32, DGRAD>(src, w, bias, addend, dst, g, relu, st);
    else launch_igemm<128, 32, 32, 32, DGRAD>(src, w, bias, addend, dst, g, relu, st);
    return sc_check_launch(DGRAD ? "conv2d_dgrad" : "conv2d_fwd");
}

static int conv_out(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }

extern "C" int scouter_conv2d_fwd_f32(const float* x, const float* w, const float* bias, const float* addend,
                                      float* y, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride,
                                      int pad, int groups, int relu, void* stream) {
    SC_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0, "conv2d_fwd: null pointer or empty shape");
    SC_REQUIRE(groups > 0 && Cin % groups == 0 && Cout % groups == 0, "conv2d_fwd: channels not divisible by groups");
    const int Cg = Cin / groups, Ng = Cout / groups;
    SC_UNSUPPORTED(Cg % 32 == 0 && Ng % 32 == 0,
                   "conv2d_fwd: per-group channels must be multiples of 32 (got %d -> %d); use scouter_conv2d_stem_*",
                   Cg, Ng);
    ConvGeom g{B, H, W, Cin, conv_out(H, kh, stride, pad), conv_out(W, kw, stride, pad), Cout, kh, kw, stride, pad,
               groups, Cg, Ng, 0, Cout, Cg * Cout};
    g.M = (long)B * g.Ho * g.Wo;
    ScProfScope prof(SC_PROF_CONV_FWD, (hipStream_t)stream, 2.0 * g.M * Cout * Cg * kh * kw,
                     4.0 * ((double)B * H * W * Cin + (double)g.M * Cout));
    return dispatch_igemm<false>(x, w, bias, addend, y, g, relu, (hipStream_t)stream);
}

extern "C" int scouter_conv2d_dgrad_f32(const float* dy, const float* w, const float* addend, float* dx, int B, int H,
                                        int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int groups,
                                        void* stream) {
    SC_REQUIRE(dy && w && dx && B > 0, "conv2d_dgrad: null pointer or empty shape");
    SC_REQUIRE(groups > 0 && Cin % groups == 0 && Cout % groups == 0, "conv2d_dgrad: channels not divisible by groups");
    const int Cig = Cin / groups, Cog = Cout / groups;
    SC_UNSUPPORTED(Cig % 32 == 0 && Cog % 32 == 0, "conv2d_dgrad: per-group channels must be multiples of 32");
    const int Ho = conv_out(H, kh, stride, pad), Wo = conv_out(W, kw, stride, pad);
    // A operand = dY [B][Ho][Wo][Cout]; GEMM rows = input pixels (H x W); columns = Cin
    ConvGeom g{B, Ho, Wo, Cout, H, W, Cin, kh, kw, stride, pad, groups, Cog, Cig, 0, Cout, Cig * Cout};
    g.M = (long)B * H * W;
    ScProfScope prof(SC_PROF_CONV_DGRAD, (hipStream_t)stream, 2.0 * g.M * Cin * Cog * kh * kw / (stride * stride),
                     4.0 * ((double)B * Ho * Wo * Cout + (double)g.M * Cin));
    return dispatch_igemm<true>(dy, w, nullptr, addend, dx, g, 0, (hipStream_t)stream);
}

struct WgradPlan { int bm, bn, ci_tiles, co_tiles, splits; long pix_per_split, tiles; };

static WgradPlan wgrad_plan(const ConvGeom& g) {
    WgradPlan p;
    p.bm = (g.Cg % 128 == 0) ? 128 : (g.Cg % 64 == 0 ? 64 : 32);
    p.bn = (g.Ng % 128 == 0) ? 128 : (g.Ng % 64 == 0 ? 64 : 32);
    if (p.bm == 128 && p.bn == 128 && (long)(g.Cg / 128) * (g.Ng / 128) * g.R * g.S * g.groups < 64) p.bm = 64;
    p.ci_tiles = g.Cg / p.bm;
    p.co_tiles = g.Ng / p.bn;
    p.tiles = (long)p.ci_tiles * p.co_tiles * g.groups * g.R * g.S;
    // enough blocks to fill 256 CUs x 2 blocks twice over, but few splits for big weight tensors: every split costs a
    // full-size slab write + read in the reduction kernel
    long want = 1024 / p.tiles;
    if (want < 1) want = 1;
    long chunks = (g.M + BK - 1) / BK;
    long cps = (chunks + want - 1) / want;               // K-chunks (of 32 pixels) per split
    if (cps < 8) cps = 8;                                 // at least 256 pixels per block
    p.pix_per_split = cps * BK;
    p.splits = (int)((g.M + p.pix_per_split - 1) / p.pix_per_split);
    return p;
}

static ConvGeom wgrad_geom(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int groups) {
    const int Cg = Cin / groups, Ng = Cout / groups;
    ConvGeom g{B, H, W, Cin, conv_out(H, kh, stride, pad), conv_out(W, kw, stride, pad), Cout, kh, kw, stride, pad,
               groups, Cg, Ng, 0, Cout, Cg * Cout};
    g.M = (long)B * g.Ho * g.Wo;
    return g;
}

extern "C" size_t scouter_conv2d_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw,
                                                       int stride, int pad, int groups) {
    if (groups <= 0 || Cin % groups || Cout % groups) return 0;
    ConvGeom g = wgrad_geom(B, H, W, Cin, Cout, kh, kw, stride, pad, groups);
    WgradPlan p = wgrad_plan(g);
    if (p.splits <= 1) return 0;
    return (size_t)p.splits * kh * kw * g.Cg * Cout * sizeof(float);
}

extern "C" int scouter_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin,
                                        int Cout, int kh, int kw, int stride, int pad, int groups, void* ws,
                                        size_t ws_bytes, void* stream) {
    SC_REQUIRE(x && dy && dw && B > 0, "conv2d_wgrad: null pointer or empty shape");
    SC_REQUIRE(groups > 0 && Cin % groups == 0 && Cout % groups == 0, "conv2d_wgrad: channels not divisible by groups");
    ConvGeom g = wgrad_geom(B, H, W, Cin, Cout, kh, kw, stride, pad, groups);
    SC_UNSUPPORTED(g.Cg % 32 == 0 && g.Ng % 32 == 0, "conv2d_wgrad: per-group channels must be multiples of 32");
    WgradPlan p = wgrad_plan(g);
    const long slab = (long)kh * kw * g.Cg * Cout;
    const size_t need = p.splits > 1 ? (size_t)p.splits * slab * sizeof(float) : 0;
    if (need > ws_bytes || (need && !ws)) {
        sc_set_error("conv2d_wgrad: workspace too small (%zu < %zu bytes)", ws_bytes, need);
        return SC_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    ScProfScope prof(SC_PROF_CONV_WGRAD, st, 2.0 * g.M * Cout * g.Cg * kh * kw,
                     4.0 * ((double)B * H * W * Cin + (double)g.M * Cout));
    float* out = p.splits > 1 ? (float*)ws : dw;
    dim3 grid((unsigned)p.tiles, (unsigned)p.splits);
#define WG(BM_, BN_, WM_, WN_)                                                                                      \
    hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, WM_, WN_>), grid, dim3(256), 0, st, x, dy, out, g, p.ci_tiles,     \
                       p.co_tiles, p.pix_per_split, slab)
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
    int rc = sc_check_launch("conv2d_wgrad");
    if (rc) return rc;
    if (p.splits > 1) {
        const long n = slab;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(sc_cdiv(n / 4, 256)), dim3(256), 0, st, (const float*)ws, dw, n,
                           p.splits, slab);
        rc = sc_check_launch("conv2d_wgrad_reduce");
    }
    return rc;
}
