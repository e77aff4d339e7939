// Train-mode BatchNorm (+ReLU, +residual) forward/backward, column sums, pooling, layout transposes.
// NHWC activations viewed as [M = B*H*W][C]; every kernel is HBM-bound: float4 accesses, lanes along C
// (coalesced), batch statistics accumulated in fp64 (E[x^2]-E[x]^2 is then safe), deterministic two-level
// reductions (per-block partials -> finalize) instead of atomics.
// Replaces nn.BatchNorm2d / ReLU / MaxPool2d / AvgPool2d of the reference backbone (timm/models/resnet.py:383,
// 404-412, 292-306; resnest.py:93,128-129) and their autograd backward.
#include "common.h"

#define MAXB 1024   // max partial blocks of a column reduction

struct ColGeom { long M; int C; int tpr; int rpb; int cslab; };

__host__ __device__ static inline ColGeom col_geom(long M, int C) {
    ColGeom g;
    g.M = M; g.C = C;
    // channels handled by one blockIdx.y: the whole row when its float4 count divides 256, else 1024-wide slabs
    g.cslab = (C / 4 <= 256 && 256 % (C / 4) == 0) ? C : 1024;
    g.tpr = g.cslab / 4;                   // threads per row (one float4 each)
    g.rpb = 256 / g.tpr;                   // rows per block pass
    return g;
}

static int col_blocks(long M, const ColGeom& g) {
    long nb = (M + (long)g.rpb * 8 - 1) / ((long)g.rpb * 8);   // >= 8 row passes per block
    if (nb > MAXB) nb = MAXB;
    if (nb < 1) nb = 1;
    return (int)nb;
}

// (ReLU sign bitmask: relu_mask_apply, common.h)

// Per-block partial column sums of up to two quantities produced by `F(row, col4) -> (float4 u, float4 v)`.
// partial layout: [block][C][2] doubles.
template <int MODE, bool XB = false, bool DB = false>   // 0: (x, x*x)   1: (g, g*xhat) with g = dy*(mask>0)   2: (x, 0)   3: (a*b, 0)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                             const float* __restrict__ p2,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd,
                                                             const unsigned long long* __restrict__ mbits,
                                                             double* __restrict__ part, ColGeom g,
                                                             float* __restrict__ direct_out = nullptr,
                                                             float direct_alpha = 1.f) {
    __shared__ double red[256 * 8];
    const int tid = threadIdx.x;
    const int cq = tid % g.tpr, rl = tid / g.tpr;
    const int c = blockIdx.y * g.cslab + cq * 4;
    double s[4] = {0, 0, 0, 0}, t[4] = {0, 0, 0, 0};
    if (rl < g.rpb && c < g.C) {
        f32x4 mu = {0, 0, 0, 0}, rs = {1, 1, 1, 1};
        if (MODE == 1) { mu = *(const f32x4*)(mean + c); rs = *(const f32x4*)(rstd + c); }
        const long rstep = (long)gridDim.x * g.rpb;
        long r = (long)blockIdx.x * g.rpb + rl;
        if (MODE == 0 || MODE == 2) {          // single-operand passes: 4 independent rows in flight per thread
            for (; r + 3 * rstep < g.M; r += 4 * rstep) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *(const f32x4*)(p0 + (r + u * rstep) * g.C + c);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        s[k] += v[u][k];
                        if (MODE == 0) t[k] += (double)v[u][k] * v[u][k];
                    }
            }
        }
        if (MODE == 1) {                       // four rows' loads in flight (one per round trip left this pass latency-bound)
            for (; r + 3 * rstep < g.M; r += 4 * rstep) {
                f32x4 av[4], xv[4], yv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long off = (r + u * rstep) * g.C + c;
                    av[u] = sc_load4<DB>(p0, off);
                    xv[u] = sc_load4<XB>(p2, off);
                    if (!mbits && p1) yv[u] = *(const f32x4*)(p1 + off);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long off = (r + u * rstep) * g.C + c;
                    f32x4 a = av[u];
                    if (mbits) {
                        relu_mask_apply(a, mbits, off >> 2);
                    } else if (p1) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) a[k] = yv[u][k] > 0.f ? a[k] : 0.f;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) { s[k] += a[k]; t[k] += (double)a[k] * ((xv[u][k] - mu[k]) * rs[k]); }
                }
            }
        }
        for (; r < g.M; r += rstep) {
            const long off = r * g.C + c;
            f32x4 a = sc_load4<(MODE == 1 && DB)>(p0, off);            // (DB: dy of MODE 1 is stored as bf16)
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { s[k] += a[k]; t[k] += (double)a[k] * a[k]; }
            } else if (MODE == 1) {
                if (mbits) {
                    relu_mask_apply(a, mbits, off >> 2);
                } else if (p1) {
                    f32x4 y = *(const f32x4*)(p1 + off);
#pragma unroll
                    for (int k = 0; k < 4; ++k) a[k] = y[k] > 0.f ? a[k] : 0.f;
                }
                const f32x4 x = sc_load4<XB>(p2, off);             // (XB: the BatchNorm input is stored as bf16)
#pragma unroll
                for (int k = 0; k < 4; ++k) { s[k] += a[k]; t[k] += (double)a[k] * ((x[k] - mu[k]) * rs[k]); }
            } else if (MODE == 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] += a[k];
            } else {
                f32x4 b = *(const f32x4*)(p1 + off);
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] += (double)a[k] * b[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[tid * 8 + k] = s[k]; red[tid * 8 + 4 + k] = t[k]; }
    __syncthreads();
    if (rl == 0 && c < g.C) {
        for (int j = 1; j < g.rpb; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) { s[k] += red[(j * g.tpr + cq) * 8 + k]; t[k] += red[(j * g.tpr + cq) * 8 + 4 + k]; }
        if (direct_out) {            // single row-block launch (small M): this IS the final sum, no finalize launch
#pragma unroll
            for (int k = 0; k < 4; ++k) direct_out[c + k] = (float)(s[k] * (double)direct_alpha);
            return;
        }
        double* o = part + ((long)blockIdx.x * g.C + c) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[k * 2] = s[k]; o[k * 2 + 1] = t[k]; }
    }
}

// Sums the per-block partials of FIN_CH channels per workgroup: FIN_LANES lanes split the nb partials (4 loads in flight
// each), fixed-order combine through LDS.
// FIN_CH = 1 (256 lanes per channel) for the narrow layers with MANY partial rows (stem / layer1: 32-64 channels, up to
// 3 430 rows): with four channels per workgroup those finalizes ran on 8-16 workgroups whose lanes each walked 54 rows
// (13 us); on this step's critical path the length of that dependent chain is the cost.
template <int FIN_CH>
__device__ __forceinline__ void partial_reduce(const double* __restrict__ part, int nb, int C, int c, int bl,
                                               double& s, double& t, double (*red)[FIN_CH][2]) {
    constexpr int FIN_LANES = 256 / FIN_CH;
    double s4[4] = {0, 0, 0, 0}, t4[4] = {0, 0, 0, 0};
    if (c < C) {
        int b = bl;
        for (; b + 3 * FIN_LANES < nb; b += 4 * FIN_LANES) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double2 v = *(const double2*)(part + ((long)(b + u * FIN_LANES) * C + c) * 2);
                s4[u] += v.x; t4[u] += v.y;
            }
        }
        for (; b < nb; b += FIN_LANES) {
            const double2 v = *(const double2*)(part + ((long)b * C + c) * 2);
            s4[0] += v.x; t4[0] += v.y;
        }
    }
    s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    t = (t4[0] + t4[1]) + (t4[2] + t4[3]);
    red[bl][threadIdx.x % FIN_CH][0] = s;
    red[bl][threadIdx.x % FIN_CH][1] = t;
    __syncthreads();
    // fixed-order tree over the lanes (these kernels are pure latency: a serial 128-term chain was most of their time)
    for (int w = FIN_LANES / 2; w >= 1; w >>= 1) {
        if (bl < w) {
            red[bl][threadIdx.x % FIN_CH][0] += red[bl + w][threadIdx.x % FIN_CH][0];
            red[bl][threadIdx.x % FIN_CH][1] += red[bl + w][threadIdx.x % FIN_CH][1];
        }
        __syncthreads();
    }
    s = red[0][threadIdx.x % FIN_CH][0];
    t = red[0][threadIdx.x % FIN_CH][1];
}

template <int FIN_CH>
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const double* __restrict__ part, int nb, long M, int C,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                                         float eps, int training, float* __restrict__ mean_o,
                                         float* __restrict__ rstd_o, float* __restrict__ scale_o,
                                         float* __restrict__ shift_o) {
    constexpr int FIN_LANES = 256 / FIN_CH;
    __shared__ double red[FIN_LANES][FIN_CH][2];
    const int c = blockIdx.x * FIN_CH + (threadIdx.x % FIN_CH), bl = threadIdx.x / FIN_CH;
    double mean, var, s = 0, t = 0;
    if (training) partial_reduce<FIN_CH>(part, nb, C, c, bl, s, t, red);
    if (bl != 0 || c >= C) return;
    if (training) {
        mean = s / (double)M;
        var = t / (double)M - mean * mean;
        if (var < 0) var = 0;
        if (rmean) {
            const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
            rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean);
            rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
        }
    } else { mean = rmean[c]; var = rvar[c]; }
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = (gamma ? gamma[c] : 1.f) * rstd;
    mean_o[c] = (float)mean;
    rstd_o[c] = rstd;
    scale_o[c] = sc;
    shift_o[c] = beta ? beta[c] : 0.f;        // applied as (x - mean) * scale + beta: no cancellation when |mean| >> std
}

// HOIST: the launch's grid stride is a multiple of C / 4, so a thread's four channels never change and the per-channel
// parameters are loaded once (round 4: these passes issue 3-6 cached parameter loads per 2-3 streamed ones; the bf16
// instantiations were bound by that instruction stream, not by bytes)
template <bool XB = false, bool YB = false, bool RB = false, bool HOIST = false>     // storage types of x / y / res (bf16 or fp32)
__global__ __launch_bounds__(256) void scale_shift_act_kernel(const void* __restrict__ x,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              const void* __restrict__ res, void* __restrict__ y,
                                                              unsigned long long* __restrict__ mbits, long n4, int C,
                                                              int relu, unsigned short* __restrict__ planes,
                                                              int nplanes, const float* __restrict__ res_bn) {
    const long i_first = (long)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 a, b, mu, qm, qs, qb;
    auto params = [&](int c) {
        a = *(const f32x4*)(scale + c); b = *(const f32x4*)(shift + c); mu = *(const f32x4*)(mean + c);
        if (res && res_bn) { qm = *(const f32x4*)(res_bn + c); qs = *(const f32x4*)(res_bn + 2 * C + c); qb = *(const f32x4*)(res_bn + 3 * C + c); }
    };
    if (HOIST) params((int)((i_first * 4) % C));
    for (long i = i_first; i < n4; i += (long)gridDim.x * blockDim.x) {
        if (!HOIST) params((int)((i * 4) % C));
        f32x4 v = sc_load4_nt<XB>(x, i * 4);
        v = bn_affine(v, mu, a, b);
        if (res) {
            f32x4 r = sc_load4<RB>(res, i * 4);
            // the residual is the RAW output of the downsample convolution: its BatchNorm is applied here (saved block
            // [mean, rstd, scale, shift][C]) -- the same fma the stand-alone apply pass uses, so the sum is bit-identical
            if (res_bn) r = bn_affine(r, qm, qs, qb);
            v += r;
        }
        if (relu) {
            if (mbits) {        // i - lane is a multiple of 64 (256-thread blocks, grid stride a multiple of 256)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned long long bits = __ballot(v[k] > 0.f);
                    if ((threadIdx.x & 63) == 0) mbits[(i >> 6) * 4 + k] = bits;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        if (y) sc_store4<YB>(y, i * 4, v);                            // (NULL: the consumers read the planes only)
        if (planes) store_planes4(planes, n4 * 4, nplanes, i, v);     // operand planes of the consuming plane convolution
    }
}

// The same pass when x, the residual (if any) and y are ALL bf16-stored (bn3 + shortcut + ReLU of a stored bottleneck):
// EIGHT elements per thread, so that every access is 16 bytes again -- the 4-element typed instantiation moves 8 bytes per
// lane and access and reaches 3.5-4.0 TB/s where the fp32 pass reaches 4.9 (tools_dev/typed_pass_bench.py).  Thread i owns
// the float4 indices 2i, 2i+1: a wave covers 128 consecutive ones = two mask words per component, the even / odd ballots
// interleaved on the scalar unit.  Same arithmetic, same bits as the 4-element kernel.
// HOIST: the grid stride is a multiple of C / 8 (power-of-two channel counts), so a thread's eight channels never change:
// the per-channel parameters are loaded ONCE.  Without it the all-bf16 pass issues ~12 cached parameter loads per 48 bytes
// of streamed data and is bound by vector-memory INSTRUCTIONS, not bytes (3.5-4.5 TB/s; tools_dev/typed_pass_bench.py).
template <bool HOIST>
__global__ __launch_bounds__(256) void scale_shift_act_bf16x8_kernel(const void* __restrict__ x,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ scale,
                                                                     const float* __restrict__ shift,
                                                                     const void* __restrict__ res, void* __restrict__ y,
                                                                     unsigned long long* __restrict__ mbits, long n8, int C,
                                                                     int relu, const float* __restrict__ res_bn) {
    const long i_first = (long)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 pm[2], ps[2], pb[2], qm[2], qs[2], qb[2];
    auto params = [&](int c) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int cc = c + 4 * q;
            pm[q] = *(const f32x4*)(mean + cc); ps[q] = *(const f32x4*)(scale + cc); pb[q] = *(const f32x4*)(shift + cc);
            if (res_bn) {
                qm[q] = *(const f32x4*)(res_bn + cc); qs[q] = *(const f32x4*)(res_bn + 2 * C + cc);
                qb[q] = *(const f32x4*)(res_bn + 3 * C + cc);
            }
        }
    };
    if (HOIST) params((int)((i_first * 8) % C));
    for (long i = i_first; i < n8; i += (long)gridDim.x * blockDim.x) {
        if (!HOIST) params((int)((i * 8) % C));
        f32x4 v[2], r[2];
        sc_load8_bf16(x, i * 8, v[0], v[1]);
        if (res) sc_load8_bf16(res, i * 8, r[0], r[1]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            v[q] = bn_affine(v[q], pm[q], ps[q], pb[q]);
            if (res) {
                if (res_bn) r[q] = bn_affine(r[q], qm[q], qs[q], qb[q]);
                v[q] += r[q];
            }
        }
        if (relu) {
            if (mbits) {        // float4 index 2i + q; i - lane is a multiple of 64 -> words (i >> 5) * 4 + k and the next four
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned long long be = __ballot(v[0][k] > 0.f), bo = __ballot(v[1][k] > 0.f);
                    if ((threadIdx.x & 63) == 0) {
                        unsigned long long* w = mbits + ((2 * i) >> 6) * 4 + k;
                        w[0] = sc_interleave32(be, bo);
                        if (2 * i + 64 < 2 * n8) w[4] = sc_interleave32(be >> 32, bo >> 32);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[q][k] = fmaxf(v[q][k], 0.f);
        }
        sc_store8_bf16(y, i * 8, v[0], v[1]);
    }
}

template <int FIN_CH>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ part, int nb, long M, int C,
                                                              int training, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ c1,
                                                              float* __restrict__ c2) {
    constexpr int FIN_LANES = 256 / FIN_CH;
    __shared__ double red[FIN_LANES][FIN_CH][2];
    const int c = blockIdx.x * FIN_CH + (threadIdx.x % FIN_CH), bl = threadIdx.x / FIN_CH;
    double s, t;
    partial_reduce<FIN_CH>(part, nb, C, c, bl, s, t, red);
    if (bl != 0 || c >= C) return;
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)t;
    c1[c] = training ? (float)(s / (double)M) : 0.f;
    c2[c] = training ? (float)(t / (double)M) : 0.f;
}

// dx = gamma*rstd * (g - c1 - xhat*c2),  g = dy * (ymask > 0);  optionally also writes g (the residual-branch grad)
template <bool XB = false, bool YB = false, bool DB = false, bool HOIST = false>   // XB / YB / DB: x / dx / dy stored as bf16
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const void* __restrict__ dy, const float* __restrict__ ymask,
                                                           const void* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ scale, const float* __restrict__ c1,
                                                           const float* __restrict__ c2,
                                                           const unsigned long long* __restrict__ mbits,
                                                           void* __restrict__ dx, float* __restrict__ gout, long n4,
                                                           int C) {
    const long i_first = (long)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 mu, rs, sc, k1, k2;
    auto params = [&](int c) {          // (HOIST: see scale_shift_act_kernel)
        mu = *(const f32x4*)(mean + c); rs = *(const f32x4*)(rstd + c); sc = *(const f32x4*)(scale + c);
        k1 = *(const f32x4*)(c1 + c); k2 = *(const f32x4*)(c2 + c);
    };
    if (HOIST) params((int)((i_first * 4) % C));
    for (long i = i_first; i < n4; i += (long)gridDim.x * blockDim.x) {
        if (!HOIST) params((int)((i * 4) % C));
        f32x4 g = sc_load4_nt<DB>(dy, i * 4);
        if (mbits) {
            relu_mask_apply(g, mbits, i);
        } else if (ymask) {
            const f32x4 y = *(const f32x4*)(ymask + i * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = y[k] > 0.f ? g[k] : 0.f;
        }
        const f32x4 xv = sc_load4_nt<XB>(x, i * 4);
        const f32x4 xh = (xv - mu) * rs;
        sc_store4<YB>(dx, i * 4, sc * (g - k1 - xh * k2));
        if (gout) *(f32x4*)(gout + i * 4) = g;
    }
}

// all-bf16 storage (dy = the stored masked gradient, x, dx): eight elements per thread, 16-byte accesses (see
// scale_shift_act_bf16x8_kernel); no mask (a bf16-stored dy is already masked)
template <bool HOIST>                                        // (see scale_shift_act_bf16x8_kernel)
__global__ __launch_bounds__(256) void bn_bwd_apply_bf16x8_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ c1, const float* __restrict__ c2,
                                                                  void* __restrict__ dx, long n8, int C) {
    const long i_first = (long)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 mu[2], rs[2], sc[2], k1[2], k2[2];
    auto params = [&](int c) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int cc = c + 4 * q;
            mu[q] = *(const f32x4*)(mean + cc); rs[q] = *(const f32x4*)(rstd + cc); sc[q] = *(const f32x4*)(scale + cc);
            k1[q] = *(const f32x4*)(c1 + cc); k2[q] = *(const f32x4*)(c2 + cc);
        }
    };
    if (HOIST) params((int)((i_first * 8) % C));
    for (long i = i_first; i < n8; i += (long)gridDim.x * blockDim.x) {
        if (!HOIST) params((int)((i * 8) % C));
        f32x4 g[2], xv[2], o[2];
        sc_load8_bf16(dy, i * 8, g[0], g[1]);
        sc_load8_bf16(x, i * 8, xv[0], xv[1]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4 xh = (xv[q] - mu[q]) * rs[q];
            o[q] = sc[q] * (g[q] - k1[q] - xh * k2[q]);
        }
        sc_store8_bf16(dx, i * 8, o[0], o[1]);
    }
}

// BatchNorm backward of a SMALL tensor (few rows: the BatchNorm between the two 1x1 layers of the split attention acts
// on one pooled vector per image) in ONE launch: a workgroup per channel slab reduces sum g, sum g * xhat over all rows
// (fp64), then applies -- instead of three launches whose latency, not their work, is the cost.
__global__ __launch_bounds__(256) void bn_bwd_small_kernel(const float* __restrict__ dy, const float* __restrict__ ymask,
                                                           const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ scale,
                                                           const unsigned long long* __restrict__ mbits, int training,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ dx, float* __restrict__ gout, ColGeom g) {
    __shared__ double red[256 * 8];
    __shared__ float coef[256 * 8];
    const int tid = threadIdx.x;
    const int cq = tid % g.tpr, rl = tid / g.tpr;
    const int c = blockIdx.y * g.cslab + cq * 4;
    const bool live = rl < g.rpb && c < g.C;
    double s[4] = {0, 0, 0, 0}, t[4] = {0, 0, 0, 0};
    f32x4 mu = {0, 0, 0, 0}, rs = {1, 1, 1, 1}, sc = {1, 1, 1, 1};
    auto masked = [&](long off) {
        f32x4 a = *(const f32x4*)(dy + off);
        if (mbits) {
            relu_mask_apply(a, mbits, off >> 2);
        } else if (ymask) {
            const f32x4 y = *(const f32x4*)(ymask + off);
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = y[k] > 0.f ? a[k] : 0.f;
        }
        return a;
    };
    if (live) {
        mu = *(const f32x4*)(mean + c); rs = *(const f32x4*)(rstd + c); sc = *(const f32x4*)(scale + c);
        for (long r = rl; r < g.M; r += g.rpb) {
            const long off = r * g.C + c;
            const f32x4 a = masked(off), xv = *(const f32x4*)(x + off);
#pragma unroll
            for (int k = 0; k < 4; ++k) { s[k] += a[k]; t[k] += (double)a[k] * ((xv[k] - mu[k]) * rs[k]); }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[tid * 8 + k] = s[k]; red[tid * 8 + 4 + k] = t[k]; }
    __syncthreads();
    if (rl == 0 && c < g.C) {
        for (int j = 1; j < g.rpb; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) { s[k] += red[(j * g.tpr + cq) * 8 + k]; t[k] += red[(j * g.tpr + cq) * 8 + 4 + k]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (dbeta) dbeta[c + k] = (float)s[k];
            if (dgamma) dgamma[c + k] = (float)t[k];
            coef[cq * 8 + k] = training ? (float)(s[k] / (double)g.M) : 0.f;
            coef[cq * 8 + 4 + k] = training ? (float)(t[k] / (double)g.M) : 0.f;
        }
    }
    __syncthreads();
    if (live) {
        f32x4 k1, k2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { k1[k] = coef[cq * 8 + k]; k2[k] = coef[cq * 8 + 4 + k]; }
        for (long r = rl; r < g.M; r += g.rpb) {
            const long off = r * g.C + c;
            const f32x4 a = masked(off), xv = *(const f32x4*)(x + off);
            const f32x4 xh = (xv - mu) * rs;
            *(f32x4*)(dx + off) = sc * (a - k1 - xh * k2);
            if (gout) *(f32x4*)(gout + off) = a;
        }
    }
}

template <int FIN_CH>
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const double* __restrict__ part, int nb, int C,
                                                              float* __restrict__ out, float alpha) {
    constexpr int FIN_LANES = 256 / FIN_CH;
    __shared__ double red[FIN_LANES][FIN_CH][2];
    const int c = blockIdx.x * FIN_CH + (threadIdx.x % FIN_CH), bl = threadIdx.x / FIN_CH;
    double s, t;
    partial_reduce<FIN_CH>(part, nb, C, c, bl, s, t, red);
    if (bl != 0 || c >= C) return;
    out[c] = (float)(s * alpha);
}

// g = dy * (y > 0)   (ReLU backward where no BN follows, e.g. the head's conv1x1+ReLU)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                       float* __restrict__ dx, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 g = *(const f32x4*)(dy + i * 4);
        const f32x4 v = *(const f32x4*)(y + i * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = v[k] > 0.f ? g[k] : 0.f;
        *(f32x4*)(dx + i * 4) = g;
    }
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    float* __restrict__ y, float alpha, float beta, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 v = *(const f32x4*)(a + i * 4) * alpha;
        if (b) v += *(const f32x4*)(b + i * 4) * beta;
        *(f32x4*)(y + i * 4) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// pooling (NHWC)
// ---------------------------------------------------------------------------------------------------------------
struct PoolGeom { int B, H, W, C, Ho, Wo, k, stride, pad, count_pad; };

// max pool: first maximum in (ky,kx) scan order wins (PyTorch CPU semantics); argmax tap stored as uint8
// bn (optional): saved block [mean, rstd, scale, shift][C] of the BatchNorm in front of the pool -- the windows are taken
// over relu(bn(x)) evaluated on the fly (the stem's bn1 + act1 + maxpool, resnet.py:404-412: the 4-byte activation and
// its sign mask are never stored; same fma as the apply pass, so values and the first-maximum choice are identical)
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          unsigned char* __restrict__ arg, PoolGeom g,
                                                          const float* __restrict__ bn) {
    const int c4n = g.C / 4;
    const long n = (long)g.B * g.Ho * g.Wo * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        long p = i / c4n;
        const int ox = (int)(p % g.Wo); p /= g.Wo;
        const int oy = (int)(p % g.Ho);
        const int b = (int)(p / g.Ho);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        f32x4 mu = {0, 0, 0, 0}, sc = mu, sh = mu;
        if (bn) { mu = *(const f32x4*)(bn + c); sc = *(const f32x4*)(bn + 2 * g.C + c); sh = *(const f32x4*)(bn + 3 * g.C + c); }
        for (int ky = 0; ky < g.k; ++ky) {
            const int iy = oy * g.stride - g.pad + ky;
            if (iy < 0 || iy >= g.H) continue;
            for (int kx = 0; kx < g.k; ++kx) {
                const int ix = ox * g.stride - g.pad + kx;
                if (ix < 0 || ix >= g.W) continue;
                f32x4 v = *(const f32x4*)(x + (((long)b * g.H + iy) * g.W + ix) * g.C + c);
                if (bn) {
                    v = bn_affine(v, mu, sc, sh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = ky * g.k + kx; }
            }
        }
        const long o = (((long)b * g.Ho + oy) * g.Wo + ox) * g.C + c;
        *(f32x4*)(y + o) = best;
        if (arg) *(uchar4*)(arg + o) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy,
                                                          const unsigned char* __restrict__ arg,
                                                          float* __restrict__ dx, PoolGeom g) {
    const int c4n = g.C / 4;
    const long n = (long)g.B * g.H * g.W * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        long p = i / c4n;
        const int ix = (int)(p % g.W); p /= g.W;
        const int iy = (int)(p % g.H);
        const int b = (int)(p / g.H);
        f32x4 acc = {0, 0, 0, 0};
        for (int ky = 0; ky < g.k; ++ky) {
            const int ty = iy + g.pad - ky;
            if (ty < 0 || ty % g.stride) continue;
            const int oy = ty / g.stride;
            if (oy >= g.Ho) continue;
            for (int kx = 0; kx < g.k; ++kx) {
                const int tx = ix + g.pad - kx;
                if (tx < 0 || tx % g.stride) continue;
                const int ox = tx / g.stride;
                if (ox >= g.Wo) continue;
                const long o = (((long)b * g.Ho + oy) * g.Wo + ox) * g.C + c;
                const uchar4 a = *(const uchar4*)(arg + o);
                const f32x4 d = *(const f32x4*)(dy + o);
                const int tap = ky * g.k + kx;
                if (a.x == tap) acc[0] += d[0];
                if (a.y == tap) acc[1] += d[1];
                if (a.z == tap) acc[2] += d[2];
                if (a.w == tap) acc[3] += d[3];
            }
        }
        *(f32x4*)(dx + i * 4) = acc;
    }
}

__device__ __forceinline__ float avg_div(const PoolGeom& g, int oy, int ox) {
    // PyTorch avg_pool2d divisor: count_include_pad -> window clipped to the padded extent; else to the input
    int y0 = oy * g.stride - g.pad, x0 = ox * g.stride - g.pad;
    int y1 = min(y0 + g.k, g.H + g.pad), x1 = min(x0 + g.k, g.W + g.pad);
    if (g.count_pad) return (float)((y1 - y0) * (x1 - x0));
    y0 = max(y0, 0); x0 = max(x0, 0); y1 = min(y1, g.H); x1 = min(x1, g.W);
    return (float)((y1 - y0) * (x1 - x0));
}

template <bool XB = false>                                   // XB: the input is stored as bf16 (output fp32)
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const void* __restrict__ x, float* __restrict__ y,
                                                          PoolGeom g) {
    const int c4n = g.C / 4;
    const long n = (long)g.B * g.Ho * g.Wo * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        long p = i / c4n;
        const int ox = (int)(p % g.Wo); p /= g.Wo;
        const int oy = (int)(p % g.Ho);
        const int b = (int)(p / g.Ho);
        f32x4 acc = {0, 0, 0, 0};
        for (int ky = 0; ky < g.k; ++ky) {
            const int iy = oy * g.stride - g.pad + ky;
            if (iy < 0 || iy >= g.H) continue;
            for (int kx = 0; kx < g.k; ++kx) {
                const int ix = ox * g.stride - g.pad + kx;
                if (ix < 0 || ix >= g.W) continue;
                acc += sc_load4<XB>(x, (((long)b * g.H + iy) * g.W + ix) * g.C + c);
            }
        }
        *(f32x4*)(y + i * 4) = acc / avg_div(g, oy, ox);
    }
}

__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                          PoolGeom g) {
    const int c4n = g.C / 4;
    const long n = (long)g.B * g.H * g.W * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        long p = i / c4n;
        const int ix = (int)(p % g.W); p /= g.W;
        const int iy = (int)(p % g.H);
        const int b = (int)(p / g.H);
        f32x4 acc = {0, 0, 0, 0};
        for (int ky = 0; ky < g.k; ++ky) {
            const int ty = iy + g.pad - ky;
            if (ty < 0 || ty % g.stride) continue;
            const int oy = ty / g.stride;
            if (oy >= g.Ho) continue;
            for (int kx = 0; kx < g.k; ++kx) {
                const int tx = ix + g.pad - kx;
                if (tx < 0 || tx % g.stride) continue;
                const int ox = tx / g.stride;
                if (ox >= g.Wo) continue;
                acc += *(const f32x4*)(dy + (((long)b * g.Ho + oy) * g.Wo + ox) * g.C + c) / avg_div(g, oy, ox);
            }
        }
        *(f32x4*)(dx + i * 4) = acc;
    }
}

// Backward gathers for stride 1 / 2 (every pool of the three backbones): one workgroup row per input pixel row
// (blockIdx.y = b * H + iy: scalar), threads along (ix, channel quad), the contributing output windows enumerated
// directly -- oy in [ceil((iy + pad - k + 1) / S), floor((iy + pad) / S)] -- instead of testing k x k taps with a
// modulo and a division each.  The generic kernels above spent ~900 VALU instructions per 16-byte result on 64-bit
// index arithmetic, tap tests and 16 divisions (3.0 TB/s); these are memory-bound.  avg: dy * (1 / divisor), the
// reciprocal taken once per window (within 1 ulp of PyTorch's dy / divisor).
// bnx / bn / coef (optional, max pool): the pool input was relu(bn(bnx)) evaluated on the fly in the forward -- the gathered
// gradient is masked by [bn(bnx) > 0] and pushed through the BatchNorm backward in the same pass:
// dx = scale * (g - c1 - xhat * c2), coef = [c1 | c2][C] from bn_bwd_finalize_kernel.
template <int S, bool MAXP, bool YB = false>                 // YB: dx is stored as bf16
__global__ __launch_bounds__(256) void pool_bwd_rows_kernel(const float* __restrict__ dy,
                                                            const unsigned char* __restrict__ arg,
                                                            void* __restrict__ dx, PoolGeom g, float inv_c4n,
                                                            const float* __restrict__ bnx,
                                                            const float* __restrict__ bn,
                                                            const float* __restrict__ coef) {
    const int c4n = g.C / 4;
    const int xid = blockIdx.x * 256 + threadIdx.x;
    if (xid >= g.W * c4n) return;
    const int ix = (int)(((float)xid + 0.5f) * inv_c4n), cq = xid - ix * c4n;      // xid < 2^20: exact
    const int row = blockIdx.y, b = row / g.H, iy = row - b * g.H;
    const int ty = iy + g.pad, tx = ix + g.pad;
    const int oy_hi = min(ty / S, g.Ho - 1), ox_hi = min(tx / S, g.Wo - 1);
    const int oy_lo = max((ty - g.k + S) / S, 0), ox_lo = max((tx - g.k + S) / S, 0);    // ceil((t - k + 1) / S), t - k + S >= 0 or result clamps
    f32x4 acc = {0, 0, 0, 0};
    for (int oy = ty - g.k + S < 0 ? 0 : oy_lo; oy <= oy_hi; ++oy) {
        const int ky = ty - oy * S;
        for (int ox = tx - g.k + S < 0 ? 0 : ox_lo; ox <= ox_hi; ++ox) {
            const int kx = tx - ox * S;
            const long o = (((long)b * g.Ho + oy) * g.Wo + ox) * g.C + cq * 4;
            const f32x4 d = *(const f32x4*)(dy + o);
            if (MAXP) {
                const uchar4 a = *(const uchar4*)(arg + o);
                const int tap = ky * g.k + kx;
                if (a.x == tap) acc[0] += d[0];
                if (a.y == tap) acc[1] += d[1];
                if (a.z == tap) acc[2] += d[2];
                if (a.w == tap) acc[3] += d[3];
            } else {
                acc += d * (1.0f / avg_div(g, oy, ox));
            }
        }
    }
    const long off = ((long)row * g.W + ix) * g.C + cq * 4;
    if (MAXP && bn) {
        const int c = cq * 4;
        const f32x4 xv = *(const f32x4*)(bnx + off);
        const f32x4 mu = *(const f32x4*)(bn + c), rs = *(const f32x4*)(bn + g.C + c), sc = *(const f32x4*)(bn + 2 * g.C + c),
                    sh = *(const f32x4*)(bn + 3 * g.C + c);
        const f32x4 k1 = *(const f32x4*)(coef + c), k2 = *(const f32x4*)(coef + g.C + c);
        const f32x4 hv = bn_affine(xv, mu, sc, sh);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = hv[e] > 0.f ? acc[e] : 0.f;
        const f32x4 xh = (xv - mu) * rs;
        acc = sc * (acc - k1 - xh * k2);
    }
    sc_store4<YB>(dx, off, acc);
}

// Reduction pass of the fused BatchNorm + ReLU + max-pool backward: g is non-zero only at the arg-max positions, so
// sum g and sum g * xhat run over the POOLED grid (a quarter of the elements), gathering x at each window's arg-max.
__global__ __launch_bounds__(256) void bn_maxpool_bwd_partial_kernel(const float* __restrict__ dy,
                                                                     const unsigned char* __restrict__ arg,
                                                                     const float* __restrict__ x,
                                                                     const float* __restrict__ bn,
                                                                     double* __restrict__ part, PoolGeom g, ColGeom cg) {
    __shared__ double red[256 * 8];
    const int tid = threadIdx.x;
    const int cq = tid % cg.tpr, rl = tid / cg.tpr;
    const int c = blockIdx.y * cg.cslab + cq * 4;
    double s[4] = {0, 0, 0, 0}, t[4] = {0, 0, 0, 0};
    if (rl < cg.rpb && c < g.C) {
        const f32x4 mu = *(const f32x4*)(bn + c), rs = *(const f32x4*)(bn + g.C + c), sc = *(const f32x4*)(bn + 2 * g.C + c),
                    sh = *(const f32x4*)(bn + 3 * g.C + c);
        const long rstep = (long)gridDim.x * cg.rpb;
        const int hwo = g.Ho * g.Wo;
        for (long r = (long)blockIdx.x * cg.rpb + rl; r < cg.M; r += rstep) {
            const int b = (int)(r / hwo), rem = (int)(r - (long)b * hwo), oy = rem / g.Wo, ox = rem - oy * g.Wo;
            const f32x4 d = *(const f32x4*)(dy + r * g.C + c);
            const uchar4 a = *(const uchar4*)(arg + r * g.C + c);
            const int taps[4] = {a.x, a.y, a.z, a.w};
            f32x4 xv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ky = taps[e] / g.k, kx = taps[e] - ky * g.k;
                const int iy = oy * g.stride - g.pad + ky, ix = ox * g.stride - g.pad + kx;
                xv[e] = x[(((long)b * g.H + iy) * g.W + ix) * g.C + c + e];
            }
            const f32x4 hv = bn_affine(xv, mu, sc, sh);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gg = hv[e] > 0.f ? d[e] : 0.f;
                s[e] += gg;
                t[e] += (double)gg * ((xv[e] - mu[e]) * rs[e]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[tid * 8 + k] = s[k]; red[tid * 8 + 4 + k] = t[k]; }
    __syncthreads();
    if (rl == 0 && c < g.C) {
        for (int j = 1; j < cg.rpb; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) { s[k] += red[(j * cg.tpr + cq) * 8 + k]; t[k] += red[(j * cg.tpr + cq) * 8 + 4 + k]; }
        double* o = part + ((long)blockIdx.x * g.C + c) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[k * 2] = s[k]; o[k * 2 + 1] = t[k]; }
    }
}

static bool pool_bwd_rows_ok(const PoolGeom& g) {
    const long xw = (long)g.W * (g.C / 4), rows = (long)g.B * g.H;
    return (g.stride == 1 || g.stride == 2) && xw < (1 << 20) && rows <= 65535;
}
template <bool MAXP>
static bool pool_bwd_rows(const float* dy, const unsigned char* arg, void* dx, const PoolGeom& g, hipStream_t st,
                          const float* bnx = nullptr, const float* bn = nullptr, const float* coef = nullptr,
                          bool dx_bf16 = false) {
    const long xw = (long)g.W * (g.C / 4), rows = (long)g.B * g.H;
    if (!pool_bwd_rows_ok(g)) return false;
    dim3 grid((unsigned)((xw + 255) / 256), (unsigned)rows);
    const float inv = 1.0f / (float)(g.C / 4);
    if (dx_bf16 && g.stride == 2)
        hipLaunchKernelGGL((pool_bwd_rows_kernel<2, MAXP, true>), grid, dim3(256), 0, st, dy, arg, dx, g, inv, bnx, bn, coef);
    else if (dx_bf16)
        hipLaunchKernelGGL((pool_bwd_rows_kernel<1, MAXP, true>), grid, dim3(256), 0, st, dy, arg, dx, g, inv, bnx, bn, coef);
    else if (g.stride == 1)
        hipLaunchKernelGGL((pool_bwd_rows_kernel<1, MAXP>), grid, dim3(256), 0, st, dy, arg, dx, g, inv, bnx, bn, coef);
    else
        hipLaunchKernelGGL((pool_bwd_rows_kernel<2, MAXP>), grid, dim3(256), 0, st, dy, arg, dx, g, inv, bnx, bn, coef);
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// layout: NCHW <-> NHWC (32x32 LDS tile transpose of the [C][HW] plane of every image)
// ---------------------------------------------------------------------------------------------------------------
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    __shared__ float t[32][33];
    const long boff = (long)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        if (r < rows && c < cols) t[j][threadIdx.x] = in[boff + (long)r * cols + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (r < rows && c < cols) out[boff + (long)c * rows + r] = t[threadIdx.x][j];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host entry points
// ---------------------------------------------------------------------------------------------------------------
static inline bool fin_narrow(int C, int nparts) { return C <= 64 && nparts > 512; }   // see partial_reduce
static inline int ew_blocks(long n) { long b = (n + 255) / 256; return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }
// grid for an elementwise pass over n vectors of `cv` = C / 4 (or C / 8) channel groups whose threads keep their channels
// (HOIST instantiations): the grid stride blocks * 256 must be a multiple of cv; 0 if no such grid (odd channel counts)
static inline int ew_blocks_hoist(long n, int cv) {
    int b = ew_blocks(n);
    if (cv <= 0) return 0;
    if (256 % cv == 0) return b;
    if (cv % 256 == 0) {                   // wider rows than a workgroup: a multiple of cv / 256 workgroups
        const int m = cv / 256;
        if (b <= 1) return b;              // one workgroup: at most one iteration per thread
        b = (b + m - 1) / m * m;
        return b;
    }
    return 0;
}

extern "C" size_t scouter_colreduce_workspace_bytes(long M, int C) { (void)M; return (size_t)MAXB * C * 2 * sizeof(double); }

#define COL_CHECKS(name)                                                                             \
    SC_REQUIRE(M > 0 && C > 0 && C % 4 == 0, name ": bad shape M=%ld C=%d (C must be a multiple of 4)", M, C); \
    ColGeom g = col_geom(M, C);                                                                      \
    SC_REQUIRE(g.cslab % 4 == 0 && 256 % g.tpr == 0, name ": unsupported channel count %d", C);      \
    const int nb = col_blocks(M, g);                                                                 \
    if (!ws || ws_bytes < (size_t)nb * C * 2 * sizeof(double)) {                                     \
        sc_set_error(name ": workspace too small");                                                  \
        return SC_ERR_WORKSPACE;                                                                     \
    }                                                                                                \
    dim3 pgrid(nb, (C + g.cslab - 1) / g.cslab);

extern "C" size_t scouter_relu_mask_words(long n) { return (size_t)((n / 4 + 63) / 64) * 4; }

// `io` (SC_IO_X_BF16 | SC_IO_Y_BF16 | SC_IO_R_BF16): storage type of x / y / residual.  A bf16 x carries the values the
// apply pass normalises; its batch statistics must come from the producing convolution's fp32 accumulators (ext_partial)
extern "C" int scouter_bn_fwd_io(const void* x, void* y, const void* residual, long M, int C, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var, float momentum,
                                 float eps, int training, int relu, float* mean_out, float* rstd_out,
                                 float* scale_out, float* shift_out, const double* ext_partial, int ext_rows,
                                 unsigned long long* relu_mask_out, void* planes_out, int nplanes,
                                 const float* residual_bn_saved, int io, void* ws, size_t ws_bytes, void* stream) {
    SC_REQUIRE(x && mean_out && rstd_out && scale_out && shift_out, "bn_fwd: null pointer");   // y == NULL: statistics only
    SC_REQUIRE((io & ~7) == 0, "bn_fwd: unknown io bits %d", io);
    SC_UNSUPPORTED(!(io & SC_IO_X_BF16) || !training || ext_partial,
                   "bn_fwd: a bf16-stored input needs the batch statistics of its producer (ext_partial)");
    SC_REQUIRE(!residual_bn_saved || residual, "bn_fwd: residual_bn_saved without residual");
    SC_REQUIRE(!planes_out || nplanes == 1 || nplanes == 3, "bn_fwd: planes_out needs 1 or 3 planes");
    SC_REQUIRE(!relu_mask_out || relu, "bn_fwd: relu_mask_out without relu");
    SC_REQUIRE(training || (running_mean && running_var), "bn_fwd: eval mode needs running statistics");
    COL_CHECKS("bn_fwd")
    hipStream_t st = (hipStream_t)stream;
    ScProfScope prof("bn_fwd(stats+finalize+apply)", st, 0,
                     ((ext_partial ? 0.0 : 4.0) + ((io & SC_IO_X_BF16) ? 2.0 : 4.0) + ((io & SC_IO_Y_BF16) ? 2.0 : 4.0)) * M * C);
    const double* part = (const double*)ws;
    int nparts = nb;
    if (training && ext_partial) { part = ext_partial; nparts = ext_rows; }     // statistics came from the conv epilogue
    else if (training)
        hipLaunchKernelGGL(colsum_partial_kernel<0>, pgrid, dim3(256), 0, st, (const float*)x, nullptr, nullptr, nullptr,
                           nullptr, nullptr, (double*)ws, g);
    if (fin_narrow(C, nparts))
        hipLaunchKernelGGL(bn_stats_finalize_kernel<1>, dim3(C), dim3(256), 0, st, part, nparts, M, C, gamma, beta,
                           running_mean, running_var, momentum, eps, training, mean_out, rstd_out, scale_out, shift_out);
    else
        hipLaunchKernelGGL(bn_stats_finalize_kernel<4>, dim3(sc_cdiv(C, 4)), dim3(256), 0, st, part, nparts, M, C, gamma, beta,
                           running_mean, running_var, momentum, eps, training, mean_out, rstd_out, scale_out, shift_out);
    const long n4 = M * C / 4;
    if (y || planes_out) {     // neither: the consumer applies (x - mean) * scale + shift itself (fused split attention)
#define SSA(XB_, YB_, RB_)                                                                                            \
        hipLaunchKernelGGL((scale_shift_act_kernel<XB_, YB_, RB_>), dim3(ew_blocks(n4)), dim3(256), 0, st, x, mean_out, \
                           scale_out, shift_out, residual, y, relu_mask_out, n4, C, relu, (unsigned short*)planes_out, \
                           nplanes, residual_bn_saved)
        const bool all_bf16 = (io & SC_IO_X_BF16) && (io & SC_IO_Y_BF16) && y && (!residual || (io & SC_IO_R_BF16)) &&
                              !planes_out && C % 8 == 0;
        if (all_bf16) {
            const long n8 = n4 / 2;
            if (256 % (C / 8) == 0)         // (the grid stride, a multiple of 256, is then a multiple of C / 8: see HOIST)
                hipLaunchKernelGGL(scale_shift_act_bf16x8_kernel<true>, dim3(ew_blocks(n8)), dim3(256), 0, st, x, mean_out,
                                   scale_out, shift_out, residual, y, relu_mask_out, n8, C, relu, residual_bn_saved);
            else
                hipLaunchKernelGGL(scale_shift_act_bf16x8_kernel<false>, dim3(ew_blocks(n8)), dim3(256), 0, st, x, mean_out,
                                   scale_out, shift_out, residual, y, relu_mask_out, n8, C, relu, residual_bn_saved);
        } else
        {
        const int hb = ew_blocks_hoist(n4, C / 4);
#define SSH(XB_, YB_, RB_)                                                                                            \
        do {                                                                                                          \
            if (hb) hipLaunchKernelGGL((scale_shift_act_kernel<XB_, YB_, RB_, true>), dim3(hb), dim3(256), 0, st, x,    \
                                       mean_out, scale_out, shift_out, residual, y, relu_mask_out, n4, C, relu,       \
                                       (unsigned short*)planes_out, nplanes, residual_bn_saved);                      \
            else SSA(XB_, YB_, RB_);                                                                                  \
        } while (0)
        switch (io & 7) {
            case 0: SSH(false, false, false); break;
            case 1: SSH(true, false, false); break;
            case 2: SSH(false, true, false); break;
            case 3: SSH(true, true, false); break;
            case 4: SSH(false, false, true); break;
            case 5: SSH(true, false, true); break;
            case 6: SSH(false, true, true); break;
            default: SSH(true, true, true); break;
        }
#undef SSH
        }
#undef SSA
    }
    return sc_check_launch("bn_fwd");
}

extern "C" int scouter_bn_fwd_f32(const float* x, float* y, const float* residual, long M, int C, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var, float momentum,
                                  float eps, int training, int relu, float* mean_out, float* rstd_out,
                                  float* scale_out, float* shift_out, const double* ext_partial, int ext_rows,
                                  unsigned long long* relu_mask_out, void* planes_out, int nplanes,
                                  const float* residual_bn_saved, void* ws, size_t ws_bytes, void* stream) {
    return scouter_bn_fwd_io(x, y, residual, M, C, gamma, beta, running_mean, running_var, momentum, eps, training, relu,
                             mean_out, rstd_out, scale_out, shift_out, ext_partial, ext_rows, relu_mask_out, planes_out,
                             nplanes, residual_bn_saved, 0, ws, ws_bytes, stream);
}

// the apply pass alone: y = [relu]((x - mean) * scale + shift) from a saved [4][C] block (mean, rstd, scale, shift)
extern "C" int scouter_bn_apply_f32(const float* x, const float* bn_saved, float* y, long M, int C, int relu,
                                    void* stream) {
    SC_REQUIRE(x && bn_saved && y && M > 0 && C > 0 && C % 4 == 0, "bn_apply: bad arguments");
    const long n4 = M * C / 4;
    hipLaunchKernelGGL((scale_shift_act_kernel<false, false, false>), dim3(ew_blocks(n4)), dim3(256), 0,
                       (hipStream_t)stream, x, bn_saved, bn_saved + 2 * C, bn_saved + 3 * C, nullptr, y, nullptr, n4, C,
                       relu, nullptr, 0, nullptr);
    return sc_check_launch("bn_apply");
}

// `io`: SC_IO_X_BF16 -- the BatchNorm input x is stored as bf16; SC_IO_Y_BF16 -- dx is stored as bf16 (RNE; for a dx whose
// only readers are bf16-input convolution kernels, which round it the same way); SC_IO_R_BF16 -- dy is stored as bf16 (the
// masked block-output gradient written by scouter_conv2d_dgrad_bnbwd_bf16_io).  gout is fp32.
extern "C" int scouter_bn_bwd_io(const void* dy, const float* ymask, const void* x, const float* mean,
                                 const float* rstd, const float* scale, const unsigned long long* relu_mask, long M,
                                 int C, int training, float* dgamma, float* dbeta, void* dx, float* gout,
                                 const double* ext_partial, int ext_rows, int io, void* ws, size_t ws_bytes, void* stream) {
    SC_REQUIRE(dy && x && mean && rstd && scale && dx, "bn_bwd: null pointer");
    SC_REQUIRE((io & ~7) == 0, "bn_bwd: unknown io bits %d", io);
    const bool xb = (io & SC_IO_X_BF16) != 0, yb = (io & SC_IO_Y_BF16) != 0, db = (io & SC_IO_R_BF16) != 0;
    SC_UNSUPPORTED(!db || !gout, "bn_bwd: a bf16-stored dy is the already masked gradient (no separate gout)");
    SC_REQUIRE(!ext_partial || (ext_rows > 0 && !ymask && !relu_mask && !gout),
               "bn_bwd: with ext_partial dy is the already masked gradient (no ymask / relu_mask / gout)");
    COL_CHECKS("bn_bwd")
    const size_t coef_off = (size_t)nb * C * 2 * sizeof(double);
    if (ws_bytes < coef_off + 2 * (size_t)C * sizeof(float)) {
        sc_set_error("bn_bwd: workspace too small");
        return SC_ERR_WORKSPACE;
    }
    float* c1 = (float*)((char*)ws + coef_off);
    float* c2 = c1 + C;
    hipStream_t st = (hipStream_t)stream;
    ScProfScope prof(ext_partial ? "bn_bwd(finalize+apply)" : "bn_bwd(reduce+finalize+apply)", st, 0,
                     ((ext_partial ? 12.0 : (ymask && !relu_mask ? 28.0 : 20.0)) + (gout ? 4.0 : 0.0) +
                      (relu_mask ? 0.25 : 0.0) - (xb ? (ext_partial ? 2.0 : 4.0) : 0.0) - (yb ? 2.0 : 0.0) -
                      (db ? (ext_partial ? 2.0 : 4.0) : 0.0)) * M * C);
    if (!ext_partial && M <= (long)g.rpb * 24 && !xb && !yb && !db) {   // few row passes: everything in one launch
        // 16-channel slabs (4 threads per row, 64 rows per pass) when the channel count allows: C / 16 workgroups with one or
        // two passes each instead of ONE workgroup walking up to 24 dependent passes (14 us on every block's critical path)
        ColGeom gs = g;
        int slabs = pgrid.y;
        if (C % 16 == 0 && C >= 32) { gs.cslab = 16; gs.tpr = 4; gs.rpb = 64; slabs = C / 16; }
        hipLaunchKernelGGL(bn_bwd_small_kernel, dim3(1, slabs), dim3(256), 0, st, (const float*)dy, ymask, (const float*)x, mean,
                           rstd, scale, relu_mask, training, dgamma, dbeta, (float*)dx, gout, gs);
        return sc_check_launch("bn_bwd");
    }
    const double* part = (const double*)ws;
    int nparts = nb;
    if (ext_partial) { part = ext_partial; nparts = ext_rows; }   // reduced by the epilogue of the kernel that produced dy
#define CSP(XB_, DB_)                                                                                                 \
        hipLaunchKernelGGL((colsum_partial_kernel<1, XB_, DB_>), pgrid, dim3(256), 0, st, (const float*)dy, ymask,          \
                           (const float*)x, mean, rstd, relu_mask, (double*)ws, g)
    else if (xb && db) CSP(true, true);
    else if (xb) CSP(true, false);
    else if (db) CSP(false, true);
    else CSP(false, false);
#undef CSP
    if (fin_narrow(C, nparts))
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<1>, dim3(C), dim3(256), 0, st, part, nparts, M, C, training, dgamma, dbeta, c1, c2);
    else
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<4>, dim3(sc_cdiv(C, 4)), dim3(256), 0, st, part, nparts, M, C, training, dgamma,
                           dbeta, c1, c2);
    const long n4 = M * C / 4;
    const int hb4 = ew_blocks_hoist(n4, C / 4);
#define BBA(XB_, YB_, DB_)                                                                                            \
    do {                                                                                                              \
        if (hb4) hipLaunchKernelGGL((bn_bwd_apply_kernel<XB_, YB_, DB_, true>), dim3(hb4), dim3(256), 0, st, dy, ymask, x,  \
                                    mean, rstd, scale, c1, c2, relu_mask, dx, gout, n4, C);                           \
        else hipLaunchKernelGGL((bn_bwd_apply_kernel<XB_, YB_, DB_>), dim3(ew_blocks(n4)), dim3(256), 0, st, dy, ymask, x,  \
                                mean, rstd, scale, c1, c2, relu_mask, dx, gout, n4, C);                               \
    } while (0)
    if (xb && yb && db && !ymask && !relu_mask && !gout && C % 8 == 0 && 256 % (C / 8) == 0)
        hipLaunchKernelGGL(bn_bwd_apply_bf16x8_kernel<true>, dim3(ew_blocks(n4 / 2)), dim3(256), 0, st, dy, x, mean, rstd, scale,
                           c1, c2, dx, n4 / 2, C);
    else if (xb && yb && db && !ymask && !relu_mask && !gout && C % 8 == 0)
        hipLaunchKernelGGL(bn_bwd_apply_bf16x8_kernel<false>, dim3(ew_blocks(n4 / 2)), dim3(256), 0, st, dy, x, mean, rstd, scale,
                           c1, c2, dx, n4 / 2, C);
    else
    switch ((xb ? 1 : 0) | (yb ? 2 : 0) | (db ? 4 : 0)) {
        case 0: BBA(false, false, false); break;
        case 1: BBA(true, false, false); break;
        case 2: BBA(false, true, false); break;
        case 3: BBA(true, true, false); break;
        case 4: BBA(false, false, true); break;
        case 5: BBA(true, false, true); break;
        case 6: BBA(false, true, true); break;
        default: BBA(true, true, true); break;
    }
#undef BBA
    return sc_check_launch("bn_bwd");
}

extern "C" int scouter_bn_bwd_f32(const float* dy, const float* ymask, const float* x, const float* mean,
                                  const float* rstd, const float* scale, const unsigned long long* relu_mask, long M,
                                  int C, int training, float* dgamma, float* dbeta, float* dx, float* gout,
                                  const double* ext_partial, int ext_rows, void* ws, size_t ws_bytes, void* stream) {
    return scouter_bn_bwd_io(dy, ymask, x, mean, rstd, scale, relu_mask, M, C, training, dgamma, dbeta, dx, gout,
                             ext_partial, ext_rows, 0, ws, ws_bytes, stream);
}

// ---- split attention fused with its BatchNorm (bn0), backward (timm/models/layers/split_attn.py:62-80 + bn0/act0).
// x0 = raw output of the radix convolution [B][HW][C2]; h0 = relu(bn0(x0)) is never stored.  One pass produces
//   g = (dout * a_r + dgap / HW) * [bn0(x0) > 0]      (gradient w.r.t. the BatchNorm output, ReLU folded in)
// is never stored either: pass 1 forms it and reduces g, g * xhat per block; pass 2 (after the finalize) forms it again
// and writes dx0 = scale * (g - c1 - xhat * c2).  16 bytes per element of the [B][HW][2C'] tensor in total; the
// unfused chain moved 26 (sa_apply_bwd writing dh0, colsum_partial<1> re-reading dh0 + x0 + mask, the apply pass).
__device__ __forceinline__ f32x4 sa_bn_g(const float* __restrict__ dout, const float* __restrict__ a,
                                         const float* __restrict__ dgap, long r, int b, int c, int cp, int C, int Cp,
                                         float inv_hw, f32x4 x, f32x4 mu, f32x4 sc, f32x4 sh) {
    const f32x4 d = *(const f32x4*)(dout + r * Cp + cp);
    const f32x4 av = *(const f32x4*)(a + (long)b * C + c);
    const f32x4 gp = *(const f32x4*)(dgap + (long)b * Cp + cp) * inv_hw;
    const f32x4 v = bn_affine(x, mu, sc, sh);
    f32x4 gg = d * av + gp;
#pragma unroll
    for (int k = 0; k < 4; ++k) gg[k] = v[k] > 0.f ? gg[k] : 0.f;
    return gg;
}
template <bool XB = false>                                   // XB: x0 is stored as bf16
__global__ __launch_bounds__(256) void sa_bn_bwd_partial_kernel(const float* __restrict__ dout, const float* __restrict__ a,
                                                                const float* __restrict__ dgap,
                                                                const void* __restrict__ x0,
                                                                const float* __restrict__ bn,   // [4][C2]
                                                                double* __restrict__ part, ColGeom g, int HW, int Cp,
                                                                float inv_hw) {
    __shared__ double red[256 * 8];
    const int tid = threadIdx.x;
    const int cq = tid % g.tpr, rl = tid / g.tpr;
    const int c = blockIdx.y * g.cslab + cq * 4;
    double s[4] = {0, 0, 0, 0}, t[4] = {0, 0, 0, 0};
    if (rl < g.rpb && c < g.C) {
        const f32x4 mu = *(const f32x4*)(bn + c), rs = *(const f32x4*)(bn + g.C + c);
        const f32x4 sc = *(const f32x4*)(bn + 2 * g.C + c), sh = *(const f32x4*)(bn + 3 * g.C + c);
        const int cp = c >= Cp ? c - Cp : c;                      // column inside the radix half
        const long rstep = (long)gridDim.x * g.rpb;
        for (long r = (long)blockIdx.x * g.rpb + rl; r < g.M; r += rstep) {
            const int b = (int)((unsigned)r / (unsigned)HW);      // M < 2^31 (host check); HBM-bound pass
            const f32x4 x = sc_load4<XB>(x0, r * g.C + c);
            const f32x4 gg = sa_bn_g(dout, a, dgap, r, b, c, cp, g.C, Cp, inv_hw, x, mu, sc, sh);
            const f32x4 xh = (x - mu) * rs;
#pragma unroll
            for (int k = 0; k < 4; ++k) { s[k] += gg[k]; t[k] += (double)gg[k] * xh[k]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[tid * 8 + k] = s[k]; red[tid * 8 + 4 + k] = t[k]; }
    __syncthreads();
    if (rl == 0 && c < g.C) {
        for (int j = 1; j < g.rpb; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) { s[k] += red[(j * g.tpr + cq) * 8 + k]; t[k] += red[(j * g.tpr + cq) * 8 + 4 + k]; }
        double* o = part + ((long)blockIdx.x * g.C + c) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[k * 2] = s[k]; o[k * 2 + 1] = t[k]; }
    }
}

template <bool XB = false, bool HOIST = false>               // XB: x0 is stored as bf16; HOIST: see scale_shift_act_kernel
__global__ __launch_bounds__(256) void sa_bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ a,
                                                              const float* __restrict__ dgap,
                                                              const void* __restrict__ x0, const float* __restrict__ bn,
                                                              const float* __restrict__ c1, const float* __restrict__ c2,
                                                              float* __restrict__ dx, long n4, int C, int HW, int Cp,
                                                              float inv_hw, unsigned short* __restrict__ planes,
                                                              int nplanes) {
    const int c4n = C / 4;
    const long i_first = (long)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 mu, rs, sc, sh, k1, k2;
    auto params = [&](int c) {
        mu = *(const f32x4*)(bn + c); rs = *(const f32x4*)(bn + C + c);
        sc = *(const f32x4*)(bn + 2 * C + c); sh = *(const f32x4*)(bn + 3 * C + c);
        k1 = *(const f32x4*)(c1 + c); k2 = *(const f32x4*)(c2 + c);
    };
    if (HOIST) params((int)(i_first % c4n) * 4);
    for (long i = i_first; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c4n;
        const int c = (int)(i - r * c4n) * 4, cp = c >= Cp ? c - Cp : c;
        const int b = (int)((unsigned)r / (unsigned)HW);
        if (!HOIST) params(c);
        const f32x4 x = sc_load4_nt<XB>(x0, i * 4);
        const f32x4 gg = sa_bn_g(dout, a, dgap, r, b, c, cp, C, Cp, inv_hw, x, mu, sc, sh);
        const f32x4 xh = (x - mu) * rs;
        const f32x4 o = sc * (gg - k1 - xh * k2);
        if (dx) *(f32x4*)(dx + i * 4) = o;                              // (NULL: dgrad and wgrad both run on the planes)
        if (planes) store_planes4(planes, n4 * 4, nplanes, i, o);      // A operand of the plane input-gradient kernel
    }
}

// bf16 mode with a stored x0 and the gradient handed over as ONE bf16 plane only: eight elements per thread, 16-byte
// accesses (see scale_shift_act_bf16x8_kernel); same arithmetic as sa_bn_bwd_apply_kernel<true>
template <bool HOIST>                                        // (see scale_shift_act_bf16x8_kernel)
__global__ __launch_bounds__(256) void sa_bn_bwd_apply_bf16x8_kernel(const float* __restrict__ dout, const float* __restrict__ a,
                                                                     const float* __restrict__ dgap,
                                                                     const void* __restrict__ x0, const float* __restrict__ bn,
                                                                     const float* __restrict__ c1, const float* __restrict__ c2,
                                                                     long n8, int C, int HW, int Cp, float inv_hw,
                                                                     unsigned short* __restrict__ plane) {
    const int c8n = C / 8;
    const long i_first = (long)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 mu[2], rs[2], sc[2], sh[2], k1[2], k2[2];
    auto params = [&](int c0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = c0 + 4 * q;
            mu[q] = *(const f32x4*)(bn + c); rs[q] = *(const f32x4*)(bn + C + c);
            sc[q] = *(const f32x4*)(bn + 2 * C + c); sh[q] = *(const f32x4*)(bn + 3 * C + c);
            k1[q] = *(const f32x4*)(c1 + c); k2[q] = *(const f32x4*)(c2 + c);
        }
    };
    if (HOIST) params((int)(i_first % c8n) * 8);
    for (long i = i_first; i < n8; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c8n;
        const int c0 = (int)(i - r * c8n) * 8;
        if (!HOIST) params(c0);
        const int b = (int)((unsigned)r / (unsigned)HW);
        f32x4 x[2], o[2];
        sc_load8_bf16(x0, i * 8, x[0], x[1]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = c0 + 4 * q, cp = c >= Cp ? c - Cp : c;
            const f32x4 gg = sa_bn_g(dout, a, dgap, r, b, c, cp, C, Cp, inv_hw, x[q], mu[q], sc[q], sh[q]);
            const f32x4 xh = (x[q] - mu[q]) * rs[q];
            o[q] = sc[q] * (gg - k1[q] - xh * k2[q]);
        }
        sc_store8_bf16(plane, i * 8, o[0], o[1]);
    }
}

// Finalize of the split-attention / bn0 backward from the per-image statistics the d(attention) pass produced
// (misc_ops.hip sa_colsum_partial_kernel<true, true>): one thread per channel sums over the images.
// (SBS_L lanes per channel share the loop over the images and are added by a fixed-order shuffle tree: one thread per
//  channel walked B dependent rounds of five loads in ONE to FOUR workgroups -- 23 us on the critical path of every block)
constexpr int SBS_L = 16;
__global__ __launch_bounds__(256) void sa_bn_bwd_sums_kernel(const double* __restrict__ sums, const float* __restrict__ a,
                                                             const float* __restrict__ dgap, int B, int C, int Cp,
                                                             float inv_hw, long M, int training,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             float* __restrict__ c1, float* __restrict__ c2) {
    const int t0 = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = t0 / SBS_L, j = t0 % SBS_L;
    const bool ok = c < C;
    const int cc = ok ? c : 0;
    const int cp = cc >= Cp ? cc - Cp : cc;
    double s = 0.0, t = 0.0;
    for (int b = j; b < B; b += SBS_L) {
        const double* q = sums + ((long)b * C + cc) * 4;
        const double av = (double)a[(long)b * C + cc], gp = (double)(dgap[(long)b * Cp + cp] * inv_hw);
        s += av * q[0] + gp * q[2];
        t += av * q[1] + gp * q[3];
    }
#pragma unroll
    for (int o = 1; o < SBS_L; o <<= 1) { s += __shfl_xor(s, o, SBS_L); t += __shfl_xor(t, o, SBS_L); }
    if (!ok || j != 0) return;
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)t;
    c1[c] = training ? (float)(s / (double)M) : 0.f;
    c2[c] = training ? (float)(t / (double)M) : 0.f;
}

// `io` & SC_IO_X_BF16: x0 (the raw radix-convolution output) is stored as bf16
extern "C" int scouter_sa_bn_bwd_io(const float* dout, const float* a, const float* dgap, const void* x0,
                                    const float* bn_saved, const double* bn_sums, int B, int HW, int Cp, int training,
                                    float* dgamma, float* dbeta, float* dx, void* dx_planes, int nplanes, int io, void* ws,
                                    size_t ws_bytes, void* stream) {
    SC_REQUIRE(dout && a && dgap && x0 && bn_saved && (dx || dx_planes) && B > 0 && HW > 0 && Cp % 4 == 0,
               "sa_bn_bwd: bad arguments");
    SC_REQUIRE((io & ~SC_IO_X_BF16) == 0, "sa_bn_bwd: unsupported io bits %d (only x0 may be bf16)", io);
    const bool xb = (io & SC_IO_X_BF16) != 0;
    const long M = (long)B * HW;
    const int C = 2 * Cp;
    SC_UNSUPPORTED(M < (1L << 31), "sa_bn_bwd: more than 2^31 pixels per batch");
    COL_CHECKS("sa_bn_bwd")
    const size_t coef_off = (size_t)nb * C * 2 * sizeof(double);
    if (ws_bytes < coef_off + 2 * (size_t)C * sizeof(float)) {
        sc_set_error("sa_bn_bwd: workspace too small");
        return SC_ERR_WORKSPACE;
    }
    float* c1 = (float*)((char*)ws + coef_off);
    float* c2 = c1 + C;
    hipStream_t st = (hipStream_t)stream;
    ScProfScope prof(bn_sums ? "sa_bn_bwd(finalize+apply)" : "sa_bn_bwd(reduce+finalize+apply)", st, 0,
                     ((bn_sums ? 8.0 : 12.0) * C + (bn_sums ? 4.0 : 8.0) * Cp - (xb ? (bn_sums ? 2.0 : 4.0) * C : 0.0)) * M);
    if (bn_sums) {
        hipLaunchKernelGGL(sa_bn_bwd_sums_kernel, dim3(sc_cdiv((long)C * SBS_L, 256)), dim3(256), 0, st, bn_sums, a, dgap, B, C, Cp,
                           1.f / HW, M, training, dgamma, dbeta, c1, c2);
    } else {
        if (xb) hipLaunchKernelGGL(sa_bn_bwd_partial_kernel<true>, pgrid, dim3(256), 0, st, dout, a, dgap, x0, bn_saved,
                                   (double*)ws, g, HW, Cp, 1.f / HW);
        else hipLaunchKernelGGL(sa_bn_bwd_partial_kernel<false>, pgrid, dim3(256), 0, st, dout, a, dgap, x0, bn_saved,
                                (double*)ws, g, HW, Cp, 1.f / HW);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<4>, dim3(sc_cdiv(C, 4)), dim3(256), 0, st, (const double*)ws, nb, M, C,
                           training, dgamma, dbeta, c1, c2);
    }
    const long n4 = M * C / 4;
    if (xb && !dx && dx_planes && nplanes == 1 && Cp % 8 == 0 && 256 % (C / 8) == 0)
        hipLaunchKernelGGL(sa_bn_bwd_apply_bf16x8_kernel<true>, dim3(ew_blocks(n4 / 2)), dim3(256), 0, st, dout, a, dgap, x0,
                           bn_saved, c1, c2, n4 / 2, C, HW, Cp, 1.f / HW, (unsigned short*)dx_planes);
    else if (xb && !dx && dx_planes && nplanes == 1 && Cp % 8 == 0)
        hipLaunchKernelGGL(sa_bn_bwd_apply_bf16x8_kernel<false>, dim3(ew_blocks(n4 / 2)), dim3(256), 0, st, dout, a, dgap, x0,
                           bn_saved, c1, c2, n4 / 2, C, HW, Cp, 1.f / HW, (unsigned short*)dx_planes);
    else {
        const int hb = ew_blocks_hoist(n4, C / 4);
#define SBA(XB_, H_, G_)                                                                                              \
        hipLaunchKernelGGL((sa_bn_bwd_apply_kernel<XB_, H_>), dim3(G_), dim3(256), 0, st, dout, a, dgap, x0, bn_saved, c1, c2, \
                           dx, n4, C, HW, Cp, 1.f / HW, (unsigned short*)dx_planes, nplanes)
        if (xb && hb) SBA(true, true, hb);
        else if (xb) SBA(true, false, ew_blocks(n4));
        else if (hb) SBA(false, true, hb);
        else SBA(false, false, ew_blocks(n4));
#undef SBA
    }
    return sc_check_launch("sa_bn_bwd");
}
extern "C" int scouter_sa_bn_bwd_f32(const float* dout, const float* a, const float* dgap, const float* x0,
                                     const float* bn_saved, const double* bn_sums, int B, int HW, int Cp, int training,
                                     float* dgamma, float* dbeta, float* dx, void* dx_planes, int nplanes, void* ws,
                                     size_t ws_bytes, void* stream) {
    return scouter_sa_bn_bwd_io(dout, a, dgap, x0, bn_saved, bn_sums, B, HW, Cp, training, dgamma, dbeta, dx, dx_planes,
                                nplanes, 0, ws, ws_bytes, stream);
}

// out[c] = alpha * sum_m a[m][c] (* b[m][c] when b != NULL)
extern "C" int scouter_colsum_f32(const float* a, const float* b, float* out, long M, int C, float alpha, void* ws,
                                  size_t ws_bytes, void* stream) {
    SC_REQUIRE(a && out, "colsum: null pointer");
    COL_CHECKS("colsum")
    hipStream_t st = (hipStream_t)stream;
    // few rows (bias gradients of the pooled-vector layers: M = batch): ONE launch, one row block per channel slab -- also
    // for the wide layers (C >= 256: 1-4 rows per pass), where a second 5-us launch costs more than the 70 sequential,
    // independent row loads of a thread
    if (M <= (long)g.rpb * 12 || M <= 256) {
        dim3 sgrid(1, pgrid.y);
        if (b) hipLaunchKernelGGL(colsum_partial_kernel<3>, sgrid, dim3(256), 0, st, a, b, nullptr, nullptr, nullptr, nullptr, (double*)ws, g, out, alpha);
        else hipLaunchKernelGGL(colsum_partial_kernel<2>, sgrid, dim3(256), 0, st, a, nullptr, nullptr, nullptr, nullptr, nullptr, (double*)ws, g, out, alpha);
        return sc_check_launch("colsum");
    }
    if (b) hipLaunchKernelGGL(colsum_partial_kernel<3>, pgrid, dim3(256), 0, st, a, b, nullptr, nullptr, nullptr, nullptr, (double*)ws, g);
    else hipLaunchKernelGGL(colsum_partial_kernel<2>, pgrid, dim3(256), 0, st, a, nullptr, nullptr, nullptr, nullptr, nullptr, (double*)ws, g);
    hipLaunchKernelGGL(colsum_finalize_kernel<4>, dim3(sc_cdiv(C, 4)), dim3(256), 0, st, (const double*)ws, nb, C, out, alpha);
    return sc_check_launch("colsum");
}

extern "C" int scouter_relu_bwd_f32(const float* dy, const float* y, float* dx, long n, void* stream) {
    SC_REQUIRE(dy && y && dx && n % 4 == 0, "relu_bwd: null pointer or n %% 4 != 0");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n / 4);
    return sc_check_launch("relu_bwd");
}

extern "C" int scouter_axpby_f32(const float* a, const float* b, float* y, float alpha, float beta, long n,
                                 void* stream) {
    SC_REQUIRE(a && y && n % 4 == 0, "axpby: null pointer or n %% 4 != 0");
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, a, b, y, alpha, beta, n / 4);
    return sc_check_launch("axpby");
}

static int pool_out(int in, int k, int s, int p, int ceil_mode) {
    int o = ceil_mode ? (in + 2 * p - k + s - 1) / s + 1 : (in + 2 * p - k) / s + 1;
    if (ceil_mode && (o - 1) * s >= in + p) --o;   // last window must start inside the input or left padding
    return o;
}
extern "C" int scouter_pool_out_size(int in, int k, int stride, int pad, int ceil_mode) {
    return pool_out(in, k, stride, pad, ceil_mode);
}

#define POOL_SETUP(name)                                                                                       \
    SC_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && k > 0 && k <= 15 && stride > 0, name ": bad shape"); \
    PoolGeom g{B, H, W, C, pool_out(H, k, stride, pad, ceil_mode), pool_out(W, k, stride, pad, ceil_mode), k, stride, \
               pad, count_include_pad};                                                                        \
    hipStream_t st = (hipStream_t)stream;

extern "C" int scouter_maxpool_fwd_f32(const float* x, float* y, unsigned char* argmax, int B, int H, int W, int C,
                                       int k, int stride, int pad, void* stream) {
    const int ceil_mode = 0, count_include_pad = 0;
    POOL_SETUP("maxpool_fwd")
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_blocks((long)B * g.Ho * g.Wo * C / 4)), dim3(256), 0, st, x, y, argmax, g,
                       (const float*)nullptr);
    return sc_check_launch("maxpool_fwd");
}

// ---- BatchNorm + ReLU + MaxPool2d in one pass each way (the stem: bn1, act1, maxpool -- resnet.py:404-412, 494-496).
// bn_saved = [4][C] block (mean, rstd, scale, shift) finalised by scouter_bn_fwd_f32(y = NULL).  Forward: pooled
// relu(bn(x)) and the arg-max taps; the activation and its sign mask are never stored.  Backward: dy (pooled gradient)
// -> dx of the BatchNorm input, dgamma / dbeta; the ReLU sign is re-evaluated with the same fma.
extern "C" int scouter_bn_maxpool_fwd_f32(const float* x, const float* bn_saved, float* y, unsigned char* argmax, int B,
                                          int H, int W, int C, int k, int stride, int pad, void* stream) {
    const int ceil_mode = 0, count_include_pad = 0;
    SC_REQUIRE(x && bn_saved && y, "bn_maxpool_fwd: null pointer");
    POOL_SETUP("bn_maxpool_fwd")
    ScProfScope prof("bn_maxpool_fwd", st, 0, 4.0 * ((double)B * H * W * C + (double)B * g.Ho * g.Wo * C * 1.25));
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_blocks((long)B * g.Ho * g.Wo * C / 4)), dim3(256), 0, st, x, y, argmax, g,
                       bn_saved);
    return sc_check_launch("bn_maxpool_fwd");
}
// `io` & SC_IO_Y_BF16: dx is stored as bf16 (a dx read by bf16-input convolution kernels only)
extern "C" int scouter_bn_maxpool_bwd_io(const float* dy, const unsigned char* argmax, const float* x,
                                         const float* bn_saved, int B, int H, int W, int C, int k, int stride, int pad,
                                         int training, float* dgamma, float* dbeta, void* dx, int io, void* ws,
                                         size_t ws_bytes, void* stream) {
    const int ceil_mode = 0, count_include_pad = 0;
    SC_REQUIRE(dy && argmax && x && bn_saved && dx, "bn_maxpool_bwd: null pointer");
    SC_REQUIRE((io & ~SC_IO_Y_BF16) == 0, "bn_maxpool_bwd: unsupported io bits %d (only dx may be bf16)", io);
    POOL_SETUP("bn_maxpool_bwd")
    SC_UNSUPPORTED(pool_bwd_rows_ok(g), "bn_maxpool_bwd: stride 1 / 2, B*H <= 65535 only");
    const long Mp = (long)B * g.Ho * g.Wo;
    ColGeom cg = col_geom(Mp, C);
    SC_REQUIRE(cg.cslab % 4 == 0 && 256 % cg.tpr == 0, "bn_maxpool_bwd: unsupported channel count %d", C);
    const int nb = col_blocks(Mp, cg);
    const size_t coef_off = (size_t)nb * C * 2 * sizeof(double);
    if (!ws || ws_bytes < coef_off + 2 * (size_t)C * sizeof(float)) {
        sc_set_error("bn_maxpool_bwd: workspace too small");
        return SC_ERR_WORKSPACE;
    }
    float* coef = (float*)((char*)ws + coef_off);
    ScProfScope prof("bn_maxpool_bwd(reduce+finalize+apply)", st, 0,
                     8.0 * (double)B * H * W * C + 9.0 * (double)Mp * C);
    dim3 pgrid(nb, (C + cg.cslab - 1) / cg.cslab);
    hipLaunchKernelGGL(bn_maxpool_bwd_partial_kernel, pgrid, dim3(256), 0, st, dy, argmax, x, bn_saved, (double*)ws, g, cg);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<4>, dim3(sc_cdiv(C, 4)), dim3(256), 0, st, (const double*)ws, nb,
                       (long)B * H * W, C, training, dgamma, dbeta, coef, coef + C);
    pool_bwd_rows<true>(dy, argmax, dx, g, st, x, bn_saved, coef, (io & SC_IO_Y_BF16) != 0);
    return sc_check_launch("bn_maxpool_bwd");
}
extern "C" int scouter_bn_maxpool_bwd_f32(const float* dy, const unsigned char* argmax, const float* x,
                                          const float* bn_saved, int B, int H, int W, int C, int k, int stride, int pad,
                                          int training, float* dgamma, float* dbeta, float* dx, void* ws,
                                          size_t ws_bytes, void* stream) {
    return scouter_bn_maxpool_bwd_io(dy, argmax, x, bn_saved, B, H, W, C, k, stride, pad, training, dgamma, dbeta, dx, 0, ws,
                                     ws_bytes, stream);
}
extern "C" int scouter_maxpool_bwd_f32(const float* dy, const unsigned char* argmax, float* dx, int B, int H, int W,
                                       int C, int k, int stride, int pad, void* stream) {
    const int ceil_mode = 0, count_include_pad = 0;
    POOL_SETUP("maxpool_bwd")
    if (pool_bwd_rows<true>(dy, argmax, dx, g, st)) return sc_check_launch("maxpool_bwd");
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_blocks((long)B * H * W * C / 4)), dim3(256), 0, st, dy, argmax, dx, g);
    return sc_check_launch("maxpool_bwd");
}
// `io` & SC_IO_X_BF16: x is stored as bf16 (the pooled tensor is always fp32)
extern "C" int scouter_avgpool_fwd_io(const void* x, float* y, int B, int H, int W, int C, int k, int stride, int pad,
                                      int ceil_mode, int count_include_pad, int io, void* stream) {
    POOL_SETUP("avgpool_fwd")
    SC_REQUIRE((io & ~SC_IO_X_BF16) == 0, "avgpool_fwd: unsupported io bits %d (only the input may be bf16)", io);
    const dim3 grid(ew_blocks((long)B * g.Ho * g.Wo * C / 4));
    if (io & SC_IO_X_BF16) hipLaunchKernelGGL(avgpool_fwd_kernel<true>, grid, dim3(256), 0, st, x, y, g);
    else hipLaunchKernelGGL(avgpool_fwd_kernel<false>, grid, dim3(256), 0, st, x, y, g);
    return sc_check_launch("avgpool_fwd");
}
extern "C" int scouter_avgpool_fwd_f32(const float* x, float* y, int B, int H, int W, int C, int k, int stride, int pad,
                                       int ceil_mode, int count_include_pad, void* stream) {
    return scouter_avgpool_fwd_io(x, y, B, H, W, C, k, stride, pad, ceil_mode, count_include_pad, 0, stream);
}
extern "C" int scouter_avgpool_bwd_f32(const float* dy, float* dx, int B, int H, int W, int C, int k, int stride,
                                       int pad, int ceil_mode, int count_include_pad, void* stream) {
    POOL_SETUP("avgpool_bwd")
    if (pool_bwd_rows<false>(dy, nullptr, dx, g, st)) return sc_check_launch("avgpool_bwd");
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(ew_blocks((long)B * H * W * C / 4)), dim3(256), 0, st, dy, dx, g);
    return sc_check_launch("avgpool_bwd");
}

// in: [batch][rows][cols] -> out: [batch][cols][rows]   (NCHW->NHWC: rows=C, cols=H*W;  NHWC->NCHW: rows=H*W, cols=C)
extern "C" int scouter_transpose_f32(const float* in, float* out, int batch, int rows, int cols, void* stream) {
    SC_REQUIRE(in && out && batch > 0 && rows > 0 && cols > 0, "transpose: bad arguments");
    dim3 grid(sc_cdiv(cols, 32), sc_cdiv(rows, 32), batch);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, (hipStream_t)stream, in, out, rows, cols);
    return sc_check_launch("transpose");
}
