// Split-attention glue (ResNeSt), stem im2col, sine positional encoding, loss head, fused AdamW, metrics.
// Reference call sites: timm/models/layers/split_attn.py:62-80 (radix sum, GAP, RadixSoftmax, weighted sum);
// sloter/utils/position_encode.py:26-46; sloter/slot_model.py:117-121 (log_softmax, NLL, lambda*area);
// train.py:146 (AdamW); tools/calculate_tool.py:4-7 (top-1).
#include "common.h"

static inline int ew_blocks(long n) { long b = (n + 255) / 256; return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }

// ---------------------------------------------------------------------------------------------------------------
// split attention
// ---------------------------------------------------------------------------------------------------------------
// Per-image column sums over the HW rows of x[b] ([HW][C2], C2 = radix*C'), optionally of x*w (w: [HW][C'] broadcast
// over the radix halves).  partial: [B][nsplit][C2] doubles.
// bn != NULL ([4][C2]: mean, rstd, scale, shift): x is the RAW convolution output and the summed quantity is
// relu(bn(x)) -- the BatchNorm apply pass of bn0 is folded into the reductions that read the tensor anyway.
// STATS (with WITH_W and bn): the pass that forms d(attention) also reduces, per image and channel, what the backward of
// bn0 needs -- g = (dout * a + dgap / HW) * [bn0(x0) > 0] is affine in (a, dgap), which are constant over an image, so
//   sum g       = sum_b  a[b,c] * S1 + dgap[b,c'] / HW * S3,     S1 = sum_hw dout * m,         S3 = sum_hw m
//   sum g xhat  = sum_b  a[b,c] * S2 + dgap[b,c'] / HW * S4,     S2 = sum_hw dout * m * xhat,  S4 = sum_hw m * xhat
// with m = [bn0(x0) > 0]: the separate reduction pass of scouter_sa_bn_bwd_f32 over (dout, x0) disappears.
// partial planes: [1 + 4][B][nsplit][C2] doubles (plane 0 = the column sums as before).
template <bool WITH_W, bool STATS, bool XB = false>      // XB: x (the raw radix-convolution output) is stored as bf16
__global__ __launch_bounds__(256) void sa_colsum_partial_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bn,
                                                                double* __restrict__ part, int HW, int C2, int Cp,
                                                                int tpr, int rpb, int rows_per_split, long plane) {
    constexpr int NQ = STATS ? 5 : 1;
    __shared__ double red[256 * 4 * NQ];
    const int tid = threadIdx.x, cq = tid % tpr, rl = tid / tpr, b = blockIdx.y, sp = blockIdx.x;
    const int c = cq * 4;
    const int r0 = sp * rows_per_split, r1 = min(HW, r0 + rows_per_split);
    double s[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) s[q][k] = 0.0;
    const long xb0 = (long)b * HW * C2;
    const float* wb = WITH_W ? w + (long)b * HW * Cp : nullptr;
    f32x4 mu = {0, 0, 0, 0}, rs = {1, 1, 1, 1}, sc = {1, 1, 1, 1}, sh = {0, 0, 0, 0};
    if (bn) {
        mu = *(const f32x4*)(bn + c); rs = *(const f32x4*)(bn + C2 + c);
        sc = *(const f32x4*)(bn + 2 * C2 + c); sh = *(const f32x4*)(bn + 3 * C2 + c);
    }
    // the rows of a thread are summed in their order, but FOUR rows' loads are requested together: one row in flight per
    // thread left the pass latency-bound (a dozen dependent round trips per workgroup: 165 us for 308 MB at batch 256)
    auto row = [&](const f32x4& xr, f32x4 d) {
        f32x4 v = xr, hv = xr;
        if (bn) {
            hv = bn_affine(xr, mu, sc, sh);                      // (the sign of THIS value is the ReLU mask everywhere)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(hv[k], 0.f);
        }
        if (WITH_W) v *= d;
#pragma unroll
        for (int k = 0; k < 4; ++k) s[0][k] += v[k];
        if (STATS) {
            const f32x4 xh = (xr - mu) * rs;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool m = hv[k] > 0.f;
                const float dm = m ? d[k] : 0.f;
                s[1][k] += dm;
                s[2][k] += (double)dm * xh[k];
                s[3][k] += m ? 1.0 : 0.0;
                s[4][k] += m ? (double)xh[k] : 0.0;
            }
        }
    };
    int r = r0 + rl;
    for (; r + 3 * rpb < r1; r += 4 * rpb) {
        f32x4 xr[4], d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xr[u] = sc_load4<XB>(x, xb0 + (long)(r + u * rpb) * C2 + c);
            d[u] = f32x4{1, 1, 1, 1};
            if (WITH_W) d[u] = *(const f32x4*)(wb + (long)(r + u * rpb) * Cp + (c % Cp));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) row(xr[u], d[u]);
    }
    for (; r < r1; r += rpb) {
        const f32x4 xr = sc_load4<XB>(x, xb0 + (long)r * C2 + c);
        f32x4 d = {1, 1, 1, 1};
        if (WITH_W) d = *(const f32x4*)(wb + (long)r * Cp + (c % Cp));
        row(xr, d);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[(q * 256 + tid) * 4 + k] = s[q][k];
    __syncthreads();
    if (rl == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            for (int j = 1; j < rpb; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) s[q][k] += red[(q * 256 + j * tpr + cq) * 4 + k];
            double* o = part + q * plane + ((long)b * gridDim.x + sp) * C2 + c;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = s[q][k];
        }
    }
}

// d(attention) and the four per-image statistics out of the five partial planes of the STATS pass: one thread per
// (image, channel); stats [B][C2][4] doubles = (S1, S2, S3, S4).
// SA_FL lanes share one output: lane j sums the splits j, j + SA_FL, ... and a fixed-order shuffle tree adds the lanes
// (deterministic).  One thread per output walked up to 32 dependent partial loads -- these finalizes sit on the critical
// path of every block, where the length of the dependent-load chain, not the work, is the cost.
constexpr int SA_FL = 8;
__global__ void sa_dattn_stats_finalize_kernel(const double* __restrict__ part, float* __restrict__ da,
                                               double* __restrict__ stats, int B, int nsplit, int C2, long plane) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long i = t / SA_FL;
    const int j = (int)(t % SA_FL);
    const bool ok = i < (long)B * C2;
    const int b = ok ? (int)(i / C2) : 0, c = ok ? (int)(i % C2) : 0;
    double acc[5] = {0, 0, 0, 0, 0};
    for (int k = j; k < nsplit; k += SA_FL) {
        const double* p = part + ((long)b * nsplit + k) * C2 + c;
#pragma unroll
        for (int q = 0; q < 5; ++q) acc[q] += p[q * plane];
    }
#pragma unroll
    for (int o = 1; o < SA_FL; o <<= 1)
#pragma unroll
        for (int q = 0; q < 5; ++q) acc[q] += __shfl_xor(acc[q], o, SA_FL);
    if (ok && j == 0) {
        da[i] = (float)acc[0];
#pragma unroll
        for (int q = 0; q < 4; ++q) stats[i * 4 + q] = acc[1 + q];
    }
}

// gap[b][c] = alpha * sum_splits (part[b][s][c] + part[b][s][Cp+c])     (fold_radix = 1)
// out[b][c2] = alpha * sum_splits part[b][s][c2]                          (fold_radix = 0)
__global__ void sa_colsum_finalize_kernel(const double* __restrict__ part, float* __restrict__ out, int B, int nsplit,
                                          int C2, int Cp, int fold_radix, float alpha) {
    const int Co = fold_radix ? Cp : C2;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long i = t / SA_FL;
    const int j = (int)(t % SA_FL);
    const bool ok = i < (long)B * Co;
    const int b = ok ? (int)(i / Co) : 0, c = ok ? (int)(i % Co) : 0;
    double acc = 0;
    for (int k = j; k < nsplit; k += SA_FL) {
        const double* p = part + ((long)b * nsplit + k) * C2;
        double v = p[c];
        if (fold_radix) for (int r = 1; r * Cp < C2; ++r) v += p[r * Cp + c];
        acc += v;
    }
#pragma unroll
    for (int o = 1; o < SA_FL; o <<= 1) acc += __shfl_xor(acc, o, SA_FL);
    if (ok && j == 0) out[i] = (float)(acc * alpha);
}

// radix-2 softmax over (z[b][c], z[b][Cp+c])  (RadixSoftmax with cardinality 1, split_attn.py:20-28)
__global__ void radix_softmax_fwd_kernel(const float* __restrict__ z, float* __restrict__ a, int B, int Cp) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * Cp) return;
    const int b = (int)(i / Cp), c = (int)(i % Cp);
    const float z0 = z[(long)b * 2 * Cp + c], z1 = z[(long)b * 2 * Cp + Cp + c];
    const float m = fmaxf(z0, z1), e0 = expf(z0 - m), e1 = expf(z1 - m), inv = 1.f / (e0 + e1);
    a[(long)b * 2 * Cp + c] = e0 * inv;
    a[(long)b * 2 * Cp + Cp + c] = e1 * inv;
}
__global__ void radix_softmax_bwd_kernel(const float* __restrict__ a, const float* __restrict__ da,
                                         float* __restrict__ dz, int B, int Cp) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * Cp) return;
    const int b = (int)(i / Cp), c = (int)(i % Cp);
    const long i0 = (long)b * 2 * Cp + c, i1 = i0 + Cp;
    const float dot = a[i0] * da[i0] + a[i1] * da[i1];
    dz[i0] = a[i0] * (da[i0] - dot);
    dz[i1] = a[i1] * (da[i1] - dot);
}

// the weighted sum of one channel quad -- ONE function for every instantiation, so that the compiler contracts the
// multiply-adds the same way everywhere (the 8-element bf16 kernel must give the bits of the 4-element one)
__device__ __forceinline__ f32x4 sa_weighted_sum(f32x4 x0, f32x4 x1, f32x4 a0, f32x4 a1) {
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = __builtin_fmaf(x0[k], a0[k], x1[k] * a1[k]);      // (what `x0*a0 + x1*a1` contracted to)
    return o;
}
// out[b,hw,c] = x[b,hw,c]*a[b,c] + x[b,hw,Cp+c]*a[b,Cp+c]
// HOIST: the grid stride is a multiple of Cp / 4 -- bn0's parameters of the thread's channels are loaded once (bn_elem.hip)
template <bool XB = false, bool YB = false, bool HOIST = false>      // storage of x (raw convolution output) / out
__global__ __launch_bounds__(256) void sa_apply_fwd_kernel(const void* __restrict__ x, const float* __restrict__ a,
                                                           const float* __restrict__ bn, void* __restrict__ out,
                                                           long n4, int HW, int Cp) {
    const int c4n = Cp / 4, C2 = 2 * Cp;
    const long i_first = (long)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 m0, s0, b0, m1, s1, b1;
    auto params = [&](int c) {
        if (!bn) return;
        m0 = *(const f32x4*)(bn + c); s0 = *(const f32x4*)(bn + 2 * C2 + c); b0 = *(const f32x4*)(bn + 3 * C2 + c);
        m1 = *(const f32x4*)(bn + Cp + c); s1 = *(const f32x4*)(bn + 2 * C2 + Cp + c); b1 = *(const f32x4*)(bn + 3 * C2 + Cp + c);
    };
    if (HOIST) params((int)(i_first % c4n) * 4);
    for (long i = i_first; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const long row = i / c4n;                 // b*HW + hw
        const int b = (int)(row / HW);
        if (!HOIST) params(c);
        f32x4 x0 = sc_load4<XB>(x, row * 2 * Cp + c), x1 = sc_load4<XB>(x, row * 2 * Cp + Cp + c);
        if (bn) {                                 // x is the raw convolution output: relu(bn0(x)) on the fly
            x0 = bn_affine(x0, m0, s0, b0);
            x1 = bn_affine(x1, m1, s1, b1);
#pragma unroll
            for (int k = 0; k < 4; ++k) { x0[k] = fmaxf(x0[k], 0.f); x1[k] = fmaxf(x1[k], 0.f); }
        }
        const f32x4 a0 = *(const f32x4*)(a + (long)b * 2 * Cp + c), a1 = *(const f32x4*)(a + (long)b * 2 * Cp + Cp + c);
        sc_store4<YB>(out, i * 4, sa_weighted_sum(x0, x1, a0, a1));
    }
}
// x and out both bf16-stored: eight channels per thread, 16-byte accesses (bn_elem.hip scale_shift_act_bf16x8_kernel);
// same arithmetic as sa_apply_fwd_kernel<true, true>
template <bool HOIST>                   // the grid stride is a multiple of Cp / 8: bn0's parameters are loaded once (bn_elem.hip)
__global__ __launch_bounds__(256) void sa_apply_fwd_bf16x8_kernel(const void* __restrict__ x, const float* __restrict__ a,
                                                                  const float* __restrict__ bn, void* __restrict__ out,
                                                                  long n8, int HW, int Cp) {
    const int c8n = Cp / 8, C2 = 2 * Cp;
    const long i_first = (long)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 m0[2], s0[2], b0[2], m1[2], s1[2], b1[2];
    auto params = [&](int c0) {
        if (!bn) return;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = c0 + 4 * q;
            m0[q] = *(const f32x4*)(bn + c); s0[q] = *(const f32x4*)(bn + 2 * C2 + c); b0[q] = *(const f32x4*)(bn + 3 * C2 + c);
            m1[q] = *(const f32x4*)(bn + Cp + c); s1[q] = *(const f32x4*)(bn + 2 * C2 + Cp + c);
            b1[q] = *(const f32x4*)(bn + 3 * C2 + Cp + c);
        }
    };
    if (HOIST) params((int)(i_first % c8n) * 8);
    for (long i = i_first; i < n8; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % c8n) * 8;
        if (!HOIST) params(c0);
        const long row = i / c8n;
        const int b = (int)(row / HW);
        f32x4 x0[2], x1[2], o[2];
        sc_load8_bf16(x, row * C2 + c0, x0[0], x0[1]);
        sc_load8_bf16(x, row * C2 + Cp + c0, x1[0], x1[1]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = c0 + 4 * q;
            if (bn) {
                x0[q] = bn_affine(x0[q], m0[q], s0[q], b0[q]);
                x1[q] = bn_affine(x1[q], m1[q], s1[q], b1[q]);
#pragma unroll
                for (int k = 0; k < 4; ++k) { x0[q][k] = fmaxf(x0[q][k], 0.f); x1[q][k] = fmaxf(x1[q][k], 0.f); }
            }
            const f32x4 a0 = *(const f32x4*)(a + (long)b * C2 + c), a1 = *(const f32x4*)(a + (long)b * C2 + Cp + c);
            o[q] = sa_weighted_sum(x0[q], x1[q], a0, a1);
        }
        sc_store8_bf16(out, i * 8, o[0], o[1]);
    }
}
// dx[b,hw,r*Cp+c] = dout[b,hw,c]*a[b,r*Cp+c] + dgap[b,c]*inv_hw
__global__ __launch_bounds__(256) void sa_apply_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ a,
                                                           const float* __restrict__ dgap, float* __restrict__ dx,
                                                           long n4, int HW, int Cp, float inv_hw) {
    const int c4n = Cp / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const long row = i / c4n;
        const int b = (int)(row / HW);
        const f32x4 d = *(const f32x4*)(dout + i * 4);
        const f32x4 gp = *(const f32x4*)(dgap + (long)b * Cp + c) * inv_hw;
        const f32x4 a0 = *(const f32x4*)(a + (long)b * 2 * Cp + c), a1 = *(const f32x4*)(a + (long)b * 2 * Cp + Cp + c);
        *(f32x4*)(dx + row * 2 * Cp + c) = d * a0 + gp;
        *(f32x4*)(dx + row * 2 * Cp + Cp + c) = d * a1 + gp;
    }
}

static int sa_plan(int HW, int C2, int* tpr, int* rpb, int* nsplit, int* rps) {
    if (C2 % 4 || C2 > 1024 || 256 % (C2 / 4)) return -1;
    *tpr = C2 / 4;
    *rpb = 256 / *tpr;
    int ns = HW / (*rpb * 8);
    if (ns < 1) ns = 1;
    if (ns > 32) ns = 32;
    *nsplit = ns;
    *rps = (HW + ns - 1) / ns;
    return 0;
}
extern "C" size_t scouter_sa_workspace_bytes(int B, int HW, int C2) { (void)HW; return (size_t)5 * B * 32 * C2 * sizeof(double); }

// mode 0: gap[b][c]  = mean_hw (x[.,c] + x[.,Cp+c])          (out: [B][Cp])
// mode 1: da[b][c2]  = sum_hw dout[b,hw,c2 % Cp] * x[b,hw,c2] (out: [B][2Cp]); with bn_sums_out (needs bn_saved) also
//         the per-image statistics [B][2Cp][4] (fp64) that let scouter_sa_bn_bwd_f32 skip its reduction pass
// `io` & SC_IO_X_BF16: x (the raw radix-convolution output) is stored as bf16
extern "C" int scouter_sa_reduce_io(const void* x, const float* dout, const float* bn_saved, float* out,
                                    double* bn_sums_out, int B, int HW, int Cp, int mode, int io, void* ws, size_t ws_bytes,
                                    void* stream) {
    const int C2 = 2 * Cp;
    int tpr, rpb, ns, rps;
    SC_REQUIRE(x && out && B > 0 && HW > 0, "sa_reduce: bad arguments");
    SC_REQUIRE((io & ~SC_IO_X_BF16) == 0, "sa_reduce: unsupported io bits %d (only x may be bf16)", io);
    const bool xb = (io & SC_IO_X_BF16) != 0;
    SC_REQUIRE(!bn_sums_out || (mode == 1 && bn_saved), "sa_reduce: bn_sums_out needs mode 1 and bn_saved");
    SC_UNSUPPORTED(sa_plan(HW, C2, &tpr, &rpb, &ns, &rps) == 0, "sa_reduce: unsupported channel count %d", C2);
    const long plane = (long)B * ns * C2;
    if (!ws || ws_bytes < (size_t)(bn_sums_out ? 5 : 1) * plane * sizeof(double)) { sc_set_error("sa_reduce: workspace too small"); return SC_ERR_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(ns, B);
#define SCP(W_, S_, DOUT_)                                                                                            \
    do {                                                                                                              \
        if (xb) hipLaunchKernelGGL((sa_colsum_partial_kernel<W_, S_, true>), grid, dim3(256), 0, st, x, DOUT_, bn_saved, (double*)ws, HW, C2, Cp, tpr, rpb, rps, plane); \
        else hipLaunchKernelGGL((sa_colsum_partial_kernel<W_, S_, false>), grid, dim3(256), 0, st, x, DOUT_, bn_saved, (double*)ws, HW, C2, Cp, tpr, rpb, rps, plane); \
    } while (0)
    if (mode == 0) {
        SCP(false, false, (const float*)nullptr);
        hipLaunchKernelGGL(sa_colsum_finalize_kernel, dim3(sc_cdiv((long)B * Cp * SA_FL, 256)), dim3(256), 0, st, (const double*)ws, out, B, ns, C2, Cp, 1, 1.f / HW);
    } else if (bn_sums_out) {
        SC_REQUIRE(dout, "sa_reduce: dout missing");
        SCP(true, true, dout);
        hipLaunchKernelGGL(sa_dattn_stats_finalize_kernel, dim3(sc_cdiv((long)B * C2 * SA_FL, 256)), dim3(256), 0, st, (const double*)ws, out, bn_sums_out, B, ns, C2, plane);
    } else {
        SC_REQUIRE(dout, "sa_reduce: dout missing");
        SCP(true, false, dout);
        hipLaunchKernelGGL(sa_colsum_finalize_kernel, dim3(sc_cdiv((long)B * C2 * SA_FL, 256)), dim3(256), 0, st, (const double*)ws, out, B, ns, C2, Cp, 0, 1.f);
    }
#undef SCP
    return sc_check_launch("sa_reduce");
}
extern "C" int scouter_sa_reduce_f32(const float* x, const float* dout, const float* bn_saved, float* out,
                                     double* bn_sums_out, int B, int HW, int Cp, int mode, void* ws, size_t ws_bytes,
                                     void* stream) {
    return scouter_sa_reduce_io(x, dout, bn_saved, out, bn_sums_out, B, HW, Cp, mode, 0, ws, ws_bytes, stream);
}
extern "C" int scouter_radix_softmax_fwd_f32(const float* z, float* a, int B, int Cp, void* stream) {
    hipLaunchKernelGGL(radix_softmax_fwd_kernel, dim3(sc_cdiv((long)B * Cp, 256)), dim3(256), 0, (hipStream_t)stream, z, a, B, Cp);
    return sc_check_launch("radix_softmax_fwd");
}
extern "C" int scouter_radix_softmax_bwd_f32(const float* a, const float* da, float* dz, int B, int Cp, void* stream) {
    hipLaunchKernelGGL(radix_softmax_bwd_kernel, dim3(sc_cdiv((long)B * Cp, 256)), dim3(256), 0, (hipStream_t)stream, a, da, dz, B, Cp);
    return sc_check_launch("radix_softmax_bwd");
}
// `io`: SC_IO_X_BF16 -- x is stored as bf16, SC_IO_Y_BF16 -- out is
extern "C" int scouter_sa_apply_fwd_io(const void* x, const float* a, const float* bn_saved, void* out, int B, int HW,
                                       int Cp, int io, void* stream) {
    SC_REQUIRE(Cp % 4 == 0, "sa_apply: Cp %% 4 != 0");
    SC_REQUIRE((io & ~(SC_IO_X_BF16 | SC_IO_Y_BF16)) == 0, "sa_apply: unknown io bits %d", io);
    const long n4 = (long)B * HW * Cp / 4;
    const dim3 grid(ew_blocks(n4));
    hipStream_t st = (hipStream_t)stream;
    if ((io & 3) == 3 && Cp % 8 == 0) {
        if (256 % (Cp / 8) == 0)
            hipLaunchKernelGGL(sa_apply_fwd_bf16x8_kernel<true>, dim3(ew_blocks(n4 / 2)), dim3(256), 0, st, x, a, bn_saved, out, n4 / 2, HW, Cp);
        else
            hipLaunchKernelGGL(sa_apply_fwd_bf16x8_kernel<false>, dim3(ew_blocks(n4 / 2)), dim3(256), 0, st, x, a, bn_saved, out, n4 / 2, HW, Cp);
        return sc_check_launch("sa_apply_fwd");
    }
    {
        const int cv = Cp / 4;
        int hb = 0;
        if (cv > 0 && 256 % cv == 0) hb = ew_blocks(n4);
        else if (cv % 256 == 0) { const int m = cv / 256; hb = ew_blocks(n4); if (hb > 1) hb = (hb + m - 1) / m * m; }
#define SAF(XB_, YB_)                                                                                                 \
        do {                                                                                                          \
            if (hb) hipLaunchKernelGGL((sa_apply_fwd_kernel<XB_, YB_, true>), dim3(hb), dim3(256), 0, st, x, a, bn_saved, out, n4, HW, Cp); \
            else hipLaunchKernelGGL((sa_apply_fwd_kernel<XB_, YB_, false>), grid, dim3(256), 0, st, x, a, bn_saved, out, n4, HW, Cp); \
        } while (0)
        switch (io & 3) {
            case 0: SAF(false, false); break;
            case 1: SAF(true, false); break;
            case 2: SAF(false, true); break;
            default: SAF(true, true); break;
        }
#undef SAF
    }
    return sc_check_launch("sa_apply_fwd");
}
extern "C" int scouter_sa_apply_fwd_f32(const float* x, const float* a, const float* bn_saved, float* out, int B, int HW,
                                        int Cp, void* stream) {
    return scouter_sa_apply_fwd_io(x, a, bn_saved, out, B, HW, Cp, 0, stream);
}
extern "C" int scouter_sa_apply_bwd_f32(const float* dout, const float* a, const float* dgap, float* dx, int B, int HW,
                                        int Cp, void* stream) {
    SC_REQUIRE(Cp % 4 == 0, "sa_apply_bwd: Cp %% 4 != 0");
    const long n4 = (long)B * HW * Cp / 4;
    hipLaunchKernelGGL(sa_apply_bwd_kernel, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream, dout, a, dgap, dx, n4, HW, Cp, 1.f / HW);
    return sc_check_launch("sa_apply_bwd");
}

// ---------------------------------------------------------------------------------------------------------------
// stem: im2col of the NCHW input image into [M = B*Ho*Wo][Kpad] rows, k = (ky*kw + kx)*Cin + ci, zero padded, so
// the small-Cin stem convolution and its weight gradient run on the generic MFMA GEMM path as a 1x1 conv.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col_nchw_kernel(const float* __restrict__ x, float* __restrict__ col, int B,
                                                          int Cin, int H, int W, int Ho, int Wo, int k, int stride,
                                                          int pad, int Kpad) {
    const long n = (long)B * Ho * Wo * Kpad;
    const int K = k * k * Cin;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % Kpad);
        long p = i / Kpad;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        float v = 0.f;
        if (kk < K) {
            const int ci = kk % Cin, tap = kk / Cin, ky = tap / k, kx = tap % k;
            const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((long)b * Cin + ci) * H + iy) * W + ix];
        }
        col[i] = v;
    }
}
// The same, one workgroup row per output pixel row (blockIdx.y = b * Ho + oy) and one 16-byte store per thread: the
// (ci, ky, kx) decomposition of the <= 64 patch columns is tabulated once per workgroup in LDS, the thread index only
// needs one exact float-reciprocal division.  (The element-wise kernel above: five integer divisions, two of them
// 64-bit, per 4-byte store -- ALU-bound at 1.6 TB/s.)
__global__ __launch_bounds__(256) void im2col_nchw_rows_kernel(const float* __restrict__ x, float* __restrict__ col, int Cin,
                                                               int H, int W, int Ho, int Wo, int k, int stride, int pad,
                                                               int Kpad, float inv_q) {
    __shared__ int tab[64];                       // kk -> ci | ky << 8 | kx << 16 | valid << 24
    const int K = k * k * Cin;
    if (threadIdx.x < 64) {
        const int kk = threadIdx.x, ci = kk % Cin, tap = kk / Cin;
        tab[kk] = kk < K ? (ci | (tap / k) << 8 | (tap % k) << 16 | 1 << 24) : 0;
    }
    __syncthreads();
    const int qn = Kpad / 4;                      // float4 per output pixel
    const int xid = blockIdx.x * 256 + threadIdx.x;
    if (xid >= Wo * qn) return;
    const int ox = (int)(((float)xid + 0.5f) * inv_q), q = xid - ox * qn;
    const int row = blockIdx.y, b = row / Ho, oy = row - b * Ho;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int t = tab[4 * q + e];
        const int iy = oy * stride - pad + ((t >> 8) & 0xff), ix = ox * stride - pad + ((t >> 16) & 0xff);
        if ((t >> 24) && iy >= 0 && iy < H && ix >= 0 && ix < W) v[e] = x[(((long)b * Cin + (t & 0xff)) * H + iy) * W + ix];
    }
    *(f32x4*)(col + ((long)row * Wo + ox) * Kpad + 4 * q) = v;
}
// wpad[Kpad][Cout] = {w[K][Cout]; 0}
__global__ void pad_rows_kernel(const float* __restrict__ w, float* __restrict__ wpad, long nvalid, long ntotal) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ntotal) wpad[i] = i < nvalid ? w[i] : 0.f;
}
extern "C" int scouter_im2col_nchw_f32(const float* x, float* col, int B, int Cin, int H, int W, int k, int stride,
                                       int pad, int Kpad, void* stream) {
    SC_REQUIRE(x && col && Kpad >= k * k * Cin && Kpad % 32 == 0, "im2col: bad arguments");
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const long n = (long)B * Ho * Wo * Kpad;
    const long xw = (long)Wo * (Kpad / 4), rows = (long)B * Ho;
    if (Kpad <= 64 && Cin < 256 && k < 256 && xw < (1 << 20) && rows <= 65535) {
        hipLaunchKernelGGL(im2col_nchw_rows_kernel, dim3((unsigned)((xw + 255) / 256), (unsigned)rows), dim3(256), 0,
                           (hipStream_t)stream, x, col, Cin, H, W, Ho, Wo, k, stride, pad, Kpad, 1.0f / (float)(Kpad / 4));
        return sc_check_launch("im2col");
    }
    hipLaunchKernelGGL(im2col_nchw_kernel, dim3(ew_blocks(n) * 2), dim3(256), 0, (hipStream_t)stream, x, col, B, Cin, H, W, Ho, Wo, k, stride, pad, Kpad);
    return sc_check_launch("im2col");
}
// ---------------------------------------------------------------------------------------------------------------
// Deep stem's first convolution, direct: 3x3 / stride 2 / pad 1, Cin = 3, NCHW image in, NHWC rows of 32 channels out
// (timm/models/resnet.py deep stem, conv1[0]; resnest.py stems).  27 x 32 FLOP pairs per output pixel against 3 x 4 bytes in and 128 bytes
// out: an HBM stream.  The im2col + GEMM route writes 128 bytes per pixel of patch rows, reads them again and pays a
// one-K-tile GEMM launch (88 + 182 us at 70 x 224^2); here a workgroup stages the 2 * ROWS + 1 image rows of its ROWS
// output rows in LDS (coalesced), eight lanes share an output pixel -- lane & 7 owns four output channels, its 27 x 4
// weights live in registers -- so a wave's store instruction writes 1 KB of consecutive NHWC bytes.  The BatchNorm
// statistics of the tile (fp64 from the first addition, as igemm_epilogue) go to bn_part[workgroup][32][2].
// Each filter row's nine terms are one fmaf chain (patch-column order kk = (ky * 3 + kx) * 3 + ci); the three chains are added in row order.
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int ROWS, bool VEC4>
__global__ __launch_bounds__(256) void stem_direct_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ y, double* __restrict__ bn_part, int H,
                                                          int W, int Ho, int Wo, int rblocks) {
    extern __shared__ __attribute__((aligned(16))) float st_lds[];      // [2 * ROWS + 1][3][WP], column j = ix + PADL
    constexpr int IR = 2 * ROWS + 1;
    constexpr int PADL = VEC4 ? 4 : 1;                // (VEC4: image columns sit 16-byte aligned in the stage)
    const int WP = VEC4 ? W + 4 : 2 * Wo + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / rblocks, oy0 = (blockIdx.x - b * rblocks) * ROWS;
    const int cg = tid & 7;
    const float* xb = x + (long)b * 3 * H * W;
    const int iy0 = 2 * oy0 - 1;
    // stage: one (image row, channel) line per wave and turn, lanes along the line (coalesced, no divisions)
    if constexpr (VEC4) {
        // W % 4 == 0, W <= 512: every line is at most two 16-byte loads per lane; all of a wave's loads are in flight
        // before the first is written to LDS (the stage is what a workgroup waits for)
        constexpr int TURNS = (IR * 3 + 3) / 4;
        f32x4 v[TURNS][2];
        const int nq = W >> 2;
#pragma unroll
        for (int t = 0; t < TURNS; ++t) {
            const int rc = wave + 4 * t, r = rc / 3, ci = rc - r * 3, iy = iy0 + r;
            const bool row_ok = rc < IR * 3 && iy >= 0 && iy < H;
            const f32x4* src = (const f32x4*)(xb + ((long)ci * H + (row_ok ? iy : 0)) * W);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int q = lane + 64 * h;
                v[t][h] = (row_ok && q < nq) ? src[q] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int t = 0; t < TURNS; ++t) {
            const int rc = wave + 4 * t;
            if (rc < IR * 3) {
                float* dst = st_lds + rc * WP;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int q = lane + 64 * h;
                    if (q < nq) *(f32x4*)(dst + 4 + 4 * q) = v[t][h];
                }
                if (lane == 0) dst[3] = 0.f;          // ix = -1 (W is even: no patch reaches ix = W)
            }
        }
    } else {
        for (int rc = wave; rc < IR * 3; rc += 4) {
            const int r = rc / 3, ci = rc - r * 3, iy = iy0 + r;
            const bool row_ok = iy >= 0 && iy < H;
            const float* src = xb + ((long)ci * H + (row_ok ? iy : 0)) * W - 1;
            float* dst = st_lds + rc * WP;
            for (int j = lane; j < WP; j += 64) dst[j] = (row_ok && j >= 1 && j <= W) ? src[j] : 0.f;
        }
    }
    f32x2 wr[27][2];                                  // this lane's four output channels of every patch column
#pragma unroll
    for (int kk = 0; kk < 27; ++kk) {
        const f32x4 q = *(const f32x4*)(w + kk * 32 + cg * 4);
        wr[kk][0] = f32x2{q[0], q[1]};
        wr[kk][1] = f32x2{q[2], q[3]};
    }
    __syncthreads();
    double as[4] = {0.0, 0.0, 0.0, 0.0}, aq[4] = {0.0, 0.0, 0.0, 0.0};
    const int npix = ROWS * Wo;
    int ro = 0, ox = tid >> 3;                        // pixel p = ro * Wo + ox, stepped by 32 without divisions
    while (ox >= Wo) { ox -= Wo; ++ro; }
    for (int p = tid >> 3; p < npix; p += 32) {
        if (oy0 + ro >= Ho) break;                    // (rows are walked in order: every later pixel of this thread is outside too)
        const float* base = st_lds + (2 * ro) * 3 * WP + 2 * ox + (PADL - 1);
        f32x2 acc[3][2];                              // one chain per filter row (nine terms each), summed in row order
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            acc[ky][0] = f32x2{0.f, 0.f}; acc[ky][1] = f32x2{0.f, 0.f};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float v = base[(ky * 3 + ci) * WP + kx];
                    const f32x2 vv = {v, v};
                    const int kk = (ky * 3 + kx) * 3 + ci;
                    acc[ky][0] = __builtin_elementwise_fma(vv, wr[kk][0], acc[ky][0]);
                    acc[ky][1] = __builtin_elementwise_fma(vv, wr[kk][1], acc[ky][1]);
                }
        }
        const f32x2 lo = (acc[0][0] + acc[1][0]) + acc[2][0], hi = (acc[0][1] + acc[1][1]) + acc[2][1];
        const f32x4 out = {lo[0], lo[1], hi[0], hi[1]};
        *(f32x4*)(y + (((long)b * Ho + oy0 + ro) * Wo + ox) * 32 + cg * 4) = out;
        if (bn_part) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double v = (double)out[e];
                as[e] += v;
                aq[e] = fma(v, v, aq[e]);
            }
        }
        ox += 32;
        while (ox >= Wo) { ox -= Wo; ++ro; }
    }
    if (!bn_part) return;
    // lanes with the same lane & 7 hold the same four channels: fixed-order butterfly over the other lane bits, then the waves
#pragma unroll
    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) { as[e] += __shfl_xor(as[e], o, 64); aq[e] += __shfl_xor(aq[e], o, 64); }
    __shared__ double red[4][32][2];
    if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[wave][cg * 4 + e][0] = as[e]; red[wave][cg * 4 + e][1] = aq[e]; }
    }
    __syncthreads();
    if (tid < 64) {
        const int c = tid >> 1, h = tid & 1;
        bn_part[((long)blockIdx.x * 32 + c) * 2 + h] = ((red[0][c][h] + red[1][c][h]) + red[2][c][h]) + red[3][c][h];
    }
}
static constexpr int STEM_ROWS = 2;
extern "C" int scouter_stem_direct_partial_rows(int B, int H) {
    const int Ho = (H + 2 - 3) / 2 + 1;
    return B * ((Ho + STEM_ROWS - 1) / STEM_ROWS);
}
extern "C" int scouter_stem_direct_fwd_f32(const float* x, const float* w, float* y, double* bn_partial, int B, int H, int W,
                                           int Cout, void* stream) {
    SC_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0, "stem_direct_fwd: bad arguments");
    SC_UNSUPPORTED(Cout == 32, "stem_direct_fwd: the direct kernel is built for 3 -> 32 channels (got Cout = %d)", Cout);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int rblocks = (Ho + STEM_ROWS - 1) / STEM_ROWS;
    const long grid = (long)B * rblocks;
    const bool vec4 = W % 4 == 0 && W <= 512 && ((uintptr_t)x & 15) == 0;
    const size_t lds = (size_t)(2 * STEM_ROWS + 1) * 3 * (vec4 ? W + 4 : 2 * Wo + 1) * sizeof(float);
    SC_UNSUPPORTED(lds <= 60 * 1024 && grid < (1L << 31) && (long)B * Ho * Wo * 32 < (1L << 40),
                   "stem_direct_fwd: image too wide for the row stage (W = %d)", W);
    hipStream_t st = (hipStream_t)stream;
    ScProfScope prof("stem_fwd<direct>", st, 2.0 * B * Ho * Wo * 27 * 32, 4.0 * ((double)B * 3 * H * W + (double)B * Ho * Wo * 32));
    if (vec4)
        hipLaunchKernelGGL((stem_direct_kernel<STEM_ROWS, true>), dim3((unsigned)grid), dim3(256), lds, st, x, w, y, bn_partial, H,
                           W, Ho, Wo, rblocks);
    else
        hipLaunchKernelGGL((stem_direct_kernel<STEM_ROWS, false>), dim3((unsigned)grid), dim3(256), lds, st, x, w, y, bn_partial, H,
                           W, Ho, Wo, rblocks);
    return sc_check_launch("stem_direct_fwd");
}
extern "C" int scouter_pad_rows_f32(const float* w, float* wpad, long nvalid, long ntotal, void* stream) {
    hipLaunchKernelGGL(pad_rows_kernel, dim3(sc_cdiv(ntotal, 256)), dim3(256), 0, (hipStream_t)stream, w, wpad, nvalid, ntotal);
    return sc_check_launch("pad_rows");
}

// ---------------------------------------------------------------------------------------------------------------
// sine positional encoding, token-major table pe[n = y*w + x][d]   (position_encode.py:26-46, fp32 like the reference)
// ---------------------------------------------------------------------------------------------------------------
__global__ void posenc_sine_kernel(float* __restrict__ pe, int h, int w, int d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h * w * d) return;
    const int c = i % d, n = i / d, y = n / w, x = n % w, npf = d / 2;
    const float eps = 1e-6f, scale = 6.283185307179586f;
    const bool from_y = c < npf;
    const int k = from_y ? c : c - npf;
    const float embed = from_y ? (float)(y + 1) / ((float)h + eps) * scale : (float)(x + 1) / ((float)w + eps) * scale;
    const float dim_t = powf(10000.f, (float)(2 * (k / 2)) / (float)npf);
    const float v = embed / dim_t;
    pe[i] = (k & 1) ? cosf(v) : sinf(v);
}
extern "C" int scouter_posenc_sine_f32(float* pe, int h, int w, int d, void* stream) {
    SC_REQUIRE(pe && h > 0 && w > 0 && d > 0 && d % 2 == 0, "posenc_sine: bad arguments");
    hipLaunchKernelGGL(posenc_sine_kernel, dim3(sc_cdiv((long)h * w * d, 256)), dim3(256), 0, (hipStream_t)stream, pe, h, w, d);
    return sc_check_launch("posenc_sine");
}

// ---------------------------------------------------------------------------------------------------------------
// loss head: log_softmax + NLL(mean) + lambda * area**power, top-1 count; and its backward.
//   stats[0]=loss  [1]=nll  [2]=area**power  [3]=#correct/B  [4]=area (mean of A_T)  [5]=#labels outside [0, C)
// A label outside [0, C) (num_classes / dataset mismatch) is never dereferenced: it makes loss and nll NaN, its row of
// dlogits NaN, and is counted in stats[5] so the host can raise like F.nll_loss does (engine.calculation checks it).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void slot_loss_fwd_kernel(const float* __restrict__ logits,
                                                            const long* __restrict__ labels,
                                                            const float* __restrict__ area_part, int n_part,
                                                            int B, int C, double area_count, float lambda,
                                                            float power,
                                                            float* __restrict__ logp, float* __restrict__ stats) {
    __shared__ float red[3][256];
    float nll = 0.f, corr = 0.f, bad = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* row = logits + (long)b * C;
        float m = -INFINITY;
        int am = 0;
        for (int c = 0; c < C; ++c) if (row[c] > m) { m = row[c]; am = c; }
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(row[c] - m);
        const float lse = m + logf(s);
        for (int c = 0; c < C; ++c) logp[(long)b * C + c] = row[c] - lse;
        if (labels) {
            const long y = labels[b];
            if (y >= 0 && y < C) {
                nll -= row[y] - lse;
                corr += (am == (int)y) ? 1.f : 0.f;
            } else {
                nll = __builtin_nanf("");
                bad += 1.f;
            }
        }
    }
    red[0][threadIdx.x] = nll; red[1][threadIdx.x] = corr; red[2][threadIdx.x] = bad;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
            red[2][threadIdx.x] += red[2][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && stats) {
        double asum = 0.0;                      // per-image partial sums added in a fixed order
        for (int k = 0; k < n_part; ++k) asum += area_part[k];
        const float area = area_part ? (float)(asum / area_count) : 0.f;
        const float term = area_part ? powf(area, power) : 0.f;
        const float n = labels ? red[0][0] / B : 0.f;
        stats[0] = n + lambda * term;
        stats[1] = n;
        stats[2] = term;
        stats[3] = labels ? red[1][0] / B : 0.f;
        stats[4] = area;
        stats[5] = red[2][0];
        stats[6] = 0.f;                          // (all eight words written: the caller hands over an uninitialised buffer)
        stats[7] = 0.f;
    }
}
// upstream grads (device scalars, NULL = 0): g_loss, g_nll, g_term ; g_logp [B][C] or NULL
//   dlogits = dlogp - softmax * rowsum(dlogp),  dlogp = g_logp - (g_loss+g_nll)/B * onehot(y)
//   g_area_sum = (lambda*g_loss + g_term) * power * area**(power-1) / area_count
__global__ __launch_bounds__(256) void slot_loss_bwd_kernel(const float* __restrict__ logp,
                                                            const long* __restrict__ labels,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ g_loss,
                                                            const float* __restrict__ g_nll,
                                                            const float* __restrict__ g_term,
                                                            const float* __restrict__ g_logp, int B, int C,
                                                            double area_count, float lambda, float power,
                                                            float* __restrict__ dlogits, float* __restrict__ g_area_sum) {
    const float gl = g_loss ? g_loss[0] : 0.f, gn = g_nll ? g_nll[0] : 0.f, gt = g_term ? g_term[0] : 0.f;
    const float wn = labels ? (gl + gn) / B : 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        float rs = 0.f;
        const long yl = labels ? labels[b] : -1;
        const int y = (yl >= 0 && yl < C) ? (int)yl : -1;
        if (labels && y < 0) {                  // out-of-range label: poisoned row, nothing dereferenced
            for (int c = 0; c < C; ++c) dlogits[(long)b * C + c] = __builtin_nanf("");
            continue;
        }
        for (int c = 0; c < C; ++c) {
            float d = g_logp ? g_logp[(long)b * C + c] : 0.f;
            if (c == y) d -= wn;
            rs += d;
        }
        for (int c = 0; c < C; ++c) {
            float d = g_logp ? g_logp[(long)b * C + c] : 0.f;
            if (c == y) d -= wn;
            dlogits[(long)b * C + c] = d - expf(logp[(long)b * C + c]) * rs;
        }
    }
    if (threadIdx.x == 0 && g_area_sum) {
        const float area = stats[4];
        const float dterm = lambda * gl + gt;
        const float dpow = power == 1.f ? 1.f : power * powf(area, power - 1.f);
        g_area_sum[0] = (float)((double)(dterm * dpow) / area_count);
    }
}
extern "C" int scouter_slot_loss_fwd_f32(const float* logits, const long* labels, const float* area_part,
                                         int n_area_part, int B, int C, double area_count, float lambda, float power,
                                         float* logp, float* stats, void* stream) {
    SC_REQUIRE(logits && logp && B > 0 && C > 0, "slot_loss_fwd: bad arguments");
    hipLaunchKernelGGL(slot_loss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, area_part, n_area_part, B, C, area_count, lambda, power, logp, stats);
    return sc_check_launch("slot_loss_fwd");
}
extern "C" int scouter_slot_loss_bwd_f32(const float* logp, const long* labels, const float* stats, const float* g_loss,
                                         const float* g_nll, const float* g_term, const float* g_logp, int B, int C,
                                         double area_count, float lambda, float power, float* dlogits,
                                         float* g_area_sum, void* stream) {
    SC_REQUIRE(logp && dlogits && B > 0 && C > 0, "slot_loss_bwd: bad arguments");
    hipLaunchKernelGGL(slot_loss_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logp, labels, stats, g_loss, g_nll, g_term, g_logp, B, C, area_count, lambda, power, dlogits, g_area_sum);
    return sc_check_launch("slot_loss_bwd");
}

// ---------------------------------------------------------------------------------------------------------------
// fused multi-tensor AdamW (torch.optim.AdamW semantics: decoupled weight decay, bias-corrected moments)
// chunk table entry i: param pointer, offset of the chunk inside the flat grad / moment arenas, length
// ---------------------------------------------------------------------------------------------------------------
struct AdamChunk { float* p; long off; int n; int pad; };
template <bool DEV>
__global__ __launch_bounds__(256) void adamw_kernel(const AdamChunk* __restrict__ chunks, const float* __restrict__ grads,
                                                    float* __restrict__ m, float* __restrict__ v, float lr, float b1,
                                                    float b2, float eps, float wd, float bc1, float rsqrt_bc2,
                                                    const float* __restrict__ dyn) {
    const AdamChunk ch = chunks[blockIdx.x];
    if (DEV) {                                  // graph-captured launch: lr and the step count come from device memory
        lr = dyn[0];
        const double t = (double)dyn[1];
        bc1 = (float)(1.0 - pow((double)b1, t));
        rsqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, t)));
    }
    const float step = lr / bc1, decay = 1.f - lr * wd;
    auto upd = [&](float g, float& p, float& mi, float& vi) {
        p *= decay;
        mi = b1 * mi + (1.f - b1) * g;
        vi = b2 * vi + (1.f - b2) * g * g;
        const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
        p -= step * (mi / denom);
    };
    // 16-byte accesses where the chunk allows it (the arena offsets are 16-byte aligned by construction; parameters
    // are whole torch allocations), scalar tail otherwise -- same arithmetic per element either way
    const int n4 = (((size_t)ch.p & 15) == 0 && (ch.off & 3) == 0) ? ch.n / 4 : 0;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const f32x4 g = *(const f32x4*)(grads + ch.off + 4 * i);
        f32x4 p = *(const f32x4*)(ch.p + 4 * i), mi = *(const f32x4*)(m + ch.off + 4 * i), vi = *(const f32x4*)(v + ch.off + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = p[e], me = mi[e], ve = vi[e];
            upd(g[e], pe, me, ve);
            p[e] = pe; mi[e] = me; vi[e] = ve;
        }
        *(f32x4*)(m + ch.off + 4 * i) = mi;
        *(f32x4*)(v + ch.off + 4 * i) = vi;
        *(f32x4*)(ch.p + 4 * i) = p;
    }
    for (int i = 4 * n4 + threadIdx.x; i < ch.n; i += 256) {
        float p = ch.p[i], mi = m[ch.off + i], vi = v[ch.off + i];
        upd(grads[ch.off + i], p, mi, vi);
        m[ch.off + i] = mi;
        v[ch.off + i] = vi;
        ch.p[i] = p;
    }
}
extern "C" int scouter_adamw_step_f32(const void* chunk_table, int nchunks, const float* grads, float* exp_avg,
                                      float* exp_avg_sq, float lr, float beta1, float beta2, float eps,
                                      float weight_decay, int step, void* stream) {
    SC_REQUIRE(chunk_table && grads && exp_avg && exp_avg_sq && nchunks > 0 && step > 0, "adamw_step: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adamw_kernel<false>, dim3(nchunks), dim3(256), 0, (hipStream_t)stream,
                       (const AdamChunk*)chunk_table, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay,
                       (float)bc1, (float)(1.0 / sqrt(bc2)), (const float*)nullptr);
    return sc_check_launch("adamw_step");
}
extern "C" int scouter_adamw_step_dev_f32(const void* chunk_table, int nchunks, const float* grads, float* exp_avg,
                                          float* exp_avg_sq, const float* dyn, float beta1, float beta2, float eps,
                                          float weight_decay, void* stream) {
    SC_REQUIRE(chunk_table && grads && exp_avg && exp_avg_sq && dyn && nchunks > 0, "adamw_step_dev: bad arguments");
    hipLaunchKernelGGL(adamw_kernel<true>, dim3(nchunks), dim3(256), 0, (hipStream_t)stream,
                       (const AdamChunk*)chunk_table, grads, exp_avg, exp_avg_sq, 0.f, beta1, beta2, eps, weight_decay,
                       1.f, 1.f, dyn);
    return sc_check_launch("adamw_step_dev");
}
extern "C" int scouter_adamw_chunk_bytes(void) { return (int)sizeof(AdamChunk); }

// ---------------------------------------------------------------------------------------------------------------
// small dense classifier (nn.Linear with few outputs: the FC baseline `use_slot=False`, slot_model.py:116-125 /
// timm resnet.py:503-509).  B x K x C is tiny (70 x 2048 x 10): one wave per output element, wave-shuffle reduce.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int B, int K, int C) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= B * C) return;
    const int b = o / C, c = o % C;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += x[(long)b * K + k] * w[(long)c * K + k];
    s = wave_sum(s);
    if (lane == 0) y[o] = s + (bias ? bias[c] : 0.f);
}
// dx[b][k] = sum_c dy[b][c] w[c][k]
__global__ __launch_bounds__(256) void linear_small_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                 float* __restrict__ dx, int B, int K, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * K) return;
    const int b = (int)(i / K), k = (int)(i % K);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += dy[(long)b * C + c] * w[(long)c * K + k];
    dx[i] = s;
}
// dw[c][k] = sum_b dy[b][c] x[b][k] ; db[c] = sum_b dy[b][c]   (fixed summation order -> deterministic)
__global__ __launch_bounds__(256) void linear_small_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 float* __restrict__ dw, float* __restrict__ db, int B,
                                                                 int K, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)C * K) {
        const int c = (int)(i / K), k = (int)(i % K);
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dy[(long)b * C + c] * x[(long)b * K + k];
        dw[i] = s;
    }
    if (db && i < C) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dy[(long)b * C + i];
        db[i] = s;
    }
}
extern "C" int scouter_linear_small_fwd_f32(const float* x, const float* w, const float* bias, float* y, int B, int K,
                                            int C, void* stream) {
    SC_REQUIRE(x && w && y && B > 0 && K > 0 && C > 0, "linear_small_fwd: bad arguments");
    hipLaunchKernelGGL(linear_small_fwd_kernel, dim3(sc_cdiv((long)B * C, 4)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, B, K, C);
    return sc_check_launch("linear_small_fwd");
}
extern "C" int scouter_linear_small_bwd_f32(const float* dy, const float* x, const float* w, float* dx, float* dw,
                                            float* db, int B, int K, int C, void* stream) {
    SC_REQUIRE(dy && x && w && B > 0 && K > 0 && C > 0, "linear_small_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (dx) hipLaunchKernelGGL(linear_small_dgrad_kernel, dim3(sc_cdiv((long)B * K, 256)), dim3(256), 0, st, dy, w, dx, B, K, C);
    if (dw) hipLaunchKernelGGL(linear_small_wgrad_kernel, dim3(sc_cdiv((long)C * K, 256)), dim3(256), 0, st, dy, x, dw, db, B, K, C);
    return sc_check_launch("linear_small_bwd");
}


// num_batches_tracked += 1 of EVERY train-mode BatchNorm of a forward in one launch (the counters are views of one flat
// int64 buffer, SlotModel._bump_tracked; reference: torch.nn.BatchNorm2d.forward, nn/modules/batchnorm.py `self.num_batches_tracked.add_(1)`)
__global__ __launch_bounds__(256) void iadd_i64_kernel(long* __restrict__ p, long n, long v) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] += v;
}
extern "C" int scouter_iadd_i64(long* p, long n, long v, void* stream) {
    SC_REQUIRE(p && n > 0, "iadd_i64: bad arguments");
    hipLaunchKernelGGL(iadd_i64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, n, v);
    return sc_check_launch("iadd_i64");
}
