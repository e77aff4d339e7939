// Small-S instantiation of the fused xSlot backward: S <= 16 slots per image, N <= 64 tokens (NTW = 1: the metric's own head,
// 10 classes x 1 slot on the 7 x 7 grid) or N <= 96 (NTW = 2: the reference's default 9 x 9 grid, train.py:39: waves 0 and 1
// walk two token tiles; the dK / dX^a sums then wait in a wave-private global park between the iterations instead of in
// registers); sloter/utils/slot_attention.py:44-96.  Included by xslot_bwd.hip; same arguments,
// same outputs as xslot_bwd_kernel; the maths is oracle/xslot_manual.py.  Register layouts and the division of labour
// between the four waves are those of xslot_small_fwd.h:
//   * slot-per-lane-16: lane (i = lane & 15, g = lane >> 4), register r of tile t <-> M^T[c = 16 t + 4 g + r][i];
//     token-per-lane: the same with a token on the lane (to_k MLP backward, dK and dX^a accumulators);
//   * wave w owns the token tile 16 w .. 16 w + 15 (QK^T, AV, dA, the token contraction of ds_t, and -- after a 16 x 16
//     bounce through a private LDS tile that puts the slot index on the MFMA k axis -- the slot contractions
//     dK += dD^T s_t / 8 and dX^a += A^T dU / d, accumulated in registers over the iterations) and the hidden units
//     16 w .. 16 w + 15 of the GRU, forward recomputation and backward: its W_ih / W_hh fragments stay in registers in
//     BOTH operand orientations (rows = gate: 96 registers; rows = input channel: 96 registers) for the whole kernel;
//   * contractions split over the waves (U and ds_t over tokens, dU and dh over gate rows) are summed through LDS in a
//     fixed order: four hand-offs per iteration, each on its own buffer so that one barrier per hand-off suffices;
//   * r_i, tau and c0 are computed redundantly by every wave (no exchange); g_i needs one 16-float exchange.
// MFMA chain per wave: 288 per GRU iteration, 96 for the last one, 64 per to_k layer, 32 cycles each.
#pragma once
#include "xslot_small_common.h"

#define XS16_LDT 20            // row stride (floats) of the 16 x 16 bounce tiles: 16-byte rows
static size_t xs_small_bwd_lds_bytes(int NTW) {
    const int NPR = NTW == 1 ? 64 : 96;
    return (size_t)((2 * NPR + 64) * XS_LD + 3 * XS16_UX_FLOATS + 2 * 16 * XS_LD + 4 * 2 * 16 * XS16_LDT + 64) * sizeof(float) +
           (size_t)(64 * 4 + 64) * sizeof(double);
}
static size_t xs_small_bwd_park_bytes(int B, int NTW) {       // (inside scouter_xslot_bwd_workspace_bytes' in-loop figure)
    return NTW == 1 ? 0 : (size_t)B * 4 * NTW * 2 * 4 * 64 * sizeof(f32x4);
}

template <int NTW>
__global__ __launch_bounds__(256) void xslot_small_bwd_kernel(XsBwdArgs a) {
    constexpr int NPR = NTW == 1 ? 64 : 96;         // token rows held in LDS (16-token tiles: 4 or 6)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    double* part = (double*)lds;                    // [64][4]
    double* ksum_s = part + 256;                    // [64]
    float* Xs = (float*)(ksum_s + 64);              // [NPR][68] X (rows >= N zero)
    float* Ks = Xs + NPR * XS_LD;                   // [NPR][68] K (rows >= N zero)
    float* uxA = Ks + NPR * XS_LD;                  // [4][16][68] partial U of the four waves
    float* uxB = uxA + XS16_UX_FLOATS;              // partial dU
    float* uxC = uxB + XS16_UX_FLOATS;              // partial ds_t
    float* sfull = uxC + XS16_UX_FLOATS;            // [16][68] s_t, row = slot
    float* dufull = sfull + 16 * XS_LD;             // [16][68] dU_t, row = slot
    float* tb = dufull + 16 * XS_LD;                // [4 waves][2][16][20] bounce tiles (dD | A), row = token
    float* gx = tb + 4 * 2 * 16 * XS16_LDT;         // [4][16] partial g_i
    float* WL = gx + 64;                            // [64][68] to_k weight matrix of the MLP backward (rotates with uxA, uxB)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, m = lane & 15, g = lane >> 4;
    const int b = blockIdx.x, N = a.N, S = a.S, T = a.T;
    const float scale = 0.125f, inv_d = 1.f / XS_D;
    const bool iok = m < S;
    // token tile u of this wave = tile w + 4 u (wave-uniform: waves 2, 3 have no second tile at NTW = 2)
    int jA[NTW], jB[NTW];                           // token on this lane in the token-per-lane / A-operand layouts; first of this
    bool uon[NTW];                                  // lane's four tokens in the accumulator layout
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        uon[u] = 16 * (w + 4 * u) < NPR;
        jA[u] = uon[u] ? 16 * (w + 4 * u) + m : 0;
        jB[u] = uon[u] ? 16 * (w + 4 * u) + 4 * g : 0;
    }
    f32x4* park = (f32x4*)a.ws + ((long)b * 4 + w) * (NTW * 2 * 4 * 64) + lane;      // [u][dK | dX][ct][lane] (NTW = 2)

    XS16_STAMP_INIT();
    XS16_STAMP();                           // 0: start
    // ---- staging: X and K of the image (LDS), this wave's GRU fragments (registers)
    {
        // (rows beyond N: the load goes to the last valid row and the value is dropped -- no predicated loads, see
        //  xslot_small_fwd.h)
        constexpr int XQ = NPR / 16;
        f32x4 xv[XQ], kv[XQ], wv[4];
#pragma unroll
        for (int k = 0; k < XQ; ++k) {
            const int c = tid + k * 256, r = min(c >> 4, N - 1), q = c & 15;
            xv[k] = *(const f32x4*)(a.X + ((long)b * N + r) * XS_D + q * 4);
            kv[k] = *(const f32x4*)(a.Ksave + ((long)b * N + r) * XS_D + q * 4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) wv[k] = *(const f32x4*)(a.tok_w[a.L - 1] + (tid + k * 256) * 4);      // the to_k backward's first weight matrix
#pragma unroll
        for (int k = 0; k < XQ; ++k) {
            const int c = tid + k * 256, r = c >> 4, q = c & 15;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *(f32x4*)(Xs + r * XS_LD + q * 4) = r < N ? xv[k] : z;
            *(f32x4*)(Ks + r * XS_LD + q * 4) = r < N ? kv[k] : z;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = tid + k * 256;
            *(f32x4*)(WL + (c >> 4) * XS_LD + (c & 15) * 4) = wv[k];
        }
    }
    f32x4 Wg[3][2][4];                  // Wg[gate][ih | hh][t] = W[64 gate + 16 w + m][16 t + 4 g ..]       (rows = gate)
    float WTi[3][4][4], WTh[3][4][4];   // WT[gate][ct][r] = W[64 gate + 16 w + 4 g + r][16 ct + m]            (rows = channel)
    f32x4 bg[4];
    if (T > 1) {
#pragma unroll
        for (int G = 0; G < 3; ++G)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                Wg[G][0][t] = *(const f32x4*)(a.w_ih + (64 * G + 16 * w + m) * XS_D + 16 * t + 4 * g);
                Wg[G][1][t] = *(const f32x4*)(a.w_hh + (64 * G + 16 * w + m) * XS_D + 16 * t + 4 * g);
            }
#pragma unroll
        for (int G = 0; G < 3; ++G)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    WTi[G][ct][r] = a.w_ih[(64 * G + 16 * w + 4 * g + r) * XS_D + 16 * ct + m];
                    WTh[G][ct][r] = a.w_hh[(64 * G + 16 * w + 4 * g + r) * XS_D + 16 * ct + m];
                }
        const int hb = 16 * w + 4 * g;
        bg[0] = *(const f32x4*)(a.b_ih + hb) + *(const f32x4*)(a.b_hh + hb);
        bg[1] = *(const f32x4*)(a.b_ih + 64 + hb) + *(const f32x4*)(a.b_hh + 64 + hb);
        bg[2] = *(const f32x4*)(a.b_ih + 128 + hb);
        bg[3] = *(const f32x4*)(a.b_hh + 128 + hb);
    }
    const float g_area = a.g_area_sum ? a.g_area_sum[0] : 0.f;
    __syncthreads();
    XS16_STAMP();                           // 1: staged
    {   // column sums of K in fp64 (the normaliser: see xs_rowdot_f64)
        const int c = tid & 63, q = tid >> 6;
        double s = 0.0;
        for (int j = q; j < NPR; j += 4) s += (double)Ks[j * XS_LD + c];
        part[c * 4 + q] = s;
    }
    xs_lds_barrier();
    if (tid < 64) ksum_s[tid] = (part[tid * 4] + part[tid * 4 + 1]) + (part[tid * 4 + 2] + part[tid * 4 + 3]);
    xs_lds_barrier();

    XS16_STAMP();                           // 2: column sums
    f32x4 dKacc[4], dXacc[4], Pds[4];       // token-per-lane sums over the iterations (NTW = 1; parked in global memory
                                            // otherwise); dL/ds_t handed down the iterations
#pragma unroll
    for (int t = 0; t < 4; ++t) { dKacc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dXacc[t] = dKacc[t]; Pds[t] = dKacc[t]; }
    float* tbd = tb + w * (2 * 16 * XS16_LDT);
    float* tba = tbd + 16 * XS16_LDT;

    // s_t of the next iteration is requested one iteration ahead (an L2 round trip per iteration otherwise)
    auto state_ptr = [&](int it) {
        return (it == 0 ? a.slots0 : a.states + ((long)(it - 1) * a.B + b) * S * XS_D) + min(m, S - 1) * XS_D + 4 * g;
    };
    f32x4 Pn[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) Pn[t] = *(const f32x4*)(state_ptr(T - 1) + 16 * t);
    for (int it = T - 1; it >= 0; --it) {
        const bool last = it == T - 1;
        f32x4 Ps[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) Ps[t] = iok ? Pn[t] : f32x4{0.f, 0.f, 0.f, 0.f};
        if (it > 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) Pn[t] = *(const f32x4*)(state_ptr(it - 1) + 16 * t);
        }
        *(f32x4*)(sfull + m * XS_LD + 16 * w + 4 * g) = xs16_pick(Ps, w);
        // ---- r_i, tau: the forward's fp64 normaliser, redundantly in every wave
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double2 k01 = *(const double2*)(ksum_s + 16 * t + 4 * g);
            const double2 k23 = *(const double2*)(ksum_s + 16 * t + 4 * g + 2);
            a0 += (double)Ps[t][0] * k01.x;
            a1 += (double)Ps[t][1] * k01.y;
            a0 += (double)Ps[t][2] * k23.x;
            a1 += (double)Ps[t][3] * k23.y;
        }
        double r64 = xs16_gsum_f64(a0 + a1) * (double)scale;
        if (!iok) r64 = 0.0;
        const float tau = (float)xs16_rowsum_f64(r64);
        const float rr = iok ? (float)r64 : 1.f;
        const float ir = xs_recip(rr);
        XS16_STAMP();                       // it.0: s_t loaded, r_i, tau
        // ---- recomputation: D^T = (K / 8) s^T, A = sigmoid(D / r_i * tau), partial U^T = (X / 64)^T A^T
        f32x4 D[NTW], A[NTW], Up[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) Up[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NTW; ++u) {
            D[u] = f32x4{0.f, 0.f, 0.f, 0.f}; A[u] = D[u];
            if (!uon[u]) continue;
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
            f32x4 kr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) kr[t] = *(const f32x4*)(Ks + jA[u] * XS_LD + 16 * t + 4 * g) * scale;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    d0 = mfma16(kr[t][r], Ps[t][r], d0);
                    d1 = mfma16(kr[t + 2][r], Ps[t + 2][r], d1);
                }
            D[u] = d0 + d1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = xs_sigmoid(xs_div(D[u][r], rr, ir) * tau);
                A[u][r] = (iok && jB[u] + r < N) ? v : 0.f;
            }
            float xa[4][4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) xa[ct][r] = Xs[(jB[u] + r) * XS_LD + 16 * ct + m] * inv_d;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) Up[ct] = mfma16(xa[ct][r], A[u][r], Up[ct]);
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) *(f32x4*)(uxA + (w * 16 + m) * XS_LD + 16 * ct + 4 * g) = Up[ct];
        f32x4 Gr, Gz, Gin, Ghn;
        if (!last) {                        // the W_hh s_t half of the gates runs between the LDS write and the barrier
            Gr = bg[0]; Gz = bg[1]; Gin = bg[2]; Ghn = bg[3];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Gr = mfma16(Wg[0][1][t][r], Ps[t][r], Gr);
                    Gz = mfma16(Wg[1][1][t][r], Ps[t][r], Gz);
                    Ghn = mfma16(Wg[2][1][t][r], Ps[t][r], Ghn);
                }
        }
        XS16_STAMP();                       // it.1: S1, A, S2, W_hh half
        xs_lds_barrier();                   // #1: U partials, s_t rows
        f32x4 PdU[4], dhp[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) dhp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!last) {
            f32x4 PU[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float* p = uxA + m * XS_LD + 16 * t + 4 * g;
                PU[t] = (*(const f32x4*)p + *(const f32x4*)(p + 16 * XS_LD)) +
                        (*(const f32x4*)(p + 32 * XS_LD) + *(const f32x4*)(p + 48 * XS_LD));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Gr = mfma16(Wg[0][0][t][r], PU[t][r], Gr);
                    Gz = mfma16(Wg[1][0][t][r], PU[t][r], Gz);
                    Gin = mfma16(Wg[2][0][t][r], PU[t][r], Gin);
                }
            XS16_STAMP();                   // it.2: barrier, U sum, W_ih half
            // ---- GRU backward of this wave's 16 hidden units (oracle/xslot_manual.py: backward, "GRU backward")
            const f32x4 dsn = xs16_pick(Pds, w), hold = xs16_pick(Ps, w), uown = xs16_pick(PU, w);
            f32x4 da_r, da_z, da_n, da_nr, dh_el;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rg = xs_sigmoid(Gr[r]);
                const float zg = xs_sigmoid(Gz[r]);
                const float hnb = Ghn[r];
                const float ng = xs_tanh(Gin[r] + rg * hnb);
                const float ds = iok ? dsn[r] : 0.f;                       // padded slots carry nothing
                da_n[r] = ds * (1.f - zg) * (1.f - ng * ng);
                da_z[r] = ds * (hold[r] - ng) * zg * (1.f - zg);
                da_r[r] = da_n[r] * hnb * rg * (1.f - rg);
                da_nr[r] = da_n[r] * rg;
                dh_el[r] = ds * zg;
            }
            if (iok) {
                const long row = ((long)it * a.B + b) * S + m;
                float* gi = a.dgi + row * 192 + 16 * w + 4 * g;
                float* gh = a.dgh + row * 192 + 16 * w + 4 * g;
                *(f32x4*)gi = da_r; *(f32x4*)(gi + 64) = da_z; *(f32x4*)(gi + 128) = da_n;
                *(f32x4*)gh = da_r; *(f32x4*)(gh + 64) = da_z; *(f32x4*)(gh + 128) = da_nr;
                *(f32x4*)(a.Usave + row * XS_D + 16 * w + 4 * g) = uown;
            }
            // partial dU^T = W_ih^T dgi^T, partial dh^T = W_hh^T dgh^T over this wave's 48 gate rows
            f32x4 dUp[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) { dUp[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; if (ct == w) dhp[ct] = dh_el; }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    dUp[ct] = mfma16(WTi[0][ct][r], da_r[r], dUp[ct]);
                    dhp[ct] = mfma16(WTh[0][ct][r], da_r[r], dhp[ct]);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    dUp[ct] = mfma16(WTi[1][ct][r], da_z[r], dUp[ct]);
                    dhp[ct] = mfma16(WTh[1][ct][r], da_z[r], dhp[ct]);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    dUp[ct] = mfma16(WTi[2][ct][r], da_n[r], dUp[ct]);
                    dhp[ct] = mfma16(WTh[2][ct][r], da_nr[r], dhp[ct]);
                }
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) *(f32x4*)(uxB + (w * 16 + m) * XS_LD + 16 * ct + 4 * g) = dUp[ct];
            XS16_STAMP();                   // it.3: gate gradients, stores, dU / dh partials
            xs_lds_barrier();               // #2: dU partials
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float* p = uxB + m * XS_LD + 16 * t + 4 * g;
                PdU[t] = (*(const f32x4*)p + *(const f32x4*)(p + 16 * XS_LD)) +
                         (*(const f32x4*)(p + 32 * XS_LD) + *(const f32x4*)(p + 48 * XS_LD));
            }
        } else {
            // logits_c = ls * sum_{s in c} sum_k U_T[s][k]  ->  dU_T[s][k] = ls * dlogits[c(s)]
            const float du = iok ? a.loss_status * a.dlogits[(long)b * a.C + m / a.spc] : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) PdU[t] = f32x4{du, du, du, du};
        }
        XS16_STAMP();                       // it.4: barrier, dU sum
        *(f32x4*)(dufull + m * XS_LD + 16 * w + 4 * g) = xs16_pick(PdU, w);
        // ---- dA^T = (X / 64) dU^T (+ area term), G = dA * A (1 - A), g_i = sum_j G_ij D_ij
        f32x4 Gm[NTW];
        float gsum = 0.f;
#pragma unroll
        for (int u = 0; u < NTW; ++u) {
            Gm[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!uon[u]) continue;
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
            f32x4 xq[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) xq[t] = *(const f32x4*)(Xs + jA[u] * XS_LD + 16 * t + 4 * g) * inv_d;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    d0 = mfma16(xq[t][r], PdU[t][r], d0);
                    d1 = mfma16(xq[t + 2][r], PdU[t + 2][r], d1);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = (d0[r] + d1[r]) + (last ? g_area : 0.f);
                Gm[u][r] = (iok && jB[u] + r < N) ? v * A[u][r] * (1.f - A[u][r]) : 0.f;
                gsum += Gm[u][r] * D[u][r];
            }
        }
        gsum = xs16_gsum(gsum);
        if (g == 0) gx[w * 16 + m] = gsum;
        XS16_STAMP();                       // it.5: dA, G, g_i partial
        xs_lds_barrier();                   // #3: g_i partials (and the dU rows)
        const float gi_ = (gx[m] + gx[16 + m]) + (gx[32 + m] + gx[48 + m]);
        const float c0 = xs16_rowsum(iok ? gi_ / rr : 0.f);
        // ---- dD = G tau / r_i - g_i tau / r_i^2 + c0 ; partial ds_t^T = (K / 8)^T dD^T (+ this wave's dh partial)
        f32x4 dD[NTW];
#pragma unroll
        for (int u = 0; u < NTW; ++u) {
            dD[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!uon[u]) continue;
            const float k1 = tau / rr, k2 = gi_ * tau / (rr * rr);
#pragma unroll
            for (int r = 0; r < 4; ++r) dD[u][r] = (iok && jB[u] + r < N) ? Gm[u][r] * k1 - k2 + c0 : 0.f;
            float kt[4][4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) kt[ct][r] = Ks[(jB[u] + r) * XS_LD + 16 * ct + m] * scale;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) dhp[ct] = mfma16(kt[ct][r], dD[u][r], dhp[ct]);
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) *(f32x4*)(uxC + (w * 16 + m) * XS_LD + 16 * ct + 4 * g) = dhp[ct];
        XS16_STAMP();                       // it.6: barrier, dD, ds partial
        // ---- contractions over the slot index: dK += (s_t / 8)^T dD, dX^a += (dU / 64)^T A, slots on the MFMA k axis
        {
            float sa[4][4], da[4][4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sa[ct][e] = sfull[(4 * g + e) * XS_LD + 16 * ct + m] * scale;
                    da[ct][e] = dufull[(4 * g + e) * XS_LD + 16 * ct + m] * inv_d;
                }
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                if (!uon[u]) continue;
                if constexpr (NTW > 1) {        // this tile's sums so far come back from the park (zero in the first iteration)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        dKacc[ct] = last ? f32x4{0.f, 0.f, 0.f, 0.f} : park[((u * 2 + 0) * 4 + ct) * 64];
                        dXacc[ct] = last ? f32x4{0.f, 0.f, 0.f, 0.f} : park[((u * 2 + 1) * 4 + ct) * 64];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    tbd[(4 * g + r) * XS16_LDT + m] = dD[u][r];
                    tba[(4 * g + r) * XS16_LDT + m] = A[u][r];
                }
                const f32x4 bd = *(const f32x4*)(tbd + m * XS16_LDT + 4 * g);        // dD[slot 4 g + e][token m]
                const f32x4 ba = *(const f32x4*)(tba + m * XS16_LDT + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        dKacc[ct] = mfma16(sa[ct][e], bd[e], dKacc[ct]);
                        dXacc[ct] = mfma16(da[ct][e], ba[e], dXacc[ct]);
                    }
                if constexpr (NTW > 1) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        park[((u * 2 + 0) * 4 + ct) * 64] = dKacc[ct];
                        park[((u * 2 + 1) * 4 + ct) * 64] = dXacc[ct];
                    }
                }
            }
        }
        XS16_STAMP();                       // it.7: slot contractions
        xs_lds_barrier();                   // #4: ds partials
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float* p = uxC + m * XS_LD + 16 * t + 4 * g;
            Pds[t] = (*(const f32x4*)p + *(const f32x4*)(p + 16 * XS_LD)) +
                     (*(const f32x4*)(p + 32 * XS_LD) + *(const f32x4*)(p + 48 * XS_LD));
        }
    }
    XS16_STAMP();                           // loop done
    if (iok) *(f32x4*)(a.ds0 + ((long)b * S + m) * XS_D + 16 * w + 4 * g) = xs16_pick(Pds, w);

    // ---- to_k MLP backward, token-per-lane: dZ_{L-1} = dK ; dH_{l-1} = dZ_l W_l ; dZ_{l-1} = dH_{l-1} * (H_l > 0).
    // The A operand W_l^T[c][o] is four ROWS x one column per lane: W_l is staged row-major in LDS (WL, then the idle hand-off
    // buffers, in rotation) and read with ds_read_b32; the next layer's matrix is requested before this layer multiplies.
    f32x4 dz[NTW][4], dxa[NTW][4];
#pragma unroll
    for (int u = 0; u < NTW; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if constexpr (NTW > 1) {
                dz[u][t] = uon[u] ? park[((u * 2 + 0) * 4 + t) * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
                dxa[u][t] = uon[u] ? park[((u * 2 + 1) * 4 + t) * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                dz[u][t] = dKacc[t];
                dxa[u][t] = dXacc[t];
            }
        }
    float* wbuf[3] = {WL, uxA, uxB};
    for (int l = a.L - 1, k = 0; l >= 0; --l, k = k == 2 ? 0 : k + 1) {
        const float* Wl = k == 0 ? wbuf[0] : k == 1 ? wbuf[1] : wbuf[2];
        float* Wnext = k == 0 ? wbuf[1] : k == 1 ? wbuf[2] : wbuf[0];
        f32x4 wv[4], hv[NTW][4];
        if (l > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) wv[q] = *(const f32x4*)(a.tok_w[l - 1] + (tid + q * 256) * 4);
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                const int jc = min(jA[u], N - 1);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    hv[u][ct] = *(const f32x4*)(a.Hsave + (((long)l * a.B + b) * N + jc) * XS_D + 16 * ct + 4 * g);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[NTW][4];
#pragma unroll
        for (int u = 0; u < NTW; ++u) {
            if (uon[u] && jA[u] < N) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    *(f32x4*)(a.dZ + (((long)l * a.B + b) * N + jA[u]) * XS_D + 16 * t + 4 * g) = dz[u][t];
            }
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[u][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float wt[4][4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) wt[ct][r] = Wl[(16 * t + 4 * g + r) * XS_LD + 16 * ct + m];
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                if (!uon[u]) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[u][ct] = mfma16(wt[ct][r], dz[u][t][r], acc[u][ct]);
            }
        }
        if (l > 0) {
#pragma unroll
            for (int u = 0; u < NTW; ++u)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dz[u][ct][r] = (uon[u] && jA[u] < N && hv[u][ct][r] > 0.f) ? acc[u][ct][r] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = tid + q * 256;
                *(f32x4*)(Wnext + (c >> 4) * XS_LD + (c & 15) * 4) = wv[q];
            }
            xs_lds_barrier();
        } else {
#pragma unroll
            for (int u = 0; u < NTW; ++u)
                if (uon[u] && jA[u] < N) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
                        *(f32x4*)(a.dX + ((long)b * N + jA[u]) * XS_D + 16 * ct + 4 * g) = acc[u][ct] + dxa[u][ct];
                }
        }
    }
    XS16_STAMP();                           // end
}
