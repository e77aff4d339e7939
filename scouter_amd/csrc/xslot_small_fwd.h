// Small-S instantiation of the fused xSlot forward: S <= 16 slots per image (the metric's own head: 10 classes x 1 slot,
// sloter/utils/slot_attention.py:44-96, recipe README.md:39-42), N <= 64 * NTW tokens.  Included by xslot_fwd.hip; same
// arguments, same outputs as xslot_fwd_kernel.
//
// Why a second kernel: with <= 16 slots the 32-slot tiles of xslot_fwd_kernel leave ONE wave pair of the workgroup
// working (10 of 32 MFMA columns live) behind a start-up that stages 98 KB of GRU weights in LDS for a tile that uses
// each weight once per iteration: 57 us at 70 x 10, of which 81 k cycles are one wave's serial MFMA chain.  Here:
//   * v_mfma_f32_16x16x4_f32 tiles, slots on the 16 MFMA columns: "slot-per-lane-16" register layout
//         lane (i = lane & 15, g = lane >> 4), register r of 16-row tile t   <->   M^T[c = 16 t + 4 g + r][i]
//     -- an accumulator set in this layout IS the B operand (k = c, one register per MFMA step e = 4 t + r) of the next
//     contraction, and an A operand row-major in c is ONE 16-byte load per tile t (W[row][16 t + 4 g .. + 3]);
//   * all four waves work: wave w owns the token tile(s) 16 (w + 4 u) .. + 15 for the to_k MLP, QK^T and AV (the token
//     contraction of AV is summed over the waves through LDS in a fixed order) and the hidden units 16 w .. 16 w + 15
//     of the GRU (all three gates, both matrices: its 96 fragments of W_ih / W_hh live in REGISTERS for the whole
//     kernel, loaded straight from L2 in operand layout -- no LDS staging of weights at all);
//   * the to_k MLP runs token-per-lane (lane = token, registers = channel) from layer to layer without touching LDS:
//     Out^T[o][j] = W[o][c] In^T[c][j] has the next layer's operand layout, and the last layer's output is the A
//     operand of QK^T;
//   * r_i and tau (fp64, as in xslot_fwd_kernel) are computed redundantly by every wave from the full slot state
//     each wave holds: no exchange; two LDS hand-offs per iteration (U partial sums, new hidden units).
// MFMA chain per wave at N = 49: MLP 3 x 64, iteration 16 + 16 + 96 (last: 32) of 32 cycles each = 18 k cycles.
#pragma once

#include "xslot_small_common.h"

// GRU gate block of 4 hidden units (slot_attention.py:60-66): G = biased pre-activations
__device__ __forceinline__ f32x4 xs16_gru(const f32x4& Gr, const f32x4& Gz, const f32x4& Gin, const f32x4& Ghn,
                                          const f32x4& hold) {
    f32x4 hn;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float rg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(Gr[r] * -XS_LOG2E));
        const float zg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(Gz[r] * -XS_LOG2E));
        const float t = (Gin[r] + rg * Ghn[r]) * (2.f * XS_LOG2E);
        const float ng = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(t));
        hn[r] = ng + zg * (hold[r] - ng);
    }
    return hn;
}

template <int NTW>
__global__ __launch_bounds__(256) void xslot_small_fwd_kernel(XsFwdArgs a) {
    constexpr int NP = 64 * NTW;
    __shared__ __attribute__((aligned(16))) float pool[NP * XS_LD];      // K / 8 for the column sums, then the U partials
    __shared__ __attribute__((aligned(16))) float hx[2][16 * XS_LD];     // new slot state, double-buffered
    __shared__ __attribute__((aligned(16))) double part[64][4];
    __shared__ __attribute__((aligned(16))) double ksum_s[64];
    __shared__ float usum[16];
    __shared__ float area_s[4];
    static_assert(NP * XS_LD >= XS16_UX_FLOATS, "the U exchange buffer aliases the K staging buffer");
    float* ux = pool;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, m = lane & 15, g = lane >> 4;
    const int b = blockIdx.x, N = a.N, S = a.S;
    const float* Xg = a.X + (long)b * N * XS_D;
    const float scale = 0.125f, inv_d = 1.f / XS_D;
    const bool iok = m < S;

    XS16_STAMP_INIT();
    XS16_STAMP();                       // 0: start
    // ---- phase 0: every global operand is requested up front.  Order matters (loads return in order): what the MLP
    // needs first, the GRU weights last.
    f32x4 Q[NTW][4];                    // token-per-lane: lane (j = tile * 16 + m, g), Q[u][t][r] = H[j][16 t + 4 g + r]
    f32x4 Wm[4][4];                     // to_k layer: Wm[ot][t] = W[16 ot + m][16 t + 4 g ..]
    // (rows beyond N / S: the load goes to the last valid row and the value is dropped -- a predicated load becomes a
    //  branch with its own s_waitcnt, i.e. one exposed round trip per load: 18 k cycles of start-up, measured)
    f32x4 Qp[NTW][4];
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        const int j = min(16 * (w + 4 * u) + m, N - 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            Q[u][t] = *(const f32x4*)(Xg + j * XS_D + 16 * t + 4 * g);
            Qp[u][t] = *(const f32x4*)(a.PE + j * XS_D + 16 * t + 4 * g);
        }
    }
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int t = 0; t < 4; ++t) Wm[ot][t] = *(const f32x4*)(a.tok_w[0] + (16 * ot + m) * XS_D + 16 * t + 4 * g);
    f32x4 Ps[4];                        // slot state, slot-per-lane-16
#pragma unroll
    for (int t = 0; t < 4; ++t) Ps[t] = *(const f32x4*)(a.slots0 + min(m, S - 1) * XS_D + 16 * t + 4 * g);
    float Xa[NTW][4][4];                // AV operand: Xa[u][ct][r] = X[16 tile + 4 g + r][16 ct + m] / d
#pragma unroll
    for (int u = 0; u < NTW; ++u)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = min(16 * (w + 4 * u) + 4 * g + r, N - 1);
                Xa[u][ct][r] = Xg[j * XS_D + 16 * ct + m];
            }
    f32x4 Wg[3][2][4];                  // GRU: Wg[gate][ih | hh][t] = W[64 gate + 16 w + m][16 t + 4 g ..]
    f32x4 bg[4];                        // br | bz | b_in | b_hn of the hidden units 16 w + 4 g ..
    if (a.T > 1) {
#pragma unroll
        for (int G = 0; G < 3; ++G)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                Wg[G][0][t] = *(const f32x4*)(a.w_ih + (64 * G + 16 * w + m) * XS_D + 16 * t + 4 * g);
                Wg[G][1][t] = *(const f32x4*)(a.w_hh + (64 * G + 16 * w + m) * XS_D + 16 * t + 4 * g);
            }
        const int hb = 16 * w + 4 * g;
        bg[0] = *(const f32x4*)(a.b_ih + hb) + *(const f32x4*)(a.b_hh + hb);
        bg[1] = *(const f32x4*)(a.b_ih + 64 + hb) + *(const f32x4*)(a.b_hh + 64 + hb);
        bg[2] = *(const f32x4*)(a.b_ih + 128 + hb);
        bg[3] = *(const f32x4*)(a.b_hh + 128 + hb);
    }
    __builtin_amdgcn_sched_barrier(0);
    XS16_STAMP();                       // 0a: loads issued
    // input of layer 0 (operand of the to_k weight gradients)
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        const int j = 16 * (w + 4 * u) + m;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            Q[u][t] = j < N ? Q[u][t] + Qp[u][t] : f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < N) *(f32x4*)(a.Hsave + ((long)b * N + j) * XS_D + 16 * t + 4 * g) = Q[u][t];
        }
    }

    XS16_STAMP();                       // 1: loads issued, Hsave stored
    // ---- phase 1: to_k MLP (slot_attention.py:37-42,49), token-per-lane from layer to layer
    f32x4 Kr[NTW][4];                   // K / 8: A operand of QK^T
    for (int l = 0; l < a.L; ++l) {
        const bool lastl = l == a.L - 1;
        f32x4 Wn[4][4];
        if (!lastl) {
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int t = 0; t < 4; ++t) Wn[ot][t] = *(const f32x4*)(a.tok_w[l + 1] + (16 * ot + m) * XS_D + 16 * t + 4 * g);
        }
        f32x4 bo[4];
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) bo[ot] = *(const f32x4*)(a.tok_b[l] + 16 * ot + 4 * g);
        // (pinned: left alone the compiler sinks the next layer's loads behind this layer's MFMAs and waits for them at
        //  the top of the next layer -- an exposed L2 round trip per layer)
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[NTW][4];
#pragma unroll
        for (int u = 0; u < NTW; ++u)
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) acc[u][ot] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int u = 0; u < NTW; ++u)
#pragma unroll
                    for (int ot = 0; ot < 4; ++ot) acc[u][ot] = mfma16(Wm[ot][t][r], Q[u][t][r], acc[u][ot]);
#pragma unroll
        for (int u = 0; u < NTW; ++u) {
            const int j = 16 * (w + 4 * u) + m;
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) {
                f32x4 v = acc[u][ot] + bo[ot];
                if (lastl) {
                    if (j >= N) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    else *(f32x4*)(a.Ksave + ((long)b * N + j) * XS_D + 16 * ot + 4 * g) = v;
                    Kr[u][ot] = v * scale;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    Q[u][ot] = v;
                    if (j < N) *(f32x4*)(a.Hsave + (((long)(l + 1) * a.B + b) * N + j) * XS_D + 16 * ot + 4 * g) = v;
                }
            }
        }
        if (!lastl) {
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int t = 0; t < 4; ++t) Wm[ot][t] = Wn[ot][t];
        }
        XS16_STAMP();                   // layer l done
    }

    XS16_STAMP();                       // 2: MLP done
    // padded tokens / slots: zero operands (their loads went to the last valid row)
#pragma unroll
    for (int u = 0; u < NTW; ++u)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) Xa[u][ct][r] = 16 * (w + 4 * u) + 4 * g + r < N ? Xa[u][ct][r] * inv_d : 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (!iok) Ps[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- phase 2: column sums of K / 8 in fp64 (the normaliser r_i = s_i . sum_j K_j / 8, see xs_rowdot_f64)
#pragma unroll
    for (int u = 0; u < NTW; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) *(f32x4*)(pool + (16 * (w + 4 * u) + m) * XS_LD + 16 * t + 4 * g) = Kr[u][t];
    xs_lds_barrier();
    {
        const int c = tid & 63, q = tid >> 6;
        double s = 0.0;
        for (int j = q; j < NP; j += 4) s += (double)pool[j * XS_LD + c];
        part[c][q] = s;
    }
    xs_lds_barrier();
    if (tid < 64) ksum_s[tid] = (part[tid][0] + part[tid][1]) + (part[tid][2] + part[tid][3]);
    xs_lds_barrier();
    double ks[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            const double2 v = *(const double2*)(ksum_s + 16 * t + 4 * g + r);
            ks[t][r] = v.x;
            ks[t][r + 1] = v.y;
        }

    XS16_STAMP();                       // 3: column sums done
    // ---- phase 3: iterations
    for (int it = 0; it < a.T; ++it) {
        const bool last = it == a.T - 1;
        const long it_row = ((long)it * a.B + b) * S;
        // r_i, tau (every wave, redundantly, from the full slot state it holds)
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a0 += (double)Ps[t][0] * ks[t][0];
            a1 += (double)Ps[t][1] * ks[t][1];
            a0 += (double)Ps[t][2] * ks[t][2];
            a1 += (double)Ps[t][3] * ks[t][3];
        }
        double r64 = xs16_gsum_f64(a0 + a1);
        if (!iok) r64 = 0.0;                                              // padded slots do not enter tau
        const float tau = (float)xs16_rowsum_f64(r64);
        const float rr = (float)r64;
        const float cc = iok ? -XS_LOG2E * (tau * xs_recip(rr)) : 0.f;
        XS16_STAMP();                   // it.0: r_i, tau
        // S1: D^T = (K / 8) s^T, two partial chains per token tile (a dependent 16x16x4 chain issues every 40 cycles, not 32)
        f32x4 D[NTW][2];
#pragma unroll
        for (int u = 0; u < NTW; ++u) { D[u][0] = f32x4{0.f, 0.f, 0.f, 0.f}; D[u][1] = D[u][0]; }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int u = 0; u < NTW; ++u) {
                    D[u][0] = mfma16(Kr[u][t][r], Ps[t][r], D[u][0]);
                    D[u][1] = mfma16(Kr[u][t + 2][r], Ps[t + 2][r], D[u][1]);
                }
        // V1: A = sigmoid(D / r_i * tau)   (slot_attention.py:56-57)
        f32x4 A[NTW];
#pragma unroll
        for (int u = 0; u < NTW; ++u) {
            const f32x4 d = D[u][0] + D[u][1];
#pragma unroll
            for (int r = 0; r < 4; ++r) A[u][r] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(d[r] * cc));
        }
        XS16_STAMP();                   // it.1: S1 + V1
        // S2: partial U^T = (X / 64)^T A^T over this wave's tokens   (slot_attention.py:59)
        f32x4 Up[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) Up[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NTW; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) Up[ct] = mfma16(Xa[u][ct][r], A[u][r], Up[ct]);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) *(f32x4*)(ux + (w * 16 + m) * XS_LD + 16 * ct + 4 * g) = Up[ct];
        // the W_hh h half of the gate pre-activations does not wait for U: it runs between the LDS write and the barrier
        f32x4 Gr, Gz, Gin, Ghn;
        if (!last) {
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
        XS16_STAMP();                   // it.2: S2 + W_hh half
        xs_lds_barrier();
        f32x4 PU[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float* p = ux + m * XS_LD + 16 * t + 4 * g;
            PU[t] = (*(const f32x4*)p + *(const f32x4*)(p + 16 * XS_LD)) +
                    (*(const f32x4*)(p + 32 * XS_LD) + *(const f32x4*)(p + 48 * XS_LD));
        }
        XS16_STAMP();                   // it.3: barrier + U sum
        if (!last) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Gr = mfma16(Wg[0][0][t][r], PU[t][r], Gr);
                    Gz = mfma16(Wg[1][0][t][r], PU[t][r], Gz);
                    Gin = mfma16(Wg[2][0][t][r], PU[t][r], Gin);
                }
            const f32x4 hn = xs16_gru(Gr, Gz, Gin, Ghn, xs16_pick(Ps, w));
            if (iok) *(f32x4*)(a.states + (it_row + m) * XS_D + 16 * w + 4 * g) = hn;
            float* hb = hx[it & 1];
            *(f32x4*)(hb + m * XS_LD + 16 * w + 4 * g) = hn;
            XS16_STAMP();               // it.4: W_ih half + gates
            xs_lds_barrier();
#pragma unroll
            for (int t = 0; t < 4; ++t) Ps[t] = *(const f32x4*)(hb + m * XS_LD + 16 * t + 4 * g);
            XS16_STAMP();               // it.5: state hand-off
        } else {
            // attention map, area, logits of the last iteration (slot_attention.py:68-96)
            float asum = 0.f;
#pragma unroll
            for (int u = 0; u < NTW; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * (w + 4 * u) + 4 * g + r;
                    if (iok && j < N) {
                        a.attn[((long)b * S + m) * N + j] = A[u][r];
                        asum += A[u][r];
                    }
                }
            asum = xs_wave_sum(asum);
            float us = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) us += (PU[t][0] + PU[t][1]) + (PU[t][2] + PU[t][3]);
            us = xs16_gsum(us);
            if (w == 0 && g == 0) usum[m] = iok ? us : 0.f;
            if (lane == 0) area_s[w] = asum;
            xs_lds_barrier();
            for (int c = tid; c < a.C; c += 256) {
                float v = 0.f;
                for (int k = 0; k < a.spc; ++k) v += usum[c * a.spc + k];
                a.logits[(long)b * a.C + c] = a.loss_status * v;
            }
            if (tid == 0) a.area_part[b] = (area_s[0] + area_s[1]) + (area_s[2] + area_s[3]);
            XS16_STAMP();               // end
        }
    }
}
