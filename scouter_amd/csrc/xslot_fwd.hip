// Fused xSlot forward: to_k MLP over the token grid, then T iterations of
//   D = s K^T d^-1/2 ; r_i = sum_j D_ij ; tau = sum_i r_i ; A = sigmoid(D / r_i * tau) ; U = A X / d ; s' = GRU(U, s)
// and the per-class logits / attention map / attention-area sum of the LAST iteration, in ONE kernel launch.
// Reference: sloter/utils/slot_attention.py:44-96 (as written: q = slots, no softmax, updates use X without the
// positional encoding, logits from the last iteration's updates, the 3rd GRU call is dead and is skipped).
//
// One workgroup (4 waves = one per SIMD, so each wave may use the whole 512-entry VGPR/AGPR file) per image.
// K, X and the GRU weights are staged in LDS once; each wave owns 32-slot tiles in the
// slot-per-lane register layout of xslot_common.h, so the QK^T -> sigmoid -> AV -> GRU chain runs from MFMA
// accumulator to MFMA operand without touching memory.  tau couples all slots of an image: one LDS reduction
// and one barrier per iteration (tiles summed in a fixed order -> deterministic).
#include "xslot_common.h"

struct XsFwdArgs {
    const float* X; const float* PE; const float* tok_w[8]; const float* tok_b[8]; const float* slots0;
    const float* w_ih; const float* w_hh; const float* b_ih; const float* b_hh;
    float* logits; float* attn; float* area_part; float* Ksave; float* Hsave; float* states;
    int B, N, S, C, spc, T, L;
    float loss_status;
};

template <int NJT, int TPW>
__global__ __launch_bounds__(256) void xslot_fwd_kernel(XsFwdArgs a) {
    constexpr int NP = 32 * NJT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Xs = lds;                        // [NP][68]  tokens X (rows >= N zero)
    float* Ks = Xs + NP * XS_LD;            // [NP][68]  K = to_k(X + PE) (rows >= N zero)
    float* Wih = Ks + NP * XS_LD;           // [192][68]
    float* Whh = Wih + 192 * XS_LD;         // [192][68]
    float* bias = Whh + 192 * XS_LD;        // br | bz | b_in | b_hn  (4 x 64)
    double* ksum = (double*)(bias + 256);   // [64]  column sums of K (fp64)
    double* tau_part = ksum + 64;           // [T<=8][16]
    float* area_s = (float*)(tau_part + 128);   // [16]
    float* usum = area_s + 16;              // [<=512] per-slot sum_k U_T[i][k]
    // MLP scratch aliases the (not yet loaded) GRU weight region
    float* H0 = Wih;                        // [NP][68]
    float* H1 = H0 + NP * XS_LD;            // [NP][68]
    float* Wt = H1 + NP * XS_LD;            // [64][68]

    const int tid = threadIdx.x, nthr = blockDim.x, NW = nthr >> 6;
    const int lane = tid & 63, wave = tid >> 6, hh = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x, N = a.N, S = a.S;
    const float* Xg = a.X + (long)b * N * XS_D;

    // ---- phase 0: tokens -> LDS ; H0 = X + PE
    for (int c = tid; c < NP * 16; c += nthr) {
        const int r = c >> 4, q = c & 15;
        f32x4 x = {0.f, 0.f, 0.f, 0.f}, p = x;
        if (r < N) { x = *(const f32x4*)(Xg + r * XS_D + q * 4); p = *(const f32x4*)(a.PE + r * XS_D + q * 4); }
        *(f32x4*)(Xs + r * XS_LD + q * 4) = x;
        *(f32x4*)(H0 + r * XS_LD + q * 4) = x + p;
        if (r < N) *(f32x4*)(a.Hsave + ((long)b * N + r) * XS_D + q * 4) = x + p;     // Hsave[0] = input of layer 0
    }
    // ---- phase 1: to_k MLP (Linear, then (ReLU, Linear)*), output tiles (jt, ot) spread over the waves
    float* Hin = H0;
    float* Hout = H1;
    for (int l = 0; l < a.L; ++l) {
        __syncthreads();
        xs_load_mat(Wt, a.tok_w[l], XS_D, tid, nthr);
        __syncthreads();
        const bool last = l == a.L - 1;
        for (int tile = wave; tile < NJT * 2; tile += NW) {
            const int jt = tile >> 1, ot = tile & 1;
            f32x16 acc;
            xs_zero(acc);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k0 = 32 * t + 8 * q + 4 * hh;
                    const f32x4 av = *(const f32x4*)(Hin + (32 * jt + l31) * XS_LD + k0);
                    const f32x4 bv = *(const f32x4*)(Wt + (32 * ot + l31) * XS_LD + k0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = mfma32(av[e], bv[e], acc);
                }
            const int o = 32 * ot + l31;
            const float bo = a.tok_b[l][o];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * jt + mfma32_row(r, lane);
                float v = acc[r] + bo;
                if (!last) v = fmaxf(v, 0.f);
                if (last) {
                    v = j < N ? v : 0.f;
                    Ks[j * XS_LD + o] = v;
                    if (j < N) a.Ksave[((long)b * N + j) * XS_D + o] = v;
                } else {
                    Hout[j * XS_LD + o] = v;
                    if (j < N) a.Hsave[(((long)(l + 1) * a.B + b) * N + j) * XS_D + o] = v;   // input of layer l+1
                }
            }
        }
        float* tmp = Hin; Hin = Hout; Hout = tmp;
    }
    __syncthreads();
    xs_colsum_f64(Ks, NP, ksum, tid);
    // ---- phase 2: GRU weights + combined biases
    xs_load_mat(Wih, a.w_ih, 192, tid, nthr);
    xs_load_mat(Whh, a.w_hh, 192, tid, nthr);
    for (int c = tid; c < 256; c += nthr) {
        const int g = c & 63, k = c >> 6;
        bias[c] = k == 0 ? a.b_ih[g] + a.b_hh[g] : k == 1 ? a.b_ih[64 + g] + a.b_hh[64 + g]
                : k == 2 ? a.b_ih[128 + g] : a.b_hh[128 + g];
    }
    __syncthreads();

    // ---- phase 3: iterations, slot tiles ti = wave + NW*tt
    const int ntiles = (S + 31) >> 5;
    const float scale = 0.125f;                 // d^-1/2, d = 64
    f32x16 h[TPW][2];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int i = (wave + NW * tt) * 32 + l31;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (i < S) v = *(const f32x4*)(a.slots0 + i * XS_D + 32 * t + 8 * q + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) h[tt][t][4 * q + e] = v[e];
            }
    }
    for (int it = 0; it < a.T; ++it) {
        const bool last = it == a.T - 1;
        float rr[TPW];
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const int ti = wave + NW * tt;
            rr[tt] = 0.f;
            if (ti < ntiles) {
                const double r64 = xs_rowdot_f64(h[tt], ksum, hh) * (double)scale;    // r_i (padded slots: exactly 0)
                rr[tt] = (float)r64;
                const double tr = xs_tilesum_f64(r64);
                if (lane == 0) tau_part[it * 16 + ti] = tr;
            }
        }
        __syncthreads();
        double tau64 = 0.0;
        for (int k = 0; k < ntiles; ++k) tau64 += tau_part[it * 16 + k];
        const float tau = (float)tau64;
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const int ti = wave + NW * tt;
            if (ti >= ntiles) continue;
            const int i = ti * 32 + l31;
            f32x16 A[NJT];
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                xs_zero(A[jt]);
                xs_mm_kc(Ks, 32 * jt, h[tt], A[jt], l31, hh);
                A[jt] *= scale;
            }
            float asum = 0.f;
            const float ir = xs_recip(rr[tt]);           // A = sigmoid(D / r_i * tau): one reciprocal per slot
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = xs_kidx(jt, r, hh);
                    float v = xs_sigmoid(xs_div(A[jt][r], rr[tt], ir) * tau);   // slot_attention.py:56-57
                    v = (i < S && j < N) ? v : 0.f;
                    A[jt][r] = v;
                    asum += v;
                    if (last && i < S && j < N) a.attn[((long)b * S + i) * N + j] = v;
                }
            f32x16 U[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                xs_zero(U[ct]);
                xs_mm_tr<NJT>(Xs, 32 * ct, A, U[ct], l31, hh);
                U[ct] *= (1.f / XS_D);                                    // slot_attention.py:59
            }
            if (last) {
                const float us = xs_rowsum<2>(U);
                if (hh == 0 && i < S) usum[i] = us;
                asum = wave_sum(asum);
                if (lane == 0) area_s[ti] = asum;
            } else {
                // ---- GRU cell, gate order r, z, n (slot_attention.py:60-66)
                f32x16 hn[2];
#pragma unroll
                for (int gt = 0; gt < 2; ++gt) {
                    f32x16 ar, az, ain, ahn;
                    xs_zero(ar); xs_zero(az); xs_zero(ain); xs_zero(ahn);
                    xs_mm_kc(Wih, 32 * gt, U, ar, l31, hh);
                    xs_mm_kc(Whh, 32 * gt, h[tt], ar, l31, hh);
                    xs_mm_kc(Wih, 64 + 32 * gt, U, az, l31, hh);
                    xs_mm_kc(Whh, 64 + 32 * gt, h[tt], az, l31, hh);
                    xs_mm_kc(Wih, 128 + 32 * gt, U, ain, l31, hh);
                    xs_mm_kc(Whh, 128 + 32 * gt, h[tt], ahn, l31, hh);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int g = xs_kidx(gt, r, hh);
                        const float rg = xs_sigmoid(ar[r] + bias[g]);
                        const float zg = xs_sigmoid(az[r] + bias[64 + g]);
                        const float ng = xs_tanh(ain[r] + bias[128 + g] + rg * (ahn[r] + bias[192 + g]));
                        hn[gt][r] = i < S ? (1.f - zg) * ng + zg * h[tt][gt][r] : 0.f;   // padded slots stay 0 (tau!)
                    }
                }
                h[tt][0] = hn[0];
                h[tt][1] = hn[1];
                if (i < S) {
                    float* sp = a.states + (((long)it * a.B + b) * S + i) * XS_D;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = h[tt][t][4 * q + e];
                            *(f32x4*)(sp + 32 * t + 8 * q + 4 * hh) = v;
                        }
                }
            }
        }
    }
    __syncthreads();
    // ---- class aggregation (slot_attention.py:87-91,96): logits_c = ls * sum_{s in c} sum_k U[s][k]
    for (int c = tid; c < a.C; c += nthr) {
        float v = 0.f;
        for (int k = 0; k < a.spc; ++k) v += usum[c * a.spc + k];
        a.logits[(long)b * a.C + c] = a.loss_status * v;
    }
    if (tid == 0) {
        float s = 0.f;
        for (int k = 0; k < ntiles; ++k) s += area_s[k];
        a.area_part[b] = s;
    }
}

static size_t xs_fwd_lds_bytes(int NJT) { return (size_t)(2 * 32 * NJT * XS_LD + 384 * XS_LD + 256 + 2 * (64 + 128) + 16 + 512) * sizeof(float); }

extern "C" int scouter_xslot_fwd_f32(const float* X, const float* PE, const float* const* tok_w,
                                     const float* const* tok_b, const float* slots0, const float* w_ih, const float* w_hh, const float* b_ih,
                                     const float* b_hh, int B, int N, int d, int S, int spc, int T, int L,
                                     float loss_status, float* logits, float* attn, float* area_part, float* Ksave,
                                     float* Hsave, float* states, void* stream) {
    SC_REQUIRE(X && PE && tok_w && tok_b && slots0 && w_ih && w_hh && b_ih && b_hh && logits && attn && area_part &&
               Ksave && (states || T <= 1) && Hsave, "xslot_fwd: null pointer");
    SC_REQUIRE(B > 0 && N > 0 && S > 0 && spc > 0 && S % spc == 0 && T >= 1 && L >= 1, "xslot_fwd: bad dims");
    SC_UNSUPPORTED(d == XS_D, "xslot_fwd: hidden_dim must be 64 (got %d)", d);
    SC_UNSUPPORTED(N <= XS_MAX_N, "xslot_fwd: at most %d tokens per image (got %d)", XS_MAX_N, N);
    SC_UNSUPPORTED(S <= 512 && T <= 8, "xslot_fwd: at most 512 slots and 8 iterations (got S=%d T=%d)", S, T);
    SC_UNSUPPORTED(L <= 8, "xslot_fwd: at most 8 to_k layers (got %d)", L);
    XsFwdArgs a{X, PE, {}, {}, slots0, w_ih, w_hh, b_ih, b_hh, logits, attn, area_part, Ksave, Hsave, states,
                B, N, S, S / spc, spc, T, L, loss_status};
    for (int l = 0; l < L; ++l) {
        SC_REQUIRE(tok_w[l] && tok_b[l], "xslot_fwd: null to_k layer %d", l);
        a.tok_w[l] = tok_w[l];
        a.tok_b[l] = tok_b[l];
    }
    const int ntiles = (S + 31) / 32, NJT = (N + 31) / 32;
    const int NW = 4, TPW = (ntiles + NW - 1) / NW;
    const size_t lds = xs_fwd_lds_bytes(NJT);
    hipStream_t st = (hipStream_t)stream;
    const double flops = (double)B * (2.0 * L * N * d * d + (double)T * 4.0 * S * N * d + (T - 1) * 12.0 * S * d * d);
    ScProfScope prof("xslot_fwd", st, flops, 4.0 * B * (2.0 * N * d + (double)S * N));
#define XS_LAUNCH(NJT_, TPW_)                                                                                       \
    do {                                                                                                            \
        auto kern = xslot_fwd_kernel<NJT_, TPW_>;                                                                   \
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
        hipLaunchKernelGGL(kern, dim3(B), dim3(NW * 64), lds, st, a);                                               \
    } while (0)
#define XS_TPW(NJT_)                                   \
    do {                                               \
        if (TPW == 1) XS_LAUNCH(NJT_, 1);              \
        else if (TPW == 2) XS_LAUNCH(NJT_, 2);         \
        else if (TPW == 3) XS_LAUNCH(NJT_, 3);         \
        else XS_LAUNCH(NJT_, 4);                       \
    } while (0)
    if (NJT == 1) XS_TPW(1);
    else if (NJT == 2) XS_TPW(2);
    else XS_TPW(3);
#undef XS_TPW
#undef XS_LAUNCH
    return sc_check_launch("xslot_fwd");
}
