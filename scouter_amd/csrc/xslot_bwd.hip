// Fused xSlot backward (hand-derived; the maths is oracle/xslot_manual.py, checked against autograd).
// One workgroup (4 waves, one per SIMD -> 512 registers per lane) per image.  For t = T-1 .. 0 the iteration's
// forward is recomputed from the saved slot state s_t, then in the slot-per-lane register layout:
//   GRU backward (t < T-1)      -> dgi, dgh (stored: operands of the weight-gradient GEMMs), dU, dh_prev
//   dA = dU X^T / d (+ area)    -> G = dA * A(1-A),  g_i = sum_j G_ij D_ij,  c0 = sum_i g_i / r_i      [barrier]
//   dD = G tau/r_i - g_i tau/r_i^2 + c0  ->  ds_t = d^-1/2 dD K + dh_prev
// The two contractions over the SLOT index (dK += dD^T s_t, dX += A^T dU / d) cannot be done with slots on the
// lanes; A_t, dD_t, dU_t go to an L2-resident scratch and a final phase contracts them tile by tile, then runs the
// to_k MLP backward.  Weight gradients of the GRU / to_k layers are plain GEMMs over all (image, slot) rows and
// are left to the generic wgrad kernel (deterministic split-K) by the host.
#include "xslot_small_common.h"

#include <stdlib.h>

struct XsBwdArgs {
    const float* X; const float* PE; const float* tok_w[8]; const float* slots0;
    const float* w_ih; const float* w_hh; const float* b_ih; const float* b_hh;
    const float* Ksave; const float* Hsave; const float* states; const float* dlogits; const float* g_area_sum;
    float* dX; float* dgi; float* dgh; float* Usave; float* ds0; float* dZ; float* ws;
    int B, N, S, C, spc, T, L;
    float loss_status;
    int halves;             // leftover slot tiles as 16-slot half tiles (xslot_bwd_kernel; SCOUTER_XSLOT_BWD_HALVES=0: off)
#ifdef XS_TIMING
    long long* stamps;      // dev build (tools_dev/xs_bwd_phase_timing.hip): [B][64] cycle stamps of wave 0
#endif
};

#ifdef XS_TIMING
#define XSB_STAMP() do { if (threadIdx.x == 0) a.stamps[blockIdx.x * 64 + nstamp] = __builtin_readcyclecounter(); ++nstamp; } while (0)
#else
#define XSB_STAMP() do { } while (0)
#endif

template <int NT>
__device__ __forceinline__ void xs_store_tile(float* base, int ld, const f32x16 (&m)[NT], int l31, int hh) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = m[t][4 * q + e];
            *(f32x4*)(base + (long)l31 * ld + 32 * t + 8 * q + 4 * hh) = v;
        }
}
template <int NT>
__device__ __forceinline__ void xs_load_tile(const float* base, int ld, f32x16 (&m)[NT], int l31, int hh, bool ok) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *(const f32x4*)(base + (long)l31 * ld + 32 * t + 8 * q + 4 * hh);
#pragma unroll
            for (int e = 0; e < 4; ++e) m[t][4 * q + e] = v[e];
        }
}
// acc += sum_{r<16} Mt[row_base + kidx(0,r,hh)][col0 + l31] * b[r]     (one 32-deep k-tile, rows of Mt = k)
__device__ __forceinline__ void xs_mm_tr_tile(const float* __restrict__ Mt, int row_base, int col0, const f32x16& b,
                                              f32x16& acc, int l31, int hh) {
    const float* base = Mt + (long)row_base * XS_LD + col0 + l31;
    float fr[5];
#pragma unroll
    for (int k = 0; k < 4; ++k) fr[k] = base[xs_kidx(0, k, hh) * XS_LD];
#pragma unroll
    for (int r = 0; r < 16; ++r) {                       // operand reads four MFMAs ahead (see xs_mm_tr_open)
        __builtin_amdgcn_sched_barrier(0);
        if (r + 4 < 16) fr[(r + 4) % 5] = base[xs_kidx(0, r + 4, hh) * XS_LD];
        acc = mfma32(fr[r % 5], b[r], acc);
    }
    XS_REGION_END();
}

// to_k MLP backward of one image: dZ_{L-1} = dK ; dH_{l-1} = dZ_l W_l ; dZ_{l-1} = dH_{l-1} * (H_{l-1} > 0);
// dX = dH_{-1} + dX^a.  dZa holds dK on entry; every dZ_l also goes to global (operand of the to_k weight gradients).
template <int NJT>
__device__ __forceinline__ void xs_mlp_bwd(const XsBwdArgs& a, float* dZa, float* dZb, const float* dXs, float* Wt,
                                           int tid, int wave, int lane, int l31, int hh, int b) {
    const int N = a.N;
    float* dZin = dZa;
    float* dZout = dZb;
    for (int l = a.L - 1; l >= 0; --l) {
        __syncthreads();
        xs_load_mat(Wt, a.tok_w[l], XS_D, tid, 256);
        for (int c = tid; c < N * 16; c += 256) {           // dZ_l -> global (operand of the to_k weight-gradient GEMM)
            const int r = c >> 4, q = c & 15;
            *(f32x4*)(a.dZ + (((long)l * a.B + b) * N + r) * XS_D + q * 4) = *(const f32x4*)(dZin + r * XS_LD + q * 4);
        }
        __syncthreads();
        for (int tile = wave; tile < NJT * 2; tile += 4) {
            const int jt = tile >> 1, ct = tile & 1;
            f32x16 acc;
            xs_zero(acc);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 av = *(const f32x4*)(dZin + (32 * jt + l31) * XS_LD + 32 * t + 8 * q + 4 * hh);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc = mfma32(av[e], Wt[(32 * t + 8 * q + 4 * hh + e) * XS_LD + 32 * ct + l31], acc);
                }
            const int c = 32 * ct + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int j = 32 * jt + mfma32_row(e, lane);
                float v = acc[e];
                if (l > 0) {
                    const float hv = j < N ? a.Hsave[(((long)l * a.B + b) * N + j) * XS_D + c] : 0.f;   // H_l = input of layer l
                    dZout[j * XS_LD + c] = hv > 0.f ? v : 0.f;
                } else if (j < N) {
                    a.dX[((long)b * N + j) * XS_D + c] = v + dXs[j * XS_LD + c];
                }
            }
        }
        float* tmp = dZin; dZin = dZout; dZout = tmp;
    }
}

// GRU backward of one 32-slot tile (slots on the lanes): gates recomputed from (U, h), pre-activation gradients
// stored as the operands of the weight-gradient GEMMs, dU += W_ih^T dgi, dh_prev += dz*ds + W_hh^T dgh.
// `row` = row of this lane's slot in the [T][B][S] gradient buffers.
// ---- the same two operand streams with the matrix in GLOBAL memory ([rows][64] row-major, L2-resident: the 48 KB of
// W_ih are read by every workgroup): used where the LDS cannot hold both GRU weight matrices (three token tiles).  Only the
// latency differs -- ~600-800 cycles instead of ~130 -- so the fragments are requested further ahead: four 16-byte
// fragments (16 MFMAs, 1 k cycles) resp. twelve 4-byte ones (768 cycles).
// (buffer loads: ONE lane-offset register per access pattern, everything else in the scalar / immediate offset -- with flat
//  pointers the compiler hoisted 240 loop-invariant 64-bit addresses out of the iteration loop and spilled them)
struct XsWihG { __amdgpu_buffer_rsrc_t rs; unsigned voff_kc, voff_tr; };
__device__ __forceinline__ XsWihG xs_wih_global(const float* w_ih, int l31, int hh) {
    XsWihG g;
    g.rs = __builtin_amdgcn_make_buffer_rsrc((void*)w_ih, 0, 192 * XS_D * 4, 0x00020000);
    g.voff_kc = (unsigned)(l31 * XS_D + 4 * hh) * 4u;            // row l31 of a 32-row block, k = 4 hh ..
    g.voff_tr = (unsigned)(4 * hh * XS_D + l31) * 4u;            // row 4 hh of an 8-row k group, column l31
    return g;
}
// load / multiply halves: the caller requests block i+1 BEFORE it multiplies block i (and the LDS-fed W_hh block that
// follows it), so an L2 round trip (~700 cycles) runs under 32-64 MFMAs instead of in front of every block
__device__ __forceinline__ void xs_kc_g_load(const XsWihG& W, int row0, f32x4 (&fr)[8]) {
#pragma unroll
    for (int f = 0; f < 8; ++f)
        fr[f] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            W.rs, W.voff_kc, (row0 * XS_D + 32 * (f >> 2) + 8 * (f & 3)) * 4, 0));
}
__device__ __forceinline__ void xs_kc_g_mma(const f32x4 (&fr)[8], const f32x16 (&b)[2], f32x16& acc) {
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma32(fr[f][e], b[f >> 2][4 * (f & 3) + e], acc);
    }
    XS_REGION_END();
}
__device__ __forceinline__ void xs_tr_g_load(const XsWihG& W, int row_base, int col0, float (&fr)[16]) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
        fr[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
            W.rs, W.voff_tr, ((row_base + 8 * (k >> 2) + (k & 3)) * XS_D + col0) * 4, 0));
}
__device__ __forceinline__ void xs_tr_g_mma(const float (&fr)[16], const f32x16& b, f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        __builtin_amdgcn_sched_barrier(0);
        acc = mfma32(fr[r], b[r], acc);
    }
    XS_REGION_END();
}

template <bool WG = false>      // WG: W_ih is read from global memory (xs_mm_kc_g / xs_mm_tr_tile_g: WihG), W_hh from LDS
__device__ __forceinline__ void xs_gru_bwd_tile(const XsBwdArgs& a, const float* Wih, const XsWihG& WihG, const float* Whh, const float* bias,
                                                const f32x16 (&h)[2], const f32x16 (&U)[2], const float* dsx_tile,
                                                float cs, const float* ksumf, f32x16 (&dU)[2], f32x16 (&dhp)[2],
                                                long row, bool iok, int l31, int hh) {
    if (iok) xs_store_tile<2>(a.Usave + (row - l31) * XS_D, XS_D, U, l31, hh);   // row - l31 = tile row 0
#pragma unroll
    for (int gt = 0; gt < 2; ++gt) {
        // ds of the 32 hidden units of this g-tile: the [32 slots][64] tile handed over by iteration t+1 (+ its
        // deferred c0 term cs * colsum(K), variant B); requested here, used after the six gate GEMMs
        f32x4 dsq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) dsq[q] = *(const f32x4*)(dsx_tile + (long)l31 * 64 + 32 * gt + 8 * q + 4 * hh);
        f32x16 ar, az, ain, ahn;
        xs_zero(ar); xs_zero(az); xs_zero(ain); xs_zero(ahn);
        if constexpr (WG) {
            // (measured, round 5: requesting block i+1 before block i multiplies -- two fragment buffers -- cost more in
            //  registers / spills than the exposed L2 round trips it removed: 552 -> 588 us at batch 256 x 300 slots)
            f32x4 f0[8];
            xs_kc_g_load(WihG, 32 * gt, f0);
            xs_kc_g_mma(f0, U, ar);
            xs_mm_kc(Whh, 32 * gt, h, ar, l31, hh);
            xs_kc_g_load(WihG, 64 + 32 * gt, f0);
            xs_kc_g_mma(f0, U, az);
            xs_mm_kc(Whh, 64 + 32 * gt, h, az, l31, hh);
            xs_kc_g_load(WihG, 128 + 32 * gt, f0);
            xs_kc_g_mma(f0, U, ain);
            xs_mm_kc(Whh, 128 + 32 * gt, h, ahn, l31, hh);
        } else {
            xs_mm_kc(Wih, 32 * gt, U, ar, l31, hh);
            xs_mm_kc(Whh, 32 * gt, h, ar, l31, hh);
            xs_mm_kc(Wih, 64 + 32 * gt, U, az, l31, hh);
            xs_mm_kc(Whh, 64 + 32 * gt, h, az, l31, hh);
            xs_mm_kc(Wih, 128 + 32 * gt, U, ain, l31, hh);
            xs_mm_kc(Whh, 128 + 32 * gt, h, ahn, l31, hh);
        }
        // gate values -> their pre-activation gradients, in place
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int g = xs_kidx(gt, e, hh);
            const float rg = xs_sigmoid(ar[e] + bias[g]);
            const float zg = xs_sigmoid(az[e] + bias[64 + g]);
            const float hnb = ahn[e] + bias[192 + g];
            const float ng = xs_tanh(ain[e] + bias[128 + g] + rg * hnb);
            const float ds = dsq[e >> 2][e & 3] + cs * ksumf[g];
            const float da_n = ds * (1.f - zg) * (1.f - ng * ng);
            const float da_z = ds * (h[gt][e] - ng) * zg * (1.f - zg);
            const float da_r = da_n * hnb * rg * (1.f - rg);
            dhp[gt][e] += ds * zg;
            ar[e] = da_r; az[e] = da_z; ain[e] = da_n; ahn[e] = da_n * rg;
        }
        if (iok) {
            float* gi = a.dgi + row * 192 + 32 * gt + 4 * hh;
            float* gh = a.dgh + row * 192 + 32 * gt + 4 * hh;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v0, v1, v2, v3;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] = ar[4 * q + e]; v1[e] = az[4 * q + e]; v2[e] = ain[4 * q + e]; v3[e] = ahn[4 * q + e]; }
                *(f32x4*)(gi + 8 * q) = v0; *(f32x4*)(gi + 64 + 8 * q) = v1; *(f32x4*)(gi + 128 + 8 * q) = v2;
                *(f32x4*)(gh + 8 * q) = v0; *(f32x4*)(gh + 64 + 8 * q) = v1; *(f32x4*)(gh + 128 + 8 * q) = v3;
            }
        }
        // dU^T += W_ih^T dgi^T ; dh_prev^T += W_hh^T dgh^T   (k = the 32 hidden units of this g-tile)
        if constexpr (WG) {
            float g0[16];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                xs_tr_g_load(WihG, 32 * gt, 32 * ct, g0);
                xs_tr_g_mma(g0, ar, dU[ct]);
                xs_tr_g_load(WihG, 64 + 32 * gt, 32 * ct, g0);
                xs_tr_g_mma(g0, az, dU[ct]);
                xs_tr_g_load(WihG, 128 + 32 * gt, 32 * ct, g0);
                xs_tr_g_mma(g0, ain, dU[ct]);
                xs_mm_tr_tile(Whh, 32 * gt, 32 * ct, ar, dhp[ct], l31, hh);
                xs_mm_tr_tile(Whh, 64 + 32 * gt, 32 * ct, az, dhp[ct], l31, hh);
                xs_mm_tr_tile(Whh, 128 + 32 * gt, 32 * ct, ahn, dhp[ct], l31, hh);
            }
        } else {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                xs_mm_tr_tile(Wih, 32 * gt, 32 * ct, ar, dU[ct], l31, hh);
                xs_mm_tr_tile(Wih, 64 + 32 * gt, 32 * ct, az, dU[ct], l31, hh);
                xs_mm_tr_tile(Wih, 128 + 32 * gt, 32 * ct, ain, dU[ct], l31, hh);
                xs_mm_tr_tile(Whh, 32 * gt, 32 * ct, ar, dhp[ct], l31, hh);
                xs_mm_tr_tile(Whh, 64 + 32 * gt, 32 * ct, az, dhp[ct], l31, hh);
                xs_mm_tr_tile(Whh, 128 + 32 * gt, 32 * ct, ahn, dhp[ct], l31, hh);
            }
        }
    }
}

// ---- variant A (N > 64 tokens: no LDS left for the transposes of variant B): A_t, dD_t, dU_t go to an L2/HBM
// scratch and a final phase contracts them over the slot index.
template <int NJT>
__global__ __launch_bounds__(256) void xslot_bwd_scratch_kernel(XsBwdArgs a) {
    constexpr int NP = 32 * NJT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Xs = lds;
    float* Ks = Xs + NP * XS_LD;
    float* Wih = Ks + NP * XS_LD;
    float* Whh = Wih + 192 * XS_LD;
    float* bias = Whh + 192 * XS_LD;        // br | bz | b_in | b_hn
    double* ksum = (double*)(bias + 256);   // [64] column sums of K (fp64)
    double* red64 = ksum + 64;              // tau_part[16]
    float* red = (float*)(red64 + 16);      // (unused)[16] | c0_part[16]
    float* r_s = red + 32;                  // [512] r_i
    float* g_s = r_s + 512;                 // [512] g_i
    // final-phase buffers alias the GRU weights
    float* dZa = Wih;                       // [NP][68]
    float* dZb = dZa + NP * XS_LD;          // [NP][68]
    float* dXs = dZb + NP * XS_LD;          // [NP][68]
    float* Wt = dXs + NP * XS_LD;           // [64][68]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x, N = a.N, S = a.S, T = a.T;
    const int ntiles = (S + 31) >> 5, Sp = ntiles * 32, TPW = (ntiles + 3) >> 2;
    const float scale = 0.125f, inv_d = 1.f / XS_D;
    int nstamp = 0;
    (void)nstamp;
    XSB_STAMP();
    // scratch of this image: dsn [Sp][64] | per t: A [Sp][NP], dD [Sp][NP], dU [Sp][64]
    const long per_img = (long)Sp * (64 + (long)T * (2 * NP + 64));
    float* dsn = a.ws + (long)b * per_img;
    auto As_t = [&](int t) { return dsn + (long)Sp * 64 + (long)t * Sp * (2 * NP + 64); };
    auto dDs_t = [&](int t) { return As_t(t) + (long)Sp * NP; };
    auto dUs_t = [&](int t) { return dDs_t(t) + (long)Sp * NP; };

    // ---- stage X, K, GRU weights
    for (int c = tid; c < NP * 16; c += 256) {
        const int r = c >> 4, q = c & 15;
        f32x4 x = {0.f, 0.f, 0.f, 0.f}, k = x;
        if (r < N) {
            x = *(const f32x4*)(a.X + ((long)b * N + r) * XS_D + q * 4);
            k = *(const f32x4*)(a.Ksave + ((long)b * N + r) * XS_D + q * 4);
        }
        *(f32x4*)(Xs + r * XS_LD + q * 4) = x;
        *(f32x4*)(Ks + r * XS_LD + q * 4) = k;
    }
    xs_load_mat_c<192, 256>(Wih, a.w_ih, tid);
    xs_load_mat_c<192, 256>(Whh, a.w_hh, tid);
    {
        const int g = tid & 63, k = tid >> 6;
        bias[tid] = k == 0 ? a.b_ih[g] + a.b_hh[g] : k == 1 ? a.b_ih[64 + g] + a.b_hh[64 + g]
                  : k == 2 ? a.b_ih[128 + g] : a.b_hh[128 + g];
    }
    const float g_area = a.g_area_sum ? a.g_area_sum[0] : 0.f;
    __syncthreads();
    xs_colsum_f64(Ks, NP, ksum, tid);
    __syncthreads();
    XSB_STAMP();

    for (int it = T - 1; it >= 0; --it) {
        const bool last = it == T - 1;
        const float* sbase = it == 0 ? a.slots0 : a.states + ((long)(it - 1) * a.B + b) * S * XS_D;
        // ================= phase A: r_i, tau
        for (int tt = 0; tt < TPW; ++tt) {
            const int ti = wave + 4 * tt;
            if (ti >= ntiles) continue;
            const int i = ti * 32 + l31;
            f32x16 h[2];
            xs_load_tile<2>(sbase + (long)ti * 32 * XS_D, XS_D, h, l31, hh, i < S);
            const double r64 = xs_rowdot_f64(h, ksum, hh) * (double)scale;       // same fp64 normaliser as the forward
            if (hh == 0) r_s[i] = (float)r64;
            const double tr = xs_tilesum_f64(r64);
            if (lane == 0) red64[ti] = tr;
        }
        xs_lds_barrier();        // LDS hand-off only: pending global stores keep flying
        double tau64 = 0.0;
        for (int k = 0; k < ntiles; ++k) tau64 += red64[k];
        const float tau = (float)tau64;
        XSB_STAMP();
        // ================= phase B1
        for (int tt = 0; tt < TPW; ++tt) {
            const int ti = wave + 4 * tt;
            if (ti >= ntiles) continue;
            const int i = ti * 32 + l31;
            const bool iok = i < S;
            f32x16 h[2], U[2], dU[2], dhp[2];
            xs_load_tile<2>(sbase + (long)ti * 32 * XS_D, XS_D, h, l31, hh, iok);
            const float r = r_s[i];
            const float ir = xs_recip(r);
            {   // D, A = sigmoid(D / r * tau); both are parked in the scratch while the GRU section runs
                f32x16 D[NJT], A[NJT];
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt) {
                    xs_zero(D[jt]);
                    xs_mm_kc(Ks, 32 * jt, h, D[jt], l31, hh);
                    D[jt] *= scale;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float v = xs_sigmoid(xs_div(D[jt][e], r, ir) * tau);
                        A[jt][e] = (iok && xs_kidx(jt, e, hh) < N) ? v : 0.f;
                    }
                }
                xs_store_tile<NJT>(As_t(it) + (long)ti * 32 * NP, NP, A, l31, hh);
                xs_store_tile<NJT>(dDs_t(it) + (long)ti * 32 * NP, NP, D, l31, hh);
                if (tt == 0) XSB_STAMP();
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    xs_zero(U[ct]);
                    xs_mm_tr<NJT>(Xs, 32 * ct, A, U[ct], l31, hh);
                    U[ct] *= inv_d;
                    xs_zero(dhp[ct]);
                }
            }
            if (tt == 0) XSB_STAMP();
            if (last) {
                const float du = iok ? a.loss_status * a.dlogits[(long)b * a.C + i / a.spc] : 0.f;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int e = 0; e < 16; ++e) dU[ct][e] = du;
            } else {
                xs_zero(dU[0]); xs_zero(dU[1]);
                const long row = ((long)it * a.B + b) * S + i;
                xs_gru_bwd_tile(a, Wih, XsWihG{}, Whh, bias, h, U, dsn + (long)ti * 32 * 64, 0.f, bias, dU, dhp, row, iok, l31, hh);
            }
            if (tt == 0) XSB_STAMP();
            if (!iok) {   // padded slots carry nothing
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) { xs_zero(dU[ct]); xs_zero(dhp[ct]); }
            }
            xs_store_tile<2>(dsn + (long)ti * 32 * 64, 64, dhp, l31, hh);         // stash dh_prev (same lanes re-read it)
            xs_store_tile<2>(dUs_t(it) + (long)ti * 32 * 64, 64, dU, l31, hh);
            // dA^T = X dU^T / d (+ area term), G = dA * A (1 - A), g_i = sum_j G_ij D_ij
            float gsum = 0.f;
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                f32x16 dA, Aj[1], Dj[1];
                xs_load_tile<1>(As_t(it) + (long)ti * 32 * NP + 32 * jt, NP, Aj, l31, hh, true);
                xs_load_tile<1>(dDs_t(it) + (long)ti * 32 * NP + 32 * jt, NP, Dj, l31, hh, true);
                xs_zero(dA);
                xs_mm_kc(Xs, 32 * jt, dU, dA, l31, hh);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float av = Aj[0][e];
                    float v = dA[e] * inv_d + (last ? g_area : 0.f);
                    v = (iok && xs_kidx(jt, e, hh) < N) ? v * av * (1.f - av) : 0.f;
                    gsum += v * Dj[0][e];
                    Aj[0][e] = v;                                   // G
                }
                xs_store_tile<1>(dDs_t(it) + (long)ti * 32 * NP + 32 * jt, NP, Aj, l31, hh);   // stash G over D
            }
            if (tt == 0) XSB_STAMP();
            gsum += __shfl_xor(gsum, 32, 64);
            if (hh == 0) g_s[i] = gsum;
            const float tc = xs_tilesum(iok ? gsum / r : 0.f);
            if (lane == 0) red[16 + ti] = tc;
        }
        xs_lds_barrier();        // LDS hand-off only: pending global stores keep flying
        float c0 = 0.f;
        for (int k = 0; k < ntiles; ++k) c0 += red[16 + k];
        XSB_STAMP();
        // ================= phase B2: dD, ds_t
        for (int tt = 0; tt < TPW; ++tt) {
            const int ti = wave + 4 * tt;
            if (ti >= ntiles) continue;
            const int i = ti * 32 + l31;
            const bool iok = i < S;
            f32x16 G[NJT], ds[2], dhp[2];
            xs_load_tile<NJT>(dDs_t(it) + (long)ti * 32 * NP, NP, G, l31, hh, true);
            xs_load_tile<2>(dsn + (long)ti * 32 * 64, 64, dhp, l31, hh, true);
            const float r = r_s[i], gi = g_s[i];
            const float k1 = tau / r, k2 = gi * tau / (r * r);
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    G[jt][e] = (iok && xs_kidx(jt, e, hh) < N) ? G[jt][e] * k1 - k2 + c0 : 0.f;
            xs_store_tile<NJT>(dDs_t(it) + (long)ti * 32 * NP, NP, G, l31, hh);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                xs_zero(ds[ct]);
                xs_mm_tr<NJT>(Ks, 32 * ct, G, ds[ct], l31, hh);
#pragma unroll
                for (int e = 0; e < 16; ++e) ds[ct][e] = iok ? ds[ct][e] * scale + dhp[ct][e] : 0.f;
            }
            if (it > 0) xs_store_tile<2>(dsn + (long)ti * 32 * 64, 64, ds, l31, hh);
            else if (iok) xs_store_tile<2>(a.ds0 + ((long)b * S + ti * 32) * XS_D, XS_D, ds, l31, hh);
        }
        xs_lds_barrier();        // LDS hand-off only: pending global stores keep flying
        XSB_STAMP();
    }

    // ================= final phase: contractions over the slot index
    //   dK[j][c]  = d^-1/2 sum_t sum_i dD_t[i][j] s_t[i][c]      dXa[j][c] = 1/d sum_t sum_i A_t[i][j] dU_t[i][c]
    __threadfence_block();
    __syncthreads();
    for (int job = wave; job < NJT * 4; job += 4) {
        const int prod = job & 1, ct = (job >> 1) & 1, jt = job >> 2;
        f32x16 acc;
        xs_zero(acc);
        for (int t = 0; t < T; ++t) {
            const float* P = (prod == 0 ? dDs_t(t) : As_t(t)) + 32 * jt + l31;          // [i][NP]
            const float* Q;
            int ldq;
            if (prod == 1) { Q = dUs_t(t); ldq = 64; }
            else { Q = t == 0 ? a.slots0 : a.states + ((long)(t - 1) * a.B + b) * S * XS_D; ldq = XS_D; }
            Q += 32 * ct + l31;
            // operands come from the L2-resident scratch: the next 16 slots are requested before the MFMAs of the
            // current ones (one wave per SIMD -- nothing else hides a load round trip)
            float pv[2][8], qv[2][8];
            auto fetch = [&](int buf, int i0) {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int i = i0 + 2 * s + hh;
                    pv[buf][s] = P[(long)i * NP];
                    qv[buf][s] = (prod == 1 || i < S) ? Q[(long)i * ldq] : 0.f;
                }
            };
            fetch(0, 0);
            for (int i0 = 0; i0 < Sp; i0 += 32) {
                if (i0 + 16 < Sp) fetch(1, i0 + 16);
#pragma unroll
                for (int s = 0; s < 8; ++s) acc = mfma32(pv[0][s], qv[0][s], acc);
                if (i0 + 16 < Sp) {
                    if (i0 + 32 < Sp) fetch(0, i0 + 32);
#pragma unroll
                    for (int s = 0; s < 8; ++s) acc = mfma32(pv[1][s], qv[1][s], acc);
                }
            }
        }
        float* dst = prod == 0 ? dZa : dXs;
        const float sc = prod == 0 ? scale : inv_d;
#pragma unroll
        for (int e = 0; e < 16; ++e) dst[(32 * jt + mfma32_row(e, lane)) * XS_LD + 32 * ct + l31] = acc[e] * sc;
    }
    XSB_STAMP();
    xs_mlp_bwd<NJT>(a, dZa, dZb, dXs, Wt, tid, wave, lane, l31, hh, b);
    XSB_STAMP();
}

// ---- variant B (N <= 64): no scratch, one barrier per iteration.
//  * the slot-index contractions run inside the loop: each wave bounces its tiles through a private 32x36 LDS
//    buffer to get them with the slot index on the MFMA k axis, and keeps its partial dK / dX^a sums in registers
//    (reduced over the four waves in a fixed order at the end);
//  * the two workgroup-wide scalars do not need their own barriers: tau_t depends on the saved slot states only, so
//    it is computed one iteration ahead; c0_t enters dD_t, ds_t and dK linearly (dD = dD' + c0), so the wave works
//    with dD' and the c0 terms are added where the values are next read:
//        ds_t = ds'_t + d^-1/2 c0_t colsum(K),      dK += d^-1/2 c0_t (1 (x) colsum(s_t)).
#define XS_LDB 36           // row stride of the transpose buffer (floats): 16B rows, conflict-free b128 reads

// t^T: the register tile [row(e,hh)][i = l31] comes back as 16 values per lane, lane (l31, hh) holding
// [row = l31][i = 16 hh + r], r = 0..15 -- the operand order of the contraction MFMAs (k-step r <-> i = 16 hh + r)
__device__ __forceinline__ void xs_transpose_tile(float* __restrict__ buf, const f32x16& v, f32x16& out, int l31, int hh) {
#pragma unroll
    for (int e = 0; e < 16; ++e) buf[xs_kidx(0, e, hh) * XS_LDB + l31] = v[e];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 x = *(const f32x4*)(buf + l31 * XS_LDB + 16 * hh + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) out[4 * q + e] = x[e];
    }
}

// WG (three token tiles, the reference's 9 x 9 grid): W_ih stays in global memory / L2 and is read as MFMA operands through
// registers -- the LDS cannot hold X, K, both GRU weight matrices and the four bounce buffers at NP = 96 (179.6 KB); without
// the 52 KB of W_ih the in-loop variant fits and the first-generation scratch kernel is retired for 64 < N <= 96.
template <int NJT, bool WG = false>
__global__ __launch_bounds__(256) void xslot_bwd_kernel(XsBwdArgs a) {
    constexpr int NP = 32 * NJT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bias = lds;                          // [256] br | bz | b_in | b_hn
    double* ksum = (double*)(bias + 256);       // [64] column sums of K (fp64)
    double* red64 = ksum + 64;                  // [2][16] tau partials, double-buffered over the iterations
    float* redc = (float*)(red64 + 32);         // [2][16] c0 partials
    float* ksumf = redc + 32;                   // [64] colsum(K) in fp32
    float* spart = ksumf + 64;                  // [2][4][64] colsum(s_t) of each wave's tiles, double-buffered
    float* corr = spart + 512;                  // [64] sum_t c0_t colsum(s_t): the c0 terms of dK
    float* Xs = corr + 64;
    float* Ks = Xs + NP * XS_LD;
    float* WihL = Ks + NP * XS_LD;              // (WG: no LDS copy of W_ih)
    float* Whh = WihL + (WG ? 0 : 192 * XS_LD);
    const float* Wih = WG ? a.w_ih : WihL;
    const XsWihG wihg = xs_wih_global(a.w_ih, threadIdx.x & 31, (threadIdx.x & 63) >> 5);
    float* bounce = Whh + 192 * XS_LD;          // [4][32][XS_LDB] per-wave transpose buffers
    // after the loop the pool is re-used: wave-reduction buffer = dK | dX^a, then the MLP buffers
    float* dZa = Xs;                            // [NP][68]
    float* dXs = dZa + NP * XS_LD;              // [NP][68]
    float* dZb = dXs + NP * XS_LD;              // [NP][68]
    float* Wt = dZb + NP * XS_LD;               // [64][68]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x, N = a.N, S = a.S, T = a.T;
    const int ntiles = (S + 31) >> 5, Sp = ntiles * 32;
    // HALF TILES (round 6).  One or two leftover tiles (ntiles % 4 = 1, 2: 300 slots = 8 + 2 tiles) used to cost the waves
    // that own them a whole tile's time while the other waves idled at the iteration's barrier (10 tiles: 3 + 3 + 2 + 2,
    // the launch paced by 3).  They are walked as 16-slot half tiles instead, one per wave, on v_mfma_f32_16x16x4_f32
    // (slot-per-lane-16 layout of xslot_small_common.h, every operand from the same LDS images): half the time of a tile,
    // no exchange between waves -- 2.5 tiles per wave.  Their slot contractions feed the SAME 32 x 32 accumulators (eight
    // k-steps instead of sixteen), so parks and the final reduction do not change.
    const int rem4 = ntiles & 3;
    const bool halves = !WG && NJT <= 2 && a.halves && (rem4 == 1 || rem4 == 2);
    const int nfull = halves ? (ntiles & ~3) : ntiles;              // tiles walked 32 slots wide
    const int TPW = (nfull + 3) >> 2;
    const int nhalf = halves ? (S - 32 * nfull + 15) >> 4 : 0;      // live half tiles (<= 4): wave w walks half w
    const int nent = nfull + nhalf;                                 // entries of the tau / c0 partial tables
    const bool my_half = wave < nhalf;
    const int hslot0 = 32 * nfull + 16 * wave, m16 = lane & 15, g4 = lane >> 4;
    const float scale = 0.125f, inv_d = 1.f / XS_D;
    int nstamp = 0;
    (void)nstamp;
    XSB_STAMP();
    float* dsn = a.ws + (long)b * Sp * 64;      // ds'_t of this image: [Sp][64], handed from iteration to iteration
    float* tbuf = bounce + wave * (32 * XS_LDB);
    auto state_t = [&](int t) { return t == 0 ? a.slots0 : a.states + ((long)(t - 1) * a.B + b) * S * XS_D; };

    // ---- stage X, K, GRU weights; column sums of K and of the slot states
    for (int c = tid; c < NP * 16; c += 256) {
        const int r = c >> 4, q = c & 15;
        f32x4 x = {0.f, 0.f, 0.f, 0.f}, k = x;
        if (r < N) {
            x = *(const f32x4*)(a.X + ((long)b * N + r) * XS_D + q * 4);
            k = *(const f32x4*)(a.Ksave + ((long)b * N + r) * XS_D + q * 4);
        }
        *(f32x4*)(Xs + r * XS_LD + q * 4) = x;
        *(f32x4*)(Ks + r * XS_LD + q * 4) = k;
    }
    if constexpr (!WG) xs_load_mat_c<192, 256>(WihL, a.w_ih, tid);
    xs_load_mat_c<192, 256>(Whh, a.w_hh, tid);
    {
        const int g = tid & 63, k = tid >> 6;
        bias[tid] = k == 0 ? a.b_ih[g] + a.b_hh[g] : k == 1 ? a.b_ih[64 + g] + a.b_hh[64 + g]
                  : k == 2 ? a.b_ih[128 + g] : a.b_hh[128 + g];
    }
    const float g_area = a.g_area_sum ? a.g_area_sum[0] : 0.f;
    __syncthreads();
    XSB_STAMP();
    xs_colsum_f64(Ks, NP, ksum, tid);
    if (tid < 64) ksumf[tid] = (float)ksum[tid];
    __syncthreads();                            // ksum is read by every wave below
    XSB_STAMP();

    // r_i of the wave's tiles for iteration `it` (registers; every tile belongs to one wave throughout) and the
    // tile partials of tau into red64[buf]
    // ... and colsum(s_t) over the wave's tiles into spart[buf][wave] (the c0 term of dK needs it)
    float rcur[4] = {0.f, 0.f, 0.f, 0.f};
    float rcur_h = 0.f;                         // r_i of the half tile's slot hslot0 + m16
    auto phase_a = [&](int it, int buf) {
        const float* sbase = state_t(it);
        f32x16 hs[2], h[4][2];
        xs_zero(hs[0]); xs_zero(hs[1]);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)          // per-lane predicates, no branch: the tiles are in flight together
            xs_load_tile<2>(sbase + (long)(wave + 4 * tt) * 32 * XS_D, XS_D, h[tt], l31, hh, (wave + 4 * tt) * 32 + l31 < S);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int ti = wave + 4 * tt;
            if (ti < nfull) {
                const double r64 = xs_rowdot_f64(h[tt], ksum, hh) * (double)scale;   // same fp64 normaliser as the forward
                rcur[tt] = (float)r64;
                const double tr = xs_tilesum_f64(r64);
                if (lane == 0) red64[buf * 16 + ti] = tr;
                hs[0] += h[tt][0]; hs[1] += h[tt][1];
            }
        }
        // sum over the slots (lanes): bounce the two tiles through the wave's transpose buffer -- lane (c, hh) gets
        // the 16 slots of its half in registers -- then add the halves
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 hT;
            xs_transpose_tile(tbuf, hs[t], hT, l31, hh);
            float v = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) v += hT[e];
            v = xs_halfsum(v);
            if (hh == 0) spart[(buf * 4 + wave) * 64 + 32 * t + l31] = v;
        }
        if (my_half) {                              // this wave's half tile: r_i, tau partial, colsum(s_t)
            const int i = hslot0 + m16;
            const float* sp = sbase + (long)min(i, S - 1) * XS_D + 4 * g4;
            f32x4 ps[4];
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                ps[t] = *(const f32x4*)(sp + 16 * t);
                if (i >= S) ps[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                const double2 k01 = *(const double2*)(ksum + 16 * t + 4 * g4);
                const double2 k23 = *(const double2*)(ksum + 16 * t + 4 * g4 + 2);
                a0 += (double)ps[t][0] * k01.x;
                a1 += (double)ps[t][1] * k01.y;
                a0 += (double)ps[t][2] * k23.x;
                a1 += (double)ps[t][3] * k23.y;
            }
            const double r64 = xs16_gsum_f64(a0 + a1) * (double)scale;
            rcur_h = (float)r64;
            const double tr = xs16_rowsum_f64(r64);
            if (lane == 0) red64[buf * 16 + nfull + wave] = tr;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = xs16_rowsum(ps[t][r]);
                    if (m16 == 0) spart[(buf * 4 + wave) * 64 + 16 * t + 4 * g4 + r] += v;
                }
        }
    };
    phase_a(T - 1, 0);
    __syncthreads();
    XSB_STAMP();

    // this wave's share of dK' and dX^a (rows = tokens, cols = channels): 8 NJT x 16 registers that live in a
    // wave-private, lane-contiguous global buffer between two contractions (re-read while the dA / ds' GEMMs of the
    // next tile run) -- kept in registers throughout they push the GRU section over the 512-register budget
    f32x4* accpark = (f32x4*)(a.ws + (long)a.B * Sp * 64) + ((long)b * 4 + wave) * (4 * NJT * 4 * 64) + lane;
    bool have_acc = false;

    int pb = 0;
    float ssum_c = 0.f, corr_c = 0.f;
    for (int it = T - 1; it >= 0; --it) {
        const bool last = it == T - 1;
        const float* sbase = state_t(it);
        double tau64 = 0.0;
        for (int k = 0; k < nent; ++k) tau64 += red64[pb * 16 + k];
        const float tau = (float)tau64;
        float cs_prev = 0.f;                    // d^-1/2 c0_{it+1}: completes the ds' handed over by that iteration
        if (!last) {
            float c0 = 0.f;
            for (int k = 0; k < nent; ++k) c0 += redc[pb * 16 + k];
            corr_c += c0 * ssum_c;              // (threads 0..63: channel tid) c0_{it+1} colsum(s_{it+1})
            cs_prev = c0 * scale;
        }
        if (tid < 64)
            ssum_c = (spart[(pb * 4 + 0) * 64 + tid] + spart[(pb * 4 + 1) * 64 + tid]) +
                     (spart[(pb * 4 + 2) * 64 + tid] + spart[(pb * 4 + 3) * 64 + tid]);
        XSB_STAMP();
        for (int tt = 0; tt < TPW; ++tt) {
            const int ti = wave + 4 * tt;
            if (ti >= nfull) continue;
            const int i = ti * 32 + l31;
            const bool iok = i < S;
            f32x16 h[2], A[NJT], D[NJT], U[2], dU[2], dhp[2];
            xs_load_tile<2>(sbase + (long)ti * 32 * XS_D, XS_D, h, l31, hh, iok);
            const float r = tt == 0 ? rcur[0] : tt == 1 ? rcur[1] : tt == 2 ? rcur[2] : rcur[3];
            const float ir = xs_recip(r);
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {          // D, A = sigmoid(D / r * tau)
                xs_zero(D[jt]);
                xs_mm_kc(Ks, 32 * jt, h, D[jt], l31, hh);
                D[jt] *= scale;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = xs_sigmoid(xs_div(D[jt][e], r, ir) * tau);
                    A[jt][e] = (iok && xs_kidx(jt, e, hh) < N) ? v : 0.f;
                }
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                xs_zero(U[ct]);
                xs_mm_tr<NJT>(Xs, 32 * ct, A, U[ct], l31, hh);
                U[ct] *= inv_d;
                xs_zero(dhp[ct]);
            }
            if (tt == 0) XSB_STAMP();
            if (last) {
                const float du = iok ? a.loss_status * a.dlogits[(long)b * a.C + i / a.spc] : 0.f;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int e = 0; e < 16; ++e) dU[ct][e] = du;
            } else {
                xs_zero(dU[0]); xs_zero(dU[1]);
                const long row = ((long)it * a.B + b) * S + i;
                xs_gru_bwd_tile<WG>(a, Wih, wihg, Whh, bias, h, U, dsn + (long)ti * 32 * 64, iok ? cs_prev : 0.f, ksumf, dU, dhp, row,
                                    iok, l31, hh);
            }
            if (tt == 0) XSB_STAMP();
            if (!iok) {   // padded slots carry nothing
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) { xs_zero(dU[ct]); xs_zero(dhp[ct]); }
            }
            if constexpr (WG) {
                // three token tiles: A (48 registers) is not kept across the GRU section -- D is, and A = sigmoid(D / r tau)
                // is re-evaluated here with the same expression (same bits): ~400 VALU per tile instead of a spill
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float v = xs_sigmoid(xs_div(D[jt][e], r, ir) * tau);
                        A[jt][e] = (iok && xs_kidx(jt, e, hh) < N) ? v : 0.f;
                    }
            }
            if constexpr (!WG) {
            f32x16 accK[NJT][2], accX[NJT][2];
            auto acc_load = [&](f32x16 (&acc)[NJT][2], int base) {
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        xs_zero(acc[jt][ct]);
                        if (have_acc) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const f32x4 v = accpark[((base + jt * 2 + ct) * 4 + q) * 64];
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[jt][ct][4 * q + e] = v[e];
                            }
                        }
                    }
            };
            auto acc_store = [&](const f32x16 (&acc)[NJT][2], int base) {
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[jt][ct][4 * q + e];
                            accpark[((base + jt * 2 + ct) * 4 + q) * 64] = v;
                        }
            };
            acc_load(accK, 0);
            // dA^T = X dU^T / d (+ area term), G = dA * A (1 - A), g_i = sum_j G_ij D_ij ; G replaces D
            float gsum = 0.f;
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                f32x16 dA;
                xs_zero(dA);
                xs_mm_kc(Xs, 32 * jt, dU, dA, l31, hh);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float av = A[jt][e];
                    float v = dA[e] * inv_d + (last ? g_area : 0.f);
                    v = (iok && xs_kidx(jt, e, hh) < N) ? v * av * (1.f - av) : 0.f;
                    gsum += v * D[jt][e];
                    D[jt][e] = v;
                }
            }
            gsum = xs_halfsum(gsum);
            const float tc = xs_tilesum(iok ? gsum / r : 0.f);
            if (lane == 0) redc[(pb ^ 1) * 16 + ti] = tc;
            // dD' = G tau / r - g tau / r^2  (the + c0 is added by the readers), ds' = d^-1/2 dD' K + dh_prev
            const float k1 = tau / r, k2 = gsum * tau / (r * r);
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    D[jt][e] = (iok && xs_kidx(jt, e, hh) < N) ? D[jt][e] * k1 - k2 : 0.f;
            {
                f32x16 ds[2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    xs_zero(ds[ct]);
                    xs_mm_tr<NJT>(Ks, 32 * ct, D, ds[ct], l31, hh);
#pragma unroll
                    for (int e = 0; e < 16; ++e) ds[ct][e] = iok ? ds[ct][e] * scale + dhp[ct][e] : 0.f;
                }
                xs_store_tile<2>(dsn + (long)ti * 32 * 64, 64, ds, l31, hh);
            }
            if (tt == 0) XSB_STAMP();
            acc_load(accX, 2 * NJT);            // in flight while dK' is contracted
            // contractions over the slots of this tile:  dK'[j][c] += dD'[i][j] s[i][c],  dX^a[j][c] += A[i][j] dU[i][c]
            {
                f32x16 qT[2], pT;
                xs_transpose_tile(tbuf, h[0], qT[0], l31, hh);
                xs_transpose_tile(tbuf, h[1], qT[1], l31, hh);
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt) {
                    xs_transpose_tile(tbuf, D[jt], pT, l31, hh);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int r16 = 0; r16 < 16; ++r16) accK[jt][ct] = mfma32(pT[r16], qT[ct][r16], accK[jt][ct]);
                    XS_REGION_END();
                }
                acc_store(accK, 0);
                xs_transpose_tile(tbuf, dU[0], qT[0], l31, hh);
                xs_transpose_tile(tbuf, dU[1], qT[1], l31, hh);
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt) {
                    xs_transpose_tile(tbuf, A[jt], pT, l31, hh);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int r16 = 0; r16 < 16; ++r16) accX[jt][ct] = mfma32(pT[r16], qT[ct][r16], accX[jt][ct]);
                    XS_REGION_END();
                }
            }
            acc_store(accX, 2 * NJT);
            } else {
            // (WG, three token tiles: the 2 x 96 accumulator registers of dK' / dX^a do not fit next to A, D, h, dU -- the
            //  parks are walked one token tile at a time, the next tile's two blocks in flight while this one contracts)
            auto park_load = [&](f32x16 (&acc)[2], int blk) {          // blocks blk, blk + 1 (ct = 0, 1)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    xs_zero(acc[ct]);
                    if (have_acc) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v = accpark[((blk + ct) * 4 + q) * 64];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[ct][4 * q + e] = v[e];
                        }
                    }
                }
            };
            auto park_store = [&](const f32x16 (&acc)[2], int blk) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[ct][4 * q + e];
                        accpark[((blk + ct) * 4 + q) * 64] = v;
                    }
            };
            // dA^T = X dU^T / d (+ area term), G = dA * A (1 - A), g_i = sum_j G_ij D_ij ; G replaces D
            float gsum = 0.f;
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                f32x16 dA;
                xs_zero(dA);
                xs_mm_kc(Xs, 32 * jt, dU, dA, l31, hh);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float av = A[jt][e];
                    float v = dA[e] * inv_d + (last ? g_area : 0.f);
                    v = (iok && xs_kidx(jt, e, hh) < N) ? v * av * (1.f - av) : 0.f;
                    gsum += v * D[jt][e];
                    D[jt][e] = v;
                }
            }
            gsum = xs_halfsum(gsum);
            const float tc = xs_tilesum(iok ? gsum / r : 0.f);
            if (lane == 0) redc[(pb ^ 1) * 16 + ti] = tc;
            // dD' = G tau / r - g tau / r^2  (the + c0 is added by the readers), ds' = d^-1/2 dD' K + dh_prev
            const float k1 = tau / r, k2 = gsum * tau / (r * r);
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    D[jt][e] = (iok && xs_kidx(jt, e, hh) < N) ? D[jt][e] * k1 - k2 : 0.f;
            {
                f32x16 ds[2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    xs_zero(ds[ct]);
                    xs_mm_tr<NJT>(Ks, 32 * ct, D, ds[ct], l31, hh);
#pragma unroll
                    for (int e = 0; e < 16; ++e) ds[ct][e] = iok ? ds[ct][e] * scale + dhp[ct][e] : 0.f;
                }
                xs_store_tile<2>(dsn + (long)ti * 32 * 64, 64, ds, l31, hh);
            }
            if (tt == 0) XSB_STAMP();
            // contractions over the slots of this tile:  dK'[j][c] += dD'[i][j] s[i][c],  dX^a[j][c] += A[i][j] dU[i][c]
            {
                f32x16 qT[2], pT, accA[2], accB[2];
                xs_transpose_tile(tbuf, h[0], qT[0], l31, hh);
                xs_transpose_tile(tbuf, h[1], qT[1], l31, hh);
                park_load(accA, 0);
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt) {
                    f32x16 (&cur)[2] = (jt & 1) ? accB : accA;
                    f32x16 (&nxt)[2] = (jt & 1) ? accA : accB;
                    park_load(nxt, jt + 1 < NJT ? 2 * (jt + 1) : 2 * NJT);       // (last: the first dX^a blocks)
                    xs_transpose_tile(tbuf, D[jt], pT, l31, hh);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int r16 = 0; r16 < 16; ++r16) cur[ct] = mfma32(pT[r16], qT[ct][r16], cur[ct]);
                    XS_REGION_END();
                    park_store(cur, 2 * jt);
                }
                xs_transpose_tile(tbuf, dU[0], qT[0], l31, hh);
                xs_transpose_tile(tbuf, dU[1], qT[1], l31, hh);
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt) {
                    // (NJT = 3: the dX^a blocks of jt = 0 were requested into the buffer that follows dK' jt = 2)
                    f32x16 (&cur)[2] = ((NJT + jt) & 1) ? accB : accA;
                    f32x16 (&nxt)[2] = ((NJT + jt) & 1) ? accA : accB;
                    if (jt + 1 < NJT) park_load(nxt, 2 * NJT + 2 * (jt + 1));
                    xs_transpose_tile(tbuf, A[jt], pT, l31, hh);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int r16 = 0; r16 < 16; ++r16) cur[ct] = mfma32(pT[r16], qT[ct][r16], cur[ct]);
                    XS_REGION_END();
                    park_store(cur, 2 * NJT + 2 * jt);
                }
            }
            }
            have_acc = true;
            if (tt == 0) XSB_STAMP();
        }

        if constexpr (!WG && NJT <= 2) {
        if (my_half) {
            XSB_STAMP();                    // half tile: start
            // ================= this wave's 16-slot half tile (slot-per-lane-16; see the note at `halves`)
            constexpr int NTK = 2 * NJT;                    // 16-token tiles
            const int i = hslot0 + m16;
            const bool iok = i < S;
            f32x4 Ps[4], PU[4], PdU[4], dhp[4], D[NTK], A[NTK];
            {
                const float* sp = sbase + (long)min(i, S - 1) * XS_D + 4 * g4;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    Ps[t] = *(const f32x4*)(sp + 16 * t);
                    if (!iok) Ps[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            const float r = rcur_h, ir = xs_recip(r);
            // (operand reads are left to the compiler's scheduler here: requesting every block one block ahead through explicit
            //  double buffers, pinned or not, measured 2.5 % slower on the whole launch -- tools_dev/xs_bwd_phase_timing.hip)
#pragma unroll
            for (int tk = 0; tk < NTK; ++tk) {              // D^T = K s^T / 8, A = sigmoid(D / r * tau)
                f32x4 kr[4], d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
                for (int t = 0; t < 4; ++t) kr[t] = *(const f32x4*)(Ks + (16 * tk + m16) * XS_LD + 16 * t + 4 * g4);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        d0 = mfma16(kr[t][e], Ps[t][e], d0);
                        d1 = mfma16(kr[t + 2][e], Ps[t + 2][e], d1);
                    }
                D[tk] = (d0 + d1) * scale;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = xs_sigmoid(xs_div(D[tk][e], r, ir) * tau);
                    A[tk][e] = (iok && 16 * tk + 4 * g4 + e < N) ? v : 0.f;
                }
            }
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) { PU[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; PdU[ct] = PU[ct]; dhp[ct] = PU[ct]; }
#pragma unroll
            for (int tk = 0; tk < NTK; ++tk) {              // U^T = X^T A^T / d
                float xa[4][4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xa[ct][e] = Xs[(16 * tk + 4 * g4 + e) * XS_LD + 16 * ct + m16];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) PU[ct] = mfma16(xa[ct][e], A[tk][e], PU[ct]);
            }
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) PU[ct] *= inv_d;
            XSB_STAMP();                    // half tile: D, A, U
            if (last) {
                const float du = iok ? a.loss_status * a.dlogits[(long)b * a.C + i / a.spc] : 0.f;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) PdU[ct] = f32x4{du, du, du, du};
            } else {
                // GRU backward, 16 hidden units at a time (xs_gru_bwd_tile, 16 wide): gates recomputed from (U, s_t)
                const long row = ((long)it * a.B + b) * S + i;
                const float cs = iok ? cs_prev : 0.f;
                if (iok) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) *(f32x4*)(a.Usave + row * XS_D + 16 * t + 4 * g4) = PU[t];
                }
#pragma unroll
                for (int ht = 0; ht < 4; ++ht) {
                    const f32x4 dsq = *(const f32x4*)(dsn + (long)i * 64 + 16 * ht + 4 * g4);
                    f32x4 Gr = {0.f, 0.f, 0.f, 0.f}, Gz = Gr, Gin = Gr, Ghn = Gr;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float* wi = Wih + (16 * ht + m16) * XS_LD + 16 * t + 4 * g4;
                        const float* wh = Whh + (16 * ht + m16) * XS_LD + 16 * t + 4 * g4;
                        const f32x4 ir_ = *(const f32x4*)wi, iz_ = *(const f32x4*)(wi + 64 * XS_LD), in_ = *(const f32x4*)(wi + 128 * XS_LD);
                        const f32x4 hr_ = *(const f32x4*)wh, hz_ = *(const f32x4*)(wh + 64 * XS_LD), hn_ = *(const f32x4*)(wh + 128 * XS_LD);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            Gr = mfma16(ir_[e], PU[t][e], Gr);
                            Gz = mfma16(iz_[e], PU[t][e], Gz);
                            Gin = mfma16(in_[e], PU[t][e], Gin);
                            Gr = mfma16(hr_[e], Ps[t][e], Gr);
                            Gz = mfma16(hz_[e], Ps[t][e], Gz);
                            Ghn = mfma16(hn_[e], Ps[t][e], Ghn);
                        }
                    }
                    const f32x4 b0 = *(const f32x4*)(bias + 16 * ht + 4 * g4), b1 = *(const f32x4*)(bias + 64 + 16 * ht + 4 * g4);
                    const f32x4 b2 = *(const f32x4*)(bias + 128 + 16 * ht + 4 * g4), b3 = *(const f32x4*)(bias + 192 + 16 * ht + 4 * g4);
                    const f32x4 kf = *(const f32x4*)(ksumf + 16 * ht + 4 * g4);
                    f32x4 da_r, da_z, da_n, da_nr;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float rg = xs_sigmoid(Gr[e] + b0[e]);
                        const float zg = xs_sigmoid(Gz[e] + b1[e]);
                        const float hnb = Ghn[e] + b3[e];
                        const float ng = xs_tanh(Gin[e] + b2[e] + rg * hnb);
                        const float ds = dsq[e] + cs * kf[e];
                        da_n[e] = ds * (1.f - zg) * (1.f - ng * ng);
                        da_z[e] = ds * (Ps[ht][e] - ng) * zg * (1.f - zg);
                        da_r[e] = da_n[e] * hnb * rg * (1.f - rg);
                        da_nr[e] = da_n[e] * rg;
                        dhp[ht][e] += ds * zg;
                    }
                    if (iok) {
                        float* gi = a.dgi + row * 192 + 16 * ht + 4 * g4;
                        float* gh = a.dgh + row * 192 + 16 * ht + 4 * g4;
                        *(f32x4*)gi = da_r; *(f32x4*)(gi + 64) = da_z; *(f32x4*)(gi + 128) = da_n;
                        *(f32x4*)gh = da_r; *(f32x4*)(gh + 64) = da_z; *(f32x4*)(gh + 128) = da_nr;
                    }
                    // dU^T += W_ih^T dgi^T ; dh_prev^T += W_hh^T dgh^T   (k = these 16 hidden units of the three gates)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float* wi = Wih + (16 * ht + 4 * g4 + e) * XS_LD + m16;
                        const float* wh = Whh + (16 * ht + 4 * g4 + e) * XS_LD + m16;
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) {
                            PdU[ct] = mfma16(wi[16 * ct], da_r[e], PdU[ct]);
                            dhp[ct] = mfma16(wh[16 * ct], da_r[e], dhp[ct]);
                            PdU[ct] = mfma16(wi[64 * XS_LD + 16 * ct], da_z[e], PdU[ct]);
                            dhp[ct] = mfma16(wh[64 * XS_LD + 16 * ct], da_z[e], dhp[ct]);
                            PdU[ct] = mfma16(wi[128 * XS_LD + 16 * ct], da_n[e], PdU[ct]);
                            dhp[ct] = mfma16(wh[128 * XS_LD + 16 * ct], da_nr[e], dhp[ct]);
                        }
                    }
                }
            }
            if (!iok) {     // padded slots carry nothing
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) { PdU[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; dhp[ct] = PdU[ct]; }
            }
            XSB_STAMP();                    // half tile: GRU
            // dA^T = X dU^T / d (+ area term), G = dA * A (1 - A), g_i = sum_j G_ij D_ij ; G replaces D
            float gsum = 0.f;
#pragma unroll
            for (int tk = 0; tk < NTK; ++tk) {
                f32x4 xq[4], d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
                for (int t = 0; t < 4; ++t) xq[t] = *(const f32x4*)(Xs + (16 * tk + m16) * XS_LD + 16 * t + 4 * g4);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        d0 = mfma16(xq[t][e], PdU[t][e], d0);
                        d1 = mfma16(xq[t + 2][e], PdU[t + 2][e], d1);
                    }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float av = A[tk][e];
                    float v = (d0[e] + d1[e]) * inv_d + (last ? g_area : 0.f);
                    v = (iok && 16 * tk + 4 * g4 + e < N) ? v * av * (1.f - av) : 0.f;
                    gsum += v * D[tk][e];
                    D[tk][e] = v;
                }
            }
            gsum = xs16_gsum(gsum);
            const float tc = xs16_rowsum(iok ? gsum / r : 0.f);
            if (lane == 0) redc[(pb ^ 1) * 16 + nfull + wave] = tc;
            // dD' = G tau / r - g tau / r^2  (the + c0 is added by the readers), ds' = d^-1/2 dD' K + dh_prev
            const float k1 = tau / r, k2 = gsum * tau / (r * r);
#pragma unroll
            for (int tk = 0; tk < NTK; ++tk)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    D[tk][e] = (iok && 16 * tk + 4 * g4 + e < N) ? D[tk][e] * k1 - k2 : 0.f;
            {
                f32x4 ds[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) ds[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tk = 0; tk < NTK; ++tk) {
                    float kt[4][4];
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                        for (int e = 0; e < 4; ++e) kt[ct][e] = Ks[(16 * tk + 4 * g4 + e) * XS_LD + 16 * ct + m16];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) ds[ct] = mfma16(kt[ct][e], D[tk][e], ds[ct]);
                }
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ds[ct][e] = iok ? ds[ct][e] * scale + dhp[ct][e] : 0.f;
                    *(f32x4*)(dsn + (long)i * 64 + 16 * ct + 4 * g4) = ds[ct];
                }
            }
            XSB_STAMP();                    // half tile: dA, G, dD, ds
            // contractions over these 16 slots into the 32 x 32 accumulators: dK'[j][c] += dD'[i][j] s[i][c],
            // dX^a[j][c] += A[i][j] dU[i][c].  A 32-row x 16-slot block (lane = slot column m16, rows 4 g4 + e and
            // 16 + 4 g4 + e) bounces through the wave's private buffer and comes back as the eight k-steps of the 32 x 32 MFMAs:
            // lane (l31, hh) holds [row l31][slot 8 hh + e].
            auto bounce16 = [&](const f32x4& lo, const f32x4& hi, f32x16& out) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    tbuf[(4 * g4 + e) * XS_LDB + m16] = lo[e];
                    tbuf[(16 + 4 * g4 + e) * XS_LDB + m16] = hi[e];
                }
                const f32x4 x0 = *(const f32x4*)(tbuf + l31 * XS_LDB + 8 * hh), x1 = *(const f32x4*)(tbuf + l31 * XS_LDB + 8 * hh + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { out[e] = x0[e]; out[4 + e] = x1[e]; }
            };
            auto acc_io = [&](f32x16 (&acc)[NJT][2], int base, bool store) {
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 v = {0.f, 0.f, 0.f, 0.f};
                            if (store) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = acc[jt][ct][4 * q + e];
                                accpark[((base + jt * 2 + ct) * 4 + q) * 64] = v;
                            } else {
                                if (have_acc) v = accpark[((base + jt * 2 + ct) * 4 + q) * 64];
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[jt][ct][4 * q + e] = v[e];
                            }
                        }
            };
            {
                f32x16 acc[NJT][2], qT[2], pT;
                xs_zero(qT[0]); xs_zero(qT[1]); xs_zero(pT);
                acc_io(acc, 0, false);
                bounce16(Ps[0], Ps[1], qT[0]);
                bounce16(Ps[2], Ps[3], qT[1]);
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt) {
                    bounce16(D[2 * jt], D[2 * jt + 1], pT);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[jt][ct] = mfma32(pT[e], qT[ct][e], acc[jt][ct]);
                    XS_REGION_END();
                }
                acc_io(acc, 0, true);
                acc_io(acc, 2 * NJT, false);
                bounce16(PdU[0], PdU[1], qT[0]);
                bounce16(PdU[2], PdU[3], qT[1]);
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt) {
                    bounce16(A[2 * jt], A[2 * jt + 1], pT);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[jt][ct] = mfma32(pT[e], qT[ct][e], acc[jt][ct]);
                    XS_REGION_END();
                }
                acc_io(acc, 2 * NJT, true);
            }
            XSB_STAMP();                    // half tile: slot contractions
            have_acc = true;
        }
        }
        if (it > 0) phase_a(it - 1, pb ^ 1);
        xs_lds_barrier();        // LDS hand-off only: pending global stores keep flying
        pb ^= 1;
        XSB_STAMP();
    }
    // c0 of iteration 0 completes ds_0 (the gradient of the initial slots); same lanes re-read their ds' tiles
    float c00 = 0.f;
    for (int k = 0; k < nent; ++k) c00 += redc[pb * 16 + k];
    if (tid < 64) corr[tid] = (corr_c + c00 * ssum_c) * scale;
    {
        f32x16 ds[4][2];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)          // (tiles beyond the wave's last one re-read tile 0: no branch, the
            xs_load_tile<2>(dsn + (long)(wave + 4 * tt < nfull ? wave + 4 * tt : 0) * 32 * 64, 64, ds[tt], l31, hh,
                            true);              //  loads of all tiles are in flight together)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int ti = wave + 4 * tt;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 kq = *(const f32x4*)(ksumf + 32 * ct + 8 * q + 4 * hh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ds[tt][ct][4 * q + e] += c00 * scale * kq[e];
                }
            if (ti < nfull && ti * 32 + l31 < S)
                xs_store_tile<2>(a.ds0 + ((long)b * S + ti * 32) * XS_D, XS_D, ds[tt], l31, hh);
        }
        if (my_half) {
            const int i = hslot0 + m16;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                f32x4 v = *(const f32x4*)(dsn + (long)i * 64 + 16 * ct + 4 * g4);
                v += (c00 * scale) * *(const f32x4*)(ksumf + 16 * ct + 4 * g4);
                if (i < S) *(f32x4*)(a.ds0 + ((long)b * S + i) * XS_D + 16 * ct + 4 * g4) = v;
            }
        }
    }
    XSB_STAMP();
    // ---- dK' and dX^a: sum the four waves' parks (fixed order) into the LDS pool
    __threadfence_block();
    __syncthreads();
    {
        const f32x4* parks = (const f32x4*)(a.ws + (long)a.B * Sp * 64) + (long)b * 4 * (4 * NJT * 4 * 64);
#pragma unroll
        for (int u = 0; u < NJT * 4; ++u) {                 // item = (tile, q): 4 NJT tiles x 4 quads = 256 * NJT * 4 / 64
            const int item = u * 4 + wave;                  // every wave takes whole (tile, q) rows: lanes stay lanes
            const int tile = item >> 2, q = item & 3;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {                   // (a wave without tiles never wrote its park: wave 0's is
                const bool has = w < nfull || w < nhalf;    //  read instead and dropped -- no branch between the loads)
                const f32x4 x = parks[(long)(has ? w : 0) * (4 * NJT * 4 * 64) + (tile * 4 + q) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += has ? x[e] : 0.f;
            }
            const int prod = tile / (2 * NJT), jt = (tile >> 1) % NJT, ct = tile & 1;
            float* dst = prod == 0 ? dZa : dXs;
            const float sc = prod == 0 ? scale : inv_d;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                dst[(32 * jt + mfma32_row(4 * q + e, lane)) * XS_LD + 32 * ct + l31] = v[e] * sc;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < N * 64; idx += 256) {         // the c0 terms of dK
        dZa[(idx >> 6) * XS_LD + (idx & 63)] += corr[idx & 63];
    }
    XSB_STAMP();
    xs_mlp_bwd<NJT>(a, dZa, dZb, dXs, Wt, tid, wave, lane, l31, hh, b);
    XSB_STAMP();
}

#include "xslot_small_bwd.h"

static bool xs_small_bwd_enabled() {       // SCOUTER_XSLOT_SMALL=0: heads with <= 16 slots stay on the 32-slot-tile kernels (A/B, tests)
    const char* e = getenv("SCOUTER_XSLOT_SMALL");
    return !(e && e[0] == '0');
}
static bool xs_bwd_scratch_variant() {       // SCOUTER_XSLOT_BWD_SCRATCH=1: the first-generation kernel for 64 < N <= 96 (A/B)
    static int v = -1;
    if (v < 0) { const char* e = getenv("SCOUTER_XSLOT_BWD_SCRATCH"); v = e && e[0] == '1'; }
    return v != 0;
}
static size_t xs_bwd_lds_bytes(int NJT) {
    if (NJT > 2 && xs_bwd_scratch_variant())
        return (size_t)(2 * 32 * NJT * XS_LD + 384 * XS_LD + 256 + 2 * (64 + 16) + 32 + 1024) * sizeof(float);
    // (three token tiles: W_ih stays in global memory -- one 192-row matrix less)
    return (size_t)(256 + 2 * 64 + 2 * 32 + 32 + 64 + 512 + 64 + 2 * 32 * NJT * XS_LD + (NJT > 2 ? 192 : 384) * XS_LD +
                    4 * 32 * XS_LDB) * sizeof(float);
}

extern "C" size_t scouter_xslot_bwd_workspace_bytes(int B, int N, int d, int S, int T) {
    (void)d;
    const long Sp = (S + 31) / 32 * 32, NP = (N + 31) / 32 * 32;
    const size_t inloop = (size_t)B * (Sp * 64 + 4 * (NP / 32) * 4 * 16 * 64) * sizeof(float);     // ds' hand-off + accumulator parks
    if (NP <= 64) return inloop;
    const size_t scratch = (size_t)B * Sp * (64 + (long)T * (2 * NP + 64)) * sizeof(float);    // + A_t, dD_t, dU_t of every iteration
    return scratch > inloop ? scratch : inloop;       // (either variant may run for 64 < N <= 96)
}

extern "C" int scouter_xslot_bwd_f32(const float* X, const float* PE, const float* const* tok_w, const float* slots0,
                                     const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                                     const float* Ksave, const float* Hsave, const float* states,
                                     const float* dlogits, const float* g_area_sum, int B, int N, int d, int S,
                                     int spc, int T, int L, float loss_status, float* dX, float* dgi, float* dgh,
                                     float* Usave, float* ds0, float* dZ, void* ws, size_t ws_bytes, void* stream) {
    SC_REQUIRE(X && tok_w && slots0 && w_ih && w_hh && b_ih && b_hh && Ksave && dlogits && dX && ds0 && dZ && ws,
               "xslot_bwd: null pointer");
    SC_REQUIRE((T <= 1) || (states && dgi && dgh && Usave), "xslot_bwd: missing GRU buffers");
    SC_REQUIRE(Hsave, "xslot_bwd: missing to_k activations");
    SC_REQUIRE(B > 0 && N > 0 && S > 0 && spc > 0 && S % spc == 0 && T >= 1 && L >= 1, "xslot_bwd: bad dims");
    SC_UNSUPPORTED(d == XS_D && N <= XS_MAX_N && S <= 512 && T <= 8, "xslot_bwd: unsupported dims d=%d N=%d S=%d T=%d",
                   d, N, S, T);
    if (ws_bytes < scouter_xslot_bwd_workspace_bytes(B, N, d, S, T)) {
        sc_set_error("xslot_bwd: workspace too small");
        return SC_ERR_WORKSPACE;
    }
    SC_UNSUPPORTED(L <= 8, "xslot_bwd: at most 8 to_k layers (got %d)", L);
    XsBwdArgs a{X, PE, {}, slots0, w_ih, w_hh, b_ih, b_hh, Ksave, Hsave, states, dlogits, g_area_sum,
                dX, dgi, dgh, Usave, ds0, dZ, (float*)ws, B, N, S, S / spc, spc, T, L, loss_status, 1};
    { const char* e = getenv("SCOUTER_XSLOT_BWD_HALVES"); if (e && e[0] == '0') a.halves = 0; }
    for (int l = 0; l < L; ++l) {
        SC_REQUIRE(tok_w[l], "xslot_bwd: null to_k layer %d", l);
        a.tok_w[l] = tok_w[l];
    }
    const int NJT = (N + 31) / 32;
    const size_t lds = xs_bwd_lds_bytes(NJT);
    hipStream_t st = (hipStream_t)stream;
    const double flops = 2.0 * (double)B * (2.0 * L * N * d * d + (double)T * 4.0 * S * N * d + (T - 1) * 12.0 * S * d * d);
    ScProfScope prof("xslot_bwd", st, flops, 4.0 * B * (3.0 * N * d + (double)S * d));
    if (S <= 16 && xs_small_bwd_enabled()) {        // the metric's own head: xslot_small_bwd.h
        const int NTW = N <= 64 ? 1 : 2;
        const int slds = (int)xs_small_bwd_lds_bytes(NTW);
        if (ws_bytes < xs_small_bwd_park_bytes(B, NTW)) {
            sc_set_error("xslot_bwd: workspace too small");
            return SC_ERR_WORKSPACE;
        }
        if (NTW == 1) {
            hipFuncSetAttribute((const void*)xslot_small_bwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, slds);
            hipLaunchKernelGGL(xslot_small_bwd_kernel<1>, dim3(B), dim3(256), slds, st, a);
        } else {
            hipFuncSetAttribute((const void*)xslot_small_bwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, slds);
            hipLaunchKernelGGL(xslot_small_bwd_kernel<2>, dim3(B), dim3(256), slds, st, a);
        }
        return sc_check_launch("xslot_small_bwd");
    }
#define XSB_LAUNCH(KERN)                                                                                   \
    do {                                                                                                   \
        auto kern = KERN;                                                                                  \
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
        hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, st, a);                                          \
    } while (0)
    if (NJT == 1) XSB_LAUNCH(xslot_bwd_kernel<1>);
    else if (NJT == 2) XSB_LAUNCH(xslot_bwd_kernel<2>);
    else if (xs_bwd_scratch_variant()) XSB_LAUNCH(xslot_bwd_scratch_kernel<3>);
    else XSB_LAUNCH((xslot_bwd_kernel<3, true>));
#undef XSB_LAUNCH
    return sc_check_launch("xslot_bwd");
}
