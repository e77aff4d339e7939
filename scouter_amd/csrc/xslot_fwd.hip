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

#include <stdlib.h>

struct XsFwdArgs {
    const float* X; const float* PE; const float* tok_w[8]; const float* tok_b[8]; const float* slots0;
    const float* w_ih; const float* w_hh; const float* b_ih; const float* b_hh;
    float* logits; float* attn; float* area_part; float* Ksave; float* Hsave; float* states;
    int B, N, S, C, spc, T, L;
    float loss_status;
#ifdef XS_TIMING
    long long* stamps;      // dev build (tools_dev/xs_phase_timing.hip): [B][64] cycle stamps of wave 0
#endif
};

#ifdef XS_TIMING
#define XS_STAMP() do { if (threadIdx.x == 0) a.stamps[blockIdx.x * 64 + nstamp] = __builtin_readcyclecounter(); ++nstamp; } while (0)
#else
#define XS_STAMP() do { } while (0)
#endif

#define XS_FWD_WAVES 4          // one wave per SIMD: each may use the whole 512-entry register file

// ---------------------------------------------------------------------------------------------------------------
// Building blocks of one (slot tile, iteration).  "S" blocks are streams of MFMAs (+ their LDS operand reads), "V"
// blocks are VALU work cut into PIECES of <= ~12 instructions (one 64-cycle MFMA shadow each).  xs_stream() issues
// one S block MFMA by MFMA, calls the piece functor of an INDEPENDENT V block after each MFMA and pins that order
// with scheduling barriers, so that -- with a single wave per SIMD -- the gate / sigmoid arithmetic of one slot tile
// executes under the matrix instructions of another instead of between them (left to itself the compiler emits the
// VALU block as one run of ~350 instructions during which the MFMA pipe idles).
//   S1: D^T = (K/8) s^T          V1: A = sigmoid(D / r_i * tau)     S2: U^T = (X/64)^T A^T
//   S3(gt): gate pre-activations of hidden units 32gt..32gt+31       V2(gt): GRU gates -> new hidden units
// K is staged pre-scaled by d^-1/2 = 1/8 and X (transposed) by 1/d = 1/64: exact power-of-two scalings, same bits.
// ---------------------------------------------------------------------------------------------------------------
#define XS_XT_FLOATS(NP_) ((NP_) * XS_LD > 64 * ((NP_) + 4) ? (NP_) * XS_LD : 64 * ((NP_) + 4))
#define XS_SB() __builtin_amdgcn_sched_barrier(0)
struct XsLds { const float* Xs; const float* Ks; const float* Wih; const float* Whh; const float* bias; };

// An S block = NFRAG operand fragments (one ds_read_b128 = the A operands of 4 consecutive MFMAs).  Fragments are
// read two ahead of use; the first two are requested by `*_pre` -- which callers issue BEFORE the VALU block that
// precedes the stream, so the LDS latency (and, for the gate streams, the bias reads that initialise the
// accumulators) is covered by that arithmetic instead of being exposed at every block boundary.
struct XsFrag { f32x4 f0, f1; };
template <typename AddrF>
__device__ __forceinline__ XsFrag xs_pre(const AddrF& addr) {
    return XsFrag{*(const f32x4*)addr(0), *(const f32x4*)addr(1)};
}
template <int NFRAG, typename AddrF, typename MmaF>
__device__ __forceinline__ void xs_stream(const AddrF& addr, const MmaF& mma, const XsFrag& pre) {
    f32x4 fr[3];
    fr[0] = pre.f0;
    fr[1] = pre.f1;
#pragma unroll
    for (int f = 0; f < NFRAG; ++f) {
        XS_SB();
        if (f + 2 < NFRAG) fr[(f + 2) % 3] = *(const f32x4*)addr(f + 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) mma(f, e, fr[f % 3][e]);
    }
    XS_SB();
}

// S1: D^T = (K/8) s^T
template <int NJT>
struct XsS1 {
    const float* base;
    __device__ __forceinline__ XsS1(const XsLds& m, int l31, int hh) : base(m.Ks + l31 * XS_LD + 4 * hh) {}
    __device__ __forceinline__ const float* operator()(int f) const {
        return base + (32 * (f >> 3)) * XS_LD + 32 * ((f >> 2) & 1) + 8 * (f & 3);
    }
    __device__ __forceinline__ void run(const f32x16 (&h)[2], f32x16 (&D)[NJT], const XsFrag& pre) const {
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) xs_zero(D[jt]);
        xs_stream<8 * NJT>(*this, [&](int f, int e, float a) {
            D[f >> 3] = mfma32(a, h[(f >> 2) & 1][4 * (f & 3) + e], D[f >> 3]); }, pre);
    }
};
// S2: U^T = (X/64)^T A^T.  X is staged TRANSPOSED (Xt[c][j], row stride 32*NJT+4): the contraction index j is
// contiguous, so one ds_read_b128 feeds four MFMAs like in the other blocks
// NG <= 4 NJT: live 8-token groups (ceil(N / 8) rounded up by the caller's instantiation): the contraction skips the
// groups of padded tokens (N = 81: 11 of 12 -- the reference's default 9x9 grid pads to 96 otherwise)
template <int NJT, int NG>
struct XsS2 {
    static constexpr int LDT = 32 * NJT + 4;
    const float* base;
    __device__ __forceinline__ XsS2(const XsLds& m, int l31, int hh) : base(m.Xs + l31 * LDT + 4 * hh) {}
    __device__ __forceinline__ const float* operator()(int f) const {
        return base + (32 * (f / NG)) * LDT + 8 * (f % NG);
    }
    __device__ __forceinline__ void run(const f32x16 (&A)[NJT], f32x16 (&U)[2], const XsFrag& pre) const {
        xs_zero(U[0]); xs_zero(U[1]);
        xs_stream<2 * NG>(*this, [&](int f, int e, float a) {
            const int ct = f / NG, g = f % NG;
            U[ct] = mfma32(a, A[g >> 2][4 * (g & 3) + e], U[ct]); }, pre);               // slot_attention.py:59
    }
};
// S3(gt): six 64-deep chains (W_ir U + W_hr h), (W_iz U + W_hz h), W_in U, W_hn h for hidden units 32gt..32gt+31;
// the accumulators start from the biases (b_ir+b_hr | b_iz+b_hz | b_in | b_hn): the gate block has no bias operands
struct XsS3 {
    const float* bi; const float* bh; const float* bias;
    __device__ __forceinline__ XsS3(const XsLds& m, int gt, int l31, int hh)
        : bi(m.Wih + (32 * gt + l31) * XS_LD + 4 * hh), bh(m.Whh + (32 * gt + l31) * XS_LD + 4 * hh),
          bias(m.bias + 32 * gt + 4 * hh) {}
    __device__ __forceinline__ const float* operator()(int f) const {
        const int c = f >> 3, g = f & 7;
        return ((c & 1) ? bh : bi) + (64 * (c >> 1)) * XS_LD + 32 * (g >> 2) + 8 * (g & 3);
    }
    __device__ __forceinline__ void init(f32x16 (&G)[4]) const {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *(const f32x4*)(bias + 64 * k + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) G[k][4 * q + e] = bv[e];
            }
    }
    __device__ __forceinline__ void run(const f32x16 (&U)[2], const f32x16 (&h)[2], f32x16 (&G)[4],
                                        const XsFrag& pre) const {
        xs_stream<48>(*this, [&](int f, int e, float a) {
            const int c = f >> 3, g = f & 7, k = c < 4 ? (c >> 1) : c - 2;
            const float bop = (c & 1) ? h[g >> 2][4 * (g & 3) + e] : U[g >> 2][4 * (g & 3) + e];
            G[k] = mfma32(a, bop, G[k]); }, pre);
    }
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
#define XS_LOG2E 1.4426950408889634f

// ---- V1: A = sigmoid(D / r_i * tau) (slot_attention.py:56-57) as rcp(1 + exp2(D * c)), c = -log2(e) * tau / r_i
// folded per slot; element pairs so that the multiply / add become packed-fp32 instructions.  VALU work cannot hide
// under this wave's own MFMAs (tools_dev/mfma_shadow_bench.hip: they serialise), so it is kept minimal: padded
// tokens need no mask (their X^T columns are zero), padded slots get c = 0 (finite garbage that is never stored).
template <int NJT, int NG>
__device__ __forceinline__ void xs_v1(f32x16 (&A)[NJT], float c) {
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            if (4 * jt + (e >> 2) >= NG) continue;           // (elements of padded-token groups: never read, see XsS2)
            f32x2 x = {A[jt][e], A[jt][e + 1]};
            x *= c;
            f32x2 ex = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
            ex += 1.f;
            A[jt][e] = __builtin_amdgcn_rcpf(ex[0]);
            A[jt][e + 1] = __builtin_amdgcn_rcpf(ex[1]);
        }
}
// ---- V2: GRU cell, gate order r, z, n (slot_attention.py:60-66); G = biased pre-activations (r | z | i_n | h_n)
__device__ __forceinline__ void xs_v2(const f32x16 (&G)[4], const f32x16& hold, f32x16& hnew) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        f32x2 xr = {G[0][r], G[0][r + 1]}, xz = {G[1][r], G[1][r + 1]};
        xr *= -XS_LOG2E; xz *= -XS_LOG2E;
        f32x2 er = {__builtin_amdgcn_exp2f(xr[0]), __builtin_amdgcn_exp2f(xr[1])};
        f32x2 ez = {__builtin_amdgcn_exp2f(xz[0]), __builtin_amdgcn_exp2f(xz[1])};
        er += 1.f; ez += 1.f;
        const f32x2 rg = {__builtin_amdgcn_rcpf(er[0]), __builtin_amdgcn_rcpf(er[1])};
        const f32x2 zg = {__builtin_amdgcn_rcpf(ez[0]), __builtin_amdgcn_rcpf(ez[1])};
        const f32x2 gi = {G[2][r], G[2][r + 1]}, gh = {G[3][r], G[3][r + 1]}, ho = {hold[r], hold[r + 1]};
        f32x2 t = (gi + rg * gh) * (2.f * XS_LOG2E);                     // tanh(x) = 1 - 2 / (1 + e^{2x})
        f32x2 et = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
        et += 1.f;
        const f32x2 rc = {__builtin_amdgcn_rcpf(et[0]), __builtin_amdgcn_rcpf(et[1])};
        const f32x2 ng = 1.f - 2.f * rc;
        const f32x2 hn = ng + zg * (ho - ng);                            // (1 - z) n + z h
        hnew[r] = hn[0];
        hnew[r + 1] = hn[1];
    }
}
__device__ __forceinline__ void xs_store_half(float* sp, const f32x16& h, int hh) {     // 32 of the 64 hidden units
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = h[4 * q + e];
        *(f32x4*)(sp + 8 * q + 4 * hh) = v;
    }
}

// per-tile view
struct XsTile { int ti; int i; bool iok; float r; };
struct XsOut {
    float* states; float* attn; float* usum; float* area_s; float* stage; float* exch;
    int S, N, lane, l31, hh, wave;
    bool vec4;
#ifdef XS_TIMING
    long long* stamps; int* nst;
#endif
};
#ifdef XS_TIMING
#define XS_STAMP_O(o) do { if (threadIdx.x == 0) (o).stamps[blockIdx.x * 64 + *(o).nst] = __builtin_readcyclecounter(); ++*(o).nst; } while (0)
#else
#define XS_STAMP_O(o) do { } while (0)
#endif

// One slot tile, one iteration.  MODE 0: the wave owns the tile.  MODE 1: the tile is SHARED by the wave pair
// (2p, 2p+1) -- both run S1 / V1 / S2 redundantly (bit-identical), each computes the GRU for 32 of the 64 hidden
// units and they swap halves through LDS: the ntiles % 4 leftover tiles cost 0.6 instead of 1.0 tile of the critical
// wave's time.  (The caller issues the workgroup barrier between xs_tile_shared_a and _b.)
// S1 -> V1 -> S2 of one tile; `next` (if given) is the gate stream that follows: its first fragments and bias
// accumulators are requested before V1 as well.  LAST (+ writer): attention map, logit partial sums, area.
template <int NJT, int NG, bool LAST>
__device__ __forceinline__ void xs_tile_head(const XsLds& m, const f32x16 (&h)[2], const XsTile& t, float tau,
                                             const XsOut& o, f32x16 (&U)[2], bool writer, const XsS3* next,
                                             f32x16 (&G)[4], XsFrag& next_pre) {
    f32x16 A[NJT];
    const XsS1<NJT> s1(m, o.l31, o.hh);
    const XsS2<NJT, NG> s2(m, o.l31, o.hh);
    s1.run(h, A, xs_pre(s1));
    if (LAST) XS_STAMP_O(o);
    const XsFrag pre2 = xs_pre(s2);
    if (!LAST) { next->init(G); next_pre = xs_pre(*next); }
    XS_SB();
    const float c = t.iok ? -XS_LOG2E * (tau * xs_recip(t.r)) : 0.f;
    xs_v1<NJT, NG>(A, c);
    if (LAST) XS_STAMP_O(o);
    s2.run(A, U, pre2);
    if (LAST) XS_STAMP_O(o);
    if (LAST && writer) {
        // attention map rows of this tile are one contiguous [rows][N] block in global memory: stage the tile in this
        // wave's LDS scratch (the GRU weights are dead in the last iteration) and copy it out with coalesced stores
        float asum = 0.f;
        float* st = o.stage;
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int j = xs_kidx(jt, e, o.hh);
                if (j < o.N) { st[o.l31 * o.N + j] = A[jt][e]; asum += A[jt][e]; }
            }
        asum = xs_wave_sum(t.iok ? asum : 0.f);
        const int rows = min(32, o.S - 32 * t.ti), cnt = rows * o.N;
        float* dst = o.attn + (long)32 * t.ti * o.N;
        if (o.vec4) {                   // 16-byte aligned block: all loads first, then all stores
            const int cnt4 = cnt >> 2;
            f32x4 buf[(32 * NJT * 32 / 4 + 63) / 64];
#pragma unroll
            for (int k = 0; k < (32 * NJT * 32 / 4 + 63) / 64; ++k)
                if (o.lane + 64 * k < cnt4) buf[k] = *(const f32x4*)(st + 4 * (o.lane + 64 * k));
#pragma unroll
            for (int k = 0; k < (32 * NJT * 32 / 4 + 63) / 64; ++k)
                if (o.lane + 64 * k < cnt4) *(f32x4*)(dst + 4 * (o.lane + 64 * k)) = buf[k];
            for (int k = 4 * cnt4 + o.lane; k < cnt; k += 64) dst[k] = st[k];
        } else {
            for (int k = o.lane; k < cnt; k += 64) dst[k] = st[k];
        }
        XS_STAMP_O(o);
        const float us = xs_rowsum<2>(U);
        if (o.hh == 0 && t.iok) o.usum[t.i] = us;
        if (o.lane == 0) o.area_s[t.ti] = asum;
    }
}
template <int NJT, int NG, bool LAST>
__device__ __forceinline__ void xs_tile_own(const XsLds& m, f32x16 (&h)[2], const XsTile& t, float tau, const XsOut& o,
                                            long it_row) {
    f32x16 U[2], G0[4], G1[4];
    const XsS3 s3a(m, 0, o.l31, o.hh), s3b(m, 1, o.l31, o.hh);
    XsFrag pa;
    xs_tile_head<NJT, NG, LAST>(m, h, t, tau, o, U, true, &s3a, G0, pa);
    if (!LAST) {
        f32x16 hn0, hn1;
        s3a.run(U, h, G0, pa);
        s3b.init(G1);
        const XsFrag pb = xs_pre(s3b);
        XS_SB();
        xs_v2(G0, h[0], hn0);
        s3b.run(U, h, G1, pb);
        xs_v2(G1, h[1], hn1); XS_SB();
        h[0] = hn0; h[1] = hn1;
        if (t.iok) {
            float* sp = o.states + (it_row + t.i) * XS_D;
            xs_store_half(sp, h[0], o.hh);
            xs_store_half(sp + 32, h[1], o.hh);
        }
    }
}
// shared tile, part a: everything up to this wave's half of the new state, parked in the exchange buffer.
// GX (three token tiles: no LDS left for the 16 KB exchange buffer): the halves are exchanged through the `states` rows the
// wave stores anyway -- the caller's barrier is then a full __syncthreads() (drains the stores) and part b re-reads the
// partner's half from L2 (same CU, same vector L1: coherent after the barrier).
template <int NJT, int NG, bool LAST, bool GX>
__device__ __forceinline__ void xs_tile_shared_a(const XsLds& m, f32x16 (&h)[2], const XsTile& t, float tau,
                                                 const XsOut& o, long it_row) {
    f32x16 U[2], G[4];
    const int gt = o.wave & 1;
    const XsS3 s3(m, gt, o.l31, o.hh);
    XsFrag pg;
    xs_tile_head<NJT, NG, LAST>(m, h, t, tau, o, U, gt == 0, &s3, G, pg);
    if (!LAST) {
        f32x16 hn;
        s3.run(U, h, G, pg);
        if (gt) xs_v2(G, h[1], hn); else xs_v2(G, h[0], hn);
        XS_SB();
        if (gt) h[1] = hn; else h[0] = hn;
        if constexpr (!GX) {
            float* ex = o.exch + o.wave * 1024 + o.lane * 4;          // [q][lane][4]
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = hn[4 * q + e];
                *(f32x4*)(ex + q * 256) = v;
            }
        }
        if (t.iok) xs_store_half(o.states + (it_row + t.i) * XS_D + 32 * gt, hn, o.hh);
    }
}
// part b (after the barrier): the partner's half
template <bool GX>
__device__ __forceinline__ void xs_tile_shared_b(f32x16 (&h)[2], const XsOut& o, const XsTile& t, long it_row) {
    const int gt = o.wave & 1;
    f32x16 hp;
    if constexpr (GX) {
        const float* sp = o.states + (it_row + (t.iok ? t.i : 0)) * XS_D + 32 * (gt ^ 1) + 4 * o.hh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *(const f32x4*)(sp + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) hp[4 * q + e] = t.iok ? v[e] : 0.f;         // (padded slots carry nothing)
        }
    } else {
        const float* ex = o.exch + (o.wave ^ 1) * 1024 + o.lane * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *(const f32x4*)(ex + q * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e) hp[4 * q + e] = v[e];
        }
    }
    if (gt) h[0] = hp; else h[1] = hp;
}

template <int NJT, int TPW, int NG = 4 * NJT>
__global__ __launch_bounds__(64 * XS_FWD_WAVES) void xslot_fwd_kernel(XsFwdArgs a) {
#ifdef XS_TIMING
    int nstamp = 0;
#endif
    XS_STAMP();
    constexpr int NP = 32 * NJT;
    constexpr int NTHR = 64 * XS_FWD_WAVES, NW = XS_FWD_WAVES;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Xs = lds;                        // [64][NP+4]  X^T / 64 (columns >= N zero)
    float* Ks = Xs + XS_XT_FLOATS(NP);            // [NP][68]  K / 8, K = to_k(X + PE) (rows >= N zero)
    float* Wih = Ks + NP * XS_LD;           // [192][68]
    float* Whh = Wih + 192 * XS_LD;         // [192][68]
    float* bias = Whh + 192 * XS_LD;        // br | bz | b_in | b_hn  (4 x 64)
    double* ksum = (double*)(bias + 256);   // [64]  column sums of K/8 (fp64)
    double* tau_part = ksum + 64;           // [T<=8][16]
    float* area_s = (float*)(tau_part + 128);   // [16]
    float* usum = area_s + 16;              // [<=512] per-slot sum_k U_T[i][k]
    // MLP scratch aliases the GRU weight region; the weights wait in registers until the MLP is done
    float* H0 = Wih;                        // [NP][68]
    float* H1 = H0 + NP * XS_LD;            // [NP][68]
    float* Wt = H1 + NP * XS_LD;            // [64][68]
    static_assert(2 * NP + 64 + 4 * 32 <= 384, "MLP scratch + slot transpose scratch fit in the GRU weight region");

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, hh = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x, N = a.N, S = a.S;
    const float* Xg = a.X + (long)b * N * XS_D;
    const float scale = 0.125f, inv_d = 1.f / XS_D;      // d^-1/2 and 1/d, d = 64

    // ---- phase 0: every global operand is requested up front (one latency, not five): tokens, first to_k weight,
    // both GRU weight matrices (W_hh parked in registers), initial slots
    constexpr int XQ = NP * 16 / NTHR, WQ = 192 * 16 / NTHR, TQ = 64 * 16 / NTHR;
    float bias_v = 0.f;
    if (tid < 256) {
        const int g = tid & 63, k = tid >> 6;
        bias_v = k == 0 ? a.b_ih[g] + a.b_hh[g] : k == 1 ? a.b_ih[64 + g] + a.b_hh[64 + g]
               : k == 2 ? a.b_ih[128 + g] : a.b_hh[128 + g];
    }
    f32x4 xr[XQ], pr[XQ], wtr[TQ], wir[WQ], whr[WQ];
#pragma unroll
    for (int k = 0; k < XQ; ++k) {
        const int c = tid + k * NTHR, r = c >> 4, q = c & 15;
        xr[k] = f32x4{0.f, 0.f, 0.f, 0.f}; pr[k] = xr[k];
        if (r < N) { xr[k] = *(const f32x4*)(Xg + r * XS_D + q * 4); pr[k] = *(const f32x4*)(a.PE + r * XS_D + q * 4); }
    }
#pragma unroll
    for (int k = 0; k < TQ; ++k) wtr[k] = *(const f32x4*)(a.tok_w[0] + (tid + k * NTHR) * 4);
#pragma unroll
    for (int k = 0; k < WQ; ++k) wir[k] = *(const f32x4*)(a.w_ih + (tid + k * NTHR) * 4);
#pragma unroll
    for (int k = 0; k < WQ; ++k) whr[k] = *(const f32x4*)(a.w_hh + (tid + k * NTHR) * 4);
    // slot tiles: wave w owns tiles w, w+4, ...; the ntiles % 4 leftover tiles go to waves 0.. as whole tiles, or --
    // when there are one or two of them -- each is SHARED by a wave pair (exchange through LDS, NJT <= 2, or through the
    // state rows in L2, NJT = 3)
    const int ntiles = (S + 31) >> 5, nfull4 = ntiles >> 2, rem = ntiles & 3;
    constexpr bool GX = NJT > 2;                // no LDS for the exchange buffer: through the `states` rows (xs_tile_shared_a)
    const bool share = (rem == 1 || rem == 2) && (!GX || a.T > 1);
    int tile_id[TPW];
    f32x16 h[TPW][2];
    // initial slots: each wave fetches its tiles as whole 256-byte rows (coalesced; a register-layout gather straight
    // from global memory would touch 4x the cache lines) and transposes them through a private LDS scratch below
    f32x4 sr[TPW][8];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        int ti = wave + NW * tt;
        if (tt >= nfull4) ti = share ? ((wave >> 1) < rem ? 4 * nfull4 + (wave >> 1) : -1)
                                     : (tt == nfull4 && wave < rem ? 4 * nfull4 + wave : -1);
        tile_id[tt] = ti;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = ti * 32 + 4 * k + (lane >> 4);
            sr[tt][k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ti >= 0 && row < S) sr[tt][k] = *(const f32x4*)(a.slots0 + row * XS_D + (lane & 15) * 4);
        }
    }
#pragma unroll
    for (int k = 0; k < XQ; ++k) {
        const int c = tid + k * NTHR, r = c >> 4, q = c & 15;
#pragma unroll
        for (int e = 0; e < 4; ++e) Xs[(q * 4 + e) * (NP + 4) + r] = xr[k][e] * inv_d;      // Xt[c][j] = X[j][c] / d
        *(f32x4*)(H0 + r * XS_LD + q * 4) = xr[k] + pr[k];
        if (r < N) *(f32x4*)(a.Hsave + ((long)b * N + r) * XS_D + q * 4) = xr[k] + pr[k];     // input of layer 0
    }
#pragma unroll
    for (int k = 0; k < TQ; ++k) { const int c = tid + k * NTHR; *(f32x4*)(Wt + (c >> 4) * XS_LD + (c & 15) * 4) = wtr[k]; }
    if (tid < 256) bias[tid] = bias_v;
    {
        float* tsc = Wt + 64 * XS_LD + wave * (32 * XS_LD);          // private [32][68] scratch behind the MLP buffers
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
#pragma unroll
            for (int k = 0; k < 8; ++k) *(f32x4*)(tsc + (4 * k + (lane >> 4)) * XS_LD + (lane & 15) * 4) = sr[tt][k];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *(const f32x4*)(tsc + l31 * XS_LD + 32 * t + 8 * q + 4 * hh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[tt][t][4 * q + e] = v[e];
                }
        }
    }
    XS_STAMP();
    // ---- phase 1: to_k MLP (Linear, then (ReLU, Linear)*), output tiles (jt, ot) spread over the waves
    float* Hin = H0;
    float* Hout = H1;
    for (int l = 0; l < a.L; ++l) {
        if (l > 0) {              // this layer's weights were requested while the previous layer multiplied (wtr)
            __syncthreads();
#pragma unroll
            for (int k = 0; k < TQ; ++k) { const int c = tid + k * NTHR; *(f32x4*)(Wt + (c >> 4) * XS_LD + (c & 15) * 4) = wtr[k]; }
        }
        __syncthreads();
        if (l + 1 < a.L) {
#pragma unroll
            for (int k = 0; k < TQ; ++k) wtr[k] = *(const f32x4*)(a.tok_w[l + 1] + (tid + k * NTHR) * 4);
        }
        const bool last = l == a.L - 1;
        for (int tile = wave; tile < NJT * 2; tile += NW) {
            const int jt = tile >> 1, ot = tile & 1;
            const float bo = a.tok_b[l][32 * ot + l31];          // (requested in front of the MFMAs, used behind them)
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
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * jt + mfma32_row(r, lane);
                float v = acc[r] + bo;
                if (!last) v = fmaxf(v, 0.f);
                if (last) {
                    v = j < N ? v : 0.f;
                    Ks[j * XS_LD + o] = v * scale;
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
    XS_STAMP();
    // ---- phase 2: GRU weights out of the registers (the MLP scratch they alias is dead), column sums of K
    xs_colsum_f64(Ks, NP, ksum, tid);
#pragma unroll
    for (int k = 0; k < WQ; ++k) {
        const int c = tid + k * NTHR;
        *(f32x4*)(Wih + (c >> 4) * XS_LD + (c & 15) * 4) = wir[k];
        *(f32x4*)(Whh + (c >> 4) * XS_LD + (c & 15) * 4) = whr[k];
    }
    __syncthreads();
    XS_STAMP();

    // ---- phase 3: iterations
    const XsLds m{Xs, Ks, Wih, Whh, bias};
    float* exch = usum + 512;                                   // [4 waves][4][64][4] (only when NJT <= 2)
    const XsOut o{a.states, a.attn + (long)b * S * N, usum, area_s, Wih + wave * (32 * XS_MAX_N), exch,
                  S, N, lane, l31, hh, wave, ((S * N) & 3) == 0 && ((size_t)a.attn & 15) == 0
#ifdef XS_TIMING
                  , a.stamps, &nstamp
#endif
    };
    for (int it = 0; it < a.T; ++it) {
        const bool last = it == a.T - 1;
        const long it_row = ((long)it * a.B + b) * S;
        XsTile tl[TPW];
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const int ti = tile_id[tt];
            tl[tt].ti = ti;
            tl[tt].i = ti * 32 + l31;
            tl[tt].iok = ti >= 0 && tl[tt].i < S;
            tl[tt].r = 0.f;
            if (ti >= 0) {
                double r64 = xs_rowdot_f64(h[tt], ksum, hh);                  // r_i = s_i . sum_j K_j / 8
                if (!tl[tt].iok) r64 = 0.0;                                   // padded slots do not enter tau
                tl[tt].r = (float)r64;
                const double tr = xs_tilesum_f64(r64);
                const bool writes = !(share && tt == TPW - 1) || !(wave & 1);
                if (lane == 0 && writes) tau_part[it * 16 + ti] = tr;
            }
        }
        XS_STAMP();
        xs_lds_barrier();
        XS_STAMP();
        double tau64 = 0.0;
        for (int k = 0; k < ntiles; ++k) tau64 += tau_part[it * 16 + k];
        const float tau = (float)tau64;
        XS_STAMP();
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            if (tl[tt].ti < 0) continue;
            if (share && tt == TPW - 1) {
                if (last) { if (!(wave & 1)) xs_tile_shared_a<NJT, NG, true, GX>(m, h[tt], tl[tt], tau, o, it_row); }
                else xs_tile_shared_a<NJT, NG, false, GX>(m, h[tt], tl[tt], tau, o, it_row);
            } else {
                if (last) xs_tile_own<NJT, NG, true>(m, h[tt], tl[tt], tau, o, it_row);
                else xs_tile_own<NJT, NG, false>(m, h[tt], tl[tt], tau, o, it_row);
            }
            XS_STAMP();
        }
        if (share && !last) {
            if constexpr (GX) __syncthreads();          // (also drains this wave's stores of its half: the partner reads them)
            else xs_lds_barrier();
            if (tl[TPW - 1].ti >= 0) xs_tile_shared_b<GX>(h[TPW - 1], o, tl[TPW - 1], it_row);
        }
    }
    __syncthreads();
    XS_STAMP();
    // ---- class aggregation (slot_attention.py:87-91,96): logits_c = ls * sum_{s in c} sum_k U[s][k]
    for (int c = tid; c < a.C; c += NTHR) {
        float v = 0.f;
        for (int k = 0; k < a.spc; ++k) v += usum[c * a.spc + k];
        a.logits[(long)b * a.C + c] = a.loss_status * v;
    }
    if (tid == 0) {
        float s = 0.f;
        for (int k = 0; k < ntiles; ++k) s += area_s[k];
        a.area_part[b] = s;
    }
    XS_STAMP();
}

#include "xslot_small_fwd.h"

// SCOUTER_XSLOT_SMALL=0 keeps heads with <= 16 slots on the 32-slot-tile kernels (A/B runs, tests of those paths)
static bool xs_small_enabled() {          // (read at every launch: the tests flip it inside one process)
    const char* e = getenv("SCOUTER_XSLOT_SMALL");
    return !(e && e[0] == '0');
}

static size_t xs_fwd_lds_bytes(int NJT) {
    return (size_t)(XS_XT_FLOATS(32 * NJT) + 32 * NJT * XS_LD + 384 * XS_LD + 256 + 2 * (64 + 128) + 16 + 512 +
                    (NJT <= 2 ? 4096 : 0)) * sizeof(float);
}

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
    const int NW = XS_FWD_WAVES, TPW = (ntiles + NW - 1) / NW;
    const size_t lds = xs_fwd_lds_bytes(NJT);
    hipStream_t st = (hipStream_t)stream;
    const double flops = (double)B * (2.0 * L * N * d * d + (double)T * 4.0 * S * N * d + (T - 1) * 12.0 * S * d * d);
    ScProfScope prof("xslot_fwd", st, flops, 4.0 * B * (2.0 * N * d + (double)S * N));
    if (S <= 16 && xs_small_enabled()) {          // the metric's own head (10 slots): xslot_small_fwd.h
        if (N <= 64) hipLaunchKernelGGL(xslot_small_fwd_kernel<1>, dim3(B), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(xslot_small_fwd_kernel<2>, dim3(B), dim3(256), 0, st, a);
        return sc_check_launch("xslot_small_fwd");
    }
#define XS_LAUNCH(NJT_, TPW_, ...)                                                                                  \
    do {                                                                                                            \
        auto kern = xslot_fwd_kernel<NJT_, TPW_, ##__VA_ARGS__>;                                                    \
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
        hipLaunchKernelGGL(kern, dim3(B), dim3(NW * 64), lds, st, a);                                               \
    } while (0)
#define XS_TPW(NJT_, ...)                                             \
    do {                                                              \
        if (TPW == 1) XS_LAUNCH(NJT_, 1, ##__VA_ARGS__);              \
        else if (TPW == 2) XS_LAUNCH(NJT_, 2, ##__VA_ARGS__);         \
        else if (TPW == 3) XS_LAUNCH(NJT_, 3, ##__VA_ARGS__);         \
        else XS_LAUNCH(NJT_, 4, ##__VA_ARGS__);                       \
    } while (0)
#ifdef XS_DEV_SINGLE_INST
    XS_LAUNCH(2, 3);      // dev builds (tools_dev/xs_phase_timing.hip): one instantiation compiles much faster
#else
    if (NJT == 1) XS_TPW(1);
    else if (NJT == 2) XS_TPW(2);
    else if (N <= 88) XS_TPW(3, 11);      // (the reference's 9x9 grid: the token contraction skips the last 8-token group)
    else XS_TPW(3);
#endif
#undef XS_TPW
#undef XS_LAUNCH
    return sc_check_launch("xslot_fwd");
}
