// Shared pieces of the fused xSlot kernels (forward: xslot_fwd.hip, backward: xslot_bwd.hip).
//
// Register choreography ("slot-per-lane"): every per-slot matrix M[i][k] (slots s, updates U, gate
// pre-activations, D^T columns ...) of a 32-slot tile is held TRANSPOSED in MFMA accumulator layout:
//     lane (i = lane&31, hh = lane>>5), register r of k-tile t   <->   M[i][ kidx(t, r, hh) ]
//     kidx(t, r, hh) = 32 t + 8 (r>>2) + 4 hh + (r&3)
// which is exactly what v_mfma_f32_32x32x2_f32 produces for  Out^T[k][i] = sum_c W[k][c] * In^T[c][i]
// (rows = k, cols = lane&31 = i).  Such a register set can be fed straight back as the B operand
// (B[k = hh][n = i]) of the next MFMA, one register per k-step, so
//     D^T = K s^T  ->  A = sigmoid(...)  ->  U^T = X^T A^T  ->  gates^T = W U^T + W h^T  ->  s'^T
// never leaves the register file.  The A operands come from LDS: either a row-major matrix whose rows are
// the OUTPUT index (k contiguous: four k-steps per ds_read_b128, "mm_kc") or one whose rows are the
// CONTRACTION index (ds_read_b32, lanes along the output index, "mm_tr").
#pragma once
#include "common.h"

#define XS_D 64          // hidden_dim (train.py:53); the kernels are specialised for d = 64
#define XS_LD 68         // LDS row stride (floats): 16B-aligned rows, conflict-free ds_read_b128 over 16 rows
#define XS_MAX_N 96      // tokens per image: N <= 96 (7x7 and the reference's 9x9 grid); NJT = ceil(N/32) token tiles

// Ends a scheduling region: without it the compiler hoists the LDS operand loads of many MFMA blocks ahead
// (everything is unrolled) and runs out of the 512-entry register file.
#define XS_REGION_END() __builtin_amdgcn_sched_barrier(0)

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS traffic (lgkmcnt(0)), not for its outstanding
// global stores -- __syncthreads() would also drain those (vmcnt(0)), i.e. stall every iteration on the write
// acknowledgements of the slot states just stored.
__device__ __forceinline__ void xs_lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ int xs_kidx(int t, int r, int hh) { return 32 * t + 8 * (r >> 2) + 4 * hh + (r & 3); }

// acc[rows row0..row0+31][i] += sum_{k<64} M[row0 + l31][k] * B^T[k][i] ; M row-major in LDS (k contiguous)
// `_open` variants leave the scheduling region open so that a caller can interleave independent VALU work into the
// MFMA shadow (xslot_fwd.hip); the plain ones close it.
__device__ __forceinline__ void xs_mm_kc_open(const float* __restrict__ M, int row0, const f32x16 (&b)[2], f32x16& acc,
                                              int l31, int hh) {
    // operand fragments are requested two ahead of use and the order is pinned: left to itself the compiler issues
    // each ds_read_b128 right before its four MFMAs and the LDS latency is exposed once per fragment (~20 %)
    const float* base = M + (row0 + l31) * XS_LD + 4 * hh;
    f32x4 fr[3];
    fr[0] = *(const f32x4*)(base);
    fr[1] = *(const f32x4*)(base + 8);
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        __builtin_amdgcn_sched_barrier(0);
        if (f + 2 < 8) fr[(f + 2) % 3] = *(const f32x4*)(base + 32 * ((f + 2) >> 2) + 8 * ((f + 2) & 3));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma32(fr[f % 3][e], b[f >> 2][4 * (f & 3) + e], acc);
    }
}
__device__ __forceinline__ void xs_mm_kc(const float* __restrict__ M, int row0, const f32x16 (&b)[2], f32x16& acc,
                                         int l31, int hh) {
    xs_mm_kc_open(M, row0, b, acc, l31, hh);
    XS_REGION_END();
}
// acc[cols col0..col0+31][i] += sum_{k<64} Mt[k][col0 + l31] * B^T[k][i] ; Mt row-major in LDS with rows = k
template <int NT>
__device__ __forceinline__ void xs_mm_tr_open(const float* __restrict__ Mt, int col0, const f32x16 (&b)[NT],
                                              f32x16& acc, int l31, int hh) {
    // one ds_read_b32 per MFMA, requested four MFMAs ahead
    const float* base = Mt + col0 + l31;
    float fr[5];
#pragma unroll
    for (int k = 0; k < 4; ++k) fr[k] = base[xs_kidx(k >> 4, k & 15, hh) * XS_LD];
#pragma unroll
    for (int k = 0; k < 16 * NT; ++k) {
        __builtin_amdgcn_sched_barrier(0);
        if (k + 4 < 16 * NT) fr[(k + 4) % 5] = base[xs_kidx((k + 4) >> 4, (k + 4) & 15, hh) * XS_LD];
        acc = mfma32(fr[k % 5], b[k >> 4][k & 15], acc);
    }
}
template <int NT>
__device__ __forceinline__ void xs_mm_tr(const float* __restrict__ Mt, int col0, const f32x16 (&b)[NT], f32x16& acc,
                                         int l31, int hh) {
    xs_mm_tr_open<NT>(Mt, col0, b, acc, l31, hh);
    XS_REGION_END();
}

__device__ __forceinline__ void xs_zero(f32x16& v) {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = 0.f;
}
// Gate non-linearities on the hardware transcendental units (v_exp_f32 / v_rcp_f32, 1 ulp each): 4 instructions per
// sigmoid instead of the ~25 of libm's expf + IEEE division, absolute error <= 2.4e-7 -- fp32-rounding level, far
// inside the 1e-4 parity bound; the backward recomputes with the same helpers, so forward and backward agree.
__device__ __forceinline__ float xs_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
// x / r from a per-slot reciprocal refined by one Newton step: q = x*ir, q += (x - r*q)*ir.  The residual is exact
// (FMA), so the quotient is the correctly rounded one in all but rare double-rounding cases -- in particular
// x == r gives exactly 1, which the reference's D / r_i produces for a single-token grid (its attention maps are then
// constant and the `--vis` normalisation 0/0; reproduced bit for bit by tests/test_xslot_gpu.py::one_token).
__device__ __forceinline__ float xs_recip(float r) {
    float ir = __builtin_amdgcn_rcpf(r);
    return __builtin_fmaf(__builtin_fmaf(-r, ir, 1.f), ir, ir);
}
__device__ __forceinline__ float xs_div(float x, float r, float ir) {
    const float q = x * ir;
    return __builtin_fmaf(__builtin_fmaf(-r, q, x), ir, q);
}
__device__ __forceinline__ float xs_tanh(float x) {        // 1 - 2 / (1 + e^{2x}); saturates correctly at +-inf
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * 2.8853900817779268f));
}

// x(lane) + x(lane ^ 32) in every lane: one v_permlane32_swap per 32-bit word (rows 2,3 of the first operand are
// exchanged with rows 0,1 of the second) instead of a ds_bpermute round trip
__device__ __forceinline__ float xs_halfsum(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ double xs_halfsum_f64(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
}
// sum over all 64 lanes, wave-uniform result: DPP row reductions + four scalar lane reads
template <int CTRL>
__device__ __forceinline__ float xs_dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float xs_wave_sum(float v) {
    v += xs_dpp_f32<0xB1>(v);
    v += xs_dpp_f32<0x4E>(v);
    v += xs_dpp_f32<0x141>(v);
    v += xs_dpp_f32<0x140>(v);
    const int b = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48)));
}

// in-lane sum of an NT-tile register set + the partner half-wave: sum over all 32*NT k of M[i][k]
template <int NT>
__device__ __forceinline__ float xs_rowsum(const f32x16 (&m)[NT]) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += m[t][r];
    return xs_halfsum(s);
}
// r_i = sum_j D_ij = d^-1/2 * s_i . (sum_j K_j): the normaliser the reference divides by (slot_attention.py:56) is
// ill-conditioned (mixed-sign terms cancel), so it is evaluated in fp64 from the fp32 operands -- by linearity this
// is the same quantity as the row sum of D, without the fp32 cancellation noise.  ksum: [64] doubles in LDS.
__device__ __forceinline__ double xs_rowdot_f64(const f32x16 (&s)[2], const double* __restrict__ ksum, int hh) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;            // four chains: the DFMAs pipeline instead of waiting
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double2 k01 = *(const double2*)(ksum + 32 * t + 8 * q + 4 * hh);
            const double2 k23 = *(const double2*)(ksum + 32 * t + 8 * q + 4 * hh + 2);
            a0 += (double)s[t][4 * q] * k01.x;
            a1 += (double)s[t][4 * q + 1] * k01.y;
            a2 += (double)s[t][4 * q + 2] * k23.x;
            a3 += (double)s[t][4 * q + 3] * k23.y;
        }
    return xs_halfsum_f64((a0 + a1) + (a2 + a3));
}
// sum over the 32 slots of a tile of a per-slot fp64 value (both half-waves hold it): data-parallel-primitive row
// reductions (quad swaps, half-row and row mirrors: ~10 cycles each) + two scalar lane reads, instead of five
// ds_bpermute round trips through the LDS crossbar.  Result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ double xs_dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double xs_tilesum_f64(double v) {
    v += xs_dpp_f64<0xB1>(v);          // quad_perm [1,0,3,2]
    v += xs_dpp_f64<0x4E>(v);          // quad_perm [2,3,0,1]
    v += xs_dpp_f64<0x141>(v);         // row_half_mirror
    v += xs_dpp_f64<0x140>(v);         // row_mirror: every lane of a 16-lane row holds the row sum
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    return r0 + r1;
}
// column sums of K (rows < NP of an [NP][XS_LD] LDS matrix; rows >= N are zero) in fp64, by threads 0..63
__device__ __forceinline__ void xs_colsum_f64(const float* __restrict__ Ks, int NP, double* __restrict__ ksum, int tid) {
    if (tid < XS_D) {
        double a = 0.0;
        for (int j = 0; j < NP; ++j) a += (double)Ks[j * XS_LD + tid];
        ksum[tid] = a;
    }
}

// sum over the 32 slots of a tile (lanes 0..31; both half-waves hold the same per-slot value)
__device__ __forceinline__ float xs_tilesum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// cooperative copy of a [rows][64] row-major global matrix into LDS [rows][XS_LD]
__device__ __forceinline__ void xs_load_mat(float* __restrict__ dst, const float* __restrict__ src, int rows,
                                            int tid, int nthreads) {
    for (int c = tid; c < rows * 16; c += nthreads) {
        const int r = c >> 4, q = c & 15;
        *(f32x4*)(dst + r * XS_LD + q * 4) = *(const f32x4*)(src + r * XS_D + q * 4);
    }
}
// the same with a compile-time shape: every 16-byte load of the thread is in flight before the first LDS store (the
// loop above leaves one load per round trip -- 12 round trips for a GRU weight matrix)
template <int ROWS, int NTHR>
__device__ __forceinline__ void xs_load_mat_c(float* __restrict__ dst, const float* __restrict__ src, int tid) {
    constexpr int IT = (ROWS * 16 + NTHR - 1) / NTHR;
    f32x4 v[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int c = tid + k * NTHR;
        if (ROWS * 16 % NTHR == 0 || c < ROWS * 16) v[k] = *(const f32x4*)(src + (c >> 4) * XS_D + (c & 15) * 4);
    }
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int c = tid + k * NTHR;
        if (ROWS * 16 % NTHR == 0 || c < ROWS * 16) *(f32x4*)(dst + (c >> 4) * XS_LD + (c & 15) * 4) = v[k];
    }
}
