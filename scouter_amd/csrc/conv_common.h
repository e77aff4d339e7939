// Pieces shared by the fp32 (conv_igemm.hip) and bf16 (conv_bf16.hip) implicit-GEMM convolution kernels: problem
// geometry, division-free pixel unflattening, and the block epilogue (LDS-staged 16-byte stores with fused bias /
// addend / ReLU and the fused fp64 BatchNorm statistics).
#pragma once
#include "common.h"

struct ConvGeom {
    int B, H, W, C;   // tensor feeding the A operand, NHWC, C = all channels
    int Ho, Wo;       // pixel grid of the GEMM rows
    int N;            // channels (all groups) of the GEMM-column tensor
    int R, S, stride, pad, groups;
    int Cg, Ng;       // per-group channels on the K side / the column side
    long M;           // B*Ho*Wo
    int wrow, wtap;   // weight strides (elements): fwd row=k -> N, tap -> Cg*N ; dgrad row=n -> C, tap -> Ng*C
    // reciprocals for the division-free pixel unflattening in the kernels (filled by launch_igemm_s)
    double inv_hw;    // 1 / (Ho*Wo)
    float inv_wo, inv_ho;
};

// floor(n / d) for 0 <= n < 2^20 with inv = 1/d: (n + 0.5) / d is at least 0.5/d away from every integer, far more
// than the float rounding error at these magnitudes, so truncation is exact (3 VALU instead of ~40 for a division)
__device__ __forceinline__ int fast_div(int n, float inv) { return (int)(((float)n + 0.5f) * inv); }


// Split-K weight-gradient grids are (tiles = taps x ci-tiles x co-tiles x groups, splits = pixel ranges).  All tiles of
// one pixel range read the same activation / gradient rows; workgroups are handed to the 8 XCDs round-robin in linear
// order, so the natural order scatters them over all eight L2s and the rows come from HBM once per tap (PMC: 1.9 GB
// fetched per launch for the 32-channel stem, 8x its algorithmic bytes).  This bijection keeps every pixel range on ONE
// XCD, its tiles back to back: the re-reads hit that XCD's L2.
__device__ __forceinline__ void wgrad_block_coords(int& tile, int& split) {
    const int T = gridDim.x, S = gridDim.y, S8 = S & ~7;
    const int L = blockIdx.x + T * blockIdx.y;
    tile = blockIdx.x;
    split = blockIdx.y;
    if (L < T * S8) {
        const int xcd = L & 7, idx = L >> 3;
        split = xcd + 8 * (idx / T);
        tile = idx % T;
    }
}

// deterministic sum of split-K slabs (conv_igemm.hip)
void sc_launch_slab_reduce(const float* part, float* dst, long n, int splits, long slab, hipStream_t st);

// ---- the split-K slab sum INSIDE the weight-gradient kernel: the LAST workgroup of an output tile to
// arrive sums that tile's `splits` partial tiles in slab order 0, 1, 2, ... (a fixed order: deterministic, whichever
// workgroup happens to be last) and writes dW -- no slab_reduce launch (35 per step, 0.36 ms).  Cross-XCD protocol (each
// XCD has its own, non-coherent L2):
//   * partial tiles are stored with agent-scope relaxed atomic stores (write-through: sc1) -- slab_store();
//   * s_waitcnt vmcnt(0) + workgroup barrier: every partial of this workgroup has reached the device coherence point;
//   * thread 0 increments the tile's arrival counter (agent-scope atomic RMW); the workgroup that reads splits - 1 is last,
//     resets the counter to 0 (the caller's buffer is zero on entry and left zero) and
//   * reads all partials back with agent-scope atomic loads (sc1: past the local L2), 8 bytes each, eight in flight.
// `arrival` == NULL: the caller launches slab_reduce_kernel as before (plain stores).
#ifndef SC_SLAB_MODE
#define SC_SLAB_MODE 1      // 0: write-through atomic stores + sc1 atomic loads (first version: -17 % images/sec, kept for the A/B);
#endif                      // 1: plain stores + agent-scope release fence, acquire fence + plain vector loads
__device__ __forceinline__ void slab_store(float* p, float v, bool coherent) {
    if (SC_SLAB_MODE == 0 && coherent) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// ws: split 0's slab; the tile = ntaps x [nrows][ncols] elements at ws + tap * tap_stride + row0 + r * row_stride + col0 + c
__device__ __forceinline__ void slab_tile_finish(const float* ws, float* dw, long slab, int splits, unsigned* counter,
                                                 int ntaps, long tap_stride, long row0, int nrows, long row_stride, int col0,
                                                 int ncols) {
    __shared__ int s_last;
    if (SC_SLAB_MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");     // partials leave this XCD's L2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == (unsigned)(splits - 1);
        if (s_last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    if (SC_SLAB_MODE == 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                        // nothing stale from this XCD's L2
        const int quads = ncols >> 2, per_tap = nrows * quads;
        for (int e = threadIdx.x; e < ntaps * per_tap; e += blockDim.x) {
            const int t = e / per_tap, r = (e - t * per_tap) / quads, c4 = e % quads;
            const long a = (long)t * tap_stride + row0 + (long)r * row_stride + col0 + 4 * c4;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            int k = 0;
            for (; k + 8 <= splits; k += 8) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(ws + (long)(k + u) * slab + a);
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; k < splits; ++k) s += *(const f32x4*)(ws + (long)k * slab + a);
            *(f32x4*)(dw + a) = s;
        }
        return;
    }
    typedef unsigned long long u64;
    const int pairs = ncols >> 1, per_tap = nrows * pairs;
    for (int e = threadIdx.x; e < ntaps * per_tap; e += blockDim.x) {
        const int t = e / per_tap, r = (e - t * per_tap) / pairs, c2 = e % pairs;
        const long a = (long)t * tap_stride + row0 + (long)r * row_stride + col0 + 2 * c2;
        float s0 = 0.f, s1 = 0.f;
        int k = 0;
        for (; k + 8 <= splits; k += 8) {
            u64 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = __hip_atomic_load((const u64*)(ws + (long)(k + u) * slab + a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s0 += __builtin_bit_cast(float, (unsigned)v[u]);
                s1 += __builtin_bit_cast(float, (unsigned)(v[u] >> 32));
            }
        }
        for (; k < splits; ++k) {
            const u64 v = __hip_atomic_load((const u64*)(ws + (long)k * slab + a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s0 += __builtin_bit_cast(float, (unsigned)v);
            s1 += __builtin_bit_cast(float, (unsigned)(v >> 32));
        }
        *(float2*)(dw + a) = float2{s0, s1};
    }
}

// BatchNorm BACKWARD reductions fused into the epilogue of the input-gradient kernel that produces the gradient of a
// block output  out = relu(bn(x1) [+ bn(x2) | + shortcut])  (resnet.py:404-412, resnest.py:128-143): the value about to
// be stored is d(out); the epilogue applies the ReLU sign (1 bit / element, written by the forward apply pass), stores
// g = d(out) * [out > 0] instead, and reduces  sum g  and  sum g * xhat1  (and  sum g * xhat2  for the BatchNorm of a
// downsample branch fed by the same gradient) per channel and M tile in fp64 -- the separate colsum pass over (dy, x)
// and the masked-gradient copy `gout` of scouter_bn_bwd_f32 disappear (12 of 32 bytes per element of that backward).
struct BnBwdFuse {
    const unsigned long long* mask;                        // may be NULL: no ReLU between the BatchNorm and the consumer
    const float* x1; const float* sv1; double* part1;      // BatchNorm input, saved [4][C] block, partial rows [mt][C][2]
    const float* x2; const float* sv2; double* part2;      // second BatchNorm (may be NULL)
    int io;                                                // bit 0 / 1 / 2: x1 / x2 / the addend is STORED as bf16 (bf16-input kernels only);
                                                           // bit 3: the OUTPUT is (plane forward kernels, which have no other flag)
};

// conv_xhalo.hip: the persistent resident-rows kernel with the operand split in registers (3x3 / stride 1 / pad 1 layers with 32
// GEMM columns per group); reached through tile 7 of the scouter_conv2d_{fwd,dgrad}_x3 entry points (conv_x3.hip)
bool sc_xhalo_ok(const ConvGeom& g);
int sc_xhalo_partial_rows(int groups);
void sc_launch_xhalo(bool dgrad, const float* a, const void* w_planes, long w_plane_elems, const float* addend, float* dst,
                     double* bn_part, const ConvGeom& g, int relu, hipStream_t st, const BnBwdFuse& fz);

// Operands of the fused BatchNorm-backward epilogue that do not depend on the GEMM (shortcut gradient, BatchNorm inputs):
// requested at the START of the workgroup (igemm_epilogue_prefetch), so their latency overlaps the K loop -- these launches
// are HBM-latency bound (tools_dev/fused_dgrad_tiles.py).  NR = rows per lane of the epilogue's row-major pass (<= 8: the
// 64x64, 128x64, 128x32 tiles; the instantiation's register count is set by the epilogue anyway, occupancy by LDS).
template <int NR>
struct EpiPre { f32x4 add[NR], x1[NR], x2[NR]; bool on; };

template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void igemm_epilogue_prefetch(EpiPre<(WM / (64 / (WN / 4)) <= 8 ? WM / (64 / (WN / 4)) : 1)>& pre,
                                                        const ConvGeom& g, const float* __restrict__ addend, long m0,
                                                        int n0, int grp, const BnBwdFuse& fz) {
    constexpr int QPR = WN / 4, RPP = 64 / QPR, NR = WM / RPP, WAVES_N = BN / WN;
    pre.on = false;
    if constexpr (NR <= 8) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wm = wave / WAVES_N, wn = wave % WAVES_N;
        const int qcol = (lane % QPR) * 4, qrow = lane / QPR;
        const int ncol = grp * g.Ng + n0 + wn * WN + qcol;
        pre.on = fz.part1 != nullptr;
        if (pre.on) {
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                const long m = m0 + wm * WM + rr * RPP + qrow;
                const long off = (m < g.M ? m : 0) * g.N + ncol;      // (rows beyond M: any valid address, never used)
                pre.add[rr] = addend ? *(const f32x4*)(addend + off) : f32x4{0.f, 0.f, 0.f, 0.f};     // (fp32 kernels only:
                pre.x1[rr] = *(const f32x4*)(fz.x1 + off);                                            //  fp32 storage)
                pre.x2[rr] = fz.part2 ? *(const f32x4*)(fz.x2 + off) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
}

// Block epilogue.  acc: this wave's (WM x WN) accumulator tiles in MFMA layout; lds: the block's K-loop LDS (free now,
// at least 4*WM*(WN+4) floats); rows m0.., columns grp*Ng + n0.. of the [M][N] output.
// (fp32 storage only: the bf16-mode kernels -- bf16-input igemm, one-plane plane kernels -- use igemm_epilogue_typed below;
// run-time storage flags in THIS function cost the parity path 0.5 % of the step when they were tried)
template <int BM, int BN, int WM, int WN, bool BWD = false, bool BWD_PREFETCH = false>
__device__ __forceinline__ void igemm_epilogue(f32x16 (&acc)[WM / 32][WN / 32], float* lds, const ConvGeom& g,
                                               const float* __restrict__ bias, const float* __restrict__ addend,
                                               float* __restrict__ dst, double* __restrict__ bn_part, int relu,
                                               long m0, int n0, int grp, int mt_id, const BnBwdFuse* fz = nullptr,
                                               const EpiPre<(WM / (64 / (WN / 4)) <= 8 ? WM / (64 / (WN / 4)) : 1)>* pre = nullptr) {
    constexpr int MT = WM / 32, NT = WN / 32, WAVES_N = BN / WN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // ---- epilogue: stage each wave's WMxWN accumulator tile through LDS (the K-loop buffers are free now) and write
    // whole 16-byte row segments: 16 dwordx4 stores (+16 vector addend loads) per lane instead of 64 scalar ones --
    // for the short-K 1x1 layers the scalar epilogue was a quarter of the block's lifetime.
    constexpr int QPR = WN / 4;                               // float4 per tile row
    constexpr int RPP = 64 / QPR;                             // rows covered per pass by the 64 lanes
    constexpr int NR = WM / RPP;                              // rows per lane
    const int qcol = (lane % QPR) * 4, qrow = lane / QPR;
    const int ncol = grp * g.Ng + n0 + wn * WN + qcol;
    constexpr bool PREF = BWD && BWD_PREFETCH && NR <= 8;      // (the fp32 kernels; the plane kernels have no registers to spare)
    const bool pref = PREF && pre != nullptr && pre->on;
    __syncthreads();
    constexpr int LDE = WN + 4;                               // padded row (16B aligned, breaks the 32-bank stride)
    float* Es = lds + wave * (WM * LDE);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) Es[(i * 32 + mfma32_row(e, lane)) * LDE + j * 32 + l31] = acc[i][j][e];
    // each wave reads back its own tile only: no block barrier needed, just this wave's LDS writes
    __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0)
    f32x4 bv4 = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv4 = *(const f32x4*)(bias + ncol);
    // BatchNorm statistics straight from the accumulators when the stored value IS the accumulator (no bias / addend --
    // every convolution in front of a BatchNorm except the split-attention fc1): in the MFMA layout a lane holds 16 rows
    // of ONE column per 32x32 block, so the column partials are lane-local: 16 fp64 adds + 16 fp64 fmas per block and ONE
    // cross-lane step (the other lane half), instead of the row-major path's fp64 accumulation per element PLUS three
    // 64-bit shuffle rounds over eight values (~170 VALU per thread -- matrix time on these short-K layers).  fp64 from
    // the first addition on: 16-term fp32 partials were tried and cost accuracy END TO END (a rounding error in a batch
    // mean shifts a whole channel coherently; log-probs moved from 5e-5 to 1.2e-4 off the fp64 reference on the 224x224
    // fixture).  Rows beyond M contribute exact zeros (their A rows read zeros).
    const bool acc_stats = bn_part && !bias && !addend && !(BWD && fz && fz->part1);
    double as[NT], aq[NT];
    if (acc_stats) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            as[j] = 0.0; aq[j] = 0.0;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const double v = (double)acc[i][j][e];
                    as[j] += v;
                    aq[j] = fma(v, v, aq[j]);
                }
            as[j] += __shfl_xor(as[j], 32, 64);               // the other 16 rows of the column (lane half h)
            aq[j] += __shfl_xor(aq[j], 32, 64);
        }
    }
    double cs[4] = {0, 0, 0, 0}, cq[4] = {0, 0, 0, 0};        // per-column sum / sum of squares (BatchNorm statistics)
    double cq2[4] = {0, 0, 0, 0};
    bool bwd = false, bwd2 = false;
    f32x4 mu1 = {0, 0, 0, 0}, rs1 = {0, 0, 0, 0}, mu2 = {0, 0, 0, 0}, rs2 = {0, 0, 0, 0};
    if constexpr (BWD) {
        bwd = fz != nullptr && fz->part1 != nullptr;
        if (bwd) {
            bn_part = fz->part1;
            bwd2 = fz->part2 != nullptr;
            mu1 = *(const f32x4*)(fz->sv1 + ncol);
            rs1 = *(const f32x4*)(fz->sv1 + g.N + ncol);
            if (bwd2) { mu2 = *(const f32x4*)(fz->sv2 + ncol); rs2 = *(const f32x4*)(fz->sv2 + g.N + ncol); }
        }
    }
#pragma unroll
    for (int rr = 0; rr < WM / RPP; ++rr) {
        const int row = rr * RPP + qrow;
        const long m = m0 + wm * WM + row;
        if (m >= g.M) continue;
        f32x4 v = *(const f32x4*)(Es + row * LDE + qcol) + bv4;
        if constexpr (PREF) {
            if (pref) v += pre->add[rr];
            else if (addend) v += *(const f32x4*)(addend + m * g.N + ncol);
        } else {
            if (addend) v += *(const f32x4*)(addend + m * g.N + ncol);
        }
        if constexpr (BWD) {
            if (bwd) {
                const long off = m * g.N + ncol;
                if (fz->mask) relu_mask_apply(v, fz->mask, off >> 2);
                f32x4 xa;
                if constexpr (PREF) xa = pre->x1[rr]; else xa = *(const f32x4*)(fz->x1 + off);
#pragma unroll
                for (int e = 0; e < 4; ++e) { cs[e] += v[e]; cq[e] += (double)v[e] * ((xa[e] - mu1[e]) * rs1[e]); }
                if (bwd2) {
                    f32x4 xb;
                    if constexpr (PREF) xb = pre->x2[rr]; else xb = *(const f32x4*)(fz->x2 + off);
#pragma unroll
                    for (int e = 0; e < 4; ++e) cq2[e] += (double)v[e] * ((xb[e] - mu2[e]) * rs2[e]);
                }
            }
        }
        if (bn_part && !acc_stats && !bwd) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { cs[e] += v[e]; cq[e] += (double)v[e] * v[e]; }
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *(f32x4*)(dst + m * g.N + ncol) = v;
    }
    if (bn_part) {
        // Fused batch statistics of the tile just written: lanes sharing a column quad (stride QPR) are combined by
        // shuffles, the BM/WM waves sharing a column range through LDS; one fp64 (sum, sumsq) pair per column and
        // M-tile goes to bn_part[mtile][channel][2] -- scouter_bn_fwd_f32 then skips its own read of the tensor.
        if (!acc_stats) {
#pragma unroll
            for (int o = QPR; o < 64; o <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) { cs[e] += __shfl_xor(cs[e], o, 64); cq[e] += __shfl_xor(cq[e], o, 64); }
        }
        __syncthreads();                                      // every wave is done reading its staged tile
        double* Ps = (double*)lds;                            // [4 waves][WN][2], re-uses the staging area
        if (acc_stats) {
            if (lane < 32) {
#pragma unroll
                for (int j = 0; j < NT; ++j) { Ps[(wave * WN + j * 32 + l31) * 2] = as[j]; Ps[(wave * WN + j * 32 + l31) * 2 + 1] = aq[j]; }
            }
        } else if (qrow == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { Ps[(wave * WN + qcol + e) * 2] = cs[e]; Ps[(wave * WN + qcol + e) * 2 + 1] = cq[e]; }
        }
        __syncthreads();
        if (tid < BN) {
            const int wnn = tid / WN, c = tid % WN;
            double a0 = 0, a1 = 0;
#pragma unroll
            for (int w = 0; w < BM / WM; ++w) {
                const int wv = w * WAVES_N + wnn;
                a0 += Ps[(wv * WN + c) * 2];
                a1 += Ps[(wv * WN + c) * 2 + 1];
            }
            double* o = bn_part + ((long)mt_id * g.N + grp * g.Ng + n0 + tid) * 2;
            o[0] = a0;
            o[1] = a1;
        }
        if constexpr (BWD) {
            if (bwd2) {           // the second BatchNorm's (sum g, sum g * xhat2): same reduction once more
#pragma unroll
                for (int o = QPR; o < 64; o <<= 1)
#pragma unroll
                    for (int e = 0; e < 4; ++e) cq2[e] += __shfl_xor(cq2[e], o, 64);
                __syncthreads();
                if (qrow == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { Ps[(wave * WN + qcol + e) * 2] = cs[e]; Ps[(wave * WN + qcol + e) * 2 + 1] = cq2[e]; }
                }
                __syncthreads();
                if (tid < BN) {
                    const int wnn = tid / WN, c = tid % WN;
                    double a0 = 0, a1 = 0;
#pragma unroll
                    for (int w = 0; w < BM / WM; ++w) {
                        const int wv = w * WAVES_N + wnn;
                        a0 += Ps[(wv * WN + c) * 2];
                        a1 += Ps[(wv * WN + c) * 2 + 1];
                    }
                    double* o = fz->part2 + ((long)mt_id * g.N + grp * g.Ng + n0 + tid) * 2;
                    o[0] = a0;
                    o[1] = a1;
                }
            }
        }
    }
}

// The block epilogue of the TYPED kernels (bf16-input igemm, one-plane plane kernels: the bf16 mode), EIGHT columns per lane:
// a bf16-stored output / addend / BatchNorm input is then one 16-byte access per lane and row (the four-column epilogue
// above moves 8 bytes per access there: the bf16-output forward launches ran at 2.0-2.4 TB/s against 4.5 of the fp32-output
// ones).  Storage types are wave-uniform run-time flags: dst_bf16 or fz->io bit 3 (output), fz->io bits 0 / 1 / 2 (x1, x2,
// addend).  Same arithmetic, same statistics layout (bn_part[mtile][channel][2]) as igemm_epilogue.
template <int BM, int BN, int WM, int WN, bool BWD>
__device__ __forceinline__ void igemm_epilogue_typed(f32x16 (&acc)[WM / 32][WN / 32], float* lds, const ConvGeom& g,
                                                     const float* __restrict__ bias, const float* __restrict__ addend,
                                                     float* __restrict__ dst, double* __restrict__ bn_part, int relu,
                                                     long m0, int n0, int grp, int mt_id, const BnBwdFuse* fz,
                                                     bool dst_bf16) {
    constexpr int MT = WM / 32, NT = WN / 32, WAVES_N = BN / WN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    constexpr int QPR = WN / 8;                               // 8-column groups per tile row
    constexpr int RPP = 64 / QPR;                             // rows covered per pass by the 64 lanes
    const int qcol = (lane % QPR) * 8, qrow = lane / QPR;
    const int ncol = grp * g.Ng + n0 + wn * WN + qcol;
    __syncthreads();
    constexpr int LDE = WN + 4;
    float* Es = lds + wave * (WM * LDE);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) Es[(i * 32 + mfma32_row(e, lane)) * LDE + j * 32 + l31] = acc[i][j][e];
    __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): each wave reads back its own tile only
    f32x4 bv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (bias) { bv[0] = *(const f32x4*)(bias + ncol); bv[1] = *(const f32x4*)(bias + ncol + 4); }
    const bool acc_stats = bn_part && !bias && !addend && !(BWD && fz && fz->part1);      // (see igemm_epilogue)
    double as[NT], aq[NT];
    if (acc_stats) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            as[j] = 0.0; aq[j] = 0.0;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const double v = (double)acc[i][j][e];
                    as[j] += v;
                    aq[j] = fma(v, v, aq[j]);
                }
            as[j] += __shfl_xor(as[j], 32, 64);
            aq[j] += __shfl_xor(aq[j], 32, 64);
        }
    }
    double cs[8], cq[8], cq2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { cs[e] = 0.0; cq[e] = 0.0; cq2[e] = 0.0; }
    bool bwd = false, bwd2 = false;
    f32x4 mu1[2], rs1[2], mu2[2], rs2[2];
    const int io = fz != nullptr ? fz->io : 0;
    const bool out_bf16 = dst_bf16 || (io & 8) != 0;
    if constexpr (BWD) {
        bwd = fz != nullptr && fz->part1 != nullptr;
        if (bwd) {
            bn_part = fz->part1;
            bwd2 = fz->part2 != nullptr;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                mu1[q] = *(const f32x4*)(fz->sv1 + ncol + 4 * q);
                rs1[q] = *(const f32x4*)(fz->sv1 + g.N + ncol + 4 * q);
                if (bwd2) { mu2[q] = *(const f32x4*)(fz->sv2 + ncol + 4 * q); rs2[q] = *(const f32x4*)(fz->sv2 + g.N + ncol + 4 * q); }
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < WM / RPP; ++rr) {
        const int row = rr * RPP + qrow;
        const long m = m0 + wm * WM + row;
        if (m >= g.M) continue;
        const long off = m * g.N + ncol;
        f32x4 v[2];
        v[0] = *(const f32x4*)(Es + row * LDE + qcol) + bv[0];
        v[1] = *(const f32x4*)(Es + row * LDE + qcol + 4) + bv[1];
        if (addend) {
            f32x4 a0, a1;
            sc_load8_rt(addend, off, (io & 4) != 0, a0, a1);
            v[0] += a0; v[1] += a1;
        }
        if constexpr (BWD) {
            if (bwd) {
                if (fz->mask) { relu_mask_apply(v[0], fz->mask, off >> 2); relu_mask_apply(v[1], fz->mask, (off >> 2) + 1); }
                f32x4 xa[2];
                sc_load8_rt(fz->x1, off, (io & 1) != 0, xa[0], xa[1]);
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        cs[4 * q + e] += v[q][e];
                        cq[4 * q + e] += (double)v[q][e] * ((xa[q][e] - mu1[q][e]) * rs1[q][e]);
                    }
                if (bwd2) {
                    f32x4 xb[2];
                    sc_load8_rt(fz->x2, off, (io & 2) != 0, xb[0], xb[1]);
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) cq2[4 * q + e] += (double)v[q][e] * ((xb[q][e] - mu2[q][e]) * rs2[q][e]);
                }
            }
        }
        if (bn_part && !acc_stats && !bwd) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) { cs[4 * q + e] += v[q][e]; cq[4 * q + e] += (double)v[q][e] * v[q][e]; }
        }
        if (relu) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[q][e] = fmaxf(v[q][e], 0.f);
        }
        sc_store8_rt(dst, off, v[0], v[1], out_bf16);
    }
    if (bn_part) {
        if (!acc_stats) {
#pragma unroll
            for (int o = QPR; o < 64; o <<= 1)
#pragma unroll
                for (int e = 0; e < 8; ++e) { cs[e] += __shfl_xor(cs[e], o, 64); cq[e] += __shfl_xor(cq[e], o, 64); }
        }
        __syncthreads();
        double* Ps = (double*)lds;                            // [4 waves][WN][2]
        if (acc_stats) {
            if (lane < 32) {
#pragma unroll
                for (int j = 0; j < NT; ++j) { Ps[(wave * WN + j * 32 + l31) * 2] = as[j]; Ps[(wave * WN + j * 32 + l31) * 2 + 1] = aq[j]; }
            }
        } else if (qrow == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { Ps[(wave * WN + qcol + e) * 2] = cs[e]; Ps[(wave * WN + qcol + e) * 2 + 1] = cq[e]; }
        }
        __syncthreads();
        if (tid < BN) {
            const int wnn = tid / WN, c = tid % WN;
            double a0 = 0, a1 = 0;
#pragma unroll
            for (int w = 0; w < BM / WM; ++w) {
                const int wv = w * WAVES_N + wnn;
                a0 += Ps[(wv * WN + c) * 2];
                a1 += Ps[(wv * WN + c) * 2 + 1];
            }
            double* o = bn_part + ((long)mt_id * g.N + grp * g.Ng + n0 + tid) * 2;
            o[0] = a0;
            o[1] = a1;
        }
        if constexpr (BWD) {
            if (bwd2) {
#pragma unroll
                for (int o = QPR; o < 64; o <<= 1)
#pragma unroll
                    for (int e = 0; e < 8; ++e) cq2[e] += __shfl_xor(cq2[e], o, 64);
                __syncthreads();
                if (qrow == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { Ps[(wave * WN + qcol + e) * 2] = cs[e]; Ps[(wave * WN + qcol + e) * 2 + 1] = cq2[e]; }
                }
                __syncthreads();
                if (tid < BN) {
                    const int wnn = tid / WN, c = tid % WN;
                    double a0 = 0, a1 = 0;
#pragma unroll
                    for (int w = 0; w < BM / WM; ++w) {
                        const int wv = w * WAVES_N + wnn;
                        a0 += Ps[(wv * WN + c) * 2];
                        a1 += Ps[(wv * WN + c) * 2 + 1];
                    }
                    double* o = fz->part2 + ((long)mt_id * g.N + grp * g.Ng + n0 + tid) * 2;
                    o[0] = a0;
                    o[1] = a1;
                }
            }
        }
    }
}
