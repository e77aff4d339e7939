// PERSISTENT plane convolution (forward / input gradient), included by conv_planes.hip behind pconv_kernel.
//
// pconv_kernel<256,128,3,2,*,4> keeps the matrix pipe ~77 % busy INSIDE its K loop, but every 512-thread workgroup owns a
// CU alone, so its prologue (row addressing + the first DMA round trip, 11-19 k cycles) and its LDS-staged epilogue (7 k
// cycles, behind a vmcnt(0) + two barriers) are exposed: 26 % of a K = 576 tile, and a launch of T tiles costs
// ceil(T / 256) whole tile times.  Here ONE workgroup per CU walks a list of tiles:
//   * the LDS-DMA pipeline never drains: while tile t computes its last K-tiles the DMAs already fetch the first K-tiles
//     of tile t + 1 (its row state is set up by the issuing side two K-tiles ahead of the computing side);
//   * the epilogue uses NO LDS and NO barrier -- both LDS stages hold the next tile's operands by then.  The 32x32
//     accumulator blocks (lane = column, 16 rows per lane) are transposed 4x4 inside lane quads with DPP moves, after
//     which a lane holds 4 consecutive columns of one row: 16-byte stores / addend loads, the row-major BatchNorm-
//     backward sums; BatchNorm statistics come lane-locally from the accumulators as before.  Partial sums are written
//     PER WAVE ROW (64 rows) instead of per 256-row tile, so there is no cross-wave reduction either: the finalize
//     kernels simply see 4 x as many partial rows.
//   * tiles are handed out so that the 32 workgroups of an XCD work on neighbouring tiles at the same time (shared
//     operand rows stay in that XCD's L2): round k, workgroup w -> tile k * G + (w & 7) * (G / 8) + (w >> 3).
// Same K order and accumulation order as pconv_kernel: outputs are bit-identical to tiles 0-4.
#pragma once
#ifndef PP_ABLATE
#define PP_ABLATE 0     // dev builds only (tools_dev/pp_ablate.sh): 1 = no DMA, 2 = no DMA wait / barrier, 4 = no LDS reads, 8 = no epilogue
#endif

__device__ __forceinline__ float dpp_quad_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_quad_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
}
// 4x4 transpose across the four lanes of a quad: lane j ends up with what were element j of lanes 0..3
__device__ __forceinline__ f32x4 quad_transpose4(float a0, float a1, float a2, float a3, int lane) {
    const bool odd = lane & 1, hi = lane & 2;
    const float r0 = dpp_quad_xor1(odd ? a0 : a1), r1 = dpp_quad_xor1(odd ? a2 : a3);
    if (odd) { a0 = r0; a2 = r1; } else { a1 = r0; a3 = r1; }
    const float u0 = dpp_quad_xor2(hi ? a0 : a2), u1 = dpp_quad_xor2(hi ? a1 : a3);
    if (hi) { a0 = u0; a1 = u1; } else { a2 = u0; a3 = u1; }
    return f32x4{a0, a1, a2, a3};
}

template <int BN, int NP, bool DGRAD>
__global__ __launch_bounds__(512, 1) void ppersist_kernel(const unsigned short* __restrict__ a_planes, long a_plane_elems,
                                                          const unsigned short* __restrict__ w_planes, long w_plane_elems,
                                                          const float* __restrict__ bias, const float* __restrict__ addend,
                                                          float* __restrict__ dst, double* __restrict__ bn_part, ConvGeom g,
                                                          int relu, int mtiles, int ntiles, BnBwdFuse fz) {
    constexpr int BM = 256, BK = 32, NWM = 4, NW = 8, WM = 64, WN = BN / 2, MT = 2, NT = WN / 32, NSTAGE = 2;
    constexpr int A_BYTES = NP * BM * 64, B_BYTES = NP * BN * 64, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int ARG = BM / 16 / NW;                          // 16-row groups of A per wave (2)
    constexpr int BGROUPS = BN / 16;                           // 16-row groups of B (8 or 4): wave w < BGROUPS fetches one
    constexpr int DMA_PER_TILE = (ARG + 1) * NP;               // (waves without a B group issue an out-of-range dummy)
    constexpr int DUMMY = NSTAGE * STAGE_BYTES;                // 1 KB landing zone of the dummy DMAs
    constexpr int NPROD = NP == 3 ? 6 : 1;
    static_assert(BN == 128 || BN == 64, "tile");
    extern __shared__ __attribute__((aligned(1024))) char lds_raw[];

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntot = mtiles * ntiles * g.groups;
    const int G = gridDim.x, wg = blockIdx.x;
    // tile list of this workgroup: one round-robin round per k (see the header comment); G <= 256, a multiple of 8 when
    // there is more than one round
    const bool single = ntot <= G;
    auto tile_of = [&](int k) -> int { return single ? xcd_remap(wg, ntot) : k * G + (wg & 7) * (G >> 3) + (wg >> 3); };
    int n_mine = 0;
    while (tile_of(n_mine) < ntot && (n_mine == 0 || !single)) ++n_mine;
    if (n_mine == 0) return;
    const int cpt = g.Cg / BK;
    const int ntaps = g.R * g.S;
    const int KT = ntaps * cpt;

    // ---- issue side: DMA source addressing of the tile the DMA stream is in
    constexpr unsigned OOB = 0x80000000u;
    const int slot = lane & 3, rin = lane >> 2;
    const int chunk = slot ^ ((lane >> 4) & 3);
    const int hw = g.Ho * g.Wo;
    const long img_elems = (long)g.H * g.W * g.C;
    const long shift = DGRAD ? ((long)(g.R - 1) * g.W + (g.S - 1)) * g.C : ((long)g.pad * g.W + g.pad) * g.C;
    const long wtap_bytes = (long)g.N * g.Cg * 2;
    const long a_plane_bytes = a_plane_elems * 2, w_plane_bytes = w_plane_elems * 2;
    unsigned a_mask[ARG], a_voff[ARG], b_voff;
    __amdgpu_buffer_rsrc_t rs_a, rs_b;
    auto setup_issue_tile = [&](int L) {
        const int grp = L % g.groups, nt_id = (L / g.groups) % ntiles, mt_id = L / (g.groups * ntiles);
        const long m0 = (long)mt_id * BM;
        const int n0 = nt_id * BN;
        const int blk_b = (int)(((double)(unsigned)m0 + 0.5) * g.inv_hw);
        const int blk_rem = (int)((unsigned)m0 - (unsigned)blk_b * (unsigned)hw);
        const int blk_y = fast_div(blk_rem, g.inv_wo), blk_x = blk_rem - blk_y * g.Wo;
        const int rows_valid = (int)(g.M - m0 < BM ? g.M - m0 : BM);
#pragma unroll
        for (int t = 0; t < ARG; ++t) {
            const int rowoff = 16 * (wave + NW * t) + rin;
            const bool okm = rowoff < rows_valid;
            const int tx = blk_x + rowoff, qx = fast_div(tx, g.inv_wo), x = tx - qx * g.Wo;
            const int ty = blk_y + qx, qy = fast_div(ty, g.inv_ho), y = ty - qy * g.Ho;
            const int ay = DGRAD ? y + g.pad : y * g.stride - g.pad, ax = DGRAD ? x + g.pad : x * g.stride - g.pad;
            unsigned colbits = 0, mask = 0;
            for (int q = 0; q < g.S; ++q) colbits |= ((unsigned)(DGRAD ? ax - q : ax + q) < (unsigned)g.W ? 1u : 0u) << q;
            for (int r = 0; r < g.R; ++r) mask |= ((unsigned)(DGRAD ? ay - r : ay + r) < (unsigned)g.H ? colbits : 0u) << (r * g.S);
            a_mask[t] = okm ? mask : 0u;
            const int rel = okm ? qy * g.H * g.W : 0;
            const int e = DGRAD ? (rel + ay * g.W + ax) * g.C : (rel + (ay + g.pad) * g.W + (ax + g.pad)) * g.C;
            a_voff[t] = (unsigned)(e + grp * g.Cg + chunk * 8) * 2u;
        }
        rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(a_planes + (long)blk_b * img_elems - shift), 0, 0x7fffffff, 0x00020000);
        rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(w_planes + (long)grp * g.Ng * g.Cg), 0, 0x7fffffff, 0x00020000);
        b_voff = wave < BGROUPS ? (unsigned)((n0 + 16 * wave + rin) * g.Cg + chunk * 8) * 2u : OOB;
    };
    constexpr bool TAP_INNER = DGRAD;                          // K order as in pconv_kernel
    auto issue = [&](int kt, int stage) {                      // DMA of K-tile kt of the ISSUE tile -> LDS `stage`
        const int chunk_i = TAP_INNER ? kt / ntaps : kt % cpt, tap = TAP_INNER ? kt - chunk_i * ntaps : kt / cpt;
        const int c0 = chunk_i * BK;
        const int r = tap / g.S, q = tap - r * g.S;
        const long toff = (DGRAD ? -((long)r * g.W + q) : ((long)r * g.W + q)) * g.C + c0;
        unsigned a_veff[ARG];
#pragma unroll
        for (int t = 0; t < ARG; ++t) a_veff[t] = ((a_mask[t] >> tap) & 1u) ? a_voff[t] : OOB;
        const long sa = (DGRAD ? shift + toff : toff) * 2;
        const long sb = tap * wtap_bytes + (long)c0 * 2;
        char* st = lds_raw + stage * STAGE_BYTES;
        char* bdst = wave < BGROUPS ? st + A_BYTES + wave * 1024 : lds_raw + DUMMY;
        const int bstep = wave < BGROUPS ? BN * 64 : 0;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int t = 0; t < ARG; ++t)
                dma16(rs_a, st + pl * (BM * 64) + (wave + NW * t) * 1024, a_veff[t], (int)(sa + pl * a_plane_bytes));
            dma16(rs_b, bdst + pl * bstep, b_voff, (int)(sb + pl * w_plane_bytes));
        }
    };

    constexpr int NACC = NP == 3 ? 2 : 1;
    f32x16 acc[MT][NT], accl[NACC == 2 ? MT : 1][NACC == 2 ? NT : 1];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    acc[i][j][e] = 0.f;
                    if (NACC == 2) accl[i][j][e] = 0.f;
                }
    };
    zero_acc();

    const int sw = (l31 >> 2) & 3;
    auto frag = [&](const char* base, int row0, int step) -> bf16x8 {
        const int c = (2 * step + h) ^ sw;
        return *(const bf16x8*)(base + (row0 + l31) * 64 + c * 16);
    };
    bf16x8 F[2][MT + NT][NP];
    auto load_frags = [&](int buf, int stage, int step) {
        const char* As = lds_raw + stage * STAGE_BYTES;
        const char* Bs = As + A_BYTES;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int i = 0; i < MT; ++i) F[buf][i][pl] = frag(As + pl * (BM * 64), wm * WM + i * 32, step);
#pragma unroll
            for (int j = 0; j < NT; ++j) F[buf][MT + j][pl] = frag(Bs + pl * (BN * 64), wn * WN + j * 32, step);
        }
    };
    auto mma = [&](int buf) {
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};      // smallest terms first
#pragma unroll
        for (int pr = (NP == 3 ? 0 : 5); pr < 6; ++pr)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (NACC == 2 && pr < 5)
                        accl[i][j] = mfma_bf16p(F[buf][i][PA[pr]], F[buf][MT + j][PB[pr]], accl[i][j]);
                    else
                        acc[i][j] = mfma_bf16p(F[buf][i][NP == 3 ? PA[pr] : 0], F[buf][MT + j][NP == 3 ? PB[pr] : 0], acc[i][j]);
                }
    };

    // ---- epilogue of the tile the COMPUTING side just finished : registers -> global, no LDS
    auto epilogue = [&](int L) {
        const int grp = L % g.groups, nt_id = (L / g.groups) % ntiles, mt_id = L / (g.groups * ntiles);
        const long m0 = (long)mt_id * BM;
        const int n0 = nt_id * BN;
        if (NACC == 2) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] += accl[i][j];
        }
        const int ncol0 = grp * g.Ng + n0 + wn * WN;
        const long prow = (long)mt_id * NWM + wm;              // partial row of this wave row (64 rows of the tile)
        bool bwd = false, bwd2 = false;
        if constexpr (DGRAD) { bwd = fz.part1 != nullptr; bwd2 = bwd && fz.part2 != nullptr; }
        // BatchNorm statistics of the stored values, lane-locally from the accumulators (lane = column): see igemm_epilogue
        if (bn_part && !bwd) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                double as = 0.0, aq = 0.0;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const double v = (double)acc[i][j][e];
                        as += v;
                        aq = fma(v, v, aq);
                    }
                as += __shfl_xor(as, 32, 64);
                aq += __shfl_xor(aq, 32, 64);
                if (lane < 32) {
                    double* o = bn_part + (prow * g.N + ncol0 + j * 32 + l31) * 2;
                    o[0] = as;
                    o[1] = aq;
                }
            }
        }
        const int jq = l31 & 3, cq = l31 >> 2;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int ncol = ncol0 + j * 32 + cq * 4;
            f32x4 bv4 = {0.f, 0.f, 0.f, 0.f};
            if (bias) bv4 = *(const f32x4*)(bias + ncol);
            double cs[4] = {0, 0, 0, 0}, cx[4] = {0, 0, 0, 0}, cx2[4] = {0, 0, 0, 0};
            f32x4 mu1 = {0, 0, 0, 0}, rs1 = {0, 0, 0, 0}, mu2 = {0, 0, 0, 0}, rs2 = {0, 0, 0, 0};
            if constexpr (DGRAD) {
                if (bwd) {
                    mu1 = *(const f32x4*)(fz.sv1 + ncol);
                    rs1 = *(const f32x4*)(fz.sv1 + g.N + ncol);
                    if (bwd2) { mu2 = *(const f32x4*)(fz.sv2 + ncol); rs2 = *(const f32x4*)(fz.sv2 + g.N + ncol); }
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = quad_transpose4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2],
                                              acc[i][j][4 * q + 3], lane);
                    const long m = m0 + wm * WM + i * 32 + 8 * q + 4 * h + jq;
                    if (m >= g.M) continue;
                    const long off = m * g.N + ncol;
                    v += bv4;
                    if (addend) v += *(const f32x4*)(addend + off);
                    if constexpr (DGRAD) {
                        if (bwd) {
                            if (fz.mask) relu_mask_apply(v, fz.mask, off >> 2);
                            const f32x4 xa = *(const f32x4*)(fz.x1 + off);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { cs[e] += v[e]; cx[e] += (double)v[e] * ((xa[e] - mu1[e]) * rs1[e]); }
                            if (bwd2) {
                                const f32x4 xb = *(const f32x4*)(fz.x2 + off);
#pragma unroll
                                for (int e = 0; e < 4; ++e) cx2[e] += (double)v[e] * ((xb[e] - mu2[e]) * rs2[e]);
                            }
                        }
                    }
                    if (relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    *(f32x4*)(dst + off) = v;
                }
            if constexpr (DGRAD) {
                if (bwd) {        // the 8 lanes (jq, h) that share a column quad; fixed order -> deterministic
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        cs[e] += __shfl_xor(cs[e], 1, 64); cx[e] += __shfl_xor(cx[e], 1, 64);
                        cs[e] += __shfl_xor(cs[e], 2, 64); cx[e] += __shfl_xor(cx[e], 2, 64);
                        cs[e] += __shfl_xor(cs[e], 32, 64); cx[e] += __shfl_xor(cx[e], 32, 64);
                        if (bwd2) {
                            cx2[e] += __shfl_xor(cx2[e], 1, 64);
                            cx2[e] += __shfl_xor(cx2[e], 2, 64);
                            cx2[e] += __shfl_xor(cx2[e], 32, 64);
                        }
                    }
                    if (jq == 0 && h == 0) {
                        double* o = fz.part1 + (prow * g.N + ncol) * 2;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { o[2 * e] = cs[e]; o[2 * e + 1] = cx[e]; }
                        if (bwd2) {
                            double* o2 = fz.part2 + (prow * g.N + ncol) * 2;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { o2[2 * e] = cs[e]; o2[2 * e + 1] = cx2[e]; }
                        }
                    }
                }
            }
        }
    };

#define SBAR() __builtin_amdgcn_sched_barrier(0)
#define SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
    constexpr int NMMA = MT * NT * NPROD;                 // MFMAs per half K-tile
    constexpr int NFR = (MT + NT) * NP;                   // fragment reads per half K-tile
    // One K-tile of the computing side; the DMA round issued under it is K-tile `ikt` of the ISSUE tile.  The body is
    // branch-free (the sched_group pipeline does not survive control flow: a first version with the tile switch inside
    // issue() had all nine DMAs and 40 scalar instructions clustered in front of the MFMAs).
    int stage = 0;
    auto ktile = [&](int ikt) {
        const int nstage = stage ^ 1;
        SBAR();
        if (!(PP_ABLATE & 4)) load_frags(1, stage, 1);            // A
        mma(0);                                                   // B
#pragma unroll
        for (int q = 0; q < NMMA; ++q) { SG(0x008, 1); if (q < NFR) { SG(0x100, 1); SG(0x006, 2); } }
        SBAR();
        __builtin_amdgcn_s_waitcnt(0xc07f);                       // C: lgkmcnt(0)
        if (!(PP_ABLATE & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      //    this wave's part of the next round has landed
            __builtin_amdgcn_s_barrier();                         // D
        }
        SBAR();
        if (!(PP_ABLATE & 1)) issue(ikt, stage);                  // E
        if (!(PP_ABLATE & 4)) load_frags(0, nstage, 0);
        mma(1);                                                   // F
#pragma unroll
        for (int q = 0; q < NMMA; ++q) {
            SG(0x008, 1);
            if (q < DMA_PER_TILE) { SG(0x020, 1); SG(0x006, 4); }
            else if (q - DMA_PER_TILE < NFR) { SG(0x100, 1); SG(0x006, 2); }
        }
        SBAR();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        stage = nstage;
    };
    setup_issue_tile(tile_of(0));
    issue(0, 0);
    issue(1, 1);                                                  // (KT >= 2: checked by the host)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_TILE) : "memory");     // round 0 has landed
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int k = 0; k < n_mine; ++k) {
        for (int kt = 0; kt < KT - 2; ++kt) ktile(kt + 2);
        // the DMA stream crosses into the next tile of the list (past its end: the last tile again, harmlessly)
        setup_issue_tile(tile_of(k + 1 < n_mine ? k + 1 : k));
        ktile(0);
        ktile(1);
        if (!(PP_ABLATE & 8)) epilogue(tile_of(k));               // results leave straight from the registers
        zero_acc();
    }
#undef SBAR
#undef SG
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the re-fetched rounds past the end of the list)
}
