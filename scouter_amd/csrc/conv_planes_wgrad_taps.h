// Tap-fused weight gradient on bf16 operand planes for same-size 3x3 convolutions.
//
//   dW[tap][ci][co] = sum_m X[m + tapoff(tap)][ci] * dY[m][co]                    (pixels m in flat NHWC order)
//
// pwgrad_kernel gives every filter tap its own workgroups: the X rows of a pixel range are fetched nine times (shifted)
// and the dY rows nine times, 48 KB of L2 -> LDS traffic per 1 584 matrix cycles of a 128 x 128 tile = 30 B / cycle / CU --
// PMC: 445-492 MB of HBM-side traffic per launch, 2.3 x the algorithmic bytes, 42-44 % MFMA-busy.  Here ONE workgroup
// accumulates all nine taps of a 64 x 64 (ci x co) tile:
//   * X rows live in an LDS RING of 256 pixel rows per plane.  In the flat pixel index the taps are the row offsets
//     (r - 1) W + (q - 1), so chunk c (32 pixels) needs rows m0 - W - 1 .. m0 + 32 + W: with W <= 63 that is inside five
//     32-row units, consecutive chunks share all but one of them -- every X row is fetched ONCE per workgroup (one 32-row
//     unit per chunk, issued seven chunks ahead) instead of nine times;
//   * dY rows are fetched once per chunk (three stages) and their fragments serve all nine taps;
//   * per 32-pixel chunk and wave: 108 MFMAs (9 taps x 2 k-steps x 6 bf16 products) against 6 DMA instructions and 60
//     transposing fragment reads -- 24 KB of L2 -> LDS traffic per 3 456 matrix cycles = 7 B / cycle / CU.
// Border / neighbouring-image taps are not masked in the data: a lane whose (pixel, tap) is invalid points its
// ds_read_b64_tr_b16 at a ZERO ROW (each lane of a transposing read addresses one pixel row), one v_bfi on the address.
// Accumulators: 9 taps x {hi*hi, corrections} x one 32 x 32 tile per wave = 288 registers (one wave per SIMD).
// Same slab / split-K output format as pwgrad_kernel (deterministic; the slabs are summed by slab_reduce_kernel).
#pragma once

template <int NP, int NACC_>
__global__ __launch_bounds__(256, 1) void pwgrad_taps_kernel(const unsigned short* __restrict__ x_planes, long x_pe,
                                                             const unsigned short* __restrict__ dy_planes, long dy_pe,
                                                             float* __restrict__ out, ConvGeom g, int ci_tiles,
                                                             int co_tiles, long pix_per_split, long slab,
                                                             float* __restrict__ dw, unsigned* __restrict__ arrival) {
    constexpr int BM = 64, BN = 64, BK = 32, NACC = NP == 3 ? NACC_ : 1;
    constexpr int RING = 256;                                   // X pixel rows resident per plane (8 units of 32)
    constexpr int XPL = RING * BM * 2 + 128;                    // plane image + its zero row (row index 256)
    constexpr int X_BYTES = NP * XPL;
    constexpr int DYPL = BK * BN * 2, DY_STAGE = NP * DYPL, NDY = 3;
    extern __shared__ __attribute__((aligned(1024))) char lds_raw[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_raw;

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int bid, split_id;
    wgrad_block_coords(bid, split_id);
    const int tile_id = bid;
    const int co_t = bid % co_tiles; bid /= co_tiles;
    const int ci_t = bid % ci_tiles;
    const int grp = bid / ci_tiles;
    const int ci0 = ci_t * BM, co0 = co_t * BN;
    const long mbeg = (long)split_id * pix_per_split;            // multiple of 32
    long mend = mbeg + pix_per_split;
    if (mend > g.M) mend = g.M;
    const int KT = (int)((mend - mbeg + BK - 1) / BK);
    const int W = g.W, H = g.H;

    // ---- zero rows (one per plane, right behind its ring)
    if (tid < NP * 8) *(f32x4*)(lds_raw + (tid >> 3) * XPL + RING * BM * 2 + (tid & 7) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- DMA geometry: one wave instruction = 8 pixel rows x 128 bytes; lane -> (row l >> 3, 16-byte slot l & 7), the
    // channel chunk it fetches carries the bank swizzle of its row (slot s of row r holds chunk s ^ 4 ((r >> 1) & 1))
    constexpr unsigned OOB = 0x80000000u;
    const int drow = lane >> 3, dchunk = (lane & 7) ^ (4 * ((drow >> 1) & 1));
    const unsigned x_lane = (unsigned)(((long)(8 * wave + drow) * g.C + grp * g.Cg + ci0 + 8 * dchunk) * 2);
    const unsigned dy_lane = (unsigned)(((long)(8 * wave + drow) * g.N + grp * g.Ng + co0 + 8 * dchunk) * 2);
    __amdgpu_buffer_rsrc_t rs_x[NP], rs_dy[NP];
    {
        const long xr = g.M * (long)g.C * 2, dr = mend * (long)g.N * 2;   // dY rows beyond this split's range read zeros
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            rs_x[pl] = __builtin_amdgcn_make_buffer_rsrc((void*)(x_planes + pl * x_pe), 0, (unsigned)xr, 0x00020000);
            rs_dy[pl] = __builtin_amdgcn_make_buffer_rsrc((void*)(dy_planes + pl * dy_pe), 0, (unsigned)dr, 0x00020000);
        }
    }
    // X unit u = pixel rows mbeg - 64 + 32 u .. + 31 (entirely before pixel 0 or not at all: mbeg is a multiple of 32)
    auto issue_x = [&](int u) {
        const long m0 = mbeg - 64 + 32L * u;
        const unsigned voff = m0 < 0 ? OOB : x_lane;             // (those rows are only ever addressed by invalid taps)
        const int soff = m0 < 0 ? 0 : (int)(m0 * g.C * 2);
        char* dst = lds_raw + ((32 * u - 64 + 8 * wave) & (RING - 1)) * (BM * 2);   // ring position = (pixel - mbeg) & 255
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) dma16(rs_x[pl], dst + pl * XPL, voff, soff);
    };
    auto issue_dy = [&](int c) {
        const long m0 = mbeg + 32L * c;
        char* dst = lds_raw + X_BYTES + (c % NDY) * DY_STAGE + wave * 1024;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) dma16(rs_dy[pl], dst + pl * DYPL, dy_lane, (int)(m0 * g.N * 2));
    };

    // ---- fragment geometry of a transposing read (see pwgrad_kernel): lane -> pixel row r2 of a 4-row group, channel
    // columns 16 g1 + 4 (l & 3) .. + 3 of the wave's 32-channel block; the 16-pixel k-step s and the 4-row sub-block u are
    // added per read
    const int r2 = (lane >> 2) & 3, g1 = (lane >> 4) & 1, c4 = 4 * (lane & 3);
    const int xcol = wm * 32 + 16 * g1 + c4, ycol = wn * 32 + 16 * g1 + c4;
    // dY: fixed lane offset inside a stage (rows 16 s + 4 u further down keep the swizzle term)
    const unsigned fb_off = (unsigned)((8 * h + r2) * (BN * 2) + (((ycol >> 3) ^ (4 * (((8 * h + r2) >> 1) & 1))) << 4) + (ycol & 7) * 2);
    // X: ring row = (32 c + 16 s + 8 h + 4 u + r2 + tapoff) & 255; its swizzle bit is bit 1 of (r2 + tapoff): fixed per tap
    unsigned xl[9];
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        toff[t] = (t / 3 - 1) * W + (t % 3 - 1);
        const int sw = 4 * (((r2 + toff[t]) >> 1) & 1);
        xl[t] = lds_base + (unsigned)((((xcol >> 3) ^ sw) << 4) + (xcol & 7) * 2);
    }
    const unsigned zrow = lds_base + RING * BM * 2 + (unsigned)(((xcol >> 3) << 4) + (xcol & 7) * 2);

    // ---- (y, x) of the lane's four pixels of a chunk: p(s, u) = m0 + 16 s + 8 h + 4 u + r2
    const int hw = H * W;
    const int blk_b = (int)(((double)(unsigned long)mbeg + 0.5) * g.inv_hw);
    const int blk_rem = (int)(mbeg - (long)blk_b * hw);
    const int blk_y = fast_div(blk_rem, g.inv_wo), blk_x = blk_rem - blk_y * W;
    int py[4], px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int off = 16 * (k >> 1) + 8 * h + 4 * (k & 1) + r2;
        const int tx = blk_x + off, qx = fast_div(tx, g.inv_wo);
        const int ty = blk_y + qx, qy = fast_div(ty, g.inv_ho);
        px[k] = tx - qx * W;
        py[k] = ty - qy * H;
    }
    const int adv_x = BK % W, adv_y = BK / W;
    int prow = 8 * h + r2;                                       // pixel row of p(0, 0) relative to mbeg (+ 32 per chunk)

    f32x16 acc[9], accl[NACC == 2 ? 9 : 1];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            acc[t][e] = 0.f;
            if (NACC == 2) accl[t][e] = 0.f;
        }

#define SBAR() __builtin_amdgcn_sched_barrier(0)
#define ACL(t) (NACC == 2 ? accl[NACC == 2 ? (t) : 0] : acc[t])
#define LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#ifdef PWT_STAMPS     // dev builds (tools_dev/pwt_stamps.py): cycles of wave 0 in {DMA wait + barrier + DMA issue, MFMA groups}
    long long st_wait = 0, st_pro = 0, st_grp = 0, st_t0 = __builtin_readcyclecounter(), st_a, st_b, st_c, st_d;
    const unsigned long long st_r0 = wall_clock64();
#define STAMP(v) v = __builtin_readcyclecounter()
#else
#define STAMP(v)
#endif
    // validity of the nine taps for the lane's pixel k of the chunk the coordinates stand at (bit 3 r + q), then advance them
    auto mask_build = [&](int k) -> unsigned {
        const unsigned cb = (px[k] >= 1 ? 1u : 0u) | 2u | (px[k] <= W - 2 ? 4u : 0u);
        return (py[k] >= 1 ? cb : 0u) | (cb << 3) | (py[k] <= H - 2 ? (cb << 6) : 0u);
    };
    auto mask_advance = [&](int k) {                             // coordinates of pixel k, one chunk (32 pixels) on
        asm volatile("" : "+v"(px[k]), "+v"(py[k]));             // (pinned to its slot, like the address math below)
        px[k] += adv_x;
        const int wrap = px[k] >= W ? 1 : 0;
        px[k] -= wrap ? W : 0;
        py[k] += adv_y + wrap;
        py[k] -= py[k] >= H ? H : 0;
    };
    auto tap_mask = [&](int k) -> unsigned { const unsigned m = mask_build(k); mask_advance(k); return m; };
    unsigned vm[4], vmn[4];                                       // this chunk's masks / the next chunk's (made in the slots)
    u64x2 Fa[3][NP];                                             // [buffer][plane] = two transposing reads (u = 0, 1)
    // ---- address of the lane's pixel row for (chunk-relative row base pr0, k-step s, sub-block u, tap t), or the zero row.
    // Three stages, each pinned to ITS slot between two MFMAs by an empty asm that makes its input opaque there (otherwise the
    // compiler hoists the arithmetic in front of the chunk's MFMAs, where it leaves the matrix pipe idle); both reads' chains
    // side by side, at most ~6 short VALU per slot: a slot with one 6-deep dependent chain delayed the next MFMA by ~12 cycles.
    auto addr_rows = [&](int pr0, int s, int t, unsigned& r0, unsigned& r1) {
        int pr = pr0;
        asm volatile("" : "+v"(pr));
        r0 = (unsigned)(pr + 16 * s + toff[t]) & (RING - 1);
        r1 = (unsigned)(pr + 16 * s + 4 + toff[t]) & (RING - 1);
    };
    auto addr_mix = [&](unsigned m0, unsigned m1, int t, unsigned& r0, unsigned& r1, unsigned& k0, unsigned& k1) {
        asm volatile("" : "+v"(r0), "+v"(r1));
        r0 = (r0 << 7) + xl[t];
        r1 = (r1 << 7) + xl[t];
        k0 = (unsigned)(((int)(m0 << (31 - t))) >> 31);          // bit t -> 0 or ~0
        k1 = (unsigned)(((int)(m1 << (31 - t))) >> 31);
    };
    auto addr_sel = [&](unsigned& r0, unsigned& r1, unsigned k0, unsigned k1) {
        asm volatile("" : "+v"(r0), "+v"(r1));
        r0 = (r0 & k0) | (zrow & ~k0);
        r1 = (r1 & k1) | (zrow & ~k1);
    };
    auto addr = [&](int pr0, unsigned mask, int s, int u, int t) -> unsigned {      // (all at once: pipeline start-up only)
        const unsigned row = (unsigned)(pr0 + 16 * s + 4 * u + toff[t]) & (RING - 1);
        const unsigned msk = (unsigned)(((int)(mask << (31 - t))) >> 31);
        const unsigned ad = (row << 7) + xl[t];
        return (ad & msk) | (zrow & ~msk);
    };
#ifdef PWT_NO_READS
    bool pwt_started = false;
#endif
    unsigned a0, a1, a0c, a1c;                                   // addresses of the reads the NEXT group issues (planes 0 / 1; plane 2)
    // read q of a group's six (NP = 3) / two fragment reads: q = 2 * plane + u
    auto read_a = [&](int buf, auto Q) {
        constexpr int q = decltype(Q)::value, pl = q >> 1, u = q & 1;
        const unsigned ad = pl < 2 ? (u ? a1 : a0) : (u ? a1c : a0c);          // (2 * XPL exceeds the 16-bit offset field)
#ifdef PWT_NO_READS       // dev ablation (results wrong, timing only): the fragment reads of the steady state are dropped
        if (pwt_started) return;
#endif
        Fa[buf][pl][u] = pl == 1 ? ds_read_tr16<XPL>(ad) : ds_read_tr16<0>(ad);
    };
    auto read_a_plane = [&](int buf, auto PL) {                  // (pipeline start-up)
        constexpr int pl = decltype(PL)::value;
        a0c = a0 + 2 * XPL; a1c = a1 + 2 * XPL;
        read_a(buf, std::integral_constant<int, 2 * pl>{});
        read_a(buf, std::integral_constant<int, 2 * pl + 1>{});
    };
    auto FA = [&](int buf, int pl) -> bf16x8 { return __builtin_bit_cast(bf16x8, Fa[buf][pl]); };
    u64x2 FbR[2][NP];
    auto read_b1 = [&](int buf, unsigned dyb, auto S, auto Q) {   // one of the 2 * NP reads of a k-step's dY fragments
        constexpr int s = decltype(S)::value, q = decltype(Q)::value, pl = q >> 1, u = q & 1;
        FbR[buf][pl][u] = ds_read_tr16<pl * DYPL + (16 * s + 4 * u) * (BN * 2)>(dyb);
    };
    auto read_b = [&](int buf, unsigned dyb, auto S) {            // (pipeline start-up)
        read_b1(buf, dyb, S, std::integral_constant<int, 0>{}); read_b1(buf, dyb, S, std::integral_constant<int, 1>{});
        if constexpr (NP == 3) {
            read_b1(buf, dyb, S, std::integral_constant<int, 2>{}); read_b1(buf, dyb, S, std::integral_constant<int, 3>{});
            read_b1(buf, dyb, S, std::integral_constant<int, 4>{}); read_b1(buf, dyb, S, std::integral_constant<int, 5>{});
        }
    };
    auto FB = [&](int buf, int pl) -> bf16x8 { return __builtin_bit_cast(bf16x8, FbR[buf][pl]); };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    auto dy_base = [&](int c) -> unsigned { return lds_base + X_BYTES + (c % NDY) * DY_STAGE + fb_off; };

    LGKM0();                                                     // (the zero rows, before the first barrier)
    if (KT > 0) {
        // Issue order matters for the counted waits.  Steady state: after the barrier inside chunk c the wave issues dY(c + 2),
        // then X(c + 7); the barrier inside chunk c needs everything up to dY(c + 1) landed = all but the last issue.
        issue_x(0); issue_x(1); issue_x(2); issue_x(3); issue_x(4); issue_dy(0);
        issue_x(5); issue_dy(1); issue_x(6);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NP) : "memory");
        __builtin_amdgcn_s_barrier();
        // pipeline start-up (once per workgroup): masks of chunk 0, the fragments of its first TWO groups, addresses of its third
#pragma unroll
        for (int k = 0; k < 4; ++k) vm[k] = tap_mask(k);
        read_b(0, dy_base(0), I0{});
        a0 = addr(prow, vm[0], 0, 0, 0); a1 = addr(prow, vm[1], 0, 1, 0);
        read_a_plane(0, I0{});
        if constexpr (NP == 3) { read_a_plane(0, I1{}); read_a_plane(0, I2{}); }
        a0 = addr(prow, vm[0], 0, 0, 1); a1 = addr(prow, vm[1], 0, 1, 1);
        read_a_plane(1, I0{});
        if constexpr (NP == 3) { read_a_plane(1, I1{}); read_a_plane(1, I2{}); }
        a0 = addr(prow, vm[0], 0, 0, 2); a1 = addr(prow, vm[1], 0, 1, 2);
        a0c = a0 + 2 * XPL; a1c = a1 + 2 * XPL;
        LGKM0();
    }
    // ---- ONE software pipeline over all chunks x 18 groups (k-step s, tap t), interleaved by hand (the transposing reads are
    // inline asm, which sched_group_barrier cannot classify): between the six MFMAs of group n the wave issues the fragment reads
    // of group n + 2 (three fragment buffers; addresses from group n - 1's slots) and computes the addresses of group n + 3 -- one
    // wave per SIMD, so whatever is not issued BETWEEN MFMAs leaves the matrix pipe idle (first version, reads + address math in
    // front of each group: 389 us; up to ~5 VALU per bf16 MFMA execute in its shadow, tools_dev/mfma_bf16_shadow_bench.hip), and
    // a group may only wait for reads issued a whole group earlier (counted lgkmcnt: with one group of distance the wave sat
    // ~50 cycles per group in s_waitcnt, the LDS pipe being shared by four waves).  The pipeline runs THROUGH the chunk
    // boundary: the barrier that publishes chunk c + 1's DMA data sits after group 15 of chunk c, the slots of groups 16 / 17
    // already fetch chunk c + 1's first fragments, and the next chunk's tap masks are made in the slots of groups 6-12 (a
    // per-chunk restart cost 1 000 of 4 800 cycles, tools_dev/pwt_stamps.py).
#ifdef PWT_NO_READS
    pwt_started = true;
#endif
    for (int c = 0; c < KT; ++c) {
        STAMP(st_a);
        const unsigned dyb = dy_base(c), dybn = dy_base(c + 1);
        const int prown = prow + BK;
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            const int s = st / 9, t = st % 9, cur = st % 3, fil = (st + 2) % 3;
            // group n + 3 and its chunk-relative coordinates (groups 15-17 compute for the NEXT chunk's groups 0-2)
            const int s3 = ((st + 3) % 18) / 9, t3 = (st + 3) % 9;
            const bool wrap3 = st + 3 >= 18;
            unsigned n0 = 0, n1 = 0, k0 = 0, k1 = 0;
            const int prn = wrap3 ? prown : prow;
            const unsigned mk0 = wrap3 ? vmn[2 * s3] : vm[2 * s3], mk1 = wrap3 ? vmn[2 * s3 + 1] : vm[2 * s3 + 1];
            const bool mk = st >= 6 && st < 14 && !(st & 1);     // groups 6, 8, 10, 12: the next chunk's mask of pixel 0..3
            const int mkk = (st - 6) >> 1;
            const bool bread = st == 7 || st == 16;              // B fragments of k-step 1 / of the next chunk's k-step 0
            using I3 = std::integral_constant<int, 3>;
            using I4 = std::integral_constant<int, 4>;
            using I5 = std::integral_constant<int, 5>;
            // one B read per slot in the two groups that fetch dY fragments (k-step 1 of this chunk / k-step 0 of the next)
            auto bq = [&](auto Q) {
                if (st == 7) read_b1(1, dyb, I1{}, Q);
                if (st == 16) read_b1(0, dybn, I0{}, Q);
            };
            if constexpr (NP == 3) {
                // every slot: ONE fragment read + <= 3 short VALU (a slot with two reads + a 6-deep VALU chain delayed the next
                // MFMA: 252 instead of 198 cycles per group)
                SBAR();
                ACL(t) = mfma_bf16p(FA(cur, 0), FB(s, 2), ACL(t));
                SBAR();
                read_a(fil, I0{}); bq(I0{});
                { int pr = prn; asm volatile("" : "+v"(pr)); n0 = (unsigned)(pr + 16 * s3 + toff[t3]) & (RING - 1); }
                SBAR();
                ACL(t) = mfma_bf16p(FA(cur, 2), FB(s, 0), ACL(t));
                SBAR();
                read_a(fil, I1{}); bq(I1{});
                { int pr = prn; asm volatile("" : "+v"(pr)); n1 = (unsigned)(pr + 16 * s3 + 4 + toff[t3]) & (RING - 1); }
                SBAR();
                ACL(t) = mfma_bf16p(FA(cur, 1), FB(s, 1), ACL(t));
                SBAR();
                read_a(fil, I2{}); bq(I2{});
                asm volatile("" : "+v"(n0));
                n0 = (n0 << 7) + xl[t3];
                k0 = (unsigned)(((int)(mk0 << (31 - t3))) >> 31);
                SBAR();
                ACL(t) = mfma_bf16p(FA(cur, 0), FB(s, 1), ACL(t));
                SBAR();
                read_a(fil, I3{}); bq(I3{});
                asm volatile("" : "+v"(n1));
                n1 = (n1 << 7) + xl[t3];
                k1 = (unsigned)(((int)(mk1 << (31 - t3))) >> 31);
                SBAR();
                ACL(t) = mfma_bf16p(FA(cur, 1), FB(s, 0), ACL(t));
                SBAR();
                read_a(fil, I4{}); bq(I4{});
                asm volatile("" : "+v"(n0), "+v"(n1));
                n0 = (n0 & k0) | (zrow & ~k0);
                n1 = (n1 & k1) | (zrow & ~k1);
                SBAR();
                acc[t] = mfma_bf16p(FA(cur, 0), FB(s, 0), acc[t]);
                SBAR();
                read_a(fil, I5{}); bq(I5{});
                if (mk) vmn[mkk] = mask_build(mkk);
                if (st >= 7 && st < 15 && (st & 1)) mask_advance((st - 7) >> 1);
                SBAR();
            } else {
                SBAR();
                read_a(fil, I0{}); read_a(fil, I1{});
                bq(I0{}); bq(I1{});
                addr_rows(prn, s3, t3, n0, n1);
                addr_mix(mk0, mk1, t3, n0, n1, k0, k1);
                addr_sel(n0, n1, k0, k1);
                if (mk) vmn[mkk] = tap_mask(mkk);
                SBAR();
                acc[t] = mfma_bf16p(FA(cur, 0), FB(s, 0), acc[t]);
                SBAR();
            }
            a0 = n0; a1 = n1; a0c = n0 + 2 * XPL; a1c = n1 + 2 * XPL;
            // the NEXT group's fragments were requested a group ago: everything but this group's own reads must have landed
            // (LDS returns in order)
            if (bread) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(4 * NP) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * NP) : "memory");
            if (st == 15) {
                // ---- chunk c + 1's operands become visible; every wave is past chunk c - 1, whose buffers the new DMAs reuse
                STAMP(st_c);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
                __builtin_amdgcn_s_barrier();
                SBAR();
                issue_dy(c + 2);                                 // stage of chunk c - 1
                issue_x(c + 7);                                  // ring slot of unit c - 1, last read by chunk c - 1
#ifdef PWT_STAMPS
                STAMP(st_d); st_wait += st_d - st_c; st_grp -= st_d - st_c;
#endif
            }
        }
        prow = prown;
#pragma unroll
        for (int k = 0; k < 4; ++k) vm[k] = vmn[k];
#ifdef PWT_STAMPS
        STAMP(st_b); st_grp += st_b - st_a;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#undef SBAR
#undef ACL
#undef LGKM0

#ifdef PWT_STAMPS
    if (tid == 0) {
        long long* sp = (long long*)(out + (long)gridDim.y * slab) + 6 * ((long)blockIdx.y * gridDim.x + blockIdx.x);
        sp[0] = st_wait; sp[1] = st_pro; sp[2] = st_grp; sp[3] = __builtin_readcyclecounter() - st_t0; sp[4] = (long long)(wall_clock64() - st_r0); sp[5] = (long long)st_r0;
    }
#endif
    float* o = out + (long)split_id * slab;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ci = ci0 + wm * 32 + mfma32_row(e, lane);
            const int co = grp * g.Ng + co0 + wn * 32 + l31;
            slab_store(o + (long)t * g.Cg * g.N + (long)ci * g.N + co,
                       NACC == 2 ? acc[t][e] + accl[NACC == 2 ? t : 0][e] : acc[t][e], arrival != nullptr);
        }
    if (arrival)        // the last workgroup of this tile sums the slabs itself (conv_common.h slab_tile_finish)
        slab_tile_finish(out, dw, slab, gridDim.y, arrival + tile_id, 9, (long)g.Cg * g.N, (long)ci0 * g.N, BM, g.N,
                         grp * g.Ng + co0, BN);
}
