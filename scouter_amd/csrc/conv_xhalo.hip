// fp32-grade 3x3 / stride 1 / pad 1 convolutions with 32 OUTPUT COLUMNS PER GROUP on the bf16 matrix cores: the three-way
// operand split done in registers ONCE per input element, the split rows RESIDENT in LDS for all nine filter taps, one
// persistent workgroup per CU (round 6).
//
// Call sites: the layers whose GEMM is 32 columns wide per group -- the deep stem's 32 -> 32 convolution forward and
// input gradient, the input gradient of its 32 -> 64 convolution (/root/reference/timm/models/resnet.py:471-489), and the
// input gradient of layer1's radix convolution, 64 -> 128 in two groups (/root/reference/timm/models/layers/split_attn.py:54-60).
// Until round 6 they ran on the exact-fp32 MFMA kernels (conv_igemm.hip: 90 TFLOP/s of a 157 peak, 0.9 ms of a 15.5 ms step):
//   * the plane kernels (conv_planes.hip) want 64-multiples of columns and a producer that writes 6-byte planes;
//   * the register-split GEMM (conv_x3.hip, CONV) re-loads and re-splits the A tile for every filter tap: with one
//     32-column MFMA block per split that is 8 VALU per MFMA -- VALU-bound at 0.64-0.77x the fp32 kernel (round 5).
// Here the split is paid once per element and serves nine taps: in the flat NHWC pixel index the taps of a same-size
// convolution are the row offsets dr*W + dq (conv_planes.hip, phalo_kernel), so the rows a 256-pixel tile needs for ALL taps
// are one contiguous range of 256 + 2W + 2 pixels.  Per 16-channel chunk of K that range is loaded as fp32 (two 16-byte loads
// per lane and 32-row group), split exactly (x = hi + mid + lo, 6.5 VALU per element -- 1.8 per MFMA at W = 112, in the shadow
// of the bf16 MFMAs) and stored as three [rows][16 ch] bf16 images (32-byte rows, the two 16-byte halves swapped on bit 3 of
// the row: 16 consecutive rows at any tap shift touch all 64 banks); nine taps x six products then read their fragments from
// it.  Border taps and the pixels of neighbouring images that the flat range drags in read a 32-byte ZERO row instead (one
// select on the ADDRESS per tap).  The weight planes of a chunk ([9 taps][3 planes][32 n][16 k] = 27 KB) come from the
// model's one split launch by LDS-DMA, double-buffered.
//
// A tile is only 864 MFMAs per 16 channels... per CU, so the kernel is PERSISTENT: one 512-thread workgroup per CU walks its
// tile list as ONE stream of chunks -- while chunk s multiplies, the rows of chunk s + 1 (possibly the next tile's) are
// split into the other image buffer and those of chunk s + 2 are on their way from L2 / HBM into registers; one barrier
// per chunk.  The epilogue goes straight from the accumulators (lane = column: 32 lanes = one 128-byte line per row): no
// LDS, no barrier; BatchNorm statistics (forward) / the BatchNorm-backward sums (input gradient, conv_common.h BnBwdFuse)
// are accumulated lane-locally in fp64 over ALL tiles of the workgroup and leave as ONE partial row per workgroup.
//
// K order: 16-channel chunk outer, tap inner, six products smallest-first, two accumulator sets (conv_planes.hip): the
// error bound of the plane kernels, a third of the exact-fp32 MFMA kernel's; not bit-identical to either (another
// summation order).  Deterministic: fixed tile lists, fixed summation orders, no atomics.
#include "conv_common.h"

#include <stdlib.h>
#include <type_traits>

typedef __bf16 hbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short hu16x8 __attribute__((ext_vector_type(8)));

#define XH_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// Development builds only (tools_dev/xhalo_ablate.sh): XH_ABLATE bits remove one ingredient of the chunk loop each -- 1 the
// MFMAs, 2 the image loads, 4 the weight DMAs, 8 the split + its LDS stores, 16 the fragment reads, 32 the barrier and the DMA
// wait, 64 the epilogue.  Results are WRONG; only the timing means something.
#ifndef XH_ABLATE
#define XH_ABLATE 0
#endif

// (NON-template helpers on purpose -- see conv_planes.hip: the builtin inside a kernel template makes hipcc's host pass drop
//  the kernel's launch stub)
__device__ __forceinline__ void xh_dma16(__amdgpu_buffer_rsrc_t rs, char* lds, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, XH_LDS_PTR(lds), 16, voff, soff, 0, 0);
}
__device__ __forceinline__ f32x16 xh_mfma(hbf16x8 a, hbf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// AG: 32-row groups of the resident image (12: maps up to 62 pixels wide, 16: up to 126).
template <int AG, bool DGRAD>
__global__ __launch_bounds__(512, 1) void xhalo_kernel(const float* __restrict__ a_f32, const unsigned short* __restrict__ w_planes,
                                                       long w_plane_elems, const float* __restrict__ addend,
                                                       float* __restrict__ dst, double* __restrict__ bn_part, ConvGeom g,
                                                       int mtiles, int relu, BnBwdFuse fz) {
    constexpr int BM = 256, NW = 8;
    constexpr int APLANE = (AG + 1) * 1024;                      // one plane of one image buffer + a zero block behind it
    constexpr int ABUF = 3 * APLANE;
    constexpr int WTAP = 3 * 1024;                               // one tap: three planes of [32 n][16 k] bf16
    constexpr int WSTAGE = 9 * WTAP;
    constexpr int W_OFF = 2 * ABUF;
    constexpr int ZERO = AG * 1024;                              // zero block of plane 0 in image buffer 0 (+ pl * APLANE)
    constexpr int WDMA = (27 + NW - 1) / NW;                     // weight DMAs per wave and chunk (27 pieces of 1 KB)
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) char lds_raw[];

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = g.W, hw = g.Ho * g.Wo;
    const int CPT = g.Cg / 16;                                   // chunks per tile
    // ---- tile list: workgroup wg serves group wg % groups; among the Gg workgroups of a group, round k hands tile
    // k * Gg + (w & 7) * (Gg / 8) + (w >> 3) to workgroup w (neighbouring tiles -- which share 2W + 2 rows -- run at the same
    // time; conv_planes_persist.h)
    const int grp = blockIdx.x % g.groups, wq = blockIdx.x / g.groups, Gg = gridDim.x / g.groups;
    auto tile_of = [&](int k) -> int { return k * Gg + ((Gg & 7) == 0 ? (wq & 7) * (Gg >> 3) + (wq >> 3) : wq); };
    int n_mine = 0;
    while (tile_of(n_mine) < mtiles) ++n_mine;
    const int S = n_mine * CPT;                                  // chunks of this workgroup's stream

    // ---- zero blocks
    if (tid < 3 * 64) *(f32x4*)(lds_raw + (tid >> 6) * APLANE + ZERO + (tid & 63) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- per-lane sums of the epilogue, over all tiles of this workgroup (fp64 from the first addition)
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;                         // fwd: sum, sum of squares; dgrad+bn: sum g, sum g xhat1, sum g xhat2
    const int col = grp * 32 + l31;                              // this lane's output column (all groups)
    const bool bwd = DGRAD && fz.part1 != nullptr, bwd2 = bwd && fz.part2 != nullptr;
    float mu1 = 0.f, rs1 = 0.f, mu2 = 0.f, rs2 = 0.f;
    if (bwd) {
        mu1 = fz.sv1[col]; rs1 = fz.sv1[g.N + col];
        if (bwd2) { mu2 = fz.sv2[col]; rs2 = fz.sv2[g.N + col]; }
    }

    if (S > 0) {
    // ---- LOAD side: image rows of the chunk stream, two 32-row groups per wave (wave, wave + 8), lane = (row lane >> 1,
    // 8-channel half lane & 1)
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a_f32, 0, (unsigned)(g.M * g.C * 4), 0x00020000);
    const int rows_needed = BM + 2 * W + 2;
    const bool g1 = wave + NW < AG;                              // (wave-uniform) this wave has a second group
    unsigned a_voff[2];
    int ld_k = 0, ld_c = 0;                                      // (tile round, chunk) of the next chunk to request
    auto setup_load_tile = [&](int k) {
        const long m0 = (long)tile_of(k) * BM;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int j = 32 * (wave + NW * v) + (lane >> 1);
            const long p = m0 - (W + 1) + j;
            const bool ok = k < n_mine && (v == 0 || g1) && j < rows_needed && p >= 0 && p < g.M;
            a_voff[v] = ok ? (unsigned)(p * g.C * 4) + (unsigned)((lane & 1) * 32) : OOB;
        }
    };
    f32x4 ra[2][2];
    auto load_group = [&](int v) {
        const int soff = (grp * g.Cg + ld_c * 16) * 4;
        ra[v][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_voff[v], soff, 0));
        ra[v][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_voff[v] + 16u, soff, 0));
    };
    auto advance_load = [&]() {
        if (++ld_c == CPT) { ld_c = 0; ++ld_k; setup_load_tile(ld_k); }
    };
    const int st_off = (lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4);      // row, swizzled half
    // exact split x = hi + mid + lo of one group's 8 values, two at a time (spread over the filter taps of the MFMA loop: ~13
    // VALU per tap next to six MFMAs), then one 16-byte piece per plane.  Waves without a second group (AG < 16) store theirs
    // into a 3-KB landing zone: no branch inside the scheduled region.
    constexpr bool ALWAYS_G1 = AG >= 2 * NW;
    constexpr int DUMMY = W_OFF + 2 * WSTAGE;
    hu16x8 sp[3];
    auto split_pair = [&](int v, int part) {
#pragma unroll
        for (int e = 2 * part; e < 2 * part + 2; ++e) {
            unsigned short a, b, c;
            split3_bf16(ra[v][e >> 2][e & 3], a, b, c);
            sp[0][e] = a; sp[1][e] = b; sp[2][e] = c;
        }
    };
    auto store_split = [&](int buf, int v) {
        const bool real = v == 0 || ALWAYS_G1 || g1;
        char* d = lds_raw + (real ? buf * ABUF + (wave + NW * v) * 1024 : DUMMY) + st_off;
        const int ps = real ? APLANE : 1024;
        *(hu16x8*)(d) = sp[0];
        *(hu16x8*)(d + ps) = sp[1];
        *(hu16x8*)(d + 2 * ps) = sp[2];
    };
    auto store_group = [&](int buf, int v) {                     // (prologue)
#pragma unroll
        for (int part = 0; part < 4; ++part) split_pair(v, part);
        store_split(buf, v);
    };
    // ---- weights of chunk c -> weight stage: piece e = tap * 3 + plane, wave w fetches e = w, w + 8, ...
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(w_planes + (long)grp * 32 * g.Cg), 0, 0x7fffffff, 0x00020000);
    const unsigned b_voff = (unsigned)(((lane >> 1) * g.Cg) * 2 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4));
    const long wtap_bytes = (long)g.N * g.Cg * 2, w_plane_bytes = w_plane_elems * 2;
    auto dma_w = [&](int c, int stage) {
#pragma unroll
        for (int u = 0; u < WDMA; ++u) {
            const int e = wave + NW * u;
            const bool real = e < 27;
            const int tap = e / 3, pl = e - 3 * tap;
            char* d = lds_raw + W_OFF + stage * WSTAGE + (real ? e : 0) * 1024;
            xh_dma16(rs_b, real ? d : lds_raw + ZERO, real ? b_voff : OOB, (int)(tap * wtap_bytes + pl * w_plane_bytes + c * 32));
        }
    };

    // ---- COMPUTE side
    f32x16 acc, accl, accm;                                      // hi * hi | the five correction products in two chains
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[e] = 0.f; accl[e] = 0.f; accm[e] = 0.f; }
    unsigned amask = 0;                                          // 9 tap bits of this lane's row (wave * 32 + l31) of the tile
    auto setup_masks = [&](int k) {
        const long m = (long)tile_of(k) * BM + wave * 32 + l31;
        const int bimg = (int)(((double)(unsigned)m + 0.5) * g.inv_hw);
        const int rem = (int)((unsigned)m - (unsigned)bimg * (unsigned)hw);
        const int y = fast_div(rem, g.inv_wo), x = rem - y * g.Wo;
        unsigned mask = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int yy = DGRAD ? y + 1 - r : y + r - 1, xx = DGRAD ? x + 1 - q : x + q - 1;
                mask |= ((unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)W ? 1u : 0u) << (r * 3 + q);
            }
        amask = m < g.M ? mask : 0u;
    };
    // image row j holds pixel m0 - (W + 1) + j; tap (r, q) of tile row t reads image row t + (W + 1) + dr * W + dq
    const int arow0 = (wave * 32 + l31 + W + 1) * 32;
    const int bfrag = W_OFF + l31 * 32 + ((h ^ ((l31 >> 3) & 1)) << 4);
    hbf16x8 F[2][2][3];
    auto load_frags = [&](int buf, int cp, int tap) {
        const int r = tap / 3, q = tap - 3 * r;
        const int shift = ((DGRAD ? 1 - r : r - 1) * W + (DGRAD ? 1 - q : q - 1)) * 32;
        const int row32 = arow0 + shift;
        const int live = cp * ABUF + row32 + ((((row32 >> 8) ^ h) & 1) << 4);
        const int addr = ((amask >> tap) & 1u) ? live : ZERO + h * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) F[buf][0][pl] = *(const hbf16x8*)(lds_raw + addr + pl * APLANE);
        const char* bs = lds_raw + bfrag + cp * WSTAGE + tap * WTAP;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) F[buf][1][pl] = *(const hbf16x8*)(bs + pl * 1024);
    };
    auto mma = [&](int buf) {
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};      // smallest terms first (conv_planes.hip)
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {
            // (one 32 x 32 block per wave: the correction products alternate between two accumulators so that no MFMA waits
            //  for the one in front of it)
            if (pr == 5) acc = xh_mfma(F[buf][0][PA[pr]], F[buf][1][PB[pr]], acc);
            else if (pr & 1) accm = xh_mfma(F[buf][0][PA[pr]], F[buf][1][PB[pr]], accm);
            else accl = xh_mfma(F[buf][0][PA[pr]], F[buf][1][PB[pr]], accl);
        }
    };
    // ---- epilogue of tile round k, straight from the accumulators: lane = column, rows mfma32_row(e, lane).
    // The operands of the BatchNorm-backward epilogue (BatchNorm input, ReLU sign words) are requested when the tile's LAST chunk
    // starts and arrive under its MFMAs: read inside the store loop (first version) every row paid its own L2 / HBM round trip
    // behind the previous row's store -- 12 us per tile, four times the tile's matrix work.
    // Addressing: buffer descriptors over the whole tensors, ONE per-lane offset for the lane's first row of a tile (row
    // wave * 32 + 4 h, its column), everything else -- tile, row of the accumulator register -- in the scalar offset; rows beyond
    // M fall outside the descriptor (loads return zeros, stores are dropped): no per-row branch, no 64-bit address arithmetic.
    const unsigned out_bytes = (unsigned)(g.M * g.N * 4);
    const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc((void*)(bwd ? fz.x1 : a_f32), 0, bwd ? out_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc((void*)(bwd2 ? fz.x2 : a_f32), 0, bwd2 ? out_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc((void*)(addend ? addend : a_f32), 0, addend ? out_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_msk = __builtin_amdgcn_make_buffer_rsrc((void*)(bwd && fz.mask ? (const void*)fz.mask : (const void*)a_f32), 0,
                                                                             bwd && fz.mask ? (unsigned)((g.M * g.N + 255) / 256 * 32) : 0u, 0x00020000);
    const unsigned lane_elem = (unsigned)((wave * 32 + 4 * h) * g.N + col);       // element index of the lane's first row, tile-relative
    const unsigned lane_off = lane_elem * 4u;
    auto row_soff = [&](int k, int e) -> int {                  // scalar byte offset of accumulator register e's row in tile round k
        return (int)(((unsigned)tile_of(k) * (unsigned)BM + (unsigned)((e & 3) + 8 * (e >> 2))) * (unsigned)g.N * 4u);
    };
    float px1[16];
    unsigned long long pmw[4];
    // (rows 4h .. 4h + 3 of an 8-row block share ONE 64-bit mask word when a row has at most 64 channels -- every layer this
    //  kernel serves; wider rows take the in-loop path)
    const bool mask_shared = g.N <= 64;
    auto mask_word = [&](unsigned elem) -> unsigned long long {  // the 64-bit word holding the ReLU bit of flat element `elem`
        const unsigned idx = ((elem >> 8) << 2) + (elem & 3u);
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 w = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_msk, idx * 8u, 0, 0));
        return (unsigned long long)w[0] | ((unsigned long long)w[1] << 32);
    };
    auto epi_prefetch = [&](int k) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
            px1[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x1, lane_off, row_soff(k, e), 0));
        if (fz.mask && mask_shared) {
            const unsigned base = (unsigned)tile_of(k) * (unsigned)BM * (unsigned)g.N + lane_elem;
#pragma unroll
            for (int q = 0; q < 4; ++q) pmw[q] = mask_word(base + (unsigned)(8 * q) * (unsigned)g.N);
        }
    };
    auto epilogue = [&](int k) {
        const unsigned base = (unsigned)tile_of(k) * (unsigned)BM * (unsigned)g.N + lane_elem;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int soff = row_soff(k, e);
            float v = acc[e] + (accl[e] + accm[e]);
            acc[e] = 0.f; accl[e] = 0.f; accm[e] = 0.f;
            if (addend) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_add, lane_off, soff, 0));
            if constexpr (DGRAD) {
                if (bwd) {
                    if (fz.mask) {
                        const unsigned elem = base + (unsigned)((e & 3) + 8 * (e >> 2)) * (unsigned)g.N;
                        const unsigned long long word = mask_shared ? pmw[e >> 2] : mask_word(elem);
                        v = ((word >> ((elem >> 2) & 63u)) & 1ull) ? v : 0.f;
                    }
                    // (rows beyond M: the accumulator is an exact zero -- their A rows read zeros -- and so are their terms)
                    const float xh1 = (px1[e] - mu1) * rs1;
                    s0 += v;
                    s1 += (double)v * xh1;
                    if (bwd2) {
                        const float x2v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x2, lane_off, soff, 0));
                        s2 += (double)v * ((x2v - mu2) * rs2);
                    }
                }
            } else {
                if (bn_part) { const double vd = (double)v; s0 += vd; s1 = fma(vd, vd, s1); }
                if (relu) v = fmaxf(v, 0.f);
            }
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_dst, lane_off, soff, 0);
        }
    };

#define SBAR() __builtin_amdgcn_sched_barrier(0)
#define SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
    // ---- prologue: chunk 0 split into image buffer 0, chunk 1 in registers, weights of chunk 0 in stage 0
    setup_load_tile(0);
    load_group(0); load_group(1);
    advance_load();
    dma_w(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_group(0, 0);
    store_group(0, 1);
    load_group(0); load_group(1);                                // chunk 1 (zeros beyond the stream's end)
    advance_load();
    setup_masks(0);
    __builtin_amdgcn_s_waitcnt(0xc07f);                          // lgkmcnt(0): this wave's LDS stores (and the zero blocks)
    __builtin_amdgcn_s_barrier();

    int kc = 0, kt = 0;                                          // compute side: chunk within the tile, tile round
    for (int s = 0; s < S; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        const int kc_next = kc + 1 == CPT ? 0 : kc + 1;
        SBAR();
        if (!(XH_ABLATE & 4)) dma_w(kc_next, nxt);               // weights of chunk s + 1 (fenced: in front of the reloads below)
        if constexpr (DGRAD) {
            if (bwd && kc + 1 == CPT) epi_prefetch(kt);          // (older than the image reloads: covered by the wait below)
        }
        SBAR();
        load_frags(0, cur, 0);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // Nine taps, six MFMAs each.  Under the MFMAs of tap t: the fragment reads of tap t + 1 (one per MFMA slot) and a quarter of
        // the split of chunk s + 1's rows (requested a whole chunk ago) -- group 0 under taps 0-3, group 1 under taps 4-7; a
        // finished group goes to the other image buffer and its registers are re-loaded with chunk s + 2 right behind.
        // Program order inside a tap = the order LDS accesses must keep (the compiler cannot tell the buffers apart):
        // fragment READS first, then the stores.
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            SBAR();
            if (tap + 1 < 9 && !(XH_ABLATE & 16)) load_frags((tap + 1) & 1, cur, tap + 1);
            if (tap < 8 && !(XH_ABLATE & 8)) split_pair(tap >> 2, tap & 3);
            if (tap == 3) { if (!(XH_ABLATE & 8)) store_split(nxt, 0); if (!(XH_ABLATE & 2)) load_group(0); }
            if (tap == 7) { if (!(XH_ABLATE & 8)) store_split(nxt, 1); if (!(XH_ABLATE & 2)) load_group(1); }
            if (!(XH_ABLATE & 1)) mma(tap & 1);
#pragma unroll
            for (int q = 0; q < 6; ++q) {                        // (the six reads go out under the first three MFMAs: landed by the sixth)
                SG(0x008, 1);
                if (tap + 1 < 9 && q < 3) SG(0x100, 2);
                SG(0x006, 4);
            }
            if (tap == 3 || tap == 7) { SG(0x200, 3); SG(0x020, 2); }
            SBAR();
            __builtin_amdgcn_s_waitcnt(0xc07f);                  // fragments of tap t + 1 (and this wave's stores)
        }
        advance_load();
        SBAR();
        if (!(XH_ABLATE & 32)) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // the weight DMAs have landed (four image loads fly on)
            __builtin_amdgcn_s_barrier();                        // chunk s + 1 complete in LDS; buffers of chunk s free
        }
        SBAR();
        if (++kc == CPT) {
            if (!(XH_ABLATE & 64)) epilogue(kt);
            kc = 0; ++kt;
            if (kt < n_mine) setup_masks(kt);
        }
    }
#undef SBAR
#undef SG
    }
    // ---- partial row of this workgroup: the two lane halves of a column, then the eight waves, in a fixed order
    if (bn_part || bwd) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        s0 += __shfl_xor(s0, 32, 64);
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        double* Ps = (double*)lds_raw;                           // [8 waves][32 columns][3]
        if (lane < 32) { Ps[(wave * 32 + l31) * 3] = s0; Ps[(wave * 32 + l31) * 3 + 1] = s1; Ps[(wave * 32 + l31) * 3 + 2] = s2; }
        __syncthreads();
        if (tid < 32) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { a0 += Ps[(w * 32 + tid) * 3]; a1 += Ps[(w * 32 + tid) * 3 + 1]; a2 += Ps[(w * 32 + tid) * 3 + 2]; }
            double* o = (bwd ? fz.part1 : bn_part) + ((long)wq * g.N + grp * 32 + tid) * 2;
            o[0] = a0; o[1] = a1;
            if (bwd2) { double* o2 = fz.part2 + ((long)wq * g.N + grp * 32 + tid) * 2; o2[0] = a0; o2[1] = a2; }
        }
    }
}

// workgroups per launch: one per CU, a multiple of 8 * groups
static int xh_grid(int groups) {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
            n = 256;
        cus = n;
    }
    const int unit = 8 * groups;
    return cus >= unit ? cus / unit * unit : groups;
}

// geometry the kernel covers (g as for the GEMM: Cg = K per group, Ng = columns per group)
bool sc_xhalo_ok(const ConvGeom& g) {
    return g.R == 3 && g.S == 3 && g.stride == 1 && g.pad == 1 && g.H == g.Ho && g.W == g.Wo && g.W <= 126 && g.Ng == 32 &&
           g.Cg % 16 == 0 && g.Cg >= 16 && g.M * g.C * 4 < (1L << 31) && g.M * g.N * 4 < (1L << 31) &&
           (long)9 * g.N * g.Cg * 2 * 3 < (1L << 31);
}
int sc_xhalo_partial_rows(int groups) { return xh_grid(groups) / groups; }

template <int AG, bool DGRAD>
static void launch_xhalo_ag(const float* a, const void* w, long w_pe, const float* addend, float* dst, double* bn_part,
                            const ConvGeom& g, int relu, hipStream_t st, const BnBwdFuse& fz) {
    ConvGeom gg = g;
    gg.inv_hw = 1.0 / ((double)g.Ho * g.Wo);
    gg.inv_wo = 1.0f / (float)g.Wo;
    gg.inv_ho = 1.0f / (float)g.Ho;
    constexpr int LDS = 2 * 3 * (AG + 1) * 1024 + 2 * 27 * 1024 + (AG < 16 ? 3 * 1024 : 0);
    auto kern = xhalo_kernel<AG, DGRAD>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(xh_grid(g.groups)), dim3(512), LDS, st, a, (const unsigned short*)w, w_pe, addend, dst, bn_part,
                       gg, sc_cdiv(g.M, 256), relu, fz);
}
void sc_launch_xhalo(bool dgrad, const float* a, const void* w, long w_pe, const float* addend, float* dst, double* bn_part,
                     const ConvGeom& g, int relu, hipStream_t st, const BnBwdFuse& fz) {
    const bool small = 256 + 2 * g.W + 2 <= 12 * 32;
    if (dgrad) {
        if (small) launch_xhalo_ag<12, true>(a, w, w_pe, addend, dst, bn_part, g, relu, st, fz);
        else launch_xhalo_ag<16, true>(a, w, w_pe, addend, dst, bn_part, g, relu, st, fz);
    } else {
        if (small) launch_xhalo_ag<12, false>(a, w, w_pe, addend, dst, bn_part, g, relu, st, fz);
        else launch_xhalo_ag<16, false>(a, w, w_pe, addend, dst, bn_part, g, relu, st, fz);
    }
}
