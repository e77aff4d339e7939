// Input pipeline on the GPU: the reference's `make_transform` (dataset/transform_func.py:101-124) =
//   Resize((S,S)) [PIL bilinear with antialiasing, via torchvision F.resize, :19-31] -> ToTensor [:51-66] -> Normalize
//   [:91-99], applied to a whole batch of decoded uint8 images of arbitrary sizes in two launches.
// The resize reproduces Pillow's 8-bit resampler (libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
// ImagingResample{Horizontal,Vertical}_8bpc) BIT FOR BIT: per output coordinate a normalised triangle filter of
// support max(in/out, 1) evaluated in fp64 (no FMA contraction: Pillow is built for baseline x86-64), converted to
// 22-bit fixed point; horizontal pass into a uint8 intermediate, vertical pass, out = clip8((2^21 + sum px*k) >> 22).
// ToTensor + Normalize collapse to a [C][256] table lookup (built by the host in float64 exactly as the reference
// computes it).  HBM-bound byte work: lanes run along the interleaved (x, channel) axis, coefficients of a block's
// output columns / rows are computed once into LDS.
#include "common.h"

#define IMG_PRECISION_BITS 22
#define IMG_TILE 64            // output columns (pass H) / output rows (pass V) whose coefficients a block keeps in LDS
#define IMG_MAX_TAPS 96        // 2*ceil(in/out)+1 <= 96: down-scaling by up to 47x

#pragma clang fp contract(off)
// coefficients of output coordinate xx for in_size -> out_size; returns the tap count, *xmin_out = first input index
__device__ static int img_coeffs(int in_size, int out_size, int xx, int* __restrict__ kk, int* xmin_out) {
    const double scale = (double)(float)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = 0.0 + (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double t = (x + xmin - center + 0.5) * ss;
        if (t < 0.0) t = -t;
        ww += t < 1.0 ? 1.0 - t : 0.0;
    }
    for (int x = 0; x < xmax; ++x) {
        double t = (x + xmin - center + 0.5) * ss;
        if (t < 0.0) t = -t;
        double w = t < 1.0 ? 1.0 - t : 0.0;
        if (ww != 0.0) w /= ww;
        kk[x] = w < 0 ? (int)(-0.5 + w * (double)(1 << IMG_PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << IMG_PRECISION_BITS));
    }
    *xmin_out = xmin;
    return xmax;
}
#pragma clang fp contract(fast)

__device__ __forceinline__ unsigned char img_clip8(int acc) {
    const int v = acc >> IMG_PRECISION_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: src [h][w][C] -> tmp [h][out_w][C];  grid (ceil(out_w / IMG_TILE), row chunks, B)
__global__ __launch_bounds__(256) void img_resize_h_kernel(const unsigned char* const* __restrict__ srcs,
                                                           const int* __restrict__ hw, unsigned char* __restrict__ tmp,
                                                           const long* __restrict__ tmp_off, int C, int out_w,
                                                           int rows_per_block) {
    __shared__ int kk[IMG_TILE][IMG_MAX_TAPS];
    __shared__ int xmin_s[IMG_TILE], cnt_s[IMG_TILE];
    const int b = blockIdx.z, h = hw[2 * b], w = hw[2 * b + 1];
    const int x0 = blockIdx.x * IMG_TILE, ncol = min(IMG_TILE, out_w - x0);
    const int y0 = blockIdx.y * rows_per_block;
    if (y0 >= h) return;
    if (threadIdx.x < ncol) cnt_s[threadIdx.x] = img_coeffs(w, out_w, x0 + threadIdx.x, kk[threadIdx.x], &xmin_s[threadIdx.x]);
    __syncthreads();
    const unsigned char* __restrict__ src = srcs[b];
    unsigned char* __restrict__ dst = tmp + tmp_off[b];
    const int y1 = min(h, y0 + rows_per_block), per_row = ncol * C;
    for (int it = threadIdx.x; it < (y1 - y0) * per_row; it += 256) {
        const int yy = y0 + it / per_row, e = it % per_row, xl = e / C, c = e - xl * C;
        const unsigned char* p = src + ((long)yy * w + xmin_s[xl]) * C + c;
        int acc = 1 << (IMG_PRECISION_BITS - 1);
        const int n = cnt_s[xl];
        for (int x = 0; x < n; ++x) acc += (int)p[x * C] * kk[xl][x];
        dst[((long)yy * out_w + x0 + xl) * C + c] = img_clip8(acc);
    }
}

// vertical pass + ToTensor/Normalize lookup: tmp [h][out_w][C] -> out [B][C][out_h][out_w] float
// grid (ceil(out_h / IMG_TILE), column chunks, B)
__global__ __launch_bounds__(256) void img_resize_v_kernel(const unsigned char* __restrict__ tmp,
                                                           const long* __restrict__ tmp_off, const int* __restrict__ hw,
                                                           const float* __restrict__ lut, float* __restrict__ out, int C,
                                                           int out_h, int out_w, int cols_per_block) {
    __shared__ int kk[IMG_TILE][IMG_MAX_TAPS];
    __shared__ int ymin_s[IMG_TILE], cnt_s[IMG_TILE];
    const int b = blockIdx.z, h = hw[2 * b];
    const int y0 = blockIdx.x * IMG_TILE, nrow = min(IMG_TILE, out_h - y0);
    const int c0 = blockIdx.y * cols_per_block, c1 = min(out_w, c0 + cols_per_block);
    if (threadIdx.x < nrow) cnt_s[threadIdx.x] = img_coeffs(h, out_h, y0 + threadIdx.x, kk[threadIdx.x], &ymin_s[threadIdx.x]);
    __syncthreads();
    const unsigned char* __restrict__ src = tmp + tmp_off[b];
    const int per_row = (c1 - c0) * C;
    for (int it = threadIdx.x; it < nrow * per_row; it += 256) {
        const int yl = it / per_row, e = it % per_row, xl = e / C, c = e - xl * C, xx = c0 + xl;
        const unsigned char* p = src + ((long)ymin_s[yl] * out_w + xx) * C + c;
        int acc = 1 << (IMG_PRECISION_BITS - 1);
        const int n = cnt_s[yl];
        for (int y = 0; y < n; ++y) acc += (int)p[(long)y * out_w * C] * kk[yl][y];
        out[(((long)b * C + c) * out_h + y0 + yl) * out_w + xx] = lut[c * 256 + img_clip8(acc)];
    }
}

extern "C" int scouter_resize_normalize_u8_f32(const unsigned char* const* srcs, const int* hw, unsigned char* tmp,
                                               const long* tmp_off, const float* lut, float* out, int B, int C, int max_h,
                                               int max_w, int out_h, int out_w, void* stream) {
    SC_REQUIRE(srcs && hw && tmp && tmp_off && lut && out, "resize_normalize: null pointer");
    SC_REQUIRE(B > 0 && (C == 1 || C == 3 || C == 4) && out_h > 0 && out_w > 0 && max_h > 0 && max_w > 0,
               "resize_normalize: bad shape B=%d C=%d out=%dx%d", B, C, out_h, out_w);
    const int taps_w = 2 * (int)((max_w + out_w - 1) / out_w) + 1, taps_h = 2 * (int)((max_h + out_h - 1) / out_h) + 1;
    SC_UNSUPPORTED(taps_w <= IMG_MAX_TAPS && taps_h <= IMG_MAX_TAPS,
                   "resize_normalize: down-scaling factor above %d is not supported (image %dx%d -> %dx%d)",
                   (IMG_MAX_TAPS - 1) / 2, max_h, max_w, out_h, out_w);
    hipStream_t st = (hipStream_t)stream;
    ScProfScope prof("resize_normalize", st, 0,
                     (double)B * ((double)max_h * max_w * C + 2.0 * max_h * out_w * C + 4.0 * out_h * out_w * C));
    const int rpb = 32;
    hipLaunchKernelGGL(img_resize_h_kernel, dim3(sc_cdiv(out_w, IMG_TILE), sc_cdiv(max_h, rpb), B), dim3(256), 0, st, srcs,
                       hw, tmp, tmp_off, C, out_w, rpb);
    const int cpb = 64;
    hipLaunchKernelGGL(img_resize_v_kernel, dim3(sc_cdiv(out_h, IMG_TILE), sc_cdiv(out_w, cpb), B), dim3(256), 0, st, tmp,
                       tmp_off, hw, lut, out, C, out_h, out_w, cpb);
    return sc_check_launch("resize_normalize");
}
