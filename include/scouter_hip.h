/* scouter_hip.h -- C ABI of libscouter_hip.so, the MI355X (gfx950 / CDNA4) implementation of SCOUTER's xSlot
 * training hot path.
 *
 * The reference (wbw520/scouter) is pure Python over PyTorch; it has NO native plugin / FFI interface to copy
 * (SURVEY.md section 8b).  Each entry point below therefore replaces a PyTorch op *call site* of the reference
 * (cited as file:line under /root/reference) and is what a binding for that call site would bind to.
 *
 * Conventions
 *   - plain C types only: device pointers, sizes, a `void* stream` (hipStream_t); no torch types;
 *   - every function is asynchronous on `stream`, re-entrant, keeps no global mutable state, never allocates
 *     device memory (callers pass workspaces sized by the *_workspace_bytes helpers), never throws or exits;
 *   - return 0 on success, <0 on error (SC_ERR_* below); the message is in scouter_last_error() (thread-local);
 *   - tensors are caller-owned, dense, 16-byte aligned fp32 unless stated;
 *   - activations are NHWC ("[M = B*H*W][C]"), convolution weights are HWIO ([kh][kw][Cin/groups][Cout]).
 */
#ifndef SCOUTER_HIP_H
#define SCOUTER_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SC_OK 0
#define SC_ERR_ARG (-1)
#define SC_ERR_LAUNCH (-2)
#define SC_ERR_UNSUPPORTED (-3)
#define SC_ERR_WORKSPACE (-4)

int scouter_abi_version(void);
const char* scouter_last_error(void);

/* ---- input pipeline on the GPU: the reference's make_transform
 * (dataset/transform_func.py:101-124) = Resize((S,S)) [:19-31, torchvision F.resize -> PIL Image.resize BILINEAR with
 * antialiasing] -> ToTensor [:51-66] -> Normalize [:91-99], for a batch of decoded uint8 HWC images of different sizes.
 * The resize is bit-identical to Pillow's 8-bit resampler (Resample.c); ToTensor+Normalize is a [C][256] float table
 * the caller fills exactly as the reference computes it (float64, then float32).
 * srcs: device array of B image pointers; hw: device [B][2] (height, width); tmp: device scratch for the horizontal
 * pass, image b at tmp + tmp_off[b], size h_b*out_w*C bytes; max_h/max_w: largest height/width in the batch (grid
 * sizing); out: float32 [B][C][out_h][out_w].  Down-scaling factors above 47 are refused. */
int scouter_resize_normalize_u8_f32(const unsigned char* const* srcs, const int* hw, unsigned char* tmp,
                                    const long* tmp_off, const float* lut, float* out, int B, int C, int max_h, int max_w,
                                    int out_h, int out_w, void* stream);

/* optional per-kernel hipEvent timing used by bench.py's roofline leg: when enabled, every launch of the conv / BN /
 * xSlot kernels is bracketed by two hipEvents on ITS stream and aggregated by kernel-instance name.
 * scouter_prof_collect writes "name\tlaunches\ttotal_ms\talgorithmic_flops\talgorithmic_bytes\n" lines into buf
 * (returns the byte count) and clears the log. */
void scouter_prof_enable(int on);
int scouter_prof_collect(char* buf, int cap);

/* ---- convolution = nn.Conv2d call sites: timm/models/resnet.py:491-501 (stem, blocks), resnest.py:111-143,
 * layers/split_attn.py:54-60 (grouped 3x3, fc1, fc2), sloter/slot_model.py:108 (conv1x1) and their autograd
 * backward (engine.py:33).  fp32 implicit GEMM on v_mfma_f32_32x32x2_f32.  Per-group channels must be multiples
 * of 32.  `bias` (per Cout), `addend` (same shape as the output) may be NULL; relu applies last.
 * bn_partial (may be NULL): the epilogue also writes per-M-tile fp64 (sum, sum of squares) of every output channel,
 * [scouter_conv2d_fwd_bn_partial_rows(...)][Cout][2], which scouter_bn_fwd_f32 accepts instead of re-reading y.
 * tile_hint: -1 = built-in heuristic, 0..3 = block tile 128x128 / 128x64 / 64x64 / 128x32 (ignored if illegal for the
 * shape).  Results are bit-identical for every tile (same K order per output element), so callers may autotune it. */
int scouter_conv2d_fwd_bn_partial_rows(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad,
                                       int groups, int tile_hint);
int scouter_conv2d_fwd_f32(const float* x, const float* w, const float* bias, const float* addend, float* y,
                           double* bn_partial, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride,
                           int pad, int groups, int relu, int tile_hint, void* stream);
int scouter_conv2d_dgrad_f32(const float* dy, const float* w, const float* addend, float* dx, int B, int H, int W,
                             int Cin, int Cout, int kh, int kw, int stride, int pad, int groups, int tile_hint,
                             void* stream);
/* Input gradient with the BatchNorm-BACKWARD reductions of its consumer fused into the epilogue.  The tensor being
 * produced is d(out) of a block output out = relu(bn(x1) [+ bn(x2) | + shortcut]) (resnet.py:404-412, resnest.py:128-143;
 * the shortcut / downsample gradient arrives through `addend`).  The epilogue applies the ReLU sign (relu_mask: the bit
 * mask scouter_bn_fwd_f32 wrote, NULL = no ReLU), stores g = d(out) * [out > 0] into dx, and writes per M tile and
 * input channel the fp64 pairs (sum g, sum g * xhat1) to part1 [rows][Cin][2] -- and (sum g, sum g * xhat2) to part2 for
 * a second BatchNorm fed by the same gradient (downsample branch; may be NULL).  x1 / x2: the BatchNorm inputs
 * [B][H][W][Cin]; saved1 / saved2: their [4][Cin] blocks {mean, rstd, scale, shift}.  rows =
 * scouter_conv2d_dgrad_bn_partial_rows(...) (fp32 / bf16-input kernels: ceil(B*H*W / 64) for tile 2, / 128 otherwise;
 * plane kernels: / 128, 128, 128, 64, 256, 256 for tile 0..5, 4 * ceil(B*H*W / 256) for tile 6).  scouter_bn_bwd_f32(ext_partial = part, ext_rows = rows) then
 * runs without its own reduction pass.  part1 == NULL: exactly scouter_conv2d_dgrad_f32.
 * tile_hint 5 (round 5; csrc/conv_pw_persist_x3.h): the PERSISTENT kernel on the bf16 matrix cores (three-way operand split:
 * fp32-grade products, another summation order) -- 1x1 / stride 1 / groups 1 with GEMM-K of 64 / 128 / 256 and 64-multiples of
 * output columns: the input gradient WITH the fused epilogue (one partial row per workgroup row:
 * scouter_conv2d_dgrad_bn_partial_rows(..., 5)), and the forward without bias / addend / ReLU
 * (scouter_conv2d_fwd_bn_partial_rows(..., 5)); named where it does not apply: SC_ERR_UNSUPPORTED, never re-routed. */
int scouter_conv2d_dgrad_bn_partial_rows(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad,
                                         int groups, int tile_hint);
int scouter_conv2d_dgrad_bnbwd_f32(const float* dy, const float* w, const float* addend, float* dx, int B, int H, int W,
                                   int Cin, int Cout, int kh, int kw, int stride, int pad, int groups, int tile_hint,
                                   const void* relu_mask, const float* x1, const float* saved1, double* part1,
                                   const float* x2, const float* saved2, double* part2, void* stream);
/* ---- bf16 matrix inputs, fp32 accumulation (mixed-precision mode of BASELINE configs[4]): same tensors (fp32 in HBM),
 * same semantics and fused epilogues as the fp32 entry points; operands are rounded to bf16 (RNE) on their way into
 * LDS and multiplied on v_mfma_f32_32x32x16_bf16.  The forward takes the weights pre-transposed to bf16
 * [kh*kw][Cout][Cin/groups] (scouter_conv2d_weight_bf16t, 2*kh*kw*Cin/groups*Cout bytes); dgrad reads the fp32 HWIO
 * weights directly (stride 1 only: strided input gradients stay on the fp32 kernel).  wgrad_bf16 covers same-size
 * stride-1 convolutions with per-group channels that are multiples of 32 (32-channel groups on ragged 64-wide tiles) and
 * returns SC_ERR_UNSUPPORTED otherwise (callers then use scouter_conv2d_wgrad_f32); it shares that function's workspace.
 *
 * ACTIVATION STORAGE (`*_io` entry points, round 4): under --precision bf16 the widest tensors of a bottleneck -- the conv3 /
 * downsample-convolution outputs and the block outputs (timm/models/resnest.py:128-143: `out = self.bn3(self.conv3(out))`,
 * `out += residual`, `out = self.act3(out)`) -- may be STORED as bf16.  The typed entry points take untyped pointers plus
 * an `io` bit set: SCOUTER_IO_X_BF16 (1) the main input, SCOUTER_IO_Y_BF16 (2) the output, SCOUTER_IO_R_BF16 (4) the
 * residual; values are widened on load and rounded RNE on store, all arithmetic (and every gradient tensor) stays fp32;
 * a bf16-stored convolution input holds exactly the values the kernel rounds an fp32 input to, so the product is the
 * same bits.  io = 0 is the fp32 entry point. */
#define SCOUTER_IO_X_BF16 1
#define SCOUTER_IO_Y_BF16 2
#define SCOUTER_IO_R_BF16 4
int scouter_conv2d_weight_bf16t(const float* w, void* wt, int kh, int kw, int Cin, int Cout, int groups, void* stream);
int scouter_conv2d_fwd_bf16(const float* x, const void* wt_bf16, const float* bias, const float* addend, float* y,
                            double* bn_partial, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride,
                            int pad, int groups, int relu, int tile_hint, void* stream);
int scouter_conv2d_fwd_bf16_io(const void* x, const void* wt_bf16, const float* bias, const float* addend, void* y,
                               double* bn_partial, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride,
                               int pad, int groups, int relu, int tile_hint, int io, void* stream);
int scouter_conv2d_dgrad_bf16(const float* dy, const float* w, const float* addend, float* dx, int B, int H, int W,
                              int Cin, int Cout, int kh, int kw, int stride, int pad, int groups, int tile_hint,
                              void* stream);
int scouter_conv2d_dgrad_bnbwd_bf16(const float* dy, const float* w, const float* addend, float* dx, int B, int H, int W,
                                    int Cin, int Cout, int kh, int kw, int stride, int pad, int groups, int tile_hint,
                                    const void* relu_mask, const float* x1, const float* saved1, double* part1,
                                    const float* x2, const float* saved2, double* part2, void* stream);
/* io bits of the typed input gradient: 1 -- x1 is stored as bf16, 2 -- x2 is (the BatchNorm inputs the fused epilogue
 * reads), 4 -- the addend is, 8 -- dy is (the values the kernel rounds an fp32 dy to: same result), 16 -- dx is */
#define SCOUTER_DGRAD_IO_X1 1
#define SCOUTER_DGRAD_IO_X2 2
#define SCOUTER_DGRAD_IO_ADDEND 4
#define SCOUTER_DGRAD_IO_DY 8
#define SCOUTER_DGRAD_IO_DX 16
int scouter_conv2d_dgrad_bnbwd_bf16_io(const void* dy, const float* w, const void* addend, void* dx, int B, int H,
                                       int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int groups,
                                       int tile_hint, const void* relu_mask, const void* x1, const float* saved1,
                                       double* part1, const void* x2, const float* saved2, double* part2, int io,
                                       void* stream);
/* tile_hint 4 of the typed FORWARD (round 5): the persistent pointwise kernel (csrc/conv_pw_persist_bf16.h pwb_fwd_kernel) --
 * 1x1 / stride 1 / groups 1, 64 / 128 / 256 / 512 input channels STORED as bf16 (SCOUTER_IO_X_BF16), 128-multiples of output
 * channels, no bias / addend / ReLU; y bf16 or fp32; bn_partial rows = scouter_conv2d_fwd_bn_partial_rows_bf16(..., 4)
 * (tile_hint 0-3: as scouter_conv2d_fwd_bn_partial_rows).  Named where it does not apply: SC_ERR_UNSUPPORTED. */
int scouter_conv2d_fwd_bn_partial_rows_bf16(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad,
                                            int groups, int tile_hint);
/* tile_hint 4 of the typed input gradient (round 5): the PERSISTENT pointwise kernel (csrc/conv_pw_persist_bf16.h) -- 1x1 /
 * stride 1 / groups 1, Cout of 64 / 128 / 256 / 512, 128-multiples of Cin, the fused BatchNorm-backward epilogue present
 * (part1 != NULL) and EVERY tensor stored as bf16 (io = DY | DX | X1 [| X2] [| ADDEND]); anything else named with tile 4 is
 * SC_ERR_UNSUPPORTED, never re-routed.  WITHOUT the fused epilogue (part1 == NULL) tile 4 is the plain persistent kernel:
 * Cout of 64 ... 1024, 64-multiples of Cin, dy stored as bf16 (DY), dx fp32, optional addend (fp32 or bf16).
 * The fused kernel writes ONE partial row per workgroup row: rows =
 * scouter_conv2d_dgrad_bn_partial_rows_bf16(..., 4, part2 != NULL) (tile_hint 0-3: the rows of the bf16-input tiles, as
 * scouter_conv2d_dgrad_bn_partial_rows). */
int scouter_conv2d_dgrad_bn_partial_rows_bf16(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad,
                                              int groups, int tile_hint, int two_batchnorms);
int scouter_conv2d_wgrad_bf16(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout, int kh,
                              int kw, int stride, int pad, int groups, int plan_hint, void* ws, size_t ws_bytes,
                              void* arrival, int arrival_slots, void* stream);
/* io: SCOUTER_IO_X_BF16 -- x is stored as bf16, SCOUTER_IO_R_BF16 -- dy is.  plan_hint < 0 (the library's own plan) with BOTH
 * operands stored as bf16 on a 3x3 / stride 1 / pad 1 layer with 32 input channels per group and rows of at most 112 pixels: the
 * tap-fused kernel (every x row fetched once per workgroup, not once per tap; csrc/conv_wgrad_taps_bf16.h; SCOUTER_BWT=0: off).
 * Same workspace bound (scouter_conv2d_wgrad_workspace_bytes with the same plan_hint) and deterministic slab sum. */
int scouter_conv2d_wgrad_bf16_io(const void* x, const void* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                                 int kh, int kw, int stride, int pad, int groups, int plan_hint, void* ws,
                                 size_t ws_bytes, void* arrival, int arrival_slots, int io, void* stream);
/* plan_hint: -1 = built-in plan; otherwise bits 0-1 = block budget {512,1024,2048,4096} (sets the split-K count),
 * bit 4 / bit 5 = halve the ci / co tile edge.  Every plan is deterministic; different plans sum the pixels in a
 * different order (results differ in the last bits), so a caller that autotunes keeps its choice for the whole run.
 * arrival (may be NULL), arrival_slots: caller-owned device buffer of arrival_slots 32-bit counters, ZERO on entry and left
 * zero on exit, private to the stream (two launches that overlap in time must not share it).  With it the split-K slabs
 * are summed INSIDE the kernel -- the last workgroup of an output tile to arrive adds the tile's partials in slab order
 * (deterministic) and writes dw; without it (or when the plan has more output tiles than slots) a separate slab-sum
 * kernel is launched.  The two paths add the slabs in different association orders: results agree to fp32 rounding. */
size_t scouter_conv2d_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad,
                                            int groups, int plan_hint);
int scouter_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout, int kh,
                             int kw, int stride, int pad, int groups, int plan_hint, void* ws, size_t ws_bytes,
                             void* arrival, int arrival_slots, void* stream);
/* stem (Cin < 32): im2col of the NCHW image into [M][Kpad] rows so the stem runs as a 1x1 conv on the same path */
int scouter_im2col_nchw_f32(const float* x, float* col, int B, int Cin, int H, int W, int k, int stride, int pad,
                            int Kpad, void* stream);
int scouter_pad_rows_f32(const float* w, float* wpad, long nvalid, long ntotal, void* stream);
/* deep stem's first convolution (timm/models/resnet.py deep stem conv1[0], the 3 -> 32 channel 3x3 / stride 2 / pad 1 layer of
 * resnest26d / resnest50d), direct: x NCHW [B][3][H][W], w HWIO [3][3][3][32], y NHWC [B][Ho][Wo][32]; no patch rows are
 * written.  bn_partial (optional): [scouter_stem_direct_partial_rows(B, H)][32][2] fp64 (sum, sum of squares) of y, the layout
 * scouter_bn_fwd_f32 takes from scouter_conv2d_fwd_f32.  Cout != 32 or an image too wide for the LDS row stage: SC_ERR_UNSUPPORTED
 * (the caller keeps the im2col route). */
int scouter_stem_direct_partial_rows(int B, int H);
int scouter_stem_direct_fwd_f32(const float* x, const float* w, float* y, double* bn_partial, int B, int H, int W, int Cout,
                                void* stream);

/* ---- BatchNorm2d (+ReLU, +residual add): timm/models/resnet.py:383 (norm_layer), BasicBlock :172-199,
 * ResNestBottleneck resnest.py:111-143.  Training mode: batch statistics (fp64 accumulation), running stats
 * updated with `momentum` and the unbiased variance; eval mode: running stats.
 * y = (x - mean) * scale + shift with scale = gamma*rstd, shift = beta; mean/rstd/scale are saved for the backward.
 * relu_mask_out (optional, needs relu): sign bitmask of y, scouter_relu_mask_words(M*C) 64-bit words (1 bit/element);
 * handing it to scouter_bn_bwd_f32 as relu_mask replaces the 4-byte-per-element read of ymask in both backward passes.
 * planes_out (optional): the output ALSO as nplanes (1 or 3) bf16 operand planes for scouter_conv2d_fwd_planes
 * (likewise dx_planes of scouter_sa_bn_bwd_f32 for scouter_conv2d_dgrad_planes).
 * residual_bn_saved (optional, with residual): `residual` is the RAW output of the downsample convolution and this is
 * its BatchNorm's saved block [4][C] (statistics finalised by a y == NULL call): y = relu(bn(x) + bn_ds(residual)) in
 * one pass -- the downsample BatchNorm's output is never stored (resnet.py:404-412 `x += residual` with
 * residual = downsample(x)). */
size_t scouter_relu_mask_words(long n);
size_t scouter_colreduce_workspace_bytes(long M, int C);
int scouter_bn_fwd_f32(const float* x, float* y, const float* residual, long M, int C, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                       int training, int relu, float* mean_out, float* rstd_out, float* scale_out, float* shift_out,
                       const double* ext_partial, int ext_rows, unsigned long long* relu_mask_out, void* planes_out,
                       int nplanes, const float* residual_bn_saved, void* ws, size_t ws_bytes, void* stream);
/* typed storage (see ACTIVATION STORAGE above): x / y / residual per `io`; a bf16 x needs ext_partial in training mode
 * (its batch statistics are those of the producing convolution's fp32 accumulators) */
int scouter_bn_fwd_io(const void* x, void* y, const void* residual, long M, int C, const float* gamma,
                      const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                      int training, int relu, float* mean_out, float* rstd_out, float* scale_out, float* shift_out,
                      const double* ext_partial, int ext_rows, unsigned long long* relu_mask_out, void* planes_out,
                      int nplanes, const float* residual_bn_saved, int io, void* ws, size_t ws_bytes, void* stream);
/* y == NULL in scouter_bn_fwd_f32: with planes_out the apply pass writes the planes only (every consumer reads planes: 6
 * instead of 10 bytes per element); without, statistics / running-stat update only.  scouter_bn_apply_f32: the apply pass alone
 * from a saved block [4][C] = {mean, rstd, scale, shift} (contiguous rows, as written through the four *_out pointers). */
int scouter_bn_apply_f32(const float* x, const float* bn_saved, float* y, long M, int C, int relu, void* stream);
/* g = dy * (y > 0), the sign taken from relu_mask if given, else from ymask (both may be NULL: no ReLU);
 * dgamma/dbeta may be NULL (frozen); gout (may be NULL) receives g.
 * ext_partial (may be NULL): [ext_rows][C][2] fp64 (sum g, sum g * xhat) already reduced by the epilogue of the
 * input-gradient kernel that produced dy (scouter_conv2d_dgrad_bnbwd_*): dy then IS g (ymask / relu_mask / gout must be
 * NULL) and the reduction pass over (dy, x) is skipped. */
int scouter_bn_bwd_f32(const float* dy, const float* ymask, const float* x, const float* mean, const float* rstd,
                       const float* scale, const unsigned long long* relu_mask, long M, int C, int training,
                       float* dgamma, float* dbeta, float* dx, float* gout, const double* ext_partial, int ext_rows,
                       void* ws, size_t ws_bytes, void* stream);
/* io: SCOUTER_IO_X_BF16 -- the BatchNorm input x is stored as bf16; SCOUTER_IO_Y_BF16 -- dx is stored as bf16 (for a dx
 * read only by bf16-input convolution kernels, which round it the same way: same results); SCOUTER_IO_R_BF16 -- dy is
 * stored as bf16 (the masked block-output gradient of scouter_conv2d_dgrad_bnbwd_bf16_io; gout must be NULL) */
int scouter_bn_bwd_io(const void* dy, const float* ymask, const void* x, const float* mean, const float* rstd,
                      const float* scale, const unsigned long long* relu_mask, long M, int C, int training,
                      float* dgamma, float* dbeta, void* dx, float* gout, const double* ext_partial, int ext_rows,
                      int io, void* ws, size_t ws_bytes, void* stream);
/* out[c] = alpha * sum_m a[m][c] * (b ? b[m][c] : 1)  -- bias gradients, d(initial_slots) */
int scouter_colsum_f32(const float* a, const float* b, float* out, long M, int C, float alpha, void* ws,
                       size_t ws_bytes, void* stream);
int scouter_relu_bwd_f32(const float* dy, const float* y, float* dx, long n, void* stream);
int scouter_axpby_f32(const float* a, const float* b, float* y, float alpha, float beta, long n, void* stream);

/* ---- pooling: MaxPool2d(3,2,1) resnet.py:412; AvgPool2d(3,s,1) resnest.py:93 (avd); AvgPool2d(2,s,ceil,
 * count_include_pad=False) resnet.py:300 (downsample_avg) */
int scouter_pool_out_size(int in, int k, int stride, int pad, int ceil_mode);
int scouter_maxpool_fwd_f32(const float* x, float* y, unsigned char* argmax, int B, int H, int W, int C, int k,
                            int stride, int pad, void* stream);
int scouter_maxpool_bwd_f32(const float* dy, const unsigned char* argmax, float* dx, int B, int H, int W, int C, int k,
                            int stride, int pad, void* stream);
/* BatchNorm + ReLU + MaxPool2d fused in both directions (the stem: bn1, act1, maxpool -- timm/models/resnet.py:404-412,
 * 494-496).  bn_saved = the [4][C] block {mean, rstd, scale, shift} finalised by scouter_bn_fwd_f32(y = NULL).
 * fwd: y = maxpool(relu(bn(x))) and the arg-max taps; the activation and its sign mask are never stored.
 * bwd: dy = gradient of the pooled tensor -> dx = gradient of the BatchNorm INPUT, dgamma / dbeta (may be NULL);
 * workspace scouter_colreduce_workspace_bytes(B*Ho*Wo, C) + 2*C*4 bytes; stride 1 / 2. */
int scouter_bn_maxpool_fwd_f32(const float* x, const float* bn_saved, float* y, unsigned char* argmax, int B, int H, int W,
                               int C, int k, int stride, int pad, void* stream);
int scouter_bn_maxpool_bwd_f32(const float* dy, const unsigned char* argmax, const float* x, const float* bn_saved, int B,
                               int H, int W, int C, int k, int stride, int pad, int training, float* dgamma,
                               float* dbeta, float* dx, void* ws, size_t ws_bytes, void* stream);
/* io & SCOUTER_IO_Y_BF16: dx is stored as bf16 */
int scouter_bn_maxpool_bwd_io(const float* dy, const unsigned char* argmax, const float* x, const float* bn_saved, int B,
                              int H, int W, int C, int k, int stride, int pad, int training, float* dgamma,
                              float* dbeta, void* dx, int io, void* ws, size_t ws_bytes, void* stream);
int scouter_avgpool_fwd_f32(const float* x, float* y, int B, int H, int W, int C, int k, int stride, int pad,
                            int ceil_mode, int count_include_pad, void* stream);
int scouter_avgpool_fwd_io(const void* x, float* y, int B, int H, int W, int C, int k, int stride, int pad,
                           int ceil_mode, int count_include_pad, int io, void* stream);    /* io: SCOUTER_IO_X_BF16 */
int scouter_avgpool_bwd_f32(const float* dy, float* dx, int B, int H, int W, int C, int k, int stride, int pad,
                            int ceil_mode, int count_include_pad, void* stream);
int scouter_transpose_f32(const float* in, float* out, int batch, int rows, int cols, void* stream);

/* ---- split attention glue: timm/models/layers/split_attn.py:62-80 (radix 2, cardinality 1).
 * bn_saved (may be NULL) = the [4][2Cp] block {mean, rstd, scale, shift} scouter_bn_fwd_f32 wrote for bn0: x is then the
 * RAW output of the radix convolution and relu(bn0(x)) is evaluated on the fly -- the activation is never stored
 * (scouter_bn_fwd_f32 with y == NULL only finalises the statistics).  scouter_sa_bn_bwd_f32 is the matching backward:
 * gradient w.r.t. the split-attention input h0 (from dout, the attention weights a and d(gap)), ReLU mask recomputed from
 * x0, BatchNorm backward of bn0 (dgamma / dbeta may be NULL) -> dx = gradient w.r.t. the convolution output.  * bn_sums (optional): the per-image statistics [B][2C'][4] (fp64) that scouter_sa_reduce_f32(mode 1, bn_sums_out) produced
 * in the d(attention) pass -- g = (dout * a + dgap / HW) * [bn0(x0) > 0] is affine in (a, dgap), constants of an image,
 * so its BatchNorm-backward sums follow from them and the reduction pass over (dout, x0) is skipped. */
size_t scouter_sa_workspace_bytes(int B, int HW, int C2);
/* typed storage of the radix convolution's raw output x / x0 (SCOUTER_IO_X_BF16) and of the attention-weighted sum `out`
 * (SCOUTER_IO_Y_BF16) -- see ACTIVATION STORAGE above */
int scouter_sa_reduce_io(const void* x, const float* dout, const float* bn_saved, float* out, double* bn_sums_out, int B,
                         int HW, int Cp, int mode, int io, void* ws, size_t ws_bytes, void* stream);
int scouter_sa_bn_bwd_io(const float* dout, const float* a, const float* dgap, const void* x0, const float* bn_saved,
                         const double* bn_sums, int B, int HW, int Cp, int training, float* dgamma, float* dbeta,
                         float* dx, void* dx_planes, int nplanes, int io, void* ws, size_t ws_bytes, void* stream);
int scouter_sa_apply_fwd_io(const void* x, const float* a, const float* bn_saved, void* out, int B, int HW, int Cp, int io,
                            void* stream);
int scouter_sa_reduce_f32(const float* x, const float* dout, const float* bn_saved, float* out, double* bn_sums_out,
                          int B, int HW, int Cp, int mode, void* ws, size_t ws_bytes, void* stream);
int scouter_sa_bn_bwd_f32(const float* dout, const float* a, const float* dgap, const float* x0, const float* bn_saved,
                          const double* bn_sums, int B, int HW, int Cp, int training, float* dgamma, float* dbeta,
                          float* dx, void* dx_planes, int nplanes, void* ws, size_t ws_bytes, void* stream);
int scouter_radix_softmax_fwd_f32(const float* z, float* a, int B, int Cp, void* stream);
int scouter_radix_softmax_bwd_f32(const float* a, const float* da, float* dz, int B, int Cp, void* stream);
int scouter_sa_apply_fwd_f32(const float* x, const float* a, const float* bn_saved, float* out, int B, int HW, int Cp,
                             void* stream);
int scouter_sa_apply_bwd_f32(const float* dout, const float* a, const float* dgap, float* dx, int B, int HW, int Cp,
                             void* stream);

/* ---- xSlot head: sloter/utils/position_encode.py:26-46; sloter/utils/slot_attention.py:44-96;
 * sloter/slot_model.py:117-121; tools/calculate_tool.py:4-7 */
int scouter_posenc_sine_f32(float* pe /* [h*w][d] token-major */, int h, int w, int d, void* stream);
/* X: [B][N][d] tokens after conv1x1+ReLU; PE: [N][d]; tok_w / tok_b: HOST arrays of L device pointers, one
 * [d][d] (out,in) weight / [d] bias per to_k Linear; slots0: [S][d];
 * GRU weights in nn.GRU layout (gate order r,z,n).  Outputs: logits [B][S/spc], attn [B][S][N] (last iteration),
 * area_part [B] (per-image sum of attn).  Saved for backward: Ksave [B][N][d], Hsave [L][B][N][d] (the INPUT of
 * every to_k layer: Hsave[0] = X+PE), states [T-1][B][S][d] (slots entering iterations 2..T). */
int scouter_xslot_fwd_f32(const float* X, const float* PE, const float* const* tok_w, const float* const* tok_b,
                          const float* slots0,
                          const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int B, int N,
                          int d, int S, int spc, int T, int L, float loss_status, float* logits, float* attn,
                          float* area_part, float* Ksave, float* Hsave, float* states, void* stream);
size_t scouter_xslot_bwd_workspace_bytes(int B, int N, int d, int S, int T);
/* dlogits: [B][S/spc]; g_area_sum: device scalar dL/d(sum attn_T).  Outputs: dX [B][N][d]; dgi, dgh
 * [T-1][B][S][3d] and Usave [T-1][B][S][d] (operands of the GRU weight-gradient GEMMs); ds0 [B][S][d];
 * dZ [L][B][N][d] (pre-activation gradients of the to_k layers). */
int scouter_xslot_bwd_f32(const float* X, const float* PE, const float* const* tok_w, const float* slots0,
                          const float* w_ih,
                          const float* w_hh, const float* b_ih, const float* b_hh, const float* Ksave,
                          const float* Hsave, const float* states, const float* dlogits, const float* g_area_sum,
                          int B, int N, int d, int S, int spc, int T, int L, float loss_status, float* dX, float* dgi,
                          float* dgh, float* Usave, float* ds0, float* dZ, void* ws, size_t ws_bytes, void* stream);
/* stats[0..4] = loss, nll, area**power, top-1 accuracy, area.  labels: int64 (may be NULL -> log_softmax only). */
int scouter_slot_loss_fwd_f32(const float* logits, const long* labels, const float* area_part, int n_area_part, int B,
                              int C, double area_count, float lambda, float power, float* logp, float* stats,
                              void* stream);
int scouter_slot_loss_bwd_f32(const float* logp, const long* labels, const float* stats, const float* g_loss,
                              const float* g_nll, const float* g_term, const float* g_logp, int B, int C,
                              double area_count, float lambda, float power, float* dlogits, float* g_area_sum,
                              void* stream);

/* ---- FC baseline head (`use_slot=False`): global average pool = scouter_avgpool_*_f32 with k = H, then
 * nn.Linear (weight [C][K], sloter/slot_model.py:116-125, timm/models/resnet.py:503-509).  dx / dw / db may be NULL. */
int scouter_linear_small_fwd_f32(const float* x, const float* w, const float* bias, float* y, int B, int K, int C,
                                 void* stream);
int scouter_linear_small_bwd_f32(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db,
                                 int B, int K, int C, void* stream);

/* ---- convolution on pre-split bf16 operand planes (scouter_amd/csrc/conv_planes.hip; same reference call sites as
 * scouter_conv2d_*_f32).  nplanes = 3: x = hi + mid + lo exactly, six bf16 MFMA products per fp32 product -> fp32
 * accuracy at 2.7x the fp32-MFMA rate;  nplanes = 1: plain bf16 inputs (BASELINE configs[4]).
 * planes: [nplanes][elements] bf16, every plane laid out like the fp32 tensor.  Weight planes (k contiguous):
 * w_fwd [nplanes][tap][Cout][Cin/groups], w_dgrad [nplanes][tap][Cin][Cout/groups]; either may be NULL.
 * tile (block tile, callers autotune it): nplanes = 3: 0 = 128x128, 2 = 128x64 (two workgroups per CU), 3 = 64x64,
 * 4 = 256x128 (eight waves), 5 = 256 x (128 | 64) with the input rows resident in LDS for all nine taps (same-size 3x3
 * layers on maps up to 63 pixels wide; falls back to tile 0 / 1 otherwise); nplanes = 1: 0 = 128x128, 1 = 128x64;
 * 6 (1 or 3 planes) = PERSISTENT 256 x (128 | 64): one workgroup per CU walks the tile list, the LDS-DMA stream runs on into
 * the next tile, register epilogue (csrc/conv_planes_persist.h; needs >= 2 K-tiles of 32 channels x taps).
 * Fused BatchNorm-statistics rows = ceil(M / BM) with BM = 128, 128, 128, 64, 256, 256 for tile 0..5; tile 6 writes one row
 * per 64-row wave row of its 256-row tiles: 4 * ceil(M / 256).  scouter_conv2d_fwd_planes_bn_partial_rows returns the
 * largest count any tile writes (4 * ceil(M / 256)).  Tile 6 with bn_partial AND bias / addend, or with fewer than two
 * K-tiles, is refused with SC_ERR_UNSUPPORTED (no silent re-routing: the row layout depends on the tile).  Tiles 0-4 and 6
 * sum every output ELEMENT in the same K order (bit-identical convolution outputs); the fused BatchNorm statistics are
 * per-tile partial sums, grouped differently per tile, so batch statistics -- and everything downstream -- agree between
 * tiles to fp64 rounding of the partial sums, not bit for bit; tile 5 sums 16-channel chunks outer / taps inner. */
int scouter_planes_split_f32(const float* x, void* planes, long n, int nplanes, void* stream);
int scouter_planes_split_weight_f32(const float* w_hwio, void* w_fwd, void* w_dgrad, int kh, int kw, int Cin, int Cout,
                                    int groups, int nplanes, void* stream);
/* The weight split of EVERY plane convolution of a model in one launch (the weights change with every optimizer step):
 * `table` = nrows device-resident rows of scouter_planes_split_weights_row_bytes() bytes each, laid out as
 * { const float* w_hwio; void* w_fwd; void* w_dgrad; int taps, cin_per_group, cout, groups; long first; } with `first` the
 * running element count and total = sum of taps * cin_per_group * cout. */
size_t scouter_planes_split_weights_row_bytes(void);
int scouter_planes_split_weights_multi(const void* table, int nrows, long total, int nplanes, void* stream);
int scouter_conv2d_fwd_planes_bn_partial_rows(int B, int H, int W, int kh, int kw, int stride, int pad);
/* io & SCOUTER_IO_Y_BF16: y is stored as bf16 (tiles 0-5; the persistent tile 6 writes fp32 only) */
int scouter_conv2d_fwd_planes_io(const void* x_planes, const void* w_planes, const float* bias, const float* addend,
                                 void* y, double* bn_partial, int B, int H, int W, int Cin, int Cout, int kh, int kw,
                                 int stride, int pad, int groups, int relu, int nplanes, int tile, int io, void* stream);
int scouter_conv2d_fwd_planes(const void* x_planes, const void* w_planes, const float* bias, const float* addend,
                              float* y, double* bn_partial, int B, int H, int W, int Cin, int Cout, int kh, int kw,
                              int stride, int pad, int groups, int relu, int nplanes, int tile, void* stream);
int scouter_conv2d_dgrad_planes(const void* dy_planes, const void* w_planes, const float* addend, float* dx, int B, int H,
                                int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int groups, int nplanes,
                                int tile, void* stream);
int scouter_conv2d_dgrad_planes_bnbwd(const void* dy_planes, const void* w_planes, const float* addend, float* dx, int B,
                                      int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int groups,
                                      int nplanes, int tile, const void* relu_mask, const float* x1,
                                      const float* saved1, double* part1, const float* x2, const float* saved2,
                                      double* part2, void* stream);

/* weight gradient on planes: same-size stride-1 convolutions (2 * pad == k - 1), 64-multiples of channels per group;
 * x_planes [nplanes][B*H*W][Cin], dy_planes [nplanes][B*H*W][Cout]; dw HWIO fp32; split-K slabs in ws, summed in a
 * fixed order.  plan_hint: -1 = static plan; else bits 0-1 = workgroup budget {512, 1024, 2048, 4096}, bit 4 / bit 5 = 64
 * instead of 128 input / output channels per tile; bit 6 (3x3, pad 1, maps up to 63 wide) = the TAP-FUSED kernel
 * (csrc/conv_planes_wgrad_taps.h: one workgroup accumulates all nine taps of a 64 x 64 tile from an LDS-resident ring of X
 * rows), bits 0-1 then = budget {256, 512, 1024, 2048} workgroups (callers autotune it; every plan is deterministic,
 * different plans sum the pixels in a different order).  arrival / arrival_slots: as for scouter_conv2d_wgrad_f32. */
size_t scouter_conv2d_wgrad_planes_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw, int groups,
                                                   int plan_hint);
int scouter_conv2d_wgrad_planes(const void* x_planes, const void* dy_planes, float* dw, int B, int H, int W, int Cin,
                                int Cout, int kh, int kw, int pad, int groups, int nplanes, int plan_hint,
                                void* ws, size_t ws_bytes, void* arrival, int arrival_slots, void* stream);

/* ---- Convolutions at fp32 accuracy on the bf16 matrix cores with the three-way operand split done in registers
 * (csrc/conv_x3.hip, round 5): the deep 1x1 layers of the bottlenecks -- conv1 / conv3 (timm/models/resnest.py:111-143),
 * the downsample convolution (resnet.py:292-306) -- and the stride-1 3x3 layers with 32 channels per group (the deep stem,
 * resnet.py:471-489, and the first radix convolution, layers/split_attn.py:54-60).  The ACTIVATION / GRADIENT is a plain fp32
 * NHWC tensor (no producer writes planes); only the WEIGHT arrives as three bf16 planes (scouter_planes_split_weight_f32 /
 * _weights_multi: forward layout [3][taps][Cout][Cin/groups], input-gradient layout [3][taps][Cin][Cout/groups]).  Results are
 * bit-identical to scouter_conv2d_fwd_planes / _dgrad_planes (nplanes = 3) on the split activation.  kh x kw: 1x1 (pad 0) or
 * an odd same-size filter (2 * pad == kh - 1), stride 1; 32-multiples of channels per group, at least two 32-channel K-tiles.
 * tile_hint: 0 = 256x128 (eight waves), 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 256x64 (eight waves), 5 = 256x32, 6 = 128x32 (two waves), else the
 * library's choice (scouter_conv2d_x3_tile tells which for N = output columns per group; every tile gives the same bits).
 * bn_partial: as scouter_conv2d_fwd_f32, one row per M tile = scouter_conv2d_x3_partial_rows(M, N, tile_hint).  The input
 * gradient takes the optional fused BatchNorm-backward epilogue of scouter_conv2d_dgrad_bnbwd_f32 (relu_mask .. part2).
 * tile_hint 7 (round 6, csrc/conv_xhalo.hip; only on request, never the library's choice): the 3x3 / pad 1 layers whose GEMM
 * is 32 columns wide per group -- forward with Cout / groups == 32, input gradient with Cin / groups == 32 (the deep stem's
 * 32 -> 32 and 32 -> 64 convolutions, resnet.py:471-489; the input gradient of layer1's radix convolution,
 * layers/split_attn.py:54-60), maps up to 126 pixels wide, no bias / forward addend -- on a PERSISTENT kernel that splits
 * every input element once and keeps the split rows resident in LDS for all nine taps.  Same error bound as the other
 * tiles, another summation order (16-channel chunk outer, tap inner): equal to fp32 rounding, not bit for bit.  Its
 * bn_partial / part1 / part2 have scouter_conv2d_x3_halo_partial_rows(groups) rows (one per workgroup of a group). */
int scouter_conv2d_x3_tile(long M, int N, int tile_hint);
int scouter_conv2d_x3_partial_rows(long M, int N, int tile_hint);
int scouter_conv2d_x3_halo_partial_rows(int groups);
int scouter_conv2d_fwd_x3(const float* x, const void* w_planes_fwd, const float* bias, const float* addend, float* y,
                          double* bn_partial, int B, int H, int W, int Cin, int Cout, int kh, int kw, int pad, int groups,
                          int relu, int tile_hint, void* stream);
int scouter_conv2d_dgrad_x3_bnbwd(const float* dy, const void* w_planes_dgrad, const float* addend, float* dx, int B,
                                  int H, int W, int Cin, int Cout, int kh, int kw, int pad, int groups, int tile_hint,
                                  const void* relu_mask, const float* x1, const float* saved1, double* part1,
                                  const float* x2, const float* saved2, double* part2, void* stream);

/* Weight gradient of the same layers, dw[Cin][Cout] = x^T dy over the B*H*W pixels (the HWIO gradient of a 1x1 layer), both
 * fp32 operands transposed (4x4 register patches) and split three-way on their way into LDS; 128-multiples of Cin and Cout.
 * Split-K over the pixels into `ws` slabs summed in slab order (deterministic).  plan_hint: -1 / bits 0-1 = workgroup budget
 * 256 << hint (callers take it from the static table; different plans sum the pixels in another order). */
size_t scouter_conv2d_wgrad_x3_workspace_bytes(int B, int H, int W, int Cin, int Cout, int plan_hint);
int scouter_conv2d_wgrad_x3(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                            int plan_hint, void* ws, size_t ws_bytes, void* stream);

/* p[i] += v for n 64-bit counters (the num_batches_tracked buffers of a model as views of one flat buffer: one launch per
 * training forward instead of one ATen launch; torch.nn.BatchNorm2d increments them one by one) */
int scouter_iadd_i64(long* p, long n, long v, void* stream);

/* ---- optimizer: torch.optim.AdamW defaults (train.py:146).  chunk table = array of {float* param; long offset
 * into the flat grad/moment arenas; int n; int pad} built by the host (scouter_adamw_chunk_bytes() each). */
int scouter_adamw_chunk_bytes(void);
int scouter_adamw_step_f32(const void* chunk_table, int nchunks, const float* grads, float* exp_avg,
                           float* exp_avg_sq, float lr, float beta1, float beta2, float eps, float weight_decay,
                           int step, void* stream);
/* The same update with the step-dependent scalars read from DEVICE memory, so that the launch can live in a captured
 * hipGraph (scouter_amd/graph.py): dyn[0] = learning rate, dyn[1] = step count t >= 1 (as a float, exact below 2^24);
 * the bias corrections 1 - beta^t are formed in the kernel.  The host advances dyn[1] before each launch. */
int scouter_adamw_step_dev_f32(const void* chunk_table, int nchunks, const float* grads, float* exp_avg,
                               float* exp_avg_sq, const float* dyn, float beta1, float beta2, float eps,
                               float weight_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif
