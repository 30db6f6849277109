/*
 * fmgan_hip.h — C-ABI of libfmgan_hip.so: the MI355X (gfx950) kernels behind the
 * 3D-FM GAN forward hot path `(photo, render) -> image`.
 *
 * This is the drop-in boundary.  Every entry point takes plain device pointers,
 * sizes and a `hipStream_t` (passed as void*), launches asynchronously on that
 * stream, never allocates, never synchronises, keeps no global state and is
 * re-entrant (one call per Python thread / per rank is fine).  The caller owns
 * all buffers; outputs must be pre-allocated (the reference allocates inside the
 * op with at::empty — here the host-side shim does that).
 *
 * Return value: FMGAN_OK (0) or a negative FMGAN_E* code.  The reference checks
 * only "is a CUDA tensor" (op/upfirdn2d.cpp:8,15-16, op/fused_bias_act.cpp:7,13-14)
 * and never calls cudaGetLastError; this library additionally validates shapes
 * and reports launch failures, and the Python shim turns any non-zero status into
 * RuntimeError (same exception type as TORCH_CHECK).
 *
 * Reference interfaces replaced (paths relative to the reference repo root):
 *   fmgan_upfirdn2d        <- upfirdn2d_op()        op/upfirdn2d_kernel.cu:209-369
 *                             (pybind `upfirdn2d`    op/upfirdn2d.cpp:12-23)
 *   fmgan_fused_bias_act   <- fused_bias_act_op()   op/fused_bias_act_kernel.cu:52-99
 *                             (pybind `fused_bias_act` op/fused_bias_act.cpp:11-21)
 *   fmgan_modconv_demod,
 *   fmgan_modconv2d        <- ModulatedConv2d.forward  stylegan2.py:250-298
 *                             (F.conv2d / F.conv_transpose2d with groups=batch)
 *   fmgan_torgb            <- ToRGB.forward            stylegan2.py:389-404
 *   fmgan_images_to_tensor <- transforms.ToTensor()+Normalize   train_3_encoder.py:233-239
 *   fmgan_resize_*         <- transforms.Resize(size) (PIL BILINEAR) train_3_encoder.py:235
 *   fmgan_tensor_to_images <- tensor2im                 Evaluation/visual_eval.py:24-38
 */
#ifndef FMGAN_HIP_H
#define FMGAN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define FMGAN_ABI_VERSION 1

/* status codes */
#define FMGAN_OK            0
#define FMGAN_EINVAL       -1   /* null pointer / non-positive or inconsistent dims */
#define FMGAN_EUNSUPPORTED -2   /* dtype / act / mode outside what the reference dispatches */
#define FMGAN_ELAUNCH      -3   /* hipGetLastError() != hipSuccess after the launch */
#define FMGAN_EOVERFLOW    -4   /* an element count does not fit the kernel's index type */

/* element types (the reference dispatches float/double/half:
 * op/upfirdn2d_kernel.cu:311, op/fused_bias_act_kernel.cu:79) */
#define FMGAN_F32 0
#define FMGAN_F64 1
#define FMGAN_F16 2

int         fmgan_abi_version(void);
const char *fmgan_status_string(int status);

/* Which upfirdn2d kernel `fmgan_upfirdn2d` would pick for these arguments:
 * 0 generic, 1 row-march (up=down=1, wide rows), 2 LDS plane-tile (up=down=1,
 * small planes), 3 up=2 polyphase.  Pure host logic, no GPU needed.
 * `force_path` in fmgan_upfirdn2d uses the same numbering (-1 = automatic). */
int fmgan_upfirdn2d_select(int dtype, int major, int in_h, int in_w, int minor,
                           int kernel_h, int kernel_w, int up_x, int up_y,
                           int down_x, int down_y, int pad_x0, int pad_x1,
                           int pad_y0, int pad_y1);

/* out_h/out_w exactly as upfirdn2d_op computes them (op/upfirdn2d_kernel.cu:237-240). */
int fmgan_upfirdn2d_out_size(int in_h, int in_w, int kernel_h, int kernel_w,
                             int up_x, int up_y, int down_x, int down_y,
                             int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                             int *out_h, int *out_w);

/*
 * upfirdn2d: zero-stuff by (up_y,up_x), pad/crop by (pad_*; negative = crop),
 * correlate with the FLIPPED `kernel` (i.e. true convolution), keep every
 * (down_y,down_x)-th sample.
 *   input  [major, in_h, in_w, minor]   contiguous, `dtype`
 *   kernel [kernel_h, kernel_w]         contiguous, `dtype` (un-flipped, as the caller holds it)
 *   out    [major, out_h, out_w, minor] contiguous, `dtype`, pre-allocated
 * Semantics: op/upfirdn2d_kernel.cu:107-207 (== upfirdn2d_native, op/upfirdn2d.py:168-209).
 * force_path: -1 automatic; otherwise a path number from fmgan_upfirdn2d_select
 * (FMGAN_EUNSUPPORTED if that path cannot serve the arguments) — used by tests/bench.
 */
int fmgan_upfirdn2d(int dtype, const void *input, const void *kernel, void *out,
                    int major, int in_h, int in_w, int minor,
                    int kernel_h, int kernel_w,
                    int up_x, int up_y, int down_x, int down_y,
                    int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                    int force_path, void *stream);

/*
 * Same op on a row/plane-strided input (minor must be 1 when strides are not the contiguous ones):
 * element (n, y, x) of the input lives at input[n*in_plane_stride + y*in_row_stride + x] (strides in elements).
 * Used for the private conv_transpose -> blur intermediate of ModulatedConv2d(upsample) (stylegan2.py:276-279):
 * its rows are 2W+1 floats, never 16-byte aligned when contiguous; with row stride round_up(2W+2, 4) and a
 * one-float left offset every dwordx4 load of the blur is aligned (4.08 -> 4.8 TB/s on the 1024^2 layer).
 */
int fmgan_upfirdn2d_strided(int dtype, const void *input, const void *kernel, void *out,
                            int major, int in_h, int in_w, int minor,
                            long long in_plane_stride, int in_row_stride,
                            int kernel_h, int kernel_w,
                            int up_x, int up_y, int down_x, int down_y,
                            int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                            int force_path, void *stream);

/*
 * The tail of an upsampling StyledConv in one pass (stylegan2.py:279 + 371-373): 4x4 (or smaller) FIR blur of the
 * transposed-conv result, then noise + bias + leaky ReLU * scale applied to the filtered value before it is stored:
 *   out[b,c,y,x] = lrelu( (blur(in)[b,c,y,x] + noise_weight[0]*noise[b or 0,y,x]) + bias[c] ) * act_scale
 * in [batch*channels planes, in_h, in_w] with element strides as in fmgan_upfirdn2d_strided; out contiguous
 * [batch,channels,out_h,out_w]; noise [noise_batch (1|batch), out_h*out_w] or NULL; bias [channels] or NULL.
 * Served by the row-march kernels (out_w >= 64) and, for small planes (in_h*in_w <= 12288: the 4^2..32^2 upsampling
 * layers), by the plane-tile kernel; FMGAN_EUNSUPPORTED otherwise — the caller then runs fmgan_upfirdn2d(_strided)
 * followed by fmgan_noise_bias_act_f32, which gives bit-identical results (same kernel family, same roundings).
 */
int fmgan_blur_noise_bias_act_f32(const float *input, const float *kernel, float *out,
                                  int batch, int channels, int in_h, int in_w,
                                  long long in_plane_stride, int in_row_stride,
                                  int kernel_h, int kernel_w,
                                  int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                  const float *noise, const float *noise_weight, const float *bias,
                                  int noise_batch, float alpha, float act_scale, void *stream);
/* Which kernel serves a fused-blur call with these arguments (host logic, nothing is launched): 5 = LDS-DMA ring
 * (path 1b), 1 = register row-march (path 1), 2 = plane-tile, FMGAN_EUNSUPPORTED = none.  bench.py names the kernel of
 * its roofline object from this. */
int fmgan_blur_noise_bias_act_select(const float *input, const float *out, const float *noise,
                                     int batch, int channels, int in_h, int in_w,
                                     long long in_plane_stride, int in_row_stride,
                                     int kernel_h, int kernel_w,
                                     int pad_x0, int pad_x1, int pad_y0, int pad_y1);
/* The same with the row-march variant chosen by the caller (tests, measurements): force_path -1 / 1 automatic,
 * 4 = register row-march (path 1), 5 = LDS-DMA ring (path 1b; FMGAN_EUNSUPPORTED when its alignment rules do not hold).
 * Both variants produce the same bits.  fmgan_upfirdn2d(_strided) accept the same two values. */
int fmgan_blur_noise_bias_act_path_f32(const float *input, const float *kernel, float *out,
                                       int batch, int channels, int in_h, int in_w,
                                       long long in_plane_stride, int in_row_stride,
                                       int kernel_h, int kernel_w,
                                       int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                       const float *noise, const float *noise_weight, const float *bias,
                                       int noise_batch, float alpha, float act_scale, int force_path, void *stream);

/*
 * fused_bias_act: out[i] = act'(x[i] + bias[(i / step_b) % size_b]; refer[i]) * scale
 *   act*10+grad: 10,11 linear; 12 zero; 30 lrelu(x); 31 (refer>0 ? x : alpha*x); 32 zero
 *   (op/fused_bias_act_kernel.cu:36-45).  bias == NULL or size_b == 0: no bias;
 *   refer == NULL: refer treated as 0 (the reference passes an empty tensor).
 *   x, refer, out: `size_x` contiguous elements of `dtype`; bias: `size_b` elements.
 *   step_b = product of x.shape[2:] (op/fused_bias_act_kernel.cu:67-71).
 */
int fmgan_fused_bias_act(int dtype, const void *x, const void *bias, const void *refer,
                         void *out, long long size_x, int size_b, int step_b,
                         int act, int grad, float alpha, float scale, void *stream);

/*
 * StyledConv epilogue in one pass (stylegan2.py:360-376: NoiseInjection then FusedLeakyReLU):
 *   out[b,c,p] = lrelu(x[b,c,p] + noise_weight[0] * noise[b or 0, p] + bias[c]) * scale
 *   x/out [batch, channel, hw] f32; noise [noise_batch (1 or batch), hw] f32 or NULL;
 *   noise_weight device scalar f32 (NoiseInjection.weight, stylegan2.py:305) or NULL; bias [channel] or NULL.
 */
int fmgan_noise_bias_act_f32(const float *x, const float *noise, const float *noise_weight,
                             const float *bias, float *out,
                             int batch, int channel, int hw, int noise_batch,
                             float alpha, float scale, void *stream);

/*
 * FusedLeakyReLUFunctionBackward in one pass (op/fused_act.py:29-50: fused_bias_act(grad_output, empty, out, 3, 1, ...)
 * followed by grad_input.sum over batch and space for the bias):
 *   grad_in[p,i] = (ref_out[p,i] > 0 ? grad_out[p,i] : alpha * grad_out[p,i]) * scale     (bits of fmgan_fused_bias_act 3/1)
 *   partial[p, blk] = sum of the block's share of grad_in[p, :]
 * over `planes` = batch*channels contiguous planes of `hw` f32 elements; partial [planes, fmgan_fused_bias_act_bwd_blocks]
 * — the caller sums it to [channels] (deterministic).  hw % 4 == 0, hw >= 64 and 16-byte aligned pointers, else
 * FMGAN_EUNSUPPORTED (fmgan_fused_bias_act_bwd_blocks returns 0): the caller then uses fmgan_fused_bias_act + a sum.
 */
int fmgan_fused_bias_act_bwd_blocks(long long planes, int hw);
int fmgan_fused_bias_act_bwd_f32(const float *grad_out, const float *ref_out, float *grad_in, float *partial,
                                 long long planes, int hw, float alpha, float scale, void *stream);

/*
 * Backward of PReLU(channels) on channels-innermost activations (the pSp encoder's units,
 * psp_encoder_model/encoders/helpers.py:107,130, run in NHWC): x/grad/grad_x [rows, channels] f32, slope [channels],
 *   grad_x = grad * (x > 0 ? 1 : slope[c]);   partial[blk, c] = sum over the block's rows of grad * x * (x <= 0)
 * partial [fmgan_prelu_backward_blocks(rows, channels), channels]: the caller sums it over dim 0 (deterministic).
 * channels % 4 == 0 and 16-byte aligned pointers, else FMGAN_EUNSUPPORTED.
 */
int fmgan_prelu_backward_blocks(long long rows, int channels);
int fmgan_prelu_backward_f32(const float *x, const float *grad, const float *slope, float *grad_x,
                             float *partial, long long rows, int channels, void *stream);

/*
 * Demodulation coefficients of ModulatedConv2d (stylegan2.py:258-262):
 *   demod[b,o] = rsqrt( sum_{i,k} (scale * weight[o,i,k] * style[b,i])^2 + eps )
 *   weight [cout, cin, ktaps] f32 (the [1,cout,cin,k,k] parameter), style [batch, cin] f32,
 *   demod [batch, cout] f32.  One wave per output channel; the sum over cin is a wave-shuffle reduction.
 */
int fmgan_modconv_demod_f32(const float *weight, const float *style, float *demod,
                            int batch, int cout, int cin, int ktaps,
                            float scale, float eps, void *stream);

/*
 * Winograd F(2x2,3x3) form of the plain 3x3 modulated conv (mode 0 of fmgan_modconv2d_f32; /root/reference/stylegan2.py:250-298):
 * 16 products per 2x2 output tile instead of 36.  Three steps; the middle one is a plain batched GEMM done by the caller
 * (M[xi] = U[xi] @ V[xi], xi = 0..15).  H and W even; T = (H/2)*(W/2) tiles per sample; fp32; results differ from the direct
 * kernel by fp32 rounding of re-associated sums.
 *   fmgan_wino_weight_f32:  wt [cin,9,cout] (fmgan_modconv_weight_prep_f32, kind 0) -> U [16, cout, cin]
 *   fmgan_wino_input_f32:   x [B,C,H,W], style [B,C] -> V [16, C, B*T]   (column b*T + ty*(W/2) + tx; modulated, zero padded)
 *   fmgan_wino_output_f32:  M [16, cout, B*T] -> out [B,cout,H,W] contiguous = epilogue(demod * At M A); demod / noise /
 *                           bias may be NULL; epilogue as fmgan_modconv2d_f32 with fuse_act
 */
int fmgan_wino_weight_f32(const float *wt, float *u, int cin, int cout, void *stream);
int fmgan_wino_input_f32(const float *x, const float *style, float *v, int batch, int c, int h, int w, void *stream);
int fmgan_wino_output_f32(const float *m, const float *demod, const float *noise, const float *noise_weight,
                          const float *bias, float *out, int batch, int cout, int h, int w, int noise_batch,
                          int fuse_act, float alpha, float act_scale, void *stream);

/*
 * EqualLinear at inference batch sizes (replaces F.linear(input, weight*scale, bias=bias*lr_mul) of
 * /root/reference/stylegan2.py:146-180 where it is a ModulatedConv2d's modulation, stylegan2.py:226):
 *   out[b,n] = sum_k x[b,k] * weight[n,k] (+ bias[n])      x [batch,k_in], weight [n_out,k_in] (already scaled), bias [n_out] or NULL
 * One wave per (n, b), fixed summation order: bit-reproducible and independent of the batch size.
 */
int fmgan_equal_linear_f32(const float *x, const float *weight, const float *bias, float *out,
                           int batch, int n_out, int k_in, void *stream);

/*
 * The same coefficients in two steps, for weights that change rarely (inference, or once per optimiser step):
 *   fmgan_modconv_wsq_f32:        wsq[o,i] = sum_k weight[o,i,k]^2          [cout, cin] f32, cached by the caller
 *   fmgan_modconv_demod_wsq_f32:  demod[b,o] = rsqrt(scale^2 * sum_i wsq[o,i] * style[b,i]^2 + eps)
 * Same fma chains in the same order as fmgan_modconv_demod_f32: bit-identical results; the per-forward kernel reads
 * cout*cin floats instead of cout*cin*ktaps.
 */
int fmgan_modconv_wsq_f32(const float *weight, float *wsq, int cout, int cin, int ktaps, void *stream);
int fmgan_modconv_demod_wsq_f32(const float *wsq, const float *style, float *demod,
                                int batch, int cout, int cin, float scale, float eps, void *stream);

/*
 * Weight layouts for the MFMA contraction (last index contiguous, so a 32-lane MFMA A-operand read is one LDS bank
 * row and staging is coalesced).  weight [cout, cin, ktaps] f32; wt pre-allocated, cout*cin*ktaps floats:
 *   kind 0  wt[i][tap][o] = scale * weight[o][i][tap]            forward (modes 0, 1, and the downsample branch, mode 2)
 *   kind 1  wt[o][tap][i] = scale * weight[o][i][ktaps-1-tap]    data-gradient of the plain conv    (run as mode 0)
 *   kind 2  wt[o][tap][i] = scale * weight[o][i][tap]            data-gradient of the transposed conv (run as mode 2)
 * Depends on the parameter only.  The host shim never caches it across forwards (in-place parameter updates
 * through `.data` give no invalidation signal): inference re-derives every layer's table with fmgan_weight_refresh_f32.
 */
int fmgan_modconv_weight_prep_f32(const float *weight, float *wt, int cout, int cin, int ktaps,
                                  float scale, int kind, void *stream);

/*
 * All derived weights of a network, re-derived from the live parameters in ONE launch per inference forward
 * (replaces the per-call `weight * scale` / `bias * lr_mul` of every EqualLinear, stylegan2.py:165-175, and the
 * per-call weight preparation of every ModulatedConv2d, stylegan2.py:257-262; nothing is cached across forwards, so
 * the reference's in-place EMA update `accumulate`, train_3_encoder.py:195-200, needs no invalidation hook).
 *   table_dev: DEVICE array of n_entries entries, sorted by block_begin (entry k owns logical blocks
 *              [block_begin_k, block_begin_k + fmgan_weight_refresh_blocks(entry k)) ); total_blocks = their sum.
 *   kind 0: dst[k] = src[k] * scale, k < n
 *   kind 1: dst = wt[i][tap][o] = scale * src[o][i][tap] (layout kind 0 of fmgan_modconv_weight_prep_f32), and, if
 *           dst2 != NULL, dst2 = wsq[o][i] = sum_tap src[o][i][tap]^2 (fmgan_modconv_wsq_f32); ktaps <= 9.
 * Bit-identical to the separate kernels (same operations in the same order).
 */
typedef struct fmgan_refresh_entry {
  const void *src;
  void *dst;
  void *dst2;
  long long n;            /* kind 0: element count */
  int kind, cout, cin, ktaps;
  float scale;
  unsigned block_begin;
} fmgan_refresh_entry;
int fmgan_refresh_entry_bytes(void);      /* sizeof(fmgan_refresh_entry), for bindings that build the table by hand */
long long fmgan_weight_refresh_blocks(int kind, int cout, int cin, int ktaps, long long n);
int fmgan_weight_refresh_f32(const fmgan_refresh_entry *table_dev, int n_entries, long long total_blocks,
                             void *stream);

/*
 * Modulated 3x3 convolution, input-modulated form with batch-shared weights
 * (algebraically equal to the reference's per-sample weight-modulated grouped conv,
 * stylegan2.py:258-293):
 *   mode 0 (plain, stylegan2.py:289-293): out[b,o,y,x] = demod[b,o] * sum_{i,ky,kx}
 *             wt[i,ky*3+kx,o] * style[b,i] * in[b,i,y+ky-1,x+kx-1]                 out [b,o,h,w]
 *   mode 1 (transposed, stride 2, pad 0; stylegan2.py:268-277):                    out [b,o,2h+1,2w+1]
 *             out[b,o,Y,X] = demod[b,o] * sum_{i, 2y+ky=Y, 2x+kx=X} wt[i,ky*3+kx,o]*style[b,i]*in[b,i,y,x]
 *   mode 2 (stride 2, pad 0; the downsample branch stylegan2.py:281-286 after its blur, and the data-gradient of
 *             mode 1):  out[b,o,y,x] = demod[b,o] * sum_{i,ky,kx} wt[i,ky*3+kx,o]*style[b,i]*in[b,i,2y+ky,2x+kx]
 *                                                                                    out [b,o,(h-3)/2+1,(w-3)/2+1]
 *   in [batch,cin,h,w], wt from fmgan_modconv_weight_prep_f32 (ktaps = 9), style [batch,cin],
 *   demod [batch,cout] or NULL (no demodulation); out pre-allocated; all f32 contiguous.
 * Optional fused StyledConv epilogue (mode 0 only; stylegan2.py:371-373), enabled by fuse_act != 0:
 *   out = lrelu( (conv + noise_weight[0]*noise[b or 0,y,x]) + bias[o] ) * act_scale
 *   noise [noise_batch (1 or batch), h*w] or NULL, noise_weight device scalar or NULL, bias [cout] or NULL.
 * The contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation).
 * out_plane_stride / out_row_stride (elements; 0 = contiguous): element (b,o,y,x) is written to
 *   out[(b*cout+o)*out_plane_stride + y*out_row_stride + x] — see fmgan_upfirdn2d_strided.
 * workspace: tiny layers (4x4..32x32 at small batch) have too few output tiles to fill 256 CUs, so the
 *   input-channel loop is split over blocks (split-K) and the partial sums are combined by a finish kernel.
 *   Pass a device buffer of at least fmgan_modconv2d_workspace_bytes() bytes (0 = no split for this shape);
 *   with workspace == NULL the call still works, unsplit.  Results do not depend on timing or placement
 *   (no atomics: each partial slab is written once and summed in a fixed order).
 */
long long fmgan_modconv2d_workspace_bytes(int batch, int cin, int cout, int h, int w, int mode);
int fmgan_modconv2d_f32(const float *in, const float *wt, const float *style,
                        const float *demod, float *out,
                        int batch, int cin, int cout, int h, int w, int mode,
                        const float *noise, const float *noise_weight, const float *bias,
                        int noise_batch, int fuse_act, float alpha, float act_scale,
                        long long out_plane_stride, int out_row_stride,
                        void *workspace, long long workspace_bytes, void *stream);

/*
 * Reduced-precision form of fmgan_modconv2d_f32 for BASELINE config 5's bf16 leg (NOT the parity path; the reference has
 * no such path: its ModulatedConv2d is F.conv2d in the tensors' dtype, stylegan2.py:250-298, and its op kernels
 * dispatch float / double / half only, op/upfirdn2d_kernel.cu:311): the same operator with both MFMA operands rounded to
 * bf16 (RNE) — the scaled weight, and the modulated activation style[b,i]*in[b,i,..] on its way into LDS — and fp32
 * accumulation on v_mfma_f32_32x32x16_bf16; fp32 tensors in HBM on both sides, same arguments and epilogue.
 *   fmgan_modconv_weight_bf16_bytes / fmgan_modconv_weight_to_bf16: convert the fp32 MFMA layout wt[cin][taps][cout]
 *       (fmgan_modconv_weight_prep_f32, any kind; cin / cout = the CONV's input / output channels) into the bf16 operand
 *       image [cin/16][tap][2][cout padded to 32][8].
 *   fmgan_modconv2d_bf16_supported: 1 when the shape is served (cin % 16 == 0, cout % 32 == 0, position grid >= 32 wide
 *       and >= 4 high, 32-bit buffer ranges); otherwise fmgan_modconv2d_bf16 returns FMGAN_EUNSUPPORTED and the caller
 *       keeps the fp32 kernel.  mode 1 runs over the (h+1) x (w+1) quad grid; the quads m = h / n = w own only the last
 *       output row / column.
 */
long long fmgan_modconv_weight_bf16_bytes(int cin, int cout, int ktaps);
int fmgan_modconv_weight_to_bf16(const float *wt, void *wt_bf16, int cin, int cout, int ktaps, void *stream);
int fmgan_modconv2d_bf16_supported(int batch, int cin, int cout, int h, int w, int mode);
int fmgan_modconv2d_bf16(const float *in, const void *wt_bf16, const float *style, const float *demod, float *out,
                         int batch, int cin, int cout, int h, int w, int mode,
                         const float *noise, const float *noise_weight, const float *bias,
                         int noise_batch, int fuse_act, float alpha, float act_scale,
                         long long out_plane_stride, int out_row_stride, void *stream);

/*
 * fp32 contraction on the bf16 matrix pipe by operand splitting ("bf16x3"; forward modes 0 and 1; a LABELLED path, not
 * the parity path): each fp32 operand is split exactly into three bf16 pieces (hi + mid + lo) and the product is
 * accumulated in fp32 from the six largest piece products (ah*bh + ah*bm + am*bh + am*bm + ah*bl + al*bh; the dropped
 * terms are O(2^-24) of the product, one fp32 rounding), on v_mfma_f32_32x32x16_bf16 — 16/6 = 2.7x the matrix-pipe rate of
 * v_mfma_f32_32x32x2_f32 at fp32 accuracy.  Same arguments and epilogue as fmgan_modconv2d_f32; wt_split from
 * fmgan_modconv_weight_to_bf16x3 (the fp32 MFMA layout split into three bf16 operand images).
 */
long long fmgan_modconv_weight_bf16x3_bytes(int cin, int cout, int ktaps);
int fmgan_modconv_weight_to_bf16x3(const float *wt, void *wt_split, int cin, int cout, int ktaps, void *stream);
int fmgan_modconv2d_bf16x3_supported(int batch, int cin, int cout, int h, int w, int mode);
int fmgan_modconv2d_bf16x3(const float *in, const void *wt_split, const float *style, const float *demod, float *out,
                           int batch, int cin, int cout, int h, int w, int mode,
                           const float *noise, const float *noise_weight, const float *bias,
                           int noise_batch, int fuse_act, float alpha, float act_scale,
                           long long out_plane_stride, int out_row_stride, void *stream);

/*
 * Plain (mode 0) modulated conv with the FOLLOWING ToRGB layer folded into its epilogue.  In the reference these are
 * two modules and three HBM passes over the activation (StyledConv conv2 -> ToRGB, stylegan2.py:646-651 calling
 * :360-376 and :393-404).  When one block holds every output channel of its pixels (cout <= the tile's channel
 * extent and every tile lies in one sample: fmgan_modconv2d_rgb_fusable() == 1; the launch runs without split-K), the 1x1 modulated conv to <= 3 channels is a reduction
 * over the accumulator rows:
 *   act[b,o,p]  = lrelu(demod*conv + noise_weight*noise + bias) * act_scale          (as fmgan_modconv2d_f32)
 *   rgb[b,c,p]  = sum_o act[b,o,p] * wmod[b,c,o] + rgb_bias[c] + rgb_skip[b,c,p]
 *   wmod[b,c,o] = rgb_scale * rgb_weight[c,o] * rgb_style[b,o]   ([batch, 3, cout], rows >= rgb_channels zero) is
 *   built once per forward by fmgan_torgb_weight_mod_f32 from the [1,3,cout,1,1] ToRGB parameter and its style;
 *   rgb_bias [rgb_channels] or NULL, rgb_skip [batch, rgb_channels, h, w] (the already upsampled skip) or NULL,
 *   rgb_out like rgb_skip.
 * out may be NULL: the activation is then never written (last layer of the synthesis network — ToRGB is its only
 * consumer).  Summation order over o is fixed (rows of a lane, lane halves, then waves): run-to-run bit-identical.
 * Returns FMGAN_EUNSUPPORTED when the shape is not fusable; the caller then runs fmgan_modconv2d_f32 + fmgan_torgb_f32.
 */
int fmgan_torgb_weight_mod_f32(const float *rgb_weight, const float *rgb_style, float *wmod,
                               int batch, int cout, int rgb_channels, float rgb_scale, void *stream);
int fmgan_modconv2d_rgb_fusable(int batch, int cin, int cout, int h, int w);
int fmgan_modconv2d_rgb_f32(const float *in, const float *wt, const float *style,
                            const float *demod, float *out,
                            int batch, int cin, int cout, int h, int w,
                            const float *noise, const float *noise_weight, const float *bias,
                            int noise_batch, int fuse_act, float alpha, float act_scale,
                            const float *rgb_wmod, const float *rgb_bias,
                            const float *rgb_skip, float *rgb_out, int rgb_channels, void *stream);

/*
 * Weight gradient of the plain (mode 0) modulated conv on the MFMA units:
 *   gw[o,i,ky,kx] = scale * sum_{b,y,x} (demod[b,o] * go[b,o,y,x]) * (style[b,i] * x[b,i,y+ky-1,x+kx-1])
 * (the conv part of d loss / d weight; the demodulation chain-rule term is [B,Cout]x[Cout,Cin] algebra on the host).
 *   go [batch,cout,h,w], demod [batch,cout] or NULL, x [batch,cin,h,w], style [batch,cin], gw [cout,cin,3,3].
 *   workspace: fmgan_modconv_wgrad_workspace_bytes() bytes (partial slabs of the split over pixels; fixed-order
 *   finish, bit-reproducible).  w < 16 returns FMGAN_EUNSUPPORTED (0 workspace bytes): tiny layers stay on MIOpen.
 */
long long fmgan_modconv_wgrad_workspace_bytes(int batch, int cin, int cout, int h, int w);
/*
 * The same for every mode of fmgan_modconv2d_f32 (h, w = the conv INPUT's size; go has that mode's output size):
 *   mode 1 (transposed, go [batch,cout,2h+1,2w+1]):  gw[o,i,ky,kx] = scale * sum (demod*go)[b,o,2y+ky,2x+kx] * (style*x)[b,i,y,x]
 *   mode 2 (stride 2,   go [batch,cout,(h-3)/2+1,(w-3)/2+1]):  ... sum (demod*go)[b,o,y,x] * (style*x)[b,i,2y+ky,2x+kx]
 * 64 x 64 output tiles whose four waves walk the same 2 x TW pixels per step; an MFMA's pixel pair comes from the two
 * rows, so consecutive K-steps move one pixel along x and the 3 x 3 window of the shifted operand slides (SP new
 * LDS reads per tap row and step).  Served: >= 48 channels on both sides and >= 16 pixels per row of the unshifted
 * operand (mode 0 also the 32 x 32-tile kernel for narrower layers); otherwise FMGAN_EUNSUPPORTED / 0 bytes.
 */
long long fmgan_modconv_wgrad_mode_workspace_bytes(int batch, int cin, int cout, int h, int w, int mode);
int fmgan_modconv_wgrad_mode_f32(const float *go, const float *demod, const float *x, const float *style,
                                 float *gw, int batch, int cin, int cout, int h, int w, int mode, float scale,
                                 void *workspace, long long workspace_bytes, void *stream);
int fmgan_modconv_wgrad_f32(const float *go, const float *demod, const float *x, const float *style,
                            float *gw, int batch, int cin, int cout, int h, int w, float scale,
                            void *workspace, long long workspace_bytes, void *stream);

/*
 * ToRGB (stylegan2.py:389-404): 1x1 modulated conv without demodulation + bias + optional skip:
 *   out[b,c,p] = sum_i scale*weight[c,i]*style[b,i]*in[b,i,p] + bias[c] (+ skip[b,c,p])
 *   in [batch,cin,hw], weight [cout,cin], style [batch,cin], bias [cout] or NULL,
 *   skip [batch,cout,hw] or NULL (already upsampled), out [batch,cout,hw]; cout <= 4.
 */
int fmgan_torgb_f32(const float *in, const float *weight, const float *style,
                    const float *bias, const float *skip, float *out,
                    int batch, int cin, int cout, int hw, float scale, void *stream);

/*
 * Backward of ToRGB's modulated 1x1 conv in one pass over its input (the reference: autograd of a grouped F.conv2d,
 * stylegan2.py:268-286 called from :389-404):
 *   grad_x[b,i,p] = sum_c scale*weight[c,i]*style[b,i] * grad_out[b,c,p]
 *   m_partial[s,b,c,i] = sum over the pixels of split s of grad_out[b,c,p] * x[b,i,p]
 * x / grad_x [batch,cin,hw], grad_out [batch,cout,hw], weight [cout,cin], style [batch,cin], cout <= 4;
 * m_partial [fmgan_torgb_backward_splits(batch,cin,hw), batch, cout, cin].  The caller sums m_partial over s (fixed order:
 * bit-reproducible) to M[b,c,i] and forms grad_weight[c,i] = scale * sum_b style[b,i]*M[b,c,i],
 * grad_style[b,i] = scale * sum_c weight[c,i]*M[b,c,i]; grad_bias and grad_skip are sums / copies of grad_out.
 * hw % 4 == 0 and 16-byte aligned pointers, else FMGAN_EUNSUPPORTED (splits() returns 0): the caller differentiates the
 * composite instead.
 */
int fmgan_torgb_backward_splits(int batch, int cin, int hw);
int fmgan_torgb_backward_f32(const float *x, const float *grad_out, const float *weight, const float *style,
                             float *grad_x, float *m_partial, int batch, int cin, int cout, int hw, float scale,
                             void *stream);

/*
 * The steps either side of the path, 3-channel images:
 *   fmgan_images_to_tensor: in uint8 [batch,h,w,3] -> out f32 [batch,3,h,w] = ((in/255) - mean) / std
 *       == transforms.ToTensor() + Normalize(mean, std) (train_3_encoder.py:233-239; Resize: fmgan_resize_* below)
 *   fmgan_tensor_to_images: in f32 [batch,3,h,w] -> out uint8 [batch,h,w,3] = (uint8)((clip(in,-1,1) + cent) * factor)
 *       == tensor2im (Evaluation/visual_eval.py:24-38), for every image of the batch.
 */
/*
 * transforms.Resize(size) of the same pipeline (train_3_encoder.py:233-239) = PIL.Image.resize(BILINEAR), whose
 * arithmetic is Pillow's ImagingResample (third-party, pinned pillow=8.2.0: src/libImaging/Resample.c): per-axis tap
 * windows of the triangle filter stretched by max(scale,1), weights in 22-bit fixed point, a horizontal then a
 * vertical integer pass with a uint8 rounding in between.  Bit-exact with Pillow.
 *   fmgan_resize_output_size : torchvision's Resize(int) rule (shorter edge -> size, longer int(size*long/short),
 *                              unchanged if the shorter edge already equals size).
 *   fmgan_resize_plan_ints / fmgan_resize_plan : HOST functions; fill a caller-owned host buffer with the coefficient
 *       tables for (in_h,in_w)->(out_h,out_w): header[8] = {ksize_x, ksize_y, in_h, in_w, out_h, out_w, max input
 *       rows per 8-row output tile, 8}, then bounds_x[2*out_w], coeff_x[out_w*ksize_x], bounds_y[2*out_h],
 *       coeff_y[out_h*ksize_y].  The caller copies it to the device once per size pair.
 *   fmgan_resize_bilinear_u8 : in uint8 [batch,in_h,in_w,3] -> out_u8 uint8 [batch,out_h,out_w,3] (what PIL returns)
 *       and/or out_f32 f32 [batch,3,out_h,out_w] = ((v/255) - mean) / std (Resize + ToTensor + Normalize in one pass);
 *       either output may be NULL.  `plan` is the DEVICE copy of the table for exactly these sizes.
 */
int fmgan_resize_output_size(int h, int w, int size, int *out_h, int *out_w);
long long fmgan_resize_plan_ints(int in_h, int in_w, int out_h, int out_w);
int fmgan_resize_plan(int in_h, int in_w, int out_h, int out_w, int *plan, long long plan_ints);
int fmgan_resize_bilinear_u8(const unsigned char *in, const int *plan, unsigned char *out_u8, float *out_f32,
                             int batch, int in_h, int in_w, int out_h, int out_w,
                             float mean, float stdv, void *stream);

int fmgan_images_to_tensor(const unsigned char *in, float *out, int batch, int h, int w,
                           float mean, float stdv, void *stream);
int fmgan_tensor_to_images(const float *in, unsigned char *out, int batch, int h, int w,
                           float cent, float factor, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FMGAN_HIP_H */
