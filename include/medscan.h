/*
 * medscan.h -- C ABI of libmedscan.so (MI355X / gfx950 selective-scan hot path).
 *
 * This is the drop-in boundary for the reference's native extension `selective_scan_cuda`
 * (pybind module, /root/reference/CrossMamba/FusionMamba/selective_scan/selective_scan.cpp:494-497)
 * and for the eager tensor ops around it inside SS2D (/root/reference/MedMamba.py:386-424,466-483).
 *
 *   - plain pointers + sizes + element strides, no torch / ATen types;
 *   - caller owns every buffer (outputs pre-allocated; gradient accumulators that are reduced with
 *     atomics -- dA, dB, dC, dD, ddelta_bias -- must be ZEROED by the caller before the call, like the
 *     reference's `torch::zeros_like`, selective_scan.cpp:460-466);
 *   - all work is enqueued on the `stream` argument (a hipStream_t passed as void*); no host
 *     synchronisation, no allocation, no global state -> re-entrant per stream and graph-capturable;
 *   - return value: MS_OK (0) or a negative MsStatus; nothing is thrown across the ABI.
 *
 * dtype: all tensors are fp32 (the reference's kernels compute in fp32 for every I/O dtype,
 * selective_scan_fwd_kernel.cuh:147-160; SS2D hard-casts the scan operands to fp32, MedMamba.py:403-409).
 */
#ifndef MEDSCAN_H_
#define MEDSCAN_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MEDSCAN_ABI_VERSION 9

typedef enum MsStatus {
    MS_OK = 0,
    MS_ERR_NULL = -1,        /* a required pointer is NULL                                  */
    MS_ERR_SHAPE = -2,       /* non-positive size, dim % n_groups != 0, ...                  */
    MS_ERR_DSTATE = -3,      /* dstate > 256 (selective_scan.cpp:262) or not tileable        */
    MS_ERR_STRIDE = -4,      /* unsupported stride combination                               */
    MS_ERR_LAUNCH = -5,      /* hipLaunchKernel failed (hipGetLastError is left set)         */
    MS_ERR_UNSUPPORTED = -6  /* feature not built (complex A, constant B/C inside the kernel)*/
} MsStatus;

/* Sequence positions between two saved states of the forward recurrence (the reference uses 2048,
 * selective_scan.cpp:307; the value is private to the implementation as long as fwd and bwd agree). */
#define MS_SCAN_CHUNK 32

/*
 * Launch parameters of the selective scan; the field set of the reference's SSMParamsBase
 * (selective_scan.h:26-69) with int64 element strides, an explicit sequence stride and group stride for
 * every activation (so (B,D,L) and channel-last (B,L,D) tensors are both first-class) and no z/complex.
 *
 *   u, delta, out : logical shape (batch, dim, seqlen); channel d = g*(dim/n_groups) + dl of group g lives at
 *                   base + b*batch_stride + g*group_stride + dl*d_stride + pos*l_stride
 *                   (reference layout (B,D,L): group_stride = (dim/n_groups)*d_stride; a group_stride of 0 lets
 *                   every group read the SAME tensor -- the 4 scan directions of SS2D over one feature map)
 *   A             : (dim, dstate)                         fp32, real
 *   B, C          : (batch, n_groups, dstate, seqlen)     channel d uses group d / (dim / n_groups)
 *   D, delta_bias : (dim) contiguous, may be NULL
 *   x             : saved states, contiguous (batch, n_chunks, dstate, dim) with
 *                   n_chunks = ms_scan_n_chunks(seqlen); x[b,c,n,d] = h after position
 *                   min((c+1)*MS_SCAN_CHUNK, seqlen)-1.   May be NULL in forward (inference).
 *                   (reference: x (batch,dim,n_chunks,2*dstate), selective_scan.cpp:313; the layout is
 *                   private to the extension there too, selective_scan_interface.py:46 only reads the
 *                   last state, which here is x[:, n_chunks-1].)
 *   map_h, map_w  : 0, 0 = plain sequences (pos = l).  H, W > 0 (H*W == seqlen, n_groups % 4 == 0) = SS2D mode:
 *                   every tensor is indexed by the PIXEL p = h*W + w and group g scans the pixels in the order of
 *                   direction g % 4 of SS2D.forward_corev0 (MedMamba.py:393-395):
 *                     0: p = l      1: p = (l % H)*W + l / H      2: p = L-1-l      3: p = map1(L-1-l)
 *                   i.e. the cross-scan is folded into the kernel's addressing and the outputs come back in pixel
 *                   order, so the cross-merge (MedMamba.py:420-424,476) is a plain sum over the 4 groups.
 */
/* bits of MsScanParams.delta_softplus */
#define MS_SCAN_SOFTPLUS 1   /* delta' = softplus(delta + delta_bias) (selective_scan.h: delta_softplus) */
#define MS_SCAN_A_IS_LOG 2   /* `A` holds A_log; the kernels use A = -exp(A_log) (what SS2D computes before every call,
                                MedMamba.py:407) and the backward accumulates the gradient w.r.t. A_log in dA */

#define MS_SCAN_ACCUMULATE 4 /* forward: out += result; backward: du += , ddelta += (same thread reads and writes each element:
                                launches over disjoint state slices of one scan can add up in place) */
#define MS_SCAN_BC_MAP(dir) (((dir) + 1) << 4)
                             /* SS2D mode only: the B/C rows (and dB/dC) are addressed through the pixel order of direction
                                `dir` (0..3) for EVERY group, while u/delta/out keep their own group's order -- the SSD layout
                                of CNN_Mamba.py:506-519, where the four directions' B/C form one concatenated state vector.
                                Requires the scalar-decay form (A_dstate_stride == 0).
                                dir == 4 (forward only): ALL four directions in one launch -- dstate = 4 * nd (nd <= 16), state
                                slice j (states j*nd .. j*nd+nd-1) reads the SAME nd columns of B / C through direction j's
                                order; the saved states `x` are laid out slice-major, (4, batch, n_chunks, nd, dim), i.e. as
                                the four `x` tensors of four one-direction launches, which is what the backward launches
                                (one per direction) then read. */

#define MS_SCAN_DT_FUSED 128  /* SS2D mode only: `delta` is NOT read; the kernels form delta[l, d] = sum_r dt_x[l, r] * dt_w[d, r]
                                (the `dt_projs` einsum, MedMamba.py:400,403-405) while they stage a tile, then add delta_bias
                                and apply softplus as usual.  dt_x: the dt_rank leading columns of the x_proj rows, addressed
                                with B's batch / group / l strides (unit stride along r); dt_w (dim, dt_rank) contiguous
                                (= dt_projs_weight (4, D, R) flattened); dt_rank <= 32; d_state == 16; dim % 4 == 0.
                                FORWARD ONLY in this build: ms_selective_scan_bwd returns MS_ERR_UNSUPPORTED when the flag is
                                set (the training path materialises delta with ms_dtproj_fwd and back-propagates with
                                ms_dtproj_bwd); MsScanBwdParams.ddt_x / ddt_w are reserved for that backward (same addressing
                                as dB / (dim, dt_rank), accumulated) and are ignored today. */

#define MS_SCAN_DELTA_OUT 1024 /* with MS_SCAN_DT_FUSED (forward): `delta` is an OUTPUT -- the kernel stores delta' = softplus(dt_x . dt_w + bias) there
                                (same addressing as the input form), which is what the TRAINING path hands to the backward launch as
                                MS_SCAN_DELTA_ACTIVATED input: the Delta projection of MedMamba.py:400,403-405 then has no forward launch
                                and no pre-activation tensor at all. */
#define MS_SCAN_DELTA_ACTIVATED 512  /* with MS_SCAN_SOFTPLUS: `delta` already holds delta' = softplus(raw + delta_bias) (ms_dtproj_fwd_act writes
                                it: the projection kernel is bandwidth-bound, the scan kernels are issue-bound, so the ~13 instructions of
                                the activation per element sit better there).  Forward and backward use it as is; the backward still
                                returns the gradient w.r.t. the PRE-activation value (times softplus' = 1 - exp(-delta'), as always) and
                                ddelta_bias; delta_bias is not read. */
#define MS_SCAN_LATTICE 256   /* SS2D mode only: the four groups scan the four (row parity, column parity) sub-lattices of the
                                map_h x map_w map (both even) instead of the four full-resolution directions -- FusionMamba's
                                stride-2 `EfficientScan` / `EfficientMerge` (CrossMamba/FusionMamba/models/cross.py:139-190,
                                34-88) as an addressing mode of the scan kernels: seqlen = (map_h/2) * (map_w/2), and step l of
                                group k visits  k = 0: (row 2i,   col 2j)    row-major (i = l / (map_w/2), j = l % (map_w/2))
                                                k = 1: (row 2b+1, col 2a)    column-major (a = l / (map_h/2), b = l % (map_h/2))
                                                k = 2: (row 2i,   col 2j+1)  row-major
                                                k = 3: (row 2b+1, col 2a+1)  column-major.
                                Every pixel belongs to exactly one group, so one pixel-order tensor (group stride 0) can hold
                                all four groups' out / du / ddelta: the merge is free.  Odd sizes: the caller zero-pads the map to
                                even sizes (what cross.py:148-156 does to the sequences). */

typedef struct MsScanParams {
    int32_t batch, dim, seqlen, dstate, n_groups;
    int32_t delta_softplus;                 /* flags: MS_SCAN_SOFTPLUS | MS_SCAN_A_IS_LOG (historically a bool: 1 = softplus) */
    int32_t map_h, map_w;
    int64_t u_batch_stride, u_group_stride, u_d_stride, u_l_stride;
    int64_t delta_batch_stride, delta_group_stride, delta_d_stride, delta_l_stride;
    int64_t out_batch_stride, out_group_stride, out_d_stride, out_l_stride;
    int64_t A_d_stride, A_dstate_stride;
    int64_t B_batch_stride, B_group_stride, B_dstate_stride, B_l_stride;
    int64_t C_batch_stride, C_group_stride, C_dstate_stride, C_l_stride;
    const float *u, *delta, *A, *B, *C, *D, *delta_bias;
    float *out;
    float *x;
    const float *dt_x, *dt_w;               /* MS_SCAN_DT_FUSED operands (else ignored) */
    int32_t dt_rank;
    int32_t segments;                       /* >= 2 (forward, SS2D fast path, inference): scan the sequence in that many segments in
                                             * parallel -- B * 4 * D / 8 waves do not fill the chip at small batch.  `x` is then a WORKSPACE of
                                             * ms_scan_seg_floats(batch, dim, segments) floats instead of the saved states (two passes over
                                             * the recurrence + a carry over the segments; same result up to rounding).  0 / 1: off. */
} MsScanParams;

/*
 * Backward; field set of SSMParamsBwd (selective_scan.h:71-101).
 *   dout, du, ddelta : (batch, dim, seqlen) with their own batch/group/d/l strides (same addressing as u)
 *   dA (dim,dstate) contiguous; dB, dC (batch,n_groups,dstate,seqlen) with their own strides, fp32;
 *   dD, ddelta_bias (dim) or NULL.  dA/dB/dC/dD/ddelta_bias are ACCUMULATED INTO (atomics).
 *   x is required when seqlen > MS_SCAN_CHUNK.
 */
typedef struct MsScanBwdParams {
    MsScanParams f;                          /* forward operands; f.out is unused, f.x is read */
    int64_t dout_batch_stride, dout_group_stride, dout_d_stride, dout_l_stride;
    int64_t du_batch_stride, du_group_stride, du_d_stride, du_l_stride;
    int64_t ddelta_batch_stride, ddelta_group_stride, ddelta_d_stride, ddelta_l_stride;
    int64_t dB_batch_stride, dB_group_stride, dB_dstate_stride, dB_l_stride;
    int64_t dC_batch_stride, dC_group_stride, dC_dstate_stride, dC_l_stride;
    const float *dout;
    float *du, *ddelta, *dA, *dB, *dC, *dD, *ddelta_bias;
    float *ddt_x, *ddt_w;                   /* reserved: MS_SCAN_DT_FUSED gradients (ignored in this build) */
} MsScanBwdParams;

/* replaces selective_scan_cuda.fwd  (selective_scan.cpp:226-336 -> selective_scan_fwd_kernel.cuh:67-303) */
int ms_selective_scan_fwd(const MsScanParams *p, void *stream);
/* replaces selective_scan_cuda.bwd  (selective_scan.cpp:338-492 -> selective_scan_bwd_kernel.cuh:75-489) */
int ms_selective_scan_bwd(const MsScanBwdParams *p, void *stream);
/* number of saved states per row for a sequence length (host-side sizing of `x`) */
int ms_scan_n_chunks(int seqlen);
/* floats of the workspace `x` of a segmented forward (MsScanParams.segments >= 2; dstate 16) */
int64_t ms_scan_seg_floats(int batch, int dim, int segments);

/*
 * 4-direction cross-scan and cross-merge (MedMamba.py:393-395 and :420-424,476), fp32, contiguous.
 *   cross_scan : x (batch, dim, H, W)      -> xs (batch, 4, dim, H*W)
 *   cross_merge: ys (batch, 4, dim, H*W)   -> y  (batch, dim, H*W) = ((y0 + flip(y2)) + T(y1)) + T(flip(y3))
 * Each is the other's adjoint, so the same two entry points serve the backward pass.
 */
int ms_cross_scan(const float *x, float *xs, int batch, int dim, int H, int W, void *stream);
int ms_cross_merge(const float *ys, float *y, int batch, int dim, int H, int W, void *stream);
/* The same on channel-LAST tensors, as the SSD blocks use them (CNN_Mamba.py:494-519 gather, :542-552 inverse + adds):
 *   cross_scan_nhwc : pix (batch, H*W, *) fp32, first C columns of rows `pixel_stride` apart -> seq (batch, H*W, 4, C)
 *   cross_merge_nhwc: seq (batch, H*W, 4, C) -> pix[b, p, 0:C] = ((s0 + s2) + s1) + s3 at the steps that visit pixel p
 * (a column range of a wider tensor on the pixel side: pass the pointer of its first column). */
int ms_cross_scan_nhwc(const float *pix, int64_t pixel_stride, float *seq, int batch, int H, int W, int C, void *stream);
int ms_cross_merge_nhwc(const float *seq, float *pix, int64_t pixel_stride, int batch, int H, int W, int C, void *stream);

/*
 * Depthwise 3x3 conv (padding 1, stride 1) + bias + SiLU, NCHW fp32 contiguous
 * (nn.Conv2d(groups=C) followed by nn.SiLU, MedMamba.py:285-294,473).
 *   fwd: y = silu(conv(x, w) + bias)          x,y (batch, C, H, W); w (C,1,3,3); bias (C) or NULL
 *   bwd: given dy, recomputes the pre-activation; writes dx; ACCUMULATES dw (C,9) and dbias (C).
 */
int ms_dwconv3x3_silu_fwd(const float *x, const float *w, const float *bias, float *y,
                          int batch, int C, int H, int W, void *stream);
int ms_dwconv3x3_silu_bwd(const float *x, const float *w, const float *bias, const float *dy,
                          float *dx, float *dw, float *dbias, int batch, int C, int H, int W, void *stream);

/*
 * The same op on CHANNEL-LAST tensors (the layout in_proj produces; removes MedMamba.py:472's permute+copy):
 *   x : (batch, H, W, *) fp32 or bf16 (x_is_bf16), pixel stride `x_pixel_stride` elements, first C channels used
 *       (so the x half of xz = in_proj(x) is read in place);  y : (batch, H, W, C) contiguous fp32.
 *   bwd: the incoming gradient is the sum of `dy_ndir` contiguous (batch,H,W,C) fp32 slabs `dy_dir_stride` elements apart
 *       (the scan's four per-direction du tensors are consumed as they are) plus `dy_extra` (same shape, or NULL);
 *       dx : fp32 or bf16 with its own pixel stride (it can be written straight into the x half of the xz gradient);
 *       scratch : ms_dwconv3x3_silu_nhwc_bwd_scratch_floats(batch, C, H, W) floats of workspace (the pre-activation
 *       gradient + per-workgroup partial sums of dw/dbias; need not be initialised); dw (C,9) and dbias (C) are ACCUMULATED.
 */
int ms_dwconv3x3_silu_nhwc_fwd(const void *x, int x_is_bf16, const float *w, const float *bias, float *y,
                               int batch, int C, int H, int W, int64_t x_pixel_stride, void *stream);
int ms_dwconv3x3_silu_nhwc_bwd(const void *x, int x_is_bf16, const float *w, const float *bias, const float *dy, int dy_ndir,
                               int64_t dy_dir_stride, const float *dy_extra, void *dx, int dx_is_bf16, int64_t dx_pixel_stride,
                               float *scratch, float *dw, float *dbias, int batch, int C, int H, int W,
                               int64_t x_pixel_stride, void *stream);
int64_t ms_dwconv3x3_silu_nhwc_bwd_scratch_floats(int batch, int C, int H, int W);

/*
 * Fused tail of SS2D (MedMamba.py:476-479): cross-merge sum of the four directions + LayerNorm(D) + SiLU gate.
 *   y4   : 4 scan outputs in pixel order, y_k at y4 + k*dir_stride, each (npix, D) contiguous fp32
 *   z    : gate, (npix, *) fp32 or bf16 (z_is_bf16) with pixel stride z_pixel_stride (reads the z half of xz in place)
 *   out  = (LN(((y0+y2)+y1)+y3) * gamma + beta) * silu(z)      (npix, D) contiguous, fp32 or bf16 (out_is_bf16)
 * bwd: dout (npix, D) fp32|bf16 -> dy (npix, D) fp32 (the dout of all four scan directions), dz (npix, D) in z's
 *      dtype, and ACCUMULATES dgamma, dbeta (D).  y and the LN statistics are recomputed from y4.
 * ms_ln_gate_fwd_keep additionally stores the merged sum ((y0+y2)+y1)+y3 in ysum (npix, D) fp32; the backward accepts it in
 * place of the four slabs: dir_stride == 0 means `y4` IS that sum (4 B instead of 16 B read per element, and the four slabs
 * need not outlive the forward).
 */
int ms_ln_gate_fwd(const float *y4, int64_t dir_stride, const void *z, int z_is_bf16, int64_t z_pixel_stride,
                   const float *gamma, const float *beta, float eps, void *out, int out_is_bf16,
                   int64_t npix, int D, void *stream);
int ms_ln_gate_fwd_keep(const float *y4, int64_t dir_stride, const void *z, int z_is_bf16, int64_t z_pixel_stride,
                        const float *gamma, const float *beta, float eps, void *out, int out_is_bf16, float *ysum,
                        int64_t npix, int D, void *stream);
int ms_ln_gate_bwd(const float *y4, int64_t dir_stride, const void *z, int z_is_bf16, int64_t z_pixel_stride,
                   const float *gamma, const float *beta, float eps, const void *dout, int dout_is_bf16,
                   float *dy, void *dz, int64_t dz_pixel_stride, float *dgamma, float *dbeta, int64_t npix, int D,
                   void *stream);

/* ---- the two-branch block around SS2D (SS_Conv_SSM, MedMamba.py:502-538) ------------------------------------
 * ms_layernorm_fwd/bwd replace `self.ln_1(right)` on the right half of `input.chunk(2, dim=-1)` (MedMamba.py:512-515):
 *   x      fp32 rows of D channels, unit channel stride, `x_pixel_stride` floats between pixels (the half is read in
 *          place from the (.., 2D) block input: no contiguous copy);  out (npix, D) contiguous, fp32 or bf16.
 *   bwd    recomputes mean/rstd from x; dx (npix, D) fp32 contiguous; dgamma/dbeta are ACCUMULATED (zero them first).
 * ms_block_tail_fwd/bwd replace drop_path, `torch.cat((left, x), -1)`, `channel_shuffle(.., groups=2)` and
 * `+ input` (MedMamba.py:515,534-538 and 486-499):
 *   out[p, 2i] = left[p, i] + input[p, 2i] ;  out[p, 2i+1] = sample_scale[b(p)] * x[p, i] + input[p, 2i+1]
 *   left, x: (npix, C/2) contiguous, fp32 or bf16; input, out: (npix, C) fp32; sample_scale: per-sample DropPath factor
 *   (mask / keep_prob) or NULL; pixels_per_sample = H*W; C % 4 == 0.
 *   bwd: dleft[p, i] = dout[p, 2i], dx[p, i] = sample_scale[b] * dout[p, 2i+1]; the gradient w.r.t. `input` is dout itself. */
int ms_layernorm_fwd(const float *x, int64_t x_pixel_stride, const float *gamma, const float *beta, float eps, void *out,
                     int out_is_bf16, int64_t npix, int D, void *stream);
int ms_layernorm_bwd(const float *x, int64_t x_pixel_stride, const float *gamma, float eps, const void *dout, int dout_is_bf16,
                     float *dx, float *dgamma, float *dbeta, int64_t npix, int D, void *stream);
int ms_block_tail_fwd(const void *left, int left_is_bf16, const void *x, int x_is_bf16, const float *input,
                      const float *sample_scale, float *out, int64_t npix, int64_t pixels_per_sample, int C, void *stream);
int ms_block_tail_bwd(const float *dout, const float *sample_scale, void *dleft, int dleft_is_bf16, void *dx, int dx_is_bf16,
                      int64_t npix, int64_t pixels_per_sample, int C, void *stream);
/* The same with the ReLU that ends the conv branch (MedMamba.py:526) folded in: `left` (dtype of dleft) is that ReLU's OUTPUT,
 * dleft[p, i] = dout[p, 2i] * [left[p, i] > 0] -- the gradient w.r.t. the ReLU's INPUT, no threshold_backward pass. */
int ms_block_tail_bwd_relu(const float *dout, const float *sample_scale, const void *left, void *dleft, int dleft_is_bf16, void *dx,
                           int dx_is_bf16, int64_t npix, int64_t pixels_per_sample, int C, void *stream);
/* Gradient of the block input in one pass (autograd: a concat for `input.chunk(2, -1)`, MedMamba.py:531, plus an add with the
 * residual edge of :537):  dinput[p, :] = dout[p, :] + cat(dleft[p, :], dright[p, :]);  dout, dinput (npix, C) fp32, the halves
 * (npix, C/2) contiguous, bf16 or fp32; C % 8 == 0.  dinput may alias dout. */
int ms_block_head_bwd(const float *dout, const void *dleft, int dleft_is_bf16, const void *dright, int dright_is_bf16, float *dinput,
                      int64_t npix, int C, void *stream);

/* ---- training-mode BatchNorm2d (+ fused ReLU) of the conv branch (MedMamba.py:517-527, 533-535) ------------------
 * y, dy, dx : (npix, C) contiguous = the memory of an NCHW tensor in channels_last format; bf16 or fp32.
 * x : npix rows of C channels, unit channel stride, `x_pixel_stride` elements between pixels (C when contiguous; the left
 *     half of the (.., 2C) block input is normalised in place, without the copy `.contiguous()` would make).
 * fwd: batch statistics (biased variance, fp32), y = [relu]((x - mean) * rstd * gamma + beta), running statistics
 *      updated as torch does (unbiased variance, `momentum`), *num_batches_tracked += 1 when not NULL; save_mean /
 *      save_rstd (C) are written for the backward.
 * bwd: dgamma, dbeta (C) are WRITTEN; dx = gamma*rstd*(dy' - mean(dy') - xhat*mean(dy'*xhat)), dy' = dy * [y > 0] when relu.
 * `input_shift` (C) or NULL: the layer normalises x + input_shift without that add ever being made -- a per-channel
 *      constant cancels in (x - mean) and only moves the running mean.  It is the bias of the convolution in front of the
 *      layer (MedMamba.py:518-523): its add and its (identically zero) gradient reduction are skipped.
 * `scratch`: ms_bn_scratch_floats(C) floats of workspace (per-workgroup partial sums; need not be initialised). */
int ms_bn_relu_nhwc_fwd(const void *x, int x_is_bf16, int64_t x_pixel_stride, const float *input_shift, const float *gamma, const float *beta,
                        float *running_mean,
                        float *running_var, int64_t *num_batches_tracked, float momentum, float eps, int relu, void *y,
                        int y_is_bf16, float *save_mean, float *save_rstd, float *scratch, int64_t npix, int C,
                        void *stream);
int ms_bn_relu_nhwc_bwd(const void *x, int x_is_bf16, int64_t x_pixel_stride, const void *dy, int dy_is_bf16, const float *gamma, const float *beta,
                        const float *save_mean, const float *save_rstd, int relu, void *dx, int dx_is_bf16, float *dgamma, float *dbeta,
                        float *scratch, int64_t npix, int C, void *stream);
int ms_bn_scratch_floats(int C);

/* PatchMerging2D's tap gather + LayerNorm(4C) in one pass (MedMamba.py:196-205): x (batch, H, W, C) fp32 contiguous, H and W even,
 * C % 4 == 0, 4C <= 1024; token (b, h2, w2) = the channels of pixels (2h2, 2w2), (2h2+1, 2w2), (2h2, 2w2+1), (2h2+1, 2w2+1) in that
 * order; out (batch * H/2 * W/2, 4C) fp32 or bf16.  bwd: dx (batch, H, W, C) fp32 is WRITTEN (every input pixel belongs to exactly
 * one token: the scatter is the same map); dgamma / dbeta (4C) are ACCUMULATED. */
int ms_layernorm_taps_fwd(const float *x, const float *gamma, const float *beta, float eps, void *out, int out_is_bf16, int batch, int H,
                          int W, int C, void *stream);
int ms_layernorm_taps_bwd(const float *x, const float *gamma, float eps, const void *dout, int dout_is_bf16, float *dx, float *dgamma,
                          float *dbeta, int batch, int H, int W, int C, void *stream);

/* ---- delta projection of SS2D (the `dt_projs` einsum, MedMamba.py:400,403-405) -------------------------------------
 * proj   (npix, 4, row_width) fp32: the x_proj output rows [dts(R) | B(N) | C(N)] of every (pixel, direction)
 * Wdt    (4, D, R) fp32 = dt_projs_weight;  delta / ddelta (4, npix, D) fp32;  R <= 32
 * fwd:   delta[k, m, d] = sum_r proj[m, k, r] * Wdt[k, d, r]            (bias and softplus stay inside the scan kernel)
 * bwd:   dproj[m, k, r] (r < R) (+)= sum_d ddelta[k, m, d] * Wdt[k, d, r]   -- written in place into the projection gradient
 *        whose B|C columns ms_selective_scan_bwd fills; dproj must be zero-initialised (shared with that kernel's rule)
 *        dWdt[k, d, r] += sum_m ddelta[k, m, d] * proj[m, k, r]             -- accumulated (zero it first)
 *        `scratch`: ms_dtproj_bwd_scratch_floats(npix, D, R) floats of workspace (need not be initialised) in which the
 *        workgroups' dWdt partial sums are kept and then added up without atomics; NULL / too small: atomics into dWdt (slower,
 *        same result up to summation order).  Ranks 1..4 use register-tiled kernels, 5..32 scalar-operand kernels (D % 4 == 0). */
int ms_dtproj_fwd(const float *proj, const float *Wdt, float *delta, int64_t npix, int D, int R, int row_width, void *stream);
/* The same with the activation the scan applies to it folded in: delta'[k, m, d] = softplus(delta[k, m, d] + dt_bias[k, d])
 * (`dt_projs_bias`, MedMamba.py:403-405), for scans run with MS_SCAN_SOFTPLUS | MS_SCAN_DELTA_ACTIVATED. */
int ms_dtproj_fwd_act(const float *proj, const float *Wdt, const float *dt_bias, float *delta, int64_t npix, int D, int R, int row_width,
                      void *stream);
int ms_dtproj_bwd(const float *ddelta, const float *proj, const float *Wdt, float *dproj, float *dWdt, float *scratch,
                  int64_t scratch_floats, int64_t npix, int D, int R, int row_width, void *stream);
int64_t ms_dtproj_bwd_scratch_floats(int64_t npix, int D, int R);

/* ---- gated RMS normalisation of the SSD blocks (mamba_ssm 2.2.2 `RMSNormGated`, norm_before_gate=False, one group, as
 * constructed at CNN_Mamba.py:430-431 and applied at :554-555), optionally with the cross-merge sum (:542-552) in front:
 *   y    : ndir (1 or 4) slabs of (npix, D) fp32, `dir_stride` elements apart; merged = y0 or ((y0 + y2) + y1) + y3
 *   z    : gate, (npix, *) fp32 or bf16 with pixel stride z_pixel_stride;   weight (D);   D <= 1024
 *   fwd  : out = g * rsqrt(mean_D(g^2) + eps) * weight,  g = merged * silu(z);  out (npix, D) fp32 or bf16
 *   bwd  : dy (npix, D) fp32 = gradient of `merged` (the same for every slab), dz (like z, own pixel stride), dweight (D)
 *          ACCUMULATED (zero-fill first). */
int ms_rms_gate_fwd(const float *y, int64_t dir_stride, int ndir, const void *z, int z_is_bf16, int64_t z_pixel_stride,
                    const float *weight, float eps, void *out, int out_is_bf16, int64_t npix, int D, void *stream);
int ms_rms_gate_bwd(const float *y, int64_t dir_stride, int ndir, const void *z, int z_is_bf16, int64_t z_pixel_stride,
                    const float *weight, float eps, const void *dout, int dout_is_bf16, float *dy, void *dz,
                    int64_t dz_pixel_stride, float *dweight, int64_t npix, int D, void *stream);

/* ---- chunked (state-space-duality) SSD evaluation: the carry of chunk states (cnn_mamba._ssd_chunked) -----------------
 * What mamba_ssm 2.2.2's `_state_passing_fwd/_bwd` Triton kernels do behind `mamba_chunk_scan_combined` (the call at
 * CNN_Mamba.py:523-537; the dependency is outside the reference tree).
 *   in, out : (batch, chunks, groups, dstate, heads_per_group, headdim) fp32 contiguous; decay : (batch, chunks, heads),
 *             heads = groups * heads_per_group, the total decay exp(sum dt*A) of each chunk; headdim % 4 == 0.
 *   reverse = 0 : out[z] = state entering chunk z    = decay[z-1] * out[z-1] + in[z-1],  out[0] = 0
 *   reverse = 1 : out[z] = gradient reaching in[z]   = decay[z+1] * out[z+1] + in[z+1],  out[last] = 0   (the adjoint);
 *                 with `fwd_out` (the forward result) and `ddecay` (batch, chunks, heads; zero-filled by the caller) both
 *                 non-NULL, also ddecay[z] += < out[z], fwd_out[z] > over the head's states (the gradient of `decay`). */
int ms_ssd_chunk_carry(const float *in, const float *decay, float *out, const float *fwd_out, float *ddecay, int batch, int chunks,
                       int groups, int dstate, int heads_per_group, int headdim, int reverse, void *stream);

/* The chunked SSD evaluation itself on the matrix cores (csrc/ssd_chunk.hip; exact-fp32 MFMA, 64-position chunks): what
 * `mamba_chunk_scan_combined(x, dt, A, B, C, chunk_size, D, z=None, dt_bias, dt_softplus)` computes at CNN_Mamba.py:523-537 /
 * CrossMamba_fusion_2b2.py:327,590.  Operands fp32 contiguous: x, y (batch, L, heads, 64); dt (batch, L, heads) raw; A, dt_bias (heads);
 * B, C (batch, L, dstate) (one group); dstate % 64 == 0; D (heads) or (heads, 64) or NULL.  Workspaces the caller owns
 * (nc = ceil(L / 64)): dtv, cum (batch, nc, heads, 64); decay (batch, nc, heads); CB (batch, nc, 64, 64); S (batch, nc, dstate, heads * 64).
 *   ms_ssd_chunk_fwd      : dtv = softplus?(dt + bias), cum = in-chunk prefix sums of dtv * A, decay = exp(cum_last), CB = C B^T per chunk,
 *                           S = every chunk's state contribution, y = the inside-the-chunk output
 *   (caller)              : S_in = ms_ssd_chunk_carry(S, decay)
 *   ms_ssd_chunk_fwd_off  : y += diag(exp(cum)) C S_in + D x */
int ms_ssd_chunk_fwd(const float *x, const float *dt, const float *A, const float *B, const float *C, const float *dt_bias, int dt_softplus,
                     float *dtv, float *cum, float *decay, float *CB, float *S, float *y, int batch, int L, int heads, int headdim,
                     int dstate, void *stream);
int ms_ssd_chunk_fwd_off(const float *x, const float *C, const float *S_in, const float *cum, const float *D, int d_has_hdim, float *y,
                         int batch, int L, int heads, int headdim, int dstate, void *stream);

/* Backward of the chunked SSD evaluation (csrc/ssd_chunk.hip).  With the forward's workspaces (dtv, cum, decay, CB, S_in):
 *   ms_ssd_chunk_bwd_off : dS_in (batch, nc, dstate, heads * 64) = gradient of the entering states, dcum_off (batch, nc, heads, 64)
 *   (caller)             : dS, ddecay = ms_ssd_chunk_carry(dS_in, decay, reverse = 1, fwd_out = S_in)     (ddecay zero-filled first)
 *   ms_ssd_chunk_bwd     : dx (batch, L, heads, 64), ddt (batch, L, heads) w.r.t. the RAW dt, dB, dC (batch, L, dstate) WRITTEN;
 *                          dA, dbias (heads), dD (heads) | (heads, 64), dCB (batch, nc, 64, 64) ACCUMULATED (zero them first; dCB is a
 *                          workspace: the heads' sum of the masked score gradient).  dbias / dD / D may be NULL. */
int ms_ssd_chunk_bwd_off(const float *dy, const float *C, const float *S_in, const float *cum, float *dS_in, float *dcum_off, int batch, int L,
                         int heads, int headdim, int dstate, void *stream);
int ms_ssd_chunk_bwd(const float *x, const float *dy, const float *B, const float *C, const float *CB, const float *S_in, const float *dS,
                     const float *dtv, const float *cum, const float *decay, const float *ddecay, const float *A, const float *D,
                     int d_has_hdim, int dt_softplus, const float *dcum_off, float *dx, float *ddt, float *dA, float *dbias, float *dD,
                     float *dCB, float *dB, float *dC, int batch, int L, int heads, int headdim, int dstate, void *stream);

/* ---- the dense projections of SS2D on the matrix cores (in_proj, x_proj, out_proj: MedMamba.py:284,326,397,469,480) --------
 * C[i][j] (+)= sum_k Aop[i][k] * Bop[j][k]   for i < M, j < N, k < K; bf16 MFMA, fp32 accumulation
 *   Aop[i][k] = a_trans ? A[k*lda + i] : A[i*lda + k]      A, B: bf16 or fp32 in memory (`*_is_f32`; fp32 operands are rounded to
 *   Bop[j][k] = b_trans ? B[k*ldb + j] : B[j*ldb + k]      bf16 while a tile is staged -- no separate cast pass, no bf16 weight copy)
 *   c_mode 0: C fp32 written; 1: C bf16 written; 2: C fp32 ACCUMULATED with atomics (zero it first); 3: like 2 into the
 *   TRANSPOSED output, C[j*ldc + i] += ... (so the taller of the two weight-gradient orientations can be the row side);
 *   2 or 3 are required when k_splits > 1 (the k range is cut into k_splits slices evaluated by different workgroups: the weight gradient
 *   dW = dy^T x reduces over all B*H*W tokens into an output of a few tiles).
 *   forward  y = x W^T : A = x, B = W;   dx = dy W : A = dy, B = W with b_trans;   dW = dy^T x : A = dy, B = x, both *_trans.
 * Built combinations: (a_trans, b_trans) in {(0,0), (0,1)} with c_mode 0 / 1 (any dtypes); (1,1) with every c_mode and any dtypes.
 * lda / ldb in elements, multiples of 8 (bf16) / 4 (fp32); A, B 16-byte aligned; ldc in elements. */
int ms_gemm_bf16(const void *A, int a_is_f32, int a_trans, int64_t lda, const void *B, int b_is_f32, int b_trans, int64_t ldb,
                 void *C, int c_mode, int64_t ldc, int M, int N, int K, int k_splits, void *stream);

/* The same product with the epilogue of a 1x1 convolution (the `nn.Conv2d(dim/2, dim/2, 1)` + `nn.ReLU()` that end the conv
 * branch, MedMamba.py:525-526, on channels_last activations = rows of C channels): C = [relu](A B^T + bias[column]); c_mode 0 or 1
 * only, no split-K.  bias (N) fp32 or NULL. */
int ms_gemm_bf16_bias_act(const void *A, int a_is_f32, int a_trans, int64_t lda, const void *B, int b_is_f32, int b_trans, int64_t ldb,
                          void *C, int c_mode, int64_t ldc, int M, int N, int K, const float *bias, int relu, void *stream);
/* Weight AND bias gradient of that 1x1 convolution in one launch (its autograd, MedMamba.py:525): dW (N, K) += dy^T x and
 * dbias (N) += column sums of dy, for dy (M, N), x (M, K) row-major (lddy, ldx elements between rows); split-K inside the kernel
 * (k_splits slices of the M tokens), fp32 atomics: zero dW and dbias first.  The column sums ride on the matrix cores (one MFMA per
 * dy fragment against an all-ones fragment) instead of a reduction pass of their own over dy. */
int ms_gemm_bf16_wgrad_bias(const void *dy, int dy_is_f32, int64_t lddy, const void *x, int x_is_f32, int64_t ldx, float *dW, int64_t lddw,
                            float *dbias, int N, int K, int M, int k_splits, void *stream);

/* The same products in the reference's own precision (it trains in fp32, train.py:57-77): `nn.Linear` / einsum of
 * MedMamba.py:284,326,397,469,480 and their autograd on the exact-fp32 matrix instruction (v_mfma_f32_16x16x4_f32: fp32 products,
 * fp32 accumulation) -- no hipBLASLt on the fp32 path.  A, B, C fp32; meaning of a_trans / b_trans / c_mode / k_splits as in
 * ms_gemm_bf16 (c_mode 0, 2 or 3).  Built: (a_trans, b_trans) = (0,0) and (0,1) with c_mode 0 (+ bias / relu epilogue), (1,1) with
 * c_mode 2 / 3 (zero C first).  lda / ldb multiples of 4, A and B 16-byte aligned. */
int ms_gemm_f32(const float *A, int a_trans, int64_t lda, const float *B, int b_trans, int64_t ldb, float *C, int c_mode, int64_t ldc,
                int M, int N, int K, int k_splits, const float *bias, int relu, void *stream);

/* ---- bf16 working copies of the fp32 master weights, all in one launch ---------------------------------------------------
 * What torch.autocast does with one `_to_copy` launch per weight per forward (plus a layout copy per convolution weight on the
 * NHWC path) around MedMamba.py:517-527 and :284,326.  `desc`: n_tensors descriptors IN DEVICE MEMORY.
 *   taps <= 1 : dst[e] = bf16(src[e]), e < n
 *   taps  > 1 : a (O, inner, kh, kw) convolution weight, taps = kh*kw, written in channels_last memory order (O, kh, kw, inner):
 *               dst[(o*taps + k)*inner + i] = bf16(src[(o*inner + i)*taps + k]),  n = O*inner*taps
 *   taps  < 0 : the same weight for the INPUT-GRADIENT convolution, |taps| spatially flipped and in / out channels swapped, memory
 *               order (inner, kh, kw, O):  dst[(i*T + (T-1-k))*O + o] = bf16(src[(o*inner + i)*T + k]),  T = |taps| */
typedef struct MsCastDesc {
    const void *src;        /* fp32 */
    void *dst;              /* bf16 */
    int64_t n;
    int32_t inner, taps;
} MsCastDesc;
/* `blocks` (device memory): n_blocks pairs (tensor index, piece index), every piece of every tensor exactly once.  Pieces of a
 * tensor: plain casts and tap counts above MS_CAST_TILE_MAX_TAPS -- ceil(n / MS_CAST_CHUNK) runs of consecutive elements;
 * convolution weights with 2..MS_CAST_TILE_MAX_TAPS taps -- ceil(O / MS_CAST_TILE_O) * ceil(inner / MS_CAST_TILE_I) tiles, piece =
 * o_tile * ceil(inner / MS_CAST_TILE_I) + i_tile. */
#define MS_CAST_CHUNK 2048
#define MS_CAST_TILE_O 64
#define MS_CAST_TILE_I 16
#define MS_CAST_TILE_MAX_TAPS 9
int ms_cast_bf16_multi(const MsCastDesc *desc, const int32_t *blocks, int n_blocks, void *stream);

/* im2col of the patch embedding `nn.Conv2d(in_chans, embed_dim, kernel_size=4, stride=4)` (MedMamba.py:146-169): x (batch, C, H, W)
 * fp32 contiguous (16-byte aligned; H, W multiples of 4) -> out (batch * H/4 * W/4, C * 16) bf16 rows, row = patch (b, h/4, w/4),
 * column = c * 16 + i * 4 + j = the order of `weight.view(embed_dim, -1)`.  The convolution is then ms_gemm_bf16_bias_act on these
 * rows and its weight / bias gradient ms_gemm_bf16_wgrad_bias. */
int ms_patchify4_bf16(const float *x, void *out, int batch, int C, int H, int W, void *stream);

/* ---- `optimizer.step()` of the training loop: Adam over all parameters in one launch (/root/reference/train.py:62,76:
 * `optim.Adam(net.parameters(), lr=0.0001)`: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) ---------------------------
 *   desc   (device memory) one entry per parameter tensor: fp32 parameter, exp_avg, exp_avg_sq and the element count
 *   blocks (device memory) n_blocks pairs (tensor index, chunk index): workgroup b updates elements
 *          [chunk * MS_ADAM_CHUNK, min(n, (chunk + 1) * MS_ADAM_CHUNK)) of its tensor; every chunk of every tensor exactly once
 *   grads  (HOST memory) n_tensors <= MS_ADAM_MAX_TENSORS device pointers to the fp32 gradients (they are new tensors after every
 *          backward pass: they travel as kernel arguments, the rest of the table is built once)
 *   m += (1 - beta1)(g - m);  v = beta2 v + (1 - beta2) g g;  p -= step_size m / (sqrt(v) / bias_correction2_sqrt + eps)
 *   with step_size = lr / (1 - beta1^t), bias_correction2_sqrt = sqrt(1 - beta2^t) formed by the caller (torch.optim.Adam's rule). */
#define MS_ADAM_CHUNK 4096
#define MS_ADAM_MAX_TENSORS 448
typedef struct MsAdamDesc {
    float *p, *m, *v;
    int64_t n;
} MsAdamDesc;
int ms_adam_multi(const MsAdamDesc *desc, const int32_t *blocks, int n_blocks, const void *const *grads, int n_tensors,
                  float step_size, float bias_correction2_sqrt, float one_minus_beta1, float beta2, float one_minus_beta2, float eps,
                  void *stream);

/* ---- dense 3x3 convolution of the conv branch (`nn.Conv2d(dim/2, dim/2, 3, padding=1)`, MedMamba.py:518-523) ---------------------
 * stride 1, padding 1, groups 1, no bias; x (batch, H, W, Ci), y (batch, H, W, Co) bf16 channels_last memory; w bf16 in
 * (Co, 3, 3, Ci) memory order (= a channels_last copy of the (Co, Ci, 3, 3) weight, what ms_cast_bf16_multi writes); fp32
 * accumulation on the matrix cores.  Ci, Co multiples of 16.  The input gradient is the same call on dy with the flipped,
 * transposed weight w'[ci][8 - tap][co].  The kernels address with 32-bit byte offsets (raw buffer loads): an image (H * W * Ci elements) or
 * the weight (Co * 9 * Ci) of 2^30 elements or more returns MS_ERR_UNSUPPORTED; for the weight gradient the limit applies to the whole
 * activation tensor (batch * H * W * max(Ci, Co)). */
int ms_conv3x3_nhwc_bf16(const void *x, const void *w, void *y, int batch, int H, int W, int Ci, int Co, void *stream);
/* Its weight gradient: dW (Co, Ci, 3, 3) fp32 contiguous is WRITTEN = sum over pixels of dy[p, co] * x[p + tap, ci] (x, dy bf16
 * channels_last memory; Ci, Co multiples of 8).  `scratch`: ms_conv3x3_wgrad_scratch_floats(..) floats of workspace (the persistent
 * workgroups' partial blocks, summed by a second kernel -- no atomics, no zero-fill, no cast passes). */
int ms_conv3x3_wgrad(const void *x, const void *dy, float *dW, float *scratch, int64_t scratch_floats, int batch, int H, int W, int Ci,
                     int Co, void *stream);
int64_t ms_conv3x3_wgrad_scratch_floats(int batch, int H, int W, int Ci, int Co);

/* ---- the conv branch's inner BatchNorms folded into the convolutions around them (ABI v9) -------------------------------------------
 * `conv3x3 -> BatchNorm2d -> ReLU -> conv3x3 -> BatchNorm2d -> ReLU` of SS_Conv_SSM.conv33conv33conv11 (MedMamba.py:518-524) in TRAINING
 * mode: the statistics pass, the finalize launch and (for the inner BatchNorm) the apply pass of ms_bn_relu_nhwc_fwd disappear.
 * MsBnFold describes ONE BatchNorm whose batch statistics travel from the convolution that produces its input (the PRODUCER, bn_out) to
 * the kernel that consumes it (the CONSUMER: the next convolution's bn_in, or ms_bn_apply_sums_nhwc) as pivoted sums:
 *   sums : MS_BN_REPLICAS x 2 x C floats, sum (y - p) and sum (y - p)^2 per channel over all pixels, split over replica rows (same-address
 *          atomics serialise), followed by the C pivots p = running_mean - shift the producer used.  ZERO-FILLED by the caller before
 *          the producer runs (MS_BN_FOLD_FLOATS(C) floats).
 *   shift: the producing convolution's bias (may be NULL).  BN(conv + bias) == BN(conv) in training mode, so the bias never touches
 *          the activation; it enters running_mean only (its gradient is identically zero).
 *   the consumer writes save_mean / save_rstd (batch mean, 1 / sqrt(biased var + eps): what ms_bn_relu_nhwc_bwd takes), updates
 *   running_mean / running_var (momentum; unbiased variance) and num_batches_tracked (may be NULL) exactly once. */
#define MS_BN_REPLICAS 16
#define MS_BN_FOLD_FLOATS(C) ((MS_BN_REPLICAS * 2 + 1) * (C))
typedef struct MsBnFold {
    float *sums;
    const float *gamma, *beta, *shift;
    float *running_mean, *running_var;
    int64_t *num_batches_tracked;
    float *save_mean, *save_rstd;
    float momentum, eps;
} MsBnFold;
/* ms_conv3x3_nhwc_bf16 with a BatchNorm folded into either side (both may be NULL: the plain convolution):
 *   bn_in  : x is the PRE-BatchNorm activation (bf16); relu(bn_in(x)) is what is convolved, zero-padded AFTER the normalisation as the
 *            module sequence does; `xhat` (bf16, like x; may be NULL) receives it -- the conv input the weight gradient needs.  Ci <= 512.
 *   bn_out : the statistics of y (as stored, bf16-rounded) are accumulated into bn_out->sums. */
int ms_conv3x3_bn_nhwc_bf16(const void *x, const void *w, void *y, int batch, int H, int W, int Ci, int Co, const MsBnFold *bn_in, void *xhat,
                            const MsBnFold *bn_out, void *stream);
/* Backward side: the reduce pass of a BatchNorm's backward (dbeta = sum dy', dgamma = sum dy' * xhat, dy' = dy * [relu passed]) in the
 * epilogue of the 3x3 convolution's INPUT-GRADIENT launch that produces dy -- ms_conv3x3_nhwc_bf16(dy_conv, w', dx) with the sums of dx
 * against the BatchNorm's pre-normalisation input x_pre ((npix, C) rows, fp32 or bf16, pixel stride x_pre_pixel_stride: a multiple of 4
 * elements) accumulated into `sums` (MS_BN_REPLICAS x 2 x C floats, ZERO-FILLED by the caller).  ms_bn_bwd_apply_sums_nhwc then writes
 * dx_bn = gamma * rstd * (dy' - dbeta / n - xhat * dgamma / n) in one pass over (x_pre, dy) and the totals dgamma / dbeta (C each). */
typedef struct MsBnBwd {
    const void *x_pre;
    int x_pre_is_f32;
    int64_t x_pre_pixel_stride;
    const float *gamma, *beta, *save_mean, *save_rstd;
    int relu;
    float *sums;
} MsBnBwd;
int ms_conv3x3_bnbwd_nhwc_bf16(const void *dy, const void *w, void *dx, int batch, int H, int W, int Ci, int Co, const MsBnBwd *red, void *stream);
int ms_bn_bwd_apply_sums_nhwc(const MsBnBwd *bn, const void *dy, void *dx, int dx_is_bf16, float *dgamma, float *dbeta, int64_t npix, int C,
                              void *stream);
/* The same reduce on the epilogue of C = A B^T... in the input-gradient form of ms_gemm_bf16 (A (M, K) fp32 / bf16, B (K, N) in memory, C (M, N)
 * stored as fp32 (c_mode 0) or bf16 (1)): the 1x1 convolution behind the last BatchNorm of the conv branch (MedMamba.py:524-525).
 * bn->x_pre: bf16, (M, N) rows; N % 4 == 0. */
int ms_gemm_bf16_bnbwd(const void *A, int a_is_f32, int64_t lda, const void *B, int b_is_f32, int64_t ldb, void *C, int c_mode, int64_t ldc, int M,
                       int N, int K, const MsBnBwd *bn, void *stream);
/* The consumer for a BatchNorm that is not followed by a 3x3 convolution: y = [relu](bn(x)) from the producer's sums, one pass over
 * x (npix, C) bf16 -> y (npix, C) bf16; also writes save_mean / save_rstd and updates the running statistics. */
int ms_bn_apply_sums_nhwc(const void *x, const MsBnFold *bn, int relu, void *y, int64_t npix, int C, void *stream);

/* ---- both backward products of a token projection from one pass over its operands (ABI v9) ---------------------------------------------
 * The autograd of `y = x W^T` (in_proj / x_proj / out_proj, MedMamba.py:284,326,397,469,480) for huge token counts and small widths:
 *   dx (M, K) = dy (M, N) W (N, K)   written (fp32 or bf16, row stride ld_dx);   dW (N, K) fp32 contiguous += dy^T x   (ZERO-FILLED by the
 * caller; fp32 atomics).  dy, x: fp32 or bf16 rows (strides ld_dy, ld_x: whole 16-byte pieces, 16-byte aligned bases); W: (N, K)
 * contiguous, fp32 or bf16.  bf16 MFMA with fp32 accumulation -- the arithmetic of two ms_gemm_bf16 calls -- but dy and x are read
 * once.  ms_linear_bwd_ok(N, K) tells whether the pair of widths is built (K a multiple of 16; see csrc/linear_bwd.hip);
 * otherwise MS_ERR_UNSUPPORTED and the caller uses the two-launch form. */
int ms_linear_bwd_ok(int N, int K);
int ms_linear_bwd_bf16(const void *dy, int dy_is_f32, int64_t ld_dy, const void *x, int x_is_f32, int64_t ld_x, const void *w, int w_is_f32,
                       void *dx, int dx_is_bf16, int64_t ld_dx, float *dW, int M, int N, int K, void *stream);

/* Diagnostic: force the workgroup tile of ms_gemm_bf16 (rows 64 / 128, columns 64 / 128 / 192; 0 = the built-in heuristic). */
int ms_debug_gemm_tile(int block_rows, int block_cols);

/* Diagnostic: one workgroup busy for `cycles` (< 2^32) shader clocks on `stream` -- used to test whether two streams
 * execute concurrently (medmamba.set_branch_streams). */
int ms_spin(long long cycles, void *stream);

int ms_abi_version(void);
const char *ms_status_string(int status);

#ifdef __cplusplus
}
#endif
#endif /* MEDSCAN_H_ */
