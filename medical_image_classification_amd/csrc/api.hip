// extern "C" entry points of libmedscan.so (see include/medscan.h for the contract and the
// reference interfaces each one replaces).
#include <hip/hip_runtime.h>
#include "medscan.h"

namespace ms {
int scan_fwd_dispatch(const MsScanParams &p, hipStream_t stream);
int scan_bwd_dispatch(const MsScanBwdParams &q, hipStream_t stream);
int cross_scan_dispatch(const float *x, float *xs, int batch, int dim, int H, int W, hipStream_t s);
int cross_merge_dispatch(const float *ys, float *y, int batch, int dim, int H, int W, hipStream_t s);
int cross_scan_nhwc_dispatch(const float *pix, int64_t pps, float *seq, int batch, int H, int W, int C, hipStream_t s);
int cross_merge_nhwc_dispatch(const float *seq, float *pix, int64_t pps, int batch, int H, int W, int C, hipStream_t s);
int dwconv_fwd_dispatch(const float *x, const float *w, const float *bias, float *y,
                        int batch, int C, int H, int W, hipStream_t s);
int dwconv_bwd_dispatch(const float *x, const float *w, const float *bias, const float *dy, float *dx,
                        float *dw, float *dbias, int batch, int C, int H, int W, hipStream_t s);
int dwconv_nhwc_fwd_dispatch(const void *x, int x_is_bf16, const float *w, const float *bias, float *y,
                             int batch, int C, int H, int W, int64_t xps, hipStream_t s);
int64_t dwconv_nhwc_bwd_scratch_floats(int batch, int C, int H, int W);
int rms_gate_fwd_dispatch(const float *y4, int64_t sk, int ndir, const void *z, int z_bf16, int64_t zps, const float *w, float eps,
                          void *out, int out_bf16, int64_t npix, int D, hipStream_t s);
int rms_gate_bwd_dispatch(const float *y4, int64_t sk, int ndir, const void *z, int z_bf16, int64_t zps, const float *w, float eps,
                          const void *dout, int dout_bf16, float *dy, void *dz, int64_t dzps, float *dweight, int64_t npix, int D,
                          hipStream_t s);
int ssd_carry_dispatch(const float *in, const float *d, float *out, const float *fwd_out, float *ddecay, int batch, int chunks,
                       int groups, int N, int hg, int P, int reverse, hipStream_t s);
int dwconv_nhwc_bwd_dispatch(const void *x, int x_is_bf16, const float *w, const float *bias, const float *dy, int ndir,
                             int64_t dir_stride, const float *dy_extra, void *dx, int dx_bf16, int64_t dxps, float *scratch,
                             float *dw, float *dbias, int batch, int C, int H, int W, int64_t xps, hipStream_t s);
int ln_gate_fwd_dispatch(const float *y4, int64_t sk, const void *z, int z_bf16, int64_t zps, const float *gamma,
                         const float *beta, float eps, void *out, int out_bf16, float *ysum, int64_t npix, int D, hipStream_t s);
int ln_gate_bwd_dispatch(const float *y4, int64_t sk, const void *z, int z_bf16, int64_t zps, const float *gamma,
                         const float *beta, float eps, const void *dout, int dout_bf16, float *dy, void *dz, int64_t dzps,
                         float *dgamma, float *dbeta, int64_t npix, int D, hipStream_t s);
int block_tail_fwd_dispatch(const void *left, int left_is_bf16, const void *x, int x_is_bf16, const float *input,
                            const float *scale, float *out, int64_t npix, int64_t hw, int C, hipStream_t s);
int block_tail_bwd_dispatch(const float *dout, const float *scale, const void *left, void *dleft, int dleft_is_bf16, void *dx, int dx_is_bf16,
                            int64_t npix, int64_t hw, int C, hipStream_t s);
int block_head_bwd_dispatch(const float *dout, const void *dl, int dl_is_bf16, const void *dr, int dr_is_bf16, float *dinp,
                            int64_t npix, int C, hipStream_t s);
void gemm_debug_tile(int bm, int bn);
int conv3x3_nhwc_dispatch(const void *x, const void *w, void *y, int batch, int H, int W, int Ci, int Co, hipStream_t s);
int conv3x3_bn_nhwc_dispatch(const void *x, const void *w, void *y, int batch, int H, int W, int Ci, int Co, const MsBnFold *bn_in, void *xhat,
                             const MsBnFold *bn_out, hipStream_t s);
int conv3x3_bnbwd_nhwc_dispatch(const void *dy, const void *w, void *dx, int batch, int H, int W, int Ci, int Co, const MsBnBwd *red, hipStream_t s);
int bn_bwd_apply_sums_dispatch(const MsBnBwd *bn, const void *dy, void *dx, int dx_bf16, float *dgamma, float *dbeta, int64_t npix, int C, hipStream_t s);
int bn_apply_sums_dispatch(const void *x, const MsBnFold *bn, int relu, void *y, int64_t npix, int C, hipStream_t s);
int gemm_bf16_bnbwd_dispatch(const void *A, int a_f32, int64_t lda, const void *B, int b_f32, int64_t ldb, void *C, int c_mode, int64_t ldc, int M,
                             int N, int K, const MsBnBwd *bn, hipStream_t stream);
int linear_bwd_ok(int N, int K);
int linear_bwd_dispatch(const void *dy, int dy_f32, int64_t ld_dy, const void *x, int x_f32, int64_t ld_x, const void *w, int w_f32, void *dx,
                        int dx_bf16, int64_t ld_dx, float *dW, int M, int N, int K, hipStream_t s);
int conv3x3_wgrad_dispatch(const void *x, const void *dy, float *dW, float *scratch, int64_t scratch_floats, int batch, int H, int W,
                           int Ci, int Co, hipStream_t s);
int64_t conv3x3_wgrad_scratch_floats(int batch, int H, int W, int Ci, int Co);
int cast_bf16_multi_dispatch(const MsCastDesc *desc, const int32_t *blocks, int n_blocks, hipStream_t s);
int patchify4_bf16_dispatch(const float *x, void *out, int batch, int C, int H, int W, hipStream_t s);
int adam_multi_dispatch(const MsAdamDesc *desc, const int32_t *blocks, int n_blocks, const void *const *grads, int n_tensors,
                        float step_size, float bc2_sqrt, float one_minus_beta1, float beta2, float one_minus_beta2, float eps,
                        hipStream_t s);
int ln_fwd_dispatch(const float *x, int64_t xps, const float *gamma, const float *beta, float eps, void *out,
                    int out_bf16, int64_t npix, int D, hipStream_t s);
int ln_bwd_dispatch(const float *x, int64_t xps, const float *gamma, float eps, const void *dout, int dout_bf16, float *dx,
                    float *dgamma, float *dbeta, int64_t npix, int D, hipStream_t s);
int ln_taps_fwd_dispatch(const float *x, const float *gamma, const float *beta, float eps, void *out, int out_bf16, int batch, int H, int W,
                         int C, hipStream_t s);
int ln_taps_bwd_dispatch(const float *x, const float *gamma, float eps, const void *dout, int dout_bf16, float *dx, float *dgamma,
                         float *dbeta, int batch, int H, int W, int C, hipStream_t s);
int dtproj_fwd_dispatch(const float *proj, const float *W, const float *bias, float *delta, int64_t npix, int D, int R, int C, hipStream_t s);
int dtproj_bwd_dispatch(const float *ddelta, const float *proj, const float *W, float *dproj, float *dW, float *scratch,
                        int64_t scratch_floats, int64_t npix, int D, int R, int C, hipStream_t s);
int64_t dtproj_bwd_scratch_floats(int64_t npix, int D, int R);
int bn_fwd_dispatch(const void *x, int x_bf16, int64_t xps, const float *shift, const float *gamma, const float *beta, float *running_mean,
                    float *running_var, long long *nbt, float momentum, float eps, int relu, void *y, int y_bf16,
                    float *save_mean, float *save_rstd, float *scratch, int64_t npix, int C, hipStream_t s);
int bn_scratch_floats(int C);
int bn_bwd_dispatch(const void *x, int x_bf16, int64_t xps, const void *dy, int dy_bf16, const float *gamma, const float *beta,
                    const float *save_mean, const float *save_rstd, int relu, void *dx, int dx_bf16, float *dgamma, float *dbeta,
                    float *scratch, int64_t npix, int C, hipStream_t s);
int gemm_bf16_dispatch(const void *A, int a_f32, int a_trans, int64_t lda, const void *B, int b_f32, int b_trans, int64_t ldb, void *C,
                       int c_mode, int64_t ldc, int M, int N, int K, int k_splits, const float *bias, int relu, hipStream_t stream);
int ssd_chunk_fwd_dispatch(const float *x, const float *dt, const float *A, const float *B, const float *C, const float *dt_bias,
                           int softplus, float *dtv, float *cum, float *decay, float *CB, float *S, float *y, int batch, int L, int H,
                           int P, int N, hipStream_t s);
int ssd_chunk_fwd_off_dispatch(const float *x, const float *C, const float *Sin, const float *cum, const float *D, int d_has_hdim, float *y,
                               int batch, int L, int H, int P, int N, hipStream_t s);
int ssd_chunk_bwd_off_dispatch(const float *dy, const float *C, const float *Sin, const float *cum, float *dSin, float *dcum, int batch, int L,
                               int H, int P, int N, hipStream_t s);
int ssd_chunk_bwd_dispatch(const float *x, const float *dy, const float *B, const float *C, const float *CB, const float *Sin, const float *dS,
                           const float *dtv, const float *cum, const float *decay, const float *ddecay, const float *A, const float *D,
                           int d_has_hdim, int softplus, const float *dcum_off, float *dx, float *ddt, float *dA, float *dbias, float *dD,
                           float *dCB, float *dB, float *dC, int batch, int L, int H, int P, int N, hipStream_t s);
int gemm_f32_dispatch(const float *A, int a_trans, int64_t lda, const float *B, int b_trans, int64_t ldb, float *C, int c_mode, int64_t ldc,
                      int M, int N, int K, int k_splits, const float *bias, int relu, hipStream_t stream);
}  // namespace ms

extern "C" {

int ms_ssd_chunk_fwd(const float *x, const float *dt, const float *A, const float *B, const float *C, const float *dt_bias, int dt_softplus,
                     float *dtv, float *cum, float *decay, float *CB, float *S, float *y, int batch, int L, int heads, int headdim,
                     int dstate, void *stream) {
    return ms::ssd_chunk_fwd_dispatch(x, dt, A, B, C, dt_bias, dt_softplus, dtv, cum, decay, CB, S, y, batch, L, heads, headdim, dstate,
                                      (hipStream_t)stream);
}

int ms_ssd_chunk_fwd_off(const float *x, const float *C, const float *S_in, const float *cum, const float *D, int d_has_hdim, float *y,
                         int batch, int L, int heads, int headdim, int dstate, void *stream) {
    return ms::ssd_chunk_fwd_off_dispatch(x, C, S_in, cum, D, d_has_hdim, y, batch, L, heads, headdim, dstate, (hipStream_t)stream);
}

int ms_ssd_chunk_bwd_off(const float *dy, const float *C, const float *S_in, const float *cum, float *dS_in, float *dcum_off, int batch, int L,
                         int heads, int headdim, int dstate, void *stream) {
    return ms::ssd_chunk_bwd_off_dispatch(dy, C, S_in, cum, dS_in, dcum_off, batch, L, heads, headdim, dstate, (hipStream_t)stream);
}

int ms_ssd_chunk_bwd(const float *x, const float *dy, const float *B, const float *C, const float *CB, const float *S_in, const float *dS,
                     const float *dtv, const float *cum, const float *decay, const float *ddecay, const float *A, const float *D,
                     int d_has_hdim, int dt_softplus, const float *dcum_off, float *dx, float *ddt, float *dA, float *dbias, float *dD,
                     float *dCB, float *dB, float *dC, int batch, int L, int heads, int headdim, int dstate, void *stream) {
    return ms::ssd_chunk_bwd_dispatch(x, dy, B, C, CB, S_in, dS, dtv, cum, decay, ddecay, A, D, d_has_hdim, dt_softplus, dcum_off, dx, ddt, dA,
                                      dbias, dD, dCB, dB, dC, batch, L, heads, headdim, dstate, (hipStream_t)stream);
}

int ms_gemm_f32(const float *A, int a_trans, int64_t lda, const float *B, int b_trans, int64_t ldb, float *C, int c_mode, int64_t ldc,
                int M, int N, int K, int k_splits, const float *bias, int relu, void *stream) {
    return ms::gemm_f32_dispatch(A, a_trans, lda, B, b_trans, ldb, C, c_mode, ldc, M, N, K, k_splits, bias, relu, (hipStream_t)stream);
}

int ms_gemm_bf16(const void *A, int a_is_f32, int a_trans, int64_t lda, const void *B, int b_is_f32, int b_trans, int64_t ldb,
                 void *C, int c_mode, int64_t ldc, int M, int N, int K, int k_splits, void *stream) {
    return ms::gemm_bf16_dispatch(A, a_is_f32, a_trans, lda, B, b_is_f32, b_trans, ldb, C, c_mode, ldc, M, N, K, k_splits,
                                  nullptr, 0, (hipStream_t)stream);
}

int ms_gemm_bf16_bias_act(const void *A, int a_is_f32, int a_trans, int64_t lda, const void *B, int b_is_f32, int b_trans, int64_t ldb,
                          void *C, int c_mode, int64_t ldc, int M, int N, int K, const float *bias, int relu, void *stream) {
    return ms::gemm_bf16_dispatch(A, a_is_f32, a_trans, lda, B, b_is_f32, b_trans, ldb, C, c_mode, ldc, M, N, K, 1, bias, relu,
                                  (hipStream_t)stream);
}

int ms_gemm_bf16_wgrad_bias(const void *dy, int dy_is_f32, int64_t lddy, const void *x, int x_is_f32, int64_t ldx, float *dW, int64_t lddw,
                            float *dbias, int N, int K, int M, int k_splits, void *stream) {
    if (!dbias) return MS_ERR_NULL;
    return ms::gemm_bf16_dispatch(dy, dy_is_f32, 1, lddy, x, x_is_f32, 1, ldx, dW, 2, lddw, N, K, M, k_splits, dbias, 0, (hipStream_t)stream);
}

int ms_conv3x3_nhwc_bf16(const void *x, const void *w, void *y, int batch, int H, int W, int Ci, int Co, void *stream) {
    return ms::conv3x3_nhwc_dispatch(x, w, y, batch, H, W, Ci, Co, (hipStream_t)stream);
}

int ms_conv3x3_bn_nhwc_bf16(const void *x, const void *w, void *y, int batch, int H, int W, int Ci, int Co, const MsBnFold *bn_in, void *xhat,
                            const MsBnFold *bn_out, void *stream) {
    return ms::conv3x3_bn_nhwc_dispatch(x, w, y, batch, H, W, Ci, Co, bn_in, xhat, bn_out, (hipStream_t)stream);
}

int ms_conv3x3_bnbwd_nhwc_bf16(const void *dy, const void *w, void *dx, int batch, int H, int W, int Ci, int Co, const MsBnBwd *red, void *stream) {
    return ms::conv3x3_bnbwd_nhwc_dispatch(dy, w, dx, batch, H, W, Ci, Co, red, (hipStream_t)stream);
}

int ms_bn_bwd_apply_sums_nhwc(const MsBnBwd *bn, const void *dy, void *dx, int dx_is_bf16, float *dgamma, float *dbeta, int64_t npix, int C,
                              void *stream) {
    return ms::bn_bwd_apply_sums_dispatch(bn, dy, dx, dx_is_bf16, dgamma, dbeta, npix, C, (hipStream_t)stream);
}

int ms_bn_apply_sums_nhwc(const void *x, const MsBnFold *bn, int relu, void *y, int64_t npix, int C, void *stream) {
    return ms::bn_apply_sums_dispatch(x, bn, relu, y, npix, C, (hipStream_t)stream);
}

int ms_gemm_bf16_bnbwd(const void *A, int a_is_f32, int64_t lda, const void *B, int b_is_f32, int64_t ldb, void *C, int c_mode, int64_t ldc, int M,
                       int N, int K, const MsBnBwd *bn, void *stream) {
    return ms::gemm_bf16_bnbwd_dispatch(A, a_is_f32, lda, B, b_is_f32, ldb, C, c_mode, ldc, M, N, K, bn, (hipStream_t)stream);
}

int ms_linear_bwd_ok(int N, int K) { return ms::linear_bwd_ok(N, K); }

int ms_linear_bwd_bf16(const void *dy, int dy_is_f32, int64_t ld_dy, const void *x, int x_is_f32, int64_t ld_x, const void *w, int w_is_f32,
                       void *dx, int dx_is_bf16, int64_t ld_dx, float *dW, int M, int N, int K, void *stream) {
    return ms::linear_bwd_dispatch(dy, dy_is_f32, ld_dy, x, x_is_f32, ld_x, w, w_is_f32, dx, dx_is_bf16, ld_dx, dW, M, N, K, (hipStream_t)stream);
}

int ms_conv3x3_wgrad(const void *x, const void *dy, float *dW, float *scratch, int64_t scratch_floats, int batch, int H, int W, int Ci,
                     int Co, void *stream) {
    return ms::conv3x3_wgrad_dispatch(x, dy, dW, scratch, scratch_floats, batch, H, W, Ci, Co, (hipStream_t)stream);
}

int64_t ms_conv3x3_wgrad_scratch_floats(int batch, int H, int W, int Ci, int Co) {
    return ms::conv3x3_wgrad_scratch_floats(batch, H, W, Ci, Co);
}

int ms_debug_gemm_tile(int bm, int bn) { ms::gemm_debug_tile(bm, bn); return MS_OK; }

int ms_cast_bf16_multi(const MsCastDesc *desc, const int32_t *blocks, int n_blocks, void *stream) {
    return ms::cast_bf16_multi_dispatch(desc, blocks, n_blocks, (hipStream_t)stream);
}

int ms_patchify4_bf16(const float *x, void *out, int batch, int C, int H, int W, void *stream) {
    return ms::patchify4_bf16_dispatch(x, out, batch, C, H, W, (hipStream_t)stream);
}

int ms_adam_multi(const MsAdamDesc *desc, const int32_t *blocks, int n_blocks, const void *const *grads, int n_tensors,
                  float step_size, float bias_correction2_sqrt, float one_minus_beta1, float beta2, float one_minus_beta2, float eps,
                  void *stream) {
    return ms::adam_multi_dispatch(desc, blocks, n_blocks, grads, n_tensors, step_size, bias_correction2_sqrt, one_minus_beta1, beta2,
                                   one_minus_beta2, eps, (hipStream_t)stream);
}

int ms_block_head_bwd(const float *dout, const void *dleft, int dleft_is_bf16, const void *dright, int dright_is_bf16, float *dinput,
                      int64_t npix, int C, void *stream) {
    return ms::block_head_bwd_dispatch(dout, dleft, dleft_is_bf16, dright, dright_is_bf16, dinput, npix, C, (hipStream_t)stream);
}

int ms_selective_scan_fwd(const MsScanParams *p, void *stream) {
    if (!p) return MS_ERR_NULL;
    return ms::scan_fwd_dispatch(*p, (hipStream_t)stream);
}

int ms_selective_scan_bwd(const MsScanBwdParams *p, void *stream) {
    if (!p) return MS_ERR_NULL;
    return ms::scan_bwd_dispatch(*p, (hipStream_t)stream);
}

int ms_scan_n_chunks(int seqlen) { return seqlen <= 0 ? 0 : (seqlen + MS_SCAN_CHUNK - 1) / MS_SCAN_CHUNK; }
int64_t ms_scan_seg_floats(int batch, int dim, int segments) {
    return (batch <= 0 || dim <= 0 || segments < 2) ? 0 : 2ll * batch * segments * 16 * dim;      // two planes of [batch][segment][16 states][dim]
}

int ms_cross_scan(const float *x, float *xs, int batch, int dim, int H, int W, void *stream) {
    return ms::cross_scan_dispatch(x, xs, batch, dim, H, W, (hipStream_t)stream);
}

int ms_cross_scan_nhwc(const float *pix, int64_t pixel_stride, float *seq, int batch, int H, int W, int C, void *stream) {
    return ms::cross_scan_nhwc_dispatch(pix, pixel_stride, seq, batch, H, W, C, (hipStream_t)stream);
}

int ms_cross_merge_nhwc(const float *seq, float *pix, int64_t pixel_stride, int batch, int H, int W, int C, void *stream) {
    return ms::cross_merge_nhwc_dispatch(seq, pix, pixel_stride, batch, H, W, C, (hipStream_t)stream);
}

int ms_cross_merge(const float *ys, float *y, int batch, int dim, int H, int W, void *stream) {
    return ms::cross_merge_dispatch(ys, y, batch, dim, H, W, (hipStream_t)stream);
}

int ms_dwconv3x3_silu_fwd(const float *x, const float *w, const float *bias, float *y,
                          int batch, int C, int H, int W, void *stream) {
    return ms::dwconv_fwd_dispatch(x, w, bias, y, batch, C, H, W, (hipStream_t)stream);
}

int ms_dwconv3x3_silu_bwd(const float *x, const float *w, const float *bias, const float *dy,
                          float *dx, float *dw, float *dbias, int batch, int C, int H, int W, void *stream) {
    return ms::dwconv_bwd_dispatch(x, w, bias, dy, dx, dw, dbias, batch, C, H, W, (hipStream_t)stream);
}

int ms_dwconv3x3_silu_nhwc_fwd(const void *x, int x_is_bf16, const float *w, const float *bias, float *y,
                               int batch, int C, int H, int W, int64_t x_pixel_stride, void *stream) {
    return ms::dwconv_nhwc_fwd_dispatch(x, x_is_bf16, w, bias, y, batch, C, H, W, x_pixel_stride, (hipStream_t)stream);
}

int ms_dwconv3x3_silu_nhwc_bwd(const void *x, int x_is_bf16, const float *w, const float *bias, const float *dy, int dy_ndir,
                               int64_t dy_dir_stride, const float *dy_extra, void *dx, int dx_is_bf16, int64_t dx_pixel_stride,
                               float *scratch, float *dw, float *dbias, int batch, int C, int H, int W,
                               int64_t x_pixel_stride, void *stream) {
    return ms::dwconv_nhwc_bwd_dispatch(x, x_is_bf16, w, bias, dy, dy_ndir, dy_dir_stride, dy_extra, dx, dx_is_bf16,
                                        dx_pixel_stride, scratch, dw, dbias, batch, C, H, W, x_pixel_stride,
                                        (hipStream_t)stream);
}

int ms_ln_gate_fwd(const float *y4, int64_t dir_stride, const void *z, int z_is_bf16, int64_t z_pixel_stride,
                   const float *gamma, const float *beta, float eps, void *out, int out_is_bf16,
                   int64_t npix, int D, void *stream) {
    return ms::ln_gate_fwd_dispatch(y4, dir_stride, z, z_is_bf16, z_pixel_stride, gamma, beta, eps, out, out_is_bf16,
                                    nullptr, npix, D, (hipStream_t)stream);
}

int ms_ln_gate_fwd_keep(const float *y4, int64_t dir_stride, const void *z, int z_is_bf16, int64_t z_pixel_stride,
                        const float *gamma, const float *beta, float eps, void *out, int out_is_bf16, float *ysum,
                        int64_t npix, int D, void *stream) {
    if (!ysum) return MS_ERR_NULL;
    return ms::ln_gate_fwd_dispatch(y4, dir_stride, z, z_is_bf16, z_pixel_stride, gamma, beta, eps, out, out_is_bf16,
                                    ysum, npix, D, (hipStream_t)stream);
}

int ms_ln_gate_bwd(const float *y4, int64_t dir_stride, const void *z, int z_is_bf16, int64_t z_pixel_stride,
                   const float *gamma, const float *beta, float eps, const void *dout, int dout_is_bf16,
                   float *dy, void *dz, int64_t dz_pixel_stride, float *dgamma, float *dbeta, int64_t npix, int D,
                   void *stream) {
    return ms::ln_gate_bwd_dispatch(y4, dir_stride, z, z_is_bf16, z_pixel_stride, gamma, beta, eps, dout, dout_is_bf16,
                                    dy, dz, dz_pixel_stride, dgamma, dbeta, npix, D, (hipStream_t)stream);
}

int ms_block_tail_fwd(const void *left, int left_is_bf16, const void *x, int x_is_bf16, const float *input,
                      const float *sample_scale, float *out, int64_t npix, int64_t pixels_per_sample, int C, void *stream) {
    return ms::block_tail_fwd_dispatch(left, left_is_bf16, x, x_is_bf16, input, sample_scale, out, npix, pixels_per_sample, C,
                                       (hipStream_t)stream);
}

int ms_block_tail_bwd(const float *dout, const float *sample_scale, void *dleft, int dleft_is_bf16, void *dx, int dx_is_bf16,
                      int64_t npix, int64_t pixels_per_sample, int C, void *stream) {
    return ms::block_tail_bwd_dispatch(dout, sample_scale, nullptr, dleft, dleft_is_bf16, dx, dx_is_bf16, npix, pixels_per_sample, C,
                                       (hipStream_t)stream);
}

int ms_block_tail_bwd_relu(const float *dout, const float *sample_scale, const void *left, void *dleft, int dleft_is_bf16, void *dx,
                           int dx_is_bf16, int64_t npix, int64_t pixels_per_sample, int C, void *stream) {
    if (!left) return MS_ERR_NULL;
    return ms::block_tail_bwd_dispatch(dout, sample_scale, left, dleft, dleft_is_bf16, dx, dx_is_bf16, npix, pixels_per_sample, C,
                                       (hipStream_t)stream);
}

int ms_layernorm_fwd(const float *x, int64_t x_pixel_stride, const float *gamma, const float *beta, float eps, void *out,
                     int out_is_bf16, int64_t npix, int D, void *stream) {
    return ms::ln_fwd_dispatch(x, x_pixel_stride, gamma, beta, eps, out, out_is_bf16, npix, D, (hipStream_t)stream);
}

int ms_layernorm_bwd(const float *x, int64_t x_pixel_stride, const float *gamma, float eps, const void *dout, int dout_is_bf16,
                     float *dx, float *dgamma, float *dbeta, int64_t npix, int D, void *stream) {
    return ms::ln_bwd_dispatch(x, x_pixel_stride, gamma, eps, dout, dout_is_bf16, dx, dgamma, dbeta, npix, D, (hipStream_t)stream);
}

int ms_layernorm_taps_fwd(const float *x, const float *gamma, const float *beta, float eps, void *out, int out_is_bf16, int batch, int H,
                          int W, int C, void *stream) {
    return ms::ln_taps_fwd_dispatch(x, gamma, beta, eps, out, out_is_bf16, batch, H, W, C, (hipStream_t)stream);
}

int ms_layernorm_taps_bwd(const float *x, const float *gamma, float eps, const void *dout, int dout_is_bf16, float *dx, float *dgamma,
                          float *dbeta, int batch, int H, int W, int C, void *stream) {
    return ms::ln_taps_bwd_dispatch(x, gamma, eps, dout, dout_is_bf16, dx, dgamma, dbeta, batch, H, W, C, (hipStream_t)stream);
}

int ms_dtproj_fwd(const float *proj, const float *Wdt, float *delta, int64_t npix, int D, int R, int row_width, void *stream) {
    return ms::dtproj_fwd_dispatch(proj, Wdt, nullptr, delta, npix, D, R, row_width, (hipStream_t)stream);
}

int ms_dtproj_fwd_act(const float *proj, const float *Wdt, const float *dt_bias, float *delta, int64_t npix, int D, int R, int row_width,
                      void *stream) {
    if (!dt_bias) return MS_ERR_NULL;
    return ms::dtproj_fwd_dispatch(proj, Wdt, dt_bias, delta, npix, D, R, row_width, (hipStream_t)stream);
}

int ms_dtproj_bwd(const float *ddelta, const float *proj, const float *Wdt, float *dproj, float *dWdt, float *scratch,
                  int64_t scratch_floats, int64_t npix, int D, int R, int row_width, void *stream) {
    return ms::dtproj_bwd_dispatch(ddelta, proj, Wdt, dproj, dWdt, scratch, scratch_floats, npix, D, R, row_width, (hipStream_t)stream);
}

int64_t ms_dtproj_bwd_scratch_floats(int64_t npix, int D, int R) { return ms::dtproj_bwd_scratch_floats(npix, D, R); }

int ms_bn_relu_nhwc_fwd(const void *x, int x_is_bf16, int64_t x_pixel_stride, const float *input_shift, const float *gamma, const float *beta,
                        float *running_mean,
                        float *running_var, int64_t *num_batches_tracked, float momentum, float eps, int relu, void *y,
                        int y_is_bf16, float *save_mean, float *save_rstd, float *scratch, int64_t npix, int C,
                        void *stream) {
    return ms::bn_fwd_dispatch(x, x_is_bf16, x_pixel_stride, input_shift, gamma, beta, running_mean, running_var, (long long *)num_batches_tracked,
                               momentum, eps, relu, y, y_is_bf16, save_mean, save_rstd, scratch, npix, C,
                               (hipStream_t)stream);
}

int ms_bn_relu_nhwc_bwd(const void *x, int x_is_bf16, int64_t x_pixel_stride, const void *dy, int dy_is_bf16, const float *gamma, const float *beta,
                        const float *save_mean, const float *save_rstd, int relu, void *dx, int dx_is_bf16, float *dgamma, float *dbeta,
                        float *scratch, int64_t npix, int C, void *stream) {
    return ms::bn_bwd_dispatch(x, x_is_bf16, x_pixel_stride, dy, dy_is_bf16, gamma, beta, save_mean, save_rstd, relu, dx, dx_is_bf16, dgamma, dbeta,
                               scratch, npix, C, (hipStream_t)stream);
}

int64_t ms_dwconv3x3_silu_nhwc_bwd_scratch_floats(int batch, int C, int H, int W) {
    return ms::dwconv_nhwc_bwd_scratch_floats(batch, C, H, W);
}

int ms_bn_scratch_floats(int C) { return ms::bn_scratch_floats(C); }

int ms_rms_gate_fwd(const float *y, int64_t dir_stride, int ndir, const void *z, int z_is_bf16, int64_t z_pixel_stride,
                    const float *weight, float eps, void *out, int out_is_bf16, int64_t npix, int D, void *stream) {
    return ms::rms_gate_fwd_dispatch(y, dir_stride, ndir, z, z_is_bf16, z_pixel_stride, weight, eps, out, out_is_bf16, npix, D,
                                     (hipStream_t)stream);
}

int ms_rms_gate_bwd(const float *y, int64_t dir_stride, int ndir, const void *z, int z_is_bf16, int64_t z_pixel_stride,
                    const float *weight, float eps, const void *dout, int dout_is_bf16, float *dy, void *dz,
                    int64_t dz_pixel_stride, float *dweight, int64_t npix, int D, void *stream) {
    return ms::rms_gate_bwd_dispatch(y, dir_stride, ndir, z, z_is_bf16, z_pixel_stride, weight, eps, dout, dout_is_bf16, dy, dz,
                                     dz_pixel_stride, dweight, npix, D, (hipStream_t)stream);
}

int ms_ssd_chunk_carry(const float *in, const float *decay, float *out, const float *fwd_out, float *ddecay, int batch, int chunks,
                       int groups, int dstate, int heads_per_group, int headdim, int reverse, void *stream) {
    return ms::ssd_carry_dispatch(in, decay, out, fwd_out, ddecay, batch, chunks, groups, dstate, heads_per_group, headdim, reverse,
                                  (hipStream_t)stream);
}

// One workgroup that keeps a CU busy for `cycles` shader clocks: a probe for whether two HIP streams really execute
// concurrently (two probes take the time of one) or alias one hardware queue (the time of two).
__global__ void ms_spin_kernel(long long cycles, int *sink) {
    const long long t0 = clock64();
    int acc = 0;
    while (clock64() - t0 < cycles) acc += 1;
    if (sink && threadIdx.x == 0 && acc == -1) *sink = acc;
}

int ms_spin(long long cycles, void *stream) {
    if (cycles < 0 || cycles > (1LL << 32)) return MS_ERR_SHAPE;           // bounded: never a runaway kernel
    hipLaunchKernelGGL(ms_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, cycles, (int *)nullptr);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int ms_abi_version(void) { return MEDSCAN_ABI_VERSION; }

const char *ms_status_string(int status) {
    switch (status) {
        case MS_OK: return "ok";
        case MS_ERR_NULL: return "required pointer is NULL";
        case MS_ERR_SHAPE: return "invalid shape";
        case MS_ERR_DSTATE: return "unsupported state dimension";
        case MS_ERR_STRIDE: return "unsupported strides";
        case MS_ERR_LAUNCH: return "kernel launch failed";
        case MS_ERR_UNSUPPORTED: return "unsupported configuration";
    }
    return "unknown status";
}

}  // extern "C"
