// Depthwise 3x3 conv (pad 1) + bias + SiLU on CHANNEL-LAST tensors for gfx950 -- the layout in_proj produces, so
// the reference's `x.permute(0,3,1,2).contiguous()` before the conv (MedMamba.py:472) and the transpose back after
// the scan (:477) disappear.
//   x : (B, H, W, *) with pixel stride `xps` elements (reads the first C channels of each pixel, so the `x` half of
//       in_proj's output xz is consumed in place);  y : (B, H, W, C) contiguous fp32.
// One wave = 64 channels x one image row; lanes are channels (256-byte coalesced rows), the 3x3 window slides along W
// in registers: 3 loads + 9 FMAs per output, every input row is fetched by 3 waves (L2 hits).
//   bwd: two streaming passes -- (1) dpre = dy * silu'(pre) with pre recomputed (dy = a sum of gradient slabs, see the kernel),
//        dw[c,k] / dbias[c] accumulated per lane over the rows, combined per workgroup into its slot of a partial-sum table that
//        a finalize kernel adds up; (2) dx = conv^T(dpre).  dpre and the table live in a caller-provided scratch tensor.
#include <hip/hip_runtime.h>
#include "medscan.h"

namespace ms {

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

template <typename T> __device__ __forceinline__ float ldf(const T *p);
template <> __device__ __forceinline__ float ldf<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ldf<unsigned short>(const unsigned short *p) {     // bf16 bits
    return __builtin_bit_cast(float, (unsigned)(*p) << 16);
}

__device__ __forceinline__ void stf(float *p, float v) { *p = v; }
__device__ __forceinline__ void stf(unsigned short *p, float v) { *p = __builtin_bit_cast(unsigned short, (__bf16)v); }

template <typename T>
__global__ void __launch_bounds__(256)
dwconv_nhwc_fwd_kernel(const T *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                       float *__restrict__ y, int C, int H, int W, int64_t xps, int rows) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);          // b*H + h
    if (row >= rows) return;
    const bool cv = c < C;
    const int cc = cv ? c : C - 1;
    const int h = row % H;
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = w[cc * 9 + i];
    const float bv = bias ? bias[cc] : 0.0f;
    const T *r1 = x + (int64_t)row * W * xps + cc;                  // row h
    const T *r0 = r1 - (int64_t)W * xps, *r2 = r1 + (int64_t)W * xps;
    const bool v0 = h > 0, v2 = h < H - 1;
    const T *r0v = v0 ? r0 : r1, *r2v = v2 ? r2 : r1;             // rows outside the image: read row h instead, zeroed after
    float a0 = 0, a1 = 0, a2 = 0, b0, b1, b2;                     // columns w-1 (a), w (b); w+1 comes from the batch below
    b0 = v0 ? ldf(r0) : 0.0f; b1 = ldf(r1); b2 = v2 ? ldf(r2) : 0.0f;
    float *yo = y + (int64_t)row * W * C + c;
    // 4 columns per trip, their 12 loads issued together (the walk along the row is latency-bound otherwise)
    for (int w0 = 0; w0 < W; w0 += 4) {
        float n0[4], n1[4], n2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int wn = w0 + q + 1;
            const bool in = wn < W;
            const int64_t o = (int64_t)(in ? wn : W - 1) * xps;
            const float t0 = ldf(r0v + o), t1 = ldf(r1 + o), t2 = ldf(r2v + o);
            n0[q] = (in && v0) ? t0 : 0.0f; n1[q] = in ? t1 : 0.0f; n2[q] = (in && v2) ? t2 : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (w0 + q < W) {
                const float c0 = n0[q], c1 = n1[q], c2 = n2[q];
                float acc = bv;
                acc = fmaf(k[0], a0, acc); acc = fmaf(k[1], b0, acc); acc = fmaf(k[2], c0, acc);
                acc = fmaf(k[3], a1, acc); acc = fmaf(k[4], b1, acc); acc = fmaf(k[5], c1, acc);
                acc = fmaf(k[6], a2, acc); acc = fmaf(k[7], b2, acc); acc = fmaf(k[8], c2, acc);
                if (cv) yo[(int64_t)(w0 + q) * C] = acc * sigm(acc);
                a0 = b0; a1 = b1; a2 = b2; b0 = c0; b1 = c1; b2 = c2;
            }
        }
    }
}

// Backward, pass 1: dpre = dy * silu'(conv(x) + b) for one image row (same sliding window as the forward: 3 loads
// per pixel), stored to a scratch tensor; the parameter gradients dw[c, 0..8], dbias[c] are accumulated per lane over the
// rows and leave the workgroup as one slot of the partial-sum table.
// pass-1 geometry, measured on the four MedMamba-T stage shapes (tools/bench_dwconv.py): rows per wave 1 / 2 / 4 -> 165 / 168 /
// 196 us at stage 0 (2: half the partial-sum slots of 1 at the same speed); columns per trip 4 / 8 -> 168 / 167 us
constexpr int kRowsPerWaveBwd1 = 2;
constexpr int kColsBwd1 = 4;

// NDIR > 0: the slab count is a compile-time constant, so the 4 x (NDIR + EXTRA) gradient loads of a trip are issued
// together with the 12 window loads (a run-time slab loop makes each add wait for its own load: measured 330 us instead
// of ~100 us at stage 0).  NDIR == 0: run-time `ndir`, any count up to 8.
template <typename T, int NDIR, bool EXTRA>
__global__ void __launch_bounds__(256)
dwconv_nhwc_bwd1_kernel(const T *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                        const float *__restrict__ dy, int ndir, int64_t dir_stride, const float *__restrict__ dy_extra,
                        float *__restrict__ dpre, float *__restrict__ part, int C, int H, int W, int64_t xps, int rows) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    __shared__ float red[3][10][64];
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const int row_first = (blockIdx.y * 4 + wv) * kRowsPerWaveBwd1;
    const bool cv = c < C;
    const int cc = cv ? c : C - 1;
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = w[cc * 9 + i];
    const float bv = bias ? bias[cc] : 0.0f;
    float acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = 0.0f;
    const int row_end = min(rows, row_first + kRowsPerWaveBwd1);
    for (int row = row_first; row < row_end; ++row) {
        const int h = row % H;
        const T *r1 = x + (int64_t)row * W * xps + cc;
        const T *r0 = r1 - (int64_t)W * xps, *r2 = r1 + (int64_t)W * xps;
        const bool v0 = h > 0, v2 = h < H - 1;
        const T *r0v = v0 ? r0 : r1, *r2v = v2 ? r2 : r1;        // rows outside the image: read row h instead, zeroed after
        float a0 = 0, a1 = 0, a2 = 0, b0, b1, b2;
        b0 = v0 ? ldf(r0) : 0.0f; b1 = ldf(r1); b2 = v2 ? ldf(r2) : 0.0f;
        const float *go = dy + (int64_t)row * W * C + cc;
        const float *ge = EXTRA ? dy_extra + (int64_t)row * W * C + cc : nullptr;
        float *po = dpre + (int64_t)row * W * C + c;
        // 4 columns per trip: the 16 loads of a trip (3 rows x 4 next columns + 4 dy) are issued together, so a wave keeps
        // 16 requests in flight instead of 4 -- the walk along the row is latency-bound otherwise (measured 245 us for a
        // 192 MB pass at stage 0)
        for (int w0 = 0; w0 < W; w0 += kColsBwd1) {
            float n0[kColsBwd1], n1[kColsBwd1], n2[kColsBwd1], gq[kColsBwd1];
#pragma unroll
            for (int q = 0; q < kColsBwd1; ++q) {
                const int wn = w0 + q + 1;                       // the column to the right of pixel w0 + q
                const bool in = wn < W;
                const int64_t o = (int64_t)(in ? wn : W - 1) * xps;
                const float t0 = ldf(r0v + o), t1 = ldf(r1 + o), t2 = ldf(r2v + o);
                n0[q] = (in && v0) ? t0 : 0.0f; n1[q] = in ? t1 : 0.0f; n2[q] = (in && v2) ? t2 : 0.0f;
                // the incoming gradient may arrive as `ndir` slabs (the scan's four per-direction du) plus one more term
                // (the x_proj backward): summed here instead of by separate reduce / add kernels
                const int64_t gi = (int64_t)min(w0 + q, W - 1) * C;
                float gs = go[gi];
                if (NDIR > 0) {
                    float gk[NDIR > 0 ? NDIR : 1];
#pragma unroll
                    for (int kd = 1; kd < NDIR; ++kd) gk[kd] = go[gi + kd * dir_stride];
                    const float gx = EXTRA ? ge[gi] : 0.0f;
#pragma unroll
                    for (int kd = 1; kd < NDIR; ++kd) gs += gk[kd];
                    gs += gx;
                } else {
                    for (int kd = 1; kd < ndir; ++kd) gs += go[gi + kd * dir_stride];
                    if (EXTRA) gs += ge[gi];
                }
                gq[q] = gs;
            }
#pragma unroll
            for (int q = 0; q < kColsBwd1; ++q) {
                if (w0 + q < W) {
                    const float c0 = n0[q], c1 = n1[q], c2 = n2[q], g = gq[q];
                    float pre = bv;
                    pre = fmaf(k[0], a0, pre); pre = fmaf(k[1], b0, pre); pre = fmaf(k[2], c0, pre);
                    pre = fmaf(k[3], a1, pre); pre = fmaf(k[4], b1, pre); pre = fmaf(k[5], c1, pre);
                    pre = fmaf(k[6], a2, pre); pre = fmaf(k[7], b2, pre); pre = fmaf(k[8], c2, pre);
                    const float sg = sigm(pre);
                    const float dp = g * (sg * (1.0f + pre * (1.0f - sg)));
                    if (cv) po[(int64_t)(w0 + q) * C] = dp;
                    acc[0] = fmaf(dp, a0, acc[0]); acc[1] = fmaf(dp, b0, acc[1]); acc[2] = fmaf(dp, c0, acc[2]);
                    acc[3] = fmaf(dp, a1, acc[3]); acc[4] = fmaf(dp, b1, acc[4]); acc[5] = fmaf(dp, c1, acc[5]);
                    acc[6] = fmaf(dp, a2, acc[6]); acc[7] = fmaf(dp, b2, acc[7]); acc[8] = fmaf(dp, c2, acc[8]);
                    acc[9] += dp;
                    a0 = b0; a1 = b1; a2 = b2; b0 = c0; b1 = c1; b2 = c2;
                }
            }
        }
    }
    // combine the block's 4 waves (same channels, different rows) in LDS; the block's 10 sums per channel go to its own
    // slot of `part` ([gridDim.y][10][gridDim.x*64]) and dwconv_nhwc_bwd_finalize_kernel adds the slots up.  (Atomics on
    // the 10*C result addresses serialise: measured 65 of 230 us at stage 0.)
    if (wv > 0) {
#pragma unroll
        for (int i = 0; i < 10; ++i) red[wv - 1][i][ln] = acc[i];
    }
    __syncthreads();
    if (wv == 0) {
        const int cpad = gridDim.x * 64;
        float *po = part + (int64_t)blockIdx.y * 10 * cpad + blockIdx.x * 64 + ln;
#pragma unroll
        for (int i = 0; i < 10; ++i) po[i * cpad] = cv ? acc[i] + red[0][i][ln] + red[1][i][ln] + red[2][i][ln] : 0.0f;
    }
}

// dw[c, 0..8] += sum over slots of part[slot][0..8][c], dbias[c] += ... [9][c].  One block = 64 columns (a column =
// (tap, channel), contiguous in `part`) x 16 slot slices; sole writer of its outputs, so plain read-modify-write.
__global__ void __launch_bounds__(1024)
dwconv_nhwc_bwd_finalize_kernel(const float *__restrict__ part, int nslots, int cpad, int C, float *__restrict__ dw,
                                float *__restrict__ dbias) {
    __shared__ float red[16][64];
    const int ln = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + ln;                       // < 10 * cpad (cpad is a multiple of 64)
    const float *p = part + col;
    const int64_t pitch = (int64_t)10 * cpad;
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int k = sl;
    for (; k + 48 < nslots; k += 64) {
        s0 += p[k * pitch]; s1 += p[(k + 16) * pitch]; s2 += p[(k + 32) * pitch]; s3 += p[(k + 48) * pitch];
    }
    for (; k < nslots; k += 16) s0 += p[k * pitch];
    red[sl][ln] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0) {
        float t = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][ln];
        const int i = col / cpad, c = col - i * cpad;
        if (c < C) {
            if (i < 9) dw[c * 9 + i] += t;
            else if (dbias) dbias[c] += t;
        }
    }
}

// Backward, pass 2: dx = conv3x3^T(dpre): dx[h][w] = sum_{i,j} dpre[h+1-i][w+1-j] * k[i][j]  (a correlation with the
// flipped kernel: again a 3-row sliding window).
template <typename TO>
__global__ void __launch_bounds__(256)
dwconv_nhwc_bwd2_kernel(const float *__restrict__ dpre, const float *__restrict__ w, TO *__restrict__ dx, int64_t dxps,
                        int C, int H, int W, int rows) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bool cv = c < C;
    const int cc = cv ? c : C - 1;
    const int h = row % H;
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = w[cc * 9 + i];
    const float *r1 = dpre + (int64_t)row * W * C + cc;
    const float *r0 = r1 - (int64_t)W * C, *r2 = r1 + (int64_t)W * C;
    const bool v0 = h > 0, v2 = h < H - 1;
    const float *r0v = v0 ? r0 : r1, *r2v = v2 ? r2 : r1;
    float a0 = 0, a1 = 0, a2 = 0, b0, b1, b2;                     // dpre columns w-1 (a), w (b); rows h-1,h,h+1
    b0 = v0 ? r0[0] : 0.0f; b1 = r1[0]; b2 = v2 ? r2[0] : 0.0f;
    TO *xo = dx + (int64_t)row * W * dxps + c;                    // dx may be a channel slice of a wider gradient tensor
    for (int w0 = 0; w0 < W; w0 += 4) {                            // 4 columns per trip, 12 loads in flight
        float n0[4], n1[4], n2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int wn = w0 + q + 1;
            const bool in = wn < W;
            const int64_t o = (int64_t)(in ? wn : W - 1) * C;
            const float t0 = r0v[o], t1 = r1[o], t2 = r2v[o];
            n0[q] = (in && v0) ? t0 : 0.0f; n1[q] = in ? t1 : 0.0f; n2[q] = (in && v2) ? t2 : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (w0 + q < W) {
                const float c0 = n0[q], c1 = n1[q], c2 = n2[q];
                // i = 0 -> row h+1 (x2), i = 2 -> row h-1 (x0);  j = 0 -> column w+1 (c), j = 2 -> column w-1 (a)
                float a = 0.0f;
                a = fmaf(k[0], c2, a); a = fmaf(k[1], b2, a); a = fmaf(k[2], a2, a);
                a = fmaf(k[3], c1, a); a = fmaf(k[4], b1, a); a = fmaf(k[5], a1, a);
                a = fmaf(k[6], c0, a); a = fmaf(k[7], b0, a); a = fmaf(k[8], a0, a);
                if (cv) stf(xo + (int64_t)(w0 + q) * dxps, a);
                a0 = b0; a1 = b1; a2 = b2; b0 = c0; b1 = c1; b2 = c2;
            }
        }
    }
}

template <typename T>
static int launch_fwd(const void *x, const float *w, const float *bias, float *y, int batch, int C, int H, int W,
                      int64_t xps, hipStream_t s) {
    const int rows = batch * H;
    hipLaunchKernelGGL((dwconv_nhwc_fwd_kernel<T>), dim3((C + 63) / 64, (rows + 3) / 4, 1), dim3(256), 0, s,
                       (const T *)x, w, bias, y, C, H, W, xps, rows);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int dwconv_nhwc_fwd_dispatch(const void *x, int x_is_bf16, const float *w, const float *bias, float *y,
                             int batch, int C, int H, int W, int64_t xps, hipStream_t s) {
    if (!x || !w || !y) return MS_ERR_NULL;
    if (batch < 0 || C <= 0 || H <= 0 || W <= 0 || xps < C) return MS_ERR_SHAPE;
    if (batch == 0) return MS_OK;
    return x_is_bf16 ? launch_fwd<unsigned short>(x, w, bias, y, batch, C, H, W, xps, s)
                     : launch_fwd<float>(x, w, bias, y, batch, C, H, W, xps, s);
}

template <typename T>
static int launch_bwd(const void *x, const float *w, const float *bias, const float *dy, int ndir, int64_t dir_stride,
                      const float *dy_extra, void *dx, int dx_bf16, int64_t dxps, float *scratch, float *dw, float *dbias,
                      int batch, int C, int H, int W, int64_t xps, hipStream_t s) {
    const int rows = batch * H;
    // dx doubles as the dpre scratch of pass 1?  No: pass 2 reads rows h-1..h+1 of dpre while writing row h of dx,
    // so dpre needs its own buffer -- the caller passes it in `scratch` ((batch, H, W, C) fp32).
    const int tasks = (rows + kRowsPerWaveBwd1 - 1) / kRowsPerWaveBwd1;
    const dim3 g1((C + 63) / 64, (tasks + 3) / 4, 1);
    float *part = scratch + (int64_t)rows * W * C;            // behind dpre: dwconv_nhwc_bwd_scratch_floats()
#define MS_BWD1(ND, EX) hipLaunchKernelGGL((dwconv_nhwc_bwd1_kernel<T, ND, EX>), g1, dim3(256), 0, s, (const T *)x, w, bias, dy, \
                                           ndir, dir_stride, dy_extra, scratch, part, C, H, W, xps, rows)
    if (ndir == 1 && !dy_extra) MS_BWD1(1, false);
    else if (ndir == 4 && dy_extra) MS_BWD1(4, true);
    else if (dy_extra) MS_BWD1(0, true);
    else MS_BWD1(0, false);
#undef MS_BWD1
    const int cpad = (int)g1.x * 64;
    hipLaunchKernelGGL(dwconv_nhwc_bwd_finalize_kernel, dim3(10 * cpad / 64), dim3(1024), 0, s, part, (int)g1.y, cpad, C, dw, dbias);
    const dim3 g2((C + 63) / 64, (rows + 3) / 4, 1);
    if (dx_bf16) hipLaunchKernelGGL((dwconv_nhwc_bwd2_kernel<unsigned short>), g2, dim3(256), 0, s, scratch, w, (unsigned short *)dx, dxps, C, H, W, rows);
    else         hipLaunchKernelGGL((dwconv_nhwc_bwd2_kernel<float>), g2, dim3(256), 0, s, scratch, w, (float *)dx, dxps, C, H, W, rows);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int64_t dwconv_nhwc_bwd_scratch_floats(int batch, int C, int H, int W) {
    if (batch < 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    const int64_t rows = (int64_t)batch * H;
    const int64_t tasks = (rows + kRowsPerWaveBwd1 - 1) / kRowsPerWaveBwd1;
    return rows * W * C + ((tasks + 3) / 4) * 10 * (((int64_t)C + 63) / 64 * 64);
}

int dwconv_nhwc_bwd_dispatch(const void *x, int x_is_bf16, const float *w, const float *bias, const float *dy, int ndir,
                             int64_t dir_stride, const float *dy_extra, void *dx, int dx_bf16, int64_t dxps, float *scratch,
                             float *dw, float *dbias, int batch, int C, int H, int W, int64_t xps, hipStream_t s) {
    if (!x || !w || !dy || !dx || !dw || !scratch) return MS_ERR_NULL;
    if (batch < 0 || C <= 0 || H <= 0 || W <= 0 || xps < C || dxps < C || ndir < 1 || ndir > 8) return MS_ERR_SHAPE;
    if (batch == 0) return MS_OK;
    return x_is_bf16 ? launch_bwd<unsigned short>(x, w, bias, dy, ndir, dir_stride, dy_extra, dx, dx_bf16, dxps, scratch, dw, dbias, batch, C, H, W, xps, s)
                     : launch_bwd<float>(x, w, bias, dy, ndir, dir_stride, dy_extra, dx, dx_bf16, dxps, scratch, dw, dbias, batch, C, H, W, xps, s);
}

}  // namespace ms
