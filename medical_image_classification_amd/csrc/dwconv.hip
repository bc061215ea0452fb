// Depthwise 3x3 conv (pad 1) + bias + SiLU, NCHW fp32, for gfx950.  In the reference this is a cuDNN
// depthwise conv followed by a separate SiLU kernel (/root/reference/MedMamba.py:285-294,473):
// two passes forward, ~five backward.  Here: one pass each way; a (b,c) plane (+halo) lives in LDS.
//   fwd: y = silu(conv3x3(x, w[c]) + bias[c])
//   bwd: dpre = dy * silu'(pre) (pre recomputed), dx = conv3x3^T(dpre, w[c]),
//        dw[c,k] += sum dpre*x_shift_k, dbias[c] += sum dpre      (atomics over batch)
#include <hip/hip_runtime.h>
#include "medscan.h"

namespace ms {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ void load_plane_halo(float *t, const float *src, int H, int W, int pitch) {
    const int HP = H + 2, WP = W + 2;
    for (int i = threadIdx.x; i < HP * WP; i += blockDim.x) {
        const int r = i / WP, cidx = i % WP;
        const int h = r - 1, w = cidx - 1;
        float v = 0.0f;
        if (h >= 0 && h < H && w >= 0 && w < W) v = src[h * W + w];
        t[r * pitch + cidx] = v;
    }
}

__global__ void __launch_bounds__(256)
dwconv3x3_silu_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                          float *__restrict__ y, int C, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float t[];
    const int pitch = W + 3;
    const int c = blockIdx.x % C;
    const int64_t plane = (int64_t)blockIdx.x * H * W;
    load_plane_halo(t, x + plane, H, W, pitch);
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = w[c * 9 + i];
    const float bv = bias ? bias[c] : 0.0f;
    __syncthreads();
    for (int p = threadIdx.x; p < H * W; p += blockDim.x) {
        const int h = p / W, ww = p % W;
        float acc = bv;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc = fmaf(k[i * 3 + j], t[(h + i) * pitch + ww + j], acc);
        y[plane + p] = acc * sigmoidf_(acc);
    }
}

__global__ void __launch_bounds__(256)
dwconv3x3_silu_bwd_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                          const float *__restrict__ dy, float *__restrict__ dx, float *__restrict__ dw,
                          float *__restrict__ dbias, int C, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int pitch = W + 3;
    float *tx = smem;                          // x with halo
    float *tg = smem + (H + 2) * pitch;        // dpre with halo
    __shared__ float red[4][10];
    const int c = blockIdx.x % C;
    const int64_t plane = (int64_t)blockIdx.x * H * W;
    load_plane_halo(tx, x + plane, H, W, pitch);
    for (int i = threadIdx.x; i < (H + 2) * pitch; i += blockDim.x) tg[i] = 0.0f;
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = w[c * 9 + i];
    const float bv = bias ? bias[c] : 0.0f;
    __syncthreads();
    float acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = 0.0f;
    for (int p = threadIdx.x; p < H * W; p += blockDim.x) {
        const int h = p / W, ww = p % W;
        float pre = bv;
        float xv[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                xv[i * 3 + j] = tx[(h + i) * pitch + ww + j];
                pre = fmaf(k[i * 3 + j], xv[i * 3 + j], pre);
            }
        const float s = sigmoidf_(pre);
        const float dpre = dy[plane + p] * (s * (1.0f + pre * (1.0f - s)));
        tg[(h + 1) * pitch + ww + 1] = dpre;
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[i] = fmaf(dpre, xv[i], acc[i]);
        acc[9] += dpre;
    }
    __syncthreads();
    // dx[h,w] = sum_{i,j} dpre[h-i+1, w-j+1] * k[i,j]
    for (int p = threadIdx.x; p < H * W; p += blockDim.x) {
        const int h = p / W, ww = p % W;
        float a = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) a = fmaf(k[i * 3 + j], tg[(h + 2 - i) * pitch + ww + 2 - j], a);
        dx[plane + p] = a;
    }
    // block reduction of the 10 parameter-gradient partials
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        float v = acc[i];
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s, 64);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        float v = 0.0f;
        for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) v += red[wv][threadIdx.x];
        if (threadIdx.x < 9) atomicAdd(dw + c * 9 + threadIdx.x, v);
        else if (dbias) atomicAdd(dbias + c, v);
    }
}

static int check(int batch, int C, int H, int W) {
    if (batch < 0 || C <= 0 || H <= 0 || W <= 0) return MS_ERR_SHAPE;
    return MS_OK;
}

int dwconv_fwd_dispatch(const float *x, const float *w, const float *bias, float *y,
                        int batch, int C, int H, int W, hipStream_t s) {
    if (!x || !w || !y) return MS_ERR_NULL;
    int rc = check(batch, C, H, W); if (rc) return rc;
    const size_t smem = sizeof(float) * (size_t)(H + 2) * (W + 3);
    if (smem > 160 * 1024) return MS_ERR_UNSUPPORTED;
    if (batch == 0) return MS_OK;
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)dwconv3x3_silu_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const int threads = H * W >= 256 ? 256 : 64;
    hipLaunchKernelGGL(dwconv3x3_silu_fwd_kernel, dim3(batch * C), dim3(threads), smem, s, x, w, bias, y, C, H, W);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int dwconv_bwd_dispatch(const float *x, const float *w, const float *bias, const float *dy, float *dx,
                        float *dw, float *dbias, int batch, int C, int H, int W, hipStream_t s) {
    if (!x || !w || !dy || !dx || !dw) return MS_ERR_NULL;
    int rc = check(batch, C, H, W); if (rc) return rc;
    const size_t smem = sizeof(float) * 2 * (size_t)(H + 2) * (W + 3);
    if (smem > 150 * 1024) return MS_ERR_UNSUPPORTED;
    if (batch == 0) return MS_OK;
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)dwconv3x3_silu_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const int threads = H * W >= 256 ? 256 : 64;
    hipLaunchKernelGGL(dwconv3x3_silu_bwd_kernel, dim3(batch * C), dim3(threads), smem, s, x, w, bias, dy, dx, dw, dbias, C, H, W);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
