// LayerNorm over the channel axis of a channel-last tensor for gfx950, reading rows IN PLACE from a wider tensor
// (the right half of the block input: `input.chunk(2, dim=-1)[1]` -> `self.ln_1(right)`, MedMamba.py:512-515) and
// writing the dtype the following projection consumes (bf16 under autocast).  The eager path needs a contiguous copy
// of the strided half, the LayerNorm, and a cast: three round trips; here one.  Backward recomputes the statistics
// (nothing saved but the input itself) and reduces dgamma/dbeta per workgroup before one atomic per channel.
// One wave = one pixel (D channels, lane owns channels lane, lane+64, ...), 4 pixels per workgroup per iteration.
#include <hip/hip_runtime.h>
#include "medscan.h"

namespace ms {

constexpr int kLnMaxVPT = 16;          // D <= 1024
constexpr int kLnPixPerWaveBwd = 8;

__device__ __forceinline__ float ln_wave_sum(float v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}
template <typename T> __device__ __forceinline__ float ln_ld(const T *p);
template <> __device__ __forceinline__ float ln_ld<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ln_ld<unsigned short>(const unsigned short *p) { return __builtin_bit_cast(float, (unsigned)*p << 16); }
template <typename T> __device__ __forceinline__ void ln_st(T *p, float v);
template <> __device__ __forceinline__ void ln_st<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void ln_st<unsigned short>(unsigned short *p, float v) { *p = __builtin_bit_cast(unsigned short, (__bf16)v); }

template <int VPT, typename TO>
__global__ void __launch_bounds__(256)
ln_fwd_kernel(const float *__restrict__ x, int64_t xps, const float *__restrict__ gamma, const float *__restrict__ beta,
              float eps, TO *__restrict__ out, int D, int64_t npix) {
    const int lane = threadIdx.x & 63;
    const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= npix) return;
    const float *xp = x + pix * xps;
    float v[VPT];
    float s1 = 0.0f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) { const int c = lane + 64 * j; v[j] = c < D ? xp[c] : 0.0f; s1 += v[j]; }
    const float mean = ln_wave_sum(s1) / (float)D;
    float s2 = 0.0f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) { const float d = (lane + 64 * j < D) ? v[j] - mean : 0.0f; s2 += d * d; }
    const float rstd = rsqrtf(ln_wave_sum(s2) / (float)D + eps);
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int c = lane + 64 * j;
        if (c < D) ln_st(out + pix * D + c, (v[j] - mean) * rstd * gamma[c] + beta[c]);
    }
}

template <int VPT, typename TG>
__global__ void __launch_bounds__(256)
ln_bwd_kernel(const float *__restrict__ x, int64_t xps, const float *__restrict__ gamma, float eps,
              const TG *__restrict__ dout, float *__restrict__ dx, float *__restrict__ dgamma, float *__restrict__ dbeta,
              int D, int64_t npix) {
    __shared__ float red[3][2][kLnMaxVPT * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float gm[VPT], dg[VPT], db[VPT];
#pragma unroll
    for (int j = 0; j < VPT; ++j) { const int c = lane + 64 * j; gm[j] = c < D ? gamma[c] : 0.0f; dg[j] = 0.0f; db[j] = 0.0f; }
    const int64_t first = ((int64_t)blockIdx.x * 4 + wv) * kLnPixPerWaveBwd;
    for (int64_t pix = first; pix < first + kLnPixPerWaveBwd && pix < npix; ++pix) {
        const float *xp = x + pix * xps;
        float v[VPT], g[VPT];
        float s1 = 0.0f;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int c = lane + 64 * j;
            v[j] = c < D ? xp[c] : 0.0f;
            g[j] = c < D ? ln_ld(dout + pix * D + c) : 0.0f;
            s1 += v[j];
        }
        const float mean = ln_wave_sum(s1) / (float)D;
        float s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < VPT; ++j) { const float d = (lane + 64 * j < D) ? v[j] - mean : 0.0f; s2 += d * d; }
        const float rstd = rsqrtf(ln_wave_sum(s2) / (float)D + eps);
        float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const bool in = lane + 64 * j < D;
            v[j] = in ? (v[j] - mean) * rstd : 0.0f;          // normalised value
            dg[j] = fmaf(g[j], v[j], dg[j]);
            db[j] += g[j];
            g[j] *= gm[j];                                     // gradient w.r.t. the normalised value
            m1 += g[j];
            m2 = fmaf(g[j], v[j], m2);
        }
        m1 = ln_wave_sum(m1) / (float)D;
        m2 = ln_wave_sum(m2) / (float)D;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int c = lane + 64 * j;
            if (c < D) dx[pix * D + c] = rstd * (g[j] - m1 - v[j] * m2);
        }
    }
    if (wv > 0) {
#pragma unroll
        for (int j = 0; j < VPT; ++j) { red[wv - 1][0][j * 64 + lane] = dg[j]; red[wv - 1][1][j * 64 + lane] = db[j]; }
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int c = lane + 64 * j;
            if (c < D) {
                atomicAdd(dgamma + c, dg[j] + red[0][0][j * 64 + lane] + red[1][0][j * 64 + lane] + red[2][0][j * 64 + lane]);
                atomicAdd(dbeta + c, db[j] + red[0][1][j * 64 + lane] + red[1][1][j * 64 + lane] + red[2][1][j * 64 + lane]);
            }
        }
    }
}

#define MS_LN_DISPATCH(VPTVAR, CALL)                                                                            \
    if (VPTVAR <= 1) { CALL(1); } else if (VPTVAR <= 2) { CALL(2); } else if (VPTVAR <= 3) { CALL(3); }           \
    else if (VPTVAR <= 4) { CALL(4); } else if (VPTVAR <= 6) { CALL(6); } else if (VPTVAR <= 8) { CALL(8); }      \
    else if (VPTVAR <= 12) { CALL(12); } else { CALL(16); }

int ln_fwd_dispatch(const float *x, int64_t xps, const float *gamma, const float *beta, float eps, void *out,
                    int out_bf16, int64_t npix, int D, hipStream_t s) {
    if (!x || !gamma || !beta || !out) return MS_ERR_NULL;
    if (D <= 0 || D > 64 * kLnMaxVPT || npix < 0 || xps < D) return MS_ERR_SHAPE;
    if (npix == 0) return MS_OK;
    const dim3 grid((unsigned)((npix + 3) / 4)), block(256);
    const int vpt = (D + 63) / 64;
    if (out_bf16) {
#define MS_C(V) hipLaunchKernelGGL((ln_fwd_kernel<V, unsigned short>), grid, block, 0, s, x, xps, gamma, beta, eps, (unsigned short *)out, D, npix)
        MS_LN_DISPATCH(vpt, MS_C)
#undef MS_C
    } else {
#define MS_C(V) hipLaunchKernelGGL((ln_fwd_kernel<V, float>), grid, block, 0, s, x, xps, gamma, beta, eps, (float *)out, D, npix)
        MS_LN_DISPATCH(vpt, MS_C)
#undef MS_C
    }
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int ln_bwd_dispatch(const float *x, int64_t xps, const float *gamma, float eps, const void *dout, int dout_bf16, float *dx,
                    float *dgamma, float *dbeta, int64_t npix, int D, hipStream_t s) {
    if (!x || !gamma || !dout || !dx || !dgamma || !dbeta) return MS_ERR_NULL;
    if (D <= 0 || D > 64 * kLnMaxVPT || npix < 0 || xps < D) return MS_ERR_SHAPE;
    if (npix == 0) return MS_OK;
    const int64_t tasks = (npix + kLnPixPerWaveBwd - 1) / kLnPixPerWaveBwd;
    const dim3 grid((unsigned)((tasks + 3) / 4)), block(256);
    const int vpt = (D + 63) / 64;
    if (dout_bf16) {
#define MS_C(V) hipLaunchKernelGGL((ln_bwd_kernel<V, unsigned short>), grid, block, 0, s, x, xps, gamma, eps, (const unsigned short *)dout, dx, dgamma, dbeta, D, npix)
        MS_LN_DISPATCH(vpt, MS_C)
#undef MS_C
    } else {
#define MS_C(V) hipLaunchKernelGGL((ln_bwd_kernel<V, float>), grid, block, 0, s, x, xps, gamma, eps, (const float *)dout, dx, dgamma, dbeta, D, npix)
        MS_LN_DISPATCH(vpt, MS_C)
#undef MS_C
    }
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
