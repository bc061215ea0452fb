// Fused tail of SS2D for gfx950: cross-merge sum + LayerNorm + SiLU gate (MedMamba.py:476-479)
//   y   = ((y0 + y2) + y1) + y3            the four directions' scan outputs, already in pixel order (ss2d mode)
//   out = (LayerNorm_D(y) * gamma + beta) * silu(z)
// In the reference this is 3 adds, a transpose copy, LayerNorm, SiLU and a multiply: ~9 elementwise / reduction
// kernels forward and ~14 backward, each a full HBM round trip.  Here: one pass forward (16 B + z read, out write per
// element) and one backward (recomputes y and the statistics instead of saving them).
// One wave = one pixel (D channels, lane owns channels lane, lane+64, ...), 4 pixels per workgroup per iteration.
#include <hip/hip_runtime.h>
#include "medscan.h"

namespace ms {

constexpr int kMaxVPT = 16;     // D <= 1024

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}
__device__ __forceinline__ float bf2f(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {           // round to nearest even (NaN stays NaN via hw cvt)
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}
template <typename T> __device__ __forceinline__ float ld(const T *p);
template <> __device__ __forceinline__ float ld<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ld<unsigned short>(const unsigned short *p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st(T *p, float v);
template <> __device__ __forceinline__ void st<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<unsigned short>(unsigned short *p, float v) { *p = f2bf(v); }

template <int VPT, typename TZ, typename TO>
__global__ void __launch_bounds__(256)
ln_gate_fwd_kernel(const float *__restrict__ y4, int64_t sk, const TZ *__restrict__ z, int64_t zps,
                   const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                   TO *__restrict__ out, float *__restrict__ ysum, int D, int64_t npix) {
    const int lane = threadIdx.x & 63;
    const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= npix) return;
    const float *yp = y4 + pix * D;
    float y[VPT];
    float s1 = 0.0f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int c = lane + 64 * j;
        y[j] = c < D ? ((yp[c] + yp[2 * sk + c]) + yp[sk + c]) + yp[3 * sk + c] : 0.0f;
        s1 += y[j];
        if (ysum && c < D) ysum[pix * D + c] = y[j];         // the merged sum, kept for the backward (4 B instead of 16 B per element there)
    }
    const float mean = wave_sum(s1) / (float)D;
    float s2 = 0.0f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) { const float d = (lane + 64 * j < D) ? y[j] - mean : 0.0f; s2 += d * d; }
    const float rstd = rsqrtf(wave_sum(s2) / (float)D + eps);
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int c = lane + 64 * j;
        if (c < D) {
            const float zz = ld(z + pix * zps + c);
            const float yh = (y[j] - mean) * rstd * gamma[c] + beta[c];
            st(out + pix * D + c, yh * (zz / (1.0f + expf(-zz))));
        }
    }
}

template <int PB>
__device__ __forceinline__ void wave_sum_n(float (&v)[PB]) {         // PB interleaved shuffle chains
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        float t[PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) t[q] = __shfl_xor(v[q], s, 64);
#pragma unroll
        for (int q = 0; q < PB; ++q) v[q] += t[q];
    }
}

// Backward: a wave works on PB pixels at a time (loads of all PB pixels first, the
// reduction chains interleaved): one pixel alone is 5 dependent wave reductions behind its global loads.
template <int VPT, int PB, typename TZ, typename TG>
__global__ void __launch_bounds__(256)
ln_gate_bwd_kernel(const float *__restrict__ y4, int64_t sk, const TZ *__restrict__ z, int64_t zps,
                   const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                   const TG *__restrict__ dout, float *__restrict__ dy, TZ *__restrict__ dz,
                   float *__restrict__ dgamma, float *__restrict__ dbeta, int D, int64_t npix, int64_t dzps) {
    __shared__ float red[3][2][kMaxVPT * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float gm[VPT], bt[VPT], dg[VPT], db[VPT];
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int c = lane + 64 * j;
        gm[j] = c < D ? gamma[c] : 0.0f; bt[j] = c < D ? beta[c] : 0.0f; dg[j] = 0.0f; db[j] = 0.0f;
    }
    // persistent waves: wave w takes pixel groups w, w + nwaves, ... and keeps its dgamma/dbeta partial sums in registers
    // over all of them, so the same-address atomics at the end number gridDim.x per channel (they, not the
    // arithmetic, bounded the one-group-per-wave version: 6272 serialized atomics per address at stage 0)
    const int64_t nwaves = (int64_t)gridDim.x * 4, last = npix;
    const float invD = 1.0f / (float)D;
    for (int64_t p0 = ((int64_t)blockIdx.x * 4 + wv) * PB; p0 < last; p0 += nwaves * PB) {
        float y[PB][VPT], zz[PB][VPT], g[PB][VPT], s1[PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int64_t pix = min(p0 + q, last - 1);           // duplicates of the last pixel are computed, not stored
            const float *yp = y4 + pix * D;
            s1[q] = 0.0f;
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                const int c = lane + 64 * j;
                // sk == 0: y4 IS the merged sum the forward kept (wave-uniform branch)
                y[q][j] = c < D ? (sk ? ((yp[c] + yp[2 * sk + c]) + yp[sk + c]) + yp[3 * sk + c] : yp[c]) : 0.0f;
                zz[q][j] = c < D ? ld(z + pix * zps + c) : 0.0f;
                g[q][j] = c < D ? ld(dout + pix * D + c) : 0.0f;
                s1[q] += y[q][j];
            }
        }
        wave_sum_n<PB>(s1);
        float s2[PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            s1[q] *= invD; s2[q] = 0.0f;
#pragma unroll
            for (int j = 0; j < VPT; ++j) { const float d = (lane + 64 * j < D) ? y[q][j] - s1[q] : 0.0f; s2[q] += d * d; }
        }
        wave_sum_n<PB>(s2);
        float m[2 * PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const float rstd = rsqrtf(s2[q] * invD + eps);
            const bool live = p0 + q < last;
            s2[q] = rstd;
            float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                const int c = lane + 64 * j;
                const bool in = c < D;
                const float sg = 1.0f / (1.0f + expf(-zz[q][j]));
                const float yn = in ? (y[q][j] - s1[q]) * rstd : 0.0f;
                const float yh = yn * gm[j] + bt[j];
                if (in && live) st(dz + (p0 + q) * dzps + c, g[q][j] * yh * (sg * (1.0f + zz[q][j] * (1.0f - sg))));
                const float dyh = in ? g[q][j] * (zz[q][j] * sg) : 0.0f;
                if (live) { dg[j] = fmaf(dyh, yn, dg[j]); db[j] += dyh; }
                y[q][j] = yn;                                    // y <- normalised value
                g[q][j] = dyh * gm[j];                           // g <- gradient w.r.t. the normalised value
                m1 += g[q][j];
                m2 = fmaf(g[q][j], yn, m2);
            }
            m[2 * q] = m1; m[2 * q + 1] = m2;
        }
        wave_sum_n<2 * PB>(m);
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            if (p0 + q < last) {
                const float m1 = m[2 * q] * invD, m2 = m[2 * q + 1] * invD;
#pragma unroll
                for (int j = 0; j < VPT; ++j) {
                    const int c = lane + 64 * j;
                    if (c < D) dy[(p0 + q) * D + c] = s2[q] * (g[q][j] - m1 - y[q][j] * m2);
                }
            }
        }
    }
    // gamma / beta gradients: combine the block's 4 waves in LDS, then one atomic per (block, channel)
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

template <typename TZ, typename TO>
static int launch_fwd(const float *y4, int64_t sk, const void *z, int64_t zps, const float *gamma, const float *beta,
                      float eps, void *out, float *ysum, int D, int64_t npix, hipStream_t s) {
    const dim3 grid((unsigned)((npix + 3) / 4)), block(256);
    const int vpt = (D + 63) / 64;
#define MS_L(V) hipLaunchKernelGGL((ln_gate_fwd_kernel<V, TZ, TO>), grid, block, 0, s, y4, sk, (const TZ *)z, zps, gamma, beta, eps, (TO *)out, ysum, D, npix)
    if (vpt <= 1) MS_L(1); else if (vpt <= 2) MS_L(2); else if (vpt <= 3) MS_L(3); else if (vpt <= 4) MS_L(4);
    else if (vpt <= 6) MS_L(6); else if (vpt <= 8) MS_L(8); else if (vpt <= 12) MS_L(12); else MS_L(16);
#undef MS_L
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int ln_gate_fwd_dispatch(const float *y4, int64_t sk, const void *z, int z_bf16, int64_t zps, const float *gamma,
                         const float *beta, float eps, void *out, int out_bf16, float *ysum, int64_t npix, int D, hipStream_t s) {
    if (!y4 || !z || !gamma || !beta || !out) return MS_ERR_NULL;
    if (D <= 0 || D > 64 * kMaxVPT || npix < 0 || zps < D) return MS_ERR_SHAPE;
    if (npix == 0) return MS_OK;
    if (z_bf16) return out_bf16 ? launch_fwd<unsigned short, unsigned short>(y4, sk, z, zps, gamma, beta, eps, out, ysum, D, npix, s)
                                : launch_fwd<unsigned short, float>(y4, sk, z, zps, gamma, beta, eps, out, ysum, D, npix, s);
    return out_bf16 ? launch_fwd<float, unsigned short>(y4, sk, z, zps, gamma, beta, eps, out, ysum, D, npix, s)
                    : launch_fwd<float, float>(y4, sk, z, zps, gamma, beta, eps, out, ysum, D, npix, s);
}

template <typename TZ, typename TG>
static int launch_bwd(const float *y4, int64_t sk, const void *z, int64_t zps, const float *gamma, const float *beta,
                      float eps, const void *dout, float *dy, void *dz, int64_t dzps, float *dgamma, float *dbeta, int D,
                      int64_t npix, hipStream_t s) {
    const int vpt = (D + 63) / 64;
    const int pb = vpt <= 2 ? 4 : vpt <= 4 ? 2 : 1;                 // = MS_PB of the dispatched VPT bucket
    const int64_t tasks = (npix + pb - 1) / pb;                     // pixel groups
    const int64_t blocks = (tasks + 3) / 4;
    const int64_t cap = npix >= 32768 ? 1024 : 512;      // measured optimum (tools/bench_ln.py), see ln.hip
    const dim3 grid((unsigned)(blocks < cap ? blocks : cap)), block(256);      // persistent
#define MS_PB(V) ((V) <= 2 ? 4 : (V) <= 4 ? 2 : 1)
#define MS_L(V) hipLaunchKernelGGL((ln_gate_bwd_kernel<V, MS_PB(V), TZ, TG>), grid, block, 0, s, y4, sk, (const TZ *)z, zps, gamma, beta, eps, (const TG *)dout, dy, (TZ *)dz, dgamma, dbeta, D, npix, dzps)
    if (vpt <= 1) MS_L(1); else if (vpt <= 2) MS_L(2); else if (vpt <= 3) MS_L(3); else if (vpt <= 4) MS_L(4);
    else if (vpt <= 6) MS_L(6); else if (vpt <= 8) MS_L(8); else if (vpt <= 12) MS_L(12); else MS_L(16);
#undef MS_L
#undef MS_PB
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int ln_gate_bwd_dispatch(const float *y4, int64_t sk, const void *z, int z_bf16, int64_t zps, const float *gamma,
                         const float *beta, float eps, const void *dout, int dout_bf16, float *dy, void *dz, int64_t dzps,
                         float *dgamma, float *dbeta, int64_t npix, int D, hipStream_t s) {
    if (!y4 || !z || !gamma || !beta || !dout || !dy || !dz || !dgamma || !dbeta) return MS_ERR_NULL;
    if (D <= 0 || D > 64 * kMaxVPT || npix < 0 || zps < D || dzps < D) return MS_ERR_SHAPE;
    if (npix == 0) return MS_OK;
    if (z_bf16) return dout_bf16 ? launch_bwd<unsigned short, unsigned short>(y4, sk, z, zps, gamma, beta, eps, dout, dy, dz, dzps, dgamma, dbeta, D, npix, s)
                                 : launch_bwd<unsigned short, float>(y4, sk, z, zps, gamma, beta, eps, dout, dy, dz, dzps, dgamma, dbeta, D, npix, s);
    return dout_bf16 ? launch_bwd<float, unsigned short>(y4, sk, z, zps, gamma, beta, eps, dout, dy, dz, dzps, dgamma, dbeta, D, npix, s)
                     : launch_bwd<float, float>(y4, sk, z, zps, gamma, beta, eps, dout, dy, dz, dzps, dgamma, dbeta, D, npix, s);
}

}  // namespace ms
