// Gated RMS normalisation of the SSD blocks for gfx950 (mamba_ssm 2.2.2 `RMSNormGated(norm_before_gate=False)` as built at
// CNN_Mamba.py:430-431 and applied at :554-555; one group, no bias), with the cross-merge sum in front of it:
//   y   = y0                       (ndir == 1)      or  ((y0 + y2) + y1) + y3   (ndir == 4: the scan's four slabs, :542-552)
//   g   = y * silu(z);   out = g * rsqrt(mean_D(g^2) + eps) * weight
// In torch this is 8 elementwise / reduction kernels forward and ~20 backward, each a full HBM round trip over a
// (pixels, D) fp32 tensor.  One wave = one pixel (lane owns channels lane, lane+64, ...): one pass forward, one backward
// (y, the gate and the statistic are recomputed from the operands instead of being saved).
#include <hip/hip_runtime.h>
#include "medscan.h"

namespace ms {
namespace {

constexpr int kMaxVPT = 16;     // D <= 1024

__device__ __forceinline__ float bf2f(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
template <typename T> __device__ __forceinline__ float ld(const T *p);
template <> __device__ __forceinline__ float ld<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ld<unsigned short>(const unsigned short *p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st(T *p, float v);
template <> __device__ __forceinline__ void st<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<unsigned short>(unsigned short *p, float v) {
    *p = __builtin_bit_cast(unsigned short, (__bf16)v);
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
__device__ __forceinline__ float merged(const float *yp, int64_t sk, int ndir, int c) {
    return ndir == 4 ? ((yp[c] + yp[2 * sk + c]) + yp[sk + c]) + yp[3 * sk + c] : yp[c];
}

}  // namespace

template <int VPT, typename TZ, typename TO>
__global__ void __launch_bounds__(256)
rms_gate_fwd_kernel(const float *__restrict__ y4, int64_t sk, int ndir, const TZ *__restrict__ z, int64_t zps,
                    const float *__restrict__ weight, float eps, TO *__restrict__ out, int D, int64_t npix) {
    const int lane = threadIdx.x & 63;
    const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= npix) return;
    const float *yp = y4 + pix * D;
    float g[VPT], s2[1] = {0.0f};
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int c = lane + 64 * j;
        if (c < D) {
            const float zz = ld(z + pix * zps + c);
            g[j] = merged(yp, sk, ndir, c) * (zz / (1.0f + expf(-zz)));
        } else {
            g[j] = 0.0f;
        }
        s2[0] = fmaf(g[j], g[j], s2[0]);
    }
    wave_sum_n<1>(s2);
    const float r = rsqrtf(s2[0] / (float)D + eps);
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int c = lane + 64 * j;
        if (c < D) st(out + pix * D + c, g[j] * r * weight[c]);
    }
}

// Backward, persistent waves (dweight partial sums stay in registers over all of a wave's pixels), PB pixels in flight:
//   v = dout * w;  m = mean(g * v);  dg = r * v - g * r^3 * m;  dy = dg * silu(z);  dz = dg * y * silu'(z);  dw += dout * g * r
template <int VPT, int PB, typename TZ, typename TG>
__global__ void __launch_bounds__(256)
rms_gate_bwd_kernel(const float *__restrict__ y4, int64_t sk, int ndir, const TZ *__restrict__ z, int64_t zps,
                    const float *__restrict__ weight, float eps, const TG *__restrict__ dout, float *__restrict__ dy,
                    TZ *__restrict__ dz, int64_t dzps, float *__restrict__ dweight, int D, int64_t npix) {
    __shared__ float red[3][kMaxVPT * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float w[VPT], dw[VPT];
#pragma unroll
    for (int j = 0; j < VPT; ++j) { const int c = lane + 64 * j; w[j] = c < D ? weight[c] : 0.0f; dw[j] = 0.0f; }
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const float invD = 1.0f / (float)D;
    for (int64_t p0 = ((int64_t)blockIdx.x * 4 + wv) * PB; p0 < npix; p0 += nwaves * PB) {
        float y[PB][VPT], zz[PB][VPT], go[PB][VPT], s2[PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int64_t pix = min(p0 + q, npix - 1);           // duplicates of the last pixel are computed, not stored
            const float *yp = y4 + pix * D;
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                const int c = lane + 64 * j;
                y[q][j] = c < D ? merged(yp, sk, ndir, c) : 0.0f;
                zz[q][j] = c < D ? ld(z + pix * zps + c) : 0.0f;
                go[q][j] = c < D ? ld(dout + pix * D + c) : 0.0f;
            }
        }
        float sg[PB][VPT], m[PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            s2[q] = 0.0f;
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                sg[q][j] = 1.0f / (1.0f + expf(-zz[q][j]));
                const float g = y[q][j] * zz[q][j] * sg[q][j];
                s2[q] = fmaf(g, g, s2[q]);
            }
        }
        wave_sum_n<PB>(s2);
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            s2[q] = rsqrtf(s2[q] * invD + eps);                  // r
            m[q] = 0.0f;
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                const float g = y[q][j] * zz[q][j] * sg[q][j];
                m[q] = fmaf(g, go[q][j] * w[j], m[q]);
                if (p0 + q < npix) dw[j] = fmaf(go[q][j], g * s2[q], dw[j]);
            }
        }
        wave_sum_n<PB>(m);
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            if (p0 + q < npix) {
                const float r = s2[q], k = r * r * r * m[q] * invD;
#pragma unroll
                for (int j = 0; j < VPT; ++j) {
                    const int c = lane + 64 * j;
                    if (c < D) {
                        const float silu = zz[q][j] * sg[q][j];
                        const float dg = r * go[q][j] * w[j] - y[q][j] * silu * k;
                        dy[(p0 + q) * D + c] = dg * silu;
                        st(dz + (p0 + q) * dzps + c, dg * y[q][j] * (sg[q][j] * (1.0f + zz[q][j] * (1.0f - sg[q][j]))));
                    }
                }
            }
        }
    }
    if (wv > 0) {
#pragma unroll
        for (int j = 0; j < VPT; ++j) red[wv - 1][j * 64 + lane] = dw[j];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int c = lane + 64 * j;
            if (c < D) atomicAdd(dweight + c, dw[j] + red[0][j * 64 + lane] + red[1][j * 64 + lane] + red[2][j * 64 + lane]);
        }
    }
}

template <typename TZ, typename TO>
static int launch_fwd(const float *y4, int64_t sk, int ndir, const void *z, int64_t zps, const float *w, float eps, void *out,
                      int D, int64_t npix, hipStream_t s) {
    const dim3 grid((unsigned)((npix + 3) / 4)), block(256);
    const int vpt = (D + 63) / 64;
#define MS_L(V) hipLaunchKernelGGL((rms_gate_fwd_kernel<V, TZ, TO>), grid, block, 0, s, y4, sk, ndir, (const TZ *)z, zps, w, eps, (TO *)out, D, npix)
    if (vpt <= 1) MS_L(1); else if (vpt <= 2) MS_L(2); else if (vpt <= 4) MS_L(4); else if (vpt <= 8) MS_L(8); else MS_L(16);
#undef MS_L
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

template <typename TZ, typename TG>
static int launch_bwd(const float *y4, int64_t sk, int ndir, const void *z, int64_t zps, const float *w, float eps,
                      const void *dout, float *dy, void *dz, int64_t dzps, float *dweight, int D, int64_t npix, hipStream_t s) {
    const int vpt = (D + 63) / 64;
    const int pb = vpt <= 4 ? 2 : 1;
    const int64_t blocks = ((npix + pb - 1) / pb + 3) / 4;
    const int64_t cap = npix >= 32768 ? 1024 : 512;               // as ln_gate.hip: bounds the same-address atomics at the end
    const dim3 grid((unsigned)(blocks < cap ? blocks : cap)), block(256);
#define MS_L(V, P) hipLaunchKernelGGL((rms_gate_bwd_kernel<V, P, TZ, TG>), grid, block, 0, s, y4, sk, ndir, (const TZ *)z, zps, w, eps, (const TG *)dout, dy, (TZ *)dz, dzps, dweight, D, npix)
    if (vpt <= 1) MS_L(1, 2); else if (vpt <= 2) MS_L(2, 2); else if (vpt <= 4) MS_L(4, 2); else if (vpt <= 8) MS_L(8, 1); else MS_L(16, 1);
#undef MS_L
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int rms_gate_fwd_dispatch(const float *y4, int64_t sk, int ndir, const void *z, int z_bf16, int64_t zps, const float *w, float eps,
                          void *out, int out_bf16, int64_t npix, int D, hipStream_t s) {
    if (!y4 || !z || !w || !out) return MS_ERR_NULL;
    if (D <= 0 || D > 64 * kMaxVPT || npix < 0 || zps < D || (ndir != 1 && ndir != 4)) return MS_ERR_SHAPE;
    if (npix == 0) return MS_OK;
    if (z_bf16) return out_bf16 ? launch_fwd<unsigned short, unsigned short>(y4, sk, ndir, z, zps, w, eps, out, D, npix, s)
                                : launch_fwd<unsigned short, float>(y4, sk, ndir, z, zps, w, eps, out, D, npix, s);
    return out_bf16 ? launch_fwd<float, unsigned short>(y4, sk, ndir, z, zps, w, eps, out, D, npix, s)
                    : launch_fwd<float, float>(y4, sk, ndir, z, zps, w, eps, out, D, npix, s);
}

int rms_gate_bwd_dispatch(const float *y4, int64_t sk, int ndir, const void *z, int z_bf16, int64_t zps, const float *w, float eps,
                          const void *dout, int dout_bf16, float *dy, void *dz, int64_t dzps, float *dweight, int64_t npix, int D,
                          hipStream_t s) {
    if (!y4 || !z || !w || !dout || !dy || !dz || !dweight) return MS_ERR_NULL;
    if (D <= 0 || D > 64 * kMaxVPT || npix < 0 || zps < D || dzps < D || (ndir != 1 && ndir != 4)) return MS_ERR_SHAPE;
    if (npix == 0) return MS_OK;
    if (z_bf16) return dout_bf16 ? launch_bwd<unsigned short, unsigned short>(y4, sk, ndir, z, zps, w, eps, dout, dy, dz, dzps, dweight, D, npix, s)
                                 : launch_bwd<unsigned short, float>(y4, sk, ndir, z, zps, w, eps, dout, dy, dz, dzps, dweight, D, npix, s);
    return dout_bf16 ? launch_bwd<float, unsigned short>(y4, sk, ndir, z, zps, w, eps, dout, dy, dz, dzps, dweight, D, npix, s)
                     : launch_bwd<float, float>(y4, sk, ndir, z, zps, w, eps, dout, dy, dz, dzps, dweight, D, npix, s);
}

}  // namespace ms
