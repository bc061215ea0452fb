// LayerNorm over the channel axis of a channel-last tensor for gfx950, reading rows IN PLACE from a wider tensor
// (the right half of the block input: `input.chunk(2, dim=-1)[1]` -> `self.ln_1(right)`, MedMamba.py:512-515) and
// writing the dtype the following projection consumes (bf16 under autocast).  The eager path needs a contiguous copy
// of the strided half, the LayerNorm, and a cast: three round trips; here one.  Backward recomputes the statistics
// (nothing saved but the input itself) and reduces dgamma/dbeta per workgroup before one atomic per channel.
// One wave = one pixel (D channels, lane owns channels lane, lane+64, ...), 4 pixels per workgroup per iteration.
#include <hip/hip_runtime.h>
#include "medscan.h"
#include "ln_common.h"

namespace ms {

constexpr int kLnMaxVPT = 32;          // D <= 2048

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

// sums of PB independent values at once: the PB shuffle chains overlap instead of running back to back
template <int PB>
__device__ __forceinline__ void ln_wave_sum_n(float (&v)[PB]) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        float t[PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) t[q] = __shfl_xor(v[q], s, 64);
#pragma unroll
        for (int q = 0; q < PB; ++q) v[q] += t[q];
    }
}

// Backward: a wave works on PB pixels at a time (all loads of the PB pixels are
// issued before the first reduction, the PB reduction chains are interleaved): the kernel is latency-bound otherwise
// (one pixel = 5 dependent wave reductions behind a global load).
template <int VPT, int PB, typename TG>
__global__ void __launch_bounds__(256)
ln_bwd_kernel(const float *__restrict__ x, int64_t xps, const float *__restrict__ gamma, float eps,
              const TG *__restrict__ dout, float *__restrict__ dx, float *__restrict__ dgamma, float *__restrict__ dbeta,
              int D, int64_t npix) {
    __shared__ float red[3][2][kLnMaxVPT * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float gm[VPT], dg[VPT], db[VPT];
#pragma unroll
    for (int j = 0; j < VPT; ++j) { const int c = lane + 64 * j; gm[j] = c < D ? gamma[c] : 0.0f; dg[j] = 0.0f; db[j] = 0.0f; }
    // persistent waves: wave w takes pixel groups w, w + nwaves, ... and keeps its dgamma/dbeta partial sums in registers
    // over all of them, so the same-address atomics at the end number gridDim.x per channel (they, not the
    // arithmetic, bounded the one-group-per-wave version: 6272 serialized atomics per address at stage 0)
    const int64_t nwaves = (int64_t)gridDim.x * 4, last = npix;
    const float invD = 1.0f / (float)D;
    for (int64_t p0 = ((int64_t)blockIdx.x * 4 + wv) * PB; p0 < last; p0 += nwaves * PB) {
        float v[PB][VPT], g[PB][VPT], s1[PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int64_t pix = min(p0 + q, last - 1);           // duplicates of the last pixel are computed, not stored
            s1[q] = 0.0f;
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                const int c = lane + 64 * j;
                v[q][j] = c < D ? x[pix * xps + c] : 0.0f;
                g[q][j] = c < D ? ln_ld(dout + pix * D + c) : 0.0f;
                s1[q] += v[q][j];
            }
        }
        ln_wave_sum_n<PB>(s1);
        float s2[PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            s1[q] *= invD; s2[q] = 0.0f;
#pragma unroll
            for (int j = 0; j < VPT; ++j) { const float d = (lane + 64 * j < D) ? v[q][j] - s1[q] : 0.0f; s2[q] += d * d; }
        }
        ln_wave_sum_n<PB>(s2);
        float m[2 * PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const float rstd = rsqrtf(s2[q] * invD + eps);
            const bool live = p0 + q < last;
            s2[q] = rstd;
            float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                const bool in = lane + 64 * j < D;
                v[q][j] = in ? (v[q][j] - s1[q]) * rstd : 0.0f;          // normalised value
                if (live) { dg[j] = fmaf(g[q][j], v[q][j], dg[j]); db[j] += g[q][j]; }
                g[q][j] *= gm[j];                                         // gradient w.r.t. the normalised value
                m1 += g[q][j];
                m2 = fmaf(g[q][j], v[q][j], m2);
            }
            m[2 * q] = m1; m[2 * q + 1] = m2;
        }
        ln_wave_sum_n<2 * PB>(m);
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            if (p0 + q < last) {
                const float m1 = m[2 * q] * invD, m2 = m[2 * q + 1] * invD;
#pragma unroll
                for (int j = 0; j < VPT; ++j) {
                    const int c = lane + 64 * j;
                    if (c < D) dx[(p0 + q) * D + c] = s2[q] * (g[q][j] - m1 - v[q][j] * m2);
                }
            }
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

// ---- sub-wave pixel groups, 16-byte accesses, VALU all-reduces (ln_common.h): the kernels every aligned call takes ------------
// TAPS: the row of output token (b, h2, w2) is the concatenation of its 2 x 2 input pixels' channels in the order (0,0), (1,0), (0,1),
// (1,1) -- PatchMerging2D's gather (MedMamba.py:196-200) as an addressing mode of the LayerNorm behind it (and of that LayerNorm's
// dx store: every input pixel belongs to exactly one token, so the scatter is the same map), instead of a permuting copy each way.
struct LnTaps { int h2n, w2n, cin; };
template <bool TAPS>
__device__ __forceinline__ int64_t ln_row_base(int64_t pix, int64_t xps, const LnTaps &tp, int64_t (&base)[4]) {
    if constexpr (!TAPS) { base[0] = pix * xps; return 0; }
    const int w2 = (int)(pix % tp.w2n);
    const int64_t t = pix / tp.w2n;
    const int h2 = (int)(t % tp.h2n);
    const int64_t b = t / tp.h2n;
    const int64_t row0 = (b * 2 * tp.h2n + 2 * h2) * (2 * tp.w2n) + 2 * w2;      // pixel (2 h2, 2 w2)
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) base[tap] = (row0 + (tap & 1) * (2 * tp.w2n) + (tap >> 1)) * tp.cin;
    return 0;
}
template <bool TAPS>
__device__ __forceinline__ int64_t ln_off(const int64_t (&base)[4], int c, const LnTaps &tp) {
    if constexpr (!TAPS) return base[0] + c;
    const int tap = c / tp.cin;
    return base[tap] + (c - tap * tp.cin);
}
template <int LPP, int V4, int PB, typename TO, bool TAPS = false>
__global__ void __launch_bounds__(256)
ln_fwd_sub_kernel(const float *__restrict__ x, int64_t xps, const float *__restrict__ gamma, const float *__restrict__ beta,
                  float eps, TO *__restrict__ out, int D, int64_t npix, LnTaps tp) {
    constexpr int PW = 64 / LPP;
    const int lane = threadIdx.x & 63, lip = lane % LPP, sub = lane / LPP;
    const int64_t p0 = (((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * PB) * PW;
    if (p0 >= npix) return;
    const float invD = 1.0f / (float)D;
    float4 v[PB][V4];
    float s1[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int64_t pix = min(p0 + q * PW + sub, npix - 1);
        int64_t rb[4];
        ln_row_base<TAPS>(pix, xps, tp, rb);
        s1[q] = 0.0f;
#pragma unroll
        for (int j = 0; j < V4; ++j) {
            const int c = 4 * (lip + LPP * j);
            v[q][j] = c < D ? ld4f(x + ln_off<TAPS>(rb, c, tp)) : make_float4(0.f, 0.f, 0.f, 0.f);
            s1[q] += sum4(v[q][j]);
        }
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) s1[q] = group_allsum<LPP>(s1[q], lane) * invD;
    float s2[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        s2[q] = 0.0f;
#pragma unroll
        for (int j = 0; j < V4; ++j) {
            if (4 * (lip + LPP * j) < D) {
                const float a = v[q][j].x - s1[q], b = v[q][j].y - s1[q], c = v[q][j].z - s1[q], d = v[q][j].w - s1[q];
                s2[q] += (a * a + b * b) + (c * c + d * d);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) s2[q] = rsqrtf(group_allsum<LPP>(s2[q], lane) * invD + eps);
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int64_t pix = p0 + q * PW + sub;
        if (pix >= npix) continue;
#pragma unroll
        for (int j = 0; j < V4; ++j) {
            const int c = 4 * (lip + LPP * j);
            if (c < D) {
                const float4 g = ld4f(gamma + c), b = ld4f(beta + c);
                st4f(out + pix * D + c, make_float4((v[q][j].x - s1[q]) * s2[q] * g.x + b.x, (v[q][j].y - s1[q]) * s2[q] * g.y + b.y,
                                                    (v[q][j].z - s1[q]) * s2[q] * g.z + b.z, (v[q][j].w - s1[q]) * s2[q] * g.w + b.w));
            }
        }
    }
}

template <int LPP, int V4, int PB, typename TG, bool TAPS = false>
__global__ void __launch_bounds__(256)
ln_bwd_sub_kernel(const float *__restrict__ x, int64_t xps, const float *__restrict__ gamma, float eps,
                  const TG *__restrict__ dout, float *__restrict__ dx, float *__restrict__ dgamma, float *__restrict__ dbeta,
                  int D, int64_t npix, LnTaps tp) {
    constexpr int PW = 64 / LPP, NC = 4 * V4 * LPP;                     // channel slots of a group
    __shared__ __attribute__((aligned(16))) float red[4][2][NC];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lip = lane % LPP, sub = lane / LPP;
    float4 gm[V4], dg[V4], db[V4];
#pragma unroll
    for (int j = 0; j < V4; ++j) {
        const int c = 4 * (lip + LPP * j);
        gm[j] = c < D ? ld4f(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        dg[j] = make_float4(0.f, 0.f, 0.f, 0.f); db[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // persistent waves (the dgamma / dbeta partial sums stay in registers over all of a wave's pixels)
    const int64_t step = (int64_t)gridDim.x * 4 * PB * PW;
    const float invD = 1.0f / (float)D;
    for (int64_t p0 = ((int64_t)blockIdx.x * 4 + wv) * PB * PW; p0 < npix; p0 += step) {
        float4 v[PB][V4], g[PB][V4];
        float s1[PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int64_t pix = min(p0 + q * PW + sub, npix - 1);       // duplicates of the last pixel are computed, not stored
            int64_t rb[4];
            ln_row_base<TAPS>(pix, xps, tp, rb);
            s1[q] = 0.0f;
#pragma unroll
            for (int j = 0; j < V4; ++j) {
                const int c = 4 * (lip + LPP * j);
                const bool in = c < D;
                v[q][j] = in ? ld4f(x + ln_off<TAPS>(rb, c, tp)) : make_float4(0.f, 0.f, 0.f, 0.f);
                g[q][j] = in ? ld4f(dout + pix * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                s1[q] += sum4(v[q][j]);
            }
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) s1[q] = group_allsum<LPP>(s1[q], lane) * invD;
        float s2[PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            s2[q] = 0.0f;
#pragma unroll
            for (int j = 0; j < V4; ++j) {
                if (4 * (lip + LPP * j) < D) {
                    const float a = v[q][j].x - s1[q], b = v[q][j].y - s1[q], c = v[q][j].z - s1[q], d = v[q][j].w - s1[q];
                    s2[q] += (a * a + b * b) + (c * c + d * d);
                }
            }
        }
        float m1[PB], m2[PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const float rstd = rsqrtf(group_allsum<LPP>(s2[q], lane) * invD + eps);
            const bool live = p0 + q * PW + sub < npix;
            s2[q] = rstd;
            m1[q] = 0.0f; m2[q] = 0.0f;
#pragma unroll
            for (int j = 0; j < V4; ++j) {
                const bool in = 4 * (lip + LPP * j) < D;
                float4 &vv = v[q][j], &gg = g[q][j];
                vv.x = in ? (vv.x - s1[q]) * rstd : 0.0f; vv.y = in ? (vv.y - s1[q]) * rstd : 0.0f;      // normalised values
                vv.z = in ? (vv.z - s1[q]) * rstd : 0.0f; vv.w = in ? (vv.w - s1[q]) * rstd : 0.0f;
                if (live) {
                    dg[j].x = fmaf(gg.x, vv.x, dg[j].x); dg[j].y = fmaf(gg.y, vv.y, dg[j].y);
                    dg[j].z = fmaf(gg.z, vv.z, dg[j].z); dg[j].w = fmaf(gg.w, vv.w, dg[j].w);
                    db[j].x += gg.x; db[j].y += gg.y; db[j].z += gg.z; db[j].w += gg.w;
                }
                gg.x *= gm[j].x; gg.y *= gm[j].y; gg.z *= gm[j].z; gg.w *= gm[j].w;                      // gradient w.r.t. the normalised value
                m1[q] += sum4(gg);
                m2[q] += (gg.x * vv.x + gg.y * vv.y) + (gg.z * vv.z + gg.w * vv.w);
            }
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) { m1[q] = group_allsum<LPP>(m1[q], lane) * invD; m2[q] = group_allsum<LPP>(m2[q], lane) * invD; }
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int64_t pix = p0 + q * PW + sub;
            if (pix >= npix) continue;
            int64_t rb[4];
            ln_row_base<TAPS>(pix, (int64_t)D, tp, rb);                 // dx: (npix, D) rows, or scattered back to the input's pixels
#pragma unroll
            for (int j = 0; j < V4; ++j) {
                const int c = 4 * (lip + LPP * j);
                if (c < D) {
                    const float4 vv = v[q][j], gg = g[q][j];
                    st4f(dx + ln_off<TAPS>(rb, c, tp), make_float4(s2[q] * (gg.x - m1[q] - vv.x * m2[q]), s2[q] * (gg.y - m1[q] - vv.y * m2[q]),
                                                       s2[q] * (gg.z - m1[q] - vv.z * m2[q]), s2[q] * (gg.w - m1[q] - vv.w * m2[q])));
                }
            }
        }
    }
    // gamma / beta gradients: the wave's pixel groups (registers), the block's 4 waves (LDS), one atomic per (block, channel)
#pragma unroll
    for (int j = 0; j < V4; ++j) {
        dg[j].x = across_groups<LPP>(dg[j].x, lane); dg[j].y = across_groups<LPP>(dg[j].y, lane);
        dg[j].z = across_groups<LPP>(dg[j].z, lane); dg[j].w = across_groups<LPP>(dg[j].w, lane);
        db[j].x = across_groups<LPP>(db[j].x, lane); db[j].y = across_groups<LPP>(db[j].y, lane);
        db[j].z = across_groups<LPP>(db[j].z, lane); db[j].w = across_groups<LPP>(db[j].w, lane);
    }
    if (sub == 0) {
#pragma unroll
        for (int j = 0; j < V4; ++j) { st4f(&red[wv][0][4 * (lip + LPP * j)], dg[j]); st4f(&red[wv][1][4 * (lip + LPP * j)], db[j]); }
    }
    __syncthreads();
    // consecutive threads -> consecutive channels: a wave's atomics cover whole 256-byte segments (4 per lane at 16-byte strides
    // quadrupled the cache lines per instruction, and the closing burst of same-address atomics is what bounds this kernel)
    for (int c = threadIdx.x; c < D; c += 256) {
        atomicAdd(dgamma + c, (red[0][0][c] + red[1][0][c]) + (red[2][0][c] + red[3][0][c]));
        atomicAdd(dbeta + c, (red[0][1][c] + red[1][1][c]) + (red[2][1][c] + red[3][1][c]));
    }
}

#define MS_LN_DISPATCH(VPTVAR, CALL)                                                                            \
    if (VPTVAR <= 1) { CALL(1); } else if (VPTVAR <= 2) { CALL(2); } else if (VPTVAR <= 3) { CALL(3); }           \
    else if (VPTVAR <= 4) { CALL(4); } else if (VPTVAR <= 6) { CALL(6); } else if (VPTVAR <= 8) { CALL(8); }      \
    else if (VPTVAR <= 12) { CALL(12); } else if (VPTVAR <= 16) { CALL(16); } else if (VPTVAR <= 24) { CALL(24); } else { CALL(32); }

int ln_fwd_dispatch(const float *x, int64_t xps, const float *gamma, const float *beta, float eps, void *out,
                    int out_bf16, int64_t npix, int D, hipStream_t s) {
    if (!x || !gamma || !beta || !out) return MS_ERR_NULL;
    if (D <= 0 || D > 64 * kLnMaxVPT || npix < 0 || xps < D) return MS_ERR_SHAPE;
    if (npix == 0) return MS_OK;
    if (D % 4 == 0 && xps % 4 == 0 && ln_aligned(x, 16) && ln_aligned(gamma, 16) && ln_aligned(beta, 16) && ln_aligned(out, out_bf16 ? 8 : 16)) {
#define MS_S(L, V, P) do { const int64_t per = 4ll * (P) * (64 / (L));                                                           \
        const dim3 g((unsigned)((npix + per - 1) / per));                                                                       \
        if (out_bf16) hipLaunchKernelGGL((ln_fwd_sub_kernel<L, V, P, unsigned short>), g, dim3(256), 0, s, x, xps, gamma, beta, eps, (unsigned short *)out, D, npix, LnTaps{0, 0, 0}); \
        else hipLaunchKernelGGL((ln_fwd_sub_kernel<L, V, P, float>), g, dim3(256), 0, s, x, xps, gamma, beta, eps, (float *)out, D, npix, LnTaps{0, 0, 0}); } while (0)
        MS_LN_SUB_DISPATCH_WIDE(D, MS_S)
#undef MS_S
        return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
    }
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
    if (D % 4 == 0 && xps % 4 == 0 && ln_aligned(x, 16) && ln_aligned(gamma, 16) && ln_aligned(dx, 16) && ln_aligned(dout, dout_bf16 ? 8 : 16)) {
        // persistent blocks: enough to keep HBM busy, few enough that the closing burst of same-address dgamma / dbeta atomics
        // (~40 ns each per address) stays short -- swept 64..2048 at the four stage shapes (tools/bench_ln.py)
        const int64_t cap2 = npix >= 32768 ? 512 : 256;
#define MS_S(L, V, P) do { const int64_t per = 4ll * (P) * (64 / (L));                                                           \
        const int64_t nb = (npix + per - 1) / per;                                                                              \
        const dim3 g((unsigned)(nb < cap2 ? nb : cap2));                                                                        \
        if (dout_bf16) hipLaunchKernelGGL((ln_bwd_sub_kernel<L, V, P, unsigned short>), g, dim3(256), 0, s, x, xps, gamma, eps, (const unsigned short *)dout, dx, dgamma, dbeta, D, npix, LnTaps{0, 0, 0}); \
        else hipLaunchKernelGGL((ln_bwd_sub_kernel<L, V, P, float>), g, dim3(256), 0, s, x, xps, gamma, eps, (const float *)dout, dx, dgamma, dbeta, D, npix, LnTaps{0, 0, 0}); } while (0)
        MS_LN_SUB_DISPATCH_WIDE(D, MS_S)
#undef MS_S
        return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
    }
    const int vpt = (D + 63) / 64;
    const int pb = vpt <= 2 ? 4 : vpt <= 4 ? 2 : 1;                 // = MS_PB of the dispatched VPT bucket
    const int64_t tasks = (npix + pb - 1) / pb;                     // pixel groups
    const int64_t blocks = (tasks + 3) / 4;
    // measured optimum (tools/bench_ln.py): enough waves to keep HBM busy, few enough that the closing burst of
    // same-address dgamma/dbeta atomics stays short
    const int64_t cap = npix >= 131072 ? 1024 : npix >= 32768 ? 512 : 256;
    const dim3 grid((unsigned)(blocks < cap ? blocks : cap)), block(256);
    // PB pixels in flight per wave: 4 for narrow rows, fewer as the row itself supplies the parallelism
#define MS_PB(V) ((V) <= 2 ? 4 : (V) <= 4 ? 2 : 1)
    if (dout_bf16) {
#define MS_C(V) hipLaunchKernelGGL((ln_bwd_kernel<V, MS_PB(V), unsigned short>), grid, block, 0, s, x, xps, gamma, eps, (const unsigned short *)dout, dx, dgamma, dbeta, D, npix)
        MS_LN_DISPATCH(vpt, MS_C)
#undef MS_C
    } else {
#define MS_C(V) hipLaunchKernelGGL((ln_bwd_kernel<V, MS_PB(V), float>), grid, block, 0, s, x, xps, gamma, eps, (const float *)dout, dx, dgamma, dbeta, D, npix)
        MS_LN_DISPATCH(vpt, MS_C)
#undef MS_C
    }
#undef MS_PB
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

// LayerNorm over the 2 x 2 tap concatenation of a (batch, H, W, C) fp32 tensor (PatchMerging2D, MedMamba.py:196-205)
int ln_taps_fwd_dispatch(const float *x, const float *gamma, const float *beta, float eps, void *out, int out_bf16, int batch, int H, int W,
                         int C, hipStream_t s) {
    if (!x || !gamma || !beta || !out) return MS_ERR_NULL;
    const int D = 4 * C;
    if (batch < 0 || H <= 0 || W <= 0 || C <= 0 || H % 2 || W % 2 || C % 4 || D > 64 * kLnMaxVPT) return MS_ERR_SHAPE;
    if (!ln_aligned(x, 16) || !ln_aligned(gamma, 16) || !ln_aligned(beta, 16) || !ln_aligned(out, out_bf16 ? 8 : 16)) return MS_ERR_STRIDE;
    const int64_t npix = (int64_t)batch * (H / 2) * (W / 2);
    if (npix == 0) return MS_OK;
    const LnTaps tp{H / 2, W / 2, C};
#define MS_S(L, V, P) do { const int64_t per = 4ll * (P) * (64 / (L));                                                           \
        const dim3 g((unsigned)((npix + per - 1) / per));                                                                       \
        if (out_bf16) hipLaunchKernelGGL((ln_fwd_sub_kernel<L, V, P, unsigned short, true>), g, dim3(256), 0, s, x, (int64_t)D, gamma, beta, eps, (unsigned short *)out, D, npix, tp); \
        else hipLaunchKernelGGL((ln_fwd_sub_kernel<L, V, P, float, true>), g, dim3(256), 0, s, x, (int64_t)D, gamma, beta, eps, (float *)out, D, npix, tp); } while (0)
    MS_LN_SUB_DISPATCH_WIDE(D, MS_S)
#undef MS_S
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int ln_taps_bwd_dispatch(const float *x, const float *gamma, float eps, const void *dout, int dout_bf16, float *dx, float *dgamma,
                         float *dbeta, int batch, int H, int W, int C, hipStream_t s) {
    if (!x || !gamma || !dout || !dx || !dgamma || !dbeta) return MS_ERR_NULL;
    const int D = 4 * C;
    if (batch < 0 || H <= 0 || W <= 0 || C <= 0 || H % 2 || W % 2 || C % 4 || D > 64 * kLnMaxVPT) return MS_ERR_SHAPE;
    if (!ln_aligned(x, 16) || !ln_aligned(gamma, 16) || !ln_aligned(dx, 16) || !ln_aligned(dout, dout_bf16 ? 8 : 16)) return MS_ERR_STRIDE;
    const int64_t npix = (int64_t)batch * (H / 2) * (W / 2);
    if (npix == 0) return MS_OK;
    const LnTaps tp{H / 2, W / 2, C};
    const int64_t cap2 = npix >= 32768 ? 512 : 256;
#define MS_S(L, V, P) do { const int64_t per = 4ll * (P) * (64 / (L));                                                           \
        const int64_t nb = (npix + per - 1) / per;                                                                              \
        const dim3 g((unsigned)(nb < cap2 ? nb : cap2));                                                                        \
        if (dout_bf16) hipLaunchKernelGGL((ln_bwd_sub_kernel<L, V, P, unsigned short, true>), g, dim3(256), 0, s, x, (int64_t)D, gamma, eps, (const unsigned short *)dout, dx, dgamma, dbeta, D, npix, tp); \
        else hipLaunchKernelGGL((ln_bwd_sub_kernel<L, V, P, float, true>), g, dim3(256), 0, s, x, (int64_t)D, gamma, eps, (const float *)dout, dx, dgamma, dbeta, D, npix, tp); } while (0)
    MS_LN_SUB_DISPATCH_WIDE(D, MS_S)
#undef MS_S
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
