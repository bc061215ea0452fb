// Training-mode BatchNorm2d (+ optional fused ReLU) on channel-last activations for gfx950 -- the three BatchNorm2d
// (two of them followed by nn.ReLU) of SS_Conv_SSM's conv branch (MedMamba.py:517-527, applied at :533-535).
// torch runs each as 3 MIOpen kernels forward + 3 backward, a clamp and a threshold_backward for the ReLU and a tiny
// int64 add for num_batches_tracked: ~10 launches of a few microseconds of work each.  Here: 2 kernels forward
// (statistics; normalise + ReLU + running-statistics update) and 2 backward (dgamma/dbeta; dx), bf16 or fp32 I/O,
// fp32 statistics.  Semantics of torch.nn.functional.batch_norm(training=True): biased variance for normalisation,
// unbiased for running_var, running = (1-momentum)*running + momentum*batch.
// Layout: x is (npix, C) with unit channel stride (the memory of an NCHW tensor in channels_last format).
#include <hip/hip_runtime.h>
#include "medscan.h"
#include "ln_common.h"

namespace ms {

template <typename T> __device__ __forceinline__ float bn_ld(const T *p);
template <> __device__ __forceinline__ float bn_ld<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float bn_ld<unsigned short>(const unsigned short *p) { return __builtin_bit_cast(float, (unsigned)*p << 16); }
template <typename T> __device__ __forceinline__ void bn_st(T *p, float v);
template <> __device__ __forceinline__ void bn_st<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void bn_st<unsigned short>(unsigned short *p, float v) { *p = __builtin_bit_cast(unsigned short, (__bf16)v); }

constexpr int kBnThreads = 256;
constexpr int kBnMaxBlocks = 1024;

// Thread t of a block owns channel c = cb*CT + t % CT (CT = min(C, 256) rounded to the block) of the pixel rows
// t / CT + k * (256 / CT): consecutive threads read consecutive channels (coalesced); a block's threads that share a
// channel are combined in LDS, then one atomic per (block, channel, statistic).
struct BnGeom { int ct, rows_per_iter, ncb; };
static BnGeom bn_geom(int C) {
    int ct = 1;
    while (ct < C && ct < kBnThreads) ct *= 2;             // power of two >= min(C, 256)
    BnGeom g; g.ct = ct; g.rows_per_iter = kBnThreads / ct; g.ncb = (C + ct - 1) / ct;
    return g;
}

// two partial sums per thread -> per-channel totals of the block -> the block's row of the partials buffer
// part[blockIdx.x][2][C] (no atomics: a finalize kernel adds the rows; 1000+ same-address atomics per channel were
// the whole cost of the first version)
__device__ __forceinline__ void bn_block_combine(float a, float b, int ct, int c, bool cv, float *part, int C) {
    __shared__ float red[2][kBnThreads];
    const int t = threadIdx.x;
    red[0][t] = a; red[1][t] = b;
    __syncthreads();
    for (int s = kBnThreads / 2; s >= ct; s >>= 1) {       // fold the row slots onto the first `ct` threads
        if (t < s) { red[0][t] += red[0][t + s]; red[1][t] += red[1][t + s]; }
        __syncthreads();
    }
    if (t < ct && cv) {
        float *row = part + (int64_t)blockIdx.x * 2 * C;
        row[c] = red[0][t]; row[C + c] = red[1][t];
    }
}

// pixels block `bx` of `nblk` visits (rows bx*rpi + r0 + k*nblk*rpi < npix, r0 < rpi): every block gets `q` full rounds of
// rpi rows plus its share of the last, partial round
__device__ __forceinline__ float bn_block_count(int bx, int nblk, int rpi, int64_t npix) {
    const int64_t per_round = (int64_t)nblk * rpi;
    const int64_t q = npix / per_round, rem = npix - q * per_round;
    const int64_t extra = rem - (int64_t)bx * rpi;
    return (float)(q * rpi + (extra < 0 ? 0 : (extra > rpi ? rpi : extra)));
}

// pass 1 forward: per block and channel (mean_b, M2_b = sum (x - mean_b)^2) of the pixels the block visits, accumulated
// around a DATA-derived pivot (the channel's value at the block's first pixel): one-pass sums of x - pivot and
// (x - pivot)^2 lose no digits however far the channel's mean is from zero -- a pivot of running_mean (0 after a reset)
// cancelled catastrophically for |mean| >> std.  The finalize kernel merges the blocks with Chan's parallel formula.
template <typename T>
__global__ void __launch_bounds__(kBnThreads)
bn_stats_kernel(const T *__restrict__ x, int64_t xps, float *__restrict__ part, int64_t npix, int C, int ct, int rpi) {
    const int t = threadIdx.x, c = blockIdx.y * ct + t % ct, r0 = t / ct;
    const bool cv = c < C;
    const int64_t pfirst = (int64_t)blockIdx.x * rpi;
    const float pv = (cv && pfirst < npix) ? bn_ld(x + pfirst * xps + c) : 0.0f;
    float s1 = 0.0f, s2 = 0.0f;
    if (cv) {
        for (int64_t p = pfirst + r0; p < npix; p += (int64_t)gridDim.x * rpi) {
            const float v = bn_ld(x + p * xps + c) - pv;
            s1 += v; s2 = fmaf(v, v, s2);
        }
    }
    // block totals of (x - pv), (x - pv)^2 -> (mean_b, M2_b); the per-thread partials share the block's pivot, so they add
    __shared__ float red[2][kBnThreads];
    red[0][t] = s1; red[1][t] = s2;
    __syncthreads();
    for (int s = kBnThreads / 2; s >= ct; s >>= 1) {
        if (t < s) { red[0][t] += red[0][t + s]; red[1][t] += red[1][t + s]; }
        __syncthreads();
    }
    if (t < ct && cv) {
        const float nb = bn_block_count(blockIdx.x, gridDim.x, rpi, npix);
        const float m = nb > 0.0f ? red[0][t] / nb : 0.0f;
        float *row = part + (int64_t)blockIdx.x * 2 * C;
        row[c] = pv + m;                                               // mean_b
        row[C + c] = fmaxf(red[1][t] - red[0][t] * m, 0.0f);           // M2_b = S2 - S1^2 / n_b
    }
}

// Finalize blocks: 1024 threads = 16 channels x 64 row-slots; slot k adds partial rows k, k+64, ... (<= 16 loads in a
// row per thread for 1024 partial rows -- a 4-slot version spent 59 us here, latency-bound), then an LDS tree over slots.
constexpr int kFinCh = 16, kFinSlots = 64;
// a wave holds 4 row-slots x 16 channels (lane = slot * 16 + channel): the slots of a wave are summed in registers
// (v_permlane16_swap / v_permlane32_swap), the 16 waves through LDS -- two barriers instead of the ten of a 1024-thread LDS tree
// (these kernels do microseconds of work 60 times per step: their time IS the barriers)
__device__ __forceinline__ void bn_fin_combine(float &a, float &b, float (*red)[16][kFinCh]) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    a = across_groups<16>(a, lane); b = across_groups<16>(b, lane);
    if (lane < kFinCh) { red[0][wv][lane] = a; red[1][wv][lane] = b; }
    __syncthreads();
    float sa = 0.0f, sb = 0.0f;
#pragma unroll
    for (int w = 0; w < 16; ++w) { sa += red[0][w][t % kFinCh]; sb += red[1][w][t % kFinCh]; }
    a = sa; b = sb;
}
__device__ __forceinline__ void bn_sum_partials(const float *part, int nblk, int C, int c, bool cv, float &a, float &b) {
    __shared__ float red[2][16][kFinCh];
    const int slot = threadIdx.x / kFinCh;
    float sa = 0.0f, sb = 0.0f;
    if (cv) for (int k = slot; k < nblk; k += kFinSlots) { sa += part[(int64_t)k * 2 * C + c]; sb += part[(int64_t)k * 2 * C + C + c]; }
    bn_fin_combine(sa, sb, red);
    a = sa; b = sb;
}

// finalize forward: Chan's merge of the per-block (n_b, mean_b, M2_b): mean = sum n_b mean_b / N, M2 = sum M2_b + n_b (mean_b - mean)^2;
// batch mean / rstd (saved for apply and backward) and the running-statistics update
__global__ void __launch_bounds__(kFinCh * kFinSlots)
bn_finalize_fwd_kernel(const float *__restrict__ part, int nblk, int rpi, const float *__restrict__ shift,
                       float *__restrict__ running_mean, float *__restrict__ running_var, long long *__restrict__ nbt, float momentum, float eps,
                       float *__restrict__ save_mean, float *__restrict__ save_rstd, int64_t npix, int C) {
    __shared__ float red[2][16][kFinCh];
    const int t = threadIdx.x, slot = t / kFinCh;
    const int c = blockIdx.x * kFinCh + t % kFinCh;
    const bool cv = c < C;
    const int64_t per_round = (int64_t)nblk * rpi;
    const int64_t q = npix / per_round, rem = npix - q * per_round;
    auto count = [&](int k) { const int64_t e = rem - (int64_t)k * rpi; return (float)(q * rpi + (e < 0 ? 0 : (e > rpi ? rpi : e))); };
    // one pass around the pivot p = block 0's mean (block means differ from each other by ~std / sqrt(n_b): no cancellation):
    //   s0 = sum n_b (mean_b - p),  s1 = sum M2_b + n_b (mean_b - p)^2   ->   mean = p + s0 / N,  M2 = s1 - s0^2 / N
    const float pvt = cv ? part[c] : 0.0f;
    float sa = 0.0f, sb = 0.0f;
    if (cv) for (int k = slot; k < nblk; k += kFinSlots) {
        const float nb = count(k), dm = part[(int64_t)k * 2 * C + c] - pvt;
        sa = fmaf(nb, dm, sa);
        sb += fmaf(nb * dm, dm, part[(int64_t)k * 2 * C + C + c]);
    }
    bn_fin_combine(sa, sb, red);
    const float s0 = sa, mean = pvt + s0 / (float)npix;
    if (cv && threadIdx.x < kFinCh) {
        const float var = fmaxf((sb - s0 * s0 / (float)npix) / (float)npix, 0.0f);       // biased variance
        const float sh = shift ? shift[c] : 0.0f;                      // the layer's logical input is x + shift
        save_mean[c] = mean; save_rstd[c] = rsqrtf(var + eps);
        const float unb = npix > 1 ? var * ((float)npix / (float)(npix - 1)) : var;
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (mean + sh);
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unb;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
}

// finalize backward: dgamma, dbeta totals
__global__ void __launch_bounds__(kFinCh * kFinSlots)
bn_finalize_bwd_kernel(const float *__restrict__ part, int nblk, float *__restrict__ dgamma, float *__restrict__ dbeta, int C) {
    const int c = blockIdx.x * kFinCh + threadIdx.x % kFinCh;
    const bool cv = c < C;
    float sg, sb;
    bn_sum_partials(part, nblk, C, c, cv, sg, sb);
    if (cv && threadIdx.x < kFinCh) { dgamma[c] = sg; dbeta[c] = sb; }
}

// pass 2 forward: y = [relu]((x - mean) * rstd * gamma + beta)
template <typename T, typename TO>
__global__ void __launch_bounds__(kBnThreads)
bn_apply_kernel(const T *__restrict__ x, int64_t xps, const float *__restrict__ gamma, const float *__restrict__ beta,
                const float *__restrict__ save_mean, const float *__restrict__ save_rstd, int relu, TO *__restrict__ y,
                int64_t npix, int C, int ct, int rpi) {
    const int t = threadIdx.x, c = blockIdx.y * ct + t % ct, r0 = t / ct;
    if (c >= C) return;
    const float g = gamma[c] * save_rstd[c], b = beta[c] - save_mean[c] * g;
    for (int64_t p = (int64_t)blockIdx.x * rpi + r0; p < npix; p += (int64_t)gridDim.x * rpi) {
        float v = fmaf(bn_ld(x + p * xps + c), g, b);
        if (relu) v = fmaxf(v, 0.0f);
        bn_st(y + p * C + c, v);
    }
}

// pass 1 backward: dbeta = sum dy', dgamma = sum dy' * xhat, dy' = dy * [y > 0]
template <typename T, typename TG>
__global__ void __launch_bounds__(kBnThreads)
bn_bwd_reduce_kernel(const T *__restrict__ x, int64_t xps, const TG *__restrict__ dy, const float *__restrict__ gamma,
                     const float *__restrict__ beta, const float *__restrict__ save_mean,
                     const float *__restrict__ save_rstd, int relu, float *__restrict__ part, int64_t npix, int C, int ct,
                     int rpi) {
    const int t = threadIdx.x, c = blockIdx.y * ct + t % ct, r0 = t / ct;
    const bool cv = c < C;
    float sg = 0.0f, sb = 0.0f;
    if (cv) {
        const float mean = save_mean[c], rstd = save_rstd[c], g = gamma[c], b = beta[c];
        for (int64_t p = (int64_t)blockIdx.x * rpi + r0; p < npix; p += (int64_t)gridDim.x * rpi) {
            const float xh = (bn_ld(x + p * xps + c) - mean) * rstd;
            float d = bn_ld(dy + p * C + c);
            if (relu && fmaf(xh, g, b) <= 0.0f) d = 0.0f;
            sg = fmaf(d, xh, sg); sb += d;
        }
    }
    bn_block_combine(sg, sb, ct, c, cv, part, C);
}

// pass 2 backward: dx = gamma * rstd * (dy' - dbeta/n - xhat * dgamma/n)
template <typename T, typename TG, typename TD>
__global__ void __launch_bounds__(kBnThreads)
bn_bwd_apply_kernel(const T *__restrict__ x, int64_t xps, const TG *__restrict__ dy, const float *__restrict__ gamma,
                    const float *__restrict__ beta, const float *__restrict__ save_mean,
                    const float *__restrict__ save_rstd, int relu, const float *__restrict__ dgamma,
                    const float *__restrict__ dbeta, TD *__restrict__ dx, int64_t npix, int C, int ct, int rpi) {
    const int t = threadIdx.x, c = blockIdx.y * ct + t % ct, r0 = t / ct;
    if (c >= C) return;
    const float inv_n = 1.0f / (float)npix;
    const float mean = save_mean[c], rstd = save_rstd[c], g = gamma[c], b = beta[c];
    const float k1 = dbeta[c] * inv_n, k2 = dgamma[c] * inv_n, gs = g * rstd;
    for (int64_t p = (int64_t)blockIdx.x * rpi + r0; p < npix; p += (int64_t)gridDim.x * rpi) {
        const float xh = (bn_ld(x + p * xps + c) - mean) * rstd;
        float d = bn_ld(dy + p * C + c);
        if (relu && fmaf(xh, g, b) <= 0.0f) d = 0.0f;
        bn_st(dx + p * C + c, gs * (d - k1 - xh * k2));
    }
}

// apply for a BatchNorm whose statistics arrive as the replica rows of pivoted sums a convolution's epilogue accumulated (MsBnFold,
// conv3x3.hip): every thread derives its channel's scale / shift from the rows (2 x MS_BN_REPLICAS loads, L2-resident), then the
// same pass as bn_apply_kernel; the first row of workgroups also writes mean / rstd for the backward and updates the running statistics
__global__ void __launch_bounds__(kBnThreads)
bn_apply_sums_kernel(const unsigned short *__restrict__ x, const float *__restrict__ sums, const float *__restrict__ gamma,
                     const float *__restrict__ beta, const float *__restrict__ shift, float *__restrict__ running_mean,
                     float *__restrict__ running_var, long long *__restrict__ nbt, float *__restrict__ save_mean, float *__restrict__ save_rstd,
                     float momentum, float eps, int relu, unsigned short *__restrict__ y, int64_t npix, int C, int ct, int rpi) {
    const int t = threadIdx.x, c = blockIdx.y * ct + t % ct, r0 = t / ct;
    if (blockIdx.x == 0 && blockIdx.y == 0 && t == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    float a = 0.0f, b2 = 0.0f;
#pragma unroll 4
    for (int r = 0; r < MS_BN_REPLICAS; ++r) { a += sums[(r * 2) * C + c]; b2 += sums[(r * 2 + 1) * C + c]; }
    const float inv_n = 1.0f / (float)npix, m1 = a * inv_n, var = fmaxf(fmaf(-m1, m1, b2 * inv_n), 0.0f);
    const float mean = sums[MS_BN_REPLICAS * 2 * C + c] + m1, rstd = rsqrtf(var + eps);
    const float g = gamma[c] * rstd, b = beta[c] - mean * g;
    if (blockIdx.x == 0 && r0 == 0) {
        save_mean[c] = mean; save_rstd[c] = rstd;
        const float sh = shift ? shift[c] : 0.0f, unb = npix > 1 ? var * ((float)npix / (float)(npix - 1)) : var;
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (mean + sh);
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unb;
    }
    for (int64_t p = (int64_t)blockIdx.x * rpi + r0; p < npix; p += (int64_t)gridDim.x * rpi) {
        float v = fmaf(bn_ld(x + p * C + c), g, b);
        if (relu) v = fmaxf(v, 0.0f);
        bn_st(y + p * C + c, v);
    }
}

// backward apply for a BatchNorm whose dgamma / dbeta arrive as replica rows a convolution's input-gradient epilogue accumulated
// (MsBnBwd, conv3x3.hip BRED): totals from the rows in the prologue, then the same pass as bn_bwd_apply_kernel; dy is bf16
template <typename T, typename TD>
__global__ void __launch_bounds__(kBnThreads)
bn_bwd_apply_sums_kernel(const T *__restrict__ x, int64_t xps, const unsigned short *__restrict__ dy, const float *__restrict__ gamma,
                         const float *__restrict__ beta, const float *__restrict__ save_mean, const float *__restrict__ save_rstd, int relu,
                         const float *__restrict__ sums, float *__restrict__ dgamma, float *__restrict__ dbeta, TD *__restrict__ dx,
                         int64_t npix, int C, int ct, int rpi) {
    const int t = threadIdx.x, c = blockIdx.y * ct + t % ct, r0 = t / ct;
    if (c >= C) return;
    float sb = 0.0f, sg = 0.0f;
#pragma unroll 4
    for (int r = 0; r < MS_BN_REPLICAS; ++r) { sb += sums[(r * 2) * C + c]; sg += sums[(r * 2 + 1) * C + c]; }
    if (blockIdx.x == 0 && r0 == 0) { dgamma[c] = sg; dbeta[c] = sb; }
    const float inv_n = 1.0f / (float)npix;
    const float mean = save_mean[c], rstd = save_rstd[c], g = gamma[c], b = beta[c];
    const float k1 = sb * inv_n, k2 = sg * inv_n, gs = g * rstd;
    for (int64_t p = (int64_t)blockIdx.x * rpi + r0; p < npix; p += (int64_t)gridDim.x * rpi) {
        const float xh = (bn_ld(x + p * xps + c) - mean) * rstd;
        float d = bn_ld(dy + p * C + c);
        if (relu && fmaf(xh, g, b) <= 0.0f) d = 0.0f;
        bn_st(dx + p * C + c, gs * (d - k1 - xh * k2));
    }
}

static unsigned bn_blocks(int64_t npix, int rpi) {
    const int64_t need = (npix + (int64_t)rpi * 8 - 1) / ((int64_t)rpi * 8);         // >= 8 rows per thread
    return (unsigned)(need < 1 ? 1 : (need > kBnMaxBlocks ? kBnMaxBlocks : need));
}

int bn_scratch_floats(int C) { return 2 * (C > 0 ? C : 0) * kBnMaxBlocks; }

int bn_fwd_dispatch(const void *x, int x_bf16, int64_t xps, const float *shift, const float *gamma, const float *beta, float *running_mean,
                    float *running_var, long long *nbt, float momentum, float eps, int relu, void *y, int y_bf16,
                    float *save_mean, float *save_rstd, float *scratch, int64_t npix, int C, hipStream_t s) {
    if (!x || !gamma || !beta || !running_mean || !running_var || !y || !save_mean || !save_rstd || !scratch) return MS_ERR_NULL;
    if (npix <= 0 || C <= 0) return npix == 0 && C > 0 ? MS_OK : MS_ERR_SHAPE;
    if (xps < C) return MS_ERR_STRIDE;
    const BnGeom g = bn_geom(C);
    const unsigned nblk = bn_blocks(npix, g.rows_per_iter);
    const dim3 grid(nblk, (unsigned)g.ncb), block(kBnThreads);
    using bf = unsigned short;
    if (x_bf16) hipLaunchKernelGGL((bn_stats_kernel<bf>), grid, block, 0, s, (const bf *)x, xps, scratch, npix, C, g.ct, g.rows_per_iter);
    else        hipLaunchKernelGGL((bn_stats_kernel<float>), grid, block, 0, s, (const float *)x, xps, scratch, npix, C, g.ct, g.rows_per_iter);
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinCh * kFinSlots), 0, s, scratch, (int)nblk, g.rows_per_iter, shift, running_mean, running_var, nbt,
                       momentum, eps, save_mean, save_rstd, npix, C);
#define MS_BN_APPLY(TI, TO) hipLaunchKernelGGL((bn_apply_kernel<TI, TO>), grid, block, 0, s, (const TI *)x, xps, gamma, beta, save_mean, save_rstd, \
        relu, (TO *)y, npix, C, g.ct, g.rows_per_iter)
    if (x_bf16 && y_bf16) MS_BN_APPLY(bf, bf); else if (x_bf16) MS_BN_APPLY(bf, float);
    else if (y_bf16) MS_BN_APPLY(float, bf); else MS_BN_APPLY(float, float);
#undef MS_BN_APPLY
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int bn_bwd_apply_sums_dispatch(const MsBnBwd *bn, const void *dy, void *dx, int dx_bf16, float *dgamma, float *dbeta, int64_t npix, int C,
                               hipStream_t s) {
    if (!bn || !bn->x_pre || !bn->gamma || !bn->beta || !bn->save_mean || !bn->save_rstd || !bn->sums || !dy || !dx || !dgamma || !dbeta)
        return MS_ERR_NULL;
    if (npix <= 0 || C <= 0) return npix == 0 && C > 0 ? MS_OK : MS_ERR_SHAPE;
    if (bn->x_pre_pixel_stride < C) return MS_ERR_STRIDE;
    const BnGeom g = bn_geom(C);
    const unsigned nblk = bn_blocks(npix, g.rows_per_iter);
    const dim3 grid(nblk, (unsigned)g.ncb), block(kBnThreads);
    using bf = unsigned short;
#define MS_BN_BAS(TI, TD)                                                                                                              \
    hipLaunchKernelGGL((bn_bwd_apply_sums_kernel<TI, TD>), grid, block, 0, s, (const TI *)bn->x_pre, bn->x_pre_pixel_stride, (const bf *)dy, bn->gamma, \
                       bn->beta, bn->save_mean, bn->save_rstd, bn->relu, bn->sums, dgamma, dbeta, (TD *)dx, npix, C, g.ct, g.rows_per_iter)
    if (bn->x_pre_is_f32) { if (dx_bf16) { MS_BN_BAS(float, bf); } else { MS_BN_BAS(float, float); } }
    else { if (dx_bf16) { MS_BN_BAS(bf, bf); } else { MS_BN_BAS(bf, float); } }
#undef MS_BN_BAS
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int bn_apply_sums_dispatch(const void *x, const MsBnFold *bn, int relu, void *y, int64_t npix, int C, hipStream_t s) {
    if (!x || !y || !bn || !bn->sums || !bn->gamma || !bn->beta || !bn->running_mean || !bn->running_var || !bn->save_mean || !bn->save_rstd)
        return MS_ERR_NULL;
    if (npix <= 0 || C <= 0) return npix == 0 && C > 0 ? MS_OK : MS_ERR_SHAPE;
    const BnGeom g = bn_geom(C);
    const unsigned nblk = bn_blocks(npix, g.rows_per_iter);
    using bf = unsigned short;
    hipLaunchKernelGGL(bn_apply_sums_kernel, dim3(nblk, (unsigned)g.ncb), dim3(kBnThreads), 0, s, (const bf *)x, bn->sums, bn->gamma, bn->beta, bn->shift,
                       bn->running_mean, bn->running_var, (long long *)bn->num_batches_tracked, bn->save_mean, bn->save_rstd, bn->momentum, bn->eps,
                       relu, (bf *)y, npix, C, g.ct, g.rows_per_iter);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int bn_bwd_dispatch(const void *x, int x_bf16, int64_t xps, const void *dy, int dy_bf16, const float *gamma, const float *beta,
                    const float *save_mean, const float *save_rstd, int relu, void *dx, int dx_bf16, float *dgamma, float *dbeta,
                    float *scratch, int64_t npix, int C, hipStream_t s) {
    if (!x || !dy || !gamma || !beta || !save_mean || !save_rstd || !dx || !dgamma || !dbeta || !scratch) return MS_ERR_NULL;
    if (npix <= 0 || C <= 0) return npix == 0 && C > 0 ? MS_OK : MS_ERR_SHAPE;
    const BnGeom g = bn_geom(C);
    const unsigned nblk = bn_blocks(npix, g.rows_per_iter);
    const dim3 grid(nblk, (unsigned)g.ncb), block(kBnThreads);
    using bf = unsigned short;
#define MS_BN_BWD(TI, TG)                                                                                                          \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<TI, TG>), grid, block, 0, s, (const TI *)x, xps, (const TG *)dy, gamma, beta, save_mean,      \
                       save_rstd, relu, scratch, npix, C, g.ct, g.rows_per_iter);                                                   \
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinCh * kFinSlots), 0, s, scratch, (int)nblk, dgamma, dbeta, C);            \
    if (dx_bf16) hipLaunchKernelGGL((bn_bwd_apply_kernel<TI, TG, bf>), grid, block, 0, s, (const TI *)x, xps, (const TG *)dy, gamma, beta, save_mean,       \
                       save_rstd, relu, dgamma, dbeta, (bf *)dx, npix, C, g.ct, g.rows_per_iter);                                     \
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<TI, TG, float>), grid, block, 0, s, (const TI *)x, xps, (const TG *)dy, gamma, beta, save_mean,       \
                       save_rstd, relu, dgamma, dbeta, (float *)dx, npix, C, g.ct, g.rows_per_iter)
    if (x_bf16 && dy_bf16) { MS_BN_BWD(bf, bf); } else if (x_bf16) { MS_BN_BWD(bf, float); }
    else if (dy_bf16) { MS_BN_BWD(float, bf); } else { MS_BN_BWD(float, float); }
#undef MS_BN_BWD
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
