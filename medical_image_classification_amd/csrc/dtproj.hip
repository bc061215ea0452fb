// Delta projection of SS2D for gfx950 (MedMamba.py:400,403-405):
//   delta[k, m, d] = sum_r dts[m, k, r] * Wdt[k, d, r]        k = direction, m = pixel, r < R = dt_rank (3..32), d < D
// and its backward
//   ddts[m, k, r] = sum_d ddelta[k, m, d] * Wdt[k, d, r]      dWdt[k, d, r] = sum_m ddelta[k, m, d] * dts[m, k, r]
// The reference runs these as einsum -> batched GEMMs with K = R; on this stack the fp32 strided-batched GEMMs with such
// shapes cost 0.2-0.5 ms of HOST time per call (11 ms of a 31 ms step, torch profiler) and a copy of dts out of the
// projection rows.  They are HBM-bound outer products, so here they are plain HIP: dts is read in place from the
// x_proj output rows [dts(R) | B(N) | C(N)] per (pixel, direction), ddts is written in place into the same columns of the
// projection gradient (whose B|C columns the scan backward fills), W lives in registers.
// One wave = up to 64*VPT channels of one direction, persistent over pixels; lane owns channels lane, lane+64, ...
#include <hip/hip_runtime.h>
#include "medscan.h"

namespace ms {

constexpr int kDtMaxBlocksX = 256;          // persistent workgroups per (direction, channel block)

// grid: x = persistent pixel workers, y = 4 directions * ncb channel blocks; block = 4 waves.
template <int VPT, int RP>
__global__ void __launch_bounds__(256)
dtproj_fwd_kernel(const float *__restrict__ proj, const float *__restrict__ W, float *__restrict__ delta,
                  int64_t npix, int D, int R, int C, int ncb) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int k = blockIdx.y / ncb, cb = blockIdx.y % ncb;
    const int dbase = cb * 64 * VPT;
    float w[VPT][RP];
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int d = dbase + lane + 64 * j;
#pragma unroll
        for (int r = 0; r < RP; ++r) w[j][r] = (d < D && r < R) ? W[((int64_t)k * D + d) * R + r] : 0.0f;
    }
    float *dk = delta + (int64_t)k * npix * D;
    constexpr int PB = 8;                       // pixels per wave (one trip: nothing to accumulate, the hardware overlaps waves)
    const int64_t p0 = ((int64_t)blockIdx.x * 4 + wv) * PB;
    if (p0 >= npix) return;
    float t[PB][RP];
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int64_t p = min(p0 + q, npix - 1);
        const float *row = proj + (p * 4 + k) * C;                // wave-uniform address
#pragma unroll
        for (int r = 0; r < RP; ++r) t[q][r] = r < R ? row[r] : 0.0f;
    }
    float *o = dk + p0 * D + dbase + lane;
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        if (p0 + q < npix) {
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                float a = 0.0f;
#pragma unroll
                for (int r = 0; r < RP; ++r) a = fmaf(t[q][r], w[j][r], a);
                if (dbase + lane + 64 * j < D) o[q * D + 64 * j] = a;
            }
        }
    }
}

template <int VPT, int RP>
__global__ void __launch_bounds__(256)
dtproj_bwd_kernel(const float *__restrict__ ddelta, const float *__restrict__ proj, const float *__restrict__ W,
                  float *__restrict__ dproj, float *__restrict__ dW, int64_t npix, int D, int R, int C, int ncb) {
    __shared__ __attribute__((aligned(16))) float sT[4][RP * 64];        // per wave: partial ddts [r][lane]
    __shared__ float sW[3][VPT * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int k = blockIdx.y / ncb, cb = blockIdx.y % ncb;
    const int dbase = cb * 64 * VPT;
    float w[VPT][RP], acc[VPT][RP];
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int d = dbase + lane + 64 * j;
#pragma unroll
        for (int r = 0; r < RP; ++r) { w[j][r] = (d < D && r < R) ? W[((int64_t)k * D + d) * R + r] : 0.0f; acc[j][r] = 0.0f; }
    }
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const float *dk = ddelta + (int64_t)k * npix * D;
    float *st = sT[wv];
    // ownership for the cross-lane sum: lanes [rr*LPR, (rr+1)*LPR) sum row rr of the tile, RP values each
    constexpr int LPR = 64 / RP;                                  // lanes per r (RP = 4, 8, 16, 32 -> 16, 8, 4, 2)
    const int rr = lane / LPR, seg = lane % LPR;
    constexpr int PB = 2;
    float g[PB][VPT], t[PB][RP], gn[PB][VPT], tn[PB][RP];
    auto load = [&](int64_t p0, float (&gg)[PB][VPT], float (&tt)[PB][RP]) {
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int64_t p = min(p0 + q, npix - 1);
            const bool live = p0 + q < npix;
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                const int d = dbase + lane + 64 * j;
                const float v = dk[p * D + min(d, D - 1)];
                gg[q][j] = (d < D && live) ? v : 0.0f;
            }
            const float *row = proj + (p * 4 + k) * C;
#pragma unroll
            for (int r = 0; r < RP; ++r) tt[q][r] = r < R ? row[r] : 0.0f;
        }
    };
    int64_t p0 = ((int64_t)blockIdx.x * 4 + wv) * PB;
    if (p0 < npix) load(p0, gn, tn);
    for (; p0 < npix; p0 += nwaves * PB) {
#pragma unroll
        for (int q = 0; q < PB; ++q) {
#pragma unroll
            for (int j = 0; j < VPT; ++j) g[q][j] = gn[q][j];
#pragma unroll
            for (int r = 0; r < RP; ++r) t[q][r] = tn[q][r];
        }
        if (p0 + nwaves * PB < npix) load(p0 + nwaves * PB, gn, tn);      // next group: in flight during this one
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            float part[RP];
#pragma unroll
            for (int r = 0; r < RP; ++r) part[r] = 0.0f;
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
#pragma unroll
                for (int r = 0; r < RP; ++r) {
                    acc[j][r] = fmaf(g[q][j], t[q][r], acc[j][r]);
                    part[r] = fmaf(g[q][j], w[j][r], part[r]);
                }
            }
            // sum part[r] over the 64 lanes: transpose through the wave's LDS tile, then a short DPP row reduction
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < RP; ++r) st[r * 64 + lane] = part[r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < RP / 4; ++i) {
                const float4 v4 = *reinterpret_cast<const float4 *>(st + rr * 64 + seg * RP + 4 * i);
                s += (v4.x + v4.y) + (v4.z + v4.w);
            }
            // lanes of one r are adjacent and LPR <= 16: finish inside the DPP row (quad_perm xor 1, xor 2, then row_ror 4 / 8:
            // after row_ror:4 only the upper quad of an 8-lane group holds the group total -> the LAST lane of the group owns it)
            if (LPR >= 2) s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(unsigned, s), 0xB1, 0xF, 0xF, true));
            if (LPR >= 4) s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(unsigned, s), 0x4E, 0xF, 0xF, true));
            if (LPR >= 8) s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(unsigned, s), 0x124, 0xF, 0xF, true));
            if (LPR >= 16) s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(unsigned, s), 0x128, 0xF, 0xF, true));
            if (seg == LPR - 1 && rr < R && p0 + q < npix) {
                float *o = dproj + ((p0 + q) * 4 + k) * C + rr;
                if (ncb == 1) *o = s; else atomicAdd(o, s);       // channel blocks of one direction add up
            }
        }
    }
    // dW: combine the 4 waves of the block per r in LDS, then one atomic per (block, channel, r)
#pragma unroll
    for (int r = 0; r < RP; ++r) {
        if (r < R) {
            if (wv > 0) {
#pragma unroll
                for (int j = 0; j < VPT; ++j) sW[wv - 1][j * 64 + lane] = acc[j][r];
            }
            __syncthreads();
            if (wv == 0) {
#pragma unroll
                for (int j = 0; j < VPT; ++j) {
                    const int d = dbase + lane + 64 * j;
                    if (d < D) atomicAdd(dW + ((int64_t)k * D + d) * R + r,
                                         acc[j][r] + sW[0][j * 64 + lane] + sW[1][j * 64 + lane] + sW[2][j * 64 + lane]);
                }
            }
            __syncthreads();
        }
    }
}

// channels per wave: as many 64-channel slabs as keep VPT * RP accumulators <= 128 registers
static int dt_vpt(int D, int rp) {
    const int need = (D + 63) / 64, cap = 128 / rp;             // rp 4 -> 32, 8 -> 16, 16 -> 8, 32 -> 4
    const int cands[7] = {1, 2, 3, 4, 6, 8, 12};
    int fit = 1;                                                 // largest candidate within the register budget
    for (int i = 0; i < 7; ++i) if (cands[i] <= cap) fit = cands[i];
    for (int i = 0; i < 7; ++i) if (cands[i] >= need && cands[i] <= fit) return cands[i];    // one channel block
    return fit;                                                  // several channel blocks of 64 * fit channels
}
static int dt_rp(int R) { return R <= 4 ? 4 : R <= 8 ? 8 : R <= 16 ? 16 : 32; }

template <int RP>
static int launch_dt(bool bwd, const float *a, const float *proj, const float *W, float *o1, float *o2, int64_t npix, int D,
                     int R, int C, hipStream_t s) {
    const int vpt = dt_vpt(D, RP);
    const int ncb = (D + 64 * vpt - 1) / (64 * vpt);
    // forward: one trip per wave (8 pixels).  backward: persistent waves (dWdt accumulators), at least 16 trips each so the
    // closing LDS-combine + atomics round is amortised, at most kDtMaxBlocksX workgroups per (direction, channel block)
    int64_t blocks = bwd ? (npix + 2 * 4 * 16 - 1) / (2 * 4 * 16) : (npix + 8 * 4 - 1) / (8 * 4);
    if (bwd && blocks > kDtMaxBlocksX) blocks = kDtMaxBlocksX;
    const dim3 grid((unsigned)(blocks < 1 ? 1 : blocks), (unsigned)(4 * ncb)), block(256);
#define MS_DT(V)                                                                                                         \
    if (bwd) hipLaunchKernelGGL((dtproj_bwd_kernel<V, RP>), grid, block, 0, s, a, proj, W, o1, o2, npix, D, R, C, ncb);    \
    else hipLaunchKernelGGL((dtproj_fwd_kernel<V, RP>), grid, block, 0, s, proj, W, o1, npix, D, R, C, ncb)
    switch (vpt) {
        case 1: MS_DT(1); break;
        case 2: MS_DT(2); break;
        case 3: MS_DT(3); break;
        case 4: MS_DT(4); break;
        case 6: if constexpr (RP <= 16) { MS_DT(6); } break;
        case 8: if constexpr (RP <= 16) { MS_DT(8); } break;
        case 12: if constexpr (RP <= 8) { MS_DT(12); } break;
    }
#undef MS_DT
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

static int dt_dispatch(bool bwd, const float *a, const float *proj, const float *W, float *o1, float *o2, int64_t npix, int D,
                       int R, int C, hipStream_t s) {
    if (npix < 0 || D <= 0 || R <= 0 || C < R) return MS_ERR_SHAPE;
    if (R > 32) return MS_ERR_SHAPE;                              // dt_rank > 32 (d_model > 512) is not tiled by this build
    if (npix == 0) return MS_OK;
    switch (dt_rp(R)) {
        case 4: return launch_dt<4>(bwd, a, proj, W, o1, o2, npix, D, R, C, s);
        case 8: return launch_dt<8>(bwd, a, proj, W, o1, o2, npix, D, R, C, s);
        case 16: return launch_dt<16>(bwd, a, proj, W, o1, o2, npix, D, R, C, s);
        default: return launch_dt<32>(bwd, a, proj, W, o1, o2, npix, D, R, C, s);
    }
}

int dtproj_fwd_dispatch(const float *proj, const float *W, float *delta, int64_t npix, int D, int R, int C, hipStream_t s) {
    if (!proj || !W || !delta) return MS_ERR_NULL;
    return dt_dispatch(false, nullptr, proj, W, delta, nullptr, npix, D, R, C, s);
}

int dtproj_bwd_dispatch(const float *ddelta, const float *proj, const float *W, float *dproj, float *dW, int64_t npix, int D,
                        int R, int C, hipStream_t s) {
    if (!ddelta || !proj || !W || !dproj || !dW) return MS_ERR_NULL;
    return dt_dispatch(true, ddelta, proj, W, dproj, dW, npix, D, R, C, s);
}

}  // namespace ms
