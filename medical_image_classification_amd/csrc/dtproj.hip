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
#include "scan_common.h"          // softplus_ref: the SAME function the scan kernels apply (bit-identical delta')

namespace ms {

constexpr int kDtMaxBlocksX = 256;          // persistent workgroups per (direction, channel block)

// grid: x = persistent pixel workers, y = 4 directions * ncb channel blocks; block = 4 waves.
template <int VPT, int RP>
__global__ void __launch_bounds__(256)
dtproj_fwd_kernel(const float *__restrict__ proj, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ delta,
                  int64_t npix, int D, int R, int C, int ncb) {
    // (wv is deliberately NOT made wave-uniform here: with scalar loads the three dts values of a pixel arrive as three s_load_dword
    // behind one lgkmcnt wait each -- 187 us against 104 with broadcast vector loads)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int k = blockIdx.y / ncb, cb = blockIdx.y % ncb;
    const int dbase = cb * 64 * VPT;
    float w[VPT][RP];
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int d = dbase + lane + 64 * j;
        if (R == RP && (reinterpret_cast<uintptr_t>(W) & 15) == 0) {
            // a channel's R weights are contiguous: RP / 4 16-byte loads instead of RP scalar ones (the prologue, not the pixels, was
            // this kernel's time on small maps: 48 scattered loads per lane for 16 pixels of work at stage 3)
            const float4 *wr = reinterpret_cast<const float4 *>(W + ((int64_t)k * D + min(d, D - 1)) * R);
#pragma unroll
            for (int r4 = 0; r4 < RP / 4; ++r4) {
                const float4 v = wr[r4];
                w[j][4 * r4] = d < D ? v.x : 0.0f; w[j][4 * r4 + 1] = d < D ? v.y : 0.0f;
                w[j][4 * r4 + 2] = d < D ? v.z : 0.0f; w[j][4 * r4 + 3] = d < D ? v.w : 0.0f;
            }
        } else {
#pragma unroll
            for (int r = 0; r < RP; ++r) w[j][r] = (d < D && r < R) ? W[((int64_t)k * D + d) * R + r] : 0.0f;
        }
    }
    float bv[VPT];                               // bias != NULL: delta' = softplus(delta + bias[k, d]) (MS_SCAN_DELTA_ACTIVATED)
#pragma unroll
    for (int j = 0; j < VPT; ++j) bv[j] = (bias && dbase + lane + 64 * j < D) ? bias[(int64_t)k * D + dbase + lane + 64 * j] : 0.0f;
    float *dk = delta + (int64_t)k * npix * D;
    // 8 pixels per trip; a wave makes several trips (grid sized for ~MS_DT_FWD_TRIPS of them) so that its prologue -- the weight and
    // bias loads above -- is paid once per ~64 pixels instead of once per 8, with the next trip's dts rows requested before this trip's
    // stores (one-trip waves: 110 us for 308 MB of stores at stage 0 of MedMamba-T)
    constexpr int PB = 8;
    const int64_t stride = (int64_t)gridDim.x * 4 * PB;
    int64_t p0 = ((int64_t)blockIdx.x * 4 + wv) * PB;
    if (p0 >= npix) return;
    float t[PB][RP], tn[PB][RP];
    auto load = [&](int64_t pb, float (&tt)[PB][RP]) {
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int64_t p = min(pb + q, npix - 1);
            const float *row = proj + (p * 4 + k) * C;            // wave-uniform address
#pragma unroll
            for (int r = 0; r < RP; ++r) tt[q][r] = r < R ? row[r] : 0.0f;
        }
    };
    load(p0, tn);
    for (; p0 < npix; p0 += stride) {
#pragma unroll
        for (int q = 0; q < PB; ++q)
#pragma unroll
            for (int r = 0; r < RP; ++r) t[q][r] = tn[q][r];
        if (p0 + stride < npix) load(p0 + stride, tn);
        float *o = dk + p0 * D + dbase + lane;
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            if (p0 + q < npix) {
#pragma unroll
                for (int j = 0; j < VPT; ++j) {
                    float a = 0.0f;
#pragma unroll
                    for (int r = 0; r < RP; ++r) a = fmaf(t[q][r], w[j][r], a);
                    if (bias) a = softplus_ref(a + bv[j]);
                    if (dbase + lane + 64 * j < D) o[q * D + 64 * j] = a;
                }
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
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: the dts rows come through scalar loads
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
#ifndef MS_DT_BWD_PB
#define MS_DT_BWD_PB 4
#endif
    constexpr int PB = MS_DT_BWD_PB;
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

// ---- ranks above 4: the same three products with the rank-R operand in SGPRs --------------------------------------------
// For R = 6..32 the kernels above lose to batched GEMMs (per-pixel LDS transposes, W re-read by every 8-pixel wave).  The small
// operand of each product is wave-uniform, so it is read with scalar loads and enters v_fma as an SGPR source -- no LDS at all:
//   forward   lane = channel (VPT slabs of 64): delta[p][d] = sum_r t[p][r] (SGPR) * w[d][r] (VGPR), 64 pixels per wave
//   dWdt      lane = channel: acc[d][r] += ddelta[p][d] * t[p][r] (SGPR), persistent waves, LDS combine + atomics at the end
//   ddts      lane = PIXEL: acc[r] += ddelta[p][d] * W[d][r] (SGPR) over the lane's own contiguous ddelta row (16-byte loads);
//             written in place into the dts columns of the projection-row gradient
// ddelta is read twice (once per backward product); at these ranks it fits the 256 MB MALL.
template <int VPT, int RP>
__global__ void __launch_bounds__(256)
dtproj_fwd_s_kernel(const float *__restrict__ proj, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ delta,
                    int64_t npix, int D, int R, int C, int ncb, int ppw) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = blockIdx.y / ncb, cb = blockIdx.y % ncb;
    const int dbase = cb * 64 * VPT;
    float w[VPT][RP];
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int d = dbase + lane + 64 * j;
        if (R == RP && (reinterpret_cast<uintptr_t>(W) & 15) == 0) {
            // a channel's R weights are contiguous: RP / 4 16-byte loads instead of RP scalar ones (the prologue, not the pixels, was
            // this kernel's time on small maps: 48 scattered loads per lane for 16 pixels of work at stage 3)
            const float4 *wr = reinterpret_cast<const float4 *>(W + ((int64_t)k * D + min(d, D - 1)) * R);
#pragma unroll
            for (int r4 = 0; r4 < RP / 4; ++r4) {
                const float4 v = wr[r4];
                w[j][4 * r4] = d < D ? v.x : 0.0f; w[j][4 * r4 + 1] = d < D ? v.y : 0.0f;
                w[j][4 * r4 + 2] = d < D ? v.z : 0.0f; w[j][4 * r4 + 3] = d < D ? v.w : 0.0f;
            }
        } else {
#pragma unroll
            for (int r = 0; r < RP; ++r) w[j][r] = (d < D && r < R) ? W[((int64_t)k * D + d) * R + r] : 0.0f;
        }
    }
    float bv[VPT];
#pragma unroll
    for (int j = 0; j < VPT; ++j) bv[j] = (bias && dbase + lane + 64 * j < D) ? bias[(int64_t)k * D + dbase + lane + 64 * j] : 0.0f;
    float *dk = delta + (int64_t)k * npix * D + dbase + lane;
    const int64_t p0 = ((int64_t)blockIdx.x * 4 + wv) * ppw;
    const int64_t p1 = p0 + ppw < npix ? p0 + ppw : npix;
#pragma unroll 2
    for (int64_t p = p0; p < p1; ++p) {
        const float *row = proj + (p * 4 + k) * C;                  // wave-uniform: scalar loads
        float t[RP];
#pragma unroll
        for (int r = 0; r < RP; ++r) t[r] = row[r];                 // r >= R: B columns of the same row (C >= RP, host-checked), times w = 0
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            float a = 0.0f;
#pragma unroll
            for (int r = 0; r < RP; ++r) a = fmaf(t[r], w[j][r], a);
            if (bias) a = softplus_ref(a + bv[j]);
            if (dbase + lane + 64 * j < D) dk[p * D + 64 * j] = a;
        }
    }
}

// dWdt: workgroup = 4 waves = 4 slabs of 64*VPT channels sharing tiles of 64 pixels; the tile's dts rows are staged in LDS once
// (broadcast ds_read_b128: RP/4 reads feed VPT*RP FMAs), ddelta rows are read coalesced 8 pixels ahead.  Persistent over tiles.
template <int VPT, int RP>
__global__ void __launch_bounds__(256)
dtproj_dw_s_kernel(const float *__restrict__ ddelta, const float *__restrict__ proj, float *__restrict__ dW, float *__restrict__ part,
                   int64_t npix, int D, int R, int C, int ncb) {
    __shared__ __attribute__((aligned(16))) float sT[64 * RP];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = blockIdx.y / ncb, cb = blockIdx.y % ncb;
    const int dbase = (cb * 4 + wv) * 64 * VPT;
    float acc[VPT][RP];
#pragma unroll
    for (int j = 0; j < VPT; ++j)
#pragma unroll
        for (int r = 0; r < RP; ++r) acc[j][r] = 0.0f;
    const float *dk = ddelta + (int64_t)k * npix * D;
    const int64_t ntiles = (npix + 63) / 64;
    constexpr int PF = 8;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t p0 = tile * 64;
        __syncthreads();                                            // the previous tile's rows have been read
        for (int idx = threadIdx.x; idx < 64 * RP; idx += 256) {
            const int px = idx / RP, r = idx - px * RP;
            const int64_t p = p0 + px;
            sT[idx] = p < npix ? proj[(p * 4 + k) * C + r] : 0.0f;   // r >= R: in-row values into accumulators never written out
        }
        __syncthreads();
        if (dbase < D) {
            // (requesting the next group's rows one group ahead through a second register set was tried: the compiler then hoists the
            // LDS reads of both groups -- 310 VGPRs, one wave per SIMD, 72 vs 66 us at stage 3)
#pragma unroll 1
            for (int q0 = 0; q0 < 64; q0 += PF) {
                float g[PF][VPT];
#pragma unroll
                for (int q = 0; q < PF; ++q) {
                    const int64_t p = p0 + q0 + q;
#pragma unroll
                    for (int j = 0; j < VPT; ++j) {
                        const int d = dbase + lane + 64 * j;
                        g[q][j] = (d < D && p < npix) ? dk[p * D + d] : 0.0f;
                    }
                }
#pragma unroll
                for (int q = 0; q < PF; ++q) {
#pragma unroll
                    for (int r4 = 0; r4 < RP / 4; ++r4) {
                        const float4 t = *reinterpret_cast<const float4 *>(sT + (q0 + q) * RP + 4 * r4);
#pragma unroll
                        for (int j = 0; j < VPT; ++j) {
                            acc[j][4 * r4 + 0] = fmaf(g[q][j], t.x, acc[j][4 * r4 + 0]);
                            acc[j][4 * r4 + 1] = fmaf(g[q][j], t.y, acc[j][4 * r4 + 1]);
                            acc[j][4 * r4 + 2] = fmaf(g[q][j], t.z, acc[j][4 * r4 + 2]);
                            acc[j][4 * r4 + 3] = fmaf(g[q][j], t.w, acc[j][4 * r4 + 3]);
                        }
                    }
                }
            }
        }
    }
    // partial sums of this workgroup: one row of `part` per blockIdx.x (summed by dtproj_dw_finalize_kernel -- same-address atomics
    // from hundreds of workgroups cost 130 us here), or atomics straight into dW when the caller gave no workspace
    float *dst = part ? part + (int64_t)blockIdx.x * 4 * D * R : dW;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int d = dbase + lane + 64 * j;
        if (d < D) {
#pragma unroll
            for (int r = 0; r < RP; ++r) {
                if (r < R) {
                    if (part) dst[((int64_t)k * D + d) * R + r] = acc[j][r];
                    else atomicAdd(dst + ((int64_t)k * D + d) * R + r, acc[j][r]);
                }
            }
        }
    }
}

// dW[i] += sum over the nrows partial rows (dW arrives zero-filled like every accumulated output of this library)
__global__ void __launch_bounds__(256)
dtproj_dw_finalize_kernel(const float *__restrict__ part, float *__restrict__ dW, int n, int nrows) {
    // 16 columns of 4 floats x 16 row slices per block, combined in LDS: a thread adds nrows / 16 rows instead of all of them
    // (one thread per element walked ~100 rows in a dependent chain: 14.5 us for 7 MB)
    __shared__ float4 red[16][16];
    const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = (blockIdx.x * 16 + col) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        if (i + 4 <= n && (n & 3) == 0) {
#pragma unroll 4
            for (int b = sl; b < nrows; b += 16) {
                const float4 v = *reinterpret_cast<const float4 *>(part + (int64_t)b * n + i);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
        } else {
            for (int b = sl; b < nrows; b += 16) {
                const float *r = part + (int64_t)b * n + i;
                a.x += r[0]; if (i + 1 < n) a.y += r[1]; if (i + 2 < n) a.z += r[2]; if (i + 3 < n) a.w += r[3];
            }
        }
    }
    red[sl][col] = a;
    __syncthreads();
#pragma unroll
    for (int s2 = 8; s2 >= 1; s2 >>= 1) {
        if (sl < s2) { float4 &m = red[sl][col]; const float4 o = red[sl + s2][col]; m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w; }
        __syncthreads();
    }
    if (sl == 0 && i < n) {
        const float4 t = red[0][col];
        dW[i] += t.x; if (i + 1 < n) dW[i + 1] += t.y; if (i + 2 < n) dW[i + 2] += t.z; if (i + 3 < n) dW[i + 3] += t.w;
    }
}

// ddts: workgroup = 4 waves = 4 slices of 64 channels over the SAME 64 pixels, lane = pixel: acc[r] += ddelta[p][d] * W[d][r] with
// the W rows as SGPR operands (scalar loads) over the lane's own contiguous piece of its ddelta row (sixteen 16-byte loads, all in
// flight before the first FMA); the four partial sums meet in LDS; channels beyond the workgroup's 256 add with atomics (the
// dts columns of the projection-row gradient arrive zero-filled).  grid (ceil(npix / 64), 4 directions * ceil(D / 256)); D % 4 == 0.
template <int RP>
__global__ void __launch_bounds__(256)
dtproj_dts_s_kernel(const float *__restrict__ ddelta, const float *__restrict__ W, float *__restrict__ dproj,
                    int64_t npix, int D, int R, int C, int ncg) {
    __shared__ float sP[4][64][RP + 1];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = blockIdx.y / ncg, cg = blockIdx.y % ncg;
    const int d_lo = (cg * 4 + wv) * 64;                             // this wave's channel slice [d_lo, d_hi)
    const int d_hi = d_lo + 64 < D ? d_lo + 64 : D;
    const int64_t p = (int64_t)blockIdx.x * 64 + lane;
    const bool live = p < npix;
    float acc[RP];
#pragma unroll
    for (int r = 0; r < RP; ++r) acc[r] = 0.0f;
    if (d_lo < D) {
        const float *grow = ddelta + ((int64_t)k * npix + (live ? p : npix - 1)) * D;
        const float *Wk = W + (int64_t)k * D * R;
        float4 g[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            g[i] = d_lo + 4 * i < d_hi ? *reinterpret_cast<const float4 *>(grow + d_lo + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
        // W rows are read RP wide (r >= R: the next row's leading weights, into accumulators that are never written out); the last
        // four rows of the whole tensor are read R wide
        const int d_safe = (k == 3 && d_hi == D) ? d_hi - 4 : d_hi;
#pragma unroll
        for (int i = 0; i < 16; ++i) {                              // fully unrolled: g[] stays in registers
            const int d0 = d_lo + 4 * i;
            if (d0 < d_safe) {                                      // wave-uniform
                const float gv[4] = {g[i].x, g[i].y, g[i].z, g[i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float *wrow = Wk + (int64_t)(d0 + e) * R; // wave-uniform: scalar loads
#pragma unroll
                    for (int r = 0; r < RP; ++r) acc[r] = fmaf(gv[e], wrow[r], acc[r]);
                }
            }
        }
        if (d_safe < d_hi) {
            const float4 gl = *reinterpret_cast<const float4 *>(grow + d_safe);
            const float gv[4] = {gl.x, gl.y, gl.z, gl.w};
            for (int e = 0; e < 4; ++e) {
                const float *wrow = Wk + (int64_t)(d_safe + e) * R;
#pragma unroll
                for (int r = 0; r < RP; ++r) acc[r] = fmaf(gv[e], (e < 3 || r < R) ? wrow[r] : 0.0f, acc[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RP; ++r) sP[wv][lane][r] = acc[r];
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * R; idx += 256) {
        const int px = idx / R, r = idx - px * R;
        const int64_t pp = (int64_t)blockIdx.x * 64 + px;
        if (pp < npix) {
            const float v = (sP[0][px][r] + sP[1][px][r]) + (sP[2][px][r] + sP[3][px][r]);
            float *o = dproj + (pp * 4 + k) * C + r;
            if (ncg == 1) *o = v; else atomicAdd(o, v);
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
    static const int fwd_trips = [] { const char *e = getenv("MEDSCAN_DT_FWD_TRIPS"); return e ? atoi(e) : 8; }();
    const int trips = npix >= 32768 ? (fwd_trips < 1 ? 1 : fwd_trips) : 1;      // small maps: one trip per wave (more, shorter waves)
    int64_t blocks = bwd ? (npix + 2 * 4 * 16 - 1) / (2 * 4 * 16) : (npix + 8 * 4 * trips - 1) / (8 * 4 * trips);
    if (bwd && blocks > kDtMaxBlocksX) blocks = kDtMaxBlocksX;
    const dim3 grid((unsigned)(blocks < 1 ? 1 : blocks), (unsigned)(4 * ncb)), block(256);
#define MS_DT(V)                                                                                                         \
    if (bwd) hipLaunchKernelGGL((dtproj_bwd_kernel<V, RP>), grid, block, 0, s, a, proj, W, o1, o2, npix, D, R, C, ncb);    \
    else hipLaunchKernelGGL((dtproj_fwd_kernel<V, RP>), grid, block, 0, s, proj, W, a, o1, npix, D, R, C, ncb)
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

// scalar-operand kernels (ranks above 4): RP = R rounded up to a multiple of 4 keeps the unrolled loops tight
static int64_t dw_workers(int64_t npix, int ncbw) {
    // dWdt: workgroups of 4 channel slabs, persistent over 64-pixel tiles: about 8 workgroups per CU in total, at least 2 tiles each
    const int64_t ntiles = (npix + 63) / 64;
    int64_t bx = (2048 + 4 * ncbw - 1) / (4 * ncbw);
    // at least 2 tiles per workgroup -- unless that leaves the chip under-filled (7 x 7 maps: 49 tiles): then one tile each
    const int64_t cap = ntiles * 4 * ncbw <= 1024 ? ntiles : (ntiles + 1) / 2;
    if (bx > cap) bx = cap;
    return bx < 1 ? 1 : bx;
}

template <int RP>
static int launch_dt_s(bool bwd, const float *a, const float *proj, const float *W, float *o1, float *o2, float *scratch,
                       int64_t scratch_floats, int64_t npix, int D, int R, int C, hipStream_t s) {
    const int vpt = (D % 128 == 0 && RP <= 32) ? 2 : 1;
    const int ncb = (D + 64 * vpt - 1) / (64 * vpt);
    if (!bwd) {
        const int ppw = npix >= 32768 ? 64 : 16;                 // few pixels: more, shorter waves
        const dim3 grid((unsigned)((npix + 4 * ppw - 1) / (4 * ppw)), (unsigned)(4 * ncb));
        if (vpt == 2) hipLaunchKernelGGL((dtproj_fwd_s_kernel<2, RP>), grid, dim3(256), 0, s, proj, W, a, o1, npix, D, R, C, ncb, ppw);
        else          hipLaunchKernelGGL((dtproj_fwd_s_kernel<1, RP>), grid, dim3(256), 0, s, proj, W, a, o1, npix, D, R, C, ncb, ppw);
        return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
    }
    // ddts: 64 pixels x 256 channels per workgroup
    const int ncg = (D + 255) / 256;
    hipLaunchKernelGGL((dtproj_dts_s_kernel<RP>), dim3((unsigned)((npix + 63) / 64), (unsigned)(4 * ncg)), dim3(256), 0, s, a, W, o1, npix,
                       D, R, C, ncg);
    const int ncbw = (D + 256 * vpt - 1) / (256 * vpt);
    const int64_t bx = dw_workers(npix, ncbw);
    const int n = 4 * D * R;
    float *part = (scratch && scratch_floats >= bx * n) ? scratch : nullptr;
    const dim3 grid((unsigned)bx, (unsigned)(4 * ncbw));
    if (vpt == 2) hipLaunchKernelGGL((dtproj_dw_s_kernel<2, RP>), grid, dim3(256), 0, s, a, proj, o2, part, npix, D, R, C, ncbw);
    else          hipLaunchKernelGGL((dtproj_dw_s_kernel<1, RP>), grid, dim3(256), 0, s, a, proj, o2, part, npix, D, R, C, ncbw);
    if (part) hipLaunchKernelGGL(dtproj_dw_finalize_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, part, o2, n, (int)bx);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

static int dt_dispatch(bool bwd, const float *a, const float *proj, const float *W, float *o1, float *o2, float *scratch,
                       int64_t scratch_floats, int64_t npix, int D, int R, int C, hipStream_t s) {
    if (npix < 0 || D <= 0 || R <= 0 || C < R) return MS_ERR_SHAPE;
    if (R > 32) return MS_ERR_SHAPE;                              // dt_rank > 32 (d_model > 512) is not tiled by this build
    if (npix == 0) return MS_OK;
    if (R > 4 && D % 4 == 0 && C >= (R + 3) / 4 * 4) {
        switch ((R + 3) / 4) {
            case 2: return launch_dt_s<8>(bwd, a, proj, W, o1, o2, scratch, scratch_floats, npix, D, R, C, s);
            case 3: return launch_dt_s<12>(bwd, a, proj, W, o1, o2, scratch, scratch_floats, npix, D, R, C, s);
            case 4: return launch_dt_s<16>(bwd, a, proj, W, o1, o2, scratch, scratch_floats, npix, D, R, C, s);
            case 5: return launch_dt_s<20>(bwd, a, proj, W, o1, o2, scratch, scratch_floats, npix, D, R, C, s);
            case 6: return launch_dt_s<24>(bwd, a, proj, W, o1, o2, scratch, scratch_floats, npix, D, R, C, s);
            case 7: return launch_dt_s<28>(bwd, a, proj, W, o1, o2, scratch, scratch_floats, npix, D, R, C, s);
            default: return launch_dt_s<32>(bwd, a, proj, W, o1, o2, scratch, scratch_floats, npix, D, R, C, s);
        }
    }
    switch (dt_rp(R)) {
        case 4: return launch_dt<4>(bwd, a, proj, W, o1, o2, npix, D, R, C, s);
        case 8: return launch_dt<8>(bwd, a, proj, W, o1, o2, npix, D, R, C, s);
        case 16: return launch_dt<16>(bwd, a, proj, W, o1, o2, npix, D, R, C, s);
        default: return launch_dt<32>(bwd, a, proj, W, o1, o2, npix, D, R, C, s);
    }
}

// bias (4, D) or NULL: forward `a` operand of dt_dispatch = the activation's bias (delta' = softplus(delta + bias))
int dtproj_fwd_dispatch(const float *proj, const float *W, const float *bias, float *delta, int64_t npix, int D, int R, int C,
                        hipStream_t s) {
    if (!proj || !W || !delta) return MS_ERR_NULL;
    return dt_dispatch(false, bias, proj, W, delta, nullptr, nullptr, 0, npix, D, R, C, s);
}

int dtproj_bwd_dispatch(const float *ddelta, const float *proj, const float *W, float *dproj, float *dW, float *scratch,
                        int64_t scratch_floats, int64_t npix, int D, int R, int C, hipStream_t s) {
    if (!ddelta || !proj || !W || !dproj || !dW) return MS_ERR_NULL;
    return dt_dispatch(true, ddelta, proj, W, dproj, dW, scratch, scratch_floats, npix, D, R, C, s);
}

// floats of workspace with which ms_dtproj_bwd sums its per-workgroup dWdt partials without atomics (0: none needed)
int64_t dtproj_bwd_scratch_floats(int64_t npix, int D, int R) {
    if (npix <= 0 || D <= 0 || R <= 4 || R > 32 || D % 4 != 0) return 0;
    const int vpt = (D % 128 == 0) ? 2 : 1;
    return dw_workers(npix, (D + 256 * vpt - 1) / (256 * vpt)) * 4 * D * R;
}

}  // namespace ms
