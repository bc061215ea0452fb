// SS2D fast path of the selective scan for gfx950 (MI355X): the kernels MedMamba's SS2D block actually runs
// (/root/reference/MedMamba.py:386-424: cross-scan, x_proj / dt_proj, selective_scan_fn, cross-merge), specialised for
//   * SS2D addressing (channel-last activations indexed by pixel through the direction's position table, projection rows
//     [dts | B | C] contiguous along the state axis), d_state == 16, dense real A;
//   * 16-byte vector loads / stores of the activations (one lane = 4 consecutive channels of one position: a wave's
//     [32 positions x CW channels] tile is ONE global_load_dwordx4 per tensor at CW = 8) and of the B / C rows;
//   * optionally the Delta projection fused in (MS_SCAN_DT_FUSED: delta = softplus(dts @ Wdt^T + bias) is formed from the
//     R = dt_rank leading columns of the projection row while the tile is staged -- `delta` / `ddelta` tensors never
//     exist; the backward returns d dts into the projection-row gradient and d Wdt)  -- MedMamba.py:400,403-405;
//   * a lane's states as packed fp32 pairs (scan_common.h).
// Work mapping as in scan_fwd.hip / scan_bwd.hip: wave = CW channels x 16 states of one (batch, direction), lane = (state
// group sg, channel c); workgroup = the waves that share a 128-byte line (32 channels).
// Everything else (reference (B,D,L) layout, other d_state, scalar-decay SSD forms, unaligned tensors) runs on the
// general kernels of scan_fwd.hip / scan_bwd.hip.
#include <cstdlib>
#include <type_traits>
#include "scan_common.h"

namespace ms {

#ifdef MS_CLOCK
// diagnostic build only (tools/build_variant.sh clock "scan_ss2d.hip" "-DMS_CLOCK"): the clock the chip holds while the forward
// kernel runs = sum over waves of (shader cycles, s_memtime) / (100 MHz reference ticks, s_memrealtime) -- MI355X_MICROARCH.md
// "DVFS give-back" item 6.  The stamps go to a buffer of their own and feed nothing else.
__device__ unsigned long long ms_clock_acc[2];
extern "C" int ms_debug_clock(unsigned long long *out, int reset) {
    unsigned long long z[2] = {0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ms_clock_acc), sizeof(z)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(ms_clock_acc), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif

constexpr int kN = 16;              // d_state of the fast path
constexpr int kRPs = kN + 4;        // row pitch (floats) of the [position][state] B / C tiles (16-byte aligned rows)
constexpr int kMaxR = 32;           // largest fused dt_rank

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
// global accesses as (wave-uniform base pointer, 32-bit per-lane ELEMENT offset): one VGPR of address per lane and the
// scalar-base addressing mode instead of a 64-bit pointer pair per lane and tensor (the host validates the ranges)
__device__ __forceinline__ const float *at(const float *base, int off) {
    return reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (uint32_t)off * 4u);
}
__device__ __forceinline__ float *at(float *base, int off) {
    return reinterpret_cast<float *>(reinterpret_cast<char *>(base) + (uint32_t)off * 4u);
}

// geometry of the 16-byte activation accesses of one wave: lane -> (position lane / QPP + k * PPI, channel quad lane % QPP)
template <int CW> struct VecIO {
    static constexpr int QPP = CW / 4, PPI = 64 / QPP, NEV = kCL / PPI;
};

// workgroup -> (batch * n_groups + group, first channel block).  Workgroups are dealt round-robin over the 8 XCDs; the
// workgroups that share one (batch, group)'s projection rows get equal blockIdx % 8 (one XCD's L2) -- speed only.
__device__ __forceinline__ void wg_to_work(int bid, int npairs, int ncg, int &pair, int &cg) {
    const int full = (npairs / 8) * 8 * ncg;
    if (bid < full) { pair = (bid / (8 * ncg)) * 8 + bid % 8; cg = (bid / 8) % ncg; }
    else            { pair = (npairs / 8) * 8 + (bid - full) / ncg; cg = (bid - full) % ncg; }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward:  replaces selective_scan_fwd_kernel (selective_scan_fwd_kernel.cuh:67-303) + the eager ops of
// MedMamba.py:393-395 (cross-scan), :400 (dt_proj, when DTF), :420-424 (inverse permutations of the cross-merge)
// ---------------------------------------------------------------------------------------------------------------------
// SEG (inference at small batch: B * 4 * D / CW waves do not fill the chip, and each walks all L positions one after the other): the
// sequence is cut into `segments` runs of `cps` chunks, blockIdx.y = the segment.
//   SEG 1: scan the segment from the ZERO state with no output; at its end store the state h_end and the product of its decays P per
//          (channel, state) into the workspace p.x: plane 0 [batch][segment][state][dim] = h_end, plane 1 = P.
//   (ss2d_seg_carry_kernel: plane 0 [s] <- the state ENTERING segment s: H = P[s-1] H + h_end[s-1], sequential over the few segments)
//   SEG 2: the plain forward over the segment, started from plane 0 [s].
// Twice the recurrence work, `segments` times the parallelism; the recurrence is linear in the state, so the result is the
// unsegmented one up to rounding (h entering a segment is formed as P * H + h_end instead of position by position).
template <int CW, bool DTF, int SEG = 0>
__global__ void __launch_bounds__(64 * (32 / CW))
ss2d_fwd_kernel(const MsScanParams p, const int n_chunks, const int cps = 0, const int segments = 1) {
    constexpr int SG = 64 / CW, NPL = kN / SG, NP2 = NPL / 2, NW = 32 / CW, NT = 64 * NW;
    using V = VecIO<CW>;
    constexpr int QPP = V::QPP, PPI = V::PPI, NEV = V::NEV;
    constexpr int NBC = 2 * kCL * (kN / 4) / NT;            // float4 pieces of the chunk's B | C rows per thread
    constexpr int kRp = kMaxR + 1;                           // dts tile pitch (odd: conflict-free column reads)
    constexpr int NDT = DTF ? (kCL * kMaxR + NT - 1) / NT : 1;
    __shared__ __attribute__((aligned(16))) float sB[kCL * kRPs];
    __shared__ __attribute__((aligned(16))) float sC[kCL * kRPs];
    __shared__ __attribute__((aligned(16))) v2f sdd_[NW][kCL * CW];       // {delta', delta' * u}
    __shared__ __attribute__((aligned(16))) float so_[NW][kCL * CW];      // y of the chunk (the store adds D * u)
    __shared__ int spos_[NW][2][kCL];
    __shared__ float sDT[DTF ? kCL * kRp : 1];                            // dts of the chunk [position][r]
    __shared__ __attribute__((aligned(16))) float sW[DTF ? kMaxR * 32 : 4];   // Wdt of the workgroup's 32 channels [r][channel]
    const int lane = threadIdx.x & 63, tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v2f *sdd = sdd_[wv];
    float *so = so_[wv];
    int (*spos)[kCL] = spos_[wv];
    const int c = lane % CW, sg = lane / CW;
    const int quad = lane % QPP, my4 = 4 * quad, lp0 = lane / QPP;

    const int L = __builtin_amdgcn_readfirstlane(p.seqlen), R = DTF ? __builtin_amdgcn_readfirstlane(p.dt_rank) : 0;
    const int dpg = p.dim / p.n_groups, ncg = (dpg + 31) / 32;
    int pair, cg;
    wg_to_work(blockIdx.x, p.batch * p.n_groups, ncg, pair, cg);
    const int g = pair % p.n_groups, b = pair / p.n_groups;
    const int cw0 = wv * CW;                                  // this wave's first channel inside the workgroup's 32
    const int c0w = cg * 32 + cw0;                            // ... inside its group
    const int nvalid = max(0, min(CW, dpg - c0w));            // dpg % 4 == 0 (host): quads are valid or invalid as a whole
    const int d0 = g * dpg + c0w;
    const bool active = c < nvalid, quad_ok = my4 < nvalid;
    const int dsafe = min(d0 + (active ? c : 0), p.dim - 1);    // channel whose parameters this lane reads

    const int seg = SEG ? (int)blockIdx.y : 0;
    const int ch_begin = SEG ? seg * cps : 0, ch_end = SEG ? min(n_chunks, ch_begin + cps) : n_chunks;
    if (SEG && ch_begin >= ch_end) return;                        // (uniform over the workgroup: before any barrier)
    float *seg_h = SEG ? p.x + (((int64_t)b * segments + seg) * kN) * p.dim : nullptr;            // plane 0 row block of this (batch, segment)
    float *seg_p = SEG ? seg_h + (int64_t)p.batch * segments * kN * p.dim : nullptr;               // plane 1
    v2f A2[NP2], h[NP2], Pd[SEG == 1 ? NP2 : 1];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int n = sg * NPL + i;
        float av = active ? p.A[dsafe * p.A_d_stride + n * p.A_dstate_stride] : 0.0f;
        if ((p.delta_softplus & MS_SCAN_A_IS_LOG) && active) av = -__expf(av);
        A2[i / 2][i % 2] = av * kLog2e;
        h[i / 2][i % 2] = (SEG == 2 && active) ? seg_h[(int64_t)n * p.dim + dsafe] : 0.0f;
        if constexpr (SEG == 1) Pd[i / 2][i % 2] = 1.0f;
    }
    // MS_SCAN_DELTA_ACTIVATED: delta already holds softplus(raw + bias) (ms_dtproj_fwd_act); wave-uniform
    const bool pre = !DTF && (p.delta_softplus & MS_SCAN_DELTA_ACTIVATED) != 0;
    float D4[4], bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int dj = min(d0 + my4 + j, p.dim - 1);
        D4[j] = (p.D && quad_ok) ? p.D[dj] : 0.0f;
        bias4[j] = (p.delta_bias && quad_ok && !pre) ? p.delta_bias[dj] : 0.0f;
    }
    if (DTF) {          // Wdt rows of the workgroup's channels -> LDS, transposed to [r][channel]
        for (int e = tid; e < 32 * R; e += NT) {
            const int ch = e / R, r = e - ch * R;
            const int dj = g * dpg + cg * 32 + ch;
            sW[r * 32 + ch] = (cg * 32 + ch < dpg) ? p.dt_w[(int64_t)dj * R + r] : 0.0f;
        }
    }

    const int c0s = nvalid > 0 ? c0w : 0;                       // a wave past the last channel block loads the group's first channels (in bounds) and stores nothing
    const float *ub = p.u + b * p.u_batch_stride + g * p.u_group_stride + c0s;
    const float *db = DTF ? nullptr : p.delta + b * p.delta_batch_stride + g * p.delta_group_stride + c0s;
    // MS_SCAN_DELTA_OUT: the fused projection's delta' is also stored (for the backward launch of a training step)
    float *dob = (DTF && (p.delta_softplus & MS_SCAN_DELTA_OUT) && p.delta) ? const_cast<float *>(p.delta) + b * p.delta_batch_stride + g * p.delta_group_stride + c0s : nullptr;
    float *ob = p.out + b * p.out_batch_stride + g * p.out_group_stride + c0s;
    const float *Bb = p.B + b * p.B_batch_stride + g * p.B_group_stride;
    const float *Cb = p.C + b * p.C_batch_stride + g * p.C_group_stride;
    const float *Tb = DTF ? p.dt_x + b * p.B_batch_stride + g * p.B_group_stride : nullptr;   // dts: the B rows' strides
    const int u_sl = (int)p.u_l_stride, dl_sl = (int)p.delta_l_stride, o_sl = (int)p.out_l_stride;
    const int B_sl = (int)p.B_l_stride, C_sl = (int)p.C_l_stride;
    PosMap pm;
    pm.mode = g & 3; pm.L = L;
    pm.H = __builtin_amdgcn_readfirstlane(p.map_h); pm.W = __builtin_amdgcn_readfirstlane(p.map_w);
    pm.invH = 1.0f / (float)p.map_h; pm.tab = nullptr; pm.tab_base = 0;
    const unsigned sp_mask = (p.delta_softplus & MS_SCAN_SOFTPLUS) ? 0xFFFFFFFFu : 0u;
    const bool accumulate = (p.delta_softplus & MS_SCAN_ACCUMULATE) != 0;
    const float invR = DTF ? 1.0f / (float)R : 0.0f;

    float4 ru[NEV], rd[NEV], rBC[NBC];
    float rdt[NDT];
    float4 uk[NEV];
#ifdef MS_CLOCK
    const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rk0 = __builtin_amdgcn_s_memrealtime();
#endif
    auto fetch = [&](int ch) {
        const int l0 = ch * kCL;
        int *tab = spos[ch & 1];
        pm.fill_table(tab, l0, lane);           // positions past L are clamped to L - 1: every address below is valid
        wave_sync();
#pragma unroll
        for (int k = 0; k < NEV; ++k) {
            const int pos = tab[lp0 + k * PPI];
            const int cq = quad_ok ? my4 : 0;
            ru[k] = ld4(at(ub, __mul24(pos, u_sl) + cq));
            if (!DTF) rd[k] = ld4(at(db, __mul24(pos, dl_sl) + cq));
        }
#pragma unroll
        for (int j = 0; j < NBC; ++j) {
            const int e = tid + j * NT, isC = e / (kCL * 4), r = e % (kCL * 4);
            rBC[j] = ld4(at(isC ? Cb : Bb, __mul24(tab[r / 4], isC ? C_sl : B_sl) + 4 * (r % 4)));
        }
        if (DTF) {
#pragma unroll
            for (int j = 0; j < NDT; ++j) {
                const int e = min(tid + j * NT, kCL * R - 1);
                const int pl = (int)(((float)e + 0.5f) * invR);
                rdt[j] = *at(Tb, __mul24(tab[pl], B_sl) + (e - pl * R));
            }
        }
    };
    fetch(ch_begin);

    for (int ch = ch_begin; ch < ch_end; ++ch) {
        const int l0 = ch * kCL, len = min(kCL, L - l0);
        __syncthreads();                        // everyone is done with the previous chunk's shared tiles
#pragma unroll
        for (int j = 0; j < NBC; ++j) {
            const int e = tid + j * NT, isC = e / (kCL * 4), r = e % (kCL * 4);
            const float4 v = (r / 4 < len) ? rBC[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            st4((isC ? sC : sB) + (r / 4) * kRPs + 4 * (r % 4), v);
        }
        if (DTF) {
#pragma unroll
            for (int j = 0; j < NDT; ++j) {
                const int e = tid + j * NT;
                const int pl = (int)(((float)e + 0.5f) * invR);
                if (e < kCL * R) sDT[pl * kRp + (e - pl * R)] = rdt[j];
            }
        }
        __syncthreads();                        // the shared tiles are staged
        // {delta', delta' * u}: delta = dts . Wdt (DTF) or the loaded tensor, + bias, softplus -- once per element
#pragma unroll
        for (int k = 0; k < NEV; ++k) {
            const int pl = lp0 + k * PPI;
            const bool ok = pl < len && quad_ok;
            float raw[4];
            if (DTF) {
                v2f a01 = (v2f){bias4[0], bias4[1]}, a23 = (v2f){bias4[2], bias4[3]};
                for (int r = 0; r < R; ++r) {
                    const float4 w = ld4(sW + r * 32 + cw0 + my4);
                    const v2f t = splat(sDT[pl * kRp + r]);
                    a01 = pk_fma(t, (v2f){w.x, w.y}, a01);
                    a23 = pk_fma(t, (v2f){w.z, w.w}, a23);
                }
                raw[0] = a01.x; raw[1] = a01.y; raw[2] = a23.x; raw[3] = a23.y;
            } else {
                raw[0] = rd[k].x + bias4[0]; raw[1] = rd[k].y + bias4[1]; raw[2] = rd[k].z + bias4[2]; raw[3] = rd[k].w + bias4[3];
            }
            const float uu[4] = {ru[k].x, ru[k].y, ru[k].z, ru[k].w};
            float dl[4], us[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float sp;
                if (!DTF && pre) sp = raw[j];                            // delta' arrives activated (bias4 == 0: raw is the loaded value)
                else sp = bits_f((f_bits(softplus_ref(raw[j])) & sp_mask) | (f_bits(raw[j]) & ~sp_mask));
                dl[j] = ok ? sp : 0.0f;
                us[j] = ok ? uu[j] : 0.0f;
            }
            uk[k] = make_float4(us[0], us[1], us[2], us[3]);
            if (DTF && dob != nullptr && ok) st4(at(dob, __mul24(spos[ch & 1][pl], dl_sl) + my4), make_float4(dl[0], dl[1], dl[2], dl[3]));
            float *dst = reinterpret_cast<float *>(sdd + pl * CW + my4);
            st4(dst, make_float4(dl[0], dl[0] * us[0], dl[1], dl[1] * us[1]));
            st4(dst + 4, make_float4(dl[2], dl[2] * us[2], dl[3], dl[3] * us[3]));
        }
        wave_sync();
        if (ch + 1 < ch_end) fetch(ch + 1);    // lands while this chunk is computed

#ifndef MS_FWD_UNROLL
#define MS_FWD_UNROLL 8
#endif
        // PARTIAL (the sequence's last chunk when L % 32 != 0): 4-position batches past the end hold the scan identity (delta' = 0:
        // a = 1, b = 0) and are skipped whole -- bit-identical, and L = 49 / 196 (MedMamba-T stages 3 / 2) do not pay for 15 / 28
        // padded positions.  Full chunks keep the unpredicated loop (one basic block: a guard per batch ends the scheduling region).
        auto sweep = [&](auto partial) {
#pragma unroll MS_FWD_UNROLL
        for (int lb = 0; lb < kCL; lb += 4) {
            if (decltype(partial)::value && lb >= len) continue;
            float y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v2f dd = sdd[(lb + j) * CW + c];
                const float *bp = sB + (lb + j) * kRPs + sg * NPL, *cp = sC + (lb + j) * kRPs + sg * NPL;
                v2f y2 = splat(0.0f);
#pragma unroll
                for (int q = 0; q < NP2; ++q) {
                    const v2f Bv = *reinterpret_cast<const v2f *>(bp + 2 * q), Cv = *reinterpret_cast<const v2f *>(cp + 2 * q);
                    const v2f a = exp2_pk(splat(dd.x) * A2[q]);
                    h[q] = pk_fma(a, h[q], splat(dd.y) * Bv);
                    if constexpr (SEG == 1) Pd[q] = Pd[q] * a;
                    else y2 = q == 0 ? Cv * h[q] : pk_fma(Cv, h[q], y2);
                }
                y[j] = y2.x + y2.y;
            }
            if constexpr (SEG == 1) continue;                       // no output in the first pass
            const float yt = sum_groups_scatter4<CW>(y, lane);
            // 8-channel waves: both lanes of the last butterfly pair (lane bit 3) hold the total and store it to the same word -- no
            // predicate, and the last DPP add folds into one v_add_f32_dpp (behind `if (owner)` it was a v_mov_b32_dpp + v_add inside
            // an exec save / restore: 3 instructions more per batch)
            so[(lb + group_slot<CW>(lane)) * CW + c] = yt;
        }
        };
        if (len == kCL) sweep(std::false_type()); else sweep(std::true_type());
        if (SEG == 0 && p.x != nullptr && active) {
#pragma unroll
            for (int i = 0; i < NPL; ++i)
                p.x[(((int64_t)b * n_chunks + ch) * kN + sg * NPL + i) * p.dim + dsafe] = h[i / 2][i % 2];
        }
        wave_sync();
        if constexpr (SEG != 1) {
            const int *tab = spos[ch & 1];
#pragma unroll
            for (int k = 0; k < NEV; ++k) {
                const int pl = lp0 + k * PPI;
                if (pl < len && quad_ok) {
                    const float4 yv = ld4(so + pl * CW + my4);
                    float *o = at(ob, __mul24(tab[pl], o_sl) + my4);
                    float4 v = make_float4(fmaf(D4[0], uk[k].x, yv.x), fmaf(D4[1], uk[k].y, yv.y), fmaf(D4[2], uk[k].z, yv.z),
                                           fmaf(D4[3], uk[k].w, yv.w));
                    if (accumulate) { const float4 t = ld4(o); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
                    st4(o, v);
                }
            }
        }
        wave_sync();
    }
    if constexpr (SEG == 1) {
        if (active) {
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                seg_h[(int64_t)(sg * NPL + i) * p.dim + dsafe] = h[i / 2][i % 2];
                seg_p[(int64_t)(sg * NPL + i) * p.dim + dsafe] = Pd[i / 2][i % 2];
            }
        }
    }
#ifdef MS_CLOCK
    if (lane == 0) {
        atomicAdd(&ms_clock_acc[0], __builtin_amdgcn_s_memtime() - ck0);
        atomicAdd(&ms_clock_acc[1], __builtin_amdgcn_s_memrealtime() - rk0);
    }
#endif
}

// plane 0 [b][s][n][d] = h_end of segment s (from zero state), plane 1 = product of its decays  ->  plane 0 [s] = state entering s
__global__ void __launch_bounds__(256)
ss2d_seg_carry_kernel(float *__restrict__ ws, int batch, int segments, int64_t row /* kN * dim */) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)batch * row) return;
    const int64_t b = e / row, r = e - b * row;
    float *h = ws + (b * segments) * row + r, *pd = h + (int64_t)batch * segments * row;
    float H = 0.0f;
    for (int s2 = 0; s2 < segments; ++s2) {
        const float he = h[(int64_t)s2 * row], pv = pd[(int64_t)s2 * row];
        h[(int64_t)s2 * row] = H;
        H = fmaf(pv, H, he);
    }
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool fits24(int64_t v);

// Can the fast path take this problem?  (anything else goes to the general kernels)
bool ss2d_fast_ok(const MsScanParams &p) {
    if (p.map_h <= 0 || p.dstate != kN || p.A_dstate_stride == 0) return false;
    if (((p.delta_softplus >> 4) & 7) != 0) return false;                         // MS_SCAN_BC_MAP: SSD forms
    if (p.delta_softplus & MS_SCAN_LATTICE) return false;                         // stride-2 sub-lattices: general kernels
    const int dpg = p.dim / p.n_groups;
    if (dpg % 4 != 0) return false;
    const bool dtf = (p.delta_softplus & MS_SCAN_DT_FUSED) != 0;
    if (dtf && (p.dt_rank < 1 || p.dt_rank > kMaxR || !p.dt_x || !p.dt_w)) return false;
    if (!dtf && !p.delta) return false;
    if ((p.delta_softplus & MS_SCAN_DELTA_OUT) && !(dtf && p.delta)) return false;
    auto act_ok = [](const float *ptr, int64_t sb, int64_t sgp, int64_t sl) {
        return aligned16(ptr) && sb % 4 == 0 && sgp % 4 == 0 && sl % 4 == 0 && fits24(sl);
    };
    if (!act_ok(p.u, p.u_batch_stride, p.u_group_stride, p.u_l_stride)) return false;
    if ((!dtf || (p.delta_softplus & MS_SCAN_DELTA_OUT)) && !act_ok(p.delta, p.delta_batch_stride, p.delta_group_stride, p.delta_l_stride)) return false;
    if (!fits24(p.B_l_stride) || !fits24(p.C_l_stride) || !fits24(p.seqlen)) return false;
    return true;
}

int ss2d_fwd_launch(const MsScanParams &p, int n_chunks, hipStream_t stream) {
    if (!aligned16(p.out) || p.out_batch_stride % 4 || p.out_group_stride % 4 || p.out_l_stride % 4 || !fits24(p.out_l_stride))
        return MS_ERR_STRIDE;
    const int dpg = p.dim / p.n_groups, ncg = (dpg + 31) / 32;
    const dim3 grid((unsigned)((int64_t)p.batch * p.n_groups * ncg));
    const bool dtf = (p.delta_softplus & MS_SCAN_DT_FUSED) != 0;
    // 8-channel waves (8 state groups, 2 states per lane) when 16-channel waves would leave the chip under-filled
    static const int cw_env = [] { const char *e = getenv("MEDSCAN_FWD_CW"); return e ? atoi(e) : 0; }();      // experiments: force 8 / 16
    const bool cw8 = cw_env ? cw_env == 8 : (int64_t)p.batch * p.n_groups * ncg * 2 < 4096;      // measured (MedMamba-T bs 64): stage 1 (3072 16-channel waves) 177 vs 190 us with 8-channel waves; stage 2 (6144) equal
    // MS_SCAN_SEGMENTS (p.segments >= 2, p.x = workspace of ms_scan_seg_floats(..) floats, no saved states): the sequence in segments
    // (inference at small batch); 8-channel waves
    const int segs = p.segments >= 2 ? (p.segments < n_chunks ? p.segments : n_chunks) : 1;
    if (segs >= 2) {
        if (!p.x || (p.delta_softplus & (MS_SCAN_ACCUMULATE | MS_SCAN_DELTA_OUT))) return MS_ERR_UNSUPPORTED;
        const int cps = (n_chunks + segs - 1) / segs;
        const dim3 g2(grid.x, (unsigned)segs);
        if (dtf) hipLaunchKernelGGL((ss2d_fwd_kernel<8, true, 1>), g2, dim3(256), 0, stream, p, n_chunks, cps, segs);
        else     hipLaunchKernelGGL((ss2d_fwd_kernel<8, false, 1>), g2, dim3(256), 0, stream, p, n_chunks, cps, segs);
        const int64_t row = (int64_t)kN * p.dim;
        hipLaunchKernelGGL(ss2d_seg_carry_kernel, dim3((unsigned)((p.batch * row + 255) / 256)), dim3(256), 0, stream, p.x, p.batch, segs, row);
        if (dtf) hipLaunchKernelGGL((ss2d_fwd_kernel<8, true, 2>), g2, dim3(256), 0, stream, p, n_chunks, cps, segs);
        else     hipLaunchKernelGGL((ss2d_fwd_kernel<8, false, 2>), g2, dim3(256), 0, stream, p, n_chunks, cps, segs);
        return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
    }
    if (cw8) {
        if (dtf) hipLaunchKernelGGL((ss2d_fwd_kernel<8, true>), grid, dim3(256), 0, stream, p, n_chunks);
        else     hipLaunchKernelGGL((ss2d_fwd_kernel<8, false>), grid, dim3(256), 0, stream, p, n_chunks);
    } else {
        if (dtf) hipLaunchKernelGGL((ss2d_fwd_kernel<16, true>), grid, dim3(128), 0, stream, p, n_chunks);
        else     hipLaunchKernelGGL((ss2d_fwd_kernel<16, false>), grid, dim3(128), 0, stream, p, n_chunks);
    }
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
