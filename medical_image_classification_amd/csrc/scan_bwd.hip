// Selective-scan backward for gfx950 (MI355X).  Replaces selective_scan_bwd_kernel
// (/root/reference/CrossMamba/FusionMamba/selective_scan/selective_scan_bwd_kernel.cuh:75-489) and the
// hand-rolled suffix scan it needs (reverse_scan.cuh:86-401).
//
// Adjoint implemented (same closed form as the reference kernel, bwd_kernel.cuh:209-216,244-329,439-475):
//   dh_l   = C_l*g_l + a_{l+1}*dh_{l+1}                     (reverse recurrence, sequential in registers)
//   du_l   = D*g_l + sum_n dh_l*delta'_l*B_l
//   ddl_l  = sum_n [ dh_l*B_l*u_l + dh_l*A_n*(a_l*h_{l-1}) ]
//   dA_n  += dh_l*delta'_l*(a_l*h_{l-1}) ;  dB += dh_l*delta'_l*u_l ;  dC += g_l*h_l ;  dD += g_l*u_l
//   ddelta = ddl * softplus'(delta+bias),  softplus'(x) = sigmoid(x) = 1 - exp(-softplus(x))
//
// One wave = 16 channels x all states, lane (sg, c) owns NPL states of channel c (scan_common.h).
// Chunks of MS_SCAN_CHUNK positions are visited last->first (the next one is prefetched into registers).
// Per chunk: (1) a forward sweep from the saved state x[b,c-1] stores h at the start of every 4-position
// batch in LDS; (2) a reverse sweep recomputes (a, h) of one batch into a register window and runs the
// adjoint recurrence backwards over it.  Sums over the state axis (du, ddelta) are permlane
// reduce-scatters; the per-(l,n) dB/dC contributions are summed over the wave's 16 channels by a DPP
// reduce-scatter into a per-chunk LDS tile, flushed with 128-byte-row global atomics.  No barriers.
#include "scan_common.h"

namespace ms {

// Reduce NV (a power of two, zero-padded by the caller) values over the CW channel lanes of a wave
// (lane bits 0 .. log2(CW)-1, all inside one DPP row).  Levels at distance CW/2 ... 1; while more than one
// value is live a level halves the set (reduce-scatter), afterwards it is a plain butterfly add.
// `owner_index(cbits, r)` tells which input index ended up in slot r of the lane with channel bits cbits.
template <int NV, int CW>
struct ChannelReduce {
    static constexpr int kOut = NV >= CW ? NV / CW : 1;
    template <int S, int CNT>
    __device__ static __forceinline__ void level(float (&v)[NV], int lane) {
        if constexpr (CNT > 1) {
#pragma unroll
            for (int i = 0; i < CNT / 2; ++i) v[i] = xchg_add<S>(v[i], v[i + CNT / 2], lane);
        } else {
            constexpr int CTRL = S == 8 ? 0x128 : S == 4 ? 0x12C : S == 2 ? 0x4E : 0xB1;
            constexpr int CTRL2 = S == 4 ? 0x124 : CTRL;
            const float up = dpp_mov<CTRL>(v[0]), dn = dpp_mov<CTRL2>(v[0]);
            v[0] += (lane & S) ? dn : up;
        }
    }
    __device__ static __forceinline__ void run(float (&v)[NV], int lane) {
        if constexpr (CW == 16) {
            level<8, NV>(v, lane);
            level<4, (NV >= 2 ? NV / 2 : 1)>(v, lane);
            level<2, (NV >= 4 ? NV / 4 : 1)>(v, lane);
            level<1, (NV >= 8 ? NV / 8 : 1)>(v, lane);
        } else {
            level<4, NV>(v, lane);
            level<2, (NV >= 2 ? NV / 2 : 1)>(v, lane);
            level<1, (NV >= 4 ? NV / 4 : 1)>(v, lane);
        }
    }
    __device__ static __forceinline__ int owner_index(int cbits, int r) {
        int idx = r, cnt = NV;
        constexpr int LV = CW == 16 ? 4 : 3;
#pragma unroll
        for (int k = 0; k < LV; ++k) {
            const int bit = (cbits >> (LV - 1 - k)) & 1;
            if (cnt > 1) { idx += bit * (cnt / 2); cnt /= 2; }
        }
        return idx;
    }
    // lanes that own distinct results (when NV < CW several lanes hold the same total)
    __device__ static __forceinline__ bool is_owner(int cbits) {
        if (NV >= CW) return true;
        return (cbits & (CW / NV - 1)) == 0;        // the low channel bits only replicate
    }
};

constexpr int next_pow2(int v) { int r = 1; while (r < v) r *= 2; return r; }

constexpr int kWPB = 4;      // waves per workgroup in the backward: independent except for the per-chunk dB/dC combine

template <int NPL, int CW, int MODE>
__global__ void __launch_bounds__(64 * kWPB)
scan_bwd_kernel(const MsScanBwdParams q, const int n_chunks, const int ncb) {
    constexpr int SG = 64 / CW, NP = SG * NPL, NB = kCL / 4, NV = next_pow2(4 * NPL);
    using Tile = TileIO<MODE, CW>;
    using Rows = RowIO<MODE, NP>;
    using CR = ChannelReduce<NV, CW>;
    constexpr int kPitch = Tile::kPitch, kTile = Tile::kTile, kCW = CW;
    const MsScanParams &p = q.f;
    __shared__ __attribute__((aligned(16))) float sB_[kWPB][NP * kRowPitch];
    __shared__ __attribute__((aligned(16))) float sC_[kWPB][NP * kRowPitch];
    __shared__ __attribute__((aligned(16))) float sdB_[kWPB][NP * kRowPitch];   // this chunk's dB / dC of each wave's channels
    __shared__ __attribute__((aligned(16))) float sdC_[kWPB][NP * kRowPitch];
    __shared__ float su_[kWPB][kTile];       // u tile      -> du tile
    __shared__ float sdl_[kWPB][kTile];      // delta' tile
    __shared__ float sg__[kWPB][kTile];      // dout tile   -> ddelta tile
    __shared__ float sck_[kWPB][NB * NPL * 64];   // h at the start of each 4-position batch
    __shared__ float sbias_[kWPB][kCW];
    __shared__ int spos_[kWPB][2][kCL];      // SS2D mode: pixel positions of the chunk being computed / being prefetched
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *sB = sB_[wv], *sC = sC_[wv], *sdB = sdB_[wv], *sdC = sdC_[wv];
    float *su = su_[wv], *sdl = sdl_[wv], *sg_ = sg__[wv], *sck = sck_[wv], *sbias = sbias_[wv];
    int (*spos)[kCL] = spos_[wv];
    const int c = lane % CW, sg = lane / CW;

    const int N = p.dstate, L = p.seqlen;
    const int dpg = p.dim / p.n_groups;
    // workgroup -> (batch, group, channel block).  Workgroups are dealt round-robin over the 8 XCDs, so the
    // waves that share one (batch, group)'s B/C rows are given equal blockIdx % 8: they hit one XCD's L2
    // instead of making all eight fetch the same rows (speed only, never correctness).
    int pair, cb;
    {
        const int ncg = (ncb + kWPB - 1) / kWPB;            // workgroups per (batch, group)
        const int npairs = p.batch * p.n_groups, bid = blockIdx.x;
        const int full = (npairs / 8) * 8 * ncg;            // pairs that form complete groups of 8
        int cg;
        if (bid < full) { pair = (bid / (8 * ncg)) * 8 + bid % 8; cg = (bid / 8) % ncg; }
        else            { pair = (npairs / 8) * 8 + (bid - full) / ncg; cg = (bid - full) % ncg; }
        cb = cg * kWPB + wv;
    }
    const int g = pair % p.n_groups;
    const int b = pair / p.n_groups;
    // a wave past the last channel block (ncb not a multiple of kWPB) computes on zeros and only joins the barriers
    const bool wave_idle = cb >= ncb;
    if (wave_idle) cb = ncb - 1;
    const int nvalid = wave_idle ? 0 : min(kCW, dpg - cb * kCW);
    const int d0 = g * dpg + cb * kCW;
    const bool active = c < nvalid;
    const int d = d0 + (active ? c : max(nvalid, 1) - 1);

    float An[NPL], A2[NPL], dhc[NPL], dAacc[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int n = sg * NPL + i;
        An[i] = n < N ? p.A[d * p.A_d_stride + n * p.A_dstate_stride] : 0.0f;
        A2[i] = An[i] * kLog2e;
        dhc[i] = 0.0f; dAacc[i] = 0.0f;
    }
    const float Dv = (p.D != nullptr && sg == 0) ? p.D[d] : 0.0f;
    const float fD = sg == 0 ? 1.0f : 0.0f;            // dD is accumulated once per channel, by group 0
    float dDacc = 0.0f, dbacc = 0.0f;
    if (lane < kCW) sbias[lane] = p.delta_bias ? p.delta_bias[d0 + min(lane, max(nvalid, 1) - 1)] : 0.0f;

    const int c0w = cb * kCW;                                   // first channel of this wave inside its group
    const float *ub = p.u + b * p.u_batch_stride + g * p.u_group_stride + c0w * p.u_d_stride;
    const float *db = p.delta + b * p.delta_batch_stride + g * p.delta_group_stride + c0w * p.delta_d_stride;
    const float *gb = q.dout + b * q.dout_batch_stride + g * q.dout_group_stride + c0w * q.dout_d_stride;
    float *dub = q.du + b * q.du_batch_stride + g * q.du_group_stride + c0w * q.du_d_stride;
    float *ddb = q.ddelta + b * q.ddelta_batch_stride + g * q.ddelta_group_stride + c0w * q.ddelta_d_stride;
    PosMap pm;
    pm.mode = MODE == kModeSS2D ? (g & 3) : -1; pm.H = p.map_h; pm.W = p.map_w; pm.L = L;
    pm.invH = MODE == kModeSS2D ? 1.0f / (float)p.map_h : 0.0f;
    pm.tab = nullptr; pm.tab_base = 0;
    const float *Bb = p.B + b * p.B_batch_stride + g * p.B_group_stride;
    const float *Cb = p.C + b * p.C_batch_stride + g * p.C_group_stride;
    float *dBb = q.dB + b * q.dB_batch_stride + g * q.dB_group_stride;
    float *dCb = q.dC + b * q.dC_batch_stride + g * q.dC_group_stride;
    const bool softplus = p.delta_softplus != 0;

    const Tile tile(lane);
    const Rows rows(lane);
    const unsigned sp_mask = softplus ? 0xFFFFFFFFu : 0u;
    float ru[Tile::NE], rd[Tile::NE], rg[Tile::NE], rB[Rows::NE], rC[Rows::NE];
    auto fetch = [&](int ch) {
        const int l0 = ch * kCL, len = min(kCL, L - l0);
        if (MODE == kModeSS2D) { pm.fill_table(spos[ch & 1], l0, lane); wave_sync(); }
        tile.fetch(ru, ub, p.u_d_stride, p.u_l_stride, l0, pm, nvalid, len);
        tile.fetch(rd, db, p.delta_d_stride, p.delta_l_stride, l0, pm, nvalid, len);
        tile.fetch(rg, gb, q.dout_d_stride, q.dout_l_stride, l0, pm, nvalid, len);
        rows.fetch(rB, Bb, p.B_dstate_stride, p.B_l_stride, l0, pm, N, len);
        rows.fetch(rC, Cb, p.C_dstate_stride, p.C_l_stride, l0, pm, N, len);
    };
    fetch(n_chunks - 1);
    wave_sync();                                           // sbias visible

    for (int ch = n_chunks - 1; ch >= 0; --ch) {
        const int l0 = ch * kCL, len = min(kCL, L - l0);
        tile.put(su, ru);
        tile.put_delta(sdl, rd, sbias, sp_mask, nvalid, len);
        tile.put(sg_, rg);
        rows.put(sB, rB);
        rows.put(sC, rC);
        float h[NPL];
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int n = sg * NPL + i;
            h[i] = (ch > 0 && n < N) ? p.x[(((int64_t)b * n_chunks + (ch - 1)) * N + n) * p.dim + d] : 0.0f;
        }
        wave_sync();
        if (ch > 0) fetch(ch - 1);                         // lands while this chunk is computed

        // ---- forward sweep: h at the start of every 4-position batch -> LDS ---------------------------
#pragma unroll 1
        for (int kb = 0; kb < NB; ++kb) {
            const int lb = kb * 4;
            float Bv[NPL][4];
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                sck[(kb * NPL + i) * 64 + lane] = h[i];
                row4(sB + (sg * NPL + i) * kRowPitch, lb, Bv[i]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dl_ = sdl[(lb + j) * kPitch + c];
                const float du_ = dl_ * su[(lb + j) * kPitch + c];
#pragma unroll
                for (int i = 0; i < NPL; ++i)
                    h[i] = fmaf(exp2_fast(dl_ * A2[i]), h[i], du_ * Bv[i][j]);
            }
        }
        // ---- reverse sweep: per batch, recompute (a, h) into a 4-position register window, then run the
        //      adjoint recurrence backwards over the window --------------------------------------------
#pragma unroll 1
        for (int kb = NB - 1; kb >= 0; --kb) {
            const int lb = kb * 4;
            float Bv[NPL][4], Cv[NPL][4], hs[NPL];
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                hs[i] = sck[(kb * NPL + i) * 64 + lane];
                row4(sB + (sg * NPL + i) * kRowPitch, lb, Bv[i]);
                row4(sC + (sg * NPL + i) * kRowPitch, lb, Cv[i]);
            }
            float dl_[4], uu[4], gg[4], av[4][NPL], hv[4][NPL];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dl_[j] = sdl[(lb + j) * kPitch + c];
                uu[j] = su[(lb + j) * kPitch + c];
                gg[j] = sg_[(lb + j) * kPitch + c];
                const float du_ = dl_[j] * uu[j];
#pragma unroll
                for (int i = 0; i < NPL; ++i) {
                    av[j][i] = exp2_fast(dl_[j] * A2[i]);
                    hv[j][i] = fmaf(av[j][i], j > 0 ? hv[j > 0 ? j - 1 : 0][i] : hs[i], du_ * Bv[i][j]);
                }
            }
            float duv[4], ddv[4], vB[NV], vC[NV];
#pragma unroll
            for (int r = 4 * NPL; r < NV; ++r) { vB[r] = 0.0f; vC[r] = 0.0f; }
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                const float du_ = dl_[j] * uu[j];
                float du_l = Dv * gg[j], dd_l = 0.0f;
                dDacc = fmaf(fD * gg[j], uu[j], dDacc);
#pragma unroll
                for (int i = 0; i < NPL; ++i) {
                    const float hprev = j > 0 ? hv[j > 0 ? j - 1 : 0][i] : hs[i];
                    const float dhn = fmaf(Cv[i][j], gg[j], dhc[i]);
                    const float w = av[j][i] * hprev;
                    const float t = dhn * Bv[i][j];
                    du_l = fmaf(t, dl_[j], du_l);
                    dd_l = fmaf(t, uu[j], dd_l);
                    const float qv = dhn * w;
                    dd_l = fmaf(qv, An[i], dd_l);
                    dAacc[i] = fmaf(qv, dl_[j], dAacc[i]);
                    vB[j * NPL + i] = dhn * du_;
                    vC[j * NPL + i] = gg[j] * hv[j][i];
                    dhc[i] = av[j][i] * dhn;
                }
                duv[j] = du_l; ddv[j] = dd_l;
            }
            // sums over the state axis: group sg receives the totals of position lb + sg
            const float du_t = sum_groups_scatter4<CW>(duv, lane);
            float dd_t = sum_groups_scatter4<CW>(ddv, lane);
            if (is_group_owner<CW>(lane)) {
                const int lo = lb + group_slot<CW>(lane);
                if (softplus) dd_t *= sigmoid_from_softplus(sdl[lo * kPitch + c]);
                dbacc += dd_t;
                su[lo * kPitch + c] = du_t;                // in place: this batch's u / dout are in registers
                sg_[lo * kPitch + c] = dd_t;
            }
            // sums over the wave's 16 channels of the per-(position, state) dB / dC terms
            CR::run(vB, lane);
            CR::run(vC, lane);
            if (CR::is_owner(c)) {
#pragma unroll
                for (int r = 0; r < CR::kOut; ++r) {
                    const int idx = CR::owner_index(c, r);
                    const int j = idx / NPL, i = idx % NPL;
                    if (4 * NPL == NV || idx < 4 * NPL) {
                        sdB[(sg * NPL + i) * kRowPitch + lb + j] = vB[r];
                        sdC[(sg * NPL + i) * kRowPitch + lb + j] = vC[r];
                    }
                }
            }
        }
        wave_sync();
        if (MODE == kModeSS2D) { pm.tab = spos[ch & 1]; pm.tab_base = l0; }      // the prefetch moved pm to the next chunk
        tile.store(su, dub, q.du_d_stride, q.du_l_stride, l0, pm, nvalid, len);
        tile.store(sg_, ddb, q.ddelta_d_stride, q.ddelta_l_stride, l0, pm, nvalid, len);
        // flush the chunk's dB / dC tile (full rows -> 128-byte atomic segments in both row layouts)
        // combine the dB / dC tiles of the workgroup's waves (same batch and group, adjacent channel blocks) and add the
        // sums to global memory: kWPB x fewer atomics than one flush per wave.  The only two barriers of the chunk.
        __syncthreads();
        {
            constexpr int NT = 64 * kWPB, TOT = 2 * NP * kCL;
            for (int idx = threadIdx.x; idx < TOT; idx += NT) {
                const int isC = idx / (NP * kCL), rem = idx % (NP * kCL);
                int n, l;
                if (MODE == kModeSS2D) {            // (n, tensor) fastest: dB|dC of a pixel are adjacent in the projection row
                    const int t = idx % (2 * NP); l = idx / (2 * NP); n = t % NP;
                    if ((t >= NP) != (isC != 0)) { /* remap: idx enumerates (l, tensor, n) in SS2D mode */ }
                    const int tc = t >= NP;
                    float v = 0.0f;
#pragma unroll
                    for (int w = 0; w < kWPB; ++w) v += (tc ? sdC_[w] : sdB_[w])[n * kRowPitch + l];
                    if (n < N && l < len) {
                        float *base = tc ? dCb : dBb;
                        const int64_t sn = tc ? q.dC_dstate_stride : q.dB_dstate_stride, sl = tc ? q.dC_l_stride : q.dB_l_stride;
                        atomicAdd(base + pm.tab[l] * sl + n * sn, v);
                    }
                    continue;
                }
                l = rem % kCL; n = rem / kCL;
                float v = 0.0f;
#pragma unroll
                for (int w = 0; w < kWPB; ++w) v += (isC ? sdC_[w] : sdB_[w])[n * kRowPitch + l];
                if (n < N && l < len) {
                    float *base = isC ? dCb : dBb;
                    const int64_t sn = isC ? q.dC_dstate_stride : q.dB_dstate_stride, sl = isC ? q.dC_l_stride : q.dB_l_stride;
                    atomicAdd(base + (int64_t)(l0 + l) * sl + n * sn, v);
                }
            }
        }
        __syncthreads();
        wave_sync();
    }

    if (active) {
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int n = sg * NPL + i;
            if (n < N) atomicAdd(q.dA + (int64_t)d * N + n, dAacc[i]);
        }
        if (q.dD != nullptr && sg == 0) atomicAdd(q.dD + d, dDacc);
        if (q.ddelta_bias != nullptr) atomicAdd(q.ddelta_bias + d, dbacc);
    }
}

int validate_scan(const MsScanParams &p);
int pick_npl(int dstate, int sg);
bool use_cw8(const MsScanParams &p, bool backward);
int pick_mode(bool l_contig, bool d_contig, int map_h);
bool act_strides_ok(int64_t sd, int64_t sl, int seqlen);

template <int NPL, int CW>
static int launch_bwd(const MsScanBwdParams &q, int n_chunks, hipStream_t stream) {
    const MsScanParams &p = q.f;
    const int dpg = p.dim / p.n_groups;
    const int ncb = (dpg + CW - 1) / CW;
    const dim3 grid((unsigned)((int64_t)p.batch * p.n_groups * ((ncb + kWPB - 1) / kWPB)));
    const bool lcontig = p.u_l_stride == 1 && p.delta_l_stride == 1 && q.dout_l_stride == 1 &&
                         q.du_l_stride == 1 && q.ddelta_l_stride == 1;
    const bool dcontig = p.u_d_stride == 1 && p.delta_d_stride == 1 && q.dout_d_stride == 1 &&
                         q.du_d_stride == 1 && q.ddelta_d_stride == 1;
    switch (pick_mode(lcontig, dcontig, p.map_h)) {
        case kModeSS2D: hipLaunchKernelGGL((scan_bwd_kernel<NPL, CW, kModeSS2D>), grid, dim3(64 * kWPB), 0, stream, q, n_chunks, ncb); break;
        case kModeCL:   hipLaunchKernelGGL((scan_bwd_kernel<NPL, CW, kModeCL>), grid, dim3(64 * kWPB), 0, stream, q, n_chunks, ncb); break;
        default:        hipLaunchKernelGGL((scan_bwd_kernel<NPL, CW, kModeBDL>), grid, dim3(64 * kWPB), 0, stream, q, n_chunks, ncb); break;
    }
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int scan_bwd_dispatch(const MsScanBwdParams &q, hipStream_t stream) {
    const MsScanParams &p = q.f;
    int rc = validate_scan(p);
    if (rc != MS_OK) return rc;
    if (!q.dout || !q.du || !q.ddelta || !q.dA || !q.dB || !q.dC) return MS_ERR_NULL;
    if (!act_strides_ok(q.dout_d_stride, q.dout_l_stride, p.seqlen) || !act_strides_ok(q.du_d_stride, q.du_l_stride, p.seqlen) ||
        !act_strides_ok(q.ddelta_d_stride, q.ddelta_l_stride, p.seqlen) ||
        !act_strides_ok(q.dB_dstate_stride * 4, q.dB_l_stride, p.seqlen) || !act_strides_ok(q.dC_dstate_stride * 4, q.dC_l_stride, p.seqlen))
        return MS_ERR_STRIDE;
    if (p.map_h > 0 && (q.dout_d_stride != 1 || q.du_d_stride != 1 || q.ddelta_d_stride != 1 ||
                        q.dB_dstate_stride != 1 || q.dC_dstate_stride != 1)) return MS_ERR_STRIDE;
    if (p.batch == 0 || p.seqlen == 0) return MS_OK;
    const int n_chunks = (p.seqlen + kCL - 1) / kCL;
    if (n_chunks > 1 && !p.x) return MS_ERR_NULL;
    if (use_cw8(p, true)) {
        switch (pick_npl(p.dstate, 8)) {
            case 1: return launch_bwd<1, 8>(q, n_chunks, stream);
            case 2: return launch_bwd<2, 8>(q, n_chunks, stream);
        }
    }
    switch (pick_npl(p.dstate, 4)) {
        case 1: return launch_bwd<1, 16>(q, n_chunks, stream);
        case 2: return launch_bwd<2, 16>(q, n_chunks, stream);
        case 3: return launch_bwd<3, 16>(q, n_chunks, stream);
        case 4: return launch_bwd<4, 16>(q, n_chunks, stream);
    }
    return MS_ERR_DSTATE;       // backward is built for dstate <= 16
}

}  // namespace ms
