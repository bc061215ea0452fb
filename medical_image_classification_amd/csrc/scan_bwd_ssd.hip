// Backward of the SSD (Mamba-2 form) blocks' 4-direction scan over ALL four direction slices of the state axis in ONE launch
// (MS_SCAN_BC_MAP(4); CNN_Mamba.py:506-537 -- heads = (direction, head), every head's state is the concatenation of the four
// directions' B / C).  Mirror of the forward's all-directions launch (scan_fwd.hip, BCM == 2); replaces four launches of
// scan_bwd_kernel<2, 8, SS2D, SA, BCM> (one per slice, MS_SCAN_ACCUMULATE from the second on), each of which re-staged the
// u / delta' / dout tiles, re-evaluated the decay exp2 of every position and read-modify-wrote du / ddelta:
//   per 32-position chunk: tiles staged ONCE, decay a_l = exp2(delta'_l * A) evaluated ONCE (scalar decay per channel),
//   then for each slice j = 0..3: its B / C rows (read through direction j's pixel order) -> forward sweep from the slice's saved
//   state -> reverse sweep (packed state pairs, register channel sums: the machinery of scan_bwd.hip's packed path) -> dB / dC
//   of the slice flushed through direction j's order; du / ddelta accumulate in the wave's LDS tiles and are stored once.
// Work mapping as scan_bwd.hip: wave = 8 channels x 16 states of a slice (2 states per lane), 4 waves per workgroup sharing the
// B / C tiles.  Saved states: the forward's slice-major layout (4, batch, n_chunks, nd, dim).
#include "scan_common.h"

namespace ms {

namespace {
constexpr int kW = 4, kCW8 = 8, kNPL = 2, kSG = 8, kNP = 16, kNB = kCL / 4, kRP = kNP + 4;
}

__global__ void __launch_bounds__(64 * kW) __attribute__((amdgpu_waves_per_eu(2, 2)))
ssd_bwd_all_kernel(const MsScanBwdParams q, const int n_chunks, const int ncb) {
    using Tile = TileIO<kModeSS2D, kCW8>;
    using Rows = RowIO<kModeSS2D, kNP, kW>;
    constexpr int kPitch = Tile::kPitch, kTile = Tile::kTile;
    constexpr int kDC = kNP * kRowPitch + 2;
    const MsScanParams &p = q.f;
    __shared__ __attribute__((aligned(16))) float sB[kCL * kRP];
    __shared__ __attribute__((aligned(16))) float sC[kCL * kRP];
    __shared__ float sdBC_[2][kW][kDC + kNP * kRowPitch];
    __shared__ float su_[kW][kTile], sdl_[kW][kTile], sg__[kW][kTile];              // du out | delta' | ddelta' out
    // per (position, channel): {a = exp2(delta' A), delta', dout, delta' u} as ONE 16-byte element (every member broadcasts into the
    // packed sweeps through op_sel) and u; the decay is evaluated HERE, once per element -- not once per lane and position
    __shared__ __attribute__((aligned(16))) float4 sQ_[kW][kTile];
    __shared__ float sui_[kW][kTile];
    __shared__ float sbias_[kW][kCW8];
    __shared__ int spos_[kW][2][kCL];
    __shared__ int sposb_[kW][2][4][kCL];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *su = su_[wv], *sdl = sdl_[wv], *sg_ = sg__[wv], *sbias = sbias_[wv];
    float4 *sQ = sQ_[wv];
    float *sui = sui_[wv];
    int (*spos)[kCL] = spos_[wv];
    const int c = lane % kCW8, sg = lane / kCW8;
    const int t_dpp = (sg * kNPL + (c & 1)) * kRowPitch + (c >> 1);
    const int N = __builtin_amdgcn_readfirstlane(p.dstate) / 4;                     // states per direction slice
    const int L = __builtin_amdgcn_readfirstlane(p.seqlen);
    const int dpg = p.dim / p.n_groups;
    int pair, cb;
    {
        const int ncg = (ncb + kW - 1) / kW;
        const int npairs = p.batch * p.n_groups, bid = blockIdx.x;
        const int full = (npairs / 8) * 8 * ncg;
        int cg;
        if (bid < full) { pair = (bid / (8 * ncg)) * 8 + bid % 8; cg = (bid / 8) % ncg; }
        else            { pair = (npairs / 8) * 8 + (bid - full) / ncg; cg = (bid - full) % ncg; }
        cb = cg * kW + wv;
    }
    const int g = pair % p.n_groups, b = pair / p.n_groups;
    const bool wave_idle = cb >= ncb;
    if (wave_idle) cb = ncb - 1;
    const int nvalid = wave_idle ? 0 : min(kCW8, dpg - cb * kCW8);
    const int d0 = g * dpg + cb * kCW8;
    const bool active = c < nvalid;
    const int d = d0 + (active ? c : max(nvalid, 1) - 1);

    float As = p.A[d * p.A_d_stride];                                              // one decay rate per channel
    if (p.delta_softplus & MS_SCAN_A_IS_LOG) As = -__expf(As);
    const float A2s = As * kLog2e;
    const v2f Anp = {As, As};
    const float Dv = (p.D != nullptr && sg == 0) ? p.D[d] : 0.0f;
    if (lane < kCW8) sbias[lane] = p.delta_bias ? p.delta_bias[d0 + min(lane, max(nvalid, 1) - 1)] : 0.0f;

    const int c0w = cb * kCW8;
    const float *ub = p.u + b * p.u_batch_stride + g * p.u_group_stride + c0w;
    const float *db = p.delta + b * p.delta_batch_stride + g * p.delta_group_stride + c0w;
    const float *gb = q.dout + b * q.dout_batch_stride + g * q.dout_group_stride + c0w;
    float *dub = q.du + b * q.du_batch_stride + g * q.du_group_stride + c0w;
    float *ddb = q.ddelta + b * q.ddelta_batch_stride + g * q.ddelta_group_stride + c0w;
    PosMap pm;
    pm.setup(g, __builtin_amdgcn_readfirstlane(p.map_h), __builtin_amdgcn_readfirstlane(p.map_w), L, false);
    const float *Bb = p.B + b * p.B_batch_stride + g * p.B_group_stride;
    const float *Cb = p.C + b * p.C_batch_stride + g * p.C_group_stride;
    float *dBb = q.dB + b * q.dB_batch_stride + g * q.dB_group_stride;
    float *dCb = q.dC + b * q.dC_batch_stride + g * q.dC_group_stride;
    const int u_sl = (int)p.u_l_stride, dl_sl = (int)p.delta_l_stride, g_sl = (int)q.dout_l_stride;
    const int du_sl = (int)q.du_l_stride, dd_sl = (int)q.ddelta_l_stride;
    const int B_sl = (int)p.B_l_stride, C_sl = (int)p.C_l_stride, dB_sl = (int)q.dB_l_stride, dC_sl = (int)q.dC_l_stride;
    const unsigned sp_mask = (p.delta_softplus & MS_SCAN_SOFTPLUS) ? 0xFFFFFFFFu : 0u;
    const bool pre = (p.delta_softplus & MS_SCAN_DELTA_ACTIVATED) != 0;

    const Tile tile(lane);
    const Rows rows(threadIdx.x);
    float ru[Tile::NE], rd[Tile::NE], rg[Tile::NE], rB[Rows::NE], rC[Rows::NE];
    float dDk = 0.0f, dbk[1] = {0.0f};
    // saved states of the four slices at the start of a chunk: x[j][b][ch - 1][n][d] (consumed slice by slice and rotated; the
    // next chunk's are loaded into the same registers during the last slice, when all four have been consumed)
    float rx[4][kNPL];
    const float *xs0[kNPL];
    const int64_t x_chunk_stride = n_chunks > 1 ? (int64_t)N * p.dim : 0;
    const int64_t x_slice_stride = n_chunks > 1 ? (int64_t)p.batch * n_chunks * N * p.dim : 0;
#pragma unroll
    for (int i = 0; i < kNPL; ++i) {
        const int n = min(sg * kNPL + i, N - 1);
        xs0[i] = n_chunks > 1 ? p.x + ((int64_t)b * n_chunks * N + n) * p.dim + d : p.A;
    }
    auto dir_map = [&](int ch, int j) { PosMap m = pm; m.mode = j; m.tab = sposb_[wv][ch & 1][j]; m.tab_base = ch * kCL; return m; };
    auto fetch_states = [&](int ch) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < kNPL; ++i) rx[j][i] = xs0[i][j * x_slice_stride + (int64_t)max(ch - 1, 0) * x_chunk_stride];
    };
    auto fetch_tiles = [&](int ch) {
        const int l0 = ch * kCL, len = min(kCL, L - l0);
        pm.fill_table(spos[ch & 1], l0, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) { PosMap pj = pm; pj.mode = j; pj.fill_table(sposb_[wv][ch & 1][j], l0, lane); }
        pm.tab = spos[ch & 1]; pm.tab_base = l0;
        wave_sync();
        tile.fetch(ru, ub, 1, u_sl, l0, pm, nvalid, len);
        tile.fetch(rd, db, 1, dl_sl, l0, pm, nvalid, len);
        tile.fetch(rg, gb, 1, g_sl, l0, pm, nvalid, len);
    };
    auto fetch_rows = [&](int ch, int j) {
        const int l0 = ch * kCL, len = min(kCL, L - l0);
        const PosMap m = dir_map(ch, j);
        rows.fetch(rB, Bb, 1, B_sl, l0, m, N, len);
        rows.fetch(rC, Cb, 1, C_sl, l0, m, N, len);
    };
    fetch_tiles(n_chunks - 1);
    fetch_states(n_chunks - 1);
    fetch_rows(n_chunks - 1, 0);
    wave_sync();

    // adjoint carries of the four slices: dhp[0] is always the CURRENT slice's (the array is rotated after every slice, so the
    // slice loop can be a real loop -- unrolled it would be four copies of the sweeps, more than the instruction cache likes)
    v2f dhp[4], dAp = {0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 4; ++j) dhp[j] = (v2f){0.0f, 0.0f};
    const float *sBl = sB + sg * kNPL, *sCl = sC + sg * kNPL;

    for (int ch = n_chunks - 1; ch >= 0; --ch) {
        const int l0 = ch * kCL, len = min(kCL, L - l0);
        // ---- the chunk's activation tiles: staged once for the four slices -------------------------------------------
        tile.put_delta(sdl, rd, sbias, sp_mask, nvalid, len, pre);
#pragma unroll
        for (int k = 0; k < Tile::NE; ++k) {
            const bool ok = tile.ok(k, nvalid, len);
            ru[k] = ok ? ru[k] : 0.0f; rg[k] = ok ? rg[k] : 0.0f;
            const float dlv = sdl[tile.soff(k)];
            sQ[tile.soff(k)] = make_float4(exp2_fast(dlv * A2s), dlv, rg[k], dlv * ru[k]);      // tile channel == this lane's channel c
            sui[tile.soff(k)] = ru[k];
            dDk = fmaf(rg[k], ru[k], dDk);
        }
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            float *sdB = sdBC_[j & 1][wv], *sdC = sdBC_[j & 1][wv] + kDC;
            rows.template put_t<kNPL, kRP>(sB, rB, N, len);
            rows.template put_t<kNPL, kRP>(sC, rC, N, len);
            v2f hp = {(ch > 0 && sg * kNPL < N) ? rx[0][0] : 0.0f, (ch > 0 && sg * kNPL + 1 < N) ? rx[0][1] : 0.0f};
            {   // rotate the saved states: rx[0] is the next slice's (four rotations per chunk = identity)
                const float t0 = rx[0][0], t1 = rx[0][1];
#pragma unroll
                for (int r = 0; r < 3; ++r) { rx[r][0] = rx[r + 1][0]; rx[r][1] = rx[r + 1][1]; }
                rx[3][0] = t0; rx[3][1] = t1;
            }
            __syncthreads();                         // the slice's B / C tiles are staged by all waves
            if (j < 3) fetch_rows(ch, j + 1);        // next slice's rows: in flight during this slice's sweeps
            if (j == 0) {
                if (ch > 0) fetch_tiles(ch - 1);                 // next chunk's tiles (tables of the other parity)
                pm.tab = spos[ch & 1]; pm.tab_base = l0;         // (fetch_tiles moved the map: back to this chunk for the stores)
            }
            if (j == 3 && ch > 0) { fetch_states(ch - 1); fetch_rows(ch - 1, 0); }      // (rx: all four consumed, rotation = identity)
            // ---- forward sweep of the slice -------------------------------------------------------------------------
            v2f ckp[kNB];
#pragma unroll
            for (int l = 0; l < kCL; ++l) {
                if ((l & 7) == 0) __builtin_amdgcn_sched_barrier(0);    // bound how far the scheduler hoists the LDS reads (VGPRs)
                if ((l & 3) == 0) ckp[l >> 2] = hp;
                const float4 qq = sQ[l * kPitch + c];
                const v2f Bp = *reinterpret_cast<const v2f *>(sBl + l * kRP);
                hp = pk_fma((v2f){qq.x, qq.x}, hp, (v2f){qq.w, qq.w} * Bp);
            }
            // ---- reverse sweep -------------------------------------------------------------------------------------
            v2f dh = dhp[0];
            const float Dj = j == 0 ? Dv : 0.0f;     // D * dout enters du once
#pragma unroll
            for (int kb = kNB - 1; kb >= 0; --kb) {
                const int lb = kb * 4;
                __builtin_amdgcn_sched_barrier(0);
                v2f Bp[4], Cp[4], bu[4], hv[4];
                float4 qq[4];
                float uu[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    Bp[jj] = *reinterpret_cast<const v2f *>(sBl + (lb + jj) * kRP);
                    Cp[jj] = *reinterpret_cast<const v2f *>(sCl + (lb + jj) * kRP);
                    qq[jj] = sQ[(lb + jj) * kPitch + c]; uu[jj] = sui[(lb + jj) * kPitch + c];
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    bu[jj] = (v2f){qq[jj].w, qq[jj].w} * Bp[jj];
                    hv[jj] = pk_fma((v2f){qq[jj].x, qq[jj].x}, jj > 0 ? hv[jj > 0 ? jj - 1 : 0] : ckp[kb], bu[jj]);
                }
                float duv[4], ddv[4], vB8[8], vC8[8];
#pragma unroll
                for (int jj = 3; jj >= 0; --jj) {
                    const v2f gg = {qq[jj].z, qq[jj].z};
                    const v2f dhn = pk_fma(Cp[jj], gg, dh);
                    const v2f w = hv[jj] - bu[jj];
                    const v2f t1 = dhn * Bp[jj];
                    const v2f qv = dhn * w;
                    const v2f t2 = qv * Anp;
                    dAp = pk_fma(qv, (v2f){qq[jj].y, qq[jj].y}, dAp);
                    const v2f vb = dhn * (v2f){qq[jj].w, qq[jj].w}, vc = gg * hv[jj];
                    vB8[2 * jj] = vb.x; vB8[2 * jj + 1] = vb.y; vC8[2 * jj] = vc.x; vC8[2 * jj + 1] = vc.y;
                    dh = (v2f){qq[jj].x, qq[jj].x} * dhn;
                    const float s1 = t1.x + t1.y, s2 = t2.x + t2.y;
                    duv[jj] = fmaf(s1, qq[jj].y, Dj * qq[jj].z);
                    ddv[jj] = fmaf(s1, uu[jj], s2);
                }
                const float du_t = sum_groups_scatter4<kCW8>(duv, lane);
                const float dd_t = sum_groups_scatter4<kCW8>(ddv, lane);
                if (is_group_owner<kCW8>(lane)) {
                    const int lo = (lb + group_slot<kCW8>(lane)) * kPitch + c;
                    su[lo] = j == 0 ? du_t : su[lo] + du_t;              // the four slices add up in the wave's own tiles
                    sg_[lo] = j == 0 ? dd_t : sg_[lo] + dd_t;
                }
                sdB[t_dpp + lb] = chan_scatter8(vB8, lane);
                sdC[t_dpp + lb] = chan_scatter8(vC8, lane);
            }
            dhp[0] = dhp[1]; dhp[1] = dhp[2]; dhp[2] = dhp[3]; dhp[3] = dh;      // rotate: dhp[0] = next slice's carry
            wave_sync();
            if (j == 3) {
                tile.store(su, dub, 1, du_sl, l0, pm, nvalid, len);
                tile.store_ddelta(sg_, sdl, sp_mask, ddb, 1, dd_sl, l0, pm, nvalid, len, dbk);
            }
            __syncthreads();
            {   // dB / dC of this slice: the 4 waves' tiles summed, flushed through direction j's pixel order
                const int t = threadIdx.x & 31, lb0 = threadIdx.x >> 5;
                const int tc = t >> 4, n = t & 15;
                const float *src = sdBC_[j & 1][0] + (tc ? kDC : 0) + n * kRowPitch + lb0;
                constexpr int kWS = kDC + kNP * kRowPitch;
                float *base = (tc ? dCb : dBb) + n;
                const int sl = tc ? dC_sl : dB_sl;
                const int *tab = sposb_[wv][ch & 1][j];
                if (n < N) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int l = lb0 + 8 * i;
                        const float v = (src[8 * i] + src[kWS + 8 * i]) + (src[2 * kWS + 8 * i] + src[3 * kWS + 8 * i]);
                        if (l < len) atomicAdd(base + __mul24(tab[l], sl), v);
                    }
                }
            }
        }
    }

    if (active) {
        // dA: one decay per channel; the (dim, nd) accumulator of the sliced launches is kept (the host sums it over the states)
        if (sg * kNPL < N) atomicAdd(q.dA + (int64_t)d * N + sg * kNPL, (p.delta_softplus & MS_SCAN_A_IS_LOG) ? dAp.x * As : dAp.x);
        if (sg * kNPL + 1 < N) atomicAdd(q.dA + (int64_t)d * N + sg * kNPL + 1, (p.delta_softplus & MS_SCAN_A_IS_LOG) ? dAp.y * As : dAp.y);
    }
    float a = dDk, e = dbk[0];
#pragma unroll
    for (int m = kCW8; m < 64; m *= 2) { a += __shfl_xor(a, m); e += __shfl_xor(e, m); }
    if (lane < kCW8 && lane < nvalid) {
        if (q.dD != nullptr) atomicAdd(q.dD + d0 + lane, a);
        if (q.ddelta_bias != nullptr) atomicAdd(q.ddelta_bias + d0 + lane, e);
    }
}

int ssd_bwd_all_launch(const MsScanBwdParams &q, int n_chunks, hipStream_t stream) {
    const MsScanParams &p = q.f;
    const int dpg = p.dim / p.n_groups;
    const int ncb = (dpg + kCW8 - 1) / kCW8;
    const dim3 grid((unsigned)((int64_t)p.batch * p.n_groups * ((ncb + kW - 1) / kW)));
    hipLaunchKernelGGL(ssd_bwd_all_kernel, grid, dim3(64 * kW), 0, stream, q, n_chunks, ncb);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
