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
// One wave = CW channels x all states, lane (sg, c) owns NPL <= 2 states of channel c (scan_common.h).
// Chunks of MS_SCAN_CHUNK positions are visited last->first (the next one is prefetched into registers).
// Per chunk, fully unrolled: (1) a forward sweep from the saved state x[b,c-1] keeps the decay a = exp2(delta'*A)
// of all 32 positions in registers (so every exp2 is evaluated once per backward) and h at the start of every
// 4-position batch; (2) a reverse sweep rebuilds h of one batch from its checkpoint (mul + fma) and runs the
// adjoint recurrence backwards over it.  Sums over the state axis (du, ddelta) are permlane reduce-scatters;
// the per-(l,n) dB/dC contributions are summed over the wave's CW channels by a transpose through a wave-private
// LDS tile (conflict-free b32 writes, b128 row reads), then over the workgroup's 4 waves, then added to global
// memory with atomics.
// The benchmark's instantiation (2 states per lane, 8-channel waves, SS2D mode) takes the PACKED path (`kPk`): state pairs as
// v2f through v_pk_* in both sweeps, the per-position scalars staged as pairs, and the channel sums as a register reduce-scatter
// (banked DPP adds, scan_common.h chan_scatter8) instead of the LDS transpose -- MS_BWD_PK / MS_BWD_DPP below, DESIGN.md 3.3.
// The SSD blocks' backward over all four direction slices in one launch is its own kernel: scan_bwd_ssd.hip.
#include <cstdlib>
#include "scan_common.h"

namespace ms {

constexpr int kWPB = 4;      // waves per workgroup in the backward: independent except for the per-chunk dB/dC combine

#ifndef MS_BWD_PK
#define MS_BWD_PK 1      // packed fp32 state pairs in the sweeps (2 states per lane, SS2D mode): see the kPk blocks
#endif
#ifndef MS_BWD_DPP
#define MS_BWD_DPP 1     // channel sums of dB / dC in registers (banked DPP adds) instead of an LDS transpose; used WITH the packed
                         // sweeps only.  Measured (MedMamba-T bs 64, ms per step of the 10 SS2D backward launches, same box):
                         // scalar + LDS 5.94 | scalar + DPP 6.17 | packed + LDS 6.20 | packed + DPP 5.42.  Either change alone loses
                         // (more VALU for the DPP sums / the LDS pipe becomes the bound once the VALU work shrinks); together they win.
#endif
// SA: scalar decay per channel (A_dstate_stride == 0, the SSD form): one exp2 and one stored decay per position.
// BCM: the B/C rows and the dB/dC flush follow the pixel order of one fixed direction (MS_SCAN_BC_MAP, see scan_fwd.hip).
template <int NPL, int CW, int MODE, bool SA = false, bool BCM = false>
__global__ void __launch_bounds__(64 * kWPB) __attribute__((amdgpu_waves_per_eu(2, 2)))
scan_bwd_kernel(const MsScanBwdParams q, const int n_chunks, const int ncb) {
    constexpr int SG = 64 / CW, NP = SG * NPL, NB = kCL / 4;
    // transpose tile: rows of CW floats.  Writes: (r*SG + sg)*CW + c = r*64 + lane, one bank per lane.  Reads: lane rr sums
    // row rr with CW/4 ds_read_b128 taken in a rotated order, so that the 16 lanes of a pass touch 16 different bank quads
    // (the sum does not care about the order); a padded pitch of CW + 4 cost a 2-way conflict on every write.
    constexpr int kRows = 4 * NPL * SG, kTP = CW, kQ = CW / 4;
    static_assert(kRows <= 64, "one lane per (position, state) of a batch");
    using Tile = TileIO<MODE, CW>;
    using Rows = RowIO<MODE, NP, kWPB>;
    constexpr int kPitch = Tile::kPitch, kTile = Tile::kTile, kCW = CW;
    const MsScanParams &p = q.f;
    constexpr int kRowsLds = NP * kRowPitch > kCL * (NP + 4) ? NP * kRowPitch : kCL * (NP + 4);       // either tile layout fits
    __shared__ __attribute__((aligned(16))) float sB[kRowsLds];                // B / C rows of the chunk: one copy per workgroup
    __shared__ __attribute__((aligned(16))) float sC[kRowsLds];
    // this chunk's dB | dC of each wave's channels; the dC tile sits 2 banks past a multiple of 64 from the dB tile, so the
    // workgroup combine (which reads dB[n][l] and dC[n][l] in one instruction) finds them in different banks
    constexpr int kDC = NP * kRowPitch + 2;
    __shared__ float sdBC_[2][kWPB][kDC + NP * kRowPitch];      // double-buffered by chunk parity (see the combine below)
    __shared__ float su_[kWPB][kTile];       // u tile      -> du tile
    __shared__ float sdl_[kWPB][kTile];      // delta' tile
    __shared__ float sg__[kWPB][kTile];      // dout tile   -> ddelta tile
    __shared__ __attribute__((aligned(16))) float sTB_[kWPB][kRows * kTP];   // channel-sum transpose tiles (one batch)
    __shared__ __attribute__((aligned(16))) float sTC_[kWPB][kRows * kTP];
    __shared__ float sbias_[kWPB][kCW];
    __shared__ int spos_[kWPB][2][kCL];      // SS2D mode: pixel positions of the chunk being computed / being prefetched
    __shared__ int sposb_[BCM ? kWPB : 1][2][kCL];   // BCM: the same for the B/C rows' direction
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    float *su = su_[wv], *sdl = sdl_[wv], *sg_ = sg__[wv], *sTB = sTB_[wv], *sTC = sTC_[wv], *sbias = sbias_[wv];
    int (*spos)[kCL] = spos_[wv];
    const int c = lane % CW, sg = lane / CW;
    // register (DPP) channel sums: lane (sg, c) owns value c of its group's 8 = position lb + c / 2, state sg * NPL + c % 2
    // packed sweeps: a lane's two states are one v2f; the B / C tiles are laid out [position][state] (pitch kRPk) so that a pair is
    // one ds_read_b64
    // (packed sweeps and register channel sums go together: either alone loses, see MS_BWD_DPP)
    constexpr bool kPk = MS_BWD_PK && MS_BWD_DPP && NPL == 2 && CW == 8 && MODE == kModeSS2D;
    constexpr bool kDppSums = kPk;
    constexpr int kRPk = NP + 4;
    const int t_dpp = (sg * NPL + (c & 1)) * kRowPitch + (c >> 1);
    // transpose-reduce ownership: lane rr sums row rr = (j*NPL + i)*SG + sg' -> position lb + j, state sg'*NPL + i
    const int t_out = ((lane % SG) * NPL + (lane / SG) % NPL) * kRowPitch + lane / (NPL * SG);

    // scalars used inside the chunk loop are copied out of the 8-dword argument tuples (readfirstlane makes a fresh
    // SGPR): when the register allocator spills them it then reloads ONE lane instead of the whole tuple
    const int N = __builtin_amdgcn_readfirstlane(p.dstate), L = __builtin_amdgcn_readfirstlane(p.seqlen);
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
        An[i] = (n < N || SA) ? p.A[d * p.A_d_stride + (SA ? 0 : n) * p.A_dstate_stride] : 0.0f;
        if ((p.delta_softplus & MS_SCAN_A_IS_LOG) && (n < N || SA)) An[i] = -__expf(An[i]);
        A2[i] = An[i] * kLog2e;
        dhc[i] = 0.0f; dAacc[i] = 0.0f;
    }
    const float Dv = (p.D != nullptr && sg == 0) ? p.D[d] : 0.0f;     // D*g enters du once per channel, through group 0
    if (lane < kCW) sbias[lane] = p.delta_bias ? p.delta_bias[d0 + min(lane, max(nvalid, 1) - 1)] : 0.0f;

    const int c0w = cb * kCW;                                   // first channel of this wave inside its group
    const float *ub = p.u + b * p.u_batch_stride + g * p.u_group_stride + c0w * p.u_d_stride;
    const float *db = p.delta + b * p.delta_batch_stride + g * p.delta_group_stride + c0w * p.delta_d_stride;
    const float *gb = q.dout + b * q.dout_batch_stride + g * q.dout_group_stride + c0w * q.dout_d_stride;
    float *dub = q.du + b * q.du_batch_stride + g * q.du_group_stride + c0w * q.du_d_stride;
    float *ddb = q.ddelta + b * q.ddelta_batch_stride + g * q.ddelta_group_stride + c0w * q.ddelta_d_stride;
    PosMap pm;
    pm.mode = -1; pm.L = L; pm.H = 0; pm.W = 0; pm.invH = 0.0f; pm.tab = nullptr; pm.tab_base = 0;
    if (MODE == kModeSS2D)
        pm.setup(g, __builtin_amdgcn_readfirstlane(p.map_h), __builtin_amdgcn_readfirstlane(p.map_w), L,
                 (p.delta_softplus & MS_SCAN_LATTICE) != 0);
    PosMap pmb = pm;                         // B/C rows (and dB/dC): same order as the activations unless BCM
    if (BCM) pmb.mode = ((p.delta_softplus >> 4) & 7) - 1;
    const float *Bb = p.B + b * p.B_batch_stride + g * p.B_group_stride;
    const float *Cb = p.C + b * p.C_batch_stride + g * p.C_group_stride;
    float *dBb = q.dB + b * q.dB_batch_stride + g * q.dB_group_stride;
    float *dCb = q.dC + b * q.dC_batch_stride + g * q.dC_group_stride;
    const bool softplus = (p.delta_softplus & MS_SCAN_SOFTPLUS) != 0;

    // 32-bit copies of the strides the chunk loop needs (one SGPR each instead of slices of the 16-dword argument
    // tuples); channel-last and SS2D modes have unit channel / state strides (validated on the host)
    constexpr bool kGen = MODE == kModeBDL;
    const int u_sd = kGen ? (int)p.u_d_stride : 1, u_sl = (int)p.u_l_stride;
    const int dl_sd = kGen ? (int)p.delta_d_stride : 1, dl_sl = (int)p.delta_l_stride;
    const int g_sd = kGen ? (int)q.dout_d_stride : 1, g_sl = (int)q.dout_l_stride;
    const int du_sd = kGen ? (int)q.du_d_stride : 1, du_sl = (int)q.du_l_stride;
    const int dd_sd = kGen ? (int)q.ddelta_d_stride : 1, dd_sl = (int)q.ddelta_l_stride;
    constexpr bool kRowN = MODE == kModeSS2D;
    const int B_sn = kRowN ? 1 : (int)p.B_dstate_stride, B_sl = (int)p.B_l_stride;
    const int C_sn = kRowN ? 1 : (int)p.C_dstate_stride, C_sl = (int)p.C_l_stride;
    const int dB_sn = kRowN ? 1 : (int)q.dB_dstate_stride, dB_sl = (int)q.dB_l_stride;
    const int dC_sn = kRowN ? 1 : (int)q.dC_dstate_stride, dC_sl = (int)q.dC_l_stride;

    const Tile tile(lane);
    const Rows rows(threadIdx.x);
    const unsigned sp_mask = softplus ? 0xFFFFFFFFu : 0u;
    float ru[Tile::NE], rd[Tile::NE], rg[Tile::NE], rB[Rows::NE], rC[Rows::NE];
    // dD = sum g*u and ddelta_bias = sum ddelta are accumulated by the lane that stages / stores the element
    // (Tile::NE elements per chunk) instead of inside the sweeps; reduced over the wave at the end
    float dDk[Tile::NA], dbk[Tile::NA];
#pragma unroll
    for (int k = 0; k < Tile::NA; ++k) { dDk[k] = 0.0f; dbk[k] = 0.0f; }
    float rx[NPL];          // saved state at the start of the prefetched chunk (x[b, ch-1]); zero for the first chunk
    const float *xs0[NPL];  // &x[b, 0, n_i, d] (or a dummy valid word when single-chunk sequences carry no saved states)
    const int64_t x_chunk_stride = n_chunks > 1 ? (int64_t)N * p.dim : 0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int n = min(sg * NPL + i, N - 1);
        xs0[i] = n_chunks > 1 ? p.x + ((int64_t)b * n_chunks * N + n) * p.dim + d : p.A;
    }
    auto fetch = [&](int ch) {
        const int l0 = ch * kCL, len = min(kCL, L - l0);
        if (MODE == kModeSS2D) {
            pm.fill_table(spos[ch & 1], l0, lane);
            if (BCM) pmb.fill_table(sposb_[wv][ch & 1], l0, lane); else pmb = pm;
            wave_sync();
        }
        // the state load goes out with (and is waited for with) the tile loads: a load consumed inside the sweeps would
        // put an s_waitcnt vmcnt(0) there and expose the whole prefetch
#pragma unroll
        for (int i = 0; i < NPL; ++i) rx[i] = xs0[i][(int64_t)max(ch - 1, 0) * x_chunk_stride];
        tile.fetch(ru, ub, u_sd, u_sl, l0, pm, nvalid, len);
        tile.fetch(rd, db, dl_sd, dl_sl, l0, pm, nvalid, len);
        tile.fetch(rg, gb, g_sd, g_sl, l0, pm, nvalid, len);
        rows.fetch(rB, Bb, B_sn, B_sl, l0, pmb, N, len);
        rows.fetch(rC, Cb, C_sn, C_sl, l0, pmb, N, len);
    };
    fetch(n_chunks - 1);
    wave_sync();                                           // sbias visible

    for (int ch = n_chunks - 1; ch >= 0; --ch) {
        const int l0 = ch * kCL, len = min(kCL, L - l0);
        float *sdB = sdBC_[ch & 1][wv], *sdC = sdBC_[ch & 1][wv] + kDC;
        // packed path: the per-(position, channel) scalars of the sweeps are staged as PAIRS, {delta', u} and {dout, delta' * u}: one
        // ds_read_b64 each, and both halves of a pair broadcast into v_pk_* through op_sel (a lone scalar in an odd register costs a
        // v_mov); delta' * u is formed once per element here instead of once per lane and position in each sweep.  The pair tiles
        // live in the transpose tiles the register channel sums no longer need; su / sg_ are pure output tiles (du, ddelta').
        v2f *sP1 = reinterpret_cast<v2f *>(sTB), *sP2 = reinterpret_cast<v2f *>(sTC);
        tile.put_delta(sdl, rd, sbias, sp_mask, nvalid, len, (p.delta_softplus & MS_SCAN_DELTA_ACTIVATED) != 0);
        if constexpr (kPk) {
#pragma unroll
            for (int k = 0; k < Tile::NE; ++k) {
                const bool ok = tile.ok(k, nvalid, len);
                ru[k] = ok ? ru[k] : 0.0f; rg[k] = ok ? rg[k] : 0.0f;
                const float dlv = sdl[tile.soff(k)];                      // this lane wrote it just above
                sP1[tile.soff(k)] = (v2f){dlv, ru[k]};
                sP2[tile.soff(k)] = (v2f){rg[k], dlv * ru[k]};
            }
        } else {
            tile.put(su, ru, nvalid, len);
            tile.put(sg_, rg, nvalid, len);
        }
#pragma unroll
        for (int k = 0; k < Tile::NE; ++k) dDk[Tile::ak(k)] = fmaf(rg[k], ru[k], dDk[Tile::ak(k)]);       // out-of-range elements are zero
        if constexpr (kPk) {
            rows.template put_t<NPL, kRPk>(sB, rB, N, len);
            rows.template put_t<NPL, kRPk>(sC, rC, N, len);
        } else {
            rows.put(sB, rB, N, len);
            rows.put(sC, rC, N, len);
        }
        float h[NPL];
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int n = sg * NPL + i;
            h[i] = (ch > 0 && n < N) ? rx[i] : 0.0f;
        }
#ifndef MS_ABL_NOBAR
        __syncthreads();                                   // the B/C tiles are staged by all waves of the workgroup
#endif
        if (ch > 0) fetch(ch - 1);                         // lands while this chunk is computed

#ifdef MS_ABL_NOSWEEP
        if constexpr (false) {
#else
        if constexpr (kPk) {
#endif
            // ================= packed sweeps (same algebra as below, two states per instruction) =================
            const v2f A2p = {A2[0], A2[1]}, Anp = {An[0], An[1]};
            v2f hp = {h[0], h[1]}, dhp = {dhc[0], dhc[1]}, dAp = {dAacc[0], dAacc[1]};
            v2f ap[kCL], ckp[NB];
            const float *sBl = sB + sg * NPL, *sCl = sC + sg * NPL;          // this lane's state pair inside a position row
            // 4-position batches past the end of the sequence are skipped whole (wave-uniform branch): their elements are the scan
            // identity (delta' = 0: a = 1, b = 0; dout = 0), so h, dh and every sum pass through them unchanged -- bit-identical, and
            // L = 49 / 196 (MedMamba-T stages 3 / 2) no longer pay for 15 / 28 padded positions of their last chunk
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
                if (kb * 4 >= len) continue;
#pragma unroll
                for (int l = kb * 4; l < kb * 4 + 4; ++l) {
                    if ((l & 3) == 0) ckp[l >> 2] = hp;
                    const v2f p1 = sP1[l * kPitch + c], p2 = sP2[l * kPitch + c];      // {delta', u}, {dout, delta' u}
                    const v2f Bp = *reinterpret_cast<const v2f *>(sBl + l * kRPk);
                    if constexpr (SA) { const float a = exp2_fast(p1.x * A2[0]); ap[l] = (v2f){a, a}; }      // one decay per channel
                    else ap[l] = exp2_pk((v2f){p1.x, p1.x} * A2p);
                    hp = pk_fma(ap[l], hp, (v2f){p2.y, p2.y} * Bp);
                }
            }
#pragma unroll
            for (int kb = NB - 1; kb >= 0; --kb) {
                const int lb = kb * 4;
                if (lb >= len) continue;
                v2f Bp[4], Cp[4], bu[4], hv[4], p1[4], p2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    Bp[j] = *reinterpret_cast<const v2f *>(sBl + (lb + j) * kRPk);
                    Cp[j] = *reinterpret_cast<const v2f *>(sCl + (lb + j) * kRPk);
                    p1[j] = sP1[(lb + j) * kPitch + c]; p2[j] = sP2[(lb + j) * kPitch + c];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bu[j] = (v2f){p2[j].y, p2[j].y} * Bp[j];
                    hv[j] = pk_fma(ap[lb + j], j > 0 ? hv[j > 0 ? j - 1 : 0] : ckp[kb], bu[j]);
                }
                float duv[4], ddv[4], vB8[8], vC8[8];
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    const v2f gg = {p2[j].x, p2[j].x};
                    const v2f dhn = pk_fma(Cp[j], gg, dhp);
                    const v2f w = hv[j] - bu[j];                         // = a_j * h_{j-1}
                    const v2f t1 = dhn * Bp[j];
                    const v2f qv = dhn * w;
                    const v2f t2 = qv * Anp;
                    dAp = pk_fma(qv, (v2f){p1[j].x, p1[j].x}, dAp);
                    const v2f vb = dhn * (v2f){p2[j].y, p2[j].y}, vc = gg * hv[j];
                    vB8[2 * j] = vb.x; vB8[2 * j + 1] = vb.y; vC8[2 * j] = vc.x; vC8[2 * j + 1] = vc.y;
                    dhp = ap[lb + j] * dhn;
                    const float s1 = t1.x + t1.y, s2 = t2.x + t2.y;
                    duv[j] = fmaf(s1, p1[j].x, Dv * p2[j].x);
                    ddv[j] = fmaf(s1, p1[j].y, s2);
                }
#ifdef MS_ABL_NOSTATE
                const float du_t = (duv[0] + duv[1]) + (duv[2] + duv[3]), dd_t = (ddv[0] + ddv[1]) + (ddv[2] + ddv[3]);
#else
                const float du_t = sum_groups_scatter4<CW>(duv, lane);
                const float dd_t = sum_groups_scatter4<CW>(ddv, lane);
#endif
                if (is_group_owner<CW>(lane)) {
                    const int lo = lb + group_slot<CW>(lane);
                    su[lo * kPitch + c] = du_t;
                    sg_[lo * kPitch + c] = dd_t;
                }
#ifdef MS_ABL_NOCHAN
                sdB[t_dpp + lb] = ((vB8[0] + vB8[1]) + (vB8[2] + vB8[3])) + ((vB8[4] + vB8[5]) + (vB8[6] + vB8[7]));
                sdC[t_dpp + lb] = ((vC8[0] + vC8[1]) + (vC8[2] + vC8[3])) + ((vC8[4] + vC8[5]) + (vC8[6] + vC8[7]));
#else
                sdB[t_dpp + lb] = chan_scatter8(vB8, lane);
                sdC[t_dpp + lb] = chan_scatter8(vC8, lane);
#endif
            }
            dhc[0] = dhp.x; dhc[1] = dhp.y; dAacc[0] = dAp.x; dAacc[1] = dAp.y;
#ifdef MS_ABL_NOSWEEP
        } else if constexpr (!kPk) {
#else
        } else {
#endif
        // ---- forward sweep: the decay a of EVERY position stays in registers (each exp2 is evaluated once per
        //      backward), h only at the start of every 4-position batch ------------------------------------
        constexpr int NA = SA ? 1 : NPL;                       // decays stored per position
        float av[kCL][NA], ck[NB][NPL];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            const int lb = kb * 4;
            float Bv[NPL][4];
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                ck[kb][i] = h[i];
                row4(sB + (sg * NPL + i) * kRowPitch, lb, Bv[i]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dl_ = sdl[(lb + j) * kPitch + c];
                const float du_ = dl_ * su[(lb + j) * kPitch + c];
#pragma unroll
                for (int i = 0; i < NPL; ++i) {
                    if (!SA || i == 0) av[lb + j][SA ? 0 : i] = exp2_fast(dl_ * A2[i]);
                    h[i] = fmaf(av[lb + j][SA ? 0 : i], h[i], du_ * Bv[i][j]);
                }
            }
        }
        // ---- reverse sweep: per batch, rebuild h of its 4 positions from the checkpoint and the stored a
        //      (mul + fma, no exp), then run the adjoint recurrence backwards over the window ---------------
        // Operands of a batch (B/C rows, delta', u, dout of its 4 positions) are read from LDS ONE BATCH AHEAD, right before
        // the previous batch's transpose-reduce (whose two LDS round trips then overlap with these loads instead of
        // following them: the wave fences around the transpose keep the compiler from doing this itself).
        float Bq[2][NPL][4], Cq[2][NPL][4], dq[2][4], uq[2][4], gq[2][4];
        auto load_batch = [&](int kb) {
            const int lb = kb * 4, pq = kb & 1;
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                row4(sB + (sg * NPL + i) * kRowPitch, lb, Bq[pq][i]);
                row4(sC + (sg * NPL + i) * kRowPitch, lb, Cq[pq][i]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dq[pq][j] = sdl[(lb + j) * kPitch + c];
                uq[pq][j] = su[(lb + j) * kPitch + c];
                gq[pq][j] = sg_[(lb + j) * kPitch + c];
            }
        };
        load_batch(NB - 1);
#pragma unroll
        for (int kb = NB - 1; kb >= 0; --kb) {
            const int lb = kb * 4;
            float (&Bv)[NPL][4] = Bq[kb & 1], (&Cv)[NPL][4] = Cq[kb & 1];
            float (&dl_)[4] = dq[kb & 1], (&uu)[4] = uq[kb & 1], (&gg)[4] = gq[kb & 1];
            float bu[4][NPL], hv[4][NPL];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float du_ = dl_[j] * uu[j];
#pragma unroll
                for (int i = 0; i < NPL; ++i) {
                    bu[j][i] = du_ * Bv[i][j];
                    hv[j][i] = fmaf(av[lb + j][SA ? 0 : i], j > 0 ? hv[j > 0 ? j - 1 : 0][i] : ck[kb][i], bu[j][i]);
                }
            }
            float duv[4], ddv[4], vB[4 * NPL], vC[4 * NPL];
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                const float du_ = dl_[j] * uu[j];
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int i = 0; i < NPL; ++i) {
                    const float dhn = fmaf(Cv[i][j], gg[j], dhc[i]);
                    const float w = hv[j][i] - bu[j][i];             // = a_j * h_{j-1}
                    s1 = fmaf(dhn, Bv[i][j], s1);
                    const float qv = dhn * w;
                    s2 = fmaf(qv, An[i], s2);
                    dAacc[i] = fmaf(qv, dl_[j], dAacc[i]);
                    vB[j * NPL + i] = dhn * du_;
                    vC[j * NPL + i] = gg[j] * hv[j][i];
                    dhc[i] = av[lb + j][SA ? 0 : i] * dhn;
                }
                duv[j] = fmaf(s1, dl_[j], Dv * gg[j]);               // du_l  = D g + delta' sum_n dh B
                ddv[j] = fmaf(s1, uu[j], s2);                         // ddl_l = u sum_n dh B + sum_n dh A (a h_prev)
            }
            // sums over the state axis: group sg receives the totals of position lb + sg
            const float du_t = sum_groups_scatter4<CW>(duv, lane);
            const float dd_t = sum_groups_scatter4<CW>(ddv, lane);
            if (is_group_owner<CW>(lane)) {
                const int lo = lb + group_slot<CW>(lane);
                su[lo * kPitch + c] = du_t;                // in place: this batch's u / dout are in registers
                sg_[lo * kPitch + c] = dd_t;               // d delta' ; the softplus derivative is applied by the store
            }
            if (kb > 0) load_batch(kb - 1);              // next batch's operands: in flight during the transpose below
            if constexpr (kDppSums) {
                float vB8[8], vC8[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) { vB8[r] = vB[r]; vC8[r] = vC[r]; }
                sdB[t_dpp + lb] = chan_scatter8(vB8, lane);
                sdC[t_dpp + lb] = chan_scatter8(vC8, lane);
                continue;
            }
            // sums over the wave's CW channels of the per-(position, state) dB / dC terms: transpose through LDS.
            // Lane (sg, c) writes its 4*NPL values into rows (r*SG + sg), column c (64 consecutive floats per r);
            // lane rr then reads row rr (CW consecutive floats), adds them and owns (position, state) = row_of(rr).
            wave_sync();
#pragma unroll
            for (int r = 0; r < 4 * NPL; ++r) {
                sTB[(r * SG + sg) * kTP + c] = vB[r];
                sTC[(r * SG + sg) * kTP + c] = vC[r];
            }
            wave_sync();
            if (kRows == 64 || lane < kRows) {
                float tb = 0.0f, tc = 0.0f;
#pragma unroll
                for (int k4 = 0; k4 < kQ; ++k4) {
                    const int q4 = 4 * ((k4 + lane / (16 / kQ)) % kQ);
                    const float4 x = *reinterpret_cast<const float4 *>(sTB + lane * kTP + q4);
                    const float4 y = *reinterpret_cast<const float4 *>(sTC + lane * kTP + q4);
                    tb += (x.x + x.y) + (x.z + x.w);
                    tc += (y.x + y.y) + (y.z + y.w);
                }
                sdB[t_out + lb] = tb;
                sdC[t_out + lb] = tc;
            }
        }
        }   // !kPk
        wave_sync();
        if (MODE == kModeSS2D) {                                                 // the prefetch moved the maps to the next chunk
            pm.tab = spos[ch & 1]; pm.tab_base = l0;
            if (BCM) { pmb.tab = sposb_[wv][ch & 1]; pmb.tab_base = l0; } else pmb = pm;
        }
#ifndef MS_ABL_NOSTORE
        if (p.delta_softplus & MS_SCAN_ACCUMULATE) {
            tile.template store<true>(su, dub, du_sd, du_sl, l0, pm, nvalid, len);
            tile.template store_ddelta<true>(sg_, sdl, sp_mask, ddb, dd_sd, dd_sl, l0, pm, nvalid, len, dbk);
        } else {
            tile.store(su, dub, du_sd, du_sl, l0, pm, nvalid, len);
            tile.store_ddelta(sg_, sdl, sp_mask, ddb, dd_sd, dd_sl, l0, pm, nvalid, len, dbk);
        }
#endif
        // flush the chunk's dB / dC tile (full rows -> 128-byte atomic segments in both row layouts)
        // combine the dB / dC tiles of the workgroup's waves (same batch and group, adjacent channel blocks) and add the
        // sums to global memory: kWPB x fewer atomics than one flush per wave.  One barrier: the tiles are double-buffered
        // by chunk parity, so the next chunk's sweeps write the other buffer while slower waves still read this one, and
        // this buffer is written again only after everyone has passed the NEXT chunk's barrier, i.e. finished this combine.
#ifndef MS_ABL_NOBAR
        __syncthreads();
#endif
#ifdef MS_ABL_NOFLUSH
        if constexpr (false) {
#else
        if constexpr (MODE == kModeSS2D && NP == 16 && kWPB == 4 && kCL == 32) {
#endif
            // thread -> (tensor, state) = tid % 32 fixed, positions tid / 32 + 8 i: a wave's atomics cover two whole projection-row
            // segments [dB(16) | dC(16)] (a version with four consecutive states per thread quadrupled the L2 atomic transactions:
            // 7.3 vs 5.3 ms per step); everything that does not depend on i is hoisted, LDS reads at immediate offsets
            const int t = threadIdx.x & 31, lb0 = threadIdx.x >> 5;
            const int tc = t >> 4, n = t & 15;
            const float *src = sdBC_[ch & 1][0] + (tc ? kDC : 0) + n * kRowPitch + lb0;
            constexpr int kWS = kDC + NP * kRowPitch;                  // one wave's dB | dC tile
            float *base = (tc ? dCb : dBb) + n;
            const int sl = tc ? dC_sl : dB_sl;
            if (n < N) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int l = lb0 + 8 * i;
                    const float v = (src[8 * i] + src[kWS + 8 * i]) + (src[2 * kWS + 8 * i] + src[3 * kWS + 8 * i]);
#ifdef MS_ABL_NOATOMIC
                    if (l < len && v == 12345.678f) atomicAdd(base + __mul24(pmb.tab[l], sl), v);      // diagnostic build: cost of the flush
#else
                    if (l < len) atomicAdd(base + __mul24(pmb.tab[l], sl), v);
#endif
                }
            }
#ifdef MS_ABL_NOFLUSH
        } else if constexpr (false) {
#else
        } else {
#endif
            constexpr int NT = 64 * kWPB, TOT = 2 * NP * kCL;
            for (int idx = threadIdx.x; idx < TOT; idx += NT) {
                const int isC = idx / (NP * kCL), rem = idx % (NP * kCL);
                int n, l;
                if (MODE == kModeSS2D) {            // (n, tensor) fastest: dB|dC of a pixel are adjacent in the projection row
                    const int t = idx % (2 * NP); l = idx / (2 * NP); n = t % NP;            // idx enumerates (l, tensor, n) here
                    const int tc = t >= NP;
                    float v = 0.0f;
#pragma unroll
                    for (int w = 0; w < kWPB; ++w) v += sdBC_[ch & 1][w][(tc ? kDC : 0) + n * kRowPitch + l];
                    if (n < N && l < len) {
                        float *base = tc ? dCb : dBb;
                        atomicAdd(base + __mul24(pmb.tab[l], tc ? dC_sl : dB_sl) + n, v);
                    }
                    continue;
                }
                l = rem % kCL; n = rem / kCL;
                float v = 0.0f;
#pragma unroll
                for (int w = 0; w < kWPB; ++w) v += sdBC_[ch & 1][w][(isC ? kDC : 0) + n * kRowPitch + l];
                if (n < N && l < len) {
                    float *base = isC ? dCb : dBb;
                    atomicAdd(base + (int64_t)(l0 + l) * (isC ? dC_sl : dB_sl) + n * (isC ? dC_sn : dB_sn), v);
                }
            }
        }
        wave_sync();
    }

    if (active) {
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int n = sg * NPL + i;
            // A = -exp(A_log)  =>  dL/dA_log = dL/dA * A
            if (n < N) atomicAdd(q.dA + (int64_t)d * N + n, (p.delta_softplus & MS_SCAN_A_IS_LOG) ? dAacc[i] * An[i] : dAacc[i]);
        }
    }
    // dD / ddelta_bias: every lane holds partial sums of the tile elements it staged / stored
    if constexpr (Tile::LCONTIG) {            // lane = one position of channels ck(k)
#pragma unroll
        for (int k = 0; k < Tile::NE; ++k) {
            float a = dDk[k], e = dbk[k];
#pragma unroll
            for (int m = 1; m < kCL; m *= 2) { a += __shfl_xor(a, m); e += __shfl_xor(e, m); }      // over the 32 positions
            if ((lane & (kCL - 1)) == 0 && tile.ck(k) < nvalid) {
                if (q.dD != nullptr) atomicAdd(q.dD + d0 + tile.ck(k), a);
                if (q.ddelta_bias != nullptr) atomicAdd(q.ddelta_bias + d0 + tile.ck(k), e);
            }
        }
    } else {                                  // lane = channel lane % CW of positions lane / CW + (64/CW) k
        float a = dDk[0], e = dbk[0];
#pragma unroll
        for (int m = CW; m < 64; m *= 2) { a += __shfl_xor(a, m); e += __shfl_xor(e, m); }
        if (lane < CW && lane < nvalid) {
            if (q.dD != nullptr) atomicAdd(q.dD + d0 + lane, a);
            if (q.ddelta_bias != nullptr) atomicAdd(q.ddelta_bias + d0 + lane, e);
        }
    }
}

int validate_scan(const MsScanParams &p);
int scan_positions(const MsScanParams &p);
int ssd_bwd_all_launch(const MsScanBwdParams &q, int n_chunks, hipStream_t stream);
bool ss2d_bwd_fast_ok(const MsScanBwdParams &q);
int ss2d_bwd_launch(const MsScanBwdParams &q, int n_chunks, hipStream_t stream);
int pick_npl(int dstate, int sg);
bool use_cw8(const MsScanParams &p, bool backward);
int pick_mode(bool l_contig, bool d_contig, bool small, int map_h);
bool fits24(int64_t v);
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
    const bool small = fits24(p.seqlen) && fits24(p.u_l_stride) && fits24(p.delta_l_stride) && fits24(q.dout_l_stride) &&
                       fits24(q.du_l_stride) && fits24(q.ddelta_l_stride) && fits24(p.B_l_stride) && fits24(p.C_l_stride) &&
                       fits24(q.dB_l_stride) && fits24(q.dC_l_stride);
    if (p.map_h > 0 && !small) return MS_ERR_STRIDE;
    const bool sa = p.A_dstate_stride == 0 && p.dstate > 1;
    const int bc_dir = ((p.delta_softplus >> 4) & 7) - 1;       // MS_SCAN_BC_MAP
    if (bc_dir >= 0 && (p.map_h <= 0 || !sa || bc_dir > 3)) return MS_ERR_SHAPE;
    switch (pick_mode(lcontig, dcontig, small, p.map_h)) {
        case kModeSS2D:
            if (bc_dir >= 0) hipLaunchKernelGGL((scan_bwd_kernel<NPL, CW, kModeSS2D, true, true>), grid, dim3(64 * kWPB), 0, stream, q, n_chunks, ncb);
            else hipLaunchKernelGGL((scan_bwd_kernel<NPL, CW, kModeSS2D>), grid, dim3(64 * kWPB), 0, stream, q, n_chunks, ncb);
            break;
        case kModeCL:
            if (sa) hipLaunchKernelGGL((scan_bwd_kernel<NPL, CW, kModeCL, true>), grid, dim3(64 * kWPB), 0, stream, q, n_chunks, ncb);
            else    hipLaunchKernelGGL((scan_bwd_kernel<NPL, CW, kModeCL>), grid, dim3(64 * kWPB), 0, stream, q, n_chunks, ncb);
            break;
        default:        hipLaunchKernelGGL((scan_bwd_kernel<NPL, CW, kModeBDL>), grid, dim3(64 * kWPB), 0, stream, q, n_chunks, ncb); break;
    }
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int scan_bwd_dispatch(const MsScanBwdParams &q, hipStream_t stream) {
    const MsScanParams &p = q.f;
    int rc = validate_scan(p);
    if (rc != MS_OK) return rc;
    if (!q.dout || !q.du || !q.ddelta || !q.dA || !q.dB || !q.dC) return MS_ERR_NULL;
    if (p.delta_softplus & MS_SCAN_DT_FUSED) return MS_ERR_UNSUPPORTED;     // forward-only this build
    const int npos = scan_positions(p);
    if (!act_strides_ok(q.dout_d_stride, q.dout_l_stride, npos) || !act_strides_ok(q.du_d_stride, q.du_l_stride, npos) ||
        !act_strides_ok(q.ddelta_d_stride, q.ddelta_l_stride, npos) ||
        !act_strides_ok(q.dB_dstate_stride * 4, q.dB_l_stride, npos) || !act_strides_ok(q.dC_dstate_stride * 4, q.dC_l_stride, npos))
        return MS_ERR_STRIDE;
    if (p.map_h > 0 && (q.dout_d_stride != 1 || q.du_d_stride != 1 || q.ddelta_d_stride != 1 ||
                        q.dB_dstate_stride != 1 || q.dC_dstate_stride != 1)) return MS_ERR_STRIDE;
    if (p.batch == 0 || p.seqlen == 0) return MS_OK;
    const int n_chunks = (p.seqlen + kCL - 1) / kCL;
    if (n_chunks > 1 && !p.x) return MS_ERR_NULL;
    if (((p.delta_softplus >> 4) & 7) - 1 == 4) {
        // MS_SCAN_BC_MAP(4): all four direction slices of the SSD state axis in one launch (scan_bwd_ssd.hip)
        if (p.map_h <= 0 || p.A_dstate_stride != 0 || p.dstate % 4 != 0 || p.dstate / 4 > 16 || p.n_groups % 4 != 0 ||
            (p.delta_softplus & (MS_SCAN_ACCUMULATE | MS_SCAN_LATTICE)) || q.dout_d_stride != 1)
            return MS_ERR_SHAPE;
        if (!fits24(p.seqlen) || !fits24(p.u_l_stride) || !fits24(p.delta_l_stride) || !fits24(q.dout_l_stride) || !fits24(q.du_l_stride) ||
            !fits24(q.ddelta_l_stride) || !fits24(p.B_l_stride) || !fits24(p.C_l_stride) || !fits24(q.dB_l_stride) || !fits24(q.dC_l_stride))
            return MS_ERR_STRIDE;
        return ssd_bwd_all_launch(q, n_chunks, stream);
    }
    // SS2D training shapes: the software-pipelined fast path (scan_ss2d_bwd.hip); MEDSCAN_BWD_FAST=0 keeps the general kernel (A/B runs)
    static const bool fast_on = [] { const char *e = getenv("MEDSCAN_BWD_FAST"); return !(e && e[0] == '0'); }();
    if (fast_on && ss2d_bwd_fast_ok(q)) return ss2d_bwd_launch(q, n_chunks, stream);
    if (use_cw8(p, true)) {
        switch (pick_npl(p.dstate, 8)) {
            case 1: return launch_bwd<1, 8>(q, n_chunks, stream);
            case 2: return launch_bwd<2, 8>(q, n_chunks, stream);
        }
    }
    switch (pick_npl(p.dstate, 4)) {
        case 1: return launch_bwd<1, 16>(q, n_chunks, stream);
        case 2: return launch_bwd<2, 16>(q, n_chunks, stream);     // dstate 5..7 (8..16 took the 8-channel waves above)
    }
    return MS_ERR_DSTATE;       // backward is built for dstate <= 16
}

}  // namespace ms
