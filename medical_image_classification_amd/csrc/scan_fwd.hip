// Selective-scan forward for gfx950 (MI355X).  Replaces selective_scan_fwd_kernel
// (/root/reference/CrossMamba/FusionMamba/selective_scan/selective_scan_fwd_kernel.cuh:67-303).
//
//   delta' = softplus(delta + bias)            (if delta_softplus; threshold 20)
//   a      = exp2(delta' * A[d,n] * log2e)     b = delta' * u * B[b,g,n,l]
//   h_l    = a*h_{l-1} + b                     y_l = sum_n C[b,g,n,l]*h_l + D[d]*u_l
//
// One wave = CW channels x all states (scan_common.h): lane (sg, c) runs NPL states of channel c
// sequentially in registers; the sum over the state axis is a 2-3 step permlane / DPP reduce-scatter per 4
// positions.  No cross-wave traffic; a workgroup = the 2 (CW 16) or 4 (CW 8) waves whose channel blocks share a
// 128-byte line, kept on the same chunk by one barrier.
// Algorithmic traffic per (b,d,l): 12 B (u, delta in; out) + B/C rows (L2-served, shared by the waves of a
// group) + dstate*4/32 B of saved state; no activation is read twice from HBM.
#include "scan_common.h"

namespace ms {

// Waves per workgroup in the forward: the waves of a workgroup own ADJACENT channel blocks, together one 128-byte line
// per position of the channel-last tensors (CW * 4 B * kWF = 128).  They share the chunk's B/C tiles (same batch and
// group) and meet at two barriers per chunk, which also keeps them on the same chunk, so a line is fetched from HBM once
// instead of once per wave whenever the waves drift further apart than the L2 can remember (measured: 3.7x over-fetch
// at stage 0 without).
template <int CW> constexpr int fwd_waves() { return 32 / CW; }

// SA ("scalar A"): every state of a channel shares one decay rate (A passed with A_dstate_stride == 0 -- the SSD /
// Mamba-2 form, CNN_Mamba.py:514): a = exp2(delta' * A) is evaluated once per position instead of once per state.
// BCM ("B/C map", SS2D mode with MS_SCAN_BC_MAP): the B/C rows follow the pixel order of ONE fixed direction for every
// group, the activations their own group's -- a second per-chunk position table.  BCM == 2 (MS_SCAN_BC_MAP(4)): ALL four
// directions in one launch -- the state axis is four slices, slice j's rows follow direction j (four tables): the whole
// concatenated SSD state (CNN_Mamba.py:506-519) in a single pass over u / delta instead of one pass per direction.
template <int NPL, int CW, int MODE, bool SA = false, int BCM = 0>
__global__ void __launch_bounds__(64 * fwd_waves<CW>())
scan_fwd_kernel(const MsScanParams p, const int n_chunks, const int ncb) {
    constexpr int SG = 64 / CW, NP = SG * NPL, kWF = fwd_waves<CW>();
    constexpr int NPLp = npl_padded<NPL>(), NP2 = NPLp / 2;     // states per lane as NP2 packed pairs
    constexpr int RP = SG * NPLp + 4;                           // row pitch (floats) of the [position][state column] B/C tiles
    using Tile = TileIO<MODE, CW>;
    constexpr int kPitch = Tile::kPitch, kTile = Tile::kTile, kCW = CW;
    using Rows = RowIO<MODE, NP, fwd_waves<CW>()>;
    __shared__ __attribute__((aligned(16))) float sB[kCL * RP];       // B / C rows of the chunk: one copy per workgroup
    __shared__ __attribute__((aligned(16))) float sC[kCL * RP];
    __shared__ __attribute__((aligned(8))) v2f sdd_[kWF][kTile];     // {delta', delta' * u} per (position, channel)
    __shared__ float so_[kWF][kTile];        // out tile (the state-group owners write y; the store adds D * u)
    __shared__ float sbias_[kWF][kCW];
    __shared__ float sDv_[kWF][kCW];
    __shared__ int spos_[kWF][2][kCL];      // SS2D mode: pixel positions of the chunk being computed / being prefetched
    __shared__ int sposb_[BCM ? kWF : 1][2][BCM == 2 ? 4 : 1][kCL];   // BCM: the same for the B/C rows' direction(s)
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v2f *sdd = sdd_[wv];
    float *so = so_[wv], *sbias = sbias_[wv], *sDv = sDv_[wv];
    int (*spos)[kCL] = spos_[wv];
    const int c = lane % CW, sg = lane / CW;

    const int N = __builtin_amdgcn_readfirstlane(p.dstate), L = __builtin_amdgcn_readfirstlane(p.seqlen);   // see scan_bwd.hip
    const int dpg = p.dim / p.n_groups;
    // workgroup -> (batch, group, channel block).  Workgroups are dealt round-robin over the 8 XCDs, so the
    // waves that share one (batch, group)'s B/C rows are given equal blockIdx % 8: they hit one XCD's L2
    // instead of making all eight fetch the same rows (speed only, never correctness).
    int pair, cb;
    {
        const int ncg = (ncb + kWF - 1) / kWF;              // workgroups per (batch, group)
        const int npairs = p.batch * p.n_groups, bid = blockIdx.x;
        const int full = (npairs / 8) * 8 * ncg;            // pairs that form complete groups of 8
        int cg;
        if (bid < full) { pair = (bid / (8 * ncg)) * 8 + bid % 8; cg = (bid / 8) % ncg; }
        else            { pair = (npairs / 8) * 8 + (bid - full) / ncg; cg = (bid - full) % ncg; }
        cb = cg * kWF + wv;
    }
    const int g = pair % p.n_groups;
    const int b = pair / p.n_groups;
    // a wave past the last channel block (ncb not a multiple of kWF) computes on zeros and only joins the barriers
    const bool wave_idle = cb >= ncb;
    if (wave_idle) cb = ncb - 1;
    const int nvalid = wave_idle ? 0 : min(kCW, dpg - cb * kCW);
    const int d0 = g * dpg + cb * kCW;
    const bool active = c < nvalid;
    const int d = d0 + (active ? c : max(nvalid, 1) - 1);

    v2f A2[NP2], h[NP2];
#pragma unroll
    for (int i = 0; i < NPLp; ++i) {
        const int n = sg * NPL + i;
        const bool real = i < NPL && (n < N || SA);
        float av = real ? p.A[d * p.A_d_stride + (SA ? 0 : n) * p.A_dstate_stride] : 0.0f;
        if ((p.delta_softplus & MS_SCAN_A_IS_LOG) && real) av = -__expf(av);
        A2[i / 2][i % 2] = av * kLog2e;
        h[i / 2][i % 2] = 0.0f;
    }
    if (lane < kCW) {
        const int dd_ = d0 + min(lane, max(nvalid, 1) - 1);
        sbias[lane] = p.delta_bias ? p.delta_bias[dd_] : 0.0f;
        sDv[lane] = p.D ? p.D[dd_] : 0.0f;
    }
    if (NPLp != NPL) {          // pad columns of the row tiles: never written by the staging, read as B = C = 0
        for (int i = threadIdx.x; i < kCL * RP; i += 64 * kWF) { sB[i] = 0.0f; sC[i] = 0.0f; }
    }

    const int c0w = cb * kCW;                                   // first channel of this wave inside its group
    const float *ub = p.u + b * p.u_batch_stride + g * p.u_group_stride + c0w * p.u_d_stride;
    const float *db = p.delta + b * p.delta_batch_stride + g * p.delta_group_stride + c0w * p.delta_d_stride;
    float *ob = p.out + b * p.out_batch_stride + g * p.out_group_stride + c0w * p.out_d_stride;
    PosMap pm;
    pm.mode = -1; pm.L = L; pm.H = 0; pm.W = 0; pm.invH = 0.0f; pm.tab = nullptr; pm.tab_base = 0;
    if (MODE == kModeSS2D)
        pm.setup(g, __builtin_amdgcn_readfirstlane(p.map_h), __builtin_amdgcn_readfirstlane(p.map_w), L,
                 (p.delta_softplus & MS_SCAN_LATTICE) != 0);
    PosMap pmb = pm;                         // B/C rows: same order as the activations unless BCM
    if (BCM == 1) pmb.mode = ((p.delta_softplus >> 4) & 7) - 1;
    const int nd = BCM == 2 ? N / 4 : N;                    // states per direction slice
    const float *Bb = p.B + b * p.B_batch_stride + g * p.B_group_stride;
    const float *Cb = p.C + b * p.C_batch_stride + g * p.C_group_stride;
    const bool softplus = (p.delta_softplus & MS_SCAN_SOFTPLUS) != 0;
    const bool accumulate = (p.delta_softplus & MS_SCAN_ACCUMULATE) != 0;

    constexpr bool kGen = MODE == kModeBDL, kRowN = MODE == kModeSS2D;      // see scan_bwd.hip: 32-bit stride copies
    const int u_sd = kGen ? (int)p.u_d_stride : 1, u_sl = (int)p.u_l_stride;
    const int dl_sd = kGen ? (int)p.delta_d_stride : 1, dl_sl = (int)p.delta_l_stride;
    const int o_sd = kGen ? (int)p.out_d_stride : 1, o_sl = (int)p.out_l_stride;
    const int B_sn = kRowN ? 1 : (int)p.B_dstate_stride, B_sl = (int)p.B_l_stride;
    const int C_sn = kRowN ? 1 : (int)p.C_dstate_stride, C_sl = (int)p.C_l_stride;
    const Tile tile(lane);
    const Rows rows(threadIdx.x);
    const unsigned sp_mask = softplus ? 0xFFFFFFFFu : 0u;
    float ru[Tile::NE], rd[Tile::NE], rB[Rows::NE], rC[Rows::NE];
    float uk[Tile::NE];                      // the u values this lane staged: out = y + D * u is formed by its store
    auto fetch = [&](int ch) {
        const int l0 = ch * kCL, len = min(kCL, L - l0);
        if (MODE == kModeSS2D) {
            pm.fill_table(spos[ch & 1], l0, lane);
            if (BCM == 1) pmb.fill_table(sposb_[wv][ch & 1][0], l0, lane);
            else if (BCM == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { PosMap pj = pm; pj.mode = j; pj.fill_table(sposb_[wv][ch & 1][j], l0, lane); }
            } else pmb = pm;
            wave_sync();
        }
        tile.fetch(ru, ub, u_sd, u_sl, l0, pm, nvalid, len);
        tile.fetch(rd, db, dl_sd, dl_sl, l0, pm, nvalid, len);
        if constexpr (BCM == 2) {
            rows.fetch_dirs(rB, Bb, B_sl, sposb_[wv][ch & 1], nd, N, len);
            rows.fetch_dirs(rC, Cb, C_sl, sposb_[wv][ch & 1], nd, N, len);
        } else {
            rows.fetch(rB, Bb, B_sn, B_sl, l0, pmb, N, len);
            rows.fetch(rC, Cb, C_sn, C_sl, l0, pmb, N, len);
        }
    };
    fetch(0);
    wave_sync();                                           // sbias / sDv visible

    for (int ch = 0; ch < n_chunks; ++ch) {
        const int l0 = ch * kCL, len = min(kCL, L - l0);
        __syncthreads();                                    // everyone is done with the previous chunk's shared B/C tiles
        // stage {delta', delta' * u}: bias + softplus once per element (a bit-select instead of a branch); elements
        // past the tile edge become the scan identity (a, b) = (1, 0) -- selective_scan_fwd_kernel.cuh:218-222
        const bool pre = (p.delta_softplus & MS_SCAN_DELTA_ACTIVATED) != 0;      // delta' arrives activated (wave-uniform branch)
#pragma unroll
        for (int k = 0; k < Tile::NE; ++k) {
            const bool ok = tile.ok(k, nvalid, len);
            float sp = rd[k];
            if (!pre) {
                const float raw = rd[k] + sbias[tile.ck(k)];
                sp = bits_f((f_bits(softplus_ref(raw)) & sp_mask) | (f_bits(raw) & ~sp_mask));
            }
            const float dl = ok ? sp : 0.0f;
            uk[k] = ok ? ru[k] : 0.0f;
            sdd[tile.soff(k)] = (v2f){dl, dl * uk[k]};
        }
        rows.template put_t<NPL, RP>(sB, rB, N, len);
        rows.template put_t<NPL, RP>(sC, rC, N, len);
        __syncthreads();                                    // the B/C tiles are staged by all waves of the workgroup
        if (ch + 1 < n_chunks) fetch(ch + 1);              // lands while this chunk is computed

#pragma unroll 2
        for (int lb = 0; lb < kCL; lb += 4) {
            float y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v2f dd = sdd[(lb + j) * kPitch + c];
                const float *bp = sB + (lb + j) * RP + sg * NPLp, *cp = sC + (lb + j) * RP + sg * NPLp;
                v2f as_ = splat(0.0f);
                if constexpr (SA) as_ = splat(exp2_fast(dd.x * A2[0].x));
                v2f y2 = splat(0.0f);
#pragma unroll
                for (int q = 0; q < NP2; ++q) {
                    const v2f Bv = *reinterpret_cast<const v2f *>(bp + 2 * q), Cv = *reinterpret_cast<const v2f *>(cp + 2 * q);
                    const v2f a = SA ? as_ : exp2_pk(splat(dd.x) * A2[q]);
                    h[q] = pk_fma(a, h[q], splat(dd.y) * Bv);
                    y2 = q == 0 ? Cv * h[q] : pk_fma(Cv, h[q], y2);
                }
                y[j] = y2.x + y2.y;
            }
            // sum over the state groups; the owner lanes of slot j end up with the result of position lb + j
            const float yt = sum_groups_scatter4<CW>(y, lane);
            if (is_group_owner<CW>(lane)) so[(lb + group_slot<CW>(lane)) * kPitch + c] = yt;
        }
        if (p.x != nullptr && active) {
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                const int n = sg * NPL + i;
                const float hv = h[i / 2][i % 2];
                if (BCM == 2) {         // slice-major checkpoints: slice j is what a one-direction backward launch reads
                    const int j = n / nd;
                    if (n < N) p.x[((((int64_t)j * p.batch + b) * n_chunks + ch) * nd + (n - j * nd)) * p.dim + d] = hv;
                } else if (n < N) p.x[(((int64_t)b * n_chunks + ch) * N + n) * p.dim + d] = hv;
            }
        }
        wave_sync();
        if (MODE == kModeSS2D) { pm.tab = spos[ch & 1]; pm.tab_base = l0; }      // the prefetch moved pm to the next chunk
        {   // out = y + D * u (added to what is there under MS_SCAN_ACCUMULATE: each element is read and written by this thread only)
            char *obb = reinterpret_cast<char *>(ob);
#pragma unroll
            for (int k = 0; k < Tile::NE; ++k)
                if (tile.ok(k, nvalid, len)) {
                    float *o = reinterpret_cast<float *>(obb + tile.goff(k, o_sd, o_sl, l0, pm));
                    const float v = fmaf(sDv[tile.ck(k)], uk[k], so[tile.soff(k)]);
                    *o = accumulate ? *o + v : v;
                }
        }
        wave_sync();
    }
}

// states per lane for a given dstate and number of state groups; 0 = unsupported
int pick_npl(int dstate, int sg) {
    const int need = (dstate + sg - 1) / sg;
    const int cands[6] = {1, 2, 3, 4, 8, 16};
    for (int k = 0; k < 6; ++k) if (cands[k] >= need) return cands[k];
    return 0;
}

// 8-channel waves (8 state groups) when 16-channel waves would leave the chip under-filled.
bool use_cw8(const MsScanParams &p, bool backward) {
    const int dpg = p.dim / p.n_groups;
    const int64_t waves16 = (int64_t)p.batch * p.n_groups * ((dpg + 15) / 16);
    // the backward keeps twice the per-lane state: 8-channel waves are the only way to 2 waves/SIMD there
    return p.dstate >= 8 && p.dstate <= 16 && (backward || waves16 < 2048);
}

// addressing mode from the strides (speed choice for plain sequences; SS2D mode when a map is given)
// `small`: sequence length and every sequence stride < 2^24 (the channel-last kernels multiply them with v_mul_i32_i24)
int pick_mode(bool l_contig, bool d_contig, bool small, int map_h) {
    if (map_h > 0) return kModeSS2D;
    return (d_contig && !l_contig && small) ? kModeCL : kModeBDL;
}
bool fits24(int64_t v) { return v >= 0 && v < (1 << 24); }

template <int NPL, int CW>
static int launch_fwd(const MsScanParams &p, int n_chunks, hipStream_t stream) {
    const int dpg = p.dim / p.n_groups;
    const int ncb = (dpg + CW - 1) / CW;
    constexpr int kWF = fwd_waves<CW>();
    const dim3 grid((unsigned)((int64_t)p.batch * p.n_groups * ((ncb + kWF - 1) / kWF)));
    const bool lcontig = p.u_l_stride == 1 && p.delta_l_stride == 1 && p.out_l_stride == 1;
    const bool dcontig = p.u_d_stride == 1 && p.delta_d_stride == 1 && p.out_d_stride == 1;
    const bool small = fits24(p.seqlen) && fits24(p.u_l_stride) && fits24(p.delta_l_stride) && fits24(p.out_l_stride) &&
                       fits24(p.B_l_stride) && fits24(p.C_l_stride);
    if (p.map_h > 0 && !small) return MS_ERR_STRIDE;
    const bool sa = p.A_dstate_stride == 0 && p.dstate > 1;     // scalar decay per channel (SSD form): channel-last variant only
    const int bc_dir = ((p.delta_softplus >> 4) & 7) - 1;       // MS_SCAN_BC_MAP
    if (bc_dir >= 0 && (p.map_h <= 0 || !sa || bc_dir > 4)) return MS_ERR_SHAPE;
    if (bc_dir == 4 && (p.dstate % 4 != 0 || p.dstate / 4 > 16 || (p.delta_softplus & MS_SCAN_ACCUMULATE))) return MS_ERR_SHAPE;
    switch (pick_mode(lcontig, dcontig, small, p.map_h)) {
        case kModeSS2D:
            if (bc_dir == 4) hipLaunchKernelGGL((scan_fwd_kernel<NPL, CW, kModeSS2D, true, 2>), grid, dim3(64 * kWF), 0, stream, p, n_chunks, ncb);
            else if (bc_dir >= 0) hipLaunchKernelGGL((scan_fwd_kernel<NPL, CW, kModeSS2D, true, 1>), grid, dim3(64 * kWF), 0, stream, p, n_chunks, ncb);
            else hipLaunchKernelGGL((scan_fwd_kernel<NPL, CW, kModeSS2D>), grid, dim3(64 * kWF), 0, stream, p, n_chunks, ncb);
            break;
        case kModeCL:
            if (sa) hipLaunchKernelGGL((scan_fwd_kernel<NPL, CW, kModeCL, true>), grid, dim3(64 * kWF), 0, stream, p, n_chunks, ncb);
            else    hipLaunchKernelGGL((scan_fwd_kernel<NPL, CW, kModeCL>), grid, dim3(64 * kWF), 0, stream, p, n_chunks, ncb);
            break;
        default:        hipLaunchKernelGGL((scan_fwd_kernel<NPL, CW, kModeBDL>), grid, dim3(64 * kWF), 0, stream, p, n_chunks, ncb); break;
    }
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

static bool fits_u32(int64_t a) { return a >= 0 && a < ((int64_t)1 << 32); }

// an activation tensor is addressed with 32-bit byte offsets relative to its (batch, group, first channel) base
bool act_strides_ok(int64_t sd, int64_t sl, int seqlen) {
    if (sd < 0 || sl < 0) return false;
    return fits_u32((sd * 16 + sl * (int64_t)(seqlen > 0 ? seqlen : 1)) * 4);
}

// positions a tensor of the problem is indexed with: the sequence length, or every pixel of the map in SS2D mode (the stride-2
// sub-lattices visit a quarter of them each)
int scan_positions(const MsScanParams &p) { return p.map_h > 0 ? p.map_h * p.map_w : p.seqlen; }

int validate_scan(const MsScanParams &p) {
    const bool dtf = (p.delta_softplus & MS_SCAN_DT_FUSED) != 0;
    if (!p.u || (!p.delta && !dtf) || !p.A || !p.B || !p.C) return MS_ERR_NULL;
    if (dtf && (!p.dt_x || !p.dt_w || p.map_h <= 0 || p.dt_rank < 1)) return MS_ERR_NULL;
    if (p.batch < 0 || p.dim <= 0 || p.seqlen < 0 || p.dstate <= 0 || p.n_groups <= 0) return MS_ERR_SHAPE;
    if (p.dim % p.n_groups != 0) return MS_ERR_SHAPE;
    if (p.dstate > 256) return MS_ERR_DSTATE;
    if (p.map_h < 0 || p.map_w < 0 || (p.map_h > 0) != (p.map_w > 0)) return MS_ERR_SHAPE;
    const bool lattice = (p.delta_softplus & MS_SCAN_LATTICE) != 0;
    if (lattice && (p.map_h <= 0 || p.map_h % 2 != 0 || p.map_w % 2 != 0 || ((p.delta_softplus >> 4) & 7) != 0 || dtf)) return MS_ERR_SHAPE;
    if (p.map_h > 0) {
        const int64_t want = lattice ? (int64_t)(p.map_h / 2) * (p.map_w / 2) : (int64_t)p.map_h * p.map_w;
        if (want != p.seqlen || p.n_groups % 4 != 0 || (int64_t)p.map_h * p.map_w >= (1 << 22)) return MS_ERR_SHAPE;
        // SS2D mode needs channel-last activations and projection rows that are contiguous along the state axis
        if (p.u_d_stride != 1 || (!dtf && p.delta_d_stride != 1) || p.B_dstate_stride != 1 || p.C_dstate_stride != 1) return MS_ERR_STRIDE;
    }
    const int npos = scan_positions(p);
    if (!act_strides_ok(p.u_d_stride, p.u_l_stride, npos) || (!dtf && !act_strides_ok(p.delta_d_stride, p.delta_l_stride, npos)) ||
        !act_strides_ok(p.B_dstate_stride * 4, p.B_l_stride, npos) || !act_strides_ok(p.C_dstate_stride * 4, p.C_l_stride, npos))
        return MS_ERR_STRIDE;
    return MS_OK;
}

bool ss2d_fast_ok(const MsScanParams &p);
int ss2d_fwd_launch(const MsScanParams &p, int n_chunks, hipStream_t stream);

int scan_fwd_dispatch(const MsScanParams &p, hipStream_t stream) {
    int rc = validate_scan(p);
    if (rc != MS_OK) return rc;
    if (!p.out) return MS_ERR_NULL;
    if (!act_strides_ok(p.out_d_stride, p.out_l_stride, scan_positions(p))) return MS_ERR_STRIDE;
    if (p.map_h > 0 && p.out_d_stride != 1) return MS_ERR_STRIDE;
    if (p.batch == 0 || p.seqlen == 0) return MS_OK;
    const int n_chunks = (p.seqlen + kCL - 1) / kCL;
    if (ss2d_fast_ok(p)) return ss2d_fwd_launch(p, n_chunks, stream);
    if (p.segments >= 2) return MS_ERR_UNSUPPORTED;                         // segments (p.x = workspace): the SS2D fast path only
    if (p.delta_softplus & MS_SCAN_DT_FUSED) return MS_ERR_UNSUPPORTED;     // the fused projection exists on the fast path only
    if (use_cw8(p, false)) {
        switch (pick_npl(p.dstate, 8)) {
            case 1: return launch_fwd<1, 8>(p, n_chunks, stream);
            case 2: return launch_fwd<2, 8>(p, n_chunks, stream);
        }
    }
    // all-direction SSD launch (64 states = 4 x 16): 8-channel waves x 8 states per lane while 16-channel waves would
    // leave the chip under-filled (same rule as use_cw8 applies to 16 states)
    if ((((p.delta_softplus >> 4) & 7) - 1) == 4 && p.dstate == 64 &&
        (int64_t)p.batch * p.n_groups * ((p.dim / p.n_groups + 15) / 16) < 2048)
        return launch_fwd<8, 8>(p, n_chunks, stream);
    switch (pick_npl(p.dstate, 4)) {
        case 1: return launch_fwd<1, 16>(p, n_chunks, stream);
        case 2: return launch_fwd<2, 16>(p, n_chunks, stream);
        case 3: return launch_fwd<3, 16>(p, n_chunks, stream);
        case 4: return launch_fwd<4, 16>(p, n_chunks, stream);
        case 8: return launch_fwd<8, 16>(p, n_chunks, stream);
        case 16: return launch_fwd<16, 16>(p, n_chunks, stream);
    }
    return MS_ERR_DSTATE;       // dstate > 64: not tiled by this build
}

}  // namespace ms
