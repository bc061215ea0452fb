// SS2D fast path of the selective-scan BACKWARD for gfx950 (MI355X): what MedMamba's SS2D block runs in training
// (/root/reference/MedMamba.py:386-424 through autograd; the reference kernel is selective_scan_bwd_kernel.cuh:75-489 with
// reverse_scan.cuh).  Same algebra and the same packed sweeps as the general kernel's fast branch (scan_bwd.hip, `kPk`); what is
// different is everything AROUND the sweeps, which the ablations of round 3 showed to be the bound (the sweeps issue
// at ~5.5 cycles per instruction, but stores, the dB/dC flush and two barriers per chunk -- 13 % of the instructions -- took 30 % of the
// time, serialised behind them):
//   * SOFTWARE-PIPELINED chunk loop: the du / ddelta stores and the dB / dC flush (4-wave combine + atomics) of chunk c+1 are issued
//     from INSIDE the sweeps of chunk c, in slices between the 4-position batches -- their LDS and memory latencies run under the
//     sweeps' VALU work instead of in phases of their own; out tiles, dB/dC tiles and the shared B/C tiles are double-buffered by
//     chunk parity, so ONE workgroup barrier per chunk orders everything (was two);
//   * 16-byte vector I/O: a lane owns 4 consecutive channels of one position -- one global_load_dwordx4 per tensor and chunk, one
//     store per output tensor (was four 4-byte accesses each, with a position-table lookup and a multiply apiece);
//   * the pixel positions of a chunk are computed once per workgroup (32 threads, LDS table, 4-deep ring) instead of per wave.
// Specialisation: SS2D addressing (channel-last activations indexed by pixel, projection rows [dts | B | C] contiguous along the
// state axis), d_state == 16, dense real A, 8-channel waves x 2 states per lane, no MS_SCAN_ACCUMULATE / BC_MAP / LATTICE.
// Everything else stays on scan_bwd.hip.
#include <cstdlib>
#include <type_traits>
#include "scan_common.h"

#ifndef MS_BWD_FULLPATH
#define MS_BWD_FULLPATH 0
#endif
#ifndef MS_BWD_W3_DEFAULT
#define MS_BWD_W3_DEFAULT false
#endif

namespace ms {
namespace {

constexpr int kNs = 16;                 // d_state
constexpr int kCWb = 8, kNWb = 4, kNTb = 64 * kNWb, kNBb = kCL / 4;
constexpr int kSGb = kNs / 2;           // state pairs per position (the sweeps' lanes hold two states each)
constexpr int kRowPb = kCL + 4;         // dB / dC tile: [state][position] row pitch
constexpr int kDCb = kNs * kRowPb + 2;  // dC tile 2 banks past the dB tile (the combine reads both in one instruction)
constexpr int kWSb = kDCb + kNs * kRowPb;

__device__ __forceinline__ float4 ld4b(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4b(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ const float *atb(const float *base, int off) {
    return reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (uint32_t)off * 4u);
}
__device__ __forceinline__ float *atb(float *base, int off) {
    return reinterpret_cast<float *>(reinterpret_cast<char *>(base) + (uint32_t)off * 4u);
}

// PRE: delta already holds delta' = softplus(raw + bias) (MS_SCAN_DELTA_ACTIVATED, the training path)
// W3: the THREE-waves-per-SIMD build (168 VGPRs, 53.8 KB of LDS per workgroup -> three workgroups per CU): the decays of a chunk are
//     not kept in registers (64 VGPRs) but evaluated again by the reverse sweep (+64 v_exp, +32 v_pk_mul per chunk, ~9 % more issue
//     slots), the out tiles and the dB / dC tiles are single-buffered -- every slice of the previous chunk rides in the FORWARD sweep
//     (which writes neither) and a second barrier separates it from the reverse sweep.  What it buys: a third wave per SIMD to cover
//     the LDS / exp / DPP latencies, and MedMamba-T's stage 0 (768 workgroups) resident in ONE round instead of 1.5.
template <bool PRE, bool W3>
__global__ void __launch_bounds__(kNTb) __attribute__((amdgpu_waves_per_eu(W3 ? 3 : 2, W3 ? 3 : 2)))
ss2d_bwd_kernel(const MsScanBwdParams q, const int n_chunks) {
    constexpr int CW = kCWb, NB = kNBb, NPAR = W3 ? 1 : 2;
    const MsScanParams &p = q.f;
    // The sweeps' LDS operands are laid out so that a lane needs ONE 16-byte read per array and position: {delta', u, dout, delta' u} of its
    // channel, and {B, C} of its state pair.  (Until round 3 they were four 8-byte arrays; the compiler paired the reads of neighbouring
    // positions into ds_read2_b64, which occupies the LDS for 8 cycles against 4 for a ds_read_b128 of the same 16 bytes -- the kernel's LDS
    // pipe was 49 % busy; measured with the reads re-pointed, values wrong: 4.67 -> 4.46 ms per step.)
    __shared__ __attribute__((aligned(16))) float4 sBCq[2][kCL * kSGb];            // [chunk parity][position][state pair] = {B0, B1, C0, C1}
    __shared__ __attribute__((aligned(16))) float4 sP_[kNWb][kCL * CW];            // [position][channel] = {delta', u, dout, delta' * u}
    __shared__ __attribute__((aligned(16))) float sOut_[NPAR][kNWb][2][kCL * CW];  // [parity][wave][du | ddelta'][position][channel]
    __shared__ float sdBC_[NPAR][kNWb][kWSb];                                       // [parity][wave] dB | dC of the wave's channels
    __shared__ int stab[4][kCL];                                                    // pixel positions of 4 chunks (ring)
    const int lane = threadIdx.x & 63, tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 *sP = sP_[wv];
    // sweep domain: lane = (state group sg, channel c); I/O domain: lane = (position pl, channel quad)
    const int c = lane & 7, sg = lane >> 3;
    const int pl = lane >> 1, q4 = (lane & 1) * 4;

    const int L = __builtin_amdgcn_readfirstlane(p.seqlen);
    const int dpg = p.dim / p.n_groups, ncg = (dpg + 31) / 32;
    int pair, cg;
    {
        const int npairs = p.batch * p.n_groups, bid = blockIdx.x;
        const int full = (npairs / 8) * 8 * ncg;            // equal blockIdx % 8 (one XCD's L2) for the workgroups of one (batch, group)
        if (bid < full) { pair = (bid / (8 * ncg)) * 8 + bid % 8; cg = (bid / 8) % ncg; }
        else            { pair = (npairs / 8) * 8 + (bid - full) / ncg; cg = (bid - full) % ncg; }
    }
    const int g = pair % p.n_groups, b = pair / p.n_groups;
    const int c0w = cg * 32 + wv * CW;                          // this wave's first channel inside its group
    const int nvalid = max(0, min(CW, dpg - c0w));              // dpg % 4 == 0 (host): quads are valid or invalid as a whole
    const int d0 = g * dpg + c0w;
    const bool active = c < nvalid, quad_ok = q4 < nvalid;
    const int dsw = min(d0 + (active ? c : 0), p.dim - 1);      // channel whose parameters this lane reads in the sweeps

    v2f A2p, Anp;
    {
        float a0 = active ? p.A[dsw * p.A_d_stride + (sg * 2) * p.A_dstate_stride] : 0.0f;
        float a1 = active ? p.A[dsw * p.A_d_stride + (sg * 2 + 1) * p.A_dstate_stride] : 0.0f;
        if ((p.delta_softplus & MS_SCAN_A_IS_LOG) && active) { a0 = -__expf(a0); a1 = -__expf(a1); }
        Anp = (v2f){a0, a1}; A2p = Anp * kLog2e;
    }
    const float Dv = (p.D != nullptr && sg == 0 && active) ? p.D[dsw] : 0.0f;       // D*g enters du once per channel, through group 0
    // (D * dout added by the du store instead -- 4 FMAs per lane and chunk against 32 multiplies -- measured: no gain, +13 VGPRs)
    float bias4[PRE ? 1 : 4];
    if constexpr (!PRE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bias4[j] = (p.delta_bias && quad_ok) ? p.delta_bias[min(d0 + q4 + j, p.dim - 1)] : 0.0f;
    }
    const unsigned sp_mask = (p.delta_softplus & MS_SCAN_SOFTPLUS) ? 0xFFFFFFFFu : 0u;

    const int c0s = nvalid > 0 ? c0w : 0;                       // a wave past the last channel block loads the group's first channels (in bounds) and stores nothing
    const float *ub = p.u + b * p.u_batch_stride + g * p.u_group_stride + c0s;
    const float *db = p.delta + b * p.delta_batch_stride + g * p.delta_group_stride + c0s;
    const float *gb = q.dout + b * q.dout_batch_stride + g * q.dout_group_stride + c0s;
    float *dub = q.du + b * q.du_batch_stride + g * q.du_group_stride + c0s;
    float *ddb = q.ddelta + b * q.ddelta_batch_stride + g * q.ddelta_group_stride + c0s;
    const float *Bb = p.B + b * p.B_batch_stride + g * p.B_group_stride;
    const float *Cb = p.C + b * p.C_batch_stride + g * p.C_group_stride;
    const int u_sl = (int)p.u_l_stride, dl_sl = (int)p.delta_l_stride, g_sl = (int)q.dout_l_stride;
    const int du_sl = (int)q.du_l_stride, dd_sl = (int)q.ddelta_l_stride;
    const int B_sl = (int)p.B_l_stride, C_sl = (int)p.C_l_stride;
    PosMap pm;
    pm.mode = g & 3; pm.L = L;
    pm.H = __builtin_amdgcn_readfirstlane(p.map_h); pm.W = __builtin_amdgcn_readfirstlane(p.map_w);
    pm.invH = 1.0f / (float)p.map_h; pm.tab = nullptr; pm.tab_base = 0;

    // flush geometry: thread -> (tensor, state) = tid % 32 fixed, positions tid / 32 + 8 i: a wave's atomics cover two whole
    // projection-row segments [dB(16) | dC(16)]
    const int ft = tid & 31, flb0 = tid >> 5, ftc = ft >> 4, fn = ft & 15;
    float *fbase = (ftc ? q.dC + b * q.dC_batch_stride + g * q.dC_group_stride : q.dB + b * q.dB_batch_stride + g * q.dB_group_stride) + fn;
    const int fsl = ftc ? (int)q.dC_l_stride : (int)q.dB_l_stride;
    const int fsrc = (ftc ? kDCb : 0) + fn * kRowPb + flb0;
    // register (DPP) channel sums: lane (sg, c) ends up with value c of its group's 8 = position lb + c / 2, state sg * 2 + c % 2
    const int t_dpp = (sg * 2 + (c & 1)) * kRowPb + (c >> 1);

    // saved states x[b, chunk, n, d]
    const float *xs0[2];
    const int64_t x_chunk_stride = n_chunks > 1 ? (int64_t)kNs * p.dim : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) xs0[i] = n_chunks > 1 ? p.x + ((int64_t)b * n_chunks * kNs + sg * 2 + i) * p.dim + dsw : p.A;

    float4 ru, rd, rg, rBC;
    float rx0, rx1;
    int pos_next = 0, pos_cur = 0, pos_prev = 0;
    const int bc_is_c = tid >> 7, bc_r = tid & 127;            // B/C staging: thread -> (tensor, position bc_r / 4, quarter bc_r % 4)
    auto prefetch = [&](int ch) {
        const int *tab = stab[ch & 3];
        const int pos = tab[pl];
        const int cq = quad_ok ? q4 : 0;
        ru = ld4b(atb(ub, __mul24(pos, u_sl) + cq));
        rd = ld4b(atb(db, __mul24(pos, dl_sl) + cq));
        rg = ld4b(atb(gb, __mul24(pos, g_sl) + cq));
        rBC = ld4b(atb(bc_is_c ? Cb : Bb, __mul24(tab[bc_r >> 2], bc_is_c ? C_sl : B_sl) + 4 * (bc_r & 3)));
        rx0 = xs0[0][(int64_t)max(ch - 1, 0) * x_chunk_stride];
        rx1 = xs0[1][(int64_t)max(ch - 1, 0) * x_chunk_stride];
        pos_next = pos;
    };
    auto fill_tab = [&](int ch) {      // positions past L are clamped to L - 1: every address formed from the table is valid
        if (tid < kCL) stab[ch & 3][tid] = pm(min(ch * kCL + tid, L - 1));
    };
    fill_tab(n_chunks - 1);
    __syncthreads();
    prefetch(n_chunks - 1);

    v2f dhp = {0.0f, 0.0f}, dAp = {0.0f, 0.0f};
    float dDk[4] = {0.f, 0.f, 0.f, 0.f}, dbk[4] = {0.f, 0.f, 0.f, 0.f};
    float4 sig_prev = make_float4(1.f, 1.f, 1.f, 1.f);
    int len_prev = 0;

    // the work of the PREVIOUS chunk (out / dB|dC tiles of parity pp, positions tabp) that rides inside the sweeps of the current one
    auto store_du = [&](int pp, auto fullp) {
        const float4 v = ld4b(sOut_[pp][wv][0] + pl * CW + q4);
#ifdef MS_ABL_NOGST
        if (pl < len_prev && quad_ok && v.x == 123.456f) st4b(atb(dub, __mul24(pos_prev, du_sl) + q4), v);
#else
        if ((decltype(fullp)::value || pl < len_prev) && quad_ok) st4b(atb(dub, __mul24(pos_prev, du_sl) + q4), v);
#endif
    };
    auto store_dd = [&](int pp, auto fullp) {
        float4 v = ld4b(sOut_[pp][wv][1] + pl * CW + q4);
        v.x *= sig_prev.x; v.y *= sig_prev.y; v.z *= sig_prev.z; v.w *= sig_prev.w;       // d delta = d delta' * softplus'
#ifdef MS_ABL_NOGST
        if (pl < len_prev && quad_ok && v.x == 123.456f) {
#else
        if ((decltype(fullp)::value || pl < len_prev) && quad_ok) {
#endif
            st4b(atb(ddb, __mul24(pos_prev, dd_sl) + q4), v);
            dbk[0] += v.x; dbk[1] += v.y; dbk[2] += v.z; dbk[3] += v.w;
        }
    };
    auto flush_piece = [&](int pp, const int *tabp, int i, auto fullp) {
        const float *src = sdBC_[pp][0] + fsrc + 8 * i;
        const float v = (src[0] + src[kWSb]) + (src[2 * kWSb] + src[3 * kWSb]);
        const int l = flb0 + 8 * i;
#ifdef MS_ABL_NOATOM
        if (l < len_prev && v == 123.456f) atomicAdd(fbase + __mul24(tabp[l], fsl), v);
#else
        if (decltype(fullp)::value || l < len_prev) atomicAdd(fbase + __mul24(tabp[l], fsl), v);
#endif
    };

    for (int ch = n_chunks - 1; ch >= 0; --ch) {
        const int par = ch & 1, len = min(kCL, L - ch * kCL);
        const bool have_prev = ch + 1 < n_chunks;
        const float4 *sBCl = sBCq[par] + sg;                          // this lane's state pair inside a position row
        const int po = W3 ? 0 : par, pp = W3 ? 0 : par ^ 1;        // tile parity of this chunk's results / of the previous chunk's
        float *su = sOut_[po][wv][0], *sgd = sOut_[po][wv][1];
        float *sdB = sdBC_[po][wv], *sdC = sdBC_[po][wv] + kDCb;
        // ---------------- stage chunk `ch` (prefetched one iteration ago) ----------------
        float4 sig_cur;
        {
            const bool ok = pl < len && quad_ok;
            float dl[4], uu[4], gg[4];
            const float rdv[4] = {rd.x, rd.y, rd.z, rd.w}, ruv[4] = {ru.x, ru.y, ru.z, ru.w}, rgv[4] = {rg.x, rg.y, rg.z, rg.w};
            float sg4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float sp;
                if constexpr (PRE) sp = rdv[j];
                else {
                    const float raw = rdv[j] + bias4[j];
                    sp = bits_f((f_bits(softplus_ref(raw)) & sp_mask) | (f_bits(raw) & ~sp_mask));
                }
                dl[j] = ok ? sp : 0.0f; uu[j] = ok ? ruv[j] : 0.0f; gg[j] = ok ? rgv[j] : 0.0f;
                sg4[j] = bits_f((f_bits(sigmoid_from_softplus(dl[j])) & sp_mask) | (f_bits(1.0f) & ~sp_mask));
                dDk[j] = fmaf(gg[j], uu[j], dDk[j]);
            }
            sig_cur = make_float4(sg4[0], sg4[1], sg4[2], sg4[3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) sP[pl * CW + q4 + j] = make_float4(dl[j], uu[j], gg[j], dl[j] * uu[j]);
            const float4 bc = (bc_r >> 2) < len ? rBC : make_float4(0.f, 0.f, 0.f, 0.f);
            // this thread's four states are two state pairs: the B thread fills the first half of their entries, the C thread the second
            float *bq = reinterpret_cast<float *>(sBCq[par] + (bc_r >> 2) * kSGb + 2 * (bc_r & 3)) + 2 * bc_is_c;
            *reinterpret_cast<v2f *>(bq) = (v2f){bc.x, bc.y};
            *reinterpret_cast<v2f *>(bq + 4) = (v2f){bc.z, bc.w};
        }
        v2f hp = ch > 0 ? (v2f){rx0, rx1} : (v2f){0.0f, 0.0f};
        pos_cur = pos_next;
        if (ch > 0) fill_tab(ch - 1);
        // ONE barrier per chunk: (i) this chunk's tiles are staged, (ii) every wave has finished the sweeps of chunk ch + 1, whose
        // dB / dC tiles (other parity) are combined below, (iii) the position table of chunk ch - 1 is written
#ifdef MS_ABL_NOBAR
        wave_sync();          // timing only: the workgroup's waves are NOT kept together (results are wrong)
#else
        __syncthreads();
#endif
#ifdef MS_ABL_NOGLD
        if (ch > 0 && ch == n_chunks + 5) prefetch(ch - 1);
#else
        if (ch > 0) prefetch(ch - 1);                      // lands while this chunk is computed
#endif
        const int *tabp = stab[(ch + 1) & 3];

        // ---------------- packed sweeps (same algebra as scan_bwd.hip) ----------------
        // 4-position batches past the end of the sequence are skipped whole (wave-uniform branch): they hold the scan identity
        // (delta' = 0 -> a = 1, b = 0, dout = 0), so h, dh and every sum pass through them unchanged -- bit-identical, and L = 49 / 196
        // (MedMamba-T stages 3 / 2) do not pay for 15 / 28 padded positions.  (One unpredicated basic block for the steady state was
        // measured as well, MS_BWD_FASTBLOCK in the history of this file: 3-4 % SLOWER -- the giant scheduling region raises the
        // register pressure to 256 with spills; the per-batch regions the guards create schedule better.)
        // FULL: this chunk and the previous one are whole (the steady state of a long sequence): the batch guards, the `have_prev` tests
        // and the length predicates of the stores / flush compile out -- 60 scalar / branch instructions per chunk, which cost this
        // kernel nearly as much as vector ones (measured: +256 s_mov per chunk = +7.8 %, +256 dependent v_add = +8.4 %).  The batches
        // stay separate scheduling regions (sched_barrier): one region per chunk needs more than 256 registers.
        auto sweeps = [&](auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        v2f ap[W3 ? 1 : kCL], ckp[NB];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            if constexpr (FULL && MS_BWD_FULLPATH == 1) __builtin_amdgcn_sched_barrier(0);
            if ((FULL && MS_BWD_FULLPATH == 1) || kb * 4 < len) {       // FULLPATH 2: the batch guards stay (same scheduling regions as the generic path)
#pragma unroll
                for (int l = kb * 4; l < kb * 4 + 4; ++l) {
                    if ((l & 3) == 0) ckp[l >> 2] = hp;
#ifdef MS_ABL_NOLDS
                    const float4 pq = sP[c], bcq = sBCl[0];
#else
                    const float4 pq = sP[l * CW + c], bcq = sBCl[l * kSGb];
#endif
                    const v2f p1 = {pq.x, pq.y}, p2 = {pq.z, pq.w}, Bp = {bcq.x, bcq.y};     // {delta', u}, {dout, delta' u}, B pair
#ifdef MS_ABL_NOEXP
                    const v2f a = pk_fma((v2f){p1.x, p1.x}, A2p, (v2f){1.0f, 1.0f});
#else
                    const v2f a = exp2_pk((v2f){p1.x, p1.x} * A2p);
#endif
                    if constexpr (!W3) ap[l] = a;
                    hp = pk_fma(a, hp, (v2f){p2.y, p2.y} * Bp);
                }
            }
            if (FULL || have_prev) {                               // wave-uniform
                if (kb == 1) store_du(pp, full_c);
                if (kb == 3) store_dd(pp, full_c);
                if (W3) {
                    if (kb >= 4) flush_piece(pp, tabp, kb - 4, full_c);
                } else {
                    if (kb == 5) flush_piece(pp, tabp, 0, full_c);
                    if (kb == 7) flush_piece(pp, tabp, 1, full_c);
                }
            }
        }
        if constexpr (W3) __syncthreads();     // every wave has read the previous chunk's dB / dC tiles: the reverse sweep may overwrite them
#pragma unroll
        for (int kb = NB - 1; kb >= 0; --kb) {
            const int lb = kb * 4;
            if constexpr (FULL && MS_BWD_FULLPATH == 1) __builtin_amdgcn_sched_barrier(0);
            if ((FULL && MS_BWD_FULLPATH == 1) || lb < len) {
                v2f Bp[4], Cp[4], bu[4], hv[4], p1[4], p2[4], aj[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#ifdef MS_ABL_NOLDS
                    const float4 pq = sP[c], bcq = sBCl[0];
#else
                    const float4 pq = sP[(lb + j) * CW + c], bcq = sBCl[(lb + j) * kSGb];
#endif
                    p1[j] = (v2f){pq.x, pq.y}; p2[j] = (v2f){pq.z, pq.w}; Bp[j] = (v2f){bcq.x, bcq.y}; Cp[j] = (v2f){bcq.z, bcq.w};
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (W3) aj[j] = exp2_pk((v2f){p1[j].x, p1[j].x} * A2p); else aj[j] = ap[lb + j];
                    bu[j] = (v2f){p2[j].y, p2[j].y} * Bp[j];
                    hv[j] = pk_fma(aj[j], j > 0 ? hv[j > 0 ? j - 1 : 0] : ckp[kb], bu[j]);
                }
                float duv[4], ddv[4], vB8[8], vC8[8];
#ifdef MS_ABL_ADDSNOP
#pragma unroll
                for (int z = 0; z < 32; ++z) asm volatile("s_nop 0");
#endif
#ifdef MS_ABL_ADDSALU
#pragma unroll
                for (int z = 0; z < 32; ++z) asm volatile("s_mov_b32 s90, 0" ::: "s90");
#endif
#ifdef MS_ABL_ADDVALU
                { float zz = 0.f;
#pragma unroll
                for (int z = 0; z < 32; ++z) asm volatile("v_add_f32 %0, %0, %0" : "+v"(zz)); }
#endif
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    const v2f gg = {p2[j].x, p2[j].x};
                    const v2f dhn = pk_fma(Cp[j], gg, dhp);
                    const v2f t1 = dhn * Bp[j];
                    dhp = aj[j] * dhn;
                    // a_j h_{j-1} dh_j as (a_j dh_j) h_{j-1}: the first factor is the carry that is formed anyway, h_{j-1} is at hand
                    // (was dh_j * (h_j - b_j u_j): one more packed instruction per position)
                    const v2f qv = dhp * (j > 0 ? hv[j > 0 ? j - 1 : 0] : ckp[kb]);
                    const v2f t2 = qv * Anp;
                    dAp = pk_fma(qv, (v2f){p1[j].x, p1[j].x}, dAp);
                    {   // the dB / dC terms: packed multiplies, the scalar operand taken from one half of the staged pair by op_sel (written
                        // as assembly: left to the compiler the splat is built with two v_mov per product -- 96 per chunk -- and as four
                        // scalar multiplies it is two instructions more per position)
                        v2f vb, vc;
                        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(vb) : "v"(dhn), "v"(p2[j]));          // dh * (delta' u)
                        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(vc) : "v"(hv[j]), "v"(p2[j]));     // h * dout
                        vB8[2 * j] = vb.x; vB8[2 * j + 1] = vb.y; vC8[2 * j] = vc.x; vC8[2 * j + 1] = vc.y;
                    }
                    const float s1 = t1.x + t1.y, s2 = t2.x + t2.y;
                    // {du, d delta'} = s1 * {delta', u} + {D dout, s2}: one packed FMA on the staged pair
                    const v2f o = pk_fma((v2f){s1, s1}, p1[j], (v2f){Dv * p2[j].x, s2});
                    duv[j] = o.x; ddv[j] = o.y;
                }
#ifdef MS_ABL_NOGROUP
                const float du_t = (duv[0] + duv[1]) + (duv[2] + duv[3]), dd_t = (ddv[0] + ddv[1]) + (ddv[2] + ddv[3]);
#else
                const float du_t = sum_groups_scatter4<CW>(duv, lane);
                const float dd_t = sum_groups_scatter4<CW>(ddv, lane);
#endif
                {   // both lanes of the last butterfly pair (lane bit 3) hold the total and store it to the same word: no predicate, and
                    // the last DPP add stays in this block where it folds into one v_add_f32_dpp (behind `if (owner)` it was sunk into the
                    // branch: v_mov_b32_dpp + v_add + an exec save / restore per batch)
                    const int lo = lb + group_slot<CW>(lane);
                    su[lo * CW + c] = du_t;
                    sgd[lo * CW + c] = dd_t;
                }
#ifdef MS_ABL_NOCHAN
                sdB[t_dpp + lb] = ((vB8[0] + vB8[1]) + (vB8[2] + vB8[3])) + ((vB8[4] + vB8[5]) + (vB8[6] + vB8[7]));
                sdC[t_dpp + lb] = ((vC8[0] + vC8[1]) + (vC8[2] + vC8[3])) + ((vC8[4] + vC8[5]) + (vC8[6] + vC8[7]));
#else
                sdB[t_dpp + lb] = chan_scatter8(vB8, lane);
                sdC[t_dpp + lb] = chan_scatter8(vC8, lane);
#endif
            }
            if (!W3 && (FULL || have_prev)) {
                if (kb == 5) flush_piece(pp, tabp, 2, full_c);
                if (kb == 2) flush_piece(pp, tabp, 3, full_c);
            }
        }
        };
        if (MS_BWD_FULLPATH && len == kCL && have_prev && len_prev == kCL) sweeps(std::true_type{});
        else sweeps(std::false_type{});
        sig_prev = sig_cur; pos_prev = pos_cur; len_prev = len;
    }
    // epilogue: chunk 0's stores and flush
    __syncthreads();
    {
        const int *tabp = stab[0];
        store_du(0, std::false_type{}); store_dd(0, std::false_type{});
#pragma unroll
        for (int i = 0; i < 4; ++i) flush_piece(0, tabp, i, std::false_type{});
    }

    if (active) {
        const bool alog = (p.delta_softplus & MS_SCAN_A_IS_LOG) != 0;         // A = -exp(A_log)  =>  dL/dA_log = dL/dA * A
        atomicAdd(q.dA + (int64_t)dsw * kNs + sg * 2, alog ? dAp.x * Anp.x : dAp.x);
        atomicAdd(q.dA + (int64_t)dsw * kNs + sg * 2 + 1, alog ? dAp.y * Anp.y : dAp.y);
    }
    // dD / ddelta_bias: a lane holds the partial sums of its channel quad over the positions it staged / stored
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a = dDk[j], e = dbk[j];
#pragma unroll
        for (int m = 2; m < 64; m *= 2) { a += __shfl_xor(a, m); e += __shfl_xor(e, m); }
        if (lane < 2 && quad_ok) {
            if (q.dD != nullptr) atomicAdd(q.dD + d0 + q4 + j, a);
            if (q.ddelta_bias != nullptr) atomicAdd(q.ddelta_bias + d0 + q4 + j, e);
        }
    }
}

bool aligned16b(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

bool ss2d_fast_ok(const MsScanParams &p);
bool fits24(int64_t v);

// Can the backward fast path take this problem?  (anything else runs on the general kernels of scan_bwd.hip)
bool ss2d_bwd_fast_ok(const MsScanBwdParams &q) {
    const MsScanParams &p = q.f;
    if (!ss2d_fast_ok(p)) return false;
    if (p.delta_softplus & (MS_SCAN_DT_FUSED | MS_SCAN_ACCUMULATE)) return false;
    if ((p.delta_softplus & MS_SCAN_DELTA_ACTIVATED) && !(p.delta_softplus & MS_SCAN_SOFTPLUS)) return false;
    auto act_ok = [](const float *ptr, int64_t sb, int64_t sgp, int64_t sd, int64_t sl) {
        return aligned16b(ptr) && sd == 1 && sb % 4 == 0 && sgp % 4 == 0 && sl % 4 == 0 && fits24(sl);
    };
    if (!act_ok(q.dout, q.dout_batch_stride, q.dout_group_stride, q.dout_d_stride, q.dout_l_stride)) return false;
    if (!act_ok(q.du, q.du_batch_stride, q.du_group_stride, q.du_d_stride, q.du_l_stride)) return false;
    if (!act_ok(q.ddelta, q.ddelta_batch_stride, q.ddelta_group_stride, q.ddelta_d_stride, q.ddelta_l_stride)) return false;
    if (p.u_d_stride != 1 || p.delta_d_stride != 1 || p.B_dstate_stride != 1 || p.C_dstate_stride != 1) return false;
    if (q.dB_dstate_stride != 1 || q.dC_dstate_stride != 1 || !fits24(q.dB_l_stride) || !fits24(q.dC_l_stride)) return false;
    return true;
}

int ss2d_bwd_launch(const MsScanBwdParams &q, int n_chunks, hipStream_t stream) {
    const MsScanParams &p = q.f;
    const int dpg = p.dim / p.n_groups, ncg = (dpg + 31) / 32;
    const dim3 grid((unsigned)((int64_t)p.batch * p.n_groups * ncg));
    static const int w3_env = [] { const char *e = getenv("MEDSCAN_BWD_W3"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
    const bool w3 = w3_env >= 0 ? w3_env == 1 : MS_BWD_W3_DEFAULT;
    const bool pre = (p.delta_softplus & MS_SCAN_DELTA_ACTIVATED) != 0;
    if (w3) {
        if (pre) hipLaunchKernelGGL((ss2d_bwd_kernel<true, true>), grid, dim3(kNTb), 0, stream, q, n_chunks);
        else hipLaunchKernelGGL((ss2d_bwd_kernel<false, true>), grid, dim3(kNTb), 0, stream, q, n_chunks);
    } else {
        if (pre) hipLaunchKernelGGL((ss2d_bwd_kernel<true, false>), grid, dim3(kNTb), 0, stream, q, n_chunks);
        else hipLaunchKernelGGL((ss2d_bwd_kernel<false, false>), grid, dim3(kNTb), 0, stream, q, n_chunks);
    }
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
