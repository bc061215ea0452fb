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
// Structure: lane = channel (scan_common.h); the NS waves of a workgroup own NPW states each.
// Chunks of MS_SCAN_CHUNK positions are visited last->first; per chunk the forward states of the
// wave's NPW states are RECOMPUTED from the saved state x[b,c-1] into registers (hist), then the
// reverse pass runs over the same registers.  Per-position sums over n are combined across waves with
// LDS float atomics.  The per-(l,n) dB/dC contributions of the 64 channels are summed by an
// in-register butterfly reduce-scatter (~2 cross-lane adds per value instead of 6 for a plain
// wave reduction), then one coalesced global atomic per (l,n) per 64-channel block.
#include <type_traits>
#include "scan_common.h"

namespace ms {

// Exchange-and-add step of the reduce-scatter at lane distance S: lanes with bit S clear keep `lo`,
// the others keep `hi`; both add the partner's copy of what they keep.
template <int S>
__device__ __forceinline__ float xchg_add(float lo, float hi, int lane) {
    const bool up = (lane & S) != 0;
    const float keep = up ? hi : lo;
    const float send = up ? lo : hi;
    return keep + __shfl_xor(send, S, 64);
}

template <int K>   // level K exchanges at distance 32 >> K
__device__ __forceinline__ float xchg_add_level(float lo, float hi, int lane) {
    return xchg_add<(32 >> K)>(lo, hi, lane);
}

__device__ __forceinline__ int bitrev(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}

// Streaming butterfly reduce-scatter of NPW values per position over the 64 lanes (channels).
// Levels 0..LI-1 (distances 32, 16) fold the state index, the remaining levels fold position bits,
// lowest bit first, so positions can be pushed one at a time in DESCENDING order and only one pending
// value per level is live.  One flush covers kSpan consecutive positions; afterwards
//   NPW == 4: lane j holds state (j >> 4),  position bitrev4(j & 15)           of the span
//   NPW == 2: lane j holds state (j >> 5),  position bitrev5(j & 31)
//   NPW == 1: lane j holds                  position bitrev5(j >> 1)           (both lanes of a pair)
template <int NPW>
struct ReduceScatter {
    static constexpr int LI = NPW == 4 ? 2 : (NPW == 2 ? 1 : 0);     // state levels
    static constexpr int LL = NPW == 4 ? 4 : 5;                      // position levels
    static constexpr int kSpan = 1 << LL;
    float pend[LL];

    template <int K>
    __device__ __forceinline__ void fold(int idx, float cur, int lane, float &result, bool &done) {
        if constexpr (K == LL) {
            if (NPW == 1) cur += __shfl_xor(cur, 1, 64);
            result = cur; done = true;
        } else {
            if (idx & 1) { pend[K] = cur; }
            else fold<K + 1>(idx >> 1, xchg_add_level<LI + K>(cur, pend[K], lane), lane, result, done);
        }
    }
    // l_in_span must be a compile-time constant after unrolling
    __device__ __forceinline__ bool push(int l_in_span, const float (&v)[NPW], int lane, float &result) {
        float cur;
        if constexpr (NPW == 4) {
            const float a = xchg_add_level<0>(v[0], v[2], lane);
            const float b = xchg_add_level<0>(v[1], v[3], lane);
            cur = xchg_add_level<1>(a, b, lane);
        } else if constexpr (NPW == 2) {
            cur = xchg_add_level<0>(v[0], v[1], lane);
        } else {
            cur = v[0];
        }
        bool done = false;
        fold<0>(l_in_span, cur, lane, result, done);
        return done;
    }
};

template <int NPW, bool LCONTIG, bool BC_CONTIG>
__global__ void __launch_bounds__(256, 2)
scan_bwd_kernel(const MsScanBwdParams q, const int n_chunks, const int nblk) {
    static_assert(kCL == 32, "chunk length is baked into the reduce-scatter spans");
    using RS = ReduceScatter<NPW>;
    const MsScanParams &p = q.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63;
    const int wave = uniform(tid >> 6);
    const int NS = nthreads >> 6;

    const int dpg = p.dim / p.n_groups;
    int bid = blockIdx.x;
    const int dblk = bid % nblk; bid /= nblk;
    const int g = bid % p.n_groups;
    const int b = bid / p.n_groups;
    const int nvalid = min(64, dpg - dblk * 64);
    const int d0 = g * dpg + dblk * 64;
    const bool active = lane < nvalid;
    const int d = d0 + (active ? lane : nvalid - 1);
    const int L = p.seqlen;

    float *su = smem;               // u tile      -> du tile
    float *sdl = su + kTile;        // delta' tile
    float *sg = sdl + kTile;        // dout tile   -> ddelta tile
    float *sbias = sg + kTile;      // [64]
    float *sdu = sbias + 64;        // [kCL][64] sum over states of du      (LDS atomics)
    float *sdd = sdu + kCL * 64;    // [kCL][64] sum over states of ddelta'
    float *sck = sdd + kCL * 64 + wave * (kCL / 4) * NPW * 64;   // per wave: h at the start of each 4-position batch

    const int n0 = wave * NPW;
    float An[NPW], A2[NPW], dhc[NPW], dAacc[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        An[i] = p.A[d * p.A_d_stride + (n0 + i) * p.A_dstate_stride];
        A2[i] = An[i] * kLog2e;
        dhc[i] = 0.0f; dAacc[i] = 0.0f;
    }
    const float Dv = (p.D != nullptr && wave == 0) ? p.D[d] : 0.0f;
    float dDacc = 0.0f, dbacc = 0.0f;
    if (wave == 0) sbias[lane] = p.delta_bias ? p.delta_bias[d] : 0.0f;
    __syncthreads();

    const float *ub = p.u + b * p.u_batch_stride + d0 * p.u_d_stride;
    const float *db = p.delta + b * p.delta_batch_stride + d0 * p.delta_d_stride;
    const float *gb = q.dout + b * q.dout_batch_stride + d0 * q.dout_d_stride;
    float *dub = q.du + b * q.du_batch_stride + d0 * q.du_d_stride;
    float *ddb = q.ddelta + b * q.ddelta_batch_stride + d0 * q.ddelta_d_stride;
    const float *Bb = p.B + b * p.B_batch_stride + g * p.B_group_stride + n0 * p.B_dstate_stride;
    const float *Cb = p.C + b * p.C_batch_stride + g * p.C_group_stride + n0 * p.C_dstate_stride;
    const int64_t sBn = p.B_dstate_stride, sBl = p.B_l_stride, sCn = p.C_dstate_stride, sCl = p.C_l_stride;
    const bool softplus = p.delta_softplus != 0;
    // (state, position-in-span) this lane owns after a reduce-scatter flush
    int rs_i, rs_l; bool rs_writer = true;
    if (NPW == 4)      { rs_i = lane >> 4; rs_l = bitrev(lane & 15, 4); }
    else if (NPW == 2) { rs_i = lane >> 5; rs_l = bitrev(lane & 31, 5); }
    else               { rs_i = 0; rs_l = bitrev(lane >> 1, 5); rs_writer = (lane & 1) == 0; }
    float *dBl = q.dB + (((int64_t)b * p.n_groups + g) * p.dstate + n0 + rs_i) * L;
    float *dCl = q.dC + (((int64_t)b * p.n_groups + g) * p.dstate + n0 + rs_i) * L;

    auto chunk = [&](auto full_tag, const int c) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int l0 = c * kCL;
        const int len = FULL ? kCL : L - l0;
        load_tile<LCONTIG>(su, ub + l0 * p.u_l_stride, p.u_d_stride, p.u_l_stride, nvalid, len, tid, nthreads);
        load_tile<LCONTIG>(sg, gb + l0 * q.dout_l_stride, q.dout_d_stride, q.dout_l_stride, nvalid, len, tid, nthreads);
#pragma unroll 4
        for (int idx = tid; idx < kCL * 64; idx += nthreads) {
            int l, dl; tile_coord<LCONTIG>(idx, l, dl);
            float v = 0.0f;
            if (l < len && dl < nvalid) {
                v = db[dl * p.delta_d_stride + (l0 + l) * p.delta_l_stride] + sbias[dl];
                if (softplus) v = softplus_ref(v);
            }
            sdl[l * kPitch + dl] = v;
            sdu[idx] = 0.0f; sdd[idx] = 0.0f;
        }
        __syncthreads();

        RS rsB, rsC;
        // ---- forward sweep: h at the start of every 4-position batch -> LDS (per-wave region) -------
        {
            float h[NPW];
#pragma unroll
            for (int i = 0; i < NPW; ++i)
                h[i] = c > 0 ? p.x[(((int64_t)b * n_chunks + (c - 1)) * p.dstate + n0 + i) * p.dim + d] : 0.0f;
#pragma unroll 1
            for (int kb = 0; kb < kCL / 4; ++kb) {
                const int lb = kb * 4;
                float Bv[NPW][4];
#pragma unroll
                for (int i = 0; i < NPW; ++i) {
                    sck[(kb * NPW + i) * 64 + lane] = h[i];
                    load_row<4, BC_CONTIG, FULL>(Bb + i * sBn, sBl, l0 + lb, L, Bv[i]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float dl_ = sdl[(lb + j) * kPitch + lane];
                    const float du_ = dl_ * su[(lb + j) * kPitch + lane];
#pragma unroll
                    for (int i = 0; i < NPW; ++i)
                        h[i] = fmaf(exp2_fast(dl_ * A2[i]), h[i], du_ * Bv[i][j]);
                }
            }
        }
        // ---- reverse sweep: per batch, recompute (a, h) into a 4-position register window, then run the
        //      adjoint recurrence backwards over the window --------------------------------------------
#pragma unroll 1
        for (int kb = kCL / 4 - 1; kb >= 0; --kb) {
            const int lb = kb * 4;
            float Bv[NPW][4], Cv[NPW][4], hs[NPW];
#pragma unroll
            for (int i = 0; i < NPW; ++i) {
                hs[i] = sck[(kb * NPW + i) * 64 + lane];
                load_row<4, BC_CONTIG, FULL>(Bb + i * sBn, sBl, l0 + lb, L, Bv[i]);
                load_row<4, BC_CONTIG, FULL>(Cb + i * sCn, sCl, l0 + lb, L, Cv[i]);
            }
            float dl_[4], uu[4], gg[4], av[4][NPW], hv[4][NPW];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dl_[j] = sdl[(lb + j) * kPitch + lane];
                uu[j] = su[(lb + j) * kPitch + lane];
                gg[j] = sg[(lb + j) * kPitch + lane];
                const float du_ = dl_[j] * uu[j];
#pragma unroll
                for (int i = 0; i < NPW; ++i) {
                    av[j][i] = exp2_fast(dl_[j] * A2[i]);
                    hv[j][i] = fmaf(av[j][i], j > 0 ? hv[j > 0 ? j - 1 : 0][i] : hs[i], du_ * Bv[i][j]);
                }
            }
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                const int l = lb + j;
                const float du_ = dl_[j] * uu[j];
                float du_l = Dv * gg[j], dd_l = 0.0f;
                dDacc = fmaf(gg[j], uu[j], dDacc);
                float vB[NPW], vC[NPW];
#pragma unroll
                for (int i = 0; i < NPW; ++i) {
                    const float hprev = j > 0 ? hv[j > 0 ? j - 1 : 0][i] : hs[i];
                    const float dhn = fmaf(Cv[i][j], gg[j], dhc[i]);
                    const float w = av[j][i] * hprev;
                    const float t = dhn * Bv[i][j];
                    du_l = fmaf(t, dl_[j], du_l);
                    dd_l = fmaf(t, uu[j], dd_l);
                    const float qv = dhn * w;
                    dd_l = fmaf(qv, An[i], dd_l);
                    dAacc[i] = fmaf(qv, dl_[j], dAacc[i]);
                    vB[i] = dhn * du_;
                    vC[i] = gg[j] * hv[j][i];
                    dhc[i] = av[j][i] * dhn;
                }
                if (NS == 1) { sdu[l * 64 + lane] = du_l; sdd[l * 64 + lane] = dd_l; }
                else { atomicAdd(&sdu[l * 64 + lane], du_l); atomicAdd(&sdd[l * 64 + lane], dd_l); }
                float rB = 0.0f, rC = 0.0f;
                const bool doneB = rsB.push(l & (RS::kSpan - 1), vB, lane, rB);
                const bool doneC = rsC.push(l & (RS::kSpan - 1), vC, lane, rC);
                if (doneB && doneC) {
                    const int span0 = l & ~(RS::kSpan - 1);
                    if (rs_writer && (FULL || span0 + rs_l < len)) {
                        atomicAdd(dBl + l0 + span0 + rs_l, rB);
                        atomicAdd(dCl + l0 + span0 + rs_l, rC);
                    }
                }
            }
        }
        __syncthreads();   // all waves are done with the u / dout tiles and with their LDS atomics

        // ---- finish ddelta (softplus'), move the sums into the pitch-65 tiles, store -------------
        for (int l = wave; l < kCL; l += NS) {
            float dd = sdd[l * 64 + lane];
            if (softplus) dd *= -expm1f(-sdl[l * kPitch + lane]);
            dbacc += dd;
            su[l * kPitch + lane] = sdu[l * 64 + lane];
            sg[l * kPitch + lane] = dd;
        }
        __syncthreads();
        store_tile<LCONTIG>(su, dub + l0 * q.du_l_stride, q.du_d_stride, q.du_l_stride, nvalid, len, tid, nthreads);
        store_tile<LCONTIG>(sg, ddb + l0 * q.ddelta_l_stride, q.ddelta_d_stride, q.ddelta_l_stride, nvalid, len, tid, nthreads);
        __syncthreads();
    };

    {
        int c = n_chunks - 1;
        if (L % kCL != 0) { chunk(std::false_type{}, c); --c; }
        for (; c >= 0; --c) chunk(std::true_type{}, c);
    }

    if (active) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) atomicAdd(q.dA + (int64_t)d * p.dstate + n0 + i, dAacc[i]);
        if (q.dD != nullptr && wave == 0) atomicAdd(q.dD + d, dDacc);
        if (q.ddelta_bias != nullptr) atomicAdd(q.ddelta_bias + d, dbacc);
    }
}

int validate_scan(const MsScanParams &p);

template <int NPW>
static int launch_bwd(const MsScanBwdParams &q, int ns, int n_chunks, hipStream_t stream) {
    const MsScanParams &p = q.f;
    const int dpg = p.dim / p.n_groups;
    const int nblk = (dpg + 63) / 64;
    const dim3 grid((unsigned)((int64_t)p.batch * p.n_groups * nblk));
    const dim3 block(64 * ns);
    const size_t smem = sizeof(float) * (3 * kTile + 64 + 2 * kCL * 64 + (size_t)ns * (kCL / 4) * NPW * 64);
    const bool lcontig = p.u_l_stride == 1 && p.delta_l_stride == 1 && q.dout_l_stride == 1 &&
                         q.du_l_stride == 1 && q.ddelta_l_stride == 1;
    const bool dcontig = p.u_d_stride == 1 && p.delta_d_stride == 1 && q.dout_d_stride == 1 &&
                         q.du_d_stride == 1 && q.ddelta_d_stride == 1;
    const bool bcc = p.B_l_stride == 1 && p.C_l_stride == 1;
#define MS_LAUNCH(LC, BC) hipLaunchKernelGGL((scan_bwd_kernel<NPW, LC, BC>), grid, block, smem, stream, q, n_chunks, nblk)
    if (lcontig || !dcontig) { if (bcc) MS_LAUNCH(true, true); else MS_LAUNCH(true, false); }
    else                     { if (bcc) MS_LAUNCH(false, true); else MS_LAUNCH(false, false); }
#undef MS_LAUNCH
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int scan_bwd_dispatch(const MsScanBwdParams &q, hipStream_t stream) {
    const MsScanParams &p = q.f;
    int rc = validate_scan(p);
    if (rc != MS_OK) return rc;
    if (!q.dout || !q.du || !q.ddelta || !q.dA || !q.dB || !q.dC) return MS_ERR_NULL;
    if (p.batch == 0 || p.seqlen == 0) return MS_OK;
    const int n_chunks = (p.seqlen + kCL - 1) / kCL;
    if (n_chunks > 1 && !p.x) return MS_ERR_NULL;
    // bwd keeps 256-thread workgroups (register budget): at most 4 waves split the state axis
    int npw = 0, ns = 0;
    const int cands[3] = {4, 2, 1};
    for (int k = 0; k < 3 && !npw; ++k)
        if (p.dstate % cands[k] == 0 && p.dstate / cands[k] <= 4) { npw = cands[k]; ns = p.dstate / cands[k]; }
    if (!npw) return MS_ERR_DSTATE;
    switch (npw) {
        case 1: return launch_bwd<1>(q, ns, n_chunks, stream);
        case 2: return launch_bwd<2>(q, ns, n_chunks, stream);
        case 4: return launch_bwd<4>(q, ns, n_chunks, stream);
    }
    return MS_ERR_DSTATE;
}

}  // namespace ms
