// The chunked (state-space-duality) evaluation of the SSD / Mamba-2 operator on the gfx950 matrix cores -- what the reference reaches
// through `mamba_chunk_scan_combined` of mamba_ssm 2.2.2 (a Triton dependency outside the reference tree; call sites
// /root/reference/CNN_Mamba.py:523-537, CrossMamba/CrossMamba_fusion_2b2.py:327,590):
//     h_t = exp(dt_t A) h_{t-1} + dt_t B_t (x) x_t ,   y_t = C_t . h_t + D x_t          (A scalar per head, dt = softplus(raw + bias))
// Because the decay is a scalar per head, the recurrence over a chunk of Q = 64 positions is a product of small matrices:
//     inside the chunk    Y   = ((C B^T) o L) X'            L[i][j] = exp(cum_i - cum_j) for j <= i,  X' = dt * x,  cum = prefix sums of dt A
//     chunk states        S_c = B^T (dec o X')              dec_j = exp(cum_last - cum_j)
//     carried states      S_in[c] = exp(cum_last[c-1]) S_in[c-1] + S[c-1]                   (ms_ssd_chunk_carry, ssd_carry.hip)
//     state -> output     Y  += diag(exp(cum)) C S_in ,  y = Y + D x
// Every product runs on v_mfma_f32_16x16x4_f32 (fp32 operands, products and sums: the operator keeps its 1e-3 fp32 contract) from LDS
// tiles of 64 x 64 floats; the elementwise pieces (decay masks, dt scaling, exp) ride in the tile staging or the accumulator
// epilogues, so none of the (chunks x heads x 64 x 64) mask / (64 x 64) score tensors of the torch formulation exists in memory.
// Shapes: headdim 64, n_groups 1, d_state (all directions) a multiple of 64 -- the reference's configurations (64 for CNN_Mamba.VSSM,
// 512 for VFEFM); anything else stays on the torch formulation (cnn_mamba._ssd_chunked).
//
// One helper does all the matrix work: a wave accumulates a 16-row x 64-column block of  out[row][col] += sum_k Aop[row][k] Bop[col][k],
// k < 64, from two LDS tiles, each addressed either as [row][k] (one ds_read_b128 per fragment) or as [k][row] (four ds_read_b32), with
// an optional per-k scale on the A fragments.  Tile pitch 68 floats: both access forms are bank-conflict free.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "medscan.h"

namespace ms {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kQc = 64;            // chunk length = tile edge
constexpr int kTPc = kQc + 4;      // LDS tile pitch (floats)
constexpr int kTilec = kQc * kTPc;

__device__ __forceinline__ float4 ld4c(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4c(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float exp_f(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// acc[nt] (rows row0 .. row0+15, columns 16 nt .. 16 nt + 15) += sum_{k<64} Aop[row][k] * Bop[col][k]
//   A_T = false: Aop[row][k] = sA[row * pitch + k]        A_T = true: Aop[row][k] = sA[k * pitch + row]      (same for B)
//   ksc: Aop[row][k] is multiplied by ksc[k]
// Result layout (as in gemm_f32.hip): lane (fr = lane & 15, fq = lane >> 4) holds out[row0 + fr][16 nt + 4 fq + r], r = 0..3.
template <bool A_T, bool B_T, bool SCALE>
__device__ __forceinline__ void mma_16x64(f32x4 (&acc)[4], const float *sA, int row0, const float *sB, const float *ksc, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int k0 = ks * 16 + fq * 4;
        f32x4 fa;
        if (!A_T) { const float4 v = ld4c(sA + (row0 + fr) * kTPc + k0); fa = (f32x4){v.x, v.y, v.z, v.w}; }
        else { const float *b = sA + k0 * kTPc + row0 + fr; fa = (f32x4){b[0], b[kTPc], b[2 * kTPc], b[3 * kTPc]}; }
        if (SCALE) { const float4 s = ld4c(ksc + k0); fa[0] *= s.x; fa[1] *= s.y; fa[2] *= s.z; fa[3] *= s.w; }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 fb;
            if (!B_T) { const float4 v = ld4c(sB + (nt * 16 + fr) * kTPc + k0); fb = (f32x4){v.x, v.y, v.z, v.w}; }
            else { const float *b = sB + k0 * kTPc + nt * 16 + fr; fb = (f32x4){b[0], b[kTPc], b[2 * kTPc], b[3 * kTPc]}; }
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[s], fa[s], acc[nt], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ void zero4(f32x4 (&a)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// A [64 rows][64 columns] fp32 tile, rows `ld` floats apart in memory, staged into LDS as s[row][col] (pitch kTPc) in two phases so that the
// NEXT tile's loads are in flight while the current one is multiplied: fetch() -> registers (thread -> 4 pieces of 16 bytes, row =
// id / 16, piece = id % 16; rows >= nrows read as zero), put() -> LDS, with an optional per-row scale `rscale` (LDS, 64 floats).
struct TileRegs {
    float4 v[4];
    __device__ __forceinline__ void fetch(const float *g, int64_t ld, int nrows, int tid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + 256 * i, row = id >> 4, pc = id & 15;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < nrows) v[i] = ld4c(g + (int64_t)row * ld + pc * 4);
        }
    }
    __device__ __forceinline__ void put(float *s, int tid, const float *rscale = nullptr) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + 256 * i, row = id >> 4, pc = id & 15;
            float4 t = v[i];
            if (rscale) { const float sc = rscale[row]; t.x *= sc; t.y *= sc; t.z *= sc; t.w *= sc; }
            st4c(s + row * kTPc + pc * 4, t);
        }
    }
};
__device__ __forceinline__ void stage_tile(float *s, const float *g, int64_t ld, int nrows, int tid, const float *rscale = nullptr) {
    TileRegs r;
    r.fetch(g, ld, nrows, tid);
    r.put(s, tid, rscale);
}

// ---- prep: dt' = softplus(dt + bias) (or dt + bias), cum = prefix sums of dt' A inside each chunk, decay = exp(cum_last) -----------
// one thread per (batch, chunk, head); dtv / cum: (batch, chunks, heads, 64)
__global__ void __launch_bounds__(256)
ssd_prep_kernel(const float *__restrict__ dt, const float *__restrict__ A, const float *__restrict__ bias, int softplus,
                float *__restrict__ dtv, float *__restrict__ cum, float *__restrict__ decay, int batch, int L, int nc, int H) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (int64_t)batch * nc * H) return;
    const int hh = (int)(id % H), c = (int)((id / H) % nc), b = (int)(id / ((int64_t)H * nc));
    const float a = A[hh], bs = bias ? bias[hh] : 0.0f;
    float run = 0.0f;
    float *dv = dtv + id * kQc, *cv = cum + id * kQc;
    for (int t = 0; t < kQc; ++t) {
        const int l = c * kQc + t;
        float v = 0.0f;
        if (l < L) {
            const float raw = dt[((int64_t)b * L + l) * H + hh] + bs;
            v = raw;
            if (softplus) {          // F.softplus: x > 20 ? x : log1p(exp(x)), compensated as in scan_common.h softplus_ref
                const float e = exp_f(raw), u = 1.0f + e, dn = u - 1.0f;
                const float lg = __builtin_amdgcn_logf(u) * 0.6931471805599453f;
                v = raw <= 20.0f ? (dn == 0.0f ? e : lg * (e * __builtin_amdgcn_rcpf(dn))) : raw;
            }
        }
        run = fmaf(v, a, run);
        dv[t] = v; cv[t] = run;
    }
    decay[id] = exp_f(run);
}

// ---- CB = C_c B_c^T per (batch, chunk): (batch, chunks, 64, 64) -------------------------------------------------------------
__global__ void __launch_bounds__(256)
ssd_cb_kernel(const float *__restrict__ Bm, const float *__restrict__ Cm, float *__restrict__ CB, int L, int nc, int N) {
    __shared__ __attribute__((aligned(16))) float sB[kTilec], sC[kTilec];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = blockIdx.x % nc, b = blockIdx.x / nc;
    const int l0 = c * kQc, nrows = min(kQc, L - l0);
    const float *Bg = Bm + ((int64_t)b * L + l0) * N, *Cg = Cm + ((int64_t)b * L + l0) * N;
    f32x4 acc[4];
    zero4(acc);
    TileRegs rb, rc;
    rb.fetch(Bg, N, nrows, tid); rc.fetch(Cg, N, nrows, tid);
    for (int n0 = 0; n0 < N; n0 += kQc) {
        __syncthreads();
        rb.put(sB, tid); rc.put(sC, tid);
        __syncthreads();
        if (n0 + kQc < N) { rb.fetch(Bg + n0 + kQc, N, nrows, tid); rc.fetch(Cg + n0 + kQc, N, nrows, tid); }
        mma_16x64<false, false, false>(acc, sC, 16 * w, sB, nullptr, lane);        // rows i: C, columns j: B, k: states
    }
    const int fr = lane & 15, fq = lane >> 4;
    float *o = CB + ((int64_t)blockIdx.x * kQc + 16 * w + fr) * kQc + 4 * fq;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) st4c(o + 16 * nt, make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]));
}

// ---- per (batch, chunk, head): the chunk's state contribution S and the inside-the-chunk output Y ------------------------------------
//   S[b][c][nn][hh*64 + p]   = sum_j B[j][nn] dec_j X'[j][p]                  (every 64-state tile of B)
//   y[b][l0 + i][hh][p]      = sum_{j <= i} CB[i][j] exp(cum_i - cum_j) X'[j][p]
__global__ void __launch_bounds__(256)
ssd_intra_kernel(const float *__restrict__ x, const float *__restrict__ Bm, const float *__restrict__ CB, const float *__restrict__ dtv,
                 const float *__restrict__ cum, float *__restrict__ S, float *__restrict__ y, int L, int nc, int H, int N) {
    __shared__ __attribute__((aligned(16))) float sB[kTilec], sXt[kTilec], sM[kTilec];
    __shared__ __attribute__((aligned(16))) float sCum[kQc], sDec[kQc];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hh = blockIdx.x % H, c = (blockIdx.x / H) % nc, b = blockIdx.x / (H * nc);
    const int l0 = c * kQc, nrows = min(kQc, L - l0);
    const int64_t bch = ((int64_t)b * nc + c) * H + hh;
    if (tid < kQc) {
        const float cv = cum[bch * kQc + tid];
        sCum[tid] = cv;
        sDec[tid] = exp_f(cum[bch * kQc + kQc - 1] - cv);
    }
    // X'^T: sXt[p][j] = dt'_j x[j][p].  lane = j, wave w takes p = 16 w .. 16 w + 15 (column writes: consecutive lanes, consecutive banks)
    {
        const int j = lane;
        const float dj = dtv[bch * kQc + j];
        const float *xr = x + (((int64_t)b * L + l0 + j) * H + hh) * kQc + 16 * w;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < nrows) v = ld4c(xr + 4 * q);
            float *d = sXt + (16 * w + 4 * q) * kTPc + j;
            d[0] = v.x * dj; d[kTPc] = v.y * dj; d[2 * kTPc] = v.z * dj; d[3 * kTPc] = v.w * dj;
        }
    }
    const int fr = lane & 15, fq = lane >> 4;
    const float *Bg = Bm + ((int64_t)b * L + l0) * N;
    float *Sg = S + (((int64_t)b * nc + c) * N) * ((int64_t)H * kQc) + (int64_t)hh * kQc;
    TileRegs rb;
    rb.fetch(Bg, N, nrows, tid);
    for (int n0 = 0; n0 < N; n0 += kQc) {
        __syncthreads();                                        // the previous tile's fragments are read (first trip: sXt / sDec written)
        rb.put(sB, tid);
        __syncthreads();
        if (n0 + kQc < N) rb.fetch(Bg + n0 + kQc, N, nrows, tid);
        f32x4 acc[4];
        zero4(acc);
        mma_16x64<true, false, true>(acc, sB, 16 * w, sXt, sDec, lane);       // rows nn: B^T (scaled by dec_j along k = j), columns p: X'^T
        float *o = Sg + (int64_t)(n0 + 16 * w + fr) * ((int64_t)H * kQc) + 4 * fq;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) st4c(o + 16 * nt, make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]));
    }
    if (y == nullptr) return;                                    // states only (the backward recomputes S / S_in, not Y)
    // M = (C B^T) o L for this wave's 16 rows -> LDS (read back only by this wave), then Y = M X'
    {
        const int i = 16 * w + fr;
        const float ci = sCum[i];
        const float *cb = CB + (((int64_t)b * nc + c) * kQc + i) * kQc + 4 * fq;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float4 v = ld4c(cb + 16 * nt), cj = ld4c(sCum + 16 * nt + 4 * fq);
            const int j0 = 16 * nt + 4 * fq;
            float4 m;
            m.x = j0 + 0 <= i ? v.x * exp_f(ci - cj.x) : 0.0f;
            m.y = j0 + 1 <= i ? v.y * exp_f(ci - cj.y) : 0.0f;
            m.z = j0 + 2 <= i ? v.z * exp_f(ci - cj.z) : 0.0f;
            m.w = j0 + 3 <= i ? v.w * exp_f(ci - cj.w) : 0.0f;
            st4c(sM + i * kTPc + j0, m);
        }
    }
    __syncthreads();
    f32x4 acc[4];
    zero4(acc);
    mma_16x64<false, false, false>(acc, sM, 16 * w, sXt, nullptr, lane);          // rows i: M, columns p: X'^T, k: j
    const int i = 16 * w + fr;
    if (i < nrows) {
        float *o = y + (((int64_t)b * L + l0 + i) * H + hh) * kQc + 4 * fq;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) st4c(o + 16 * nt, make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]));
    }
}

// ---- per (batch, chunk, head): y += diag(exp(cum)) C S_in + D x -----------------------------------------------------------------------
__global__ void __launch_bounds__(256)
ssd_off_kernel(const float *__restrict__ x, const float *__restrict__ Cm, const float *__restrict__ Sin, const float *__restrict__ cum,
               const float *__restrict__ D, int d_has_hdim, float *__restrict__ y, int L, int nc, int H, int N) {
    __shared__ __attribute__((aligned(16))) float sC[kTilec], sS[kTilec];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hh = blockIdx.x % H, c = (blockIdx.x / H) % nc, b = blockIdx.x / (H * nc);
    const int l0 = c * kQc, nrows = min(kQc, L - l0);
    const int64_t bch = ((int64_t)b * nc + c) * H + hh;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[4];
    zero4(acc);
    if (c > 0) {                                                                    // chunk 0 starts from the zero state
        const float *Cg = Cm + ((int64_t)b * L + l0) * N;
        const float *Sg = Sin + (((int64_t)b * nc + c) * N) * ((int64_t)H * kQc) + (int64_t)hh * kQc;
        const int64_t HS = (int64_t)H * kQc;
        TileRegs rc, rs;
        rc.fetch(Cg, N, nrows, tid); rs.fetch(Sg, HS, kQc, tid);
        for (int n0 = 0; n0 < N; n0 += kQc) {
            __syncthreads();
            rc.put(sC, tid); rs.put(sS, tid);
            __syncthreads();
            if (n0 + kQc < N) { rc.fetch(Cg + n0 + kQc, N, nrows, tid); rs.fetch(Sg + (int64_t)(n0 + kQc) * HS, HS, kQc, tid); }
            mma_16x64<false, true, false>(acc, sC, 16 * w, sS, nullptr, lane);        // rows i: C, columns p: S_in^T (stored [nn][p]), k: nn
        }
    }
    const int i = 16 * w + fr;
    if (i < nrows) {
        const float e = exp_f(cum[bch * kQc + i]);
        const int64_t off = (((int64_t)b * L + l0 + i) * H + hh) * kQc + 4 * fq;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float4 yi = ld4c(y + off + 16 * nt), xv = ld4c(x + off + 16 * nt);
            float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (D) {
                if (d_has_hdim) dv = ld4c(D + (int64_t)hh * kQc + 16 * nt + 4 * fq);
                else { const float d0 = D[hh]; dv = make_float4(d0, d0, d0, d0); }
            }
            st4c(y + off + 16 * nt, make_float4(fmaf(dv.x, xv.x, fmaf(e, acc[nt][0], yi.x)), fmaf(dv.y, xv.y, fmaf(e, acc[nt][1], yi.y)),
                                                 fmaf(dv.z, xv.z, fmaf(e, acc[nt][2], yi.z)), fmaf(dv.w, xv.w, fmaf(e, acc[nt][3], yi.w))));
        }
    }
}


// =====================================================================================================================================
// Backward.  With E_i = exp(cum_i), dec_j = exp(cum_last - cum_j), L, M, X' as above and W = C S_in:
//   off path      dS_in[nn][p] = sum_i C[i][nn] E_i dy[i][p]          dcum_i += E_i sum_p dy[i][p] W[i][p]
//   carry         dS = reverse carry of dS_in, d decay                                        (ms_ssd_chunk_carry, reverse)
//   state path    G = B dS ;  dX' += dec o G ;  ddec_j = sum_p X'[j][p] G[j][p] ;  dcum_j -= dec_j ddec_j ;  dtot += sum_j dec_j ddec_j
//   inside        dM = dy X'^T ;  T = dM o M ;  dcum_i += rowsum_i T - colsum_i T ;  dCB += dM o L ;  dX' += M^T dy
//   per position  dx = dt' dX' + D dy ;  ddt' = sum_p dX' x + A da ;  da_t = dtot + sum_{s >= t} dcum_s ;  dA += sum_t da_t dt'_t
//   B, C          dC[i][nn] = sum_h E_i dy_h[i] . S_in,h[nn]  +  sum_j dCB[i][j] B[j][nn]
//                 dB[j][nn] = sum_h dec_j X'_h[j] . dS_h[nn]  +  sum_i dCB[i][j] C[i][nn]          (one workgroup per (batch, chunk, state
//                 tile) sums every head's contribution: plain stores, no atomics over the heads)
// =====================================================================================================================================

// ---- per (batch, chunk, head): dS_in and the off-path part of dcum -------------------------------------------------------------------
__global__ void __launch_bounds__(256)
ssd_bwd_off_kernel(const float *__restrict__ dy, const float *__restrict__ Cm, const float *__restrict__ Sin, const float *__restrict__ cum,
                   float *__restrict__ dSin, float *__restrict__ dcum, int L, int nc, int H, int N) {
    __shared__ __attribute__((aligned(16))) float sC[kTilec], sS[kTilec], sDY[kTilec];
    __shared__ __attribute__((aligned(16))) float sE[kQc];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hh = blockIdx.x % H, c = (blockIdx.x / H) % nc, b = blockIdx.x / (H * nc);
    const int l0 = c * kQc, nrows = min(kQc, L - l0);
    const int64_t bch = ((int64_t)b * nc + c) * H + hh;
    if (c == 0) {                                               // zero entering state: W = 0, and the carry never reads dS_in[0]
        if (tid < kQc) dcum[bch * kQc + tid] = 0.0f;
        return;
    }
    if (tid < kQc) sE[tid] = exp_f(cum[bch * kQc + tid]);
    stage_tile(sDY, dy + (((int64_t)b * L + l0) * H + hh) * kQc, (int64_t)H * kQc, nrows, tid);
    const int fr = lane & 15, fq = lane >> 4;
    const float *Cg = Cm + ((int64_t)b * L + l0) * N;
    const int64_t HS = (int64_t)H * kQc;
    const float *Sg = Sin + (((int64_t)b * nc + c) * N) * HS + (int64_t)hh * kQc;
    float *dSg = dSin + (((int64_t)b * nc + c) * N) * HS + (int64_t)hh * kQc;
    f32x4 wacc[4];
    zero4(wacc);
    TileRegs rc, rs;
    rc.fetch(Cg, N, nrows, tid); rs.fetch(Sg, HS, kQc, tid);
    for (int n0 = 0; n0 < N; n0 += kQc) {
        __syncthreads();
        rc.put(sC, tid); rs.put(sS, tid);
        __syncthreads();
        if (n0 + kQc < N) { rc.fetch(Cg + n0 + kQc, N, nrows, tid); rs.fetch(Sg + (int64_t)(n0 + kQc) * HS, HS, kQc, tid); }
        mma_16x64<false, true, false>(wacc, sC, 16 * w, sS, nullptr, lane);             // W: rows i (C), columns p (S_in^T), k: nn
        f32x4 acc[4];
        zero4(acc);
        mma_16x64<true, true, true>(acc, sC, 16 * w, sDY, sE, lane);                    // dS_in: rows nn (C^T scaled by E_i along k = i), columns p (dy^T)
        float *o = dSg + (int64_t)(n0 + 16 * w + fr) * HS + 4 * fq;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) st4c(o + 16 * nt, make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]));
    }
    const int i = 16 * w + fr;
    float part = 0.0f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const float4 g = ld4c(sDY + i * kTPc + 16 * nt + 4 * fq);
        part += (g.x * wacc[nt][0] + g.y * wacc[nt][1]) + (g.z * wacc[nt][2] + g.w * wacc[nt][3]);
    }
    part += __shfl_xor(part, 16); part += __shfl_xor(part, 32);
    if (fq == 0) dcum[bch * kQc + i] = part * sE[i];
}

// ---- per (batch, chunk, head): everything but dB / dC ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
ssd_bwd_main_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ Bm, const float *__restrict__ CB,
                    const float *__restrict__ dS, const float *__restrict__ dtv, const float *__restrict__ cum,
                    const float *__restrict__ decay, const float *__restrict__ ddecay, const float *__restrict__ A,
                    const float *__restrict__ D, int d_has_hdim, int softplus, const float *__restrict__ dcum_off,
                    float *__restrict__ dx, float *__restrict__ ddt, float *__restrict__ dA, float *__restrict__ dbias, float *__restrict__ dD,
                    float *__restrict__ dCB, int L, int nc, int H, int N) {
    __shared__ __attribute__((aligned(16))) float sX[kTilec], sDY[kTilec], sB[kTilec], sU[kTilec];       // sB: B tile, then M; sU: dS tile, then T
    __shared__ __attribute__((aligned(16))) float sCum[kQc], sDec[kQc], sDtv[kQc], sDdec[kQc], sRow[kQc], sCol[kQc], sDdtv[kQc], sRed[kQc];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hh = blockIdx.x % H, c = (blockIdx.x / H) % nc, b = blockIdx.x / (H * nc);
    const int l0 = c * kQc, nrows = min(kQc, L - l0);
    const int64_t bch = ((int64_t)b * nc + c) * H + hh;
    const int64_t HS = (int64_t)H * kQc;
    if (tid < kQc) {
        const float cv = cum[bch * kQc + tid];
        sCum[tid] = cv;
        sDec[tid] = exp_f(cum[bch * kQc + kQc - 1] - cv);
        sDtv[tid] = dtv[bch * kQc + tid];
        sRed[tid] = 0.0f;
    }
    __syncthreads();
    const int64_t row0 = (((int64_t)b * L + l0) * H + hh) * kQc;
    stage_tile(sX, x + row0, HS, nrows, tid, sDtv);                                        // X'[j][p]
    stage_tile(sDY, dy + row0, HS, nrows, tid);
    const int fr = lane & 15, fq = lane >> 4;
    const int r = 16 * w + fr;                                                              // this lane's tile row (i or j)
    // ---- state path: G = B dS -----------------------------------------------------------------------------------------------------
    f32x4 G[4];
    zero4(G);
    {
        const float *Bg = Bm + ((int64_t)b * L + l0) * N;
        const float *dSg = dS + (((int64_t)b * nc + c) * N) * HS + (int64_t)hh * kQc;
        TileRegs rb, rs;
        rb.fetch(Bg, N, nrows, tid); rs.fetch(dSg, HS, kQc, tid);
        for (int n0 = 0; n0 < N; n0 += kQc) {
            __syncthreads();
            rb.put(sB, tid); rs.put(sU, tid);
            __syncthreads();
            if (n0 + kQc < N) { rb.fetch(Bg + n0 + kQc, N, nrows, tid); rs.fetch(dSg + (int64_t)(n0 + kQc) * HS, HS, kQc, tid); }
            mma_16x64<false, true, false>(G, sB, 16 * w, sU, nullptr, lane);                // rows j (B), columns p (dS^T), k: nn
        }
    }
    f32x4 dXp[4];
    {
        const float dj = sDec[r];
        float part = 0.0f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float4 xv = ld4c(sX + r * kTPc + 16 * nt + 4 * fq);
            part += (xv.x * G[nt][0] + xv.y * G[nt][1]) + (xv.z * G[nt][2] + xv.w * G[nt][3]);
            dXp[nt] = G[nt] * dj;
        }
        part += __shfl_xor(part, 16); part += __shfl_xor(part, 32);
        if (fq == 0) sDdec[r] = part;
    }
    // ---- inside the chunk: M, L for this lane's 16 entries of row i = r ------------------------------------------------------------
    f32x4 Mr[4], Lr[4];
    {
        const float ci = sCum[r];
        const float *cb = CB + (((int64_t)b * nc + c) * kQc + r) * kQc + 4 * fq;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float4 v = ld4c(cb + 16 * nt), cj = ld4c(sCum + 16 * nt + 4 * fq);
            const int j0 = 16 * nt + 4 * fq;
            Lr[nt][0] = j0 + 0 <= r ? exp_f(ci - cj.x) : 0.0f; Lr[nt][1] = j0 + 1 <= r ? exp_f(ci - cj.y) : 0.0f;
            Lr[nt][2] = j0 + 2 <= r ? exp_f(ci - cj.z) : 0.0f; Lr[nt][3] = j0 + 3 <= r ? exp_f(ci - cj.w) : 0.0f;
            Mr[nt][0] = v.x * Lr[nt][0]; Mr[nt][1] = v.y * Lr[nt][1]; Mr[nt][2] = v.z * Lr[nt][2]; Mr[nt][3] = v.w * Lr[nt][3];
        }
    }
    __syncthreads();                                            // every wave is done with the last B / dS tiles
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) st4c(sB + r * kTPc + 16 * nt + 4 * fq, make_float4(Mr[nt][0], Mr[nt][1], Mr[nt][2], Mr[nt][3]));     // sB := M
    f32x4 dM[4];
    zero4(dM);
    mma_16x64<false, false, false>(dM, sDY, 16 * w, sX, nullptr, lane);                     // dM: rows i (dy), columns j (X'), k: p
    {
        float rs = 0.0f;
        float *dcb = dCB + (((int64_t)b * nc + c) * kQc + r) * kQc + 4 * fq;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const f32x4 T = dM[nt] * Mr[nt];
            rs += (T[0] + T[1]) + (T[2] + T[3]);
            st4c(sU + r * kTPc + 16 * nt + 4 * fq, make_float4(T[0], T[1], T[2], T[3]));      // sU := T = dM o M
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = dM[nt][q] * Lr[nt][q];
                if (16 * nt + 4 * fq + q <= r) atomicAdd(dcb + 16 * nt + q, v);                // dCB += dM o L (summed over the heads)
            }
        }
        rs += __shfl_xor(rs, 16); rs += __shfl_xor(rs, 32);
        if (fq == 0) sRow[r] = rs;
    }
    __syncthreads();                                            // M and T are complete
    mma_16x64<true, true, false>(dXp, sB, 16 * w, sDY, nullptr, lane);                      // dX' += M^T dy: rows j, columns p, k: i
    if (tid < kQc) {
        float cs = 0.0f;
        for (int i = 0; i < kQc; ++i) cs += sU[i * kTPc + tid];
        sCol[tid] = cs;
    }
    // ---- per position: dx, ddt' partial, dD -----------------------------------------------------------------------------------------
    {
        const float dtj = sDtv[r];
        float pdt = 0.0f, pdd = 0.0f;
        const bool live = r < nrows;
        const float d0 = (D && !d_has_hdim) ? D[hh] : 0.0f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int pc = 16 * nt + 4 * fq;
            float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) xv = ld4c(x + row0 + (int64_t)r * HS + pc);
            const float4 g = ld4c(sDY + r * kTPc + pc);
            float4 dv = make_float4(d0, d0, d0, d0);
            if (D && d_has_hdim) dv = ld4c(D + (int64_t)hh * kQc + pc);
            pdt += (dXp[nt][0] * xv.x + dXp[nt][1] * xv.y) + (dXp[nt][2] * xv.z + dXp[nt][3] * xv.w);
            if (live) st4c(dx + row0 + (int64_t)r * HS + pc, make_float4(fmaf(dtj, dXp[nt][0], dv.x * g.x), fmaf(dtj, dXp[nt][1], dv.y * g.y),
                                                                        fmaf(dtj, dXp[nt][2], dv.z * g.z), fmaf(dtj, dXp[nt][3], dv.w * g.w)));
            if (dD) {
                if (d_has_hdim) {       // per (head, p): sums over the rows through LDS, one global atomic per column and workgroup
                    atomicAdd(sRed + pc, g.x * xv.x); atomicAdd(sRed + pc + 1, g.y * xv.y);
                    atomicAdd(sRed + pc + 2, g.z * xv.z); atomicAdd(sRed + pc + 3, g.w * xv.w);
                } else pdd += (g.x * xv.x + g.y * xv.y) + (g.z * xv.z + g.w * xv.w);
            }
        }
        pdt += __shfl_xor(pdt, 16); pdt += __shfl_xor(pdt, 32);
        if (fq == 0) sDdtv[r] = pdt;
        if (dD && !d_has_hdim) {
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) pdd += __shfl_xor(pdd, m);
            if (lane == 0) atomicAdd(dD + hh, pdd);
        }
    }
    __syncthreads();
    if (dD && d_has_hdim && tid < kQc) atomicAdd(dD + (int64_t)hh * kQc + tid, sRed[tid]);
    // ---- the chunk's 64 positions: dcum -> da (suffix sums), ddt, dA, dbias ---------------------------------------------------------------
    if (w == 0) {
        const int t = lane;
        const float dec = sDec[t], dde = sDdec[t];
        float dc = dcum_off[bch * kQc + t] + sRow[t] - sCol[t] - dec * dde;
        float tt = dec * dde;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) tt += __shfl_xor(tt, m);
        const float dtot = tt + ddecay[bch] * decay[bch];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { const float o = __shfl_down(dc, m); if (t + m < 64) dc += o; }      // inclusive suffix sum
        const float da = dc + dtot;
        const float dv = sDtv[t];
        const bool live = t < nrows;
        const float sig = softplus ? 1.0f - exp_f(-dv) : 1.0f;                               // softplus' = sigmoid(raw) = 1 - exp(-softplus(raw))
        const float draw = live ? fmaf(da, A[hh], sDdtv[t]) * sig : 0.0f;
        if (live) ddt[((int64_t)b * L + l0 + t) * H + hh] = draw;
        float pa = live ? da * dv : 0.0f, pb = draw;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { pa += __shfl_xor(pa, m); pb += __shfl_xor(pb, m); }
        if (lane == 0) {
            atomicAdd(dA + hh, pa);
            if (dbias) atomicAdd(dbias + hh, pb);
        }
    }
}

// ---- per (batch, chunk, NTW 64-state tiles): dB and dC, every head's contribution summed in registers ------------------------------------
// NTW tiles per workgroup: a head's dy / x tiles are staged once for NTW state tiles (with one tile per workgroup the kernel re-read dy
// and x N / 64 times: 3.3 of its 6.4 GB per stage-0 scan of VFEFM); the S_in / dS tiles stream through registers one step ahead.
template <int NTW>
__global__ void __launch_bounds__(256)
ssd_bwd_bc_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ Bm, const float *__restrict__ Cm,
                  const float *__restrict__ Sin, const float *__restrict__ dS, const float *__restrict__ dtv, const float *__restrict__ cum,
                  const float *__restrict__ dCB, float *__restrict__ dB, float *__restrict__ dC, int L, int nc, int H, int N) {
    __shared__ __attribute__((aligned(16))) float sA1[kTilec], sB1[kTilec], sA2[kTilec], sB2[kTilec];
    __shared__ __attribute__((aligned(16))) float sE[kQc], sSc[kQc];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ngr = N / (kQc * NTW);
    const int t0 = blockIdx.x % ngr, c = (blockIdx.x / ngr) % nc, b = blockIdx.x / (ngr * nc);
    const int l0 = c * kQc, nrows = min(kQc, L - l0), n0 = t0 * kQc * NTW;
    const int64_t HS = (int64_t)H * kQc;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 aC[NTW][4], aB[NTW][4];
#pragma unroll
    for (int t = 0; t < NTW; ++t) { zero4(aC[t]); zero4(aB[t]); }
    const float *Sg = Sin + (((int64_t)b * nc + c) * N + n0) * HS, *dSg = dS + (((int64_t)b * nc + c) * N + n0) * HS;
    TileRegs r1, r2;                                                                           // the next (S_in, dS) tiles
    if (c > 0) r1.fetch(Sg, HS, kQc, tid);
    r2.fetch(dSg, HS, kQc, tid);
    for (int hh = 0; hh < H; ++hh) {
        const int64_t bch = ((int64_t)b * nc + c) * H + hh;
        __syncthreads();                                                                       // the previous head's tiles are read
        if (tid < kQc) {
            const float cv = cum[bch * kQc + tid];
            sE[tid] = exp_f(cv);
            sSc[tid] = dtv[bch * kQc + tid] * exp_f(cum[bch * kQc + kQc - 1] - cv);          // dt'_j dec_j
        }
        __syncthreads();
        const int64_t row0 = (((int64_t)b * L + l0) * H + hh) * kQc;
        if (c > 0) stage_tile(sA1, dy + row0, HS, nrows, tid, sE);                             // E_i dy[i][p]
        stage_tile(sA2, x + row0, HS, nrows, tid, sSc);                                        // dec_j X'[j][p]
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            if (t > 0) __syncthreads();                                                        // the previous state tiles are read
            if (c > 0) r1.put(sB1, tid);                                                       // S_in[nn][p]
            r2.put(sB2, tid);                                                                  // dS[nn][p]
            __syncthreads();
            {       // the next pair of state tiles: (hh, t + 1), or (hh + 1, 0)
                const int tn = t + 1 < NTW ? t + 1 : 0, hn = t + 1 < NTW ? hh : hh + 1;
                if (hn < H) {
                    const int64_t off = (int64_t)tn * kQc * HS + (int64_t)hn * kQc;
                    if (c > 0) r1.fetch(Sg + off, HS, kQc, tid);
                    r2.fetch(dSg + off, HS, kQc, tid);
                }
            }
            if (c > 0) mma_16x64<false, false, false>(aC[t], sA1, 16 * w, sB1, nullptr, lane);  // rows i, columns nn, k: p
            mma_16x64<false, false, false>(aB[t], sA2, 16 * w, sB2, nullptr, lane);             // rows j, columns nn, k: p
        }
    }
    __syncthreads();
    stage_tile(sA1, dCB + ((int64_t)b * nc + c) * kQc * kQc, kQc, kQc, tid);                   // dCB[i][j]
    const int r = 16 * w + fr;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        if (t > 0) __syncthreads();
        stage_tile(sB1, Bm + ((int64_t)b * L + l0) * N + n0 + t * kQc, N, nrows, tid);         // B[j][nn]
        stage_tile(sA2, Cm + ((int64_t)b * L + l0) * N + n0 + t * kQc, N, nrows, tid);         // C[i][nn]
        __syncthreads();
        mma_16x64<false, true, false>(aC[t], sA1, 16 * w, sB1, nullptr, lane);                 // dC += dCB B:    rows i, columns nn (B^T of [j][nn]), k: j
        mma_16x64<true, true, false>(aB[t], sA1, 16 * w, sA2, nullptr, lane);                  // dB += dCB^T C:  rows j (dCB^T), columns nn, k: i
        if (r < nrows) {
            float *oc = dC + ((int64_t)b * L + l0 + r) * N + n0 + t * kQc + 4 * fq, *ob = dB + ((int64_t)b * L + l0 + r) * N + n0 + t * kQc + 4 * fq;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                st4c(oc + 16 * nt, make_float4(aC[t][nt][0], aC[t][nt][1], aC[t][nt][2], aC[t][nt][3]));
                st4c(ob + 16 * nt, make_float4(aB[t][nt][0], aB[t][nt][1], aB[t][nt][2], aB[t][nt][3]));
            }
        }
    }
}

}  // namespace

static bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// phase 0: prep + CB + intra / states (everything before the carry).  workspace tensors are the caller's.
int ssd_chunk_fwd_dispatch(const float *x, const float *dt, const float *A, const float *B, const float *C, const float *dt_bias,
                           int softplus, float *dtv, float *cum, float *decay, float *CB, float *S, float *y, int batch, int L, int H,
                           int P, int N, hipStream_t s) {
    if (!x || !dt || !A || !B || !C || !dtv || !cum || !decay || !CB || !S) return MS_ERR_NULL;      // y == NULL: states only
    if (batch < 0 || L <= 0 || H <= 0 || P != kQc || N <= 0 || N % kQc != 0) return MS_ERR_SHAPE;
    if (!al16(x) || !al16(B) || !al16(C) || !al16(S) || (y && !al16(y)) || !al16(CB) || !al16(cum) || !al16(dtv)) return MS_ERR_STRIDE;
    if (batch == 0) return MS_OK;
    const int nc = (L + kQc - 1) / kQc;
    const int64_t nbh = (int64_t)batch * nc * H;
    if (nbh >= (1LL << 31)) return MS_ERR_SHAPE;
    hipLaunchKernelGGL(ssd_prep_kernel, dim3((unsigned)((nbh + 255) / 256)), dim3(256), 0, s, dt, A, dt_bias, softplus, dtv, cum, decay, batch, L, nc, H);
    hipLaunchKernelGGL(ssd_cb_kernel, dim3((unsigned)(batch * nc)), dim3(256), 0, s, B, C, CB, L, nc, N);
    hipLaunchKernelGGL(ssd_intra_kernel, dim3((unsigned)nbh), dim3(256), 0, s, x, B, CB, dtv, cum, S, y, L, nc, H, N);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

// phase 1 (after ms_ssd_chunk_carry produced S_in): y += diag(exp(cum)) C S_in + D x
int ssd_chunk_fwd_off_dispatch(const float *x, const float *C, const float *Sin, const float *cum, const float *D, int d_has_hdim, float *y,
                               int batch, int L, int H, int P, int N, hipStream_t s) {
    if (!x || !C || !Sin || !cum || !y) return MS_ERR_NULL;
    if (batch < 0 || L <= 0 || H <= 0 || P != kQc || N <= 0 || N % kQc != 0) return MS_ERR_SHAPE;
    if (!al16(x) || !al16(C) || !al16(Sin) || !al16(y) || (D && d_has_hdim && !al16(D))) return MS_ERR_STRIDE;
    if (batch == 0) return MS_OK;
    const int nc = (L + kQc - 1) / kQc;
    hipLaunchKernelGGL(ssd_off_kernel, dim3((unsigned)((int64_t)batch * nc * H)), dim3(256), 0, s, x, C, Sin, cum, D, d_has_hdim, y, L, nc, H, N);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

// backward phase 0 (before the reverse carry): dS_in, off-path dcum
int ssd_chunk_bwd_off_dispatch(const float *dy, const float *C, const float *Sin, const float *cum, float *dSin, float *dcum, int batch, int L,
                               int H, int P, int N, hipStream_t s) {
    if (!dy || !C || !Sin || !cum || !dSin || !dcum) return MS_ERR_NULL;
    if (batch < 0 || L <= 0 || H <= 0 || P != kQc || N <= 0 || N % kQc != 0) return MS_ERR_SHAPE;
    if (!al16(dy) || !al16(C) || !al16(Sin) || !al16(dSin)) return MS_ERR_STRIDE;
    if (batch == 0) return MS_OK;
    const int nc = (L + kQc - 1) / kQc;
    hipLaunchKernelGGL(ssd_bwd_off_kernel, dim3((unsigned)((int64_t)batch * nc * H)), dim3(256), 0, s, dy, C, Sin, cum, dSin, dcum, L, nc, H, N);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

// backward phase 1 (after the reverse carry produced dS and ddecay): dx, ddt, dA, dbias, dD, then dB, dC.  dCB (batch, nc, 64, 64), dA, dbias,
// dD are ACCUMULATED (zero them); dx, ddt, dB, dC are written.
int ssd_chunk_bwd_dispatch(const float *x, const float *dy, const float *B, const float *C, const float *CB, const float *Sin, const float *dS,
                           const float *dtv, const float *cum, const float *decay, const float *ddecay, const float *A, const float *D,
                           int d_has_hdim, int softplus, const float *dcum_off, float *dx, float *ddt, float *dA, float *dbias, float *dD,
                           float *dCB, float *dB, float *dC, int batch, int L, int H, int P, int N, hipStream_t s) {
    if (!x || !dy || !B || !C || !CB || !Sin || !dS || !dtv || !cum || !decay || !ddecay || !A || !dcum_off || !dx || !ddt || !dA || !dCB || !dB || !dC)
        return MS_ERR_NULL;
    if (batch < 0 || L <= 0 || H <= 0 || P != kQc || N <= 0 || N % kQc != 0) return MS_ERR_SHAPE;
    if (!al16(x) || !al16(dy) || !al16(B) || !al16(C) || !al16(CB) || !al16(Sin) || !al16(dS) || !al16(dx) || !al16(dCB) || !al16(dB) || !al16(dC) ||
        (D && d_has_hdim && !al16(D))) return MS_ERR_STRIDE;
    if (batch == 0) return MS_OK;
    const int nc = (L + kQc - 1) / kQc;
    hipLaunchKernelGGL(ssd_bwd_main_kernel, dim3((unsigned)((int64_t)batch * nc * H)), dim3(256), 0, s, x, dy, B, CB, dS, dtv, cum, decay, ddecay, A, D,
                       d_has_hdim, softplus, dcum_off, dx, ddt, dA, dbias, dD, dCB, L, nc, H, N);
    // state tiles per workgroup of the dB / dC kernel: as many as divide N / 64, up to 4, while the grid still covers the chip
    const int ntl = N / kQc;
    if (ntl % 4 == 0 && (int64_t)batch * nc * (ntl / 4) >= 512)
        hipLaunchKernelGGL((ssd_bwd_bc_kernel<4>), dim3((unsigned)((int64_t)batch * nc * (ntl / 4))), dim3(256), 0, s, x, dy, B, C, Sin, dS, dtv, cum, dCB, dB, dC, L, nc, H, N);
    else if (ntl % 2 == 0 && (int64_t)batch * nc * (ntl / 2) >= 512)
        hipLaunchKernelGGL((ssd_bwd_bc_kernel<2>), dim3((unsigned)((int64_t)batch * nc * (ntl / 2))), dim3(256), 0, s, x, dy, B, C, Sin, dS, dtv, cum, dCB, dB, dC, L, nc, H, N);
    else
        hipLaunchKernelGGL((ssd_bwd_bc_kernel<1>), dim3((unsigned)((int64_t)batch * nc * ntl)), dim3(256), 0, s, x, dy, B, C, Sin, dS, dtv, cum, dCB, dB, dC, L, nc, H, N);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
