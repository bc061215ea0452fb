// Projection GEMMs of SS2D on the gfx950 matrix cores: in_proj, x_proj, out_proj of MedMamba.py:284,326,397,469,480 and
// their input / weight gradients -- token matrices with a huge row count (B*H*W = 3 136 .. 200 704) and small feature
// dimensions (48 .. 1536): every one of them is bound by streaming the activation once, not by the matrix cores.
//
//   C[i][j] (+)= sum_k Aop[i][k] * Bop[j][k]          bf16 MFMA (v_mfma_f32_16x16x32_bf16), fp32 accumulation
//     Aop[i][k] = a_trans ? A[k*lda + i] : A[i*lda + k]        A, B: bf16 or fp32 in memory (fp32 is rounded to bf16 while
//     Bop[j][k] = b_trans ? B[k*ldb + j] : B[j*ldb + k]        the tile is staged: no cast kernels, no bf16 weight copies)
//   forward      y  = x  @ W^T : A = x  (M x K),          B = W (N x K)
//   input grad   dx = dy @ W   : A = dy (M x N),          B = W (N x K) read transposed (b_trans)
//   weight grad  dW = dy^T @ x : A = dy read transposed,  B = x read transposed; the reduction over the tokens is split over
//                                blockIdx.z and the partial tiles are added with fp32 atomics (split-K inside the kernel:
//                                no partial-sum tensor, no reduction launch)
// Tile: 128 x BN outputs per 256-thread workgroup, 64-deep k-steps staged through LDS in the canonical [row][k] form
// (row pitch 80 bf16 = 160 B.  gfx950 serves a ds_read_b128 in groups of 16 lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32: a
// fragment read has rows 0-3 / 12-15 with k-piece q and rows 4-11 with piece q + 1 in one group.  The 144-byte pitch of rounds 1-2 --
// conflict-free for 16 CONSECUTIVE lanes -- costs 8 LDS cycles per fragment under that grouping, 160 bytes the minimal 4:
// tools/lds_swizzle_search.py), next
// k-step's global loads in flight while the current one is multiplied.  The MFMA's A operand is the Bop tile and its B
// operand the Aop tile, so a lane ends up with FOUR CONSECUTIVE output columns of one output row (16-byte / 8-byte stores).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "medscan.h"
#include "gemm_common.h"

namespace ms {

constexpr int kBM = 128, kBK = 64, kPitch = kBK + 16;      // bf16 elements (160-byte rows: see the row-pitch note at TileStage)

// Stage one [ROWS x 64] tile of op(X) into LDS.  Two phases so that the global loads of the next k-step can be in flight
// during the MFMAs: fetch() -> registers, put() -> LDS.  Both kinds are staged with 16-byte pieces in the order they have in
// memory (coalesced loads, one ds_write_b128 per piece):
//   plain      : X[row*ld + k] -> s[row][k]  (pitch kPitch);  MFMA fragment = one ds_read_b128 (8 consecutive k of a row)
//   transposed : X[k*ld + row] -> s[k][row]  (pitch ROWS + 16); MFMA fragment = two ds_read_b64_tr_b16 (the hardware's
//                transposing LDS read: a 16-lane group fetches 4 k-rows x 16 columns and every lane receives the 4 k-values
//                of ITS column) -- no scattered 2-byte LDS writes, no transposed copy of the weight in memory
template <int ROWS, bool F32, bool TR>
struct TileStage {
    static constexpr int NP = ROWS * kBK / 8 / 256;      // pieces per thread
    static constexpr int kPT = ROWS + 16;                // pitch of the transposed image (bf16)
    static constexpr int kLds = TR ? kBK * kPT : ROWS * kPitch;
    bf16x8 r[NP];
    __device__ __forceinline__ void fetch(const void *X, int64_t ld, int row0, int k0, int nrows, int K, int tid) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int id = tid + 256 * i;
            if (!TR) {
                const int row = id / (kBK / 8), kc = id % (kBK / 8);
                const int gr = row0 + row, gk = k0 + kc * 8;
                r[i] = load_piece<F32>(X, (int64_t)gr * ld + gk, gr < nrows ? min(8, max(0, K - gk)) : 0);
            } else {
                const int k = id / (ROWS / 8), rc = id % (ROWS / 8);
                const int gk = k0 + k, gr = row0 + rc * 8;
                r[i] = load_piece<F32>(X, (int64_t)gk * ld + gr, gk < K ? min(8, max(0, nrows - gr)) : 0);
            }
        }
    }
    __device__ __forceinline__ void put(unsigned short *s, int tid) const {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int id = tid + 256 * i;
            if (!TR) {
                const int row = id / (kBK / 8), kc = id % (kBK / 8);
                *reinterpret_cast<bf16x8 *>(s + row * kPitch + kc * 8) = r[i];
            } else {
                const int k = id / (ROWS / 8), rc = id % (ROWS / 8);
                *reinterpret_cast<bf16x8 *>(s + k * kPT + rc * 8) = r[i];
            }
        }
    }
    // the MFMA operand fragment of the tile's rows [16 t, 16 t + 16), k-step ks: lane (fr = lane & 15, fq = lane >> 4) gets
    // op(X)[16 t + fr][32 ks + 8 fq + j], j = 0..7
    // ALTK (both operands transposed -- the weight-gradient form): a lane's 8 k-values are rows 4 fq .. 4 fq + 3 and 16 + 4 fq .. of the
    // k-step instead of 8 fq .. 8 fq + 7.  Any assignment works as long as both operands use the same one; with this one the 32 lanes a
    // ds_read_b64_tr_b16 serves together read 8 CONSECUTIVE k-rows, which the pitch spreads over all banks -- the default assignment
    // (forced by the plain operand's contiguous 8 k when only one side is transposed) has rows r and r + 8 in one pass, and those
    // collide at every pitch that keeps rows 16-byte aligned (a third of these instantiations' LDS cycles were conflicts).
    template <bool ALTK = false>
    static __device__ __forceinline__ bf16x8 frag(const unsigned short *s, int t, int ks, int lane) {
        const int fr = lane & 15, fq = lane >> 4;
        if (!TR) return *reinterpret_cast<const bf16x8 *>(s + (t * 16 + fr) * kPitch + ks * 32 + fq * 8);
        // lane 4q + p of a 16-lane group addresses row q, columns 4p .. 4p+3 of the 4 x 16 block; it receives column (lane & 15)
        const unsigned short *b = s + (ks * 32 + fq * (ALTK ? 4 : 8) + (fr >> 2)) * kPT + t * 16 + 4 * (fr & 3);
        typedef bf16x4 __attribute__((address_space(3))) *lds_p;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(b));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(b + (ALTK ? 16 : 4) * kPT));
        return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
};

// The reduce pass of a BatchNorm's backward riding on the store modes' epilogue (ms_gemm_bf16_bnbwd: dx = dz W of the conv branch's 1x1
// convolution, MedMamba.py:525, whose result is the gradient w.r.t. the output of the BatchNorm + ReLU in front of it): per output
// column (= channel) sum dy' and sum dy' * xhat over the rows (= pixels), dy' = C[m][n] as stored * [relu passed], xhat from the
// pre-normalisation activation xpre[m][n] (bf16) -- into MS_BN_REPLICAS replica rows, as conv3x3.hip's BRED epilogue does.
struct GemmRed {
    const unsigned short *xpre; int64_t xps;
    const float *gamma, *beta, *mean, *rstd;
    int relu;
    float *sums;                                   // nullptr: no reduction
};
template <int CTRL> __device__ __forceinline__ float gemm_row_ror_add(float x) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(unsigned, x), CTRL, 0xF, 0xF, true));
}

// CMODE: 0 = fp32 store, 1 = bf16 store, 2 = fp32 atomic add (split-K partial), 3 = fp32 atomic add into C^T
// BM x BN outputs per workgroup; the 4 waves split the ROWS (MT = BM / 64 tiles of 16 rows each), every wave spans all BN
// columns (TNT = BN / 16 tiles): an activation row block is read once for up to 192 output columns.
template <int BM, int BN, bool AF32, bool BF32, bool ATR, bool BTR, int CMODE, bool RED = false>
__global__ void __launch_bounds__(256)
gemm_bf16_kernel(const void *A, const void *B, void *C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int k_per_split,
                 const float *__restrict__ bias, int relu, GemmRed red) {
    constexpr int TNT = BN / 16, MT = BM / 64;
    __shared__ __attribute__((aligned(16))) unsigned short sA[TileStage<BM, AF32, ATR>::kLds];
    __shared__ __attribute__((aligned(16))) unsigned short sB[TileStage<BN, BF32, BTR>::kLds];
    // accumulating modes: the workgroup's fp32 tile goes through LDS so that a wave's atomics cover 64 CONSECUTIVE addresses (the
    // MFMA layout would scatter them over 16 rows) and start at a position that depends on the k-slice (concurrent slices of one
    // output tile then hit different addresses instead of queueing on the same ones)
    constexpr int kCP = BN + 1;
    __shared__ float sCt[CMODE >= 2 ? BM * kCP : 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kbeg = blockIdx.z * k_per_split, kend = min(K, kbeg + k_per_split);
    if (kbeg >= kend) return;

    f32x4 acc[TNT][MT];
#pragma unroll
    for (int a = 0; a < TNT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // CMODE 2 with `bias` set: `bias` is an OUTPUT -- rowsum[m] += sum_k Aop[m][k] (the bias gradient of a 1x1 convolution next to
    // its weight gradient: Aop = dy^T, so the row sums are dy's column sums).  One more MFMA per A fragment against an all-ones
    // fragment in the first column block's workgroups, instead of a separate reduction pass over dy.
    const bool row_sums = CMODE == 2 && bias != nullptr && blockIdx.y == 0;
    f32x4 accs[CMODE == 2 ? MT : 1];
#pragma unroll
    for (int b = 0; b < (CMODE == 2 ? MT : 1); ++b) accs[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

    // register staging two k-steps deep: the loads of steps k+1 and k+2 are in flight while step k is multiplied (with
    // K <= 128 -- in_proj / x_proj of the first stages -- the whole operand is requested before the first MFMA)
    using TA = TileStage<BM, AF32, ATR>;
    using TB = TileStage<BN, BF32, BTR>;
    TA ta[2];
    TB tb[2];
    ta[0].fetch(A, lda, m0, kbeg, M, kend, tid);
    tb[0].fetch(B, ldb, n0, kbeg, N, kend, tid);
    if (kbeg + kBK < kend) {
        ta[1].fetch(A, lda, m0, kbeg + kBK, M, kend, tid);
        tb[1].fetch(B, ldb, n0, kbeg + kBK, N, kend, tid);
    }
    auto step = [&](TA &sa_, TB &sb_, int k0) {          // stage set (sa_, sb_) holds k-step k0
        __syncthreads();                                 // the previous k-step's fragments have been read
        sa_.put(sA, tid);
        sb_.put(sB, tid);
        __syncthreads();
        if (k0 + 2 * kBK < kend) {                      // refill this set with k-step k0 + 2
            sa_.fetch(A, lda, m0, k0 + 2 * kBK, M, kend, tid);
            sb_.fetch(B, ldb, n0, k0 + 2 * kBK, N, kend, tid);
        }
#pragma unroll
        for (int ks = 0; ks < kBK / 32; ++ks) {
            bf16x8 fa[MT];
#pragma unroll
            for (int b = 0; b < MT; ++b) fa[b] = TA::template frag<ATR && BTR>(sA, w * MT + b, ks, lane);
            if constexpr (CMODE == 2) {
                if (row_sums) {
#pragma unroll
                    for (int b = 0; b < MT; ++b) accs[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[b], accs[b], 0, 0, 0);
                }
            }
#pragma unroll
            for (int a = 0; a < TNT; ++a) {
                const bf16x8 fb = TB::template frag<ATR && BTR>(sB, a, ks, lane);
#pragma unroll
                for (int b = 0; b < MT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa[b], acc[a][b], 0, 0, 0);
            }
        }
    };
    const int fr = lane & 15, fq = lane >> 4;
    for (int k0 = kbeg; k0 < kend; k0 += 2 * kBK) {
        step(ta[0], tb[0], k0);
        if (k0 + kBK < kend) step(ta[1], tb[1], k0 + kBK);
    }
    if constexpr (CMODE >= 2) {
        if constexpr (CMODE == 2) {
            if (row_sums && fq == 0) {                   // D[i][j] = sum_k Aop[m0 + .. + j][k] in every row i: lane (fr, 0) holds column fr
#pragma unroll
                for (int b = 0; b < MT; ++b) {
                    const int m = m0 + w * (16 * MT) + b * 16 + fr;
                    if (m < M) atomicAdd(const_cast<float *>(bias) + m, accs[b][0]);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < MT; ++b)
#pragma unroll
            for (int a = 0; a < TNT; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) sCt[(w * (16 * MT) + b * 16 + fr) * kCP + a * 16 + fq * 4 + r] = acc[a][b][r];
        __syncthreads();
        constexpr int kTot = BM * BN;
        const int rot = (int)(blockIdx.z % (kTot / 256)) * 256;
        float *Cf = static_cast<float *>(C);
        for (int i = 0; i < kTot / 256; ++i) {
            const int e = (i * 256 + tid + rot) % kTot;
            int ml, nl;
            if (CMODE == 2) { nl = e % BN; ml = e / BN; }            // consecutive threads along n: rows of C
            else            { ml = e % BM; nl = e / BM; }            // consecutive threads along m: rows of C^T
            const int m = m0 + ml, n = n0 + nl;
            if (m < M && n < N) {
                const float v = sCt[ml * kCP + nl];
                atomicAdd(CMODE == 2 ? Cf + (int64_t)m * ldc + n : Cf + (int64_t)n * ldc + m, v);
            }
        }
        return;
    }
    if constexpr (RED) {        // its own instantiations: as a run-time branch the accumulators below cost EVERY input-gradient launch
        {                       // registers (x_proj's dx at stage 0: 65 -> 88 us)
            float s1[TNT][4], s2[TNT][4];
#pragma unroll
            for (int a = 0; a < TNT; ++a) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { s1[a][r] = 0.f; s2[a][r] = 0.f; }
                const int n = n0 + a * 16 + fq * 4;                 // N % 4 == 0 (host): a lane's 4 columns are in or out together
                if (n >= N) continue;
                const float4 mu = *reinterpret_cast<const float4 *>(red.mean + n), rs = *reinterpret_cast<const float4 *>(red.rstd + n);
                const float4 ga = *reinterpret_cast<const float4 *>(red.gamma + n), be = *reinterpret_cast<const float4 *>(red.beta + n);
                const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
                const float gav[4] = {ga.x, ga.y, ga.z, ga.w}, bev[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
                for (int b = 0; b < MT; ++b) {
                    const int m = m0 + w * (16 * MT) + b * 16 + fr;
                    if (m >= M) continue;
                    const uint2 t = *reinterpret_cast<const uint2 *>(red.xpre + (int64_t)m * red.xps + n);
                    const float xv[4] = {__builtin_bit_cast(float, t.x << 16), __builtin_bit_cast(float, t.x & 0xFFFF0000u),
                                         __builtin_bit_cast(float, t.y << 16), __builtin_bit_cast(float, t.y & 0xFFFF0000u)};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xh = (xv[r] - muv[r]) * rsv[r];
                        float d = CMODE == 1 ? (float)(__bf16)acc[a][b][r] : acc[a][b][r];
                        if (red.relu && fmaf(xh, gav[r], bev[r]) <= 0.f) d = 0.f;
                        s1[a][r] += d; s2[a][r] = fmaf(d, xh, s2[a][r]);
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < TNT; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {                       // sum over the 16 row lanes (lane bits 0-3)
                    s1[a][r] = gemm_row_ror_add<0x121>(gemm_row_ror_add<0x122>(gemm_row_ror_add<0x124>(gemm_row_ror_add<0x128>(s1[a][r]))));
                    s2[a][r] = gemm_row_ror_add<0x121>(gemm_row_ror_add<0x122>(gemm_row_ror_add<0x124>(gemm_row_ror_add<0x128>(s2[a][r]))));
                }
            __syncthreads();                                        // every wave has read its last fragments: sA is free
            float *sr = reinterpret_cast<float *>(sA);              // [wave][2][BN]
            if (fr == 0) {
#pragma unroll
                for (int a = 0; a < TNT; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sr[(w * 2 + 0) * BN + a * 16 + fq * 4 + r] = s1[a][r];
                        sr[(w * 2 + 1) * BN + a * 16 + fq * 4 + r] = s2[a][r];
                    }
            }
            __syncthreads();
            for (int e = tid; e < 2 * BN; e += 256) {
                const int st = e / BN, c = e - st * BN;
                if (n0 + c < N)
                    atomicAdd(red.sums + ((blockIdx.x % MS_BN_REPLICAS) * 2 + st) * N + n0 + c,
                              (sr[(0 * 2 + st) * BN + c] + sr[(1 * 2 + st) * BN + c]) + (sr[(2 * 2 + st) * BN + c] + sr[(3 * 2 + st) * BN + c]));
            }
        }
    }
    // D[i][j]: i = n within the tile (row 4*fq + r of the accumulator), j = m within the tile (column fr)
#pragma unroll
    for (int b = 0; b < MT; ++b) {
        const int m = m0 + w * (16 * MT) + b * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int a = 0; a < TNT; ++a) {
            const int n = n0 + a * 16 + fq * 4;
            if (n >= N) continue;
            f32x4 v = acc[a][b];
            if (CMODE < 2) {                        // epilogue of the store modes: + bias[column], ReLU (1x1 convolution + bias + ReLU)
                if (bias) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < N) v[r] += bias[n + r];
                }
                if (relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
                }
            }
            if (CMODE == 2) {
                float *c = static_cast<float *>(C) + (int64_t)m * ldc + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < N) atomicAdd(c + r, v[r]);
            } else if (CMODE == 3) {             // accumulate into the TRANSPOSED output: C[j][i] += (16 lanes = 16 consecutive i)
                float *c = static_cast<float *>(C) + (int64_t)n * ldc + m;
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < N) atomicAdd(c + (int64_t)r * ldc, v[r]);
            } else if (CMODE == 0) {
                float *c = static_cast<float *>(C) + (int64_t)m * ldc + n;
                if (n + 4 <= N && (ldc & 3) == 0) *reinterpret_cast<float4 *>(c) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < N) c[r] = v[r];
                }
            } else {
                unsigned short *c = static_cast<unsigned short *>(C) + (int64_t)m * ldc + n;
                if (n + 4 <= N && (ldc & 3) == 0) {
                    uint2 pk;
                    pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
                    pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                    *reinterpret_cast<uint2 *>(c) = pk;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < N) c[r] = f2bf(v[r]);
                }
            }
        }
    }
}

template <int BM, int BN, bool AF32, bool BF32, bool ATR, bool BTR>
static void launch_c(int c_mode, dim3 grid, hipStream_t s, const void *A, const void *B, void *C, int M, int N, int K, int64_t lda,
                     int64_t ldb, int64_t ldc, int kps, const float *bias, int relu, GemmRed red) {
    // the accumulating modes exist for the weight gradient (both operands transposed) only
    if constexpr (!(ATR && BTR)) { if (c_mode >= 2) return; }
    if (red.sums != nullptr) {              // BatchNorm backward reduce in the epilogue: input-gradient form, 64-column tiles, bf16 store
        if constexpr (!ATR && BTR && BN == 64)
            if (c_mode == 1)
                hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, AF32, BF32, ATR, BTR, 1, true>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, kps, bias, relu, red);
        return;
    }
    switch (c_mode) {
        case 0: hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, AF32, BF32, ATR, BTR, 0>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, kps, bias, relu, red); break;
        case 1: hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, AF32, BF32, ATR, BTR, 1>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, kps, bias, relu, red); break;
        default:
            if constexpr (ATR && BTR) {
                if (c_mode == 2) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, AF32, BF32, ATR, BTR, 2>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, kps, bias, relu, red);
                else hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, AF32, BF32, ATR, BTR, 3>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, kps, bias, relu, red);
            }
            break;
    }
}

template <int BM, int BN>
static void launch_layout(bool af32, bool bf32, bool atr, bool btr, int c_mode, dim3 grid, hipStream_t s, const void *A, const void *B,
                          void *C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int kps, const float *bias, int relu, GemmRed red) {
#define MS_GEMM_CASE(AF, BF, AT, BT) \
    if (af32 == AF && bf32 == BF && atr == AT && btr == BT) { launch_c<BM, BN, AF, BF, AT, BT>(c_mode, grid, s, A, B, C, M, N, K, lda, ldb, ldc, kps, bias, relu, red); return; }
    // the combinations the projections use (activations bf16 or fp32, weights fp32)
    MS_GEMM_CASE(false, false, false, false)  // y = x(bf16) W(bf16 copy)^T
    MS_GEMM_CASE(true, false, false, false)
    MS_GEMM_CASE(false, false, false, true)   // dx = dy(bf16) W(bf16 copy)
    MS_GEMM_CASE(true, false, false, true)
    MS_GEMM_CASE(false, true, false, false)   // y = x(bf16) W^T
    MS_GEMM_CASE(true, true, false, false)    // y = x(fp32) W^T
    MS_GEMM_CASE(false, true, false, true)    // dx = dy(bf16) W
    MS_GEMM_CASE(true, true, false, true)     // dx = dy(fp32) W
    MS_GEMM_CASE(false, false, true, true)    // dW = dy(bf16)^T x(bf16)
    MS_GEMM_CASE(true, false, true, true)     // dW = dy(fp32)^T x(bf16)
    MS_GEMM_CASE(false, true, true, true)     // dW = dy(bf16)^T x(fp32)
    MS_GEMM_CASE(true, true, true, true)      // dW = dy(fp32)^T x(fp32)
#undef MS_GEMM_CASE
}

static bool combo_built(bool af32, bool bf32, bool atr, bool btr, int c_mode) {
    if (!atr) return c_mode < 2;              // forward / input gradient: store modes
    return btr;                               // weight gradient: every mode
}

static int g_force_bm = 0, g_force_bn = 0;   // experiments (ms_debug_gemm_tile): 0 = the heuristic below
void gemm_debug_tile(int bm, int bn) { g_force_bm = bm; g_force_bn = bn; }

static int gemm_impl(const void *A, int a_f32, int a_trans, int64_t lda, const void *B, int b_f32, int b_trans, int64_t ldb, void *C,
                     int c_mode, int64_t ldc, int M, int N, int K, int k_splits, const float *bias, int relu, GemmRed red, hipStream_t stream) {
    if (!A || !B || !C) return MS_ERR_NULL;
    // the epilogue belongs to the store modes; in c_mode 2 `bias` is the row-sum OUTPUT of the weight-gradient form (see the kernel)
    if ((relu && c_mode >= 2) || (bias && c_mode == 3) || (bias && c_mode == 2 && !(a_trans && b_trans))) return MS_ERR_SHAPE;
    if (M <= 0 || N <= 0 || K <= 0 || k_splits < 1 || c_mode < 0 || c_mode > 3) return MS_ERR_SHAPE;
    if (k_splits > 1 && c_mode < 2) return MS_ERR_SHAPE;
    if (!combo_built(a_f32, b_f32, a_trans, b_trans, c_mode)) return MS_ERR_UNSUPPORTED;
    // 16-byte pieces: leading dimensions in units of 8 bf16 / 4 fp32, 16-byte aligned bases
    const int64_t ga = a_f32 ? 4 : 8, gb = b_f32 ? 4 : 8;
    if (lda % ga || ldb % gb || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return MS_ERR_STRIDE;
    int kps = (K + k_splits - 1) / k_splits;
    kps = (kps + kBK - 1) / kBK * kBK;
    const int nz = (K + kps - 1) / kps;
    // Tile choice, from cold-cache sweeps on MI355X (tools/bench_gemm_cold.py: inside the training step every operand comes from
    // HBM / MALL, not from a warm L2): 64-column blocks win almost everywhere -- more workgroups in flight hide the load latency,
    // and the re-read of the A rows per column block is cheap while A is small.  Only when the A operand itself is large
    // (>= 48 MB: x_proj at stage 0 streams 77 MB of fp32 activations) is it read as few times as possible: the narrowest of
    // 64 / 128 / 192 that covers N in the fewest passes.
    const int64_t a_bytes = (int64_t)M * K * (a_f32 ? 4 : 2);
    int bn = 64;
    if (!a_trans && a_bytes >= (48ll << 20)) {
        const int passes = (N + 191) / 192;
        const int per = (N + passes - 1) / passes;
        bn = per <= 64 ? 64 : per <= 128 ? 128 : 192;
    }
    if (red.sums != nullptr) bn = 64;           // the reduce epilogue is built for 64-column tiles
    else if (g_force_bn) bn = g_force_bn;
    const int ny = (N + bn - 1) / bn;
    // row block 64 instead of 128 when 128-row blocks would not give every CU four workgroups
    const bool small = (int64_t)((M + 127) / 128) * ny * nz < 1024;
    const int bm = g_force_bm ? g_force_bm : (small ? 64 : 128);
    const dim3 grid((M + bm - 1) / bm, ny, nz);
#define MS_GEMM_TILE(BM_, BN_) \
    if (bm == BM_ && bn == BN_) launch_layout<BM_, BN_>(a_f32, b_f32, a_trans, b_trans, c_mode, grid, stream, A, B, C, M, N, K, lda, ldb, ldc, kps, bias, relu, red);
    MS_GEMM_TILE(128, 64) MS_GEMM_TILE(128, 128) MS_GEMM_TILE(128, 192) MS_GEMM_TILE(64, 64) MS_GEMM_TILE(64, 128) MS_GEMM_TILE(64, 192)
#undef MS_GEMM_TILE
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int gemm_bf16_dispatch(const void *A, int a_f32, int a_trans, int64_t lda, const void *B, int b_f32, int b_trans, int64_t ldb, void *C,
                       int c_mode, int64_t ldc, int M, int N, int K, int k_splits, const float *bias, int relu, hipStream_t stream) {
    return gemm_impl(A, a_f32, a_trans, lda, B, b_f32, b_trans, ldb, C, c_mode, ldc, M, N, K, k_splits, bias, relu, GemmRed{}, stream);
}

// C = A B (b_trans form: B is (K, N) in memory), stored as fp32 / bf16, with the BatchNorm backward reduce of the result in the epilogue
int gemm_bf16_bnbwd_dispatch(const void *A, int a_f32, int64_t lda, const void *B, int b_f32, int64_t ldb, void *C, int c_mode, int64_t ldc, int M,
                             int N, int K, const MsBnBwd *bn, hipStream_t stream) {
    if (!bn || !bn->x_pre || !bn->gamma || !bn->beta || !bn->save_mean || !bn->save_rstd || !bn->sums) return MS_ERR_NULL;
    if (c_mode < 0 || c_mode > 1) return MS_ERR_SHAPE;
    if (bn->x_pre_is_f32 || N % 4 != 0 || c_mode != 1) return MS_ERR_UNSUPPORTED;
    if (bn->x_pre_pixel_stride < N || bn->x_pre_pixel_stride % 4 != 0 || (reinterpret_cast<uintptr_t>(bn->x_pre) & 7) != 0) return MS_ERR_STRIDE;
    GemmRed red;
    red.xpre = static_cast<const unsigned short *>(bn->x_pre); red.xps = bn->x_pre_pixel_stride; red.gamma = bn->gamma; red.beta = bn->beta;
    red.mean = bn->save_mean; red.rstd = bn->save_rstd; red.relu = bn->relu; red.sums = bn->sums;
    return gemm_impl(A, a_f32, 0, lda, B, b_f32, 1, ldb, C, c_mode, ldc, M, N, K, 1, nullptr, 0, red, stream);
}

}  // namespace ms
