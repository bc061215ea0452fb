// The SS2D projections in the reference's OWN precision (fp32: the reference never uses autocast, train.py:57-77) on the gfx950 matrix
// cores: in_proj / x_proj / out_proj of MedMamba.py:284,326,397,469,480 and their input / weight gradients with
// v_mfma_f32_16x16x4_f32 -- exact fp32 products and fp32 accumulation (the instruction is an fmaf chain, bit for bit), 64 FLOP per
// clock and SIMD = 157 TF: these token-matrix products (M = 3 136 .. 200 704 rows, 48 .. 1 536 features) are bound by streaming
// the activation once, as their bf16 siblings in gemm.hip are.  Same interface, tiling and epilogues as gemm.hip:
//
//   C[i][j] (+)= sum_k Aop[i][k] * Bop[j][k]       Aop = a_trans ? A^T : A,  Bop = b_trans ? B^T : B, all fp32
//   forward      y  = x  @ W^T : plain / plain
//   input grad   dx = dy @ W   : plain / transposed
//   weight grad  dW = dy^T @ x : transposed / transposed, split over blockIdx.z, fp32 atomics (c_mode 2, or 3 into C^T)
//
// Tile: BM x BN outputs per 256-thread workgroup, 64-deep k-steps staged through LDS (32-deep ones measured 10-30 % slower: two barriers per step) with 16-byte pieces in memory order
// (plain operands [row][k] at a pitch of 68 floats; transposed ones [k][row] at pitch rows + 4).
// One MFMA consumes ONE k per lane (k = lane / 16); four consecutive MFMAs of a lane take k = 4 q + s (q = lane / 16, s = 0..3)
// instead of 4 s + q -- both operands use the same assignment, so the sum over the 16 k is unchanged -- which makes a plain
// operand's fragment for 16 k ONE ds_read_b128 (a transposed one: four ds_read_b32, conflict-free at pitch = 4 mod 8).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "medscan.h"

namespace ms {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef MS_F32_BK
#define MS_F32_BK 64
#endif
constexpr int kBKf = MS_F32_BK, kPf = kBKf + 4;        // floats

__device__ __forceinline__ float4 load_piece4(const float *base, int64_t off, int nvalid) {
    if (nvalid >= 4) return *reinterpret_cast<const float4 *>(base + off);
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid > 0) r.x = base[off];
    if (nvalid > 1) r.y = base[off + 1];
    if (nvalid > 2) r.z = base[off + 2];
    return r;
}

template <int ROWS, bool TR>
struct TileStageF {
    static constexpr int NP = ROWS * kBKf / 4 / 256;     // float4 pieces per thread
    static constexpr int kPT = ROWS + 4;                 // pitch of the transposed image
    static constexpr int kLds = TR ? kBKf * kPT : ROWS * kPf;
    float4 r[NP];
    __device__ __forceinline__ void fetch(const float *X, int64_t ld, int row0, int k0, int nrows, int K, int tid) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int id = tid + 256 * i;
            if (!TR) {
                const int row = id / (kBKf / 4), kc = id % (kBKf / 4);
                const int gr = row0 + row, gk = k0 + kc * 4;
                r[i] = load_piece4(X, (int64_t)gr * ld + gk, gr < nrows ? min(4, max(0, K - gk)) : 0);
            } else {
                const int k = id / (ROWS / 4), rc = id % (ROWS / 4);
                const int gk = k0 + k, gr = row0 + rc * 4;
                r[i] = load_piece4(X, (int64_t)gk * ld + gr, gk < K ? min(4, max(0, nrows - gr)) : 0);
            }
        }
    }
    __device__ __forceinline__ void put(float *s, int tid) const {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int id = tid + 256 * i;
            if (!TR) {
                const int row = id / (kBKf / 4), kc = id % (kBKf / 4);
                *reinterpret_cast<float4 *>(s + row * kPf + kc * 4) = r[i];
            } else {
                const int k = id / (ROWS / 4), rc = id % (ROWS / 4);
                *reinterpret_cast<float4 *>(s + k * kPT + rc * 4) = r[i];
            }
        }
    }
    // operand fragment of tile rows [16 t, 16 t + 16), 16-deep k-group ks: lane (fr, fq) gets op(X)[16 t + fr][16 ks + 4 fq + s], s = 0..3
    static __device__ __forceinline__ f32x4 frag(const float *s, int t, int ks, int lane) {
        const int fr = lane & 15, fq = lane >> 4;
        if (!TR) {
            const float4 v = *reinterpret_cast<const float4 *>(s + (t * 16 + fr) * kPf + ks * 16 + fq * 4);
            return (f32x4){v.x, v.y, v.z, v.w};
        }
        const float *b = s + (ks * 16 + fq * 4) * kPT + t * 16 + fr;
        return (f32x4){b[0], b[kPT], b[2 * kPT], b[3 * kPT]};
    }
};

// CMODE: 0 = store, 2 = atomic add (split-K partial), 3 = atomic add into C^T
template <int BM, int BN, bool ATR, bool BTR, int CMODE>
__global__ void __launch_bounds__(256)
gemm_f32_kernel(const float *A, const float *B, float *C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int k_per_split,
                const float *__restrict__ bias, int relu) {
    constexpr int TNT = BN / 16, MT = BM / 64;
    using TA = TileStageF<BM, ATR>;
    using TB = TileStageF<BN, BTR>;
    __shared__ __attribute__((aligned(16))) float sA[TA::kLds];
    __shared__ __attribute__((aligned(16))) float sB[TB::kLds];
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

    TA ta[2];
    TB tb[2];
    ta[0].fetch(A, lda, m0, kbeg, M, kend, tid);
    tb[0].fetch(B, ldb, n0, kbeg, N, kend, tid);
    if (kbeg + kBKf < kend) {
        ta[1].fetch(A, lda, m0, kbeg + kBKf, M, kend, tid);
        tb[1].fetch(B, ldb, n0, kbeg + kBKf, N, kend, tid);
    }
    auto step = [&](TA &sa_, TB &sb_, int k0) {
        __syncthreads();
        sa_.put(sA, tid);
        sb_.put(sB, tid);
        __syncthreads();
        if (k0 + 2 * kBKf < kend) {
            sa_.fetch(A, lda, m0, k0 + 2 * kBKf, M, kend, tid);
            sb_.fetch(B, ldb, n0, k0 + 2 * kBKf, N, kend, tid);
        }
#pragma unroll
        for (int ks = 0; ks < kBKf / 16; ++ks) {
            f32x4 fa[MT];
#pragma unroll
            for (int b = 0; b < MT; ++b) fa[b] = TA::frag(sA, w * MT + b, ks, lane);
#pragma unroll
            for (int a = 0; a < TNT; ++a) {
                const f32x4 fb = TB::frag(sB, a, ks, lane);
#pragma unroll
                for (int b = 0; b < MT; ++b)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[s], fa[b][s], acc[a][b], 0, 0, 0);
            }
        }
    };
    const int fr = lane & 15, fq = lane >> 4;
    for (int k0 = kbeg; k0 < kend; k0 += 2 * kBKf) {
        step(ta[0], tb[0], k0);
        if (k0 + kBKf < kend) step(ta[1], tb[1], k0 + kBKf);
    }
    if constexpr (CMODE >= 2) {
        // the workgroup's tile goes through LDS so that a wave's atomics cover 64 consecutive addresses, starting at a position that
        // depends on the k-slice (concurrent slices of one output tile hit different addresses)
#pragma unroll
        for (int b = 0; b < MT; ++b)
#pragma unroll
            for (int a = 0; a < TNT; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) sCt[(w * (16 * MT) + b * 16 + fr) * kCP + a * 16 + fq * 4 + r] = acc[a][b][r];
        __syncthreads();
        constexpr int kTot = BM * BN;
        const int rot = (int)(blockIdx.z % (kTot / 256)) * 256;
        for (int i = 0; i < kTot / 256; ++i) {
            const int e = (i * 256 + tid + rot) % kTot;
            int ml, nl;
            if (CMODE == 2) { nl = e % BN; ml = e / BN; }
            else            { ml = e % BM; nl = e / BM; }
            const int m = m0 + ml, n = n0 + nl;
            if (m < M && n < N) atomicAdd(CMODE == 2 ? C + (int64_t)m * ldc + n : C + (int64_t)n * ldc + m, sCt[ml * kCP + nl]);
        }
        return;
    }
    // D[i][j]: i = n within the tile (row 4 fq + r of the accumulator), j = m within the tile (column fr)
#pragma unroll
    for (int b = 0; b < MT; ++b) {
        const int m = m0 + w * (16 * MT) + b * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int a = 0; a < TNT; ++a) {
            const int n = n0 + a * 16 + fq * 4;
            if (n >= N) continue;
            f32x4 v = acc[a][b];
            if (bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < N) v[r] += bias[n + r];
            }
            if (relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
            }
            float *c = C + (int64_t)m * ldc + n;
            if (n + 4 <= N && (ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0) *reinterpret_cast<float4 *>(c) = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < N) c[r] = v[r];
            }
        }
    }
}

template <int BM, int BN>
static void launch_f32(bool atr, bool btr, int c_mode, dim3 grid, hipStream_t s, const float *A, const float *B, float *C, int M, int N,
                       int K, int64_t lda, int64_t ldb, int64_t ldc, int kps, const float *bias, int relu) {
    if (!atr && !btr) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, false, 0>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, kps, bias, relu);
    else if (!atr && btr) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, true, 0>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, kps, bias, relu);
    else if constexpr (BN == 64) {       // the weight gradient always runs 64-column blocks
        if (c_mode == 2) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, true, 2>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, kps, bias, relu);
        else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, true, 3>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, kps, bias, relu);
    }
}

int gemm_f32_dispatch(const float *A, int a_trans, int64_t lda, const float *B, int b_trans, int64_t ldb, float *C, int c_mode, int64_t ldc,
                      int M, int N, int K, int k_splits, const float *bias, int relu, hipStream_t stream) {
    if (!A || !B || !C) return MS_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0 || k_splits < 1 || (c_mode != 0 && c_mode != 2 && c_mode != 3)) return MS_ERR_SHAPE;
    if ((k_splits > 1 && c_mode == 0) || ((bias || relu) && c_mode != 0)) return MS_ERR_SHAPE;
    // built: forward (plain, plain) and input gradient (plain, transposed) in the store mode; weight gradient (transposed, transposed)
    // in the accumulating modes
    if (a_trans ? !(b_trans && c_mode >= 2) : c_mode != 0) return MS_ERR_UNSUPPORTED;
    if (lda % 4 || ldb % 4 || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return MS_ERR_STRIDE;
    int kps = (K + k_splits - 1) / k_splits;
    kps = (kps + kBKf - 1) / kBKf * kBKf;
    const int nz = (K + kps - 1) / kps;
    // tile choice as in gemm.hip: 64-column blocks (more workgroups in flight), except for a large plain A operand, which is read in as few
    // column passes as possible (x_proj at stage 0 streams 77 MB of activations for 140 output columns)
    const int64_t a_bytes = (int64_t)M * K * 4;
    int bn = 64;
    if (!a_trans && a_bytes >= (48ll << 20) && N > 64) bn = N <= 128 ? 128 : 192;
    const int ny = (N + bn - 1) / bn;
    const bool small = (int64_t)((M + 127) / 128) * ny * nz < 1024;
    const int bm = small ? 64 : 128;
    const dim3 grid((M + bm - 1) / bm, ny, nz);
#define MS_F32_TILE(BM_, BN_) \
    if (bm == BM_ && bn == BN_) launch_f32<BM_, BN_>(a_trans, b_trans, c_mode, grid, stream, A, B, C, M, N, K, lda, ldb, ldc, kps, bias, relu);
    MS_F32_TILE(128, 64) MS_F32_TILE(64, 64) MS_F32_TILE(128, 128) MS_F32_TILE(64, 128) MS_F32_TILE(128, 192) MS_F32_TILE(64, 192)
#undef MS_F32_TILE
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
