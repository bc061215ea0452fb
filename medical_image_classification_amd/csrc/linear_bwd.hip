// Both backward products of a token projection from ONE pass over its operands -- the autograd of `in_proj` / `x_proj` / `out_proj`
// (MedMamba.py:284,326,397,469,480) at the early stages, where the token count is huge (B*H*W = 200 704 at stage 0 of MedMamba-T, batch 64)
// and the feature dimensions are small:
//     dx[m][k] = sum_n dy[m][n] W[n][k]            (M x K)
//     dW[n][k] = sum_m dy[m][n] x[m][k]            (N x K, reduction over all tokens)
// As two ms_gemm_bf16 launches each product streams dy (and the weight gradient re-reads it: x_proj at stage 0 moved 455 MB for 189 MB of
// operands, 101 us against 42 for the library's kernel).  Here a workgroup owns slabs of 64 tokens: dy and x of a slab are staged ONCE
// (16-byte pieces, fp32 rounded to bf16 on the way, the next slab's loads in flight during the products), the weight sits in LDS for the
// whole kernel, and both products read the same LDS images -- dx through plain fragments of the dy rows and transposing reads of W,
// dW through transposing reads (ds_read_b64_tr_b16) of dy and x, accumulated in registers over all of the workgroup's slabs and added to
// the fp32 output once at the end (whole rows through LDS: consecutive addresses per atomic instruction).
// bf16 MFMA (v_mfma_f32_16x16x32_bf16), fp32 accumulation: the same arithmetic as the two-launch form.
// Shapes: N <= 16 NT (NT even: the contraction of dx runs in 32-deep steps), K == 16 KT, for the (NT, KT) pairs instantiated below.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include "medscan.h"
#include "gemm_common.h"

namespace ms {
namespace {

constexpr int kSlab = 64;                       // tokens per slab

__device__ __forceinline__ bf16x8 tr_frag8(const unsigned short *img, int pitch, int k0, int row0, int lane) {
    // image [contraction index][row], fragment of rows row0 .. row0 + 15, contraction k0 .. k0 + 31: lane (fr, fq) receives
    // op[row0 + fr][k0 + 8 fq + j], j = 0..7 (two transposing 8-byte reads, as TileStage::frag in gemm.hip)
    const int fr = lane & 15, fq = lane >> 4;
    const unsigned short *b = img + (k0 + fq * 8 + (fr >> 2)) * pitch + row0 + 4 * (fr & 3);
    typedef bf16x4 __attribute__((address_space(3))) *lds_p;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(b));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(b + 4 * pitch));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int NT, int KT>
__global__ void __launch_bounds__(256)
linear_bwd_kernel(const void *__restrict__ dy, int dy_f32, int64_t ld_dy, const void *__restrict__ x, int x_f32, int64_t ld_x,
                  const void *__restrict__ w, int w_f32, void *__restrict__ dx, int dx_bf16, int64_t ld_dx, float *__restrict__ dW,
                  int M, int N, int n_slabs) {
    constexpr int NP = NT * 16, KP = KT * 16;
    constexpr int PDY = NP + 8, PX = KP + 8, PW = KP + 8;                 // LDS pitches (bf16 elements; rows stay 16-byte aligned)
    constexpr int kStage = (kSlab * PDY + kSlab * PX + NP * PW) * 2, kOut = NP * KP * 4;
    constexpr int kMain = kStage > kOut ? kStage : kOut;
    // + the slab's dx tile [64][KP] fp32: the MFMA layout gives a lane 16 bytes of one row -- a wave-level store of it touches sixteen
    // 64-byte fragments 4 K bytes apart (the counters showed the waves stalled on ISSUE for half of the kernel); through LDS every
    // wave-level store is 1 KB of consecutive addresses
    constexpr int kBytes = kMain + kSlab * KP * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[kBytes];
    unsigned short *sDY = reinterpret_cast<unsigned short *>(smem), *sX = sDY + kSlab * PDY, *sW = sX + kSlab * PX;
    float *sDX = reinterpret_cast<float *>(smem + kMain);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;

    // slab staging through registers, one slab ahead.  The pieces stay RAW (two 16-byte halves for fp32, one for bf16) until they are
    // written to LDS one iteration later: rounding them when they are fetched puts a wait for the loads into the fetch and the
    // "next slab in flight during the products" is gone (first version: 21 us per slab)
    constexpr int NPY = (kSlab * (NP / 8) + 255) / 256, NPX = (kSlab * (KP / 8) + 255) / 256;
    uint4 rya[NPY], ryb[NPY], rxa[NPX], rxb[NPX];
    const int ey = dy_f32 ? 4 : 2, ex = x_f32 ? 4 : 2;          // element sizes
    auto fetch = [&](int slab) {
        const int m0 = slab * kSlab;
#pragma unroll
        for (int i = 0; i < NPY; ++i) {
            const int id = tid + 256 * i, row = id / (NP / 8), nc = id % (NP / 8), gm = m0 + row;
            const int nv = (id < kSlab * (NP / 8) && gm < M) ? N - nc * 8 : 0;            // valid elements of the piece (>= 8: whole)
            const char *p = static_cast<const char *>(dy) + ((int64_t)gm * ld_dy + nc * 8) * ey;
            rya[i] = make_uint4(0, 0, 0, 0); ryb[i] = make_uint4(0, 0, 0, 0);
            if (nv >= (dy_f32 ? 4 : 8)) rya[i] = *reinterpret_cast<const uint4 *>(p);      // N % 4 == 0 (fp32) / % 8 == 0 (bf16): host check
            if (dy_f32 && nv >= 8) ryb[i] = *reinterpret_cast<const uint4 *>(p + 16);
        }
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int id = tid + 256 * i, row = id / (KP / 8), kc = id % (KP / 8), gm = m0 + row;
            const bool in = id < kSlab * (KP / 8) && gm < M;
            const char *p = static_cast<const char *>(x) + ((int64_t)gm * ld_x + kc * 8) * ex;
            rxa[i] = make_uint4(0, 0, 0, 0); rxb[i] = make_uint4(0, 0, 0, 0);
            if (in) rxa[i] = *reinterpret_cast<const uint4 *>(p);
            if (in && x_f32) rxb[i] = *reinterpret_cast<const uint4 *>(p + 16);
        }
    };
    auto to_bf16 = [](uint4 a, uint4 b, bool f32) -> bf16x8 {
        if (!f32) return __builtin_bit_cast(bf16x8, a);
        auto f = [](unsigned v) { return __builtin_bit_cast(float, v); };
        return (bf16x8){(short)f2bf(f(a.x)), (short)f2bf(f(a.y)), (short)f2bf(f(a.z)), (short)f2bf(f(a.w)),
                        (short)f2bf(f(b.x)), (short)f2bf(f(b.y)), (short)f2bf(f(b.z)), (short)f2bf(f(b.w))};
    };
    // weight-gradient tiles of this wave: n tiles a = wn, wn + 2, ..; k tiles b = wk, wk + 2, ..
    const int wn = wv & 1, wk = wv >> 1;
    constexpr int NA = (NT + 1) / 2, NB = (KT + 1) / 2;
    f32x4 accw[NA][NB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) accw[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if ((int)blockIdx.x < n_slabs) fetch(blockIdx.x);
    {   // the weight [N][K] -> sW[n][k] once (rows >= N are zero): all of a thread's pieces requested before the first is written (a
        // load -> store loop exposed eight dependent round trips at the start of every workgroup)
        constexpr int NPW = (NP * (KP / 8) + 255) / 256;
        uint4 wa[NPW], wb[NPW];
        const int ew = w_f32 ? 4 : 2;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int id = tid + 256 * i, n = id / (KP / 8), kc = id % (KP / 8);
            const bool in = id < NP * (KP / 8) && n < N;
            const char *p = static_cast<const char *>(w) + ((int64_t)n * KP + kc * 8) * ew;
            wa[i] = make_uint4(0, 0, 0, 0); wb[i] = make_uint4(0, 0, 0, 0);
            if (in) wa[i] = *reinterpret_cast<const uint4 *>(p);
            if (in && w_f32) wb[i] = *reinterpret_cast<const uint4 *>(p + 16);
        }
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int id = tid + 256 * i, n = id / (KP / 8), kc = id % (KP / 8);
            if (id < NP * (KP / 8)) *reinterpret_cast<bf16x8 *>(sW + n * PW + kc * 8) = to_bf16(wa[i], wb[i], w_f32 != 0);
        }
    }
    for (int slab = blockIdx.x; slab < n_slabs; slab += gridDim.x) {
        __syncthreads();                                   // the previous slab's fragments have been read (first trip: sW is written)
#pragma unroll
        for (int i = 0; i < NPY; ++i) {
            const int id = tid + 256 * i, row = id / (NP / 8), nc = id % (NP / 8);
            if (id < kSlab * (NP / 8)) *reinterpret_cast<bf16x8 *>(sDY + row * PDY + nc * 8) = to_bf16(rya[i], ryb[i], dy_f32 != 0);
        }
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int id = tid + 256 * i, row = id / (KP / 8), kc = id % (KP / 8);
            if (id < kSlab * (KP / 8)) *reinterpret_cast<bf16x8 *>(sX + row * PX + kc * 8) = to_bf16(rxa[i], rxb[i], x_f32 != 0);
        }
        __syncthreads();
        if (slab + (int)gridDim.x < n_slabs) fetch(slab + gridDim.x);
        // ---- dx rows 16 wv .. 16 wv + 15 of the slab: contraction over n in 32-deep steps ----
        {
            f32x4 accx[KT];
#pragma unroll
            for (int b = 0; b < KT; ++b) accx[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NP / 32; ++ks) {
                const bf16x8 fa = *reinterpret_cast<const bf16x8 *>(sDY + (wv * 16 + fr) * PDY + ks * 32 + fq * 8);
#pragma unroll
                for (int b = 0; b < KT; ++b) {
                    const bf16x8 fb = tr_frag8(sW, PW, ks * 32, b * 16, lane);
                    accx[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, accx[b], 0, 0, 0);
                }
            }
            // acc[r] = dx[16 wv + fr][16 b + 4 fq + r] -> sDX (read back in row order after the weight-gradient steps below)
#pragma unroll
            for (int b = 0; b < KT; ++b)
                *reinterpret_cast<float4 *>(sDX + (wv * 16 + fr) * KP + b * 16 + fq * 4) = make_float4(accx[b][0], accx[b][1], accx[b][2], accx[b][3]);
        }
        // ---- dW += dy^T x over the slab's 64 tokens (two 32-deep steps) ----
#pragma unroll
        for (int ks = 0; ks < kSlab / 32; ++ks) {
            bf16x8 fa[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a)
                if (wn + 2 * a < NT) fa[a] = tr_frag8(sDY, PDY, ks * 32, (wn + 2 * a) * 16, lane);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (wk + 2 * b < KT) {
                    const bf16x8 fb = tr_frag8(sX, PX, ks * 32, (wk + 2 * b) * 16, lane);
#pragma unroll
                    for (int a = 0; a < NA; ++a)
                        if (wn + 2 * a < NT) accw[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa[a], accw[a][b], 0, 0, 0);
                }
            }
        }
        // ---- the slab's dx rows, 16 bytes per lane in row order (the barrier at the top of the next trip orders the next writes) ----
        __syncthreads();
        {
            const int m0 = slab * kSlab;
            constexpr int kQ = kSlab * KP / 4;                   // float4 pieces of the tile
#pragma unroll
            for (int i = 0; i < (kQ + 255) / 256; ++i) {
                const int q = tid + 256 * i, row = q / (KP / 4), c4 = q % (KP / 4);
                if (q < kQ && m0 + row < M) {
                    const float4 v = *reinterpret_cast<const float4 *>(sDX + row * KP + c4 * 4);
                    if (dx_bf16) {
                        uint2 pk;
                        pk.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
                        pk.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
                        *reinterpret_cast<uint2 *>(static_cast<unsigned short *>(dx) + (int64_t)(m0 + row) * ld_dx + c4 * 4) = pk;
                    } else {
                        *reinterpret_cast<float4 *>(static_cast<float *>(dx) + (int64_t)(m0 + row) * ld_dx + c4 * 4) = v;
                    }
                }
            }
        }
    }
    // ---- the workgroup's dW tile -> LDS rows -> fp32 atomics on consecutive addresses (acc[r] = dW[16 a' + fr][16 b' + 4 fq + r]) ----
    __syncthreads();
    float *sOut = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (wn + 2 * a < NT && wk + 2 * b < KT)
                *reinterpret_cast<float4 *>(sOut + ((wn + 2 * a) * 16 + fr) * KP + (wk + 2 * b) * 16 + fq * 4) =
                    make_float4(accw[a][b][0], accw[a][b][1], accw[a][b][2], accw[a][b][3]);
    __syncthreads();
    const int tot = N * KP;                                 // dW is (N, K) contiguous, K == KP
    const int rot = (int)(((int64_t)blockIdx.x * 1024) % tot);   // concurrent workgroups start at different rows
    for (int e0 = tid; e0 < tot; e0 += 256) {
        int e = e0 + rot; if (e >= tot) e -= tot;
        atomicAdd(dW + e, sOut[e]);
    }
}

struct Shape { int nt, kt; };
constexpr Shape kShapes[] = {{10, 6}, {12, 3}, {4, 6}, {10, 8}, {16, 4}, {4, 8}};

bool pick(int N, int K, int &nt, int &kt) {
    if (K % 16 != 0 || N <= 0) return false;
    int best = -1;
    for (int i = 0; i < (int)(sizeof(kShapes) / sizeof(kShapes[0])); ++i)
        if (kShapes[i].kt * 16 == K && kShapes[i].nt * 16 >= N && (best < 0 || kShapes[i].nt < kShapes[best].nt)) best = i;
    if (best < 0) return false;
    nt = kShapes[best].nt; kt = kShapes[best].kt;
    return true;
}

}  // namespace

int linear_bwd_ok(int N, int K) { int a, b; return pick(N, K, a, b) ? 1 : 0; }

int linear_bwd_dispatch(const void *dy, int dy_f32, int64_t ld_dy, const void *x, int x_f32, int64_t ld_x, const void *w, int w_f32, void *dx,
                        int dx_bf16, int64_t ld_dx, float *dW, int M, int N, int K, hipStream_t s) {
    if (!dy || !x || !w || !dx || !dW) return MS_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0) return MS_ERR_SHAPE;
    int nt, kt;
    if (!pick(N, K, nt, kt)) return MS_ERR_UNSUPPORTED;
    // 16-byte pieces: row strides in units of 8 bf16 / 4 fp32, 16-byte aligned bases; dx rows hold 16-byte (fp32) / 8-byte (bf16) stores
    const int64_t gy = dy_f32 ? 4 : 8, gx = x_f32 ? 4 : 8;
    auto mis = [](const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) != 0; };
    if (N % (int)gy) return MS_ERR_UNSUPPORTED;                    // the last piece of a dy row must end on a 16-byte half
    if (ld_dy % gy || ld_x % gx || ld_dx % 4 || mis(dy, 16) || mis(x, 16) || mis(w, 16) || mis(dx, dx_bf16 ? 8 : 16) || ld_dy < N || ld_x < K ||
        ld_dx < K)
        return MS_ERR_STRIDE;
    const int n_slabs = (M + kSlab - 1) / kSlab;
    static const int wg_env = [] { const char *e = getenv("MEDSCAN_LINEAR_BWD_WGS"); return e ? atoi(e) : 0; }();
    // persistent workgroups, ONE per CU: every workgroup ends with N x K atomics, and 512 / 768 / 1024 workgroups measured 104 / 111 /
    // 116 us against 101 at x_proj's stage-0 shape (in_proj: 68 / 76 / 89 against 57)
    const int wgs = wg_env > 0 ? wg_env : 256;
    const int grid = n_slabs < wgs ? n_slabs : wgs;
#define MS_LB(NTv, KTv)                                                                                                                  \
    if (nt == NTv && kt == KTv)                                                                                                          \
        hipLaunchKernelGGL((linear_bwd_kernel<NTv, KTv>), dim3((unsigned)grid), dim3(256), 0, s, dy, dy_f32, ld_dy, x, x_f32, ld_x, w, w_f32, dx, \
                           dx_bf16, ld_dx, dW, M, N, n_slabs)
    MS_LB(10, 6); MS_LB(12, 3); MS_LB(4, 6); MS_LB(10, 8); MS_LB(16, 4); MS_LB(4, 8);
#undef MS_LB
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
