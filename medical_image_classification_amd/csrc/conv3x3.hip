// Dense 3x3 convolution (stride 1, padding 1, groups 1, no bias) on channels_last bf16 activations for gfx950 -- the two
// `nn.Conv2d(dim/2, dim/2, 3, padding=1)` of SS_Conv_SSM's conv branch (MedMamba.py:518-523) at 48 / 96 / 192 / 384 channels.
// A direct (implicit-GEMM) MFMA kernel: out[p, co] = sum_{tap, ci} x[p + tap, ci] * w[co, tap, ci], bf16 operands, fp32
// accumulation.  The same kernel gives the input gradient (dx = conv3x3(dy, w') with the spatially flipped, in/out-transposed
// weight w'[ci, 8 - tap, co], which the weight-copy kernel writes next to the forward copy).
//   workgroup = 4 waves = an 8 x 16 pixel tile of one image x a block of 48 output channels; wave = 2 image rows (32 pixels)
//   LDS: the tile's 10 x 18 halo of input pixels for a slice of 32 input channels, and the weights of that channel slice for
//        all 9 taps and the workgroup's output channels
//   MFMA: v_mfma_f32_16x16x16_bf16 -- K = 16 input channels per instruction divides every channel count of the model
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "medscan.h"

namespace ms {

typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int kTH = 8, kTW = 16;                 // output tile (pixels)
constexpr int kHH = kTH + 2, kHW = kTW + 2;      // halo tile
constexpr int kKS = 32;                          // input channels per LDS slice (39 KB of LDS per workgroup)
// LDS images: [row][32 channels], 64-byte rows WITHOUT padding; the row's four 16-byte pieces are stored at piece ^ ((row >> 1) & 3).
// gfx950 serves a ds_read_b128 in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- i.e. an MFMA
// fragment read (lane = row fr, piece fq) puts rows 0-3 / 12-15 with piece q and rows 4-11 with piece q + 1 into one group.  The padded
// 80-byte pitch of rounds 1-2 was laid out for groups of 16 CONSECUTIVE lanes: under the real grouping every fragment read took 8 LDS
// cycles instead of 4 (SQ_LDS_BANK_CONFLICT = 39 % of the kernel's LDS cycles).  With this swizzle every fragment read of the kernel --
// any halo row offset, any tap -- and every staging write is conflict-free (tools/lds_swizzle_search.py enumerates the layouts).
constexpr int kXP = kKS;                         // halo pixel pitch (bf16)
constexpr int kWP = kKS;                         // weight row pitch (bf16)
__device__ __forceinline__ int swz(int row, int piece) { return (piece ^ ((row >> 1) & 3)) * 8; }   // bf16 offset of a piece within its row
}

typedef short bf16x8 __attribute__((ext_vector_type(8)));

// ---- training-mode BatchNorm folded into the convolution on both sides (MedMamba.py:518-524: conv3x3 -> BatchNorm2d -> ReLU -> conv3x3 -> ...) --
// STATS (producer side): the epilogue accumulates, per output channel, sum (y - p) and sum (y - p)^2 of the bf16-ROUNDED outputs it stores
//   (p = running_mean - conv bias: a pivot near the batch mean once training runs, any value is valid) into one of kBnRep replica rows of
//   `sums` with fp32 atomics (replicas: 1 792 workgroups adding to one address serialise at the memory side); the statistics and finalize
//   launches of that BatchNorm disappear.
// BNIN (consumer side): the input tensor is the PRE-BatchNorm activation; every workgroup derives scale / shift of all input channels
//   from the replica rows while its first slice's loads are in flight, applies relu(x * scale + shift) to the halo pieces on their way
//   from registers to LDS (out-of-image pixels stay zero: the convolution pads the NORMALISED activation), and the workgroups of the
//   first output-channel block write the normalised interior of their tile to `xhat` (what the backward's weight gradient reads), so the
//   apply pass (a read and a write of the activation) disappears too.  Workgroup (0, 0) writes the batch mean / rstd for the backward and
//   updates the running statistics (torch semantics: biased variance to normalise, unbiased into running_var).
constexpr int kBnRep = MS_BN_REPLICAS;
constexpr int kBnMaxC = 512;                     // input channels whose scale / shift fit the LDS tables
struct BnFoldDev {
    float *sums;                                 // [kBnRep][2][C] sums, then [C] pivots
    const float *gamma, *beta, *shift;
    float *running_mean, *running_var;
    long long *nbt;
    float *save_mean, *save_rstd;
    float momentum, eps, inv_n, unbias;          // inv_n = 1 / pixels, unbias = n / (n - 1)
};
// BRED (the input-gradient launch, dx = conv3x3(dy, w')): the BatchNorm BEHIND this convolution's input (the one whose output the forward
//   convolved) needs sum dy' and sum dy' * xhat over all pixels, dy' = dx * [its ReLU passed]: the epilogue has dx in registers, loads the
//   pre-normalisation activation of the same pixels, and accumulates both sums into replica rows -- the bn_bwd_reduce pass (a read of x
//   and of dx) and its finalize launch disappear; ms_bn_bwd_apply_sums_nhwc reads the rows.
struct BnBwdDev {
    const void *xpre; int xpre_f32; int64_t xps;
    const float *gamma, *beta, *mean, *rstd;
    int relu;
    float *sums;                                 // [kBnRep][2][C]: sum dy' (dbeta), sum dy' * xhat (dgamma)
};
__device__ __forceinline__ float bf_lo(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf_hi(unsigned v) { return __builtin_bit_cast(float, v & 0xFFFF0000u); }
__device__ __forceinline__ unsigned bf_pack(float a, float b) {
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)b) << 16);
}
template <int CTRL> __device__ __forceinline__ float row_ror_add(float x) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(unsigned, x), CTRL, 0xF, 0xF, true));
}
#ifndef MS_CONV_ABL
#define MS_CONV_ABL 0                    // timing ablations (wrong results): 1 no weight loads / 2 no halo loads / 3 neither after the first slice, 4 no products, 5 no output stores, 6 first slice only
#endif
#ifndef MS_CONV_PF
#define MS_CONV_PF 1                     // slices of prefetch distance (register sets)
#endif
#ifndef MS_CONV_WAVES
#define MS_CONV_WAVES 3                  // 50 KB of LDS: three workgroups per CU
#endif
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifdef MS_CONV_CLOCK
__device__ long long g_conv_clock[8192 * 32];      // per workgroup: s_memtime stamps (diagnostic build only)
#define MS_STAMP(i) do { if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 8192) g_conv_clock[blockIdx.x * 32 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define MS_STAMP(i) do {} while (0)
#endif
constexpr unsigned kOob = 0x80000000u;   // a byte offset beyond every buffer: the load returns zeros

// NB = output-channel tiles of 16 per workgroup (3: 48 channels).
// Staging is software-pipelined through registers: ALL 16-byte loads of a channel slice (halo + weights: 10 per thread) are
// issued back to back into registers, and the NEXT slice's loads are in flight while the current slice is multiplied.
// The loads are raw buffer loads (one descriptor for the image, one for the weights): every thread's byte offsets -- pixel /
// weight row and 16-byte piece -- are worked out ONCE before the slice loop, the slice's channel offset rides in the scalar
// offset, and everything out of the picture (padding ring, rows past the tile, channels past Ci in a 16-channel tail slice) is
// an out-of-range offset that the hardware answers with zeros.  (Until round 3 every slice recomputed the addresses and
// wrapped each load in its own exec-mask branch: ~350 vector + 60 scalar-branch instructions per slice beside 54 MFMAs, and
// the rocprofv3 counters showed the kernel bound by exactly that: matrix cores 15 % busy, LDS 23 %, waves issuing or parked.)
// Every slice is multiplied with v_mfma_f32_16x16x32_bf16; a 16-channel tail (Ci = 48) has zeros in the upper half of both
// operands (the K = 16 instruction occupies the matrix core for the same 16 cycles, so a separate path buys nothing).
template <int NB, bool BNIN = false, bool STATS = false, bool BRED = false, int PF = MS_CONV_PF>
// (NB = 4, MedMamba-B's 64-channel blocks: 61 KB of LDS and 3 x 16 accumulator registers more -- two workgroups per CU is what fits)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NB >= 4 || PF >= 2 ? 2 : MS_CONV_WAVES, NB >= 4 || PF >= 2 ? 2 : MS_CONV_WAVES)))
conv3x3_nhwc_kernel(const unsigned short *__restrict__ x, const unsigned short *__restrict__ w, unsigned short *__restrict__ y,
                    int H, int W, int Ci, int Co, int tiles_w, int tiles_per_img, BnFoldDev bin, unsigned short *__restrict__ xhat,
                    BnFoldDev bout, BnBwdDev bred) {
    constexpr int kRowsX = (kHH * kHW + 15) / 16 * 16;                  // halo pixels padded to whole groups of 16 rows (192)
    constexpr int kRowsW = (9 * NB * 16 + 63) / 64 * 64;                // weight rows padded to whole passes of the workgroup (448 / 576)
    constexpr int kNX = kRowsX * 4 / 256, kNW = kRowsW * 4 / 256;       // 16-byte pieces per thread: halo (3) and weights (7 / 9)
    static_assert(kRowsX * 4 % 256 == 0 && kRowsW * 4 % 256 == 0, "whole passes");
    __shared__ __attribute__((aligned(16))) unsigned short sX[kRowsX * kXP];
    __shared__ __attribute__((aligned(16))) unsigned short sW[kRowsW * kWP];      // (whole passes: the staging writes need no guard)
    __shared__ __attribute__((aligned(16))) float sScale[BNIN ? kBnMaxC : 4], sShift[BNIN ? kBnMaxC : 4];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // (scalar: the wave-uniform guards below become scalar branches)
    const int tile = blockIdx.x, img = tile / tiles_per_img, tt = tile - img * tiles_per_img;
    const int h0 = (tt / tiles_w) * kTH, w0 = (tt % tiles_w) * kTW;
    const int co0 = blockIdx.y * (NB * 16);
    const int fr = lane & 15, fq = lane >> 4;

    f32x4 acc[2][NB];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // thread -> (pixel or weight row, 16-byte piece): within every 64 consecutive work items the ROW is the fast index (16 rows x 4
    // pieces), so that the 16 lanes of an LDS write group hold the same piece of 16 consecutive rows; a wave-level load covers
    // 16 rows x 64 bytes.  Work item tid + 256 i: row = (wave + 4 i) * 16 + (lane & 15), piece = lane >> 4 -- the piece is the
    // thread's own for all of its loads.
    const int64_t img_elems = (int64_t)H * W * Ci;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(x + (int64_t)img * img_elems), 0, (int)(img_elems * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(w), 0, (int)((int64_t)Co * 9 * Ci * 2), 0x00020000);
    unsigned offx[kNX], offw[kNW];
    unsigned inner = 0;                                                  // bit i: halo piece i is one of the tile's own pixels (xhat)
#pragma unroll
    for (int i = 0; i < kNX; ++i) {
        const int pix = (wv + 4 * i) * 16 + fr, ph = pix / kHW, pw = pix - ph * kHW, hh = h0 - 1 + ph, ww = w0 - 1 + pw;
        const bool ok = pix < kHH * kHW && hh >= 0 && hh < H && ww >= 0 && ww < W;
        offx[i] = ok ? (unsigned)(((hh * W + ww) * Ci + fq * 8) * 2) : kOob;
        if (ok && ph >= 1 && ph <= kTH && pw >= 1 && pw <= kTW) inner |= 1u << i;
    }
#pragma unroll
    for (int i = 0; i < kNW; ++i) {
        const int row = (wv + 4 * i) * 16 + fr, tap = row / (NB * 16), col = row - tap * (NB * 16);
        offw[i] = (row < 9 * NB * 16 && co0 + col < Co) ? (unsigned)((((co0 + col) * 9 + tap) * Ci + fq * 8) * 2) : kOob;
    }
    u32x4 rxa[kNX], rwa[kNW];
    u32x4 rxb[PF >= 2 ? kNX : 1], rwb[PF >= 2 ? kNW : 1];      // PF = 2: the second register set (two slices of prefetch distance)
    auto fetch = [&](auto &rxa, auto &rwa, int k0) {
        const unsigned kill = (k0 + fq * 8 >= Ci) ? kOob : 0u;           // the upper pieces of a 16-channel tail slice
#pragma unroll
        for (int i = 0; i < kNX; ++i) {
#if MS_CONV_ABL == 2 || MS_CONV_ABL == 3
            if (k0 > 0) break;
#endif
            rxa[i] = __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)(offx[i] | kill), k0 * 2, 0);
        }
#pragma unroll
        for (int i = 0; i < kNW; ++i) {
#if MS_CONV_ABL == 1 || MS_CONV_ABL == 3
            if (k0 > 0) break;
#endif
            rwa[i] = __builtin_amdgcn_raw_buffer_load_b128(rsw, (int)(offw[i] | kill), k0 * 2, 0);
        }
    };
    auto put = [&](const auto &rxa, const auto &rwa, int k0) {
#pragma unroll
        for (int i = 0; i < kNX; ++i) {
            u32x4 v = rxa[i];
            if constexpr (BNIN) {
                const int c8 = k0 + fq * 8;
                const bool ok = (int)offx[i] >= 0 && c8 < Ci;
                const float4 s0 = *reinterpret_cast<const float4 *>(sScale + c8), s1 = *reinterpret_cast<const float4 *>(sScale + c8 + 4);
                const float4 t0 = *reinterpret_cast<const float4 *>(sShift + c8), t1 = *reinterpret_cast<const float4 *>(sShift + c8 + 4);
                v.x = bf_pack(fmaxf(fmaf(bf_lo(v.x), s0.x, t0.x), 0.f), fmaxf(fmaf(bf_hi(v.x), s0.y, t0.y), 0.f));
                v.y = bf_pack(fmaxf(fmaf(bf_lo(v.y), s0.z, t0.z), 0.f), fmaxf(fmaf(bf_hi(v.y), s0.w, t0.w), 0.f));
                v.z = bf_pack(fmaxf(fmaf(bf_lo(v.z), s1.x, t1.x), 0.f), fmaxf(fmaf(bf_hi(v.z), s1.y, t1.y), 0.f));
                v.w = bf_pack(fmaxf(fmaf(bf_lo(v.w), s1.z, t1.z), 0.f), fmaxf(fmaf(bf_hi(v.w), s1.w, t1.w), 0.f));
                if (!ok) v = (u32x4){0u, 0u, 0u, 0u};
                // the tile's own pixels (not the halo ring), once per tile: the first output-channel block's workgroup
                if (xhat != nullptr && blockIdx.y == 0 && ok && ((inner >> i) & 1u))
                    *reinterpret_cast<u32x4 *>(xhat + (int64_t)img * img_elems + (offx[i] >> 1) + k0) = v;
            }
            *reinterpret_cast<u32x4 *>(sX + ((wv + 4 * i) * 16 + fr) * kXP + swz(fr, fq)) = v;
        }
#pragma unroll
        for (int i = 0; i < kNW; ++i) *reinterpret_cast<u32x4 *>(sW + ((wv + 4 * i) * 16 + fr) * kWP + swz(fr, fq)) = rwa[i];
    };
    // One slice's products, fragment reads one tap ahead of the MFMAs that use them.  Taps run column-major (dx outer): the wave's two
    // output rows need halo rows 0..3 of a column offset, two of them per tap, so every tap after the first of a column reads ONE new
    // pixel fragment (12 per slice instead of 18) plus its NB weight fragments.  The scheduling barriers keep the next tap's ds_reads
    // in front of this tap's MFMAs (left alone, the compiler sinks every read to its first use and the wave eats the LDS latency
    // ~20 times per slice: 1.9 of the 2.8 us a slice took at stage 2 with the loads ablated away).
    auto mac = [&]() {
        bf16x8 A[4], B[2][NB];
        auto lda = [&](int r, int dx) {
            const int row = (wv * 2 + r) * kHW + fr + dx;
            A[r] = *reinterpret_cast<const bf16x8 *>(sX + row * kXP + swz(row, fq));
        };
        auto ldb = [&](int buf, int tap) {
#pragma unroll
            for (int n = 0; n < NB; ++n) B[buf][n] = *reinterpret_cast<const bf16x8 *>(sW + ((tap * NB + n) * 16 + fr) * kWP + swz(fr, fq));
        };
        lda(0, 0); lda(1, 0); ldb(0, 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dx = t / 3, dy = t % 3;
            if (t + 1 < 9) {
                const int dxn = (t + 1) / 3, dyn = (t + 1) % 3;
                if (dyn == 0) { lda(0, dxn); lda(1, dxn); } else lda(dyn + 1, dxn);
                ldb((t + 1) & 1, dyn * 3 + dxn);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[t & 1][n], A[dy + m], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    MS_STAMP(0);
    fetch(rxa, rwa, 0);
    // PF = 2 (dispatched for Ci % 64 == 0 only: an even number of slices): every fetch and every slice below is unconditional -- a fetch
    // past Ci is all out-of-range offsets -- so that the compiler can COUNT the loads in flight: with a conditional fetch it has to
    // assume the younger set absent and waits for all of it (s_waitcnt vmcnt(9..0) instead of vmcnt(19..10)) before the first LDS write.
    if constexpr (PF >= 2) { __builtin_amdgcn_sched_barrier(0); fetch(rxb, rwb, kKS); }                    // (in this order: the first put waits for set a only)
    MS_STAMP(1);
    if constexpr (BNIN) {
        // scale / shift of every input channel from the replica rows (the first slice's loads are in flight meanwhile)
        for (int c = tid; c < Ci; c += 256) {
            float a = 0.f, b = 0.f;
#pragma unroll 4
            for (int r = 0; r < kBnRep; ++r) { a += bin.sums[(r * 2) * Ci + c]; b += bin.sums[(r * 2 + 1) * Ci + c]; }
            const float m1 = a * bin.inv_n, var = fmaxf(fmaf(-m1, m1, b * bin.inv_n), 0.f);
            const float mean = bin.sums[kBnRep * 2 * Ci + c] + m1, rstd = rsqrtf(var + bin.eps);
            const float sc = bin.gamma[c] * rstd;
            sScale[c] = sc; sShift[c] = fmaf(-mean, sc, bin.beta[c]);
            if (blockIdx.x == 0 && blockIdx.y == 0) {
                bin.save_mean[c] = mean; bin.save_rstd[c] = rstd;
                const float sh = bin.shift ? bin.shift[c] : 0.f;             // the layer's logical input is x + shift (the conv bias)
                bin.running_mean[c] = (1.f - bin.momentum) * bin.running_mean[c] + bin.momentum * (mean + sh);
                bin.running_var[c] = (1.f - bin.momentum) * bin.running_var[c] + bin.momentum * var * bin.unbias;
            }
        }
        if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && bin.nbt) *bin.nbt += 1;
        // (a 16-channel tail reads the tables up to the next multiple of 32: the entries past Ci are never used -- `ok` is false there)
    }
    // one slice: barrier (the previous slice's fragments have been read; first trip: the tables are written), registers -> LDS, barrier,
    // the loads of the slice PF ahead into the freed register set, the products
    auto slice = [&](auto &rx, auto &rw, int k0) {
        __syncthreads();
        MS_STAMP(2 + (k0 / kKS) * 4);
        put(rx, rw, k0);
        MS_STAMP(3 + (k0 / kKS) * 4);
        __syncthreads();
        MS_STAMP(4 + (k0 / kKS) * 4);
        if (PF >= 2 || k0 + PF * kKS < Ci) fetch(rx, rw, k0 + PF * kKS);
#if MS_CONV_ABL == 4
        if (k0 == 0)
#endif
        mac();
        MS_STAMP(5 + (k0 / kKS) * 4);
    };
#if MS_CONV_ABL == 6
    for (int k0 = 0; k0 < kKS; k0 += PF * kKS) {
#else
    for (int k0 = 0; k0 < Ci; k0 += PF * kKS) {
#endif
        slice(rxa, rwa, k0);
        if constexpr (PF >= 2) slice(rxb, rwb, k0 + kKS);
    }
    // D = B^T-major product: row index (4 * fq + r) = output channel within the tile, column fr = pixel
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int hh = h0 + wv * 2 + m, ww = w0 + fr;
#if MS_CONV_ABL == 5
        if (H > 0) continue;
#endif
        if (hh >= H || ww >= W) continue;
        unsigned short *yo = y + (((int64_t)img * H + hh) * W + ww) * Co + co0;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int c = n * 16 + fq * 4;
            if (co0 + c < Co) {
                const f32x4 v = acc[m][n];
                uint2 pk;
                pk.x = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[0]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[1]) << 16);
                pk.y = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[2]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[3]) << 16);
                *reinterpret_cast<uint2 *>(yo + c) = pk;
            }
        }
    }
    MS_STAMP(31);
    if constexpr (STATS || BRED) {
        float s1[NB][4], s2[NB][4];
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[n][r] = 0.f; s2[n][r] = 0.f; }
        if constexpr (STATS) {
            // per output channel: sum (y - p), sum (y - p)^2 over the tile's in-image pixels, y as stored (bf16-rounded)
            float pv[NB][4];
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = co0 + n * 16 + fq * 4 + r;
                    pv[n][r] = c < Co ? bout.running_mean[c] - (bout.shift ? bout.shift[c] : 0.f) : 0.f;
                }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const bool valid = h0 + wv * 2 + m < H && w0 + fr < W;
#pragma unroll
                for (int n = 0; n < NB; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float yv = (float)(__bf16)acc[m][n][r];
                        const float d = valid ? yv - pv[n][r] : 0.f;
                        s1[n][r] += d; s2[n][r] = fmaf(d, d, s2[n][r]);
                    }
            }
        } else {
            // per channel: s1 = sum dy', s2 = sum dy' * xhat with dy' = dx (as stored) * [relu passed], xhat from the pre-normalisation tensor
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const int c = co0 + n * 16 + fq * 4;
                if (c >= Co) continue;
                const float4 mu = *reinterpret_cast<const float4 *>(bred.mean + c), rs = *reinterpret_cast<const float4 *>(bred.rstd + c);
                const float4 ga = *reinterpret_cast<const float4 *>(bred.gamma + c), be = *reinterpret_cast<const float4 *>(bred.beta + c);
                const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
                const float gav[4] = {ga.x, ga.y, ga.z, ga.w}, bev[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int hh = h0 + wv * 2 + m, ww = w0 + fr;
                    if (hh >= H || ww >= W) continue;
                    const int64_t pix = ((int64_t)img * H + hh) * W + ww;
                    float xv[4];
                    if (bred.xpre_f32) {
                        const float4 t = *reinterpret_cast<const float4 *>(static_cast<const float *>(bred.xpre) + pix * bred.xps + c);
                        xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
                    } else {
                        const uint2 t = *reinterpret_cast<const uint2 *>(static_cast<const unsigned short *>(bred.xpre) + pix * bred.xps + c);
                        xv[0] = bf_lo(t.x); xv[1] = bf_hi(t.x); xv[2] = bf_lo(t.y); xv[3] = bf_hi(t.y);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xh = (xv[r] - muv[r]) * rsv[r];
                        float d = (float)(__bf16)acc[m][n][r];
                        if (bred.relu && fmaf(xh, gav[r], bev[r]) <= 0.f) d = 0.f;
                        s1[n][r] += d; s2[n][r] = fmaf(d, xh, s2[n][r]);
                    }
                }
            }
        }
        // sum over the 16 pixel lanes of a row (lane bits 0-3): four row rotations, every lane ends with the row's total
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1[n][r] = row_ror_add<0x121>(row_ror_add<0x122>(row_ror_add<0x124>(row_ror_add<0x128>(s1[n][r]))));
                s2[n][r] = row_ror_add<0x121>(row_ror_add<0x122>(row_ror_add<0x124>(row_ror_add<0x128>(s2[n][r]))));
            }
        __syncthreads();                                        // every wave has read its last fragments: sX is free
        float *red = reinterpret_cast<float *>(sX);             // [wave][2][NB * 16]
        if (fr == 0) {
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    red[(wv * 2 + 0) * (NB * 16) + n * 16 + fq * 4 + r] = s1[n][r];
                    red[(wv * 2 + 1) * (NB * 16) + n * 16 + fq * 4 + r] = s2[n][r];
                }
        }
        __syncthreads();
        if (tid < 2 * NB * 16) {
            const int st = tid / (NB * 16), c = tid - st * (NB * 16);
            if (co0 + c < Co) {
                const float v = (red[(0 * 2 + st) * (NB * 16) + c] + red[(1 * 2 + st) * (NB * 16) + c]) +
                                (red[(2 * 2 + st) * (NB * 16) + c] + red[(3 * 2 + st) * (NB * 16) + c]);
                atomicAdd((STATS ? bout.sums : bred.sums) + ((blockIdx.x % kBnRep) * 2 + st) * Co + co0 + c, v);
                if (STATS && blockIdx.x == 0 && st == 0)        // the pivot these sums are taken around, for the consumer
                    bout.sums[kBnRep * 2 * Co + co0 + c] = bout.running_mean[co0 + c] - (bout.shift ? bout.shift[co0 + c] : 0.f);
            }
        }
    }
}

int conv3x3_nhwc_dispatch(const void *x, const void *w, void *y, int batch, int H, int W, int Ci, int Co, hipStream_t s) {
    if (!x || !w || !y) return MS_ERR_NULL;
    if (batch < 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || Ci % 16 != 0 || Co % 16 != 0) return MS_ERR_SHAPE;
    if ((int64_t)H * W * Ci >= (1ll << 30) || (int64_t)Co * 9 * Ci >= (1ll << 30)) return MS_ERR_UNSUPPORTED;    // 32-bit byte offsets within an image / the weight
    if (batch == 0) return MS_OK;
    const int tiles_w = (W + kTW - 1) / kTW, tiles_h = (H + kTH - 1) / kTH;
    const int tiles_per_img = tiles_w * tiles_h;
    // output channels per workgroup: 48 (MedMamba-T's 48 / 96 / 192 / 384), or 64 where that divides the count and 48 does not
    // (MedMamba-B's 64 / 128 / 256 / 512: 64 = 48 + 16 would run a second, three-quarters empty block)
    const int cb = (Co % 64 == 0 && Co % 48 != 0) ? 64 : 48;
    const dim3 grid((unsigned)(batch * tiles_per_img), (unsigned)((Co + cb - 1) / cb));
    using bf = unsigned short;
    const BnFoldDev none = {};
    const BnBwdDev nob = {};
    if (cb == 64) hipLaunchKernelGGL((conv3x3_nhwc_kernel<4>), grid, dim3(256), 0, s, (const bf *)x, (const bf *)w, (bf *)y, H, W, Ci, Co, tiles_w, tiles_per_img, none, (bf *)nullptr, none, nob);
    else hipLaunchKernelGGL((conv3x3_nhwc_kernel<3>), grid, dim3(256), 0, s, (const bf *)x, (const bf *)w, (bf *)y, H, W, Ci, Co, tiles_w, tiles_per_img, none, (bf *)nullptr, none, nob);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

// dx = conv3x3(dy, w') (the input gradient) with the reduce pass of the BatchNorm behind the convolution's input in its epilogue
int conv3x3_bnbwd_nhwc_dispatch(const void *dy, const void *w, void *dx, int batch, int H, int W, int Ci, int Co, const MsBnBwd *red, hipStream_t s) {
    if (!dy || !w || !dx || !red || !red->x_pre || !red->gamma || !red->beta || !red->save_mean || !red->save_rstd || !red->sums) return MS_ERR_NULL;
    if (batch < 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || Ci % 16 != 0 || Co % 16 != 0) return MS_ERR_SHAPE;
    if ((int64_t)H * W * Ci >= (1ll << 30) || (int64_t)Co * 9 * Ci >= (1ll << 30)) return MS_ERR_UNSUPPORTED;    // 32-bit byte offsets within an image / the weight
    if (red->x_pre_pixel_stride < Co || red->x_pre_pixel_stride % 4 != 0 ||
        (reinterpret_cast<uintptr_t>(red->x_pre) & (red->x_pre_is_f32 ? 15 : 7)) != 0)
        return MS_ERR_STRIDE;
    if (batch == 0) return MS_OK;
    const int tiles_w = (W + kTW - 1) / kTW, tiles_h = (H + kTH - 1) / kTH;
    const int tiles_per_img = tiles_w * tiles_h;
    const int cb = (Co % 64 == 0 && Co % 48 != 0) ? 64 : 48;
    const dim3 grid((unsigned)(batch * tiles_per_img), (unsigned)((Co + cb - 1) / cb));
    using bf = unsigned short;
    const BnFoldDev none = {};
    BnBwdDev d = {};
    d.xpre = red->x_pre; d.xpre_f32 = red->x_pre_is_f32; d.xps = red->x_pre_pixel_stride; d.gamma = red->gamma; d.beta = red->beta;
    d.mean = red->save_mean; d.rstd = red->save_rstd; d.relu = red->relu; d.sums = red->sums;
    if (cb == 64) hipLaunchKernelGGL((conv3x3_nhwc_kernel<4, false, false, true>), grid, dim3(256), 0, s, (const bf *)dy, (const bf *)w, (bf *)dx, H, W, Ci, Co, tiles_w, tiles_per_img, none, (bf *)nullptr, none, d);
    else hipLaunchKernelGGL((conv3x3_nhwc_kernel<3, false, false, true>), grid, dim3(256), 0, s, (const bf *)dy, (const bf *)w, (bf *)dx, H, W, Ci, Co, tiles_w, tiles_per_img, none, (bf *)nullptr, none, d);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

static bool bn_fold_ok(const MsBnFold *f) {
    return f->sums && f->gamma && f->beta && f->running_mean && f->running_var && f->save_mean && f->save_rstd;
}
static BnFoldDev bn_fold_dev(const MsBnFold *f, int64_t npix) {
    BnFoldDev d = {};
    if (!f) return d;
    d.sums = f->sums; d.gamma = f->gamma; d.beta = f->beta; d.shift = f->shift; d.running_mean = f->running_mean;
    d.running_var = f->running_var; d.nbt = (long long *)f->num_batches_tracked; d.save_mean = f->save_mean; d.save_rstd = f->save_rstd;
    d.momentum = f->momentum; d.eps = f->eps; d.inv_n = 1.0f / (float)npix; d.unbias = npix > 1 ? (float)npix / (float)(npix - 1) : 1.0f;
    return d;
}

// The same convolution with a training-mode BatchNorm folded into its input side (bn_in: x is the PRE-BatchNorm tensor, relu(bn(x)) is
// convolved and, when xhat != NULL, also written out) and / or its output side (bn_out: the statistics of y go to bn_out->sums).
int conv3x3_bn_nhwc_dispatch(const void *x, const void *w, void *y, int batch, int H, int W, int Ci, int Co, const MsBnFold *bn_in,
                             void *xhat, const MsBnFold *bn_out, hipStream_t s) {
    if (!x || !w || !y) return MS_ERR_NULL;
    if ((bn_in && !bn_fold_ok(bn_in)) || (bn_out && !bn_fold_ok(bn_out))) return MS_ERR_NULL;
    if (batch < 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || Ci % 16 != 0 || Co % 16 != 0) return MS_ERR_SHAPE;
    if ((int64_t)H * W * Ci >= (1ll << 30) || (int64_t)Co * 9 * Ci >= (1ll << 30)) return MS_ERR_UNSUPPORTED;    // 32-bit byte offsets within an image / the weight
    if (bn_in && Ci > kBnMaxC) return MS_ERR_UNSUPPORTED;
    if (batch == 0) return MS_OK;
    if (!bn_in && !bn_out) return conv3x3_nhwc_dispatch(x, w, y, batch, H, W, Ci, Co, s);
    const int tiles_w = (W + kTW - 1) / kTW, tiles_h = (H + kTH - 1) / kTH;
    const int tiles_per_img = tiles_w * tiles_h;
    const int cb = (Co % 64 == 0 && Co % 48 != 0) ? 64 : 48;
    const dim3 grid((unsigned)(batch * tiles_per_img), (unsigned)((Co + cb - 1) / cb));
    using bf = unsigned short;
    const int64_t npix = (int64_t)batch * H * W;
    const BnFoldDev di = bn_fold_dev(bn_in, npix), dz = bn_fold_dev(bn_out, npix);
    const BnBwdDev nob = {};
#define MS_CONV_BN(NBv, I, O) hipLaunchKernelGGL((conv3x3_nhwc_kernel<NBv, I, O>), grid, dim3(256), 0, s, (const bf *)x, (const bf *)w, (bf *)y, H, W, Ci, Co, \
        tiles_w, tiles_per_img, di, (bf *)xhat, dz, nob)
    if (cb == 64) { if (bn_in && bn_out) MS_CONV_BN(4, true, true); else if (bn_in) MS_CONV_BN(4, true, false); else MS_CONV_BN(4, false, true); }
    else          { if (bn_in && bn_out) MS_CONV_BN(3, true, true); else if (bn_in) MS_CONV_BN(3, true, false); else MS_CONV_BN(3, false, true); }
#undef MS_CONV_BN
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}


// ---- weight gradient -----------------------------------------------------------------------------------------------------
//   dW[co][tap][ci] = sum over (image, h, w) of dy[h, w, co] * x[h + dy - 1, w + dx - 1, ci]
// The reduction axis of the MFMA is the PIXEL axis: one k-step = the 16 pixels of one image row of the 8 x 16 tile.  Both operands
// sit in LDS as they are in memory, [pixel][channel], and are read with ds_read_b64_tr_b16 (the transposing LDS read: a 16-lane
// group fetches 4 pixels x 16 channels, every lane receives the 4 pixel values of ITS channel) -- no transposed copies.
//   workgroup = a block of 48 output x 48 input channels, persistent over pixel tiles; the 3 x 27 products (3 co tiles x 9 taps x
//   3 ci tiles) are dealt to the 4 waves by product column (27 = 7 + 7 + 7 + 6 columns, every wave all 3 co tiles): per k-step a
//   wave reads 3 + 7 fragments for 21 MFMAs and keeps 21 accumulator tiles (84 VGPRs).
//   The workgroup's partial block goes to a workspace row; ms_conv3x3_wgrad sums the rows into the fp32 (Co, Ci, 3, 3) gradient.
typedef short bf16x4_t __attribute__((ext_vector_type(4)));

// GC = channels per block on both sides: 48 (MedMamba-T's 48 / 96 / 192 / 384), or 64 where that divides both channel counts and 48 does
// not (MedMamba-B's 64 / 128 / 256 / 512: 48-channel blocks covered 64 channels with 2 x 2 blocks of which 44 % was real work -- the
// weight gradients were 5.1 of that configuration's 74 ms); 4 co tiles x 9 taps x 4 ci tiles = 36 product columns, 9 per wave.
constexpr int wgrad_gc(int Ci, int Co) { return (Ci % 64 == 0 && Co % 64 == 0 && (Ci % 48 != 0 || Co % 48 != 0)) ? 64 : 48; }

__device__ __forceinline__ bf16x4 tr_frag(const unsigned short *s_pix0, int pitch, int lane) {
    // fragment of 16 channels (starting at s_pix0's channel) x 16 pixels (rows of the [pixel][channel] image starting at s_pix0):
    // lane (fr = channel, fq = pixel group) receives pixels 4 fq .. 4 fq + 3 of channel fr
    const int fr = lane & 15, fq = lane >> 4;
    typedef bf16x4 __attribute__((address_space(3))) *lds_p;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(s_pix0 + (4 * fq + (fr >> 2)) * pitch + 4 * (fr & 3)));
}

// (64-channel blocks: 36 accumulator tiles per wave = 300 registers, one workgroup per CU; capped at 256 registers it spills and
// measured 157 / 129 / 124 / 133 us against 126 / 120 / 121 / 133 at MedMamba-B's four stages, bs 32 512 x 512)
template <int kGC>
__global__ void __launch_bounds__(256, kGC == 64 ? 1 : 2)
conv3x3_wgrad_kernel(const unsigned short *__restrict__ x, const unsigned short *__restrict__ dy, float *__restrict__ part,
                     int H, int W, int Ci, int Co, int tiles_w, int tiles_per_img, int n_tiles, int nci) {
    constexpr int kGP = 80;                          // LDS pixel pitch (bf16): 160 B = 32 B x 5 -- a ds_read_b64_tr_b16 serves 32 lanes = 8 pixel rows x 32 B at once,
                                                     // and rows 160 B apart fall into 8 different 32-byte bank groups (112 / 144 B: two of the 8 overlap; 50 % of the
                                                     // kernel's LDS cycles were conflicts)
    static_assert(kGC <= 64, "pixel pitch");
    constexpr int NT = kGC / 16, kCols = 9 * NT;     // channel tiles per side; product columns (tap, ci tile)
    constexpr int kPC = kGC / 8;                                        // 16-byte pieces per pixel
    constexpr int kNX = (kHH * kHW * kPC + 255) / 256, kNG = (kTH * kTW * kPC + 255) / 256;      // 16-byte loads per thread and tile: halo, dy
    constexpr int kPixX = (kNX * 256 + kPC - 1) / kPC, kPixG = (kNG * 256 + kPC - 1) / kPC;      // LDS pixels incl. the last pass's overhang (unguarded writes)
    __shared__ __attribute__((aligned(16))) unsigned short sX[kPixX * kGP];          // x halo tile, kGC input channels
    __shared__ __attribute__((aligned(16))) unsigned short sG[kPixG * kGP];          // dy tile, kGC output channels
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cob = blockIdx.y / nci, cib = blockIdx.y % nci;
    const int co0 = cob * kGC, ci0 = cib * kGC;
    // this wave's product columns: col = tap * NT + ci tile, col % 4 == wv
    constexpr int kMaxCols = (kCols + 3) / 4;
    f32x4 acc[NT][kMaxCols];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int j = 0; j < kMaxCols; ++j) acc[a][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging through registers, one tile ahead: the next tile's 16-byte loads (<= 10 per thread) are in flight while this tile is
    // multiplied.  Raw buffer loads; what does not depend on the tile -- a piece's position inside the tile, its byte offset relative to
    // the tile's first pixel, its LDS address -- is worked out once per thread, per tile only the scalar tile origin and the edge
    // tests are left (the first version redid the index arithmetic of every load for every tile: ~400 vector instructions per tile and
    // wave beside 84 MFMAs).
    const int batch = n_tiles / tiles_per_img;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(x), 0, (int)((int64_t)batch * H * W * Ci * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(dy), 0, (int)((int64_t)batch * H * W * Co * 2), 0x00020000);
    int relx[kNX], hwx[kNX], ldx[kNX], relg[kNG], hwg[kNG], ldg[kNG];      // hw*: (row << 16 | column) inside the tile, -1 = not a piece of the tile
#pragma unroll
    for (int i = 0; i < kNX; ++i) {
        const int idx = tid + i * 256, pix = idx / kPC, pc = idx - pix * kPC, ph = pix / kHW - 1, pw = pix % kHW - 1;
        relx[i] = ((ph * W + pw) * Ci + ci0 + pc * 8) * 2;
        hwx[i] = (idx < kHH * kHW * kPC && ci0 + pc * 8 < Ci) ? ((ph + 1) << 16) | (pw + 1) : -1;
        ldx[i] = pix * kGP + pc * 8;
    }
#pragma unroll
    for (int i = 0; i < kNG; ++i) {
        const int idx = tid + i * 256, pix = idx / kPC, pc = idx - pix * kPC, ph = pix / kTW, pw = pix % kTW;
        relg[i] = ((ph * W + pw) * Co + co0 + pc * 8) * 2;
        hwg[i] = (idx < kTH * kTW * kPC && co0 + pc * 8 < Co) ? (ph << 16) | pw : -1;
        ldg[i] = pix * kGP + pc * 8;
    }
    u32x4 rx[kNX], rg[kNG];
    auto fetch = [&](int tile) {
        const int img = tile / tiles_per_img, tt = tile - img * tiles_per_img;
        const int h0 = (tt / tiles_w) * kTH, w0 = (tt % tiles_w) * kTW;
        const int pix0 = (img * H + h0) * W + w0;                       // the tile's first output pixel
        const int bx = pix0 * Ci * 2, bg = pix0 * Co * 2;
#pragma unroll
        for (int i = 0; i < kNX; ++i) {
            const int hh = h0 - 1 + (hwx[i] >> 16), ww = w0 - 1 + (hwx[i] & 0xFFFF);
            const bool ok = hwx[i] >= 0 && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
            rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rsx, ok ? bx + relx[i] : (int)kOob, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < kNG; ++i) {
            const int hh = h0 + (hwg[i] >> 16), ww = w0 + (hwg[i] & 0xFFFF);
            const bool ok = hwg[i] >= 0 && hh < H && ww < W;
            rg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsg, ok ? bg + relg[i] : (int)kOob, 0, 0);
        }
    };
    if ((int)blockIdx.x < n_tiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kNX; ++i) *reinterpret_cast<u32x4 *>(sX + ldx[i]) = rx[i];
#pragma unroll
        for (int i = 0; i < kNG; ++i) *reinterpret_cast<u32x4 *>(sG + ldg[i]) = rg[i];
        __syncthreads();
        if (tile + (int)gridDim.x < n_tiles) fetch(tile + gridDim.x);
        // one k-step = the 32 pixels of TWO image rows (v_mfma_f32_16x16x32_bf16: the K = 16 form occupies the matrix core for the same 16
        // cycles): a lane's 8 k-values are pixels 4 fq .. 4 fq + 3 of row r and of row r + 1 -- the same assignment on both operands
        auto frag2 = [&](const unsigned short *s_row, int row_pitch) {
            const bf16x4 lo = tr_frag(s_row, kGP, lane), hi = tr_frag(s_row + row_pitch * kGP, kGP, lane);
            return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
#pragma unroll 2
        for (int r = 0; r < kTH; r += 2) {
            bf16x8 ga[NT];
#pragma unroll
            for (int a = 0; a < NT; ++a) ga[a] = frag2(sG + (r * kTW) * kGP + a * 16, kTW);
#pragma unroll
            for (int j = 0; j < kMaxCols; ++j) {
                const int col = wv + 4 * j;                       // compile-time j, wave-uniform col
                if (col < kCols) {
                    const int tap = col / NT, ct = col - tap * NT;
                    const int dyy = tap / 3, dxx = tap - dyy * 3;
                    const bf16x8 xb = frag2(sX + ((r + dyy) * kHW + dxx) * kGP + ct * 16, kHW);
#pragma unroll
                    for (int a = 0; a < NT; ++a) acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga[a], xb, acc[a][j], 0, 0, 0);
                }
            }
        }
    }
    // D[co][ci]: lane (ci = fr) holds co = 4 fq + r.  Partial block layout: [co (kGC)][tap (9)][ci (kGC)] fp32 per (blockIdx.x, blockIdx.y)
    const int fr = lane & 15, fq = lane >> 4;
    float *pb = part + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * (kGC * 9 * kGC);
#pragma unroll
    for (int j = 0; j < kMaxCols; ++j) {
        const int col = wv + 4 * j;
        if (col < kCols) {
            const int tap = col / NT, ct = col - tap * NT;
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    pb[((a * 16 + 4 * fq + rr) * 9 + tap) * kGC + ct * 16 + fr] = acc[a][j][rr];
        }
    }
}

// dW (Co, Ci, 3, 3) fp32 contiguous = sum of the workers' partial blocks [worker][co block][ci block][48][9][48].
// The partial rows are read in THEIR order (16-byte pieces, 16 consecutive pieces per slice of a block: 256-byte segments) by
// 16 worker slices per block, combined in LDS; only the final write is scattered (it is 1 / nworkers of the traffic).  The first
// version walked dW's (co, ci, tap) order with one thread per element: 192-byte strides between neighbouring lanes and 81
// blocks for the whole chip at stage 0 -- 34 us for 42 MB.
constexpr int kFinCols = 16, kFinSlices = 16;
template <int kGC>
__global__ void __launch_bounds__(kFinCols * kFinSlices)
conv3x3_wgrad_finalize_kernel(const float *__restrict__ part, float *__restrict__ dW, int Ci, int Co, int nci, int nblk, int nworkers) {
    __shared__ float4 red[kFinSlices][kFinCols];
    constexpr int kBlkQ = kGC * 9 * kGC / 4;                                // 16-byte pieces per partial block
    const int col = threadIdx.x % kFinCols, sl = threadIdx.x / kFinCols;
    const int64_t q = (int64_t)blockIdx.x * kFinCols + col, nq = (int64_t)nblk * kBlkQ;
    const float4 *p4 = reinterpret_cast<const float4 *>(part) + (q < nq ? q : 0);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int wk = sl; wk < nworkers; wk += kFinSlices) {
        const float4 v = p4[(int64_t)wk * nq];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    red[sl][col] = a;
    __syncthreads();
#pragma unroll
    for (int s = kFinSlices / 2; s >= 1; s >>= 1) {
        if (sl < s) {
            const float4 o = red[sl + s][col];
            float4 &m = red[sl][col];
            m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
        }
        __syncthreads();
    }
    if (sl != 0 || q >= nq) return;
    const float4 t = red[0][col];
    const int blk = (int)(q / kBlkQ), rem = (int)(q % kBlkQ);
    const int co = (blk / nci) * kGC + rem / (9 * (kGC / 4)), tap = (rem / (kGC / 4)) % 9, ci = (blk % nci) * kGC + (rem % (kGC / 4)) * 4;
    if (co >= Co) return;
    const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (ci + i < Ci) dW[((int64_t)co * Ci + ci + i) * 9 + tap] = v[i];
}

static int wgrad_workers(int n_tiles, int nblk) {
    static const int total = [] { const char *e = getenv("MEDSCAN_WGRAD_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
    int wk = (total + nblk - 1) / nblk;                     // about two workgroups per CU in total (256: main 55 + finalize 18 us; 512: 35 + 34)
    if (wk > n_tiles) wk = n_tiles;
    return wk < 1 ? 1 : wk;
}

int64_t conv3x3_wgrad_scratch_floats(int batch, int H, int W, int Ci, int Co) {
    if (batch <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0) return 0;
    const int n_tiles = batch * ((W + kTW - 1) / kTW) * ((H + kTH - 1) / kTH);
    const int kGC = wgrad_gc(Ci, Co);
    const int nblk = ((Co + kGC - 1) / kGC) * ((Ci + kGC - 1) / kGC);
    return (int64_t)wgrad_workers(n_tiles, nblk) * nblk * (kGC * 9 * kGC);
}

int conv3x3_wgrad_dispatch(const void *x, const void *dy, float *dW, float *scratch, int64_t scratch_floats, int batch, int H, int W,
                           int Ci, int Co, hipStream_t s) {
    if (!x || !dy || !dW || !scratch) return MS_ERR_NULL;
    if (batch <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || Ci % 8 != 0 || Co % 8 != 0) return MS_ERR_SHAPE;
    if (scratch_floats < conv3x3_wgrad_scratch_floats(batch, H, W, Ci, Co)) return MS_ERR_SHAPE;
    if ((int64_t)batch * H * W * (Ci > Co ? Ci : Co) >= (1ll << 30)) return MS_ERR_UNSUPPORTED;          // 32-bit byte offsets into x / dy
    const int tiles_w = (W + kTW - 1) / kTW, tiles_h = (H + kTH - 1) / kTH;
    const int tiles_per_img = tiles_w * tiles_h, n_tiles = batch * tiles_per_img;
    const int gc = wgrad_gc(Ci, Co);
    const int nco = (Co + gc - 1) / gc, nci = (Ci + gc - 1) / gc, nblk = nco * nci;
    const int wk = wgrad_workers(n_tiles, nblk);
    using bf = unsigned short;
    const int64_t nq = (int64_t)nblk * (gc * 9 * gc / 4);
#define MS_WGRAD(GCv)                                                                                                                        \
    hipLaunchKernelGGL((conv3x3_wgrad_kernel<GCv>), dim3((unsigned)wk, (unsigned)nblk), dim3(256), 0, s, (const bf *)x, (const bf *)dy, scratch, H, W, \
                       Ci, Co, tiles_w, tiles_per_img, n_tiles, nci);                                                                          \
    hipLaunchKernelGGL((conv3x3_wgrad_finalize_kernel<GCv>), dim3((unsigned)((nq + kFinCols - 1) / kFinCols)), dim3(kFinCols * kFinSlices), 0, s,    \
                       scratch, dW, Ci, Co, nci, nblk, wk)
    if (gc == 64) { MS_WGRAD(64); } else { MS_WGRAD(48); }
#undef MS_WGRAD
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms

#ifdef MS_CONV_CLOCK
extern "C" int ms_debug_conv_clock(long long *host_out, int n) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ms::g_conv_clock), sizeof(long long) * n, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif
