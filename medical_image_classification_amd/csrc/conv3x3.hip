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
#include "medscan.h"

namespace ms {

typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int kTH = 8, kTW = 16;                 // output tile (pixels)
constexpr int kHH = kTH + 2, kHW = kTW + 2;      // halo tile
constexpr int kKS = 32;                          // input channels per LDS slice (49 KB of LDS per workgroup: 3 per CU)
constexpr int kXP = kKS + 8;                     // halo pixel pitch (bf16): 144 B = 36 banks -> the 16 pixels of a fragment spread out
constexpr int kWP = kKS + 8;                     // weight row pitch (bf16)
}

// NB = output-channel tiles of 16 per workgroup (3 or 6: 48 or 96 channels)
template <int NB>
__global__ void __launch_bounds__(256)
conv3x3_nhwc_kernel(const unsigned short *__restrict__ x, const unsigned short *__restrict__ w, unsigned short *__restrict__ y,
                    int H, int W, int Ci, int Co, int tiles_w, int tiles_per_img) {
    __shared__ __attribute__((aligned(16))) unsigned short sX[kHH * kHW * kXP];
    __shared__ __attribute__((aligned(16))) unsigned short sW[9 * NB * 16 * kWP];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tile = blockIdx.x, img = tile / tiles_per_img, tt = tile - img * tiles_per_img;
    const int h0 = (tt / tiles_w) * kTH, w0 = (tt % tiles_w) * kTW;
    const int co0 = blockIdx.y * (NB * 16);
    const int fr = lane & 15, fq = lane >> 4;

    f32x4 acc[2][NB];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned short *xi = x + (int64_t)img * H * W * Ci;
    for (int k0 = 0; k0 < Ci; k0 += kKS) {
        const int ks = min(kKS, Ci - k0);                       // channels in this slice (multiple of 16)
        __syncthreads();                                        // the previous slice's fragments have been read
        // halo tile: kHH x kHW pixels x ks channels, 16-byte pieces; out-of-image pixels are zero
        const int pieces = ks / 8;
        for (int idx = tid; idx < kHH * kHW * pieces; idx += 256) {
            const int pix = idx / pieces, pc = idx - pix * pieces;
            const int hh = h0 - 1 + pix / kHW, ww = w0 - 1 + pix % kHW;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (hh >= 0 && hh < H && ww >= 0 && ww < W) v = *reinterpret_cast<const uint4 *>(xi + ((int64_t)hh * W + ww) * Ci + k0 + pc * 8);
            *reinterpret_cast<uint4 *>(sX + pix * kXP + pc * 8) = v;
        }
        // weights of this channel slice: [tap][co (NB * 16)][ks]; w is (Co, 9, Ci)
        for (int idx = tid; idx < 9 * NB * 16 * pieces; idx += 256) {
            const int row = idx / pieces, pc = idx - row * pieces;       // row = tap * (NB*16) + col
            const int tap = row / (NB * 16), col = row - tap * (NB * 16);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (co0 + col < Co) v = *reinterpret_cast<const uint4 *>(w + ((int64_t)(co0 + col) * 9 + tap) * Ci + k0 + pc * 8);
            *reinterpret_cast<uint4 *>(sW + row * kWP + pc * 8) = v;
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            for (int kk = 0; kk < ks; kk += 16) {
                // A fragments: 16 pixels of one image row (fr) x 4 consecutive channels (fq) -> rows (wv*2 + m + dy), cols (fr + dx)
                bf16x4 a[2];
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    a[m] = *reinterpret_cast<const bf16x4 *>(sX + ((wv * 2 + m + dy) * kHW + fr + dx) * kXP + kk + fq * 4);
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const bf16x4 b = *reinterpret_cast<const bf16x4 *>(sW + ((tap * NB + n) * 16 + fr) * kWP + kk + fq * 4);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(b, a[m], acc[m][n], 0, 0, 0);
                }
            }
        }
    }
    // D = B^T-major product: row index (4 * fq + r) = output channel within the tile, column fr = pixel
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int hh = h0 + wv * 2 + m, ww = w0 + fr;
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
}

int conv3x3_nhwc_dispatch(const void *x, const void *w, void *y, int batch, int H, int W, int Ci, int Co, hipStream_t s) {
    if (!x || !w || !y) return MS_ERR_NULL;
    if (batch < 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || Ci % 16 != 0 || Co % 16 != 0) return MS_ERR_SHAPE;
    if (batch == 0) return MS_OK;
    const int tiles_w = (W + kTW - 1) / kTW, tiles_h = (H + kTH - 1) / kTH;
    const int tiles_per_img = tiles_w * tiles_h;
    const int cb = 48;                                       // output channels per workgroup
    const dim3 grid((unsigned)(batch * tiles_per_img), (unsigned)((Co + cb - 1) / cb));
    using bf = unsigned short;
    hipLaunchKernelGGL((conv3x3_nhwc_kernel<3>), grid, dim3(256), 0, s, (const bf *)x, (const bf *)w, (bf *)y, H, W, Ci, Co, tiles_w, tiles_per_img);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
