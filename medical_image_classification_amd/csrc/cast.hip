// bf16 working copies of the fp32 master weights for gfx950, all of them in ONE launch.
// Under bf16 autocast every weight that feeds a MIOpen convolution or a library GEMM is cast once per step
// (torch: one `aten::_to_copy` launch per weight per forward, ~90 launches of 4-5 us for MedMamba-T, plus a layout copy per
// convolution weight because MIOpen's NHWC kernels want (Co, kh, kw, Ci) memory).  Here: a table of (src, dst, shape)
// descriptors in device memory, one grid over all of them; convolution weights are written in channels_last order directly.
// Replaces the per-use casts of torch.autocast around MedMamba.py:517-527 (conv branch) and :284,326 (projections).
#include <hip/hip_runtime.h>
#include "medscan.h"

namespace ms {

// Work list: workgroup b serves blocks[b] = (tensor, piece).
//   plain casts (|taps| <= 1) and tap counts above 9: piece = MS_CAST_CHUNK consecutive elements (in destination order);
//   convolution weights with 2..9 taps: piece = a tile of 64 output channels x 16 input channels x taps, staged through LDS so
//   that the fp32 rows are READ in their memory order and both bf16 layouts are WRITTEN in runs (the first version walked the
//   destination order with 4-byte reads 36 B .. 14 KB apart -- every element its own sector -- from at most 64 workgroups per
//   tensor: 175 us per step for MedMamba-T's 23 M elements).
// Tensor t: dst[(o * taps + k) * inner + i] = bf16(src[(o * inner + i) * taps + k]) (taps > 1), the flipped / transposed form
// for taps < -1 (see medscan.h), a plain elementwise cast otherwise.
constexpr int kTO = 64, kTI = 16, kMaxTaps = 9;
static_assert(MS_CAST_CHUNK == 2048 && MS_CAST_TILE_O == kTO && MS_CAST_TILE_I == kTI && MS_CAST_TILE_MAX_TAPS == kMaxTaps, "medscan.h");

__device__ __forceinline__ unsigned short to_bf16(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

__global__ void __launch_bounds__(256)
cast_bf16_multi_kernel(const MsCastDesc *__restrict__ desc, const int2 *__restrict__ blocks) {
    __shared__ float tile[kTO][kTI * kMaxTaps + 1];
    const int2 bt = blocks[blockIdx.x];
    const MsCastDesc d = desc[bt.x];
    const float *__restrict__ src = static_cast<const float *>(d.src);
    unsigned short *__restrict__ dst = static_cast<unsigned short *>(d.dst);
    const int taps = d.taps < 0 ? -d.taps : d.taps, inner = d.inner;
    if (taps >= 2 && taps <= kMaxTaps) {
        const int row = inner * taps;                                   // floats per output channel
        const int outer = (int)(d.n / row);
        const int nit = (inner + kTI - 1) / kTI;
        const int o0 = (bt.y / nit) * kTO, i0 = (bt.y % nit) * kTI;
        const int tw = min(kTI, inner - i0) * taps;                     // valid floats per tile row
        const int W = kTI * taps;
        for (int idx = threadIdx.x; idx < kTO * W; idx += 256) {
            const int oo = idx / W, c = idx - oo * W;
            tile[oo][c] = (o0 + oo < outer && c < tw) ? src[(int64_t)(o0 + oo) * row + i0 * taps + c] : 0.0f;
        }
        __syncthreads();
        if (d.taps > 0) {           // channels_last copy: (o, k, i), i fastest
            for (int idx = threadIdx.x; idx < kTO * W; idx += 256) {
                const int ii = idx % kTI, ok = idx / kTI, k = ok % taps, oo = ok / taps;
                if (o0 + oo < outer && i0 + ii < inner)
                    dst[((int64_t)(o0 + oo) * taps + k) * inner + i0 + ii] = to_bf16(tile[oo][ii * taps + k]);
            }
        } else {                    // input-gradient weight: (i, flipped k, o), o fastest
            for (int idx = threadIdx.x; idx < kTO * W; idx += 256) {
                const int oo = idx % kTO, ik = idx / kTO, k = ik % taps, ii = ik / taps;
                if (o0 + oo < outer && i0 + ii < inner)
                    dst[((int64_t)(i0 + ii) * taps + (taps - 1 - k)) * outer + o0 + oo] = to_bf16(tile[oo][ii * taps + k]);
            }
        }
        return;
    }
    const int64_t e0 = (int64_t)bt.y * MS_CAST_CHUNK;
    const int cnt = (int)min((int64_t)MS_CAST_CHUNK, d.n - e0);
    if (taps <= 1) {
        for (int o = threadIdx.x; o < cnt; o += 256) dst[e0 + o] = to_bf16(src[e0 + o]);
        return;
    }
    // other tap counts (the 4x4 patch embedding): element by element in destination order
    const int64_t outer = d.n / ((int64_t)inner * taps);
    for (int o = threadIdx.x; o < cnt; o += 256) {
        const int64_t e = e0 + o;
        if (d.taps < 0) {
            const int64_t ik = e / outer, oo = e - ik * outer, i = ik / taps;
            const int k = taps - 1 - (int)(ik - i * taps);
            dst[e] = to_bf16(src[(oo * inner + i) * taps + k]);
        } else {
            const int64_t ok = e / inner, oo = ok / taps;
            const int i = (int)(e - ok * inner), k = (int)(ok - oo * taps);
            dst[e] = to_bf16(src[(oo * inner + i) * taps + k]);
        }
    }
}

// Non-overlapping 4 x 4 patches of an NCHW fp32 image batch as bf16 rows [patch][c * 16 + i * 4 + j] -- the im2col of the patch
// embedding `nn.Conv2d(in_chans, embed_dim, 4, stride 4)` (MedMamba.py:146-169), whose product is then ONE GEMM on ms_gemm_bf16
// (torch ran a layout copy of the images, MIOpen's implicit-GEMM kernel with five transposes around it, a cast of the output and
// a reduction for the bias gradient: ~0.35 ms per step).  One thread = one 4-float row of a patch: 16-byte loads that are
// consecutive along the image row, 8-byte stores.
__global__ void __launch_bounds__(256)
patchify4_bf16_kernel(const float *__restrict__ x, unsigned short *__restrict__ out, int C, int H, int W, int64_t n_rows) {
    const int pw_n = W / 4, ph_n = H / 4;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n_rows; t += (int64_t)gridDim.x * 256) {
        // t = (((b * ph_n + ph) * C + c) * 4 + i) * pw_n + pw : pw fastest (coalesced reads along the image row)
        const int pw = (int)(t % pw_n);
        int64_t r = t / pw_n;
        const int i = (int)(r & 3); r >>= 2;
        const int c = (int)(r % C); r /= C;
        const int ph = (int)(r % ph_n);
        const int64_t b = r / ph_n;
        const float4 v = *reinterpret_cast<const float4 *>(x + ((b * C + c) * H + ph * 4 + i) * (int64_t)W + pw * 4);
        uint2 o;
        o.x = (unsigned)to_bf16(v.x) | ((unsigned)to_bf16(v.y) << 16);
        o.y = (unsigned)to_bf16(v.z) | ((unsigned)to_bf16(v.w) << 16);
        *reinterpret_cast<uint2 *>(out + ((b * ph_n + ph) * pw_n + pw) * (int64_t)(C * 16) + c * 16 + i * 4) = o;
    }
}

int patchify4_bf16_dispatch(const float *x, void *out, int batch, int C, int H, int W, hipStream_t s) {
    if (!x || !out) return MS_ERR_NULL;
    if (batch < 0 || C <= 0 || H <= 0 || W <= 0 || H % 4 != 0 || W % 4 != 0) return MS_ERR_SHAPE;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 7)) return MS_ERR_STRIDE;
    const int64_t n_rows = (int64_t)batch * C * H * (W / 4);
    if (n_rows == 0) return MS_OK;
    const int64_t blocks = (n_rows + 255) / 256;
    hipLaunchKernelGGL(patchify4_bf16_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, s, x, (unsigned short *)out, C, H, W,
                       n_rows);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int cast_bf16_multi_dispatch(const MsCastDesc *desc, const int32_t *blocks, int n_blocks, hipStream_t s) {
    if (n_blocks < 0) return MS_ERR_SHAPE;
    if (n_blocks == 0) return MS_OK;
    if (!desc || !blocks) return MS_ERR_NULL;
    hipLaunchKernelGGL(cast_bf16_multi_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, desc, reinterpret_cast<const int2 *>(blocks));
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
