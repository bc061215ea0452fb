// Fused tail of the two-branch block SS_Conv_SSM for gfx950 (MedMamba.py:515,534-538):
//   x   = drop_path(ssm_branch)                per-sample scale s[b] (timm DropPath: mask / keep_prob), optional
//   out = channel_shuffle(cat(left, x), 2) + input        i.e.  out[p, 2i] = left[p, i] + input[p, 2i]
//                                                               out[p, 2i+1] = s[b] x[p, i] + input[p, 2i+1]
// The reference runs a concat, a transpose copy (the shuffle), a broadcast multiply and an add: four HBM round trips over
// the block's activation; here one pass (read left, x, input once, write out once), and one pass backward
// (d_left[p,i] = dout[p,2i], d_x[p,i] = s[b] dout[p,2i+1]; d_input IS dout, no copy).
// HBM-bound elementwise work: one thread = 4 consecutive output channels (16-byte loads/stores of the fp32 streams).
#include <hip/hip_runtime.h>
#include "medscan.h"

namespace ms {

__device__ __forceinline__ float bt_bf2f(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
__device__ __forceinline__ unsigned short bt_f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

template <typename T> __device__ __forceinline__ float2 ld2(const T *p);
template <> __device__ __forceinline__ float2 ld2<float>(const float *p) { return *reinterpret_cast<const float2 *>(p); }
template <> __device__ __forceinline__ float2 ld2<unsigned short>(const unsigned short *p) {
    const unsigned v = *reinterpret_cast<const unsigned *>(p);
    return make_float2(bt_bf2f((unsigned short)(v & 0xFFFFu)), bt_bf2f((unsigned short)(v >> 16)));
}
template <typename T> __device__ __forceinline__ void st2(T *p, float a, float b);
template <> __device__ __forceinline__ void st2<float>(float *p, float a, float b) { *reinterpret_cast<float2 *>(p) = make_float2(a, b); }
template <> __device__ __forceinline__ void st2<unsigned short>(unsigned short *p, float a, float b) {
    *reinterpret_cast<unsigned *>(p) = (unsigned)bt_f2bf(a) | ((unsigned)bt_f2bf(b) << 16);
}

// n4 = npix * C / 4 work items; item t covers output channels 4q..4q+3 of pixel p (q = t % (C/4), p = t / (C/4)),
// i.e. half-channels 2q, 2q+1 of both branches.
template <typename TL, typename TX>
__global__ void __launch_bounds__(256)
block_tail_fwd_kernel(const TL *__restrict__ left, const TX *__restrict__ x, const float *__restrict__ input,
                      const float *__restrict__ scale, float *__restrict__ out, int64_t n4, int c4, int64_t hw) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += stride) {
        const int64_t p = t / c4;
        const int q = (int)(t - p * c4);
        const int64_t hoff = p * (2 * c4) + 2 * q;                  // offset in the half-width tensors
        const float2 l = ld2(left + hoff), xv = ld2(x + hoff);
        const float4 in = *reinterpret_cast<const float4 *>(input + 4 * t);
        const float s = scale ? scale[p / hw] : 1.0f;
        float4 o;
        o.x = l.x + in.x; o.y = fmaf(s, xv.x, in.y); o.z = l.y + in.z; o.w = fmaf(s, xv.y, in.w);
        *reinterpret_cast<float4 *>(out + 4 * t) = o;
    }
}

// `left` non-NULL: the left half is the output of a ReLU (the conv branch ends with one, MedMamba.py:526) and its gradient is
// masked here, dleft = dout * [left > 0], instead of by a threshold_backward pass of its own.
template <typename TL, typename TX>
__global__ void __launch_bounds__(256)
block_tail_bwd_kernel(const float *__restrict__ dout, const float *__restrict__ scale, const TL *__restrict__ left, TL *__restrict__ dleft,
                      TX *__restrict__ dx, int64_t n4, int c4, int64_t hw) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += stride) {
        const int64_t p = t / c4;
        const int q = (int)(t - p * c4);
        const int64_t hoff = p * (2 * c4) + 2 * q;
        const float4 g = *reinterpret_cast<const float4 *>(dout + 4 * t);
        const float s = scale ? scale[p / hw] : 1.0f;
        float gl0 = g.x, gl1 = g.z;
        if (left) { const float2 l = ld2(left + hoff); gl0 = l.x > 0.0f ? gl0 : 0.0f; gl1 = l.y > 0.0f ? gl1 : 0.0f; }
        st2(dleft + hoff, gl0, gl1);
        st2(dx + hoff, s * g.y, s * g.w);
    }
}

// Gradient of the block INPUT in one pass: d_input = dout (the residual edge) + cat(d_left_half, d_right_half) (the edge
// through `input.chunk(2, -1)`).  autograd would run a concat of the two halves and then an add with the residual's
// gradient: two more round trips over the block's activation.  One thread = 4 consecutive channels of one pixel.
template <typename TL, typename TR>
__global__ void __launch_bounds__(256)
block_head_bwd_kernel(const float *__restrict__ dout, const TL *__restrict__ dl, const TR *__restrict__ dr,
                      float *__restrict__ dinp, int64_t n4, int c4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int h4 = c4 / 2;                                              // float4 groups per half
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += stride) {
        const int64_t p = t / c4;
        const int q = (int)(t - p * c4);
        float4 g = *reinterpret_cast<const float4 *>(dout + 4 * t);
        float2 a, b;
        if (q < h4) { const TL *s = dl + (p * h4 + q) * 4; a = ld2(s); b = ld2(s + 2); }
        else        { const TR *s = dr + (p * h4 + (q - h4)) * 4; a = ld2(s); b = ld2(s + 2); }
        g.x += a.x; g.y += a.y; g.z += b.x; g.w += b.y;
        *reinterpret_cast<float4 *>(dinp + 4 * t) = g;
    }
}

static unsigned tail_grid(int64_t n4) {
    const int64_t blocks = (n4 + 255) / 256;
    return (unsigned)(blocks < 1 ? 1 : (blocks > 256 * 32 ? 256 * 32 : blocks));      // grid-stride beyond 32 blocks per CU
}

int block_tail_fwd_dispatch(const void *left, int left_is_bf16, const void *x, int x_is_bf16, const float *input,
                            const float *scale, float *out, int64_t npix, int64_t hw, int C, hipStream_t s) {
    if (!left || !x || !input || !out) return MS_ERR_NULL;
    if (npix < 0 || hw <= 0 || C <= 0 || C % 4 != 0 || npix % hw != 0) return MS_ERR_SHAPE;
    if (npix == 0) return MS_OK;
    const int64_t n4 = npix * (C / 4);
    const dim3 grid(tail_grid(n4)), block(256);
    using bf = unsigned short;
    if (left_is_bf16 && x_is_bf16)
        hipLaunchKernelGGL((block_tail_fwd_kernel<bf, bf>), grid, block, 0, s, (const bf *)left, (const bf *)x, input, scale, out, n4, C / 4, hw);
    else if (left_is_bf16)
        hipLaunchKernelGGL((block_tail_fwd_kernel<bf, float>), grid, block, 0, s, (const bf *)left, (const float *)x, input, scale, out, n4, C / 4, hw);
    else if (x_is_bf16)
        hipLaunchKernelGGL((block_tail_fwd_kernel<float, bf>), grid, block, 0, s, (const float *)left, (const bf *)x, input, scale, out, n4, C / 4, hw);
    else
        hipLaunchKernelGGL((block_tail_fwd_kernel<float, float>), grid, block, 0, s, (const float *)left, (const float *)x, input, scale, out, n4, C / 4, hw);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int block_tail_bwd_dispatch(const float *dout, const float *scale, const void *left, void *dleft, int dleft_is_bf16, void *dx, int dx_is_bf16,
                            int64_t npix, int64_t hw, int C, hipStream_t s) {
    if (!dout || !dleft || !dx) return MS_ERR_NULL;
    if (npix < 0 || hw <= 0 || C <= 0 || C % 4 != 0 || npix % hw != 0) return MS_ERR_SHAPE;
    if (npix == 0) return MS_OK;
    const int64_t n4 = npix * (C / 4);
    const dim3 grid(tail_grid(n4)), block(256);
    using bf = unsigned short;
    if (dleft_is_bf16 && dx_is_bf16)
        hipLaunchKernelGGL((block_tail_bwd_kernel<bf, bf>), grid, block, 0, s, dout, scale, (const bf *)left, (bf *)dleft, (bf *)dx, n4, C / 4, hw);
    else if (dleft_is_bf16)
        hipLaunchKernelGGL((block_tail_bwd_kernel<bf, float>), grid, block, 0, s, dout, scale, (const bf *)left, (bf *)dleft, (float *)dx, n4, C / 4, hw);
    else if (dx_is_bf16)
        hipLaunchKernelGGL((block_tail_bwd_kernel<float, bf>), grid, block, 0, s, dout, scale, (const float *)left, (float *)dleft, (bf *)dx, n4, C / 4, hw);
    else
        hipLaunchKernelGGL((block_tail_bwd_kernel<float, float>), grid, block, 0, s, dout, scale, (const float *)left, (float *)dleft, (float *)dx, n4, C / 4, hw);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int block_head_bwd_dispatch(const float *dout, const void *dl, int dl_is_bf16, const void *dr, int dr_is_bf16, float *dinp,
                            int64_t npix, int C, hipStream_t s) {
    if (!dout || !dl || !dr || !dinp) return MS_ERR_NULL;
    if (npix < 0 || C <= 0 || C % 8 != 0) return MS_ERR_SHAPE;          // each half is a whole number of 4-channel groups
    if (npix == 0) return MS_OK;
    const int64_t n4 = npix * (C / 4);
    const dim3 grid(tail_grid(n4)), block(256);
    using bf = unsigned short;
    if (dl_is_bf16 && dr_is_bf16)
        hipLaunchKernelGGL((block_head_bwd_kernel<bf, bf>), grid, block, 0, s, dout, (const bf *)dl, (const bf *)dr, dinp, n4, C / 4);
    else if (dl_is_bf16)
        hipLaunchKernelGGL((block_head_bwd_kernel<bf, float>), grid, block, 0, s, dout, (const bf *)dl, (const float *)dr, dinp, n4, C / 4);
    else if (dr_is_bf16)
        hipLaunchKernelGGL((block_head_bwd_kernel<float, bf>), grid, block, 0, s, dout, (const float *)dl, (const bf *)dr, dinp, n4, C / 4);
    else
        hipLaunchKernelGGL((block_head_bwd_kernel<float, float>), grid, block, 0, s, dout, (const float *)dl, (const float *)dr, dinp, n4, C / 4);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
