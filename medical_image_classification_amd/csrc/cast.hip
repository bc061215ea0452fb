// bf16 working copies of the fp32 master weights for gfx950, all of them in ONE launch.
// Under bf16 autocast every weight that feeds a MIOpen convolution or a library GEMM is cast once per step
// (torch: one `aten::_to_copy` launch per weight per forward, ~90 launches of 4-5 us for MedMamba-T, plus a layout copy per
// convolution weight because MIOpen's NHWC kernels want (Co, kh, kw, Ci) memory).  Here: a table of (src, dst, shape)
// descriptors in device memory, one grid over all of them; convolution weights are written in channels_last order directly.
// Replaces the per-use casts of torch.autocast around MedMamba.py:517-527 (conv branch) and :284,326 (projections).
#include <hip/hip_runtime.h>
#include "medscan.h"

namespace ms {

// grid = (blocks_per_tensor, n_tensors); tensor t: dst[(o * taps + k) * inner + i] = bf16(src[(o * inner + i) * taps + k])
// for element index e = (o * inner + i) * taps + k < n; taps == 1: a plain elementwise cast (any shape).
__global__ void __launch_bounds__(256)
cast_bf16_multi_kernel(const MsCastDesc *__restrict__ desc) {
    const MsCastDesc d = desc[blockIdx.y];
    const float *__restrict__ src = static_cast<const float *>(d.src);
    unsigned short *__restrict__ dst = static_cast<unsigned short *>(d.dst);
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (d.taps >= 0 && d.taps <= 1) {
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < d.n; e += stride)
            dst[e] = __builtin_bit_cast(unsigned short, (__bf16)src[e]);
        return;
    }
    if (d.taps < 0) {
        // the weight of the INPUT-GRADIENT convolution: dst[(i * taps + (taps - 1 - k)) * outer + o] = src[(o * inner + i) * taps + k]
        // (taps spatially flipped, in / out channels swapped), outer = n / (inner * taps)
        const int taps = -d.taps, inner = d.inner;
        const int64_t outer = d.n / ((int64_t)inner * taps);
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < d.n; e += stride) {
            const int64_t ik = e / outer;                 // i * taps + k'
            const int64_t o = e - ik * outer;
            const int64_t i = ik / taps;
            const int k = taps - 1 - (int)(ik - i * taps);
            dst[e] = __builtin_bit_cast(unsigned short, (__bf16)src[(o * inner + i) * taps + k]);
        }
        return;
    }
    // iterate in DESTINATION order (coalesced 2-byte stores; the strided 4-byte reads hit the same lines `taps` times in a row)
    const int taps = d.taps, inner = d.inner;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < d.n; e += stride) {
        const int64_t ok = e / inner;                 // o * taps + k
        const int i = (int)(e - ok * inner);
        const int64_t o = ok / taps;
        const int k = (int)(ok - o * taps);
        dst[e] = __builtin_bit_cast(unsigned short, (__bf16)src[(o * inner + i) * taps + k]);
    }
}

int cast_bf16_multi_dispatch(const MsCastDesc *desc, int n_tensors, int blocks_per_tensor, hipStream_t s) {
    if (n_tensors < 0 || blocks_per_tensor < 1 || blocks_per_tensor > 65535 || n_tensors > 65535) return MS_ERR_SHAPE;
    if (n_tensors == 0) return MS_OK;
    if (!desc) return MS_ERR_NULL;
    hipLaunchKernelGGL(cast_bf16_multi_kernel, dim3((unsigned)blocks_per_tensor, (unsigned)n_tensors), dim3(256), 0, s, desc);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
