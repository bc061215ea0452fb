// Adam update of ALL parameters of a model in ONE launch for gfx950 -- `optimizer.step()` of the reference's training loop
// (`optim.Adam(net.parameters(), lr=0.0001)`, /root/reference/train.py:62,76; ddp_train.py:137,165): no weight decay, no amsgrad.
// torch's fused Adam walks the 355 tensors of MedMamba-T in 8 multi_tensor_apply launches of ~75 workgroups each (its per-launch
// metadata holds four pointers per tensor): 0.34 ms for a 405 MB update that the memory system moves in 0.1 ms.  Here the stable
// pointers (parameter, exp_avg, exp_avg_sq) and the block -> (tensor, chunk) map live in device memory, built once; only the
// gradient pointers -- fresh tensors every backward pass -- travel as kernel arguments (one pointer per tensor: 448 fit).
// Arithmetic = torch's `adam_math` (ATen/native/cuda/fused_adam_utils.cuh) for fp32, maximize = False, amsgrad = False:
//   m = m + (1 - b1) (g - m);  v = b2 v + (1 - b2) g g;  p -= (step_size m) / (sqrt(v) / sqrt(bc2) + eps),  step_size = lr / bc1
#include <hip/hip_runtime.h>
#include "medscan.h"

namespace ms {

constexpr int kAdamThreads = 256, kAdamVec = 4;
static_assert(MS_ADAM_CHUNK == kAdamThreads * kAdamVec * 4, "a block = 256 threads x 4 float4");

struct AdamGrads { const float *g[MS_ADAM_MAX_TENSORS]; };

// b1c = 1 - beta1, b2c = 1 - beta2 (formed in double on the host, like torch's double-typed betas)
__device__ __forceinline__ void adam_one(float &p, float &m, float &v, float g, float step_size, float bc2_sqrt, float b1c, float b2,
                                         float b2c, float eps) {
    m = m + b1c * (g - m);
    v = b2 * v + b2c * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - (step_size * m) / denom;
}

__global__ void __launch_bounds__(kAdamThreads)
adam_multi_kernel(const MsAdamDesc *__restrict__ desc, const int2 *__restrict__ blocks, const AdamGrads grads, float step_size,
                  float bc2_sqrt, float b1c, float b2, float b2c, float eps) {
    const int2 bt = blocks[blockIdx.x];                      // (tensor, chunk)
    const MsAdamDesc d = desc[bt.x];
    const float *__restrict__ g = grads.g[bt.x];
    const int64_t e0 = (int64_t)bt.y * MS_ADAM_CHUNK;
    const int64_t left = d.n - e0;
    float *p = d.p + e0, *m = d.m + e0, *v = d.v + e0;
    g += e0;
    const bool vec = (((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)g) & 15) == 0;
    if (vec && left >= MS_ADAM_CHUNK) {
        float4 rp[kAdamVec], rm[kAdamVec], rv[kAdamVec], rg[kAdamVec];
#pragma unroll
        for (int k = 0; k < kAdamVec; ++k) {                 // 16 loads in flight per thread
            const int o = (k * kAdamThreads + threadIdx.x) * 4;
            rp[k] = *reinterpret_cast<const float4 *>(p + o); rm[k] = *reinterpret_cast<const float4 *>(m + o);
            rv[k] = *reinterpret_cast<const float4 *>(v + o); rg[k] = *reinterpret_cast<const float4 *>(g + o);
        }
#pragma unroll
        for (int k = 0; k < kAdamVec; ++k) {
            const int o = (k * kAdamThreads + threadIdx.x) * 4;
            adam_one(rp[k].x, rm[k].x, rv[k].x, rg[k].x, step_size, bc2_sqrt, b1c, b2, b2c, eps);
            adam_one(rp[k].y, rm[k].y, rv[k].y, rg[k].y, step_size, bc2_sqrt, b1c, b2, b2c, eps);
            adam_one(rp[k].z, rm[k].z, rv[k].z, rg[k].z, step_size, bc2_sqrt, b1c, b2, b2c, eps);
            adam_one(rp[k].w, rm[k].w, rv[k].w, rg[k].w, step_size, bc2_sqrt, b1c, b2, b2c, eps);
            *reinterpret_cast<float4 *>(p + o) = rp[k]; *reinterpret_cast<float4 *>(m + o) = rm[k];
            *reinterpret_cast<float4 *>(v + o) = rv[k];
        }
        return;
    }
    const int cnt = (int)(left < MS_ADAM_CHUNK ? left : MS_ADAM_CHUNK);      // a tensor's last chunk, or unaligned views
    for (int o = threadIdx.x; o < cnt; o += kAdamThreads) {
        float pp = p[o], mm = m[o], vv = v[o];
        adam_one(pp, mm, vv, g[o], step_size, bc2_sqrt, b1c, b2, b2c, eps);
        p[o] = pp; m[o] = mm; v[o] = vv;
    }
}

int adam_multi_dispatch(const MsAdamDesc *desc, const int32_t *blocks, int n_blocks, const void *const *grads, int n_tensors,
                        float step_size, float bc2_sqrt, float one_minus_beta1, float beta2, float one_minus_beta2, float eps,
                        hipStream_t s) {
    if (n_tensors < 0 || n_tensors > MS_ADAM_MAX_TENSORS || n_blocks < 0) return MS_ERR_SHAPE;
    if (n_tensors == 0 || n_blocks == 0) return MS_OK;
    if (!desc || !blocks || !grads) return MS_ERR_NULL;
    AdamGrads ag;
    for (int i = 0; i < n_tensors; ++i) {
        if (!grads[i]) return MS_ERR_NULL;
        ag.g[i] = static_cast<const float *>(grads[i]);
    }
    for (int i = n_tensors; i < MS_ADAM_MAX_TENSORS; ++i) ag.g[i] = nullptr;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)n_blocks), dim3(kAdamThreads), 0, s, desc, reinterpret_cast<const int2 *>(blocks), ag,
                       step_size, bc2_sqrt, one_minus_beta1, beta2, one_minus_beta2, eps);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
