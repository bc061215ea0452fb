// 4-direction cross-scan / cross-merge for gfx950.  In the reference these are eager tensor ops
// (/root/reference/MedMamba.py:393-395 stack/transpose/flip/cat; :420-424,476 flips/transposes/adds):
// ~9 full-tensor HBM round trips.  Here each is ONE pass: a (b,d) plane is staged in LDS once, the
// four orderings are produced from it with coalesced global accesses on both sides.
//   xs[b,0,d,hW+w] = x[b,d,h,w]   xs[b,1,d,wH+h] = x[b,d,h,w]   xs[b,2,d,l] = xs[b,0,d,L-1-l]   xs[b,3,d,l] = xs[b,1,d,L-1-l]
//   y[b,d,hW+w]    = ((ys0[p] + ys2[L-1-p]) + ys1[q]) + ys3[L-1-q],  p = hW+w, q = wH+h     (add order of MedMamba.py:476)
// Pure data movement: bit-exact by construction.
#include <hip/hip_runtime.h>
#include "medscan.h"

namespace ms {

__global__ void __launch_bounds__(256)
cross_scan_kernel(const float *__restrict__ x, float *__restrict__ xs, int dim, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [H][W+1]
    const int L = H * W, pitch = W + 1;
    const int d = blockIdx.x % dim, b = blockIdx.x / dim;
    const float *xp = x + ((int64_t)b * dim + d) * L;
    float *o0 = xs + (((int64_t)b * 4 + 0) * dim + d) * L;
    float *o1 = xs + (((int64_t)b * 4 + 1) * dim + d) * L;
    float *o2 = xs + (((int64_t)b * 4 + 2) * dim + d) * L;
    float *o3 = xs + (((int64_t)b * 4 + 3) * dim + d) * L;
    for (int p = threadIdx.x; p < L; p += blockDim.x) {
        const float v = xp[p];
        tile[(p / W) * pitch + (p % W)] = v;
        o0[p] = v;
        o2[L - 1 - p] = v;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < L; q += blockDim.x) {
        const int w = q / H, h = q % H;
        const float v = tile[h * pitch + w];
        o1[q] = v;
        o3[L - 1 - q] = v;
    }
}

__global__ void __launch_bounds__(256)
cross_merge_kernel(const float *__restrict__ ys, float *__restrict__ y, int dim, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // two [W][H+1] planes (ys1, flipped ys3)
    const int L = H * W, pitch = H + 1;
    float *t1 = tile, *t3 = tile + W * pitch;
    const int d = blockIdx.x % dim, b = blockIdx.x / dim;
    const float *o0 = ys + (((int64_t)b * 4 + 0) * dim + d) * L;
    const float *o1 = ys + (((int64_t)b * 4 + 1) * dim + d) * L;
    const float *o2 = ys + (((int64_t)b * 4 + 2) * dim + d) * L;
    const float *o3 = ys + (((int64_t)b * 4 + 3) * dim + d) * L;
    float *yp = y + ((int64_t)b * dim + d) * L;
    for (int q = threadIdx.x; q < L; q += blockDim.x) {
        const int w = q / H, h = q % H;
        t1[w * pitch + h] = o1[q];
        t3[w * pitch + h] = o3[L - 1 - q];
    }
    __syncthreads();
    for (int p = threadIdx.x; p < L; p += blockDim.x) {
        const int h = p / W, w = p % W;
        yp[p] = ((o0[p] + o2[L - 1 - p]) + t1[w * pitch + h]) + t3[w * pitch + h];
    }
}

int cross_scan_dispatch(const float *x, float *xs, int batch, int dim, int H, int W, hipStream_t s) {
    if (!x || !xs) return MS_ERR_NULL;
    if (batch < 0 || dim <= 0 || H <= 0 || W <= 0) return MS_ERR_SHAPE;
    const size_t smem = sizeof(float) * (size_t)H * (W + 1);
    if (smem > 160 * 1024) return MS_ERR_UNSUPPORTED;
    if (batch == 0) return MS_OK;
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)cross_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(cross_scan_kernel, dim3(batch * dim), dim3(256), smem, s, x, xs, dim, H, W);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int cross_merge_dispatch(const float *ys, float *y, int batch, int dim, int H, int W, hipStream_t s) {
    if (!ys || !y) return MS_ERR_NULL;
    if (batch < 0 || dim <= 0 || H <= 0 || W <= 0) return MS_ERR_SHAPE;
    const size_t smem = sizeof(float) * 2 * (size_t)W * (H + 1);
    if (smem > 160 * 1024) return MS_ERR_UNSUPPORTED;
    if (batch == 0) return MS_OK;
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)cross_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(cross_merge_kernel, dim3(batch * dim), dim3(256), smem, s, ys, y, dim, H, W);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
