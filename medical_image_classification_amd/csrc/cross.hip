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

// ---- channel-last variants: pixels (B, H*W, *) <-> the four scan sequences (B, H*W, 4, C) ---------------------------------
// The SSD blocks (CNN_Mamba.py:494-519,542-552) gather the conv output [x | B | C | dt] of every pixel into the four scan
// orders with stack/transpose/flip/cat and bring the results back with the inverse permutations and three adds.  Channel-
// last both sides, so a (pixel, column range) row moves as one coalesced segment: the scan is a gather (each output row
// reads one pixel), the merge a 4-term sum per pixel in the reference's add order -- each the other's adjoint.
//   seq[b, l, 0, c] = pix[b, l, c]                  seq[b, l, 2, c] = pix[b, L-1-l, c]
//   seq[b, l, 1, c] = pix[b, (l % H) * W + l / H, c]   seq[b, l, 3, c] = pix[b, col(L-1-l), c]
__device__ __forceinline__ int seq_pixel(int k, int l, int H, int W, int L) {
    const int t = (k & 2) ? L - 1 - l : l;
    return (k & 1) ? (t % H) * W + t / H : t;
}
__device__ __forceinline__ int pixel_seq(int k, int pix, int H, int W, int L) {          // inverse of seq_pixel
    const int t = (k & 1) ? (pix % W) * H + pix / W : pix;
    return (k & 2) ? L - 1 - t : t;
}

__global__ void __launch_bounds__(256)
cross_scan_nhwc_kernel(const float *__restrict__ pix, int64_t pps, float *__restrict__ seq, int H, int W, int C, int64_t total) {
    const int L = H * W;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const int64_t r = e / C;                      // (b * L + l) * 4 + k
        const int k = (int)(r & 3);
        const int64_t bl = r >> 2;
        const int l = (int)(bl % L);
        const int64_t b = bl / L;
        seq[e] = pix[(b * L + seq_pixel(k, l, H, W, L)) * pps + c];
    }
}

__global__ void __launch_bounds__(256)
cross_merge_nhwc_kernel(const float *__restrict__ seq, float *__restrict__ pix, int64_t pps, int H, int W, int C, int64_t total) {
    const int L = H * W;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const int64_t bp = e / C;                     // b * L + pixel
        const int p = (int)(bp % L);
        const int64_t b = bp / L;
        const float *sb = seq + b * L * 4 * C + c;
        const float v0 = sb[((int64_t)pixel_seq(0, p, H, W, L) * 4 + 0) * C], v1 = sb[((int64_t)pixel_seq(1, p, H, W, L) * 4 + 1) * C];
        const float v2 = sb[((int64_t)pixel_seq(2, p, H, W, L) * 4 + 2) * C], v3 = sb[((int64_t)pixel_seq(3, p, H, W, L) * 4 + 3) * C];
        pix[bp * pps + c] = ((v0 + v2) + v1) + v3;    // add order of CNN_Mamba.py:552 (= MedMamba.py:476)
    }
}

static unsigned grid_for(int64_t total) {
    const int64_t blocks = (total + 255) / 256;
    return (unsigned)(blocks < 65536 ? blocks : 65536);
}

int cross_scan_nhwc_dispatch(const float *pix, int64_t pps, float *seq, int batch, int H, int W, int C, hipStream_t s) {
    if (!pix || !seq) return MS_ERR_NULL;
    if (batch < 0 || H <= 0 || W <= 0 || C <= 0 || pps < C || (int64_t)H * W >= (1LL << 30)) return MS_ERR_SHAPE;
    const int64_t total = (int64_t)batch * H * W * 4 * C;
    if (total == 0) return MS_OK;
    hipLaunchKernelGGL(cross_scan_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, s, pix, pps, seq, H, W, C, total);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

int cross_merge_nhwc_dispatch(const float *seq, float *pix, int64_t pps, int batch, int H, int W, int C, hipStream_t s) {
    if (!pix || !seq) return MS_ERR_NULL;
    if (batch < 0 || H <= 0 || W <= 0 || C <= 0 || pps < C || (int64_t)H * W >= (1LL << 30)) return MS_ERR_SHAPE;
    const int64_t total = (int64_t)batch * H * W * C;
    if (total == 0) return MS_OK;
    hipLaunchKernelGGL(cross_merge_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, s, seq, pix, pps, H, W, C, total);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
