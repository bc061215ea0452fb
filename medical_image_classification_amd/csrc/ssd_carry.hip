// Chunk-state carry of the chunked (state-space-duality) SSD evaluation for gfx950.
//   S    : (batch, chunks, F) fp32   the state each chunk adds, F = groups * N * heads_per_group * headdim flattened in the
//                                    layout the state GEMMs produce / consume: f = ((g*N + n)*hg + h)*P + p
//   d    : (batch, chunks, heads)    total decay of each chunk, per head (head = g*hg + h)
//   out  : (batch, chunks, F)        forward : out[z] = state entering chunk z   = d[z-1]*out[z-1] + S[z-1], out[0] = 0
//                                    reverse : out[z] = gradient reaching S[z]    = d[z+1]*out[z+1] + in[z+1], out[last] = 0
//                                              (the adjoint sweep: same recurrence, chunks visited last to first)
// As a GEMM this carry is a (chunks x chunks) decay matrix times the state tensor, which needs the tensor permuted head-major
// and back (two copies of a 1.6 GB tensor per scan for VFEFM's stage 0) and is memory-bound at skinny tiles.  Here every
// lane owns 4 consecutive f (one head: P % 4 == 0), walks the chunks once with the carry in registers, loads of the next
// chunks in flight: 4 B read + 4 B written per state element, the HBM floor of the operation.
#include <hip/hip_runtime.h>
#include "medscan.h"

namespace ms {

template <bool REVERSE>
__global__ void __launch_bounds__(256)
ssd_carry_kernel(const float *__restrict__ in, const float *__restrict__ d, float *__restrict__ out, int nc, int64_t F,
                 int heads, int nhp, int hg, int P) {
    const int64_t f4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (f4 >= F) return;
    const int b = blockIdx.y;
    const int head = (int)(f4 / nhp) * hg + (int)((f4 / P) % hg);           // nhp = N * hg * P
    const float *ip = in + (int64_t)b * nc * F + f4;
    float *op = out + (int64_t)b * nc * F + f4;
    const float *dp = d + (int64_t)b * nc * heads + head;
    float4 carry = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int U = 4;                                                    // chunks per trip: U loads in flight per lane
    for (int t0 = 0; t0 < nc; t0 += U) {
        float4 v[U];
        float dz[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u < nc ? t0 + u : nc - 1;
            const int z = REVERSE ? nc - 1 - t : t;
            v[u] = *reinterpret_cast<const float4 *>(ip + (int64_t)z * F);
            dz[u] = dp[(int64_t)z * heads];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (t0 + u < nc) {
                const int z = REVERSE ? nc - 1 - (t0 + u) : t0 + u;
                *reinterpret_cast<float4 *>(op + (int64_t)z * F) = carry;
                carry.x = fmaf(carry.x, dz[u], v[u].x); carry.y = fmaf(carry.y, dz[u], v[u].y);
                carry.z = fmaf(carry.z, dz[u], v[u].z); carry.w = fmaf(carry.w, dz[u], v[u].w);
            }
        }
    }
}

int ssd_carry_dispatch(const float *in, const float *d, float *out, int batch, int chunks, int groups, int N, int hg, int P,
                       int reverse, hipStream_t s) {
    if (!in || !d || !out) return MS_ERR_NULL;
    if (batch < 0 || chunks <= 0 || groups <= 0 || N <= 0 || hg <= 0 || P <= 0 || P % 4 != 0 || batch > 65535) return MS_ERR_SHAPE;
    if (batch == 0) return MS_OK;
    const int64_t F = (int64_t)groups * N * hg * P;
    if ((int64_t)N * hg * P >= (1LL << 31)) return MS_ERR_SHAPE;
    const dim3 grid((unsigned)((F / 4 + 255) / 256), (unsigned)batch);
    if (reverse) hipLaunchKernelGGL((ssd_carry_kernel<true>), grid, dim3(256), 0, s, in, d, out, chunks, F, groups * hg, N * hg * P, hg, P);
    else         hipLaunchKernelGGL((ssd_carry_kernel<false>), grid, dim3(256), 0, s, in, d, out, chunks, F, groups * hg, N * hg * P, hg, P);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
