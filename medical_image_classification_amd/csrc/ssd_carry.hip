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

// DD (reverse only): also accumulate d/d decay[z] = < gradient reaching in[z], forward out[z] > into `ddecay` (zero-filled by
// the caller): the lanes that share a head (P/4 of them, a power of two here) reduce with DPP/shuffles, one atomic per
// (head segment of the wave, chunk).  As separate torch ops (multiply + reduce over the state tensor) this cost more than
// both sweeps together.
template <bool REVERSE, bool DD>
__global__ void __launch_bounds__(256)
ssd_carry_kernel(const float *__restrict__ in, const float *__restrict__ d, float *__restrict__ out,
                 const float *__restrict__ fwd_out, float *__restrict__ ddecay, int nc, int64_t F,
                 int heads, int nhp, int hg, int P, int group) {
    int64_t f4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const bool valid = f4 < F;
    if (!valid) f4 = 0;                                                     // idle lanes shadow element 0 (they take part in the shuffles)
    const int b = blockIdx.y;
    const int head = (int)(f4 / nhp) * hg + (int)((f4 / P) % hg);           // nhp = N * hg * P
    const float *ip = in + (int64_t)b * nc * F + f4;
    const float *wp = DD ? fwd_out + (int64_t)b * nc * F + f4 : nullptr;
    float *op = out + (int64_t)b * nc * F + f4;
    const float *dp = d + (int64_t)b * nc * heads + head;
    float *ddp = DD ? ddecay + (int64_t)b * nc * heads + head : nullptr;
    const int lane = threadIdx.x & 63;
    float4 carry = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int U = 4;                                                    // chunks per trip: U loads in flight per lane
    for (int t0 = 0; t0 < nc; t0 += U) {
        float4 v[U], w[U];
        float dz[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u < nc ? t0 + u : nc - 1;
            const int z = REVERSE ? nc - 1 - t : t;
            v[u] = *reinterpret_cast<const float4 *>(ip + (int64_t)z * F);
            if (DD) w[u] = *reinterpret_cast<const float4 *>(wp + (int64_t)z * F);
            dz[u] = dp[(int64_t)z * heads];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (t0 + u < nc) {                                              // uniform across the block
                const int z = REVERSE ? nc - 1 - (t0 + u) : t0 + u;
                if (valid) *reinterpret_cast<float4 *>(op + (int64_t)z * F) = carry;
                if (DD) {
                    float part = valid ? carry.x * w[u].x + carry.y * w[u].y + carry.z * w[u].z + carry.w * w[u].w : 0.0f;
                    if (group > 0) {
                        for (int m = 1; m < group; m <<= 1) part += __shfl_xor(part, m);
                        if ((lane & (group - 1)) == 0 && valid) atomicAdd(ddp + (int64_t)z * heads, part);
                    } else if (valid) {
                        atomicAdd(ddp + (int64_t)z * heads, part);
                    }
                }
                carry.x = fmaf(carry.x, dz[u], v[u].x); carry.y = fmaf(carry.y, dz[u], v[u].y);
                carry.z = fmaf(carry.z, dz[u], v[u].z); carry.w = fmaf(carry.w, dz[u], v[u].w);
            }
        }
    }
}

int ssd_carry_dispatch(const float *in, const float *d, float *out, const float *fwd_out, float *ddecay, int batch, int chunks,
                       int groups, int N, int hg, int P, int reverse, hipStream_t s) {
    if (!in || !d || !out) return MS_ERR_NULL;
    if ((fwd_out == nullptr) != (ddecay == nullptr) || (ddecay && !reverse)) return MS_ERR_NULL;
    if (batch < 0 || chunks <= 0 || groups <= 0 || N <= 0 || hg <= 0 || P <= 0 || P % 4 != 0 || batch > 65535) return MS_ERR_SHAPE;
    if (batch == 0) return MS_OK;
    const int64_t F = (int64_t)groups * N * hg * P;
    if ((int64_t)N * hg * P >= (1LL << 31)) return MS_ERR_SHAPE;
    const dim3 grid((unsigned)((F / 4 + 255) / 256), (unsigned)batch);
    const int gl = P / 4;                                                   // lanes per head segment
    const int group = (gl & (gl - 1)) == 0 ? (gl < 64 ? gl : 64) : 0;       // 0: not a power of two -> one atomic per lane
    const int heads = groups * hg, nhp = N * hg * P;
    if (ddecay)       hipLaunchKernelGGL((ssd_carry_kernel<true, true>), grid, dim3(256), 0, s, in, d, out, fwd_out, ddecay, chunks, F, heads, nhp, hg, P, group);
    else if (reverse) hipLaunchKernelGGL((ssd_carry_kernel<true, false>), grid, dim3(256), 0, s, in, d, out, fwd_out, ddecay, chunks, F, heads, nhp, hg, P, group);
    else              hipLaunchKernelGGL((ssd_carry_kernel<false, false>), grid, dim3(256), 0, s, in, d, out, fwd_out, ddecay, chunks, F, heads, nhp, hg, P, group);
    return hipGetLastError() == hipSuccess ? MS_OK : MS_ERR_LAUNCH;
}

}  // namespace ms
