// Which packed-fp32 forms run at the scalar issue cost on gfx950?  Variants of the scan recurrence body, one state PAIR
// per lane, everything in registers.  build: hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize pk_rate.hip -o pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float x) { return (v2f){x, x}; }
template <int VAR, int NP2>
__global__ void __launch_bounds__(256) probe(const float *in, float *out, int iters) {
    v2f A2[NP2], h[NP2], Bv[NP2], Cv[NP2];
    for (int q = 0; q < NP2; ++q) { A2[q] = (v2f){-1.f - q, -1.5f - q}; h[q] = splat(0.f); Bv[q] = (v2f){in[q], in[q + 1]}; Cv[q] = (v2f){in[q + 2], in[q + 3]};
        asm volatile("" : "+v"(A2[q]), "+v"(Bv[q]), "+v"(Cv[q])); }
    float dl = in[threadIdx.x & 63] * 0.01f, du = in[(threadIdx.x & 63) + 1], ya = 0.f;
    v2f y2a = splat(0.f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            dl *= 1.0001f; du *= 0.9999f;
#pragma unroll
            for (int q = 0; q < NP2; ++q) {
                if constexpr (VAR == 0) {          // full packed body
                    const v2f x = splat(dl) * A2[q];
                    const v2f a = (v2f){__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                    h[q] = pk_fma(a, h[q], splat(du) * Bv[q]);
                    y2a = pk_fma(Cv[q], h[q], y2a);
                } else if constexpr (VAR == 1) {   // no exp: a = x
                    const v2f a = splat(dl) * A2[q];
                    h[q] = pk_fma(a, h[q], splat(du) * Bv[q]);
                    y2a = pk_fma(Cv[q], h[q], y2a);
                } else if constexpr (VAR == 2) {   // scalar body on the two states
                    const float a0 = __builtin_amdgcn_exp2f(dl * A2[q].x), a1 = __builtin_amdgcn_exp2f(dl * A2[q].y);
                    h[q].x = fmaf(a0, h[q].x, du * Bv[q].x); h[q].y = fmaf(a1, h[q].y, du * Bv[q].y);
                    ya = fmaf(Cv[q].x, h[q].x, ya); ya = fmaf(Cv[q].y, h[q].y, ya);
                } else if constexpr (VAR == 3) {   // packed, no broadcast operands (full pairs)
                    const v2f x = A2[q] * Bv[q];
                    const v2f a = (v2f){__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                    h[q] = pk_fma(a, h[q], Cv[q] * Bv[q]);
                    y2a = pk_fma(Cv[q], h[q], y2a);
                } else if constexpr (VAR == 4) {   // only the two pk_mul with broadcast + pk_fma chain (no exp, no y)
                    const v2f a = splat(dl) * A2[q];
                    h[q] = pk_fma(a, h[q], splat(du) * Bv[q]);
                }
            }
        }
    }
    float s = ya + y2a.x + y2a.y;
    for (int q = 0; q < NP2; ++q) s += h[q].x + h[q].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int VAR, int NP2> void run(const char *name, int n_instr, const float *in, float *out) {
    const int iters = 2000;
    for (int k = 2; k <= 4; ++k) {
        hipLaunchKernelGGL((probe<VAR, NP2>), dim3(256 * k), dim3(256), 0, 0, in, out, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<VAR, NP2>), dim3(256 * k), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ns = ms * 1e6 / ((double)iters * 8 * k);
        printf("%-44s pairs %d waves/SIMD %d: %7.3f ms, %6.1f cyc(2.3GHz) per position per SIMD (%d instr -> %.2f cyc/instr)\n", name, NP2, k, ms, ns * 2.3, n_instr, ns * 2.3 / n_instr);
    }
}
int main() {
    float *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 1024 * 256 * 4);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0.3f + 0.001f * i; hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    run<0, 1>("packed body: pk_mul,2exp,pk_mul,pk_fma,pk_fma", 2 + 6, in, out);
    run<1, 1>("packed, no exp", 2 + 4, in, out);
    run<2, 1>("scalar body (2 states)", 2 + 10, in, out);
    run<3, 1>("packed, no broadcast operands", 2 + 6, in, out);
    run<4, 1>("pk_mul(bcast) x2 + pk_fma", 2 + 3, in, out);
    run<0, 2>("packed body", 2 + 12, in, out);
    run<2, 2>("scalar body (4 states)", 2 + 20, in, out);
    run<0, 8>("packed body", 2 + 48, in, out);
    run<2, 8>("scalar body (16 states)", 2 + 80, in, out);
    return 0;
}
