// The forward scan's inner loop in isolation (LDS tiles filled once, no global traffic inside the timed region):
// cycles per position per wave for the scalar (round-1) and the packed formulation, NPL = 2 / CW = 8 and NPL = 4 / CW = 16,
// at 1..4 waves per SIMD.  build: hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize scan_inner.hip -o scan_inner
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float x) { return (v2f){x, x}; }
__device__ __forceinline__ float bits_f(unsigned v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ unsigned f_bits(float v) { return __builtin_bit_cast(unsigned, v); }
template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) { return bits_f(__builtin_amdgcn_update_dpp(0u, f_bits(x), CTRL, 0xF, 0xF, true)); }
template <int S> __device__ __forceinline__ float xchg_add(float lo, float hi) {
    if constexpr (S == 32) { const auto r = __builtin_amdgcn_permlane32_swap(f_bits(lo), f_bits(hi), false, false); return bits_f(r[0]) + bits_f(r[1]); }
    else { const auto r = __builtin_amdgcn_permlane16_swap(f_bits(lo), f_bits(hi), false, false); return bits_f(r[0]) + bits_f(r[1]); }
}
template <int CW> __device__ __forceinline__ float sum_groups_scatter4(const float (&v)[4]) {
    const float a = xchg_add<32>(v[0], v[2]), b = xchg_add<32>(v[1], v[3]);
    float r = xchg_add<16>(a, b);
    if constexpr (CW == 8) r += dpp_mov<0x128>(r);
    return r;
}
constexpr int kCL = 32;
// VAR 0: packed, 1: scalar (round 1 layout [n][l] rows + separate dl / u tiles), 2: packed without the group reduction,
// 3: packed without LDS reads of B/C (register constants)
template <int NPL, int CW, int VAR>
__global__ void __launch_bounds__(256) inner(const float *in, float *out, long long *cyc, int iters) {
    constexpr int SG = 64 / CW, NP2 = NPL / 2, RP = SG * NPL + 4, kRowPitch = kCL + 4;
    __shared__ __attribute__((aligned(16))) float sB[kCL * (RP > SG * NPL * 0 + kRowPitch ? RP : kRowPitch) * 2], sC[kCL * 40 * 2];
    __shared__ __attribute__((aligned(8))) v2f sdd_[4][kCL * CW];
    __shared__ float so_[4][kCL * CW], sdl_[4][kCL * CW], su_[4][kCL * CW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane % CW, sg = lane / CW;
    v2f *sdd = sdd_[wv]; float *so = so_[wv], *sdl = sdl_[wv], *su = su_[wv];
    for (int i = threadIdx.x; i < kCL * 40 * 2; i += 256) { sB[i] = in[i & 1023] * 0.01f; sC[i] = in[(i + 7) & 1023]; }
    for (int i = lane; i < kCL * CW; i += 64) { sdd[i] = (v2f){0.01f + 0.001f * in[i & 1023], in[(i + 3) & 1023]}; sdl[i] = sdd[i].x; su[i] = sdd[i].y; }
    __syncthreads();
    v2f A2[NP2 ? NP2 : 1], h[NP2 ? NP2 : 1]; float A1[NPL], h1[NPL];
    for (int i = 0; i < NPL; ++i) { A1[i] = -1.0f - i - sg; h1[i] = 0.f; if (NP2) { A2[i / 2][i % 2] = A1[i]; h[i / 2][i % 2] = 0.f; } }
    const float Dv = in[lane];
    v2f pdd[4], pB[4][NP2 ? NP2 : 1], pC[4][NP2 ? NP2 : 1];
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll (VAR == 5 ? 8 : 2)
        for (int lb = 0; lb < kCL; lb += 4) {
            float y[4];
            if constexpr (VAR == 1) {
                float dl_[4], du_[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { dl_[j] = sdl[(lb + j) * CW + c]; const float uu = su[(lb + j) * CW + c]; du_[j] = dl_[j] * uu; y[j] = Dv * uu; }
#pragma unroll
                for (int i = 0; i < NPL; ++i) {
                    const float4 Bv = *reinterpret_cast<const float4 *>(sB + (sg * NPL + i) * kRowPitch + lb);
                    const float4 Cv = *reinterpret_cast<const float4 *>(sC + (sg * NPL + i) * kRowPitch + lb);
                    const float Bq[4] = {Bv.x, Bv.y, Bv.z, Bv.w}, Cq[4] = {Cv.x, Cv.y, Cv.z, Cv.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a = __builtin_amdgcn_exp2f(dl_[j] * A1[i]);
                        h1[i] = fmaf(a, h1[i], du_[j] * Bq[j]);
                        y[j] = fmaf(Cq[j], h1[i], y[j]);
                    }
                }
            } else if constexpr (VAR == 4) {
                // operands of the NEXT batch are read from LDS before this batch is computed
                static_assert(NP2 >= 1, "");
                v2f ddn[4], Bn[4][NP2 ? NP2 : 1], Cn[4][NP2 ? NP2 : 1];
                if (lb == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { pdd[j] = sdd[j * CW + c];
#pragma unroll
                        for (int q = 0; q < NP2; ++q) { pB[j][q] = *reinterpret_cast<const v2f *>(sB + j * RP + sg * NPL + 2 * q); pC[j][q] = *reinterpret_cast<const v2f *>(sC + j * RP + sg * NPL + 2 * q); } }
                }
                const int ln = (lb + 4) % kCL;
#pragma unroll
                for (int j = 0; j < 4; ++j) { ddn[j] = sdd[(ln + j) * CW + c];
#pragma unroll
                    for (int q = 0; q < NP2; ++q) { Bn[j][q] = *reinterpret_cast<const v2f *>(sB + (ln + j) * RP + sg * NPL + 2 * q); Cn[j][q] = *reinterpret_cast<const v2f *>(sC + (ln + j) * RP + sg * NPL + 2 * q); } }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v2f y2 = splat(0.f);
#pragma unroll
                    for (int q = 0; q < NP2; ++q) {
                        const v2f x = splat(pdd[j].x) * A2[q];
                        const v2f a = (v2f){__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                        h[q] = pk_fma(a, h[q], splat(pdd[j].y) * pB[j][q]);
                        y2 = q == 0 ? pC[j][q] * h[q] : pk_fma(pC[j][q], h[q], y2);
                    }
                    y[j] = y2.x + y2.y;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { pdd[j] = ddn[j];
#pragma unroll
                    for (int q = 0; q < NP2; ++q) { pB[j][q] = Bn[j][q]; pC[j][q] = Cn[j][q]; } }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v2f dd = sdd[(lb + j) * CW + c];
                    const float *bp = sB + (lb + j) * RP + sg * NPL, *cp = sC + (lb + j) * RP + sg * NPL;
                    v2f y2 = splat(0.f);
#pragma unroll
                    for (int q = 0; q < NP2; ++q) {
                        const v2f Bv = VAR == 3 ? A2[q] : *reinterpret_cast<const v2f *>(bp + 2 * q);
                        const v2f Cv = VAR == 3 ? A2[q] : *reinterpret_cast<const v2f *>(cp + 2 * q);
                        const v2f x = splat(dd.x) * A2[q];
                        const v2f a = (v2f){__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                        h[q] = pk_fma(a, h[q], splat(dd.y) * Bv);
                        y2 = q == 0 ? Cv * h[q] : pk_fma(Cv, h[q], y2);
                    }
                    y[j] = y2.x + y2.y;
                }
            }
            if constexpr (VAR == 2) { so[(lb + (lane >> 4)) * CW + c] = y[0] + y[1] + y[2] + y[3]; }
            else {
                const float yt = sum_groups_scatter4<CW>(y);
                if (CW == 16 || (lane & 8) == 0) so[(lb + (lane >> 4)) * CW + c] = yt;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = so[lane];
    for (int i = 0; i < NPL; ++i) s += h1[i] + (NP2 ? h[i / 2][i % 2] : 0.f);
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wv] = t1 - t0;
}
template <int NPL, int CW, int VAR> void run(const char *name, const float *in, float *out, long long *cyc) {
    const int iters = 200;
    for (int k = 1; k <= 4; ++k) {
        const int blocks = 256 * k;
        hipLaunchKernelGGL((inner<NPL, CW, VAR>), dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((inner<NPL, CW, VAR>), dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per_pos_ns = ms * 1e6 / ((double)iters * kCL * k);     // per wave-position per SIMD
        printf("%-34s NPL %d CW %2d waves/SIMD %d: %7.3f ms  %6.1f ns = %6.1f cyc(2.3GHz) per wave-position per SIMD, %5.2f per lane-state\n",
               name, NPL, CW, k, ms, per_pos_ns, per_pos_ns * 2.3, per_pos_ns * 2.3 / NPL);
    }
}
int main() {
    float *in, *out; long long *cyc;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 4096 * 8);
    std::vector<float> h(4096); for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 37) % 101) / 101.f - 0.5f;
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<2, 8, 1>("scalar (round 1)", in, out, cyc);
    run<2, 8, 0>("packed", in, out, cyc);
    run<2, 8, 2>("packed, no group reduction", in, out, cyc);
    run<2, 8, 3>("packed, B/C from registers", in, out, cyc);
    run<2, 8, 4>("packed, LDS one batch ahead", in, out, cyc);
    run<2, 8, 5>("packed, full unroll", in, out, cyc);
    run<4, 16, 1>("scalar (round 1)", in, out, cyc);
    run<4, 16, 0>("packed", in, out, cyc);
    run<4, 16, 3>("packed, B/C from registers", in, out, cyc);
    run<4, 16, 4>("packed, LDS one batch ahead", in, out, cyc);
    run<4, 16, 5>("packed, full unroll", in, out, cyc);
    return 0;
}
