// VALU / transcendental / packed / MFMA issue-rate probe for gfx950: cycles per wave-instruction per SIMD as a
// function of waves per SIMD and of the instruction-level parallelism inside one wave.
// build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

enum { K_FMA = 0, K_EXP, K_PKFMA, K_SCAN, K_SCAN_MFMA, K_MUL_EXP, K_FMA_MFMA };

template <int KIND, int ILP>
__global__ void __launch_bounds__(256) probe(float *out, long long *cyc, int iters, float a, float b) {
    float acc[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = a * (float)(threadIdx.x + i + 1) * 1e-3f;
    f4 macc = {0.f, 0.f, 0.f, 0.f};
    float dl = a * 0.01f, du = b, y = 0.0f;
    float A2[ILP], Bv[ILP], Cv[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) { A2[i] = -a * (i + 1); Bv[i] = b + i; Cv[i] = a - i; asm volatile("" : "+v"(A2[i]), "+v"(Bv[i]), "+v"(Cv[i])); }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            if constexpr (KIND == K_FMA) {
#pragma unroll
                for (int i = 0; i < ILP; ++i) acc[i] = fmaf(acc[i], a, b);
            } else if constexpr (KIND == K_EXP) {
#pragma unroll
                for (int i = 0; i < ILP; ++i) acc[i] = __builtin_amdgcn_exp2f(acc[i]);
            } else if constexpr (KIND == K_PKFMA) {
#pragma unroll
                for (int i = 0; i + 1 < ILP; i += 2) {
                    f2 v = {acc[i], acc[i + 1]}, aa = {a, a}, bb = {b, b};
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(aa), "v"(bb));
                    acc[i] = v.x; acc[i + 1] = v.y;
                }
            } else if constexpr (KIND == K_MUL_EXP) {
#pragma unroll
                for (int i = 0; i < ILP; ++i) acc[i] = __builtin_amdgcn_exp2f(acc[i] * A2[i]);
            } else if constexpr (KIND == K_SCAN || KIND == K_SCAN_MFMA) {
                // one scan position: ILP states per lane
                dl = dl * 1.0001f; du = du * 0.9999f;
#pragma unroll
                for (int i = 0; i < ILP; ++i) {
                    const float e = __builtin_amdgcn_exp2f(dl * A2[i]);
                    acc[i] = fmaf(e, acc[i], du * Bv[i]);
                    y = fmaf(Cv[i], acc[i], y);
                }
                if constexpr (KIND == K_SCAN_MFMA) { macc = __builtin_amdgcn_mfma_f32_16x16x4f32(y, 1.0f, macc, 0, 0, 0); y = 0.0f; }
            } else if constexpr (KIND == K_FMA_MFMA) {
#pragma unroll
                for (int i = 0; i < ILP; ++i) acc[i] = fmaf(acc[i], a, b);
                macc = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[0], 1.0f, macc, 0, 0, 0);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = y + macc.x + macc.y + macc.z + macc.w;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int ILP>
void run(const char *name, int per_rep_instr, float *out, long long *cyc) {
    const int iters = 2000;
    for (int k = 1; k <= 4; ++k) {
        const int blocks = 256 * k;
        hipLaunchKernelGGL((probe<KIND, ILP>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 0.5f, 0.25f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<KIND, ILP>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 0.5f, 0.25f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks * 4);
        hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        const double instr = (double)iters * 8 * per_rep_instr;
        printf("%-28s ILP %2d waves/SIMD %d: %8.3f ms, median wave %9.0f cyc, %6.2f cyc/instr/wave, %6.2f cyc/instr/SIMD (clock %.2f GHz)\n",
               name, ILP, k, ms, med, med / instr, med / instr / k, med / (ms * 1e6));
    }
}

int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 1024 * 256 * sizeof(float)); hipMalloc(&cyc, 1024 * 4 * sizeof(long long));
    run<K_FMA, 8>("v_fma_f32", 8, out, cyc);
    run<K_FMA, 2>("v_fma_f32", 2, out, cyc);
    run<K_EXP, 8>("v_exp_f32", 8, out, cyc);
    run<K_PKFMA, 8>("v_pk_fma_f32 (4 instr)", 4, out, cyc);
    run<K_MUL_EXP, 8>("v_mul+v_exp", 16, out, cyc);
    run<K_SCAN, 2>("scan body (5/state+2)", 12, out, cyc);
    run<K_SCAN, 4>("scan body (5/state+2)", 22, out, cyc);
    run<K_SCAN, 16>("scan body (5/state+2)", 82, out, cyc);
    run<K_SCAN_MFMA, 4>("scan body + 1 mfma16x16x4", 23, out, cyc);
    run<K_SCAN_MFMA, 16>("scan body + 1 mfma16x16x4", 83, out, cyc);
    run<K_FMA_MFMA, 8>("8 fma + 1 mfma16x16x4", 9, out, cyc);
    run<K_FMA_MFMA, 4>("4 fma + 1 mfma16x16x4", 5, out, cyc);
    return 0;
}
