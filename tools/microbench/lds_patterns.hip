// LDS access-pattern microbenchmark for gfx950: cycles per wave-level LDS instruction for the address patterns the scan
// kernels use.  One workgroup of 1 wave per launch (no contention), 16 independent ops per iteration.
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench/lds_patterns.hip -o gpurun_out/lds_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kIters = 2000, kUnroll = 16;

template <int KIND>
__global__ void __launch_bounds__(1024) bench(const int *__restrict__ offs, long long *out, float *sink, int nwaves) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const unsigned addr = (unsigned)(uintptr_t)lds + 4u * (unsigned)(offs[lane] + (wv & 3) * 4096);   // LDS byte address
    float acc = 0.0f;
    float4 acc4 = {0, 0, 0, 0};
    long long w0 = wall_clock64();
    long long t0 = clock64();
    for (int it = 0; it < kIters; ++it) {
        if (KIND == 0) {
            float v[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) asm volatile("ds_read_b32 %0, %1" : "=v"(v[u]) : "v"(addr));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) acc += v[u];
        } else if (KIND == 1) {
            float4 v[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(addr));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) acc += v[u].x + v[u].w;
        } else if (KIND == 2) {
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(acc));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 8) {
            float v[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(v[u]) : "v"(acc));
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(v[u]) : "v"(v[u]));
            acc += v[0] * 1e-30f;
        } else if (KIND == 9) {
            float v[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) asm volatile("v_exp_f32 %0, %1" : "=v"(v[u]) : "v"(acc));
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) asm volatile("v_exp_f32 %0, %1" : "=v"(v[u]) : "v"(v[u]));
            acc += v[0] * 1e-30f;
        } else if (KIND == 4) {
            float2 v[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) asm volatile("ds_read2_b32 %0, %1 offset1:9" : "=v"(v[u]) : "v"(addr));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) acc += v[u].x + v[u].y;
        } else if (KIND == 5) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f w4 = {acc, acc, acc, acc};
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(w4));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 6) {
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) asm volatile("ds_write2_b32 %0, %1, %1 offset1:64" :: "v"(addr), "v"(acc));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            float2 v[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) asm volatile("ds_read_b64 %0, %1" : "=v"(v[u]) : "v"(addr));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) acc += v[u].x + v[u].y;
        }
    }
    long long t1 = clock64();
    long long w1 = wall_clock64();
    if (lane == 0 && wv == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
    sink[threadIdx.x] = acc + acc4.x + acc4.y + acc4.z + acc4.w;
}

struct Pat { const char *name; int kind; std::vector<int> offs; };

int main() {
    std::vector<Pat> pats;
    auto mk = [&](const char *n, int kind, auto f) { Pat p{n, kind, std::vector<int>(64)}; for (int l = 0; l < 64; ++l) p.offs[l] = f(l); pats.push_back(p); };
    mk("b32  read  consecutive (lane)", 0, [](int l) { return l; });
    mk("b32  read  all lanes same address", 0, [](int l) { return 5; });
    mk("b32  read  tile pitch 9, c=lane%8 (8 addr, bcast x8)", 0, [](int l) { return 3 * 9 + l % 8; });
    mk("b32  read  tile pitch 17, c=lane%16 (16 addr, bcast x4)", 0, [](int l) { return 3 * 17 + l % 16; });
    mk("b32  read  stride 2 (2-way conflict expected)", 0, [](int l) { return 2 * l; });
    mk("b32  read  stride 64 (64-way conflict)", 0, [](int l) { return 64 * l; });
    mk("b128 read  consecutive 16B (lane*4)", 1, [](int l) { return 4 * l; });
    mk("b128 read  all lanes same address", 1, [](int l) { return 8; });
    mk("b128 read  rows pitch 36, row=(lane/8)*2 (8 addr, bcast x8)", 1, [](int l) { return (l / 8) * 2 * 36 + 8; });
    mk("b128 read  rows pitch 36, row=(lane/16)*4 (4 addr, bcast x16)", 1, [](int l) { return (l / 16) * 4 * 36 + 8; });
    mk("b128 read  rows pitch 36, row=lane%8 interleaved (lane%8)*2", 1, [](int l) { return (l % 8) * 2 * 36 + 8; });
    mk("b128 read  lane*12 (transpose rows pitch 12)", 1, [](int l) { return 12 * l; });
    mk("b128 read  lane*8  (transpose rows pitch 8)", 1, [](int l) { return 8 * l; });
    mk("b128 read  lane*20 (transpose rows pitch 20)", 1, [](int l) { return 20 * l; });
    mk("b128 read  rows pitch 40, row=(lane/8)*2", 1, [](int l) { return (l / 8) * 2 * 40 + 8; });
    mk("b128 read  rows pitch 32+4*?: row=(lane/8), pitch 68", 1, [](int l) { return (l / 8) * 68 + 8; });
    mk("b64  read  consecutive (lane*2)", 3, [](int l) { return 2 * l; });
    mk("b64  read  rows pitch 36 row=(lane/8)*2 (bcast)", 3, [](int l) { return (l / 8) * 2 * 36 + 8; });
    mk("VALU v_fma_f32 x2 (ticks per PAIR of instructions)", 8, [](int l) { return l; });
    mk("VALU v_exp_f32 x2 (ticks per PAIR of instructions)", 9, [](int l) { return l; });
    mk("read2_b32 tile pitch 9 (two positions), c=lane%8", 4, [](int l) { return 3 * 9 + l % 8; });
    mk("read2_b32 consecutive", 4, [](int l) { return l; });
    mk("b128 write consecutive 16B (lane*4)", 5, [](int l) { return 4 * l; });
    mk("write2_b32 consecutive (+64)", 6, [](int l) { return l; });
    mk("b32  write consecutive (lane)", 2, [](int l) { return l; });
    mk("b32  write (lane/8)*12 + lane%8 (transpose write pitch 12)", 2, [](int l) { return (l / 8) * 12 + l % 8; });
    mk("b32  write tile pitch 9: (lane/8)*9 + lane%8", 2, [](int l) { return (l / 8) * 9 + l % 8; });
    mk("b32  write tile pitch 17: (lane/16)*17 + lane%16", 2, [](int l) { return (l / 16) * 17 + l % 16; });
    mk("b32  write rows pitch 36: (lane%16)*36 + lane/16", 2, [](int l) { return (l % 16) * 36 + l / 16; });
    int *d_offs; long long *d_out; float *d_sink;
    hipMalloc(&d_offs, 64 * sizeof(int)); hipMalloc(&d_out, 2 * sizeof(long long)); hipMalloc(&d_sink, 1024 * sizeof(float));
    for (int nw : {1, 4, 8, 16}) {
        printf("---- %d wave(s) per workgroup, cycles per wave-level instruction (s_memtime ticks at 100 MHz -> scaled by clock ratio is NOT applied; compare relatively) ----\n", nw);
        for (auto &p : pats) {
            hipMemcpy(d_offs, p.offs.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
            long long best = 1LL << 60, bestw = 0;
            for (int rep = 0; rep < 3; ++rep) {
                switch (p.kind) {
                    case 0: hipLaunchKernelGGL(bench<0>, dim3(1), dim3(64 * nw), 0, 0, d_offs, d_out, d_sink, nw); break;
                    case 1: hipLaunchKernelGGL(bench<1>, dim3(1), dim3(64 * nw), 0, 0, d_offs, d_out, d_sink, nw); break;
                    case 2: hipLaunchKernelGGL(bench<2>, dim3(1), dim3(64 * nw), 0, 0, d_offs, d_out, d_sink, nw); break;
                    case 3: hipLaunchKernelGGL(bench<3>, dim3(1), dim3(64 * nw), 0, 0, d_offs, d_out, d_sink, nw); break;
                    case 8: hipLaunchKernelGGL(bench<8>, dim3(1), dim3(64 * nw), 0, 0, d_offs, d_out, d_sink, nw); break;
                    case 9: hipLaunchKernelGGL(bench<9>, dim3(1), dim3(64 * nw), 0, 0, d_offs, d_out, d_sink, nw); break;
                    case 4: hipLaunchKernelGGL(bench<4>, dim3(1), dim3(64 * nw), 0, 0, d_offs, d_out, d_sink, nw); break;
                    case 5: hipLaunchKernelGGL(bench<5>, dim3(1), dim3(64 * nw), 0, 0, d_offs, d_out, d_sink, nw); break;
                    case 6: hipLaunchKernelGGL(bench<6>, dim3(1), dim3(64 * nw), 0, 0, d_offs, d_out, d_sink, nw); break;
                }
                hipDeviceSynchronize();
                long long tt[2]; hipMemcpy(tt, d_out, sizeof(tt), hipMemcpyDeviceToHost);
                const long long t = tt[0];
                if (t < best) { best = t; bestw = tt[1]; }
            }
            printf("%-70s %8.3f ticks/instr  (%.1f ns/instr by the 100 MHz wall clock; %.2f ticks per 10 ns)\n", p.name,
                   (double)best / (kIters * kUnroll), 10.0 * bestw / (kIters * kUnroll), (double)best / bestw);
        }
    }
    return 0;
}
