// Shared helpers of the LayerNorm-family kernels (ln.hip, ln_gate.hip) for gfx950: a pixel's D channels live on a GROUP of
// LPP = 16 / 32 / 64 lanes (so a wave holds 4 / 2 / 1 pixels at once), every lane owns V4 runs of 4 consecutive channels
// (16-byte loads / stores), and the sums over a pixel's channels are pure-VALU all-reduces: four DPP row rotations inside a
// 16-lane row, then v_permlane16_swap / v_permlane32_swap between rows.  The first versions of these kernels put one pixel on a
// whole wave with 4-byte accesses and reduced with __shfl_xor (six ds_bpermute round trips per sum): at D = 48..96 a quarter to
// a half of the lanes idled and the backward ran five dependent LDS-latency chains per pixel group -- 0.8-1.4 TB/s.
#pragma once
#include "scan_common.h"

namespace ms {

template <int LPP>
__device__ __forceinline__ float group_allsum(float v, int lane) {      // every lane of the group ends with the group's sum
    v += dpp_mov<0x128>(v); v += dpp_mov<0x124>(v); v += dpp_mov<0x122>(v); v += dpp_mov<0x121>(v);      // row_ror 8, 4, 2, 1
    if constexpr (LPP >= 32) v = xchg_add<16>(v, v, lane);
    if constexpr (LPP >= 64) v = xchg_add<32>(v, v, lane);
    return v;
}
// sum over the wave's 64 / LPP groups of the values held by lanes with equal position inside their group
template <int LPP>
__device__ __forceinline__ float across_groups(float v, int lane) {
    if constexpr (LPP <= 16) v = xchg_add<16>(v, v, lane);
    if constexpr (LPP <= 32) v = xchg_add<32>(v, v, lane);
    return v;
}

__device__ __forceinline__ float4 ld4f(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 ld4f(const unsigned short *p) {          // 4 bf16 -> 4 floats
    const uint2 r = *reinterpret_cast<const uint2 *>(p);
    return make_float4(bits_f(r.x << 16), bits_f(r.x & 0xFFFF0000u), bits_f(r.y << 16), bits_f(r.y & 0xFFFF0000u));
}
__device__ __forceinline__ void st4f(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ void st4f(unsigned short *p, float4 v) {
    uint2 r;
    r.x = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v.x) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v.y) << 16);
    r.y = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v.z) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v.w) << 16);
    *reinterpret_cast<uint2 *>(p) = r;
}
__device__ __forceinline__ float sum4(float4 v) { return (v.x + v.y) + (v.z + v.w); }

// (LPP, V4, PB) for D channels: the narrowest group that holds a row in <= 4 runs per lane; PB pixel slots in flight per wave
// so that a lane keeps ~8 loads outstanding
#define MS_LN_SUB_DISPATCH(D, CALL)                                                                        \
    if ((D) <= 64) { CALL(16, 1, 4); } else if ((D) <= 128) { CALL(32, 1, 4); } else if ((D) <= 256) { CALL(64, 1, 4); } \
    else if ((D) <= 512) { CALL(64, 2, 2); } else if ((D) <= 768) { CALL(64, 3, 1); } else { CALL(64, 4, 1); }

// ln.hip only: rows up to 2048 channels (PatchMerging2D's LayerNorm over 4 x 384 / 4 x 512 channels in front of stage 3, MedMamba.py:196-205)
#define MS_LN_SUB_DISPATCH_WIDE(D, CALL)                                                                   \
    if ((D) <= 1024) { MS_LN_SUB_DISPATCH(D, CALL) } else if ((D) <= 1536) { CALL(64, 6, 1); } else { CALL(64, 8, 1); }

static inline bool ln_aligned(const void *p, unsigned a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

}  // namespace ms
