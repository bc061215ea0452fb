// Shared pieces of the bf16 MFMA GEMM kernels (gemm.hip, linear_bwd.hip): bf16 rounding and the 16-byte operand pieces.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ms {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short f2bf(float f) {      // round-to-nearest-even; NaN stays NaN (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}

// One 8-element piece of a tile row -> 8 bf16.  `p` points at element 0 of the piece; `nvalid` (0..8) of them exist.
template <bool F32>
__device__ __forceinline__ bf16x8 load_piece(const void *base, int64_t off, int nvalid) {
    bf16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
    if (nvalid >= 8) {
        if (F32) {
            const float4 a = *reinterpret_cast<const float4 *>(static_cast<const float *>(base) + off);
            const float4 b = *reinterpret_cast<const float4 *>(static_cast<const float *>(base) + off + 4);
            r[0] = f2bf(a.x); r[1] = f2bf(a.y); r[2] = f2bf(a.z); r[3] = f2bf(a.w);
            r[4] = f2bf(b.x); r[5] = f2bf(b.y); r[6] = f2bf(b.z); r[7] = f2bf(b.w);
        } else {
            r = *reinterpret_cast<const bf16x8 *>(static_cast<const unsigned short *>(base) + off);
        }
    } else if (nvalid > 0) {
        // ragged row end: the leading half as one vector when it is whole (x_proj's 4 (R + 2N) = 140.. columns end on a
        // 16-byte boundary), single elements for the rest
        int i0 = 0;
        if (F32 && nvalid >= 4) {
            const float4 a = *reinterpret_cast<const float4 *>(static_cast<const float *>(base) + off);
            r[0] = f2bf(a.x); r[1] = f2bf(a.y); r[2] = f2bf(a.z); r[3] = f2bf(a.w);
            i0 = 4;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i >= i0 && i < nvalid)
                r[i] = F32 ? (short)f2bf(static_cast<const float *>(base)[off + i]) : (short)static_cast<const unsigned short *>(base)[off + i];
    }
    return r;
}

typedef short bf16x4 __attribute__((ext_vector_type(4)));

// runtime-typed form: `f32` selects the memory type of the operand (one uniform branch per piece)
__device__ __forceinline__ bf16x8 load_piece_rt(const void *base, int64_t off, int nvalid, bool f32) {
    return f32 ? load_piece<true>(base, off, nvalid) : load_piece<false>(base, off, nvalid);
}

}  // namespace ms
