// Shared device helpers for the gfx950 selective-scan kernels (wave64, lane = channel).
//
// Mapping used by every scan kernel in this directory (DESIGN.md "Kernel design"):
//   * one workgroup = one block of 64 channels of one (batch, group); lane <-> channel, so the
//     recurrence h_l = a_l*h_{l-1} + b_l runs sequentially IN REGISTERS along L with no cross-lane
//     scan at all, and B[b,g,n,l] / C[b,g,n,l] are wave-uniform -> scalar (SMEM) loads, SGPR operands;
//   * the workgroup's NS waves split the dstate axis (NPW states per wave); the per-position partial
//     sums over n are combined through LDS;
//   * activations are staged per chunk of MS_SCAN_CHUNK positions as an LDS tile [l][lane]
//     (pitch 65 floats: conflict-free both for the coalesced global<->LDS copy of a (B,D,L) tensor
//     and for the per-lane reads), which also serves channel-last tensors (pure stride change).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "medscan.h"

namespace ms {

constexpr int kCL = MS_SCAN_CHUNK;   // positions per chunk / saved state
constexpr int kPitch = 65;           // LDS tile pitch in floats
constexpr int kTile = kCL * kPitch;  // floats per LDS tile
constexpr float kLog2e = 1.4426950408889634f;

__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// softplus exactly as the reference: x <= 20 ? log1pf(expf(x)) : x
// (selective_scan_fwd_kernel.cuh:153-156; F.softplus threshold 20 in selective_scan_interface.py:112-113)
__device__ __forceinline__ float softplus_ref(float x) { return x <= 20.0f ? log1pf(expf(x)) : x; }

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Element <-> thread mapping of a [kCL positions][64 channels] tile so that consecutive threads touch
// consecutive addresses of the tensor: along L for (B,D,L) tensors, along D for channel-last ones.
template <bool LCONTIG>
__device__ __forceinline__ void tile_coord(int idx, int &l, int &dl) {
    if (LCONTIG) { l = idx % kCL; dl = idx / kCL; }
    else         { dl = idx & 63; l = idx >> 6; }
}

// global -> LDS tile; rows past `nvalid` channels are zero (so idle lanes add nothing to lane reductions), positions past
// `len` are zero-filled (the scan identity (a,b) = (1,0); selective_scan_fwd_kernel.cuh:218-222).
template <bool LCONTIG>
__device__ __forceinline__ void load_tile(float *s, const float *base, int64_t sd, int64_t sl,
                                          int nvalid, int len, int tid, int nthreads) {
#pragma unroll 4
    for (int idx = tid; idx < kCL * 64; idx += nthreads) {
        int l, dl; tile_coord<LCONTIG>(idx, l, dl);
        float v = 0.0f;
        if (l < len && dl < nvalid) v = base[dl * sd + l * sl];
        s[l * kPitch + dl] = v;
    }
}

template <bool LCONTIG>
__device__ __forceinline__ void store_tile(const float *s, float *base, int64_t sd, int64_t sl,
                                           int nvalid, int len, int tid, int nthreads) {
#pragma unroll 4
    for (int idx = tid; idx < kCL * 64; idx += nthreads) {
        int l, dl; tile_coord<LCONTIG>(idx, l, dl);
        if (l < len && dl < nvalid) base[dl * sd + l * sl] = s[l * kPitch + dl];
    }
}

// Read-only operands that are uniform across the wave (B, C) are read through the constant address
// space: hipcc then issues them on the scalar unit (s_load_dword*, results in SGPRs, no VALU/VMEM slot).
// Without this the in-kernel stores make the compiler fall back to 64-lane vector loads of one address.
typedef const float __attribute__((address_space(4))) *cfloat_ptr;
__device__ __forceinline__ cfloat_ptr as_const(const float *p) { return (cfloat_ptr)(uintptr_t)p; }

// Wave-uniform loads of NL consecutive positions l = lstart .. lstart+NL-1 of one B/C row (scalar loads).
// FULL: the positions are known to be inside the row -> straight-line, mergeable into s_load_dwordx4.
// !FULL (last, partial chunk only): positions are clamped to L-1, so nothing outside the row is ever
// touched; the clamped values only ever multiply the zero-filled u / delta' / dout of the padding.
template <int NL, bool CONTIG, bool FULL>
__device__ __forceinline__ void load_row(const float *__restrict__ row, int64_t sl, int lstart, int L, float (&v)[NL]) {
    cfloat_ptr p = as_const(row);
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int l = FULL ? lstart + j : min(lstart + j, L - 1);
        v[j] = CONTIG ? p[l] : p[l * sl];
    }
}

}  // namespace ms
