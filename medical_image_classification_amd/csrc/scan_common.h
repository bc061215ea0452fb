// Shared device helpers for the gfx950 selective-scan kernels (wave64).
//
// Work mapping:
//   * one WAVE = CW channels of one (batch, group) x all dstate states, CW = 16 or 8.  lane = sg*CW + c:
//     c = lane % CW is the channel, sg = lane / CW one of SG = 64/CW state groups; a lane owns
//     NPL = dstate/SG states of its channel.  (CW = 8 doubles the number of waves and halves the per-lane
//     state: used when 16-channel waves would not fill the chip, and always by the backward.)  The recurrence
//     h_l = a_l*h_{l-1} + b_l runs sequentially IN REGISTERS along L -- no cross-lane scan; sums over the state axis
//     are 2-3 step VALU exchanges (v_permlane32_swap / v_permlane16_swap / DPP); sums over the wave's channels
//     (backward: dB, dC) go through a wave-private LDS transpose.
//   * waves exchange no data in the forward.  A workgroup holds the waves whose channel blocks share a 128-byte
//     line of the channel-last tensors and keeps them on the same chunk with one barrier (fetch each line once).
//   * per chunk of MS_SCAN_CHUNK positions a wave stages u / delta' (/ dout) as [l][c] tiles (pitch CW+1)
//     and the chunk's B/C rows as [n][l] (pitch 36: the state groups' ds_read_b128 hit disjoint bank quads),
//     and prefetches the NEXT chunk into registers (use-free loads: nothing touches them until the next staging).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "medscan.h"

namespace ms {

constexpr int kCL = MS_SCAN_CHUNK;   // positions per chunk / saved state
constexpr int kRowPitch = kCL + 4;   // B/C row pitch (floats, 16-byte aligned rows)
constexpr float kLog2e = 1.4426950408889634f;

__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// Packed fp32: on gfx950 v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 issue in the 4 cycles of their scalar forms and do two
// lanes' worth of work (measured, tools/microbench/valu_rate.hip: v_fma_f32 4.3 cycles per wave-instruction per SIMD,
// v_pk_fma_f32 4.4, v_exp_f32 8.2) -- the kernels are VALU-issue bound, so a lane's states are kept as explicit PAIRS
// (ext_vector_type(2) in an aligned register pair; a scalar operand is broadcast with op_sel, no v_mov).  The SLP
// vectorizer stays off (-fno-slp-vectorize): pairs it forms from unrelated scalars need repacking moves.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float x) { return (v2f){x, x}; }
__device__ __forceinline__ v2f exp2_pk(v2f x) { return (v2f){exp2_fast(x.x), exp2_fast(x.y)}; }
// states per lane padded to whole pairs, and the column of state n in the [position][column] B/C row tiles
// (lane group sg owns columns sg*NPLp .. sg*NPLp + NPL - 1; an odd NPL leaves one always-zero pad column per group)
template <int NPL> constexpr int npl_padded() { return (NPL + 1) / 2 * 2; }
template <int NPL> __device__ __forceinline__ int state_col(int n) { return NPL % 2 == 0 ? n : (n / NPL) * npl_padded<NPL>() + n % NPL; }

// softplus with the reference's definition, x <= 20 ? log1p(exp(x)) : x
// (selective_scan_fwd_kernel.cuh:153-156; F.softplus threshold 20 in selective_scan_interface.py:112-113),
// evaluated with the hardware exp2/log2 and Kahan's compensated log1p: log1p(e) = log(1+e) * e / ((1+e) - 1).
// ~10 VALU ops instead of two libm calls (which cost as much per element as the whole 16-state recurrence);
// relative error ~3e-7 over the whole range, including tiny e where log(1+e) alone would lose everything.
__device__ __forceinline__ float softplus_ref(float x) {
    const float e = exp2_fast(x * kLog2e);
    const float u = 1.0f + e, dnm = u - 1.0f;
    const float lg = __builtin_amdgcn_logf(u) * 0.6931471805599453f;
    const float r = dnm == 0.0f ? e : lg * (e * __builtin_amdgcn_rcpf(dnm));
    return x <= 20.0f ? r : x;
}

// softplus'(x) = sigmoid(x) = 1 - exp(-softplus(x)), from the staged delta' (no second tile for the raw delta);
// series below 2^-6 so the subtraction never cancels.
__device__ __forceinline__ float sigmoid_from_softplus(float sp) {
    const float direct = 1.0f - exp2_fast(-sp * kLog2e);
    const float series = sp * (1.0f - sp * (0.5f - sp * (1.0f / 6.0f - sp * (1.0f / 24.0f))));
    return sp < 0.015625f ? series : direct;
}

__device__ __forceinline__ float bits_f(unsigned v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ unsigned f_bits(float v) { return __builtin_bit_cast(unsigned, v); }

// The compiler must not move LDS accesses of one lane across the point where another lane of the same
// wave produced/consumes the data; the hardware executes a wave's LDS instructions in order.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Sequence index -> memory position.  mode < 0: identity (plain sequences).  mode 0..3: the pixel visited at
// step l by direction `mode` of SS2D's cross-scan over an H x W map (MedMamba.py:393-395):
//   0: l    1: (l % H)*W + l / H    2: L-1-l    3: map1(L-1-l)
// mode 4..7 (MS_SCAN_LATTICE): sub-lattice k = mode - 4 of FusionMamba's stride-2 scan (cross.py:139-190), see medscan.h; then
// H holds the INNER count of the sequence order (map_w/2 for the row-major lattices 0, 2; map_h/2 for the column-major 1, 3),
// invH its reciprocal and W the full map width.
struct PosMap {
    int mode, H, W, L;
    float invH;
    const int *tab;        // SS2D mode: LDS table of the current chunk's kCL positions (filled by fill_table)
    int tab_base;          // first sequence index the table covers
    // SS2D-mode setup of group g: the four cross-scan directions, or the four stride-2 sub-lattices
    __device__ __forceinline__ void setup(int g, int map_h, int map_w, int L_, bool lattice) {
        L = L_; tab = nullptr; tab_base = 0;
        if (!lattice) { mode = g & 3; H = map_h; W = map_w; invH = 1.0f / (float)map_h; return; }
        mode = 4 + (g & 3);
        H = (g & 1) ? map_h / 2 : map_w / 2;
        W = map_w; invH = 1.0f / (float)H;
    }
    __device__ __forceinline__ void fill_table(int *t, int lbase, int lane) {
        if (lane < kCL) t[lane] = (*this)(min(lbase + lane, L - 1));
        tab = t; tab_base = lbase;
    }
    __device__ __forceinline__ int operator()(int l) const {
        if (mode < 0) return l;
        if (mode >= 4) {
            const int q = (int)(((float)l + 0.5f) * invH);      // l / inner, exact for l < 2^22
            const int r = l - __mul24(q, H);
            const int odd = mode & 1, hi = (mode >> 1) & 1;
            const int row = odd ? 2 * r + 1 : 2 * q;
            const int col = (odd ? 2 * q : 2 * r) + hi;
            return __mul24(row, W) + col;
        }
        int t = (mode & 2) ? L - 1 - l : l;
        if (mode & 1) {
            const int w = (int)(((float)t + 0.5f) * invH);      // exact for t < 2^22
            t = __mul24(t - __mul24(w, H), W) + w;           // all factors < 2^22 (validated)
        }
        return t;
    }
};

// position * stride.  v_mul_lo_u32 is a quarter-rate instruction; the channel-last modes (whose sequence lengths and
// sequence strides the host validates to be < 2^24) use the full-rate 24-bit multiply instead.
template <bool SMALL>
__device__ __forceinline__ int pos_times(int pos, int stride) { return SMALL ? __mul24(pos, stride) : pos * stride; }

// kernel addressing modes
constexpr int kModeBDL = 0;      // activations contiguous along L (reference (B,D,L) layout); B/C rows contiguous along L
constexpr int kModeCL = 1;       // channel-last activations, B/C rows contiguous along L
constexpr int kModeSS2D = 2;     // channel-last activations indexed by pixel through PosMap, B/C rows contiguous along n

// One [kCL positions][CW channels] activation tile moved by the 64 lanes of a wave:
// global -> registers (asynchronous until first use), registers -> LDS, LDS -> global.
// (B,D,L) tensors: lane owns position lane%32 of channels lane/32 + 2k (128-byte row segments);
// channel-last tensors: lane owns channel lane%CW of positions lane/CW + (64/CW)k.
// Channels past `nvalid` and positions past `len` read as zero (the scan identity (a,b) = (1,0),
// selective_scan_fwd_kernel.cuh:218-222; idle lanes then add nothing to lane sums).
// `base` points at (batch, group, first channel of the wave), position 0; `lbase` is the chunk's first position.
template <int MODE, int CW>
struct TileIO {
    static constexpr bool LCONTIG = MODE == kModeBDL;
    // LDS tile pitch (floats).  Channel-last tiles: lane -> (l = lane / CW, c = lane % CW) lands on address `lane`:
    // conflict-free at pitch CW (pitch CW + 1 wrapped 64..70 onto banks 0..6: a 2-way conflict on every put / store).
    // (B,D,L) tiles: lane -> position lane % 32 of two channels: needs an odd pitch.
    static constexpr int kPitch = LCONTIG ? CW + 1 : CW;
    static constexpr int kTile = kCL * kPitch;
    static constexpr int NE = kCL * CW / 64;                     // elements per lane
    static constexpr int STEP = LCONTIG ? 64 / kCL : 64 / CW;    // channels (LCONTIG) or positions per k
    int l0_, c0_;          // this lane's first (position, channel); one geometry serves every tensor
    __device__ __forceinline__ explicit TileIO(int lane) {
        if (LCONTIG) { l0_ = lane % kCL; c0_ = lane / kCL; } else { c0_ = lane % CW; l0_ = lane / CW; }
    }
    __device__ __forceinline__ int soff(int k) const { return l0_ * kPitch + c0_ + k * STEP * (LCONTIG ? 1 : kPitch); }
    __device__ __forceinline__ int lk(int k) const { return LCONTIG ? l0_ : l0_ + k * STEP; }
    __device__ __forceinline__ int ck(int k) const { return LCONTIG ? c0_ + k * STEP : c0_; }
    // strides are 32-bit element counts (validated on the host); channel-last modes have a unit channel stride
    __device__ __forceinline__ uint32_t goff(int k, int sd, int sl, int lbase, const PosMap &pm) const {
        const int pos = MODE == kModeSS2D ? pm.tab[lk(k)] : lbase + lk(k);
        return (uint32_t)(ck(k) * (LCONTIG ? sd : 1) + pos_times<!LCONTIG>(pos, sl)) * 4u;
    }
    __device__ __forceinline__ bool ok(int k, int nvalid, int len) const { return lk(k) < len && ck(k) < nvalid; }
    // branch-free and use-free: out-of-range elements read the tile's first element (always valid); NOTHING consumes the
    // loaded registers here -- the zeroing of out-of-range elements happens in put()/put_delta() one chunk later, so no
    // s_waitcnt can be scheduled next to the loads and the prefetch really overlaps the chunk being computed
    __device__ __forceinline__ void fetch(float (&r)[NE], const float *base, int sd, int sl, int lbase,
                                          const PosMap &pm, int nvalid, int len) const {
        const char *b = reinterpret_cast<const char *>(base);
        const uint32_t safe = first_valid(sd, sl, lbase, pm);
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const uint32_t off = goff(k, sd, sl, lbase, pm);
            r[k] = *reinterpret_cast<const float *>(b + (ok(k, nvalid, len) ? off : safe));
        }
    }
    // byte offset of an element that is always inside the tensor: (first position of the chunk, channel 0)
    __device__ __forceinline__ uint32_t first_valid(int sd, int sl, int lbase, const PosMap &pm) const {
        const int pos = MODE == kModeSS2D ? pm.tab[0] : lbase;
        return (uint32_t)pos_times<!LCONTIG>(pos, sl) * 4u;
    }
    // registers -> LDS; channels past `nvalid` and positions past `len` become zero; `r` is zeroed likewise
    __device__ __forceinline__ void put(float *s, float (&r)[NE], int nvalid, int len) const {
#pragma unroll
        for (int k = 0; k < NE; ++k) { r[k] = ok(k, nvalid, len) ? r[k] : 0.0f; s[soff(k)] = r[k]; }
    }
    // delta tile: bias + softplus applied once per element on the way into LDS (sp_mask = all ones / zero:
    // a bit-select instead of a branch per element)
    // pre: the values are already activated (MS_SCAN_DELTA_ACTIVATED) -- wave-uniform, so a real branch skips the work
    __device__ __forceinline__ void put_delta(float *s, const float (&r)[NE], const float *sbias, unsigned sp_mask,
                                              int nvalid, int len, bool pre = false) const {
        if (pre) {
#pragma unroll
            for (int k = 0; k < NE; ++k) s[soff(k)] = ok(k, nvalid, len) ? r[k] : 0.0f;
            return;
        }
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const float raw = r[k] + sbias[ck(k)];
            const float v = bits_f((f_bits(softplus_ref(raw)) & sp_mask) | (f_bits(raw) & ~sp_mask));
            s[soff(k)] = ok(k, nvalid, len) ? v : 0.0f;
        }
    }
    // per-lane accumulators over the elements a lane owns: one per channel it touches (NE in the (B,D,L) layout where
    // the channel changes with k, a single one in the channel-last layouts)
    static constexpr int NA = LCONTIG ? NE : 1;
    static __device__ __forceinline__ int ak(int k) { return LCONTIG ? k : 0; }
    // backward: ddelta = d delta' * softplus'(delta + bias), the derivative recovered from the staged delta' tile
    // (sp_mask = 0: no softplus, factor 1); acc collects the stored values (ddelta_bias partial sums of this lane)
    template <bool ACC = false>
    __device__ __forceinline__ void store_ddelta(const float *s, const float *sdl, unsigned sp_mask, float *base, int sd, int sl,
                                                 int lbase, const PosMap &pm, int nvalid, int len, float (&acc)[NA]) const {
        char *b = reinterpret_cast<char *>(base);
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const float f = bits_f((f_bits(sigmoid_from_softplus(sdl[soff(k)])) & sp_mask) | (f_bits(1.0f) & ~sp_mask));
            const float v = s[soff(k)] * f;
            if (ok(k, nvalid, len)) {
                float *o = reinterpret_cast<float *>(b + goff(k, sd, sl, lbase, pm));
                *o = ACC ? *o + v : v; acc[ak(k)] += v;
            }
        }
    }
    // ACC: add to what is there (MS_SCAN_ACCUMULATE); each element is read and written by this thread only
    template <bool ACC = false>
    __device__ __forceinline__ void store(const float *s, float *base, int sd, int sl, int lbase,
                                          const PosMap &pm, int nvalid, int len) const {
        char *b = reinterpret_cast<char *>(base);
#pragma unroll
        for (int k = 0; k < NE; ++k)
            if (ok(k, nvalid, len)) {
                float *o = reinterpret_cast<float *>(b + goff(k, sd, sl, lbase, pm));
                *o = ACC ? *o + s[soff(k)] : s[soff(k)];
            }
    }
};

// The chunk's B (or C) rows: NP (padded) states x kCL positions, LDS layout [n][kRowPitch]; rows past the real
// dstate and positions past `len` are zero.  The tile is SHARED by the NW waves of a workgroup (they serve the same
// (batch, group), hence the same rows): thread tid owns elements idx = tid + 64*NW*k.  Rows contiguous along L:
// l = idx % 32, n = idx / 32 (128-byte segments).  SS2D mode (rows contiguous along n, one projection row per pixel):
// n = idx % NP, l = idx / NP (NP*4-byte segments).
template <int MODE, int NP, int NW>
struct RowIO {
    static constexpr bool NCONTIG = MODE == kModeSS2D;
    static constexpr int NT = 64 * NW;                           // threads sharing the tile (the workgroup)
    static constexpr int NE = (NP * kCL + NT - 1) / NT;          // elements per thread
    int tid_;
    __device__ __forceinline__ explicit RowIO(int tid) : tid_(tid) {}
    __device__ __forceinline__ int idx(int k) const { return tid_ + NT * k; }
    __device__ __forceinline__ bool in(int k) const { return (NP * kCL) % NT == 0 || idx(k) < NP * kCL; }
    __device__ __forceinline__ int nk(int k) const { return NCONTIG ? idx(k) % NP : (idx(k) / kCL) % NP; }
    __device__ __forceinline__ int lk(int k) const { return NCONTIG ? (idx(k) / NP) % kCL : idx(k) % kCL; }
    __device__ __forceinline__ uint32_t goff(int k, int sn, int sl, int lbase, const PosMap &pm) const {
        const int pos = NCONTIG ? pm.tab[lk(k)] : lbase + lk(k);
        return (uint32_t)(nk(k) * (NCONTIG ? 1 : sn) + pos_times<MODE != kModeBDL>(pos, sl)) * 4u;
    }
    // branch-free and use-free like TileIO::fetch: every offset (including its LDS position lookup) is formed
    // unconditionally, out-of-range elements are redirected to (state 0, first position); put() zeroes them
    __device__ __forceinline__ void fetch(float (&r)[NE], const float *base, int sn, int sl, int lbase,
                                          const PosMap &pm, int N, int len) const {
        const char *b = reinterpret_cast<const char *>(base);
        const uint32_t safe = (uint32_t)pos_times<MODE != kModeBDL>(NCONTIG ? pm.tab[0] : lbase, sl) * 4u;
        uint32_t off[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) off[k] = goff(k, sn, sl, lbase, pm);
#pragma unroll
        for (int k = 0; k < NE; ++k)
            r[k] = *reinterpret_cast<const float *>(b + ((in(k) && nk(k) < N && lk(k) < len) ? off[k] : safe));
    }
    // SS2D mode, "all directions" (MS_SCAN_BC_MAP(4)): the state axis is four slices of `nd` states, slice j read through
    // direction j's position table (tabs[j]) from the SAME nd columns of the row tensor -- the SSD blocks' concatenation of
    // the four directions' B / C (CNN_Mamba.py:506-519) without materialising it.
    __device__ __forceinline__ void fetch_dirs(float (&r)[NE], const float *base, int sl, const int (*tabs)[kCL], int nd, int N,
                                               int len) const {
        static_assert(NCONTIG, "fetch_dirs: SS2D mode only");
        const char *b = reinterpret_cast<const char *>(base);
        const uint32_t safe = (uint32_t)__mul24(tabs[0][0], sl) * 4u;
        uint32_t off[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int n = nk(k);
            const int j = min(n / nd, 3);
            off[k] = (uint32_t)(n - j * nd + __mul24(tabs[j][lk(k)], sl)) * 4u;
        }
#pragma unroll
        for (int k = 0; k < NE; ++k)
            r[k] = *reinterpret_cast<const float *>(b + ((in(k) && nk(k) < N && lk(k) < len) ? off[k] : safe));
    }
    __device__ __forceinline__ void put(float *s, const float (&r)[NE], int N, int len) const {
#pragma unroll
        for (int k = 0; k < NE; ++k)
            if (in(k)) s[nk(k) * kRowPitch + lk(k)] = (nk(k) < N && lk(k) < len) ? r[k] : 0.0f;
    }
    // [position][state column] layout (row pitch RP floats): a lane's state PAIR of one position is one ds_read_b64
    // (four states one ds_read_b128); the pad columns of an odd NPL are zeroed once by the kernel
    template <int NPL, int RP>
    __device__ __forceinline__ void put_t(float *s, const float (&r)[NE], int N, int len) const {
#pragma unroll
        for (int k = 0; k < NE; ++k)
            if (in(k)) s[lk(k) * RP + state_col<NPL>(nk(k))] = (nk(k) < N && lk(k) < len) ? r[k] : 0.0f;
    }
};

// 4 consecutive positions of one staged B/C row: one ds_read_b128 (4 distinct addresses per wave, one per
// state group, 16 lanes each broadcast).
__device__ __forceinline__ void row4(const float *srow, int lb, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4 *>(srow + lb);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}

// ---- cross-lane exchange-and-add (pure VALU: v_permlane32/16_swap, DPP) ------------------------------
// Lanes with bit S clear keep `lo`, the others keep `hi`; each adds its partner's (lane ^ S) copy of the
// value it keeps.  Building block of the butterfly reduce-scatters.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return bits_f(__builtin_amdgcn_update_dpp(0u, f_bits(x), CTRL, 0xF, 0xF, true));
}
template <int S>
__device__ __forceinline__ float xchg_add(float lo, float hi, int lane) {
    if constexpr (S == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(f_bits(lo), f_bits(hi), false, false);
        return bits_f(r[0]) + bits_f(r[1]);
    } else if constexpr (S == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(f_bits(lo), f_bits(hi), false, false);
        return bits_f(r[0]) + bits_f(r[1]);
    } else {
        // DPP row_ror:n: destination lane j reads source lane (j - n) mod 16 of its row.
        // S=8: ror:8 both ways; S=4: ror:12 reads j+4, ror:4 reads j-4; S=2/1: quad_perm [2,3,0,1] / [1,0,3,2]
        constexpr int UP = S == 8 ? 0x128 : S == 4 ? 0x12C : S == 2 ? 0x4E : 0xB1;
        constexpr int DN = S == 8 ? 0x128 : S == 4 ? 0x124 : S == 2 ? 0x4E : 0xB1;
        const float a = lo + dpp_mov<UP>(lo);     // valid where bit S is clear (partner = lane + S)
        const float b = hi + dpp_mov<DN>(hi);     // valid where bit S is set   (partner = lane - S)
        return (lane & S) ? b : a;
    }
}

// Sum over the SG = 64/CW state groups of 4 values, scattered: on return the lanes with
// group_slot(lane) == j (and is_group_owner) hold the total of v[j].
template <int CW>
__device__ __forceinline__ float sum_groups_scatter4(const float (&v)[4], int lane) {
    const float a = xchg_add<32>(v[0], v[2], lane);
    const float b = xchg_add<32>(v[1], v[3], lane);
    float r = xchg_add<16>(a, b, lane);
    if constexpr (CW == 8) r += dpp_mov<0x128>(r);          // third group bit (lane bit 3): plain butterfly add
    return r;
}
// Sum over the wave's 8 channel lanes (lane bits 0-2) of the 8 (position, state) values of a batch, scattered: lane c ends up
// with the total of v[c].  Pure VALU (DPP row_ror / quad_perm adds + selects, 21 instructions): replaces a transpose through a
// wave-private LDS tile (16 ds_write_b32 + 4 ds_read_b128 and two dependent LDS round trips per batch and tensor).
// The lane-bit-2 step needs no select: two DPP adds write disjoint bank sets of one register (banks 0, 2 = lanes with bit 2
// clear read lane + 4 through row_ror:12; banks 1, 3 read lane - 4 through row_ror:4).  Inline assembly: the builtin only offers
// the move form.  s_nop 1 = the two wait states a DPP read of a just-written VGPR needs (the compiler's hazard recognizer does
// not look inside asm blocks).
__device__ __forceinline__ float chan_scatter8(const float (&v)[8], int lane) {
    float a0, a1, a2, a3;
    // one block for the four (v[q], v[q+4]) pairs: a single s_nop covers the freshest input, the DPP adds are independent
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %4, %4 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %1, %5, %5 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %2, %6, %6 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %3, %7, %7 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %0, %8, %8 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
                 "v_add_f32_dpp %1, %9, %9 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
                 "v_add_f32_dpp %2, %10, %10 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
                 "v_add_f32_dpp %3, %11, %11 row_ror:4 row_mask:0xf bank_mask:0xa"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
    const float b0 = xchg_add<2>(a0, a2, lane), b1 = xchg_add<2>(a1, a3, lane);
    return xchg_add<1>(b0, b1, lane);
}

template <int CW> __device__ __forceinline__ int group_slot(int lane) { return lane >> 4; }          // = lane bits 5,4
template <int CW> __device__ __forceinline__ bool is_group_owner(int lane) { return CW == 16 || (lane & 8) == 0; }

}  // namespace ms
