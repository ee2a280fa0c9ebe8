// Shared device helpers for the gfx950 (MI355X, CDNA4) dequant-matmul kernels.  wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "qpal.h"

namespace qpal {

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x3a __attribute__((ext_vector_type(3), aligned(4)));
typedef uint32_t u32x2a __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;
constexpr int kNumCU = 256;  // MI355X: 8 XCDs x 32 CUs

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// Pointers that a kernel reads from MEMORY (the phase table of a chain launch) are generic to the compiler: their loads
// become flat_load, which counts in BOTH vmcnt and lgkmcnt and completes out of order — every LDS wait of the decode then
// waits for the weight stream as well (measured: 7x slower).  Everything the kernels touch through such pointers lives in
// global memory: say so.
template <class T>
using gptr = __attribute__((address_space(1))) T *;
template <class T>
__device__ __forceinline__ gptr<T> as_global(T *p) {
    return (gptr<T>)p;
}

// Streamed-once weights: KV (or `bits`) consecutive dwords per lane, 4-byte aligned, non-temporal.
template <int NW>
__device__ __forceinline__ void load_words_nt(const uint32_t *__restrict__ pg, uint32_t (&w)[NW]) {
    const gptr<const uint32_t> p = as_global(pg);
    constexpr int Q = NW / 4, R = NW % 4;
    static_for<0, Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const u32x4a v = __builtin_nontemporal_load((gptr<const u32x4a>)(p + 4 * q));
        w[4 * q + 0] = v.x;
        w[4 * q + 1] = v.y;
        w[4 * q + 2] = v.z;
        w[4 * q + 3] = v.w;
    });
    if constexpr (R == 3) {
        const u32x3a v = __builtin_nontemporal_load((gptr<const u32x3a>)(p + 4 * Q));
        w[4 * Q + 0] = v.x;
        w[4 * Q + 1] = v.y;
        w[4 * Q + 2] = v.z;
    } else if constexpr (R == 2) {
        const u32x2a v = __builtin_nontemporal_load((gptr<const u32x2a>)(p + 4 * Q));
        w[4 * Q + 0] = v.x;
        w[4 * Q + 1] = v.y;
    } else if constexpr (R == 1) {
        w[4 * Q] = __builtin_nontemporal_load(p + 4 * Q);
    }
}

// 32 bits of the lane's little-endian word array starting at compile-time bit POS (reads past the
// end as zero: only ever the don't-care bits above a 16-bit window).  WIDTH: how many of the low result bits the caller
// looks at — when they all lie inside one dword the extract is a plain v_lshrrev_b32 (or nothing), which issues at full
// rate on gfx950; v_alignbit_b32, like every three-operand integer op, at half rate (profiles/r02_valu_rate2.txt).
template <int POS, int NW, int WIDTH = 32>
__device__ __forceinline__ uint32_t ext32(const uint32_t (&w)[NW]) {
    static_assert(POS >= 0, "negative bit position");
    constexpr int i = POS >> 5, sh = POS & 31;
    const uint32_t lo = (i < NW) ? w[i < NW ? i : 0] : 0u;
    if constexpr (sh == 0) {
        return lo;
#ifndef QPAL_NO_NARROW_EXT
    } else if constexpr (sh + WIDTH <= 32) {
        return lo >> sh;
#endif
    } else {
        const uint32_t hi = (i + 1 < NW) ? w[(i + 1 < NW) ? i + 1 : 0] : 0u;
        return __builtin_amdgcn_alignbit(hi, lo, sh);
    }
}
template <int POS, int WIDTH, int NW>
__device__ __forceinline__ uint32_t extw(const uint32_t (&w)[NW]) {
    return ext32<POS, NW, WIDTH>(w);
}

// value of lane+1 inside each row of 16 lanes (wraps 15 -> 0): one DPP move, no LDS traffic.
__device__ __forceinline__ uint32_t row16_next(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x12F /* row_ror:15 */, 0xF, 0xF, true);
}

__device__ __forceinline__ float fdot2(uint32_t w, uint32_t x, float acc) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, w), __builtin_bit_cast(h2_t, x), acc, false);
}

__device__ __forceinline__ float wave_xor_add(float v, int mask) {
    return v + __shfl_xor(v, mask, 64);
}

// Value of lane ^ MASK as ONE VALU instruction (DPP) where the hardware has one — `__shfl_xor` compiles to `ds_bpermute_b32`, an
// LDS-pipe operation with the full LDS round trip (round 4: the GEMV epilogue's four dependent exchange rounds were 0.5 us of every
// launch, queued behind the other waves' codebook gathers).  MASK 1, 2: quad_perm; 7, 15: row_half_mirror / row_mirror (inside a
// reduction tree they do what xor 4 / xor 8 do); 16, 32: the gfx950 permlane swaps; anything else falls back to the permute.
template <int MASK>
__device__ __forceinline__ float lane_xor(float v) {
    const int i = __builtin_bit_cast(int, v);
    if constexpr (MASK == 1) return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(i, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true));
    else if constexpr (MASK == 2) return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(i, 0x4E /* quad_perm:[2,3,0,1] */, 0xF, 0xF, true));
    else if constexpr (MASK == 7) return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(i, 0x141 /* row_half_mirror */, 0xF, 0xF, true));
    else if constexpr (MASK == 15) return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(i, 0x140 /* row_mirror */, 0xF, 0xF, true));
    else if constexpr (MASK == 16) {
        // gfx950 v_permlane16_swap: rows 1 / 3 of the first operand trade places with rows 0 / 2 of the second.  With v in both:
        // r[0] = [row0, row0, row2, row2], r[1] = [row1, row1, row3, row3] -> lane ^ 16 is r[1] in the even rows, r[0] in the odd ones
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)i, (unsigned)i, false, false);
        return __builtin_bit_cast(float, (__lane_id() & 16) ? r[0] : r[1]);
    } else if constexpr (MASK == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)i, (unsigned)i, false, false);  // (likewise for the wave's halves)
        return __builtin_bit_cast(float, (__lane_id() & 32) ? r[0] : r[1]);
    } else return __shfl_xor(v, MASK, 64);
}
// Sum over the 64 lanes of a wave, the same value in every lane: four DPP steps inside the rows of 16, then the four row totals
// through scalar registers (v_readlane) — ~11 VALU instructions against six dependent LDS round trips of the shuffle tree.
// (The summation ORDER differs from the xor tree's: not for results that must be bit-identical to an earlier build.)
// sum / max over the aligned group of G lanes a lane belongs to, the same value in all of them: DPP only up to G = 16
template <int G>
__device__ __forceinline__ float group_sum(float v) {
    static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "a power of two");
    if constexpr (G >= 2) v += lane_xor<1>(v);
    if constexpr (G >= 4) v += lane_xor<2>(v);
    if constexpr (G >= 8) v += lane_xor<7>(v);
    if constexpr (G >= 16) v += lane_xor<15>(v);
    if constexpr (G >= 32) v += lane_xor<16>(v);
    if constexpr (G >= 64) v += lane_xor<32>(v);
    return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
    static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16, "inside one row of 16 lanes");
    if constexpr (G >= 2) v = fmaxf(v, lane_xor<1>(v));
    if constexpr (G >= 4) v = fmaxf(v, lane_xor<2>(v));
    if constexpr (G >= 8) v = fmaxf(v, lane_xor<7>(v));
    if constexpr (G >= 16) v = fmaxf(v, lane_xor<15>(v));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = group_max<16>(v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += lane_xor<1>(v);
    v += lane_xor<2>(v);
    v += lane_xor<7>(v);
    v += lane_xor<15>(v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

}  // namespace qpal
