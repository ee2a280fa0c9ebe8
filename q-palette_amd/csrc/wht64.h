// Walsh-Hadamard transform of hd = 64 * R elements (R = 16 * RT, RT in {1, 2, 4, 8}: hd = 1024 ... 8192) on the
// matrix pipe, without LDS and without barriers.  Four waves ("a quad") share one transform; wave ct owns output
// columns [16 ct, 16 ct + 16) of the [R][64] view of the vector (element i = rho * 64 + c).
//
// Why: a butterfly network costs ~25 VALU/LDS instructions per element; on the ONE compute unit a single
// transform can use, instruction issue (4-5 cycles per wave instruction) makes that microseconds.  H_hd = H_R (x) H_64
// is two small matrix products instead, and both +-1 matrices are generated in registers from lane ids:
//
//   stage 1  D1[rho][c'] = sum_c  x[rho][c] * H64[c][c']      A = x rows, straight from memory in fragment order
//                                                              (8 consecutive halves per lane), B = generated signs
//   stage 2  D2[rho'][c'] = sum_rho H_R[rho'][rho] * D1[rho][c']
//            The accumulator layout of stage 1 (lane (q, j) holds rows 4q..4q+3 of column j) IS a B operand of
//            stage 2 if the k slots of a 32-row chunk are numbered rho = 32 kc + 16 (e >> 2) + 4 q + (e & 3):
//            no data movement between the stages, the permutation only changes which signs A holds.
//            D1 goes in as an fp16 hi + lo pair (two MFMAs): fp32-grade, the result differs from an fp32 butterfly
//            network only in the last fp32 bits.  (D1 / 8 must fit fp16: |x| < 8188 — activations behind an RMSNorm are
//            orders of magnitude below; the LDS butterfly kernel has no such bound.)
//
// v_mfma_f32_16x16x32_f16 operand layouts: A lane (q = lane >> 4, r = lane & 15) = A[row r][k = 8q + e];
// B lane (q, j) = B[k = 8q + e][col j]; D lane (q, j) = D[row 4q + i][col j], i = 0..3.
#pragma once
#include <type_traits>

#include "qpal_common.h"

namespace qpal {

typedef _Float16 wht_half8 __attribute__((ext_vector_type(8)));
typedef float wht_float4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t wht_parity(uint32_t v) { return __builtin_popcount(v) & 1u; }

// 8 halves of +-1.0: slot e is negative where bit e of `neg` is set
__device__ __forceinline__ u32x4 wht_signs(uint32_t neg) {
    u32x4 f;
#pragma unroll
    for (int d = 0; d < 4; d++)
        f[d] = 0x3C003C00u | (((neg >> (2 * d)) & 1u) << 15) | (((neg >> (2 * d + 1)) & 1u) << 31);
    return f;
}

__device__ __forceinline__ u32x4 wht_flip(u32x4 f, uint32_t lo_half, uint32_t hi_half) {
    // negate slots 0..3 (dwords 0, 1) and / or slots 4..7 (dwords 2, 3)
    const uint32_t ml = lo_half ? 0x80008000u : 0u, mh = hi_half ? 0x80008000u : 0u;
    return u32x4{f[0] ^ ml, f[1] ^ ml, f[2] ^ mh, f[3] ^ mh};
}

// load_a(t, kc) -> the 8 consecutive inputs x[(16 t + (lane & 15)) * 64 + 32 kc + 8 (lane >> 4) + e] as fp16
// (pre-multiplied by SU, already rounded);  store(tp, i, elem, v): v = sum * scale (fp32) of element `elem` of the
// transform, the i-th value (0..3) of this lane in output row tile tp.
// A-operand input as an fp16 hi + lo pair (fp32-grade inputs: two MFMAs per tile in stage 1)
struct wht_hilo {
    wht_half8 hi, lo;
};

// (after_stage1: called once when every input has been loaded — the place to finish a reduction over the inputs that the
// store needs, e.g. an RMSNorm's sum of squares)
template <int RT, int TCH_MAX = 4, class LoadA, class Store, class After = void (*)()>
__device__ __forceinline__ void wht64_quad(int ct, int lane, float scale, LoadA &&load_a, Store &&store, After &&after_stage1 = [] {}) {
    using AT = decltype(load_a(0, 0));
    constexpr bool HILO = std::is_same_v<AT, wht_hilo>;
    const uint32_t q = lane >> 4, j = lane & 15;
    constexpr uint32_t M = 0x80008000u;

    // ---- stage 1: B[k = c][col = c'] = (-1)^popc(c & c'), c = 32 kc + 8 q + e, c' = 16 ct + j
    uint32_t neg = 0;
#pragma unroll
    for (uint32_t e = 0; e < 8; e++) neg |= wht_parity(e & (j & 7u)) << e;
    const uint32_t f_lane = ((q & 1u) & (j >> 3)) ^ ((q >> 1) & ((uint32_t)ct & 1u));
    u32x4 b1[2];
    b1[0] = wht_flip(wht_signs(neg), f_lane, f_lane);
    const uint32_t f1 = f_lane ^ (((uint32_t)ct >> 1) & 1u);
    b1[1] = wht_flip(wht_signs(neg), f1, f1);

    wht_float4 d1[RT];
    constexpr int TCH = RT < TCH_MAX ? RT : TCH_MAX;  // row tiles whose loads are in flight together
#pragma unroll
    for (int t0 = 0; t0 < RT; t0 += TCH) {
        AT a[TCH][2];
#pragma unroll
        for (int t = 0; t < TCH; t++) {
            a[t][0] = load_a(t0 + t, 0);
            a[t][1] = load_a(t0 + t, 1);
        }
#pragma unroll
        for (int t = 0; t < TCH; t++) {
            wht_float4 acc{0.f, 0.f, 0.f, 0.f};
            if constexpr (HILO) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][0].hi, __builtin_bit_cast(wht_half8, b1[0]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][0].lo, __builtin_bit_cast(wht_half8, b1[0]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][1].hi, __builtin_bit_cast(wht_half8, b1[1]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][1].lo, __builtin_bit_cast(wht_half8, b1[1]), acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][0], __builtin_bit_cast(wht_half8, b1[0]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][1], __builtin_bit_cast(wht_half8, b1[1]), acc, 0, 0, 0);
            }
            d1[t0 + t] = acc;
        }
    }
    after_stage1();

    // ---- stage 2 operands.  B: tiles (2 kc, 2 kc + 1) of this lane, scaled by 1/8 (exact; keeps fp16 in range), as
    // hi + lo.  A[row rho' = 16 t' + r][slot (q, e)] = (-1)^popc(rho' & rho), rho = 32 kc + 16 (e >> 2) + 4 q + (e & 3):
    //   popc parity = par((r & 3) & (e & 3)) ^ par((r >> 2) & q) ^ ((t' & 1) & (e >> 2)) ^ par((t' >> 1) & kc)
    constexpr int NKC = (RT + 1) / 2;
    wht_half8 bh[NKC], bl[NKC];
#pragma unroll
    for (int kc = 0; kc < NKC; kc++) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int t = 2 * kc + (e >> 2);
            const float v = t < RT ? d1[t < RT ? t : 0][e & 3] * 0.125f : 0.f;
            const _Float16 hi = (_Float16)v;
            bh[kc][e] = hi;
            bl[kc][e] = (_Float16)(v - (float)hi);
        }
    }
    const uint32_t r = j;  // as an A operand the low lane bits are the row
    uint32_t neg2 = 0;
#pragma unroll
    for (uint32_t e = 0; e < 8; e++) neg2 |= wht_parity((r & 3u) & (e & 3u)) << e;
    const uint32_t g_lane = wht_parity((r >> 2) & q);
    const u32x4 a_even = wht_flip(wht_signs(neg2), g_lane, g_lane);        // t' even
    const u32x4 a_odd = wht_flip(wht_signs(neg2), g_lane, g_lane ^ 1u);    // t' odd: slots 4..7 (rho bit 4) flip
#pragma unroll
    for (int tp = 0; tp < RT; tp++) {
        wht_float4 acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < NKC; kc++) {
            u32x4 a = (tp & 1) ? a_odd : a_even;
            if (__builtin_popcount((tp >> 1) & kc) & 1) a = u32x4{a[0] ^ M, a[1] ^ M, a[2] ^ M, a[3] ^ M};
            const wht_half8 ah = __builtin_bit_cast(wht_half8, a);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[kc], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[kc], acc, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
            store(tp, i, (16 * tp + 4 * (int)q + i) * 64 + 16 * ct + (int)j, acc[i] * (8.0f * scale));
    }
}


// The same transform spread over a whole workgroup (used where the other waves would only wait: the GEMV prologue at
// batch 1).  Stage 1: wave t < RT loads row tile t ONCE and produces its four column tiles (8 MFMAs) into `d1buf`
// ([RT][4][64] float4 = RT KiB * 4 in LDS); after the caller's barrier, wave w < 4 RT = (t' = w >> 2, ct = w & 3) forms
// one output tile (2 NKC MFMAs).  x is read once per workgroup instead of four times, and no wave runs more than
// ~100 instructions.  Call wht64_wg_stage1, __syncthreads(), wht64_wg_stage2.
template <int RT, class LoadA>
__device__ __forceinline__ void wht64_wg_stage1(int wave, int lane, wht_float4 *d1buf, LoadA &&load_a) {
    if (wave >= RT) return;
    const uint32_t q = lane >> 4, j = lane & 15;
    uint32_t neg = 0;
#pragma unroll
    for (uint32_t e = 0; e < 8; e++) neg |= wht_parity(e & (j & 7u)) << e;
    const u32x4 base = wht_signs(neg);
    const wht_half8 a0 = load_a(wave, 0), a1 = load_a(wave, 1);
#pragma unroll
    for (uint32_t ct = 0; ct < 4; ct++) {
        const uint32_t f0 = ((q & 1u) & (j >> 3)) ^ ((q >> 1) & (ct & 1u)), f1 = f0 ^ ((ct >> 1) & 1u);
        wht_float4 acc{0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, __builtin_bit_cast(wht_half8, wht_flip(base, f0, f0)), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(wht_half8, wht_flip(base, f1, f1)), acc, 0, 0, 0);
        d1buf[(wave * 4 + ct) * 64 + lane] = acc;
    }
}

// stage 1 on a row tile the caller has already loaded (a0 / a1: the two 32-column halves as A fragments)
__device__ __forceinline__ void wht64_wg_stage1_pre(int wave, int lane, wht_float4 *d1buf, wht_half8 a0, wht_half8 a1) {
    const uint32_t q = lane >> 4, j = lane & 15;
    uint32_t neg = 0;
#pragma unroll
    for (uint32_t e = 0; e < 8; e++) neg |= wht_parity(e & (j & 7u)) << e;
    const u32x4 base = wht_signs(neg);
#pragma unroll
    for (uint32_t ct = 0; ct < 4; ct++) {
        const uint32_t f0 = ((q & 1u) & (j >> 3)) ^ ((q >> 1) & (ct & 1u)), f1 = f0 ^ ((ct >> 1) & 1u);
        wht_float4 acc{0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, __builtin_bit_cast(wht_half8, wht_flip(base, f0, f0)), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(wht_half8, wht_flip(base, f1, f1)), acc, 0, 0, 0);
        d1buf[(wave * 4 + ct) * 64 + lane] = acc;
    }
}

template <int RT, class Store>
__device__ __forceinline__ void wht64_wg_stage2(int wave, int lane, float scale, const wht_float4 *d1buf, Store &&store) {
    if (wave >= 4 * RT) return;
    const int tp = wave >> 2, ct = wave & 3;
    const uint32_t q = lane >> 4, r = lane & 15;
    constexpr uint32_t M = 0x80008000u;
    constexpr int NKC = (RT + 1) / 2;
    uint32_t neg2 = 0;
#pragma unroll
    for (uint32_t e = 0; e < 8; e++) neg2 |= wht_parity((r & 3u) & (e & 3u)) << e;
    const uint32_t g_lane = wht_parity((r >> 2) & q);
    const u32x4 a_tp = wht_flip(wht_signs(neg2), g_lane, g_lane ^ ((uint32_t)tp & 1u));
    wht_float4 acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < NKC; kc++) {
        wht_half8 bh, bl;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int t = 2 * kc + half;
            wht_float4 d{0.f, 0.f, 0.f, 0.f};
            if (t < RT) d = d1buf[(t * 4 + ct) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float v = d[i] * 0.125f;
                const _Float16 hi = (_Float16)v;
                bh[4 * half + i] = hi;
                bl[4 * half + i] = (_Float16)(v - (float)hi);
            }
        }
        u32x4 a = a_tp;
        if (__builtin_popcount((tp >> 1) & kc) & 1) a = u32x4{a[0] ^ M, a[1] ^ M, a[2] ^ M, a[3] ^ M};
        const wht_half8 ah = __builtin_bit_cast(wht_half8, a);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) store(tp, i, (16 * tp + 4 * (int)q + i) * 64 + 16 * ct + (int)r, acc[i] * (8.0f * scale));
}

}  // namespace qpal
