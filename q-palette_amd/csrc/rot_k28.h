// Rotation of a 14336-vector (hadK(28) (x) H_512, the down_proj input of Llama-3.1-8B) inside the GEMV's x staging, spread over
// the 16 waves of the workgroup (round 3).  Replaces a launch of its own in front of every down_proj (qpal_hadamard,
// had_mixfirst_kernel: 5.6 us inside a decode step) — every workgroup of the GEMV rotates the whole vector itself, on the matrix
// pipe, while its first weights are in flight.
//
// The reference's pipeline (lib/utils/matmul_had.py:137-148 matmul_hadU_cuda, called from lib/linear/incoherent_linear.py:336):
//   a = fp16(x) * SU;  t = fp16( H_512(a viewed [28][512]) / sqrt(14336) );  y = fp16( hadK @ t );  staged = fp16( y * post )
// with exactly these fp16 rounding points (oracle/incoherent.py left_input).  Here:
//   stage A  (waves 0..13, one 16-row tile of the [224][64] view each): u = a H_64 (x rows straight from memory as MFMA A
//            fragments, the sign matrix generated in registers — wht64.h stage 1), then H_8 over the 8 rows of a 512-block as
//            butterflies: two levels inside the lane (the 4 accumulator rows), one across lanes q ^ 1 (ds_swizzle);
//            t = fp16(. / sqrt(k)) goes to LDS as tb[column 0..511][i 0..31] (rows of 80 bytes: conflict-free 16-byte reads)
//   stage B  (all 16 waves, 4 of the 64 output tiles each): y[j][col] = sum_i hadK[j][i] t[i][col], ONE MFMA K-step (28 -> 32,
//            zero padded); fp16(fp16(y) * post) lands in the GEMV's staged x.
// tb (40 KiB) ALIASES the codebook image: the image's table entries are requested first and held in registers, and written
// after stage B (one more barrier).  fp32 accumulation throughout; differs from the separate launch (fp32 until one final
// rounding) by the reference's own rounding of t.
#pragma once
#include "wht64.h"

namespace qpal {

constexpr int kK28 = 28, kP28 = 512, kN28 = kK28 * kP28;  // 14336
constexpr int kTbRow = 80;                                 // bytes per column row of tb: 32 halves + 16 pad

// x: fp16 [14336] (global), su: fp16 [14336] or null, hadk: fp16 [28][28] (y[j] = sum_i hadk[j][i] t[i]), xs: staged x (LDS,
// fp16 [14336] + 32 zero halves), tb: >= 512 * 80 bytes of LDS (the image region), XsIndex: position of element i in xs.
// Contains two workgroup barriers; the caller adds the one after the image is in place.
// after_loads(): called by every wave once its own global loads are out (the place for the caller's less urgent requests: a
// burst of table reads in FRONT of the x loads delays the whole chain — tc_kernels.h build_image)
// The rotation's global inputs as one wave holds them (round 4: requested at the wave's first instruction from preloaded kernel
// arguments, from inline asm — tc_kernels.h — so that they are in before the first weights are requested and no wait for them
// waits for the weight stream; the codebook image's table entries likewise, by the caller).
struct RotK28Regs {
    u32x2 hk01, hk23;   // stage B's A operand (hadk row pieces; garbage where the row / piece does not exist: rot_k28_regs zeroes)
    u32x4 a0, a1;       // stage A: the two 32-column halves of this lane's row of the [224][64] view
    u32x4 s0, s1;       // the sign vector likewise (when there is one)
};
__device__ __forceinline__ void rot_k28_issue(RotK28Regs &r, const uint16_t *x, const uint16_t *su, const uint16_t *hadk, int wave, int lane) {
    const uint32_t q = lane >> 4, j = lane & 15;
    const int jrow = 16 * (wave & 1) + (int)j;
    const uint32_t hoff = (uint32_t)(((jrow < kK28 ? jrow : 0) * kK28 + 8 * (int)q) * 2);
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(r.hk01) : "v"(hoff), "s"(hadk));
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(r.hk23) : "v"(q < 3 ? hoff + 8u : hoff), "s"(hadk));
    if (wave < 14) {
        const uint32_t off0 = (uint32_t)(((16 * wave + (int)j) * 64 + 8 * (int)q) * 2);
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r.a0) : "v"(off0), "s"(x));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(r.a1) : "v"(off0), "s"(x));
        if (su) {
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r.s0) : "v"(off0), "s"(su));
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(r.s1) : "v"(off0), "s"(su));
        }
    }
}
// every value of `r` passes through an empty volatile asm: its uses are ordered behind the caller's wait
__device__ __forceinline__ void rot_k28_landed(RotK28Regs &r) {
    asm volatile("" : "+v"(r.hk01), "+v"(r.hk23), "+v"(r.a0), "+v"(r.a1), "+v"(r.s0), "+v"(r.s1));
}

template <bool PRE, class XsIndex, class AfterLoads>
__device__ __forceinline__ void rot_k28_impl(const RotK28Regs &pre_regs, bool has_su, const uint16_t *x, const uint16_t *su, const uint16_t *hadk, float pre,
                                             float post, uint16_t *xs, unsigned char *tb, int wave, int lane, XsIndex &&xs_idx, AfterLoads &&after_loads);

template <class XsIndex, class AfterLoads>
__device__ __forceinline__ void rot_k28(const uint16_t *x, const uint16_t *su, const uint16_t *hadk, float pre, float post,
                                        uint16_t *xs, unsigned char *tb, int wave, int lane, XsIndex &&xs_idx, AfterLoads &&after_loads) {
    const RotK28Regs none{};
    rot_k28_impl<false>(none, su != nullptr, x, su, hadk, pre, post, xs, tb, wave, lane, xs_idx, after_loads);
}
// the same on inputs that are already in registers (has_su: whether r.s0 / r.s1 hold a sign vector)
template <class XsIndex>
__device__ __forceinline__ void rot_k28_regs(const RotK28Regs &r, bool has_su, float pre, float post, uint16_t *xs, unsigned char *tb, int wave,
                                             int lane, XsIndex &&xs_idx) {
    rot_k28_impl<true>(r, has_su, nullptr, nullptr, nullptr, pre, post, xs, tb, wave, lane, xs_idx, [] {});
}

template <bool PRE, class XsIndex, class AfterLoads>
__device__ __forceinline__ void rot_k28_impl(const RotK28Regs &pre_regs, bool has_su, const uint16_t *x, const uint16_t *su, const uint16_t *hadk, float pre,
                                             float post, uint16_t *xs, unsigned char *tb, int wave, int lane, XsIndex &&xs_idx, AfterLoads &&after_loads) {
    const uint32_t q = lane >> 4, j = lane & 15;
    // ---- stage B's A operand: hadk[16 (wave & 1) + j][8 q + e], zero outside 28 x 28; requested now, used after the barrier
    const int jrow = 16 * (wave & 1) + (int)j;
    u32x2 hk01{0u, 0u}, hk23{0u, 0u};  // a row is 28 halves = 56 bytes: 8-byte pieces; the piece at i = 28..31 is the next row's: zero
    if constexpr (PRE) {
        if (jrow < kK28) {
            hk01 = pre_regs.hk01;
            if (q < 3) hk23 = pre_regs.hk23;
        }
    } else if (jrow < kK28) {
        const gptr<const uint16_t> row = as_global(hadk) + jrow * kK28 + 8 * (int)q;
        hk01 = *(gptr<const u32x2>)(row);
        if (q < 3) hk23 = *(gptr<const u32x2>)(row + 4);
    }
    // ---- stage A
    if (wave < 14) {
        const int off0 = (16 * wave + (int)j) * 64 + 8 * (int)q;  // row (16 wave + j) of the [224][64] view, k slots 8 q ..
        wht_half8 a0, a1, s0v, s1v;
        if constexpr (PRE) {
            a0 = __builtin_bit_cast(wht_half8, pre_regs.a0);
            a1 = __builtin_bit_cast(wht_half8, pre_regs.a1);
            s0v = __builtin_bit_cast(wht_half8, pre_regs.s0);
            s1v = __builtin_bit_cast(wht_half8, pre_regs.s1);
        } else {
            a0 = *reinterpret_cast<const wht_half8 *>(x + off0);
            a1 = *reinterpret_cast<const wht_half8 *>(x + off0 + 32);
            if (has_su) {
                s0v = *reinterpret_cast<const wht_half8 *>(su + off0);
                s1v = *reinterpret_cast<const wht_half8 *>(su + off0 + 32);
            }
        }
        after_loads();
        if (has_su) {
            a0 = a0 * s0v;
            a1 = a1 * s1v;
        }
        uint32_t neg = 0;
#pragma unroll
        for (uint32_t e = 0; e < 8; e++) neg |= wht_parity(e & (j & 7u)) << e;
        const u32x4 base = wht_signs(neg);
        const int i_row = 2 * wave + (int)(q >> 1);  // the accumulator rows 4 q + r of this lane: i = 2 wave + (q >> 1), b = 4 (q & 1) + r
#pragma unroll
        for (uint32_t ct = 0; ct < 4; ct++) {
            const uint32_t f0 = ((q & 1u) & (j >> 3)) ^ ((q >> 1) & (ct & 1u)), f1 = f0 ^ ((ct >> 1) & 1u);
            wht_float4 acc{0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, __builtin_bit_cast(wht_half8, wht_flip(base, f0, f0)), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(wht_half8, wht_flip(base, f1, f1)), acc, 0, 0, 0);
            // H_8 over b = 4 (q & 1) + r: levels r & 1 and r >> 1 in the lane, level q & 1 with the lane 16 away
            const float s0 = acc[0] + acc[1], s1 = acc[0] - acc[1], s2 = acc[2] + acc[3], s3 = acc[2] - acc[3];
            float w4[4] = {s0 + s2, s1 + s3, s0 - s2, s1 - s3};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float o = lane_xor<16>(w4[r]);
                const float t = (q & 1u) ? o - w4[r] : w4[r] + o;
                const int col = (4 * (int)(q & 1u) + r) * 64 + 16 * (int)ct + (int)j;
                *reinterpret_cast<_Float16 *>(tb + col * kTbRow + 2 * i_row) = (_Float16)(t * pre);
            }
        }
    } else {  // waves 14, 15: the k slots 28..31 of every column row are zeros (the A operand is zero there, but 0 * garbage may be NaN)
        after_loads();
        const int t2 = (wave - 14) * 64 + lane;
#pragma unroll
        for (int c = 0; c < 4; c++) *reinterpret_cast<u32x2 *>(tb + (t2 * 4 + c) * kTbRow + 56) = u32x2{0u, 0u};
    }
    __syncthreads();
    // ---- stage B: wave w forms output tiles (row tile w & 1, column tiles 4 (w >> 1) .. + 3)
    const u32x4 afrag{hk01.x, hk01.y, hk23.x, hk23.y};
#pragma unroll
    for (int tt = 0; tt < 4; tt++) {
        const int nt = 4 * (wave >> 1) + tt;
        const u32x4 b = *reinterpret_cast<const u32x4 *>(tb + (16 * nt + (int)j) * kTbRow + 16 * (int)q);
        wht_float4 acc{0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wht_half8, afrag), __builtin_bit_cast(wht_half8, b), acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int jo = 16 * (wave & 1) + 4 * (int)q + r;
            if (jo < kK28) {
                const int idx = jo * kP28 + 16 * nt + (int)j;
                xs[xs_idx(idx)] = __builtin_bit_cast(uint16_t, (_Float16)((float)(_Float16)acc[r] * post));
            }
        }
    }
    __syncthreads();  // tb is free: the caller writes the codebook image over it
}

}  // namespace qpal
