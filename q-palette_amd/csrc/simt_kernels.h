// GEMV / dequant kernels for the reference's "SIMT" packed formats (row-major rows, 32-lane blocks):
//   vec_sz 1  : sq_gemm_fp16 / pack_dequant_kbit_store   kernels/sq-cuda-kernels/gemm_routines.cu:393-622
//   vec_sz 2,4: vq_pack_gemm_fp16 / vq_pack_dequant_kbit_store  kernels/vq-cuda-kernels/src/gemm_routines.cu:1913-2120
// Layout (lib/quantizer/pack_op.py:288-335, quant_op.py:15-78): a row is cut into blocks of 32 lanes;
// lane t owns NGRP = 4*VEC groups of 8 consecutive weights at  blk*B + g*8*W + 8t  (W = 32, or the
// lane count of a trailing partial block; B = 256*NGRP); its 32 codes are packed LSB-first into BITS
// u32 words, word j stored at  blk*BITS*32 + t + W*j.
//
// wave64 mapping: the two 32-lane halves of a wave work on two different output rows (so vec_sz 4
// with one block per row still fills the wave); the per-row reduction is a 5-step xor butterfly
// inside each half.  fp32 accumulation (the reference accumulates in fp16), one rounding to fp16.
#pragma once
#include "qpal_common.h"

namespace qpal {

struct SimtParams {
    uint16_t *out;      // gemv: fp16 [n][m]; dequant: fp16 [m][k]
    const uint32_t *q;  // packed codes [m][BITS*k/32/VEC]
    const uint16_t *x;  // fp16 [n][k]
    const void *lut;    // fp16 [2^BITS][VEC]
    int n, m, k;
};

template <int BITS, int VEC>
struct SimtCodec {
    static constexpr bool PAIR = (VEC == 1 && BITS <= 6);
    static constexpr int EDW = (VEC == 4) ? 2 : 1;  // dwords per table entry
    static constexpr int IDXBITS = PAIR ? 2 * BITS : BITS;
    static constexpr int LOG2C_MAX = 15 - IDXBITS - (EDW - 1);
    static constexpr int LOG2C = LOG2C_MAX < 5 ? LOG2C_MAX : 5;
    static constexpr int C = 1 << LOG2C;
    static constexpr int LDS_DWORDS = (1 << IDXBITS) * C * EDW;
    static_assert(LOG2C >= 1, "table too large");
    static constexpr int NGRP = 4 * VEC;
    static constexpr int BLOCK = 256 * NGRP;  // elements per full block

    static __device__ __forceinline__ void build(uint32_t *lds, const void *tab, int tid, int nthreads) {
        const uint16_t *__restrict__ l16 = static_cast<const uint16_t *>(tab);
        const uint32_t *__restrict__ l32 = static_cast<const uint32_t *>(tab);
        // entry e, copy c at dword (e*C + c)*EDW
        for (int i = tid; i < (1 << IDXBITS) * C; i += nthreads) {
            const int e = i >> LOG2C;
            if constexpr (VEC == 4) {
                reinterpret_cast<u32x2 *>(lds)[i] = u32x2{l32[2 * e], l32[2 * e + 1]};
            } else if constexpr (VEC == 2) {
                lds[i] = l32[e];
            } else if constexpr (PAIR) {
                lds[i] = (uint32_t)l16[e & ((1 << BITS) - 1)] | ((uint32_t)l16[e >> BITS] << 16);
            } else {
                lds[i] = l16[e];
            }
        }
    }

    template <int POS, int NB_>
    static __device__ __forceinline__ uint32_t index(const uint32_t (&w)[BITS]) {
        if constexpr ((POS & 31) + NB_ <= 32) return __builtin_amdgcn_ubfe(w[POS >> 5], POS & 31, NB_);
        else return __builtin_amdgcn_ubfe(ext32<POS>(w), 0, NB_);
    }

    // the 8 halves (4 dwords) of group G of this lane
    template <int G>
    static __device__ __forceinline__ void group(const uint32_t *lds, uint32_t laneoff, const uint32_t (&w)[BITS],
                                                 uint32_t (&h)[4]) {
        const char *base = reinterpret_cast<const char *>(lds);
        if constexpr (VEC == 4) {
            static_for<0, 2>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                const uint32_t e = index<(2 * G + q) * BITS, BITS>(w);
                const u32x2 v = *reinterpret_cast<const u32x2 *>(base + ((e << (LOG2C + 3)) | laneoff));
                h[2 * q] = v.x;
                h[2 * q + 1] = v.y;
            });
        } else if constexpr (VEC == 2) {
            static_for<0, 4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                const uint32_t e = index<(4 * G + q) * BITS, BITS>(w);
                h[q] = *reinterpret_cast<const uint32_t *>(base + ((e << (LOG2C + 2)) | laneoff));
            });
        } else if constexpr (PAIR) {
            static_for<0, 4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                const uint32_t e = index<(8 * G + 2 * q) * BITS, 2 * BITS>(w);
                h[q] = *reinterpret_cast<const uint32_t *>(base + ((e << (LOG2C + 2)) | laneoff));
            });
        } else {
            static_for<0, 4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                const uint32_t e0 = index<(8 * G + 2 * q) * BITS, BITS>(w);
                const uint32_t e1 = index<(8 * G + 2 * q + 1) * BITS, BITS>(w);
                const uint32_t lo = *reinterpret_cast<const uint32_t *>(base + ((e0 << (LOG2C + 2)) | laneoff));
                const uint32_t hi = *reinterpret_cast<const uint32_t *>(base + ((e1 << (LOG2C + 2)) | laneoff));
                h[q] = lo | (hi << 16);
            });
        }
    }
};

// MODE 0: gemv (fp16 out[n][m]); MODE 1: dequant (fp16 out[m][k])
template <int BITS, int VEC, int NB, int MODE>
__global__ __launch_bounds__(1024) void simt_kernel(const SimtParams p) {
    using Cd = SimtCodec<BITS, VEC>;
    __shared__ __attribute__((aligned(16))) uint32_t lut[Cd::LDS_DWORDS];
    const int tid = threadIdx.x, lane = tid & 63, t = lane & 31, half = lane >> 5;
    const uint32_t laneoff = (uint32_t)(t & (Cd::C - 1)) << (Cd::EDW == 2 ? 3 : 2);
    Cd::build(lut, p.lut, tid, 1024);
    __syncthreads();

    const long row_words = (long)p.k * BITS / 32 / VEC;
    const int nblk = (p.k + Cd::BLOCK - 1) / Cd::BLOCK;
    const int nfull = p.k / Cd::BLOCK;
    const int wtail = (p.k % Cd::BLOCK) / (32 * VEC);
    const int gw = blockIdx.x * 16 + (tid >> 6), nw = gridDim.x * 16;

    for (int rp = gw; rp * 2 < p.m; rp += nw) {
        const int row = rp * 2 + half;
        const bool row_ok = row < p.m;
        float acc[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) acc[b] = 0.f;
        for (int blk = 0; blk < nblk; blk++) {
            const int W = blk < nfull ? 32 : wtail;
            if (row_ok && t < W) {
                uint32_t w[BITS];
                const uint32_t *src = p.q + (long)row * row_words + (long)blk * BITS * 32 + t;
#pragma unroll
                for (int j = 0; j < BITS; j++) w[j] = __builtin_nontemporal_load(src + (long)W * j);
                const long e0 = (long)blk * Cd::BLOCK + 8 * t;
                static_for<0, Cd::NGRP>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    uint32_t h[4];
                    Cd::template group<g>(lut, laneoff, w, h);
                    const long elem = e0 + (long)g * 8 * W;
                    if constexpr (MODE == 1) {
                        *reinterpret_cast<u32x4 *>(p.out + (long)row * p.k + elem) = u32x4{h[0], h[1], h[2], h[3]};
                    } else {
#pragma unroll
                        for (int b = 0; b < NB; b++) {
                            const int bb = b < p.n ? b : p.n - 1;
                            const u32x4 xv = *reinterpret_cast<const u32x4 *>(p.x + (long)bb * p.k + elem);
                            acc[b] = fdot2(h[0], xv.x, acc[b]);
                            acc[b] = fdot2(h[1], xv.y, acc[b]);
                            acc[b] = fdot2(h[2], xv.z, acc[b]);
                            acc[b] = fdot2(h[3], xv.w, acc[b]);
                        }
                    }
                });
            }
        }
        if constexpr (MODE == 0) {
#pragma unroll
            for (int b = 0; b < NB; b++) {
                float v = acc[b];
                v = wave_xor_add(v, 1);
                v = wave_xor_add(v, 2);
                v = wave_xor_add(v, 4);
                v = wave_xor_add(v, 8);
                v = wave_xor_add(v, 16);
                if (t == 0 && row_ok && b < p.n) {
                    const _Float16 hv = (_Float16)v;
                    p.out[(long)b * p.m + row] = __builtin_bit_cast(uint16_t, hv);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Tensor-core format -> SIMT format re-pack (load-time; VQLinearPackSIMT.gen_layer_from_info).
// One thread per code; dst must be zeroed (done by the C-ABI wrapper).  vec in {1, 2}.
template <int kUnused = 0>
__global__ void tc_to_simt_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, int m, int k, int bits,
                                  int vec) {
    const long ncodes = (long)m * k / vec;
    const int per_tile = 256 / vec;          // codes per 16x16 tile
    const int per_lane = 8 / vec;            // codes per (reference lane, tile)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ncodes; i += (long)gridDim.x * blockDim.x) {
        // source position i enumerates (sr, sc, lane, ksub, msub, q) exactly as memory does
        long r = i;
        const int q = (int)(r % per_lane); r /= per_lane;
        const int msub = (int)(r & 1); r >>= 1;
        const int ksub = (int)(r & 1); r >>= 1;
        const int lane = (int)(r & 31); r >>= 5;
        const int nsc = k / 32;
        const int sc = (int)(r % nsc);
        const int sr = (int)(r / nsc);
        (void)per_tile;
        // read the code: LE bit stream, `bits` per code
        const long sbit = i * bits;
        const uint64_t two = (uint64_t)src[sbit >> 5] | ((uint64_t)src[((sbit >> 5) + 1 < (long)m * k * bits / 32 / vec) ? (sbit >> 5) + 1 : (sbit >> 5)] << 32);
        const uint32_t code = (uint32_t)(two >> (sbit & 31)) & ((1u << bits) - 1u);
        // matrix position of the code's first weight
        const int j = vec == 1 ? (q >> 1) : q, e = vec == 1 ? (q & 1) : 0;
        const int row = sr * 32 + msub * 16 + (lane >> 2) + 8 * (j & 1);
        const int col = sc * 32 + ksub * 16 + 2 * (lane & 3) + 8 * (j >> 1) + e;
        // SIMT position
        const int ngrp = 4 * vec, B = 256 * ngrp;
        const int blk = col / B, inb = col % B;
        const int W = (blk < k / B) ? 32 : (k % B) / (32 * vec);
        const int g = inb / (8 * W), t = (inb % (8 * W)) / 8, off = inb % 8;
        const int c = (g * 8 + off) / vec;                 // code index inside the lane (0..31)
        const long dbit = (long)c * bits;
        const long row_words = (long)k * bits / 32 / vec;
        uint32_t *drow = dst + (long)row * row_words + (long)blk * bits * 32 + t;
        const int wj = (int)(dbit >> 5), sh = (int)(dbit & 31);
        atomicOr(drow + (long)W * wj, code << sh);
        if (sh + bits > 32) atomicOr(drow + (long)W * (wj + 1), code >> (32 - sh));
    }
}

}  // namespace qpal
