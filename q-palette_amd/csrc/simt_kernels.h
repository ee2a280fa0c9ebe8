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

// launch geometry of simt_gemv_kernel (host side; see the kernel's header comment)
struct SimtGeometry {
    int grid, threads, dyn_lds, x_lds, split;
};

inline int simt_table_bytes(int bits, int vec) {  // = SimtCodec<bits, vec>::LDS_DWORDS * 4
    const int idx = (vec == 1 && bits <= 6) ? 2 * bits : bits, edw = vec == 4 ? 2 : 1;
    const int lmax = 15 - idx - (edw - 1), log2c = lmax < 5 ? lmax : 5;
    return ((1 << idx) << log2c) * edw * 4;
}

inline SimtGeometry simt_gemv_geometry(int m, int n, int k, int bits, int vec, int num_cu) {
    const int table = simt_table_bytes(bits, vec);
    SimtGeometry g{};
    g.x_lds = n == 1 && table + 2 * k <= 156 * 1024 && k % 8 == 0;
    g.dyn_lds = g.x_lds ? 2 * k : 0;
    const int lds = table + g.dyn_lds;
    g.threads = lds <= 20 * 1024 ? 256 : lds <= 40 * 1024 ? 512 : 1024;
    int per_cu = 160 * 1024 / (lds > 1024 ? lds : 1024);
    if (per_cu > 1024 / g.threads) per_cu = 1024 / g.threads;  // 16 waves per CU resident (<= 128 VGPRs)
    if (per_cu < 1) per_cu = 1;
    // a row per wave (halves on alternate blocks) when two rows per wave would leave resident wave slots empty; it costs a
    // sixth reduction step per row, which shows on long layers of short rows
    const int nblk = (k + 1024 * vec - 1) / (1024 * vec);
    g.split = nblk >= 2 && (m + 1) / 2 < num_cu * 16;
    const int items = g.split ? m : (m + 1) / 2, wpg = g.threads / 64;
    g.grid = (items + wpg - 1) / wpg;
    if (g.grid > num_cu * per_cu) g.grid = num_cu * per_cu;
    return g;
}

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

    // same image, any workgroup size: 16-byte chunks (C >= 4 copies of a 4-byte entry, C >= 2 of an 8-byte one, are
    // adjacent), eight table reads in flight per thread before the first LDS write
    static constexpr int CHUNKS = LDS_DWORDS / 4;
    static_assert(LOG2C + (EDW - 1) >= 2, "a 16-byte chunk holds copies of one entry");
    static __device__ __forceinline__ u32x4 chunk(const void *tab, int c) {
        const gptr<const uint16_t> l16 = as_global(static_cast<const uint16_t *>(tab));
        const gptr<const uint32_t> l32 = as_global(static_cast<const uint32_t *>(tab));
        const int e = (c * 4 / EDW) >> LOG2C;
        if constexpr (VEC == 4) {
            const uint32_t lo = l32[2 * e], hi = l32[2 * e + 1];
            return u32x4{lo, hi, lo, hi};
        } else {
            uint32_t v;
            if constexpr (VEC == 2) v = l32[e];
            else if constexpr (PAIR) v = (uint32_t)l16[e & ((1 << BITS) - 1)] | ((uint32_t)l16[e >> BITS] << 16);
            else v = l16[e];
            return u32x4{v, v, v, v};
        }
    }
    static __device__ __forceinline__ void build_chunks(uint32_t *lds, const void *tab, int tid, int nthreads) {
        for (int c0 = tid; c0 < CHUNKS; c0 += nthreads * 8) {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int c = c0 + j * nthreads;
                v[j] = chunk(tab, c < CHUNKS ? c : CHUNKS - 1);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int c = c0 + j * nthreads;
                if (c < CHUNKS) reinterpret_cast<u32x4 *>(lds)[c] = v[j];
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
                v = group_sum<32>(v);  // (DPP / permlane swaps: the same tree as xor 1, 2, 4, 8, 16, bit for bit, without the LDS round trips)
                if (t == 0 && row_ok && b < p.n) {
                    const _Float16 hv = (_Float16)v;
                    p.out[(long)b * p.m + row] = __builtin_bit_cast(uint16_t, hv);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// GEMV on the SIMT format, MI355X form.  The packed rows are tiny (k = 4096 at 3 bits per pair: 768 B), so the kernel is
// a latency problem before it is a bandwidth problem: what counts is how many independent loads are in flight per CU.
//   * one ROW PER WAVE when the row has >= 2 blocks (the two 32-lane halves take alternate blocks, one more xor step in
//     the reduction); two rows per wave otherwise (vec_sz 4 with k = 4096: one block per row);
//   * the first unit's packed words are requested BEFORE the codebook image is built, and every later unit's words while
//     the previous unit is being decoded (register double buffer);
//   * the workgroup size follows the LDS footprint (host: simt_gemv_geometry) so that small layers still spread over all
//     CUs and big tables still have >= 16 waves behind them; the grid is persistent (several workgroups per CU, rows
//     strided over waves): the image is built once per workgroup;
//   * batch 1 stages x in LDS (dynamic, after the image); larger batches read it through L1.
template <int BITS, int VEC, int NB, bool XL>
__global__ __launch_bounds__(1024) void simt_gemv_kernel(const SimtParams p, const int split_rows) {
    static_assert(!XL || NB == 1, "x is staged in LDS for batch 1 only");
    using Cd = SimtCodec<BITS, VEC>;
    __shared__ __attribute__((aligned(16))) uint32_t lut[Cd::LDS_DWORDS];
    extern __shared__ __attribute__((aligned(16))) uint16_t xs_dyn[];
    const int tid = threadIdx.x, lane = tid & 63, t = lane & 31, half = lane >> 5;
    const int nthreads = blockDim.x, wpg = nthreads >> 6;
    const uint32_t laneoff = (uint32_t)(t & (Cd::C - 1)) << (Cd::EDW == 2 ? 3 : 2);

    const long row_words = (long)p.k * BITS / 32 / VEC;
    const int nblk = (p.k + Cd::BLOCK - 1) / Cd::BLOCK;
    const int nfull = p.k / Cd::BLOCK;
    const int wtail = (p.k % Cd::BLOCK) / (32 * VEC);
    const bool split = split_rows != 0;                // wave-uniform (host: only when the row has >= 2 blocks)
    const int nrp = split ? p.m : (p.m + 1) / 2;       // wave-level work items
    const int niter = split ? (nblk + 1) / 2 : nblk;   // units per item and half
    const int gw = blockIdx.x * wpg + (tid >> 6), nw = gridDim.x * wpg;
    const gptr<const uint32_t> q = as_global(p.q);

    // unit (rp, i) of this half-wave: row, block, lane count; `false` when the half has nothing there
    auto issue = [&](int rp, int i, uint32_t (&w)[BITS]) -> bool {
        const int row = split ? rp : rp * 2 + half;
        const int blk = split ? 2 * i + half : i;
        const int W = blk < nfull ? 32 : wtail;
        const bool ok = rp < nrp && row < p.m && blk < nblk && t < W;
        if (ok) {
            const gptr<const uint32_t> src = q + (long)row * row_words + (long)blk * BITS * 32 + t;
#pragma unroll
            for (int j = 0; j < BITS; j++) w[j] = __builtin_nontemporal_load(src + (long)W * j);
        }
        return ok;
    };

    uint32_t wa[BITS], wb[BITS];
    bool oka = issue(gw, 0, wa);
    if constexpr (XL) {
        const int total = p.n * p.k;  // multiple of 8 halves
        for (int i0 = tid * 8; i0 < total; i0 += nthreads * 32) {
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int i = i0 + j * nthreads * 8;
                v[j] = *(gptr<const u32x4>)as_global(p.x + (i < total ? i : 0));
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int i = i0 + j * nthreads * 8;
                if (i < total) *reinterpret_cast<u32x4 *>(xs_dyn + i) = v[j];
            }
        }
    }
    Cd::build_chunks(lut, p.lut, tid, nthreads);
    __syncthreads();

    const gptr<const uint16_t> xg = as_global(p.x);
    auto unit = [&](int i, const uint32_t (&w)[BITS], float (&acc)[NB]) {
        const int blk = split ? 2 * i + half : i;
        const int W = blk < nfull ? 32 : wtail;
        const int e0 = blk * Cd::BLOCK + 8 * t;  // 32-bit element offsets: 64-bit ones cost a register pair per x read
        static_for<0, Cd::NGRP>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            uint32_t h[4];
            Cd::template group<g>(lut, laneoff, w, h);
            const int elem = e0 + g * 8 * W;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const int bb = b < p.n ? b : p.n - 1;
                u32x4 xv;
                if constexpr (XL) xv = *reinterpret_cast<const u32x4 *>(xs_dyn + elem);
                else xv = *(gptr<const u32x4>)(xg + (bb * p.k + elem));
                acc[b] = fdot2(h[0], xv.x, acc[b]);
                acc[b] = fdot2(h[1], xv.y, acc[b]);
                acc[b] = fdot2(h[2], xv.z, acc[b]);
                acc[b] = fdot2(h[3], xv.w, acc[b]);
            }
            // vec_sz 4 has 16 groups per unit: left alone the scheduler hoists every x read (64 registers) and spills
            if constexpr (Cd::NGRP > 8 && (g & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        });
    };

    // flattened (item, unit) sequence, two units per trip so that the double buffer needs no register copies
    int rp = gw, i = 0;
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] = 0.f;
    auto finish_row = [&]() {
#pragma unroll
        for (int b = 0; b < NB; b++) {
            float v = acc[b];
            v = group_sum<32>(v);
            if (split) v += lane_xor<32>(v);
            const int row = split ? rp : rp * 2 + half;
            if (t == 0 && (!split || half == 0) && row < p.m && b < p.n)
                p.out[(long)b * p.m + row] = __builtin_bit_cast(uint16_t, (_Float16)v);
            acc[b] = 0.f;
        }
    };
    while (rp < nrp) {  // wave-uniform
        {
            int rn = rp, in = i + 1;
            if (in >= niter) { rn = rp + nw; in = 0; }
            const bool okb = issue(rn, in, wb);
            if (oka) unit(i, wa, acc);
            if (in == 0) finish_row();
            rp = rn; i = in;
            oka = okb;  // (name reused below: wb now holds the current unit)
        }
        if (rp >= nrp) break;
        {
            int rn = rp, in = i + 1;
            if (in >= niter) { rn = rp + nw; in = 0; }
            const bool okn = issue(rn, in, wa);
            if (oka) unit(i, wb, acc);
            if (in == 0) finish_row();
            rp = rn; i = in;
            oka = okn;
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
