// Fused decode+GEMV and decode-to-fp16 kernels for the "tensor-core tile order" packed formats
// (TCQ trellis streams and VQ/SQ LUT codes), written for gfx950 wave64.
//
// Replaces, for TCQ, kernel_decompress_gemm{,_comb,_combt} / kernel_decompress{,_comb,_combt} and
// load_reg_cs<R> (reference: kernels/tcq-kernels/src/inference.cu:204-1819) and, for VQ/SQ, the
// kernels of kernels/vq-tensor-kernels/src/inference.cu:205-1108.
//
// Mapping chosen for CDNA4 (not the reference's warp32 / mma.m16n8k16 tiling):
//  * Both formats store, per 32x32 "supertile", 32 reference-lanes x R u16 (or u32) words in
//    mma-fragment order.  One wave-lane owns TWO consecutive reference lanes (always a whole number
//    of dwords, 4-byte aligned also for odd R), so a supertile is exactly one 16-lane DPP row and a
//    wave covers 4 supertiles along K per "step": 32 rows x 128 cols, 64 weights per lane, one
//    contiguous 64*NW*4-byte span per wave.
//  * TCQ: the 16 look-ahead bits a lane needs from its successor in the tail-biting stream come from
//    lane+1 of the same DPP row (row_ror:15; the wrap 15->0 is the tail bite): one v_mov_dpp, no LDS.
//    Each 16-bit trellis state is cut straight out of the raw dwords with one v_alignbit_b32 (bits
//    above the window are don't-cares: h = s*(s+1) mod 2^16 only sees the low 16 bits).
//  * Codebooks live in LDS replicated per lane of a 32-lane ds_read_b32 group (bank = lane), so the
//    random gathers are conflict-free; for TCQ the sign flip is folded into the table (index = top
//    S+1 bits of h).  Per 2 weights the VALU work is: alignbit, mad_u24, shift, and_or, dot2.
//  * fp32 accumulation with v_dot2_f32_f16.  MFMA is not used at n<=8: a 16x16x32 MFMA holds the
//    VALU issue port as long as the dot2s it would replace.
#pragma once
#include "qpal_common.h"

namespace qpal {

struct TcParams {
    // outputs
    float *out;      // gemv: fp32 [n][ldo]
    uint16_t *wout;  // dequant: fp16 [m][ldw]
    long ldo;        // gemv out row stride (floats)
    long ldw;        // dequant row stride (halves)
    // inputs
    const uint32_t *c1;
    const uint32_t *c2;
    const uint16_t *x;  // fp16 [n][k]
    const void *tab;    // codebook: TCQ fp16 pairs [2^S]; LUT fp16 [2^bits][vec]
    int n;              // live batch rows
    int k;              // x row stride (= in_features)
    int nrows;          // supertile rows (m/32)
    int nsc1, nsc2;     // supertile columns of stream 1 / stream 2
    int st1, st2;       // steps (4 supertiles) per supertile row, per stream
    int col2;           // first column of stream 2 (k/2 for combt)
    int log2_wpr;       // waves per supertile row inside a workgroup = 1 << log2_wpr (<= 16)
    int sk;             // split-K factor across workgroups (atomics when > 1)
    int nitems;         // work items = ceil(nrows / rows_per_wg) * sk
};

// ================================================================================================
// TCQ codec.  Bit surgery on one lane's KV dwords (reference lanes A = bits [0,16KV), B = [16KV,32KV))
// for tile group G = ksub*2 + msub; state I in 0..7 (0..3 = lane A's j, 4..7 = lane B's j).
template <int S, int KV>
struct TcqCodec {
    static constexpr int NW = KV;          // dwords per lane per supertile
    static constexpr int L4 = 4 * KV;      // stream bits per reference lane per tile
    static constexpr bool kNeedsNext = true;
    // codebook image: entry e (S+1 bits: sign flag | S index bits) x C copies, copy c at dword e*C+c
    static constexpr int LOG2C = 14 - S;   // 32 / 16 / 8 copies -> 128 KiB for every S
    static constexpr int C = 1 << LOG2C;
    static constexpr int LDS_DWORDS = (1 << (S + 1)) * C;

    template <int G>
    static constexpr int baseA() { return G * L4; }
    template <int G>
    static constexpr int baseB() { return 16 * KV + G * L4; }

    static __device__ __forceinline__ void build(uint32_t *lds, const void *tab, int tid, int nthreads) {
        const uint32_t *__restrict__ tlut = static_cast<const uint32_t *>(tab);
        constexpr int CHUNKS = LDS_DWORDS / 4;  // 16-byte chunks: conflict-free ds_write_b128
        for (int c = tid; c < CHUNKS; c += nthreads) {
            const int e = (c * 4) >> LOG2C;
            const uint32_t v = tlut[e & ((1 << S) - 1)] ^ (((uint32_t)e >> S) << 15);
            reinterpret_cast<u32x4 *>(lds)[c] = u32x4{v, v, v, v};
        }
    }

    // (A << L4) | B in the low 2*L4 bits (KV <= 4 only: 2*L4 <= 32)
    template <int G>
    static __device__ __forceinline__ uint32_t joined(const uint32_t (&w)[NW]) {
        return __builtin_amdgcn_alignbit(ext32<baseA<G>()>(w), ext32<baseB<G>() + L4 - 32>(w), 32 - L4);
    }

    // top-aligned head of this lane's stream for tile G (its first >= 16 bits): what lane-1 needs
    template <int G>
    static __device__ __forceinline__ uint32_t head(const uint32_t (&w)[NW]) {
        if constexpr (KV <= 4) {
            if constexpr (2 * L4 == 32) return joined<G>(w);
            else return joined<G>(w) << (32 - 2 * L4);
        } else if constexpr (baseA<G>() + L4 >= 32) {
            return ext32<baseA<G>() + L4 - 32>(w);
        } else {
            return w[0] << (32 - L4);
        }
    }

    // trellis state in the LOW 16 bits, garbage above
    template <int G, int I>
    static __device__ __forceinline__ uint32_t window(const uint32_t (&w)[NW], uint32_t next_head) {
        if constexpr (KV <= 4) {
            constexpr int n = I * KV + 16 - 2 * L4;  // bits needed from the next lane
            const uint32_t y = joined<G>(w);
            if constexpr (n <= 0) return y >> (-n);
            else return __builtin_amdgcn_alignbit(y, next_head, 32 - n);
        } else if constexpr (I < 4) {
            constexpr int t0 = I * KV;
            if constexpr (t0 + 16 <= L4) {
                return ext32<baseA<G>() + L4 - 16 - t0>(w);
            } else {
                constexpr int n = t0 + 16 - L4;
                return __builtin_amdgcn_alignbit(ext32<baseA<G>()>(w), ext32<baseB<G>() + L4 - 32>(w), 32 - n);
            }
        } else {
            constexpr int t0 = (I - 4) * KV;
            if constexpr (t0 + 16 <= L4) {
                return ext32<baseB<G>() + L4 - 16 - t0>(w);
            } else {
                constexpr int n = t0 + 16 - L4;
                return __builtin_amdgcn_alignbit(ext32<baseB<G>()>(w), next_head, 32 - n);
            }
        }
    }

    // half2 weight pair of vector (G, I)
    template <int G, int I>
    static __device__ __forceinline__ uint32_t pair(const uint32_t *lds, uint32_t laneoff, const uint32_t (&w)[NW],
                                                    uint32_t next_head) {
        const uint32_t s = window<G, I>(w, next_head);
        uint32_t h;
        asm("v_mad_u32_u24 %0, %1, %1, %1" : "=v"(h) : "v"(s));  // s*s + s: low 16 bits exact
        const uint32_t e = __builtin_amdgcn_ubfe(h, 15 - S, S + 1);
        const uint32_t a = (e << (LOG2C + 2)) | laneoff;
        return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lds) + a);
    }
};

// ================================================================================================
// VQ/SQ LUT codec (tensor-core packed format, lib/quantizer/quant_op.py:101-162).
//   VEC == 2 : `BITS` dwords per lane; code j of reference lane r, tile G at LE bit r*16B + G*4B + j*B
//   VEC == 1 : 2*BITS dwords per lane; code q = 2j+e at LE bit r*32B + G*8B + q*B.  For BITS <= 6 the
//              two codes of a half2 are adjacent, so one gather in a pair table of 2^(2B) half2
//              entries decodes both (the reference does this for BITS <= 4 only: LOAD_TYPE::DUP).
template <int BITS, int VEC>
struct LutCodec {
    static_assert(VEC == 1 || VEC == 2, "tensor-core format has vec_sz 1 or 2");
    static constexpr bool PAIR = (VEC == 1 && BITS <= 6);
    static constexpr int NW = (VEC == 2) ? BITS : 2 * BITS;
    static constexpr bool kNeedsNext = false;
    static constexpr int IDXBITS = (VEC == 2) ? BITS : (PAIR ? 2 * BITS : BITS);
    static constexpr int LOG2C = (15 - IDXBITS) < 5 ? (15 - IDXBITS) : 5;  // <= 128 KiB
    static constexpr int C = 1 << LOG2C;
    static constexpr int LDS_DWORDS = (1 << IDXBITS) * C;
    static_assert(LOG2C >= 2, "table build writes 4 copies per 16-byte chunk");

    static __device__ __forceinline__ void build(uint32_t *lds, const void *tab, int tid, int nthreads) {
        const uint16_t *__restrict__ l16 = static_cast<const uint16_t *>(tab);
        const uint32_t *__restrict__ l32 = static_cast<const uint32_t *>(tab);
        constexpr int CHUNKS = LDS_DWORDS / 4;
        for (int c = tid; c < CHUNKS; c += nthreads) {
            const int e = (c * 4) >> LOG2C;
            uint32_t v;
            if constexpr (VEC == 2) v = l32[e];
            else if constexpr (PAIR) v = (uint32_t)l16[e & ((1 << BITS) - 1)] | ((uint32_t)l16[e >> BITS] << 16);
            else v = l16[e];
            reinterpret_cast<u32x4 *>(lds)[c] = u32x4{v, v, v, v};
        }
    }

    template <int G>
    static __device__ __forceinline__ uint32_t head(const uint32_t (&)[NW]) { return 0u; }

    template <int POS, int NB_>
    static __device__ __forceinline__ uint32_t gather(const uint32_t *lds, uint32_t laneoff, const uint32_t (&w)[NW]) {
        // NB_ index bits at LE bit POS; stay inside one dword when possible (single v_bfe_u32)
        uint32_t e;
        if constexpr ((POS & 31) + NB_ <= 32) e = __builtin_amdgcn_ubfe(w[POS >> 5], POS & 31, NB_);
        else e = __builtin_amdgcn_ubfe(ext32<POS>(w), 0, NB_);
        const uint32_t a = (e << (LOG2C + 2)) | laneoff;
        return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lds) + a);
    }

    template <int G, int I>
    static __device__ __forceinline__ uint32_t pair(const uint32_t *lds, uint32_t laneoff, const uint32_t (&w)[NW],
                                                    uint32_t) {
        constexpr int r = I >> 2, j = I & 3;
        if constexpr (VEC == 2) {
            return gather<r * 16 * BITS + G * 4 * BITS + j * BITS, BITS>(lds, laneoff, w);
        } else if constexpr (PAIR) {
            return gather<r * 32 * BITS + G * 8 * BITS + 2 * j * BITS, 2 * BITS>(lds, laneoff, w);
        } else {
            constexpr int pos = r * 32 * BITS + G * 8 * BITS + 2 * j * BITS;
            const uint32_t lo = gather<pos, BITS>(lds, laneoff, w);
            const uint32_t hi = gather<pos + BITS, BITS>(lds, laneoff, w);
            return lo | (hi << 16);
        }
    }
};

// ================================================================================================
// One stream of one supertile row as seen by one wave.
struct StreamView {
    const uint32_t *base;  // first dword of this supertile row of the stream
    int nsc;               // supertile columns in the stream
    int col0;              // first x column of the stream
};

template <int NW>
__device__ __forceinline__ void load_step_w(const StreamView &sv, int step, int lane, uint32_t (&w)[NW]) {
    const int sc = step * 4 + (lane >> 4);
    if (sc < sv.nsc) {
        load_words_nt<NW>(sv.base + ((long)step * 64 + lane) * NW, w);
    } else {
#pragma unroll
        for (int i = 0; i < NW; i++) w[i] = 0u;
    }
}

// x fragments of a step: xv[b][ksub*2+jh] = the 4 halves at col0 + 128*step + 32*sc + 16*ksub + 8*jh + 4*u
// (.x feeds reference lane A's vector, .y lane B's)
template <int NB>
__device__ __forceinline__ void load_step_x(const StreamView &sv, const uint16_t *__restrict__ x, int k, int n,
                                            int step, int lane, u32x2 (&xv)[NB][4]) {
    const int sc = step * 4 + (lane >> 4);
    const bool live = sc < sv.nsc;
    const long col = (long)sv.col0 + (long)sc * 32 + 4 * (lane & 1);
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int bb = b < n ? b : n - 1;
        const uint16_t *row = x + (long)bb * k + col;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            xv[b][q] = live ? *reinterpret_cast<const u32x2 *>(row + 8 * q) : u32x2{0u, 0u};
        }
    }
}

template <class Codec, int NB>
__device__ __forceinline__ void gemv_step(const uint32_t *lut, uint32_t laneoff, const uint32_t (&w)[Codec::NW],
                                          const u32x2 (&xv)[NB][4], float (&acc)[NB][4]) {
    static_for<0, 4>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int ksub = g >> 1, msub = g & 1;
        uint32_t nh = 0u;
        if constexpr (Codec::kNeedsNext) nh = row16_next(Codec::template head<g>(w));
        static_for<0, 8>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int j = i & 3, jl = j & 1, jh = j >> 1, isB = i >> 2;
            const uint32_t wv = Codec::template pair<g, i>(lut, laneoff, w, nh);
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const uint32_t xx = isB ? xv[b][ksub * 2 + jh].y : xv[b][ksub * 2 + jh].x;
                acc[b][msub * 2 + jl] = fdot2(wv, xx, acc[b][msub * 2 + jl]);
            }
        });
    });
}

// all steps [s0, s1) of one stream for this wave, one-step-ahead register prefetch
template <class Codec, int NB>
__device__ __forceinline__ void gemv_stream(const uint32_t *lut, uint32_t laneoff, const StreamView &sv,
                                            const uint16_t *__restrict__ x, int k, int n, int s0, int s1, int lane,
                                            float (&acc)[NB][4]) {
    constexpr int NW = Codec::NW;
    constexpr bool XPRE = (NB <= 2) && (NW <= 10);  // prefetch x with the weights while registers allow
    if (s0 >= s1) return;
    uint32_t w[NW];
    u32x2 xv[NB][4];
    load_step_w<NW>(sv, s0, lane, w);
    load_step_x<NB>(sv, x, k, n, s0, lane, xv);
    for (int s = s0; s < s1; s++) {
        uint32_t wn[NW];
        u32x2 xn[NB][4];
        const bool more = s + 1 < s1;
        if (more) {
            load_step_w<NW>(sv, s + 1, lane, wn);
            if constexpr (XPRE) load_step_x<NB>(sv, x, k, n, s + 1, lane, xn);
        }
        gemv_step<Codec, NB>(lut, laneoff, w, xv, acc);
        if (more) {
#pragma unroll
            for (int i = 0; i < NW; i++) w[i] = wn[i];
            if constexpr (XPRE) {
#pragma unroll
                for (int b = 0; b < NB; b++)
#pragma unroll
                    for (int q = 0; q < 4; q++) xv[b][q] = xn[b][q];
            } else {
                load_step_x<NB>(sv, x, k, n, s + 1, lane, xv);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fused decode + GEMV.  1024 threads (16 waves, 4 per SIMD), one workgroup per CU (LDS-bound).
// C2 == void: single stream.  Otherwise combt (columns [0,col2) from c1 via C1, the rest from c2 via C2);
// both codecs share one codebook image.
template <class C1, class C2, int NB>
__global__ __launch_bounds__(1024) void tc_gemv_kernel(const TcParams p) {
    __shared__ __attribute__((aligned(16))) uint32_t lut[C1::LDS_DWORDS];
    __shared__ float red[16][NB][32];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t laneoff = (uint32_t)(lane & (C1::C - 1)) << 2;

    C1::build(lut, p.tab, tid, 1024);
    __syncthreads();

    const int wpr = 1 << p.log2_wpr;
    const int rloc = wave >> p.log2_wpr;  // supertile row inside the workgroup's row group
    const int wr = wave & (wpr - 1);      // this wave's K-chunk inside the row
    const int rows_per_wg = 16 >> p.log2_wpr;
    const int st = p.st1 + p.st2;
    const int nchunk = wpr * p.sk;

    for (int item = blockIdx.x; item < p.nitems; item += gridDim.x) {
        const int rg = item / p.sk, ks = item - rg * p.sk;
        const int sr = rg * rows_per_wg + rloc;
        float acc[NB][4];
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int a = 0; a < 4; a++) acc[b][a] = 0.f;

        if (sr < p.nrows) {
            const int c = ks * wpr + wr;
            const int s0 = (int)(((long)st * c) / nchunk), s1 = (int)(((long)st * (c + 1)) / nchunk);
            {
                const StreamView sv{p.c1 + (long)sr * p.nsc1 * 16 * C1::NW, p.nsc1, 0};
                gemv_stream<C1, NB>(lut, laneoff, sv, p.x, p.k, p.n, s0, s1 < p.st1 ? s1 : p.st1, lane, acc);
            }
            if constexpr (!std::is_void_v<C2>) {
                const StreamView sv{p.c2 + (long)sr * p.nsc2 * 16 * C2::NW, p.nsc2, p.col2};
                gemv_stream<C2, NB>(lut, laneoff, sv, p.x, p.k, p.n, (s0 > p.st1 ? s0 : p.st1) - p.st1, s1 - p.st1,
                                    lane, acc);
            }
        }
        // lanes {p, p^1} x the 4 DPP rows hold partials of the same 4 output rows
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int a = 0; a < 4; a++) {
                float v = acc[b][a];
                v = wave_xor_add(v, 1);
                v = wave_xor_add(v, 16);
                v = wave_xor_add(v, 32);
                if (lane < 16 && (lane & 1) == 0) red[wave][b][(lane >> 1) + 8 * (a & 1) + 16 * (a >> 1)] = v;
            }
        __syncthreads();
        if (tid < rows_per_wg * 32) {
            const int rl = tid >> 5, r = tid & 31;
            const int srow = rg * rows_per_wg + rl;
            if (srow < p.nrows) {
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    if (b < p.n) {
                        float v = 0.f;
                        for (int q = 0; q < wpr; q++) v += red[rl * wpr + q][b][r];
                        float *dst = p.out + (long)b * p.ldo + (long)srow * 32 + r;
                        if (p.sk == 1) *dst = v;
                        else atomicAdd(dst, v);
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Decode to fp16 W (bit-exact integer + LUT path).  Each lane stores 8-byte pieces: lane A's and
// lane B's vector of the same j are adjacent columns.
template <class Codec>
__device__ __forceinline__ void dequant_step(const uint32_t *lut, uint32_t laneoff, const uint32_t (&w)[Codec::NW],
                                             uint16_t *__restrict__ wrow0, long ldw) {
    // wrow0 -> W[32*sr + (p>>1)][col0 + 32*sc + 4*u]
    static_for<0, 4>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int ksub = g >> 1, msub = g & 1;
        uint32_t nh = 0u;
        if constexpr (Codec::kNeedsNext) nh = row16_next(Codec::template head<g>(w));
        static_for<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int jl = j & 1, jh = j >> 1;
            const uint32_t va = Codec::template pair<g, j>(lut, laneoff, w, nh);
            const uint32_t vb = Codec::template pair<g, j + 4>(lut, laneoff, w, nh);
            uint16_t *dst = wrow0 + (long)(8 * jl + 16 * msub) * ldw + 16 * ksub + 8 * jh;
            *reinterpret_cast<u32x2 *>(dst) = u32x2{va, vb};
        });
    });
}

template <class Codec>
__device__ __forceinline__ void dequant_stream(const uint32_t *lut, uint32_t laneoff, const uint32_t *__restrict__ c,
                                               int nsc, int nst, int col0, uint16_t *__restrict__ wout, long ldw,
                                               int sr, int wave, int lane) {
    constexpr int NW = Codec::NW;
    const StreamView sv{c + (long)sr * nsc * 16 * NW, nsc, col0};
    for (int s = wave; s < nst; s += 16) {
        uint32_t w[NW];
        load_step_w<NW>(sv, s, lane, w);
        const int sc = s * 4 + (lane >> 4);
        if (sc < nsc) {  // whole 16-lane DPP rows are live or idle together
            uint16_t *wrow0 = wout + ((long)sr * 32 + ((lane & 15) >> 1)) * ldw + col0 + (long)sc * 32 + 4 * (lane & 1);
            dequant_step<Codec>(lut, laneoff, w, wrow0, ldw);
        }
    }
}

template <class C1, class C2>
__global__ __launch_bounds__(1024) void tc_dequant_kernel(const TcParams p) {
    __shared__ __attribute__((aligned(16))) uint32_t lut[C1::LDS_DWORDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t laneoff = (uint32_t)(lane & (C1::C - 1)) << 2;
    C1::build(lut, p.tab, tid, 1024);
    __syncthreads();
    for (int sr = blockIdx.x; sr < p.nrows; sr += gridDim.x) {
        dequant_stream<C1>(lut, laneoff, p.c1, p.nsc1, p.st1, 0, p.wout, p.ldw, sr, wave, lane);
        if constexpr (!std::is_void_v<C2>)
            dequant_stream<C2>(lut, laneoff, p.c2, p.nsc2, p.st2, p.col2, p.wout, p.ldw, sr, wave, lane);
    }
}

}  // namespace qpal
