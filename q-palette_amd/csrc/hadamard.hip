// Incoherence rotation either side of the packed GEMV: sign flip, (blocked) Hadamard transform, scales,
// optional SwiGLU of the previous projection's output — one launch, one workgroup per transformed block.
//
// Replaces matmul_hadU_cuda / matmul_hadU_head_cuda (reference: lib/utils/matmul_had.py:95-110, 137-151; the
// butterflies are the third-party fast_hadamard_transform there) together with the elementwise torch ops
// around them in lib/linear/incoherent_linear.py:81, 106, 325-337, 488-503.
//
// A block of hd = K * P elements (P a power of two) is viewed as [K][P]:
//   1. t = WHT_P over the P axis (Sylvester order) * hd^-1/2, fp32 radix-4 butterflies in LDS;
//   2. K > 1: u = hadK @ t over the K axis on the matrix pipe (v_mfma_f32_16x16x32_f16, hadK entries are
//      +-1: exact in fp16).  round_mid = 1 rounds t to fp16 first, like the reference's fp16 pipeline
//      (matmul_hadU_cuda); round_mid = 0 feeds t as an fp16 hi + lo pair, i.e. fp32-grade like
//      matmul_hadU_head_cuda's float path.
//   3. out = fp16( fp16(u) * post_scale [* sv] ).
#include "qpal_common.h"
#include "wht64.h"

namespace qpal {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));

struct HadParams {
    uint16_t *out;         // fp16 [rows][n]
    const void *in;        // fp16 [rows][n] | fp32 [rows][n] | fp32 [rows][2n] (up | gate)
    const uint16_t *su;    // fp16 [n] pre-multiplier or null
    const uint16_t *sv;    // fp16 [n] post-multiplier or null
    const uint16_t *hadk;  // fp16 [K][K] row-major (null when K == 1)
    int rows, n, hd, K, logP;
    int in_mode, round_mid;
    float pre_scale, post_scale;
    int npass;             // butterfly passes; pass i handles r[i] index bits in registers (sum r = logP)
    int r[4];
    unsigned long long *dbg;  // HAD_STAMPS diagnostic builds: per-wave s_memtime stamps
    // RMSNorm in front of the transform (qpal_hadamard_rms; in_mode F32, hd == n): x <- x * rsqrt(mean(x^2) + eps) * w.  The norm's
    // scalar commutes with the transform: the passes run on x * w * 2^-6 and 64 / rms multiplies the result (had_kernel)
    float rms_eps;
    const uint16_t *rms_w;    // fp16 [n] or null
};

#ifdef HAD_STAMPS
#define HAD_STAMP(k)                                                                       \
    do {                                                                                   \
        if (p.dbg && (threadIdx.x & 63) == 0) p.dbg[(threadIdx.x >> 6) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define HAD_STAMP(k) do {} while (0)
#endif

__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ float round_f16(float v) { return (float)(_Float16)v; }
__device__ __forceinline__ uint16_t f2h(float v) { return __builtin_bit_cast(uint16_t, (_Float16)v); }

// LDS index of element i: one pad word per 32, so that the first pass (every thread writes ITS OWN run of 2^r
// consecutive elements: lane stride 32 words) spreads over the banks instead of hitting one
__device__ __forceinline__ int pad(int i) { return i + (i >> 5); }

// 2^R-point Walsh-Hadamard butterflies in registers (Sylvester order: bit s of j <-> index bit b0 + s)
template <int R>
__device__ __forceinline__ void butterfly(float (&v)[1 << R]) {
    static_for<0, R>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
#pragma unroll
        for (int j = 0; j < (1 << R); j++) {
            if (!(j & (1 << s))) {
                const float a = v[j], b = v[j | (1 << s)];
                v[j] = a + b;
                v[j | (1 << s)] = a - b;
            }
        }
    });
}

// 2^R consecutive inputs of row `row` starting at column `col` (+ SwiGLU, + sign flip), with the reference's
// fp16 rounding points.  16-byte vector loads where the run is long enough (alignment checked by the C-ABI).
template <int R>
__device__ __forceinline__ void load_input(const HadParams &p, int row, int col, float (&v)[1 << R], float *ss = nullptr) {
    constexpr int E = 1 << R;
    auto load_f32 = [&](const float *src, float (&dst)[E]) {
        if constexpr (E >= 4) {
#pragma unroll
            for (int j = 0; j < E; j += 4) {
                const float4_t t = *reinterpret_cast<const float4_t *>(src + j);
                dst[j] = t[0], dst[j + 1] = t[1], dst[j + 2] = t[2], dst[j + 3] = t[3];
            }
        } else {
#pragma unroll
            for (int j = 0; j < E; j++) dst[j] = src[j];
        }
    };
    auto load_f16 = [&](const uint16_t *src, float (&dst)[E]) {
        if constexpr (E >= 8) {
#pragma unroll
            for (int j = 0; j < E; j += 8) {
                const half8_t t = *reinterpret_cast<const half8_t *>(src + j);
#pragma unroll
                for (int e = 0; e < 8; e++) dst[j + e] = (float)t[e];
            }
        } else {
#pragma unroll
            for (int j = 0; j < E; j++) dst[j] = h2f(src[j]);
        }
    };
    if (p.in_mode == QPAL_IN_F16) {
        load_f16(static_cast<const uint16_t *>(p.in) + (long)row * p.n + col, v);
    } else if (p.in_mode == QPAL_IN_F32) {
        load_f32(static_cast<const float *>(p.in) + (long)row * p.n + col, v);
        if (p.rms_eps > 0.f) {  // RMSNorm: sum of squares of the fp32 input, weight and a power-of-two guard scale, ONE rounding
            float w[E];
#pragma unroll
            for (int j = 0; j < E; j++) w[j] = 1.0f;
            if (p.rms_w) {
#pragma unroll
                for (int j = 0; j < E; j++) w[j] = h2f(p.rms_w[col + j]);
            }
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < E; j++) {
                acc += v[j] * v[j];
                v[j] = round_f16(v[j] * w[j] * 0.015625f);
            }
            if (ss) *ss += acc;
        } else {
#pragma unroll
            for (int j = 0; j < E; j++) v[j] = round_f16(v[j]);
        }
    } else {
        const float *src = static_cast<const float *>(p.in) + (long)row * 2 * p.n + col;
        constexpr int CH = E < 4 ? E : 4;
#pragma unroll
        for (int j = 0; j < E; j += CH) {
            float up[CH], gate[CH];
            if constexpr (CH == 4) {
                const float4_t tu = *reinterpret_cast<const float4_t *>(src + j);
                const float4_t tg = *reinterpret_cast<const float4_t *>(src + p.n + j);
#pragma unroll
                for (int e = 0; e < 4; e++) up[e] = tu[e], gate[e] = tg[e];
            } else {
#pragma unroll
                for (int e = 0; e < CH; e++) up[e] = src[j + e], gate[e] = src[p.n + j + e];
            }
#pragma unroll
            for (int e = 0; e < CH; e++) {
                const float u = round_f16(up[e]), gt = round_f16(gate[e]);
                v[j + e] = round_f16(round_f16(gt / (1.0f + __expf(-gt))) * u);
            }
        }
    }
    if (p.su) {
        if constexpr (E >= 8) {
#pragma unroll
            for (int j = 0; j < E; j += 8) {
                const half8_t t = *reinterpret_cast<const half8_t *>(p.su + col + j);
#pragma unroll
                for (int e = 0; e < 8; e++) v[j + e] = round_f16(v[j + e] * (float)t[e]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < E; j++) v[j] = round_f16(v[j] * h2f(p.su[col + j]));
        }
    }
}

// One butterfly pass over index bits [b0, b0 + R): every thread owns whole 2^R-element groups in registers.
// FROM_GLOBAL: the first pass (b0 = 0) reads its contiguous run straight from the input; TO_OUT: the last pass of
// a K = 1 transform scales, rounds and stores fp16 without going back through LDS.
template <int R, bool FROM_GLOBAL, bool TO_OUT, int NT>
__device__ __forceinline__ void had_pass(const HadParams &p, float *buf, int b0, int row, int col0, int tid, float *ss = nullptr) {
    constexpr int E = 1 << R;
    const int ngroups = p.hd >> R;
    for (int g = tid; g < ngroups; g += NT) {
        const int i0 = ((g >> b0) << (b0 + R)) | (g & ((1 << b0) - 1));
        float v[E];
        if constexpr (FROM_GLOBAL) {
            load_input<R>(p, row, col0 + i0, v, ss);
        } else {
#pragma unroll
            for (int j = 0; j < E; j++) v[j] = buf[pad(i0 + (j << b0))];
        }
        butterfly<R>(v);
        if constexpr (TO_OUT) {
            uint16_t *orow = p.out + (long)row * p.n + col0;
#pragma unroll
            for (int j = 0; j < E; j++) {
                const int i = i0 + (j << b0);
                float u = round_f16(v[j] * p.pre_scale) * p.post_scale;
                if (p.sv) u = round_f16(u) * h2f(p.sv[col0 + i]);
                orow[i] = f2h(u);
            }
        } else {
#pragma unroll
            for (int j = 0; j < E; j++) buf[pad(i0 + (j << b0))] = v[j];
        }
    }
}

template <bool FROM_GLOBAL, bool TO_OUT, int NT>
__device__ __forceinline__ void had_pass_r(int r, const HadParams &p, float *buf, int b0, int row, int col0, int tid, float *ss = nullptr) {
    switch (r) {
        case 1: had_pass<1, FROM_GLOBAL, TO_OUT, NT>(p, buf, b0, row, col0, tid, ss); break;
        case 2: had_pass<2, FROM_GLOBAL, TO_OUT, NT>(p, buf, b0, row, col0, tid, ss); break;
        case 3: had_pass<3, FROM_GLOBAL, TO_OUT, NT>(p, buf, b0, row, col0, tid, ss); break;
        case 4: had_pass<4, FROM_GLOBAL, TO_OUT, NT>(p, buf, b0, row, col0, tid, ss); break;
        default: had_pass<5, FROM_GLOBAL, TO_OUT, NT>(p, buf, b0, row, col0, tid, ss); break;
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void had_kernel(HadParams p) {
    extern __shared__ float buf[];  // hd floats, padded (pad()) [+ NT / 64 partial sums of squares: RMSNorm]
    const int tid = threadIdx.x;
    const int bpr = p.n / p.hd;
    const int row = blockIdx.x / bpr, blk = blockIdx.x - row * bpr;
    const int col0 = blk * p.hd;
    const int lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, c16 = lane & 15;

    HAD_STAMP(0);
    // hadK A fragments of the common K <= 32 case go out first: their latency hides behind the input's
    const bool hoist = p.K > 1 && p.K <= 32;
    uint16_t araw[2][8];  // raw loads only: converting here would wait for them before the input is even requested
    if (hoist) {
#pragma unroll
        for (int jt = 0; jt < 2; jt++) {
            const int ja = (jt << 4) + c16;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int i = (q << 3) + e;
                araw[jt][e] = p.hadk[(ja < p.K ? ja : p.K - 1) * p.K + (i < p.K ? i : p.K - 1)];
            }
        }
    }

    // ---- WHT over the low logP index bits
    const bool direct = p.K == 1;  // the last pass writes the output itself
    int b0 = 0;
    for (int ps = 0; ps < p.npass; ps++) {
        const int r = p.r[ps];
        const bool last = ps == p.npass - 1;
        if (ps == 0) {
            if (last && direct) had_pass_r<true, true, NT>(r, p, buf, 0, row, col0, tid);
            else if (p.rms_eps > 0.f) {  // (the host gives an RMSNorm launch at least two passes)
                float ss = 0.f;
                had_pass_r<true, false, NT>(r, p, buf, 0, row, col0, tid, &ss);
                ss = wave_sum(ss);
                float *part = buf + p.hd + (p.hd >> 5) + 1;
                if (lane == 0) part[wave] = ss;
                __syncthreads();
                float tot = 0.f;
                for (int w = 0; w < NT / 64; w++) tot += part[w];
                p.pre_scale *= __builtin_amdgcn_rsqf(tot / (float)p.hd + p.rms_eps) * 64.0f;  // the later passes scale by it
            } else had_pass_r<true, false, NT>(r, p, buf, 0, row, col0, tid);
        } else {
            if (last && direct) had_pass_r<false, true, NT>(r, p, buf, b0, row, col0, tid);
            else had_pass_r<false, false, NT>(r, p, buf, b0, row, col0, tid);
        }
        b0 += r;
        HAD_STAMP(1 + 2 * ps);
        if (!(last && direct)) __syncthreads();
        HAD_STAMP(2 + 2 * ps);
    }
    if (direct) return;

    // ---- hadK over the K axis: D[j][c] = sum_i hadK[j][i] * t[i][c], 16 columns per MFMA tile.
    // All loads are unconditional (clamped index, masked value) so that they issue back to back.
    uint16_t *orow = p.out + (long)row * p.n + col0;
    const int P = 1 << p.logP;
    const int njt = (p.K + 15) >> 4, nkc = (p.K + 31) >> 5;
    half8_t afr[2];
    if (hoist) {
#pragma unroll
        for (int jt = 0; jt < 2; jt++) {
#pragma unroll
            for (int e = 0; e < 8; e++)
                afr[jt][e] = ((q << 3) + e < p.K && (jt << 4) + c16 < p.K) ? __builtin_bit_cast(_Float16, araw[jt][e])
                                                                           : (_Float16)0.f;
        }
    }
    auto b_fragment = [&](int kc, int c, half8_t &bh, half8_t &bl) {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int i = (kc << 5) + (q << 3) + e;
            t[e] = buf[pad((i < p.K ? i : p.K - 1) * P + c)];
        }
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int i = (kc << 5) + (q << 3) + e;
            const float v = i < p.K ? t[e] * p.pre_scale : 0.f;
            const _Float16 hi = (_Float16)v;
            bh[e] = hi;
            bl[e] = p.round_mid ? (_Float16)0.f : (_Float16)(v - (float)hi);
        }
    };
    auto store_tile = [&](int jt, int c, const float4_t &acc) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int j = (jt << 4) + (q << 2) + r;
            if (j < p.K) {
                float u = round_f16(acc[r]) * p.post_scale;
                if (p.sv) u = round_f16(u) * h2f(p.sv[col0 + j * P + c]);
                orow[j * P + c] = f2h(u);
            }
        }
    };
    for (int ct = wave; ct < (P >> 4); ct += NT / 64) {
        const int c = (ct << 4) + c16;
        if (hoist) {  // K <= 32: one B fragment serves both row tiles
            half8_t bh, bl;
            b_fragment(0, c, bh, bl);
#pragma unroll
            for (int jt = 0; jt < 2; jt++) {
                if (jt < njt) {
                    float4_t acc{0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[jt], bh, acc, 0, 0, 0);
                    if (!p.round_mid) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[jt], bl, acc, 0, 0, 0);
                    store_tile(jt, c, acc);
                }
            }
            continue;
        }
        for (int jt = 0; jt < njt; jt++) {
            float4_t acc{0.f, 0.f, 0.f, 0.f};
            const int ja = (jt << 4) + c16;  // A operand: row of hadK
            for (int kc = 0; kc < nkc; kc++) {
                half8_t a, bh, bl;
                uint16_t ah[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int i = (kc << 5) + (q << 3) + e;
                    ah[e] = p.hadk[(ja < p.K ? ja : p.K - 1) * p.K + (i < p.K ? i : p.K - 1)];
                }
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int i = (kc << 5) + (q << 3) + e;
                    a[e] = (i < p.K && ja < p.K) ? __builtin_bit_cast(_Float16, ah[e]) : (_Float16)0.f;
                }
                b_fragment(kc, c, bh, bl);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bh, acc, 0, 0, 0);
                if (!p.round_mid) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bl, acc, 0, 0, 0);
            }
            store_tile(jt, c, acc);
        }
    }
    HAD_STAMP(7);
}

// K = 1, hd = 1024 .. 8192: the whole transform on the matrix pipe (wht64.h), one wave quad per block of hd
template <int RT>
__global__ __launch_bounds__(256) void had_mfma_kernel(HadParams p) {
    const int bpr = p.n / p.hd;
    const int row = blockIdx.x / bpr, blk = blockIdx.x - row * bpr;
    const int col0 = blk * p.hd;
    const int lane = threadIdx.x & 63, ct = threadIdx.x >> 6;
    uint16_t *orow = p.out + (long)row * p.n + col0;
    wht64_quad<RT>(
        ct, lane, p.pre_scale,
        [&](int t, int kc) {  // fp16 input only: x * su is one packed multiply per dword (rounds like the reference)
            const long off = (long)row * p.n + col0 + (16 * t + (lane & 15)) * 64 + 32 * kc + 8 * (lane >> 4);
            wht_half8 h = *reinterpret_cast<const wht_half8 *>(static_cast<const uint16_t *>(p.in) + off);
            if (p.su) h = h * *reinterpret_cast<const wht_half8 *>(p.su + (off - (long)row * p.n));
            return h;
        },
        [&](int, int, int i, float v) {
            float u = round_f16(v) * p.post_scale;
            if (p.sv) u = round_f16(u) * h2f(p.sv[col0 + i]);
            orow[i] = f2h(u);
        });
}

// K = 1, hd = G * 4096 (8192, 16384, 32768): H_hd = H_G (x) H_4096 and the factors commute — workgroup g of a block first forms
// y = sum_j (-1)^popc(g & j) x'_j over the G sub-blocks of 4096 (x' = x * su [* w * 2^-6: RMSNorm, scalar applied at the end];
// it reads the whole block: L2), then runs the 4096-point transform on the matrix pipe (wht64_quad<4>).  y goes in as an
// fp16 hi + lo pair (staged in LDS by the whole workgroup): sums of a few fp16 values are exact in that form, so the result
// is fp32-grade like the butterflies'.
// One workgroup doing all of 8192 on one CU: 12.8 us per launch inside a 70B decode step (had_kernel<768>), 3 per layer.
template <int G, int MODE, bool RMS>
__global__ __launch_bounds__(256) void had_split_mfma_kernel(HadParams p) {
    // y as fp16 hi / lo, formed ONCE per workgroup (every wave of the quad reads all of the transform's input: with the
    // fp32 + RMSNorm arithmetic inside the tile loader the kernel was instruction-bound, 16.6 us)
    __shared__ __attribute__((aligned(16))) uint16_t yh[4096], yl[4096];
    __shared__ float part[4];
    const int per_row = (p.n / p.hd) * G;
    const int row = blockIdx.x / per_row, rem = blockIdx.x - row * per_row;
    const int blk = rem / G, g = rem - blk * G;
    const int col0 = blk * p.hd;
    const int tid = threadIdx.x, lane = tid & 63, ct = tid >> 6;
    uint16_t *orow = p.out + (long)row * p.n + col0 + g * 4096;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int off = 8 * (tid + 256 * c);
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; e++) y[e] = 0.f;
#pragma unroll
        for (int j = 0; j < G; j++) {
            const long at = (long)row * p.n + col0 + j * 4096 + off;
            float f[8];
            if constexpr (MODE == QPAL_IN_F16) {
                const wht_half8 h = *reinterpret_cast<const wht_half8 *>(static_cast<const uint16_t *>(p.in) + at);
#pragma unroll
                for (int e = 0; e < 8; e++) f[e] = (float)h[e];
            } else {
                const float *src = static_cast<const float *>(p.in) + at;
                const float4_t a = *reinterpret_cast<const float4_t *>(src), b = *reinterpret_cast<const float4_t *>(src + 4);
#pragma unroll
                for (int e = 0; e < 4; e++) f[e] = a[e], f[4 + e] = b[e];
                if constexpr (RMS) {
                    wht_half8 w;
                    if (p.rms_w) w = *reinterpret_cast<const wht_half8 *>(p.rms_w + col0 + j * 4096 + off);
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        ss += f[e] * f[e];
                        f[e] = round_f16(f[e] * (p.rms_w ? (float)w[e] : 1.0f) * 0.015625f);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e++) f[e] = round_f16(f[e]);  // the reference's .half()
                }
            }
            if (p.su) {
                const wht_half8 s8 = *reinterpret_cast<const wht_half8 *>(p.su + col0 + j * 4096 + off);
#pragma unroll
                for (int e = 0; e < 8; e++) f[e] = round_f16(f[e] * (float)s8[e]);
            }
            const bool neg = __builtin_popcount(g & j) & 1;
#pragma unroll
            for (int e = 0; e < 8; e++) y[e] += neg ? -f[e] : f[e];
        }
        wht_half8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            hi[e] = (_Float16)y[e];
            lo[e] = (_Float16)(y[e] - (float)hi[e]);
        }
        *reinterpret_cast<wht_half8 *>(yh + off) = hi;
        *reinterpret_cast<wht_half8 *>(yl + off) = lo;
    }
    float mul = 1.0f;
    if constexpr (RMS) {
        ss = wave_sum(ss);
        if (lane == 0) part[ct] = ss;
    }
    __syncthreads();
    if constexpr (RMS) mul = __builtin_amdgcn_rsqf((part[0] + part[1] + part[2] + part[3]) / (float)p.hd + p.rms_eps) * 64.0f;
    wht64_quad<4>(
        ct, lane, p.pre_scale,
        [&](int t, int kc) {
            const int off = (16 * t + (lane & 15)) * 64 + 32 * kc + 8 * (lane >> 4);
            return wht_hilo{*reinterpret_cast<const wht_half8 *>(yh + off), *reinterpret_cast<const wht_half8 *>(yl + off)};
        },
        [&](int, int, int i, float v) {
            float u = round_f16(v * mul) * p.post_scale;
            if (p.sv) u = round_f16(u) * h2f(p.sv[col0 + g * 4096 + i]);
            orow[i] = f2h(u);
        });
}

// K > 1 in ONE launch without a single-workgroup bottleneck: the two factors commute, so workgroup (row, block, j) first
// forms row j of the hadK product, m[c] = sum_i hadK[j][i] a[i][c] (it reads the whole block: 2 .. 8 bytes per element from
// L2, coalesced, every load of a thread in flight at once — K workgroups read what one workgroup read before), then runs
// WHT_P on its own P values in LDS and stores segment j.  fp32 until the one final rounding (the reference's fp16
// intermediate t is not formed: round_mid callers get a result at least as close to the exact transform).
// The 14336-vector with SwiGLU: two launches (4.9 + 4.9 us inside a token's graph) -> one.
template <int NT, int MODE, bool SU>
__global__ __launch_bounds__(NT) void had_mixfirst_kernel(HadParams p) {
    extern __shared__ float buf[];  // [parts][P] partial rows, then pad(P) floats for the butterflies
    const int tid = threadIdx.x;
    const int P = 1 << p.logP;
    const int bpr = p.n / p.hd;
    int b = blockIdx.x;
    const int j = b % p.K;
    b /= p.K;
    const int row = b / bpr, blk = b - row * bpr;
    const int col0 = blk * p.hd;
    const int parts = NT >> p.logP;  // host: P <= NT, both powers of two
    // P >= 64: the row part is wave-uniform, so the hadK entries are scalar loads and every address is a uniform base plus
    // one per-lane 32-bit offset
    const int c = tid & (P - 1), rp = __builtin_amdgcn_readfirstlane(tid >> p.logP);
    float acc = 0.f;
    // Raw loads first, arithmetic after; input mode and sign flip are template parameters: a run-time branch anywhere in
    // the batch makes the compiler wait for every load issued before it (measured: 13 us for the 14336-vector instead of
    // one round trip).  Index clamped and weight masked past K.  The workgroup is instruction-bound (14 elements per thread,
    // four waves per SIMD): the SwiGLU uses the hardware reciprocal (1 ulp of fp32, then rounded to fp16 as the reference's
    // fp16 silu is) instead of a full division.
    const gptr<const uint16_t> hk = as_global(p.hadk) + j * p.K;
    for (int i0 = rp; i0 < p.K; i0 += 16 * parts) {
        float v[16];
        uint16_t sraw[16], kraw[16];
        uint32_t at[16];  // element offsets inside the row (n < 2^31: checked by the C-ABI)
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int i = i0 + u * parts, ic = i < p.K ? i : p.K - 1;
            at[u] = (uint32_t)(col0 + ic * P) + (uint32_t)c;
            kraw[u] = hk[ic];
        }
        if constexpr (SU) {
            const gptr<const uint16_t> su = as_global(p.su);
#pragma unroll
            for (int u = 0; u < 16; u++) sraw[u] = su[at[u]];
        }
        if constexpr (MODE == QPAL_IN_SWIGLU_F32) {
            const gptr<const float> src = as_global(static_cast<const float *>(p.in)) + (long)row * 2 * p.n;
            const gptr<const float> srg = src + p.n;
            float up[16], gate[16];
#pragma unroll
            for (int u = 0; u < 16; u++) up[u] = src[at[u]], gate[u] = srg[at[u]];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const float uu = round_f16(up[u]), gt = round_f16(gate[u]);
                const float sg = gt * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gt * -1.44269504f));
                v[u] = round_f16(round_f16(sg) * uu);
            }
        } else if constexpr (MODE == QPAL_IN_F32) {
            const gptr<const float> src = as_global(static_cast<const float *>(p.in)) + (long)row * p.n;
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = src[at[u]];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = round_f16(v[u]);
        } else {
            const gptr<const uint16_t> src = as_global(static_cast<const uint16_t *>(p.in)) + (long)row * p.n;
            uint16_t raw[16];
#pragma unroll
            for (int u = 0; u < 16; u++) raw[u] = src[at[u]];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = h2f(raw[u]);
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int i = i0 + u * parts;
            if constexpr (SU) v[u] = round_f16(v[u] * h2f(sraw[u]));
            acc += (i < p.K ? h2f(kraw[u]) : 0.f) * v[u];
        }
    }
    float *m = buf + parts * P;
    if (parts > 1) {
        buf[rp * P + c] = acc;
        __syncthreads();
        if (tid < P) {
            float t = 0.f;
            for (int q = 0; q < parts; q++) t += buf[q * P + tid];
            m[pad(tid)] = t;
        }
    } else {
        m[pad(c)] = acc;
    }
    __syncthreads();
    HadParams s = p;  // the segment as a K = 1 transform of its own (pre_scale stays the block's hd^-1/2)
    s.hd = P;
    s.K = 1;
    int b0 = 0;
    for (int ps = 0; ps < p.npass; ps++) {
        const int r = p.r[ps];
        if (ps == p.npass - 1) {
            had_pass_r<false, true, NT>(r, s, m, b0, row, col0 + j * P, tid);
        } else {
            had_pass_r<false, false, NT>(r, s, m, b0, row, col0 + j * P, tid);
            __syncthreads();
        }
        b0 += r;
    }
}

// Butterfly pass plan: how many index bits each pass handles in registers.  First pass 5 bits (the next pass then strides
// by >= 32 floats: conflict-free LDS columns) — but only 3 for small transforms, which would otherwise keep just
// hd / 32 threads busy (a 512-segment: 16 lanes doing all the SwiGLU and conversion work); the rest in even shares <= 5.
void plan_passes(HadParams &p) {
    const int first = p.hd <= 1024 ? 3 : p.hd <= 8192 ? 4 : 5;
    int left = p.logP;
    p.npass = 0;
    if (left > 0) {
        p.r[p.npass++] = left < first ? left : first;
        left -= p.r[0];
        const int more = (left + 4) / 5;
        for (int i = 0; i < more; i++) {
            const int take = (left + (more - i) - 1) / (more - i);
            p.r[p.npass++] = take;
            left -= take;
        }
    }
}

int launch_hadamard(const HadParams &p, hipStream_t stream) {
    const bool rms = p.rms_eps > 0.f;  // the general kernel only (one workgroup per row: it needs the whole row's sum of squares)
    if (!rms && p.K > 1 && p.K <= 32 && p.round_mid && p.logP >= 6 && p.logP <= 10 && p.hd > 4096) {
        // One launch, K workgroups per block (had_mixfirst_kernel).  History: one workgroup for the whole 14336-vector with
        // SwiGLU 12.4 us; segment transforms + in-place hadK mix as two launches 2 x 4.9 us; a last-arriving workgroup
        // doing the mix behind device-scope fences 18.7 us.
        HadParams a = p;
        HadParams seg = p;
        seg.hd = 1 << p.logP;
        plan_passes(seg);
        a.npass = seg.npass;
        for (int i = 0; i < 4; i++) a.r[i] = seg.r[i];
        constexpr int NT = 1024;
        const int P = 1 << p.logP, parts = NT / P;
        const size_t lds = sizeof(float) * (size_t)(parts * P + P + (P >> 5) + 1);
        const dim3 grid(p.rows * (p.n / p.hd) * p.K);
#define QPAL_MIXFIRST(M_)                                                                                          \
    if (p.in_mode == M_) {                                                                                         \
        if (p.su) hipLaunchKernelGGL((had_mixfirst_kernel<NT, M_, true>), grid, dim3(NT), lds, stream, a);         \
        else hipLaunchKernelGGL((had_mixfirst_kernel<NT, M_, false>), grid, dim3(NT), lds, stream, a);             \
    }
        QPAL_MIXFIRST(QPAL_IN_F16) QPAL_MIXFIRST(QPAL_IN_F32) QPAL_MIXFIRST(QPAL_IN_SWIGLU_F32)
#undef QPAL_MIXFIRST
        return (int)hipGetLastError();
    }
    if (!rms && p.K == 1 && p.in_mode == QPAL_IN_F16 && (p.hd == 1024 || p.hd == 2048 || p.hd == 4096)) {  // 8192: the butterflies win (measured)
        const int g = p.rows * (p.n / p.hd);
        if (p.hd == 1024) hipLaunchKernelGGL((had_mfma_kernel<1>), dim3(g), dim3(256), 0, stream, p);
        else if (p.hd == 2048) hipLaunchKernelGGL((had_mfma_kernel<2>), dim3(g), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((had_mfma_kernel<4>), dim3(g), dim3(256), 0, stream, p);
        return (int)hipGetLastError();
    }
    if (p.K == 1 && (p.hd == 8192 || p.hd == 16384 || p.hd == 32768) && p.in_mode != QPAL_IN_SWIGLU_F32 && (!rms || p.hd == p.n)) {
        const int G = p.hd / 4096;
        const dim3 grid(p.rows * (p.n / p.hd) * G);
#define QPAL_HSPLIT(G_)                                                                                              \
    if (G == G_) {                                                                                                   \
        if (rms) hipLaunchKernelGGL((had_split_mfma_kernel<G_, QPAL_IN_F32, true>), grid, dim3(256), 0, stream, p);  \
        else if (p.in_mode == QPAL_IN_F32) hipLaunchKernelGGL((had_split_mfma_kernel<G_, QPAL_IN_F32, false>), grid, dim3(256), 0, stream, p); \
        else hipLaunchKernelGGL((had_split_mfma_kernel<G_, QPAL_IN_F16, false>), grid, dim3(256), 0, stream, p);     \
    }
        QPAL_HSPLIT(2) QPAL_HSPLIT(4) QPAL_HSPLIT(8)
#undef QPAL_HSPLIT
        return (int)hipGetLastError();
    }
    const int grid = p.rows * (p.n / p.hd);
    const size_t lds = sizeof(float) * (size_t)(p.hd + (p.hd >> 5) + 1 + 16);
    if (p.hd <= 4096) {  // <= 256 groups of 16: four waves do it
        hipLaunchKernelGGL((had_kernel<256>), dim3(grid), dim3(256), lds, stream, p);
    } else {
        // > 64 KiB of dynamic LDS needs the opt-in, per DEVICE (function attributes belong to the device's code object);
        // idempotent, races are harmless
        static bool attr_set[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
        if (dev < 0 || !attr_set[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&had_kernel<768>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            if (dev >= 0) attr_set[dev] = true;
        }
        hipLaunchKernelGGL((had_kernel<768>), dim3(grid), dim3(768), lds, stream, p);
    }
    return (int)hipGetLastError();
}

}  // namespace qpal

using namespace qpal;

#ifdef HAD_STAMPS
static unsigned long long *g_had_dbg = nullptr;
extern "C" void qpal_debug_had_stamps(void *buf) { g_had_dbg = static_cast<unsigned long long *>(buf); }
#endif

static int hadamard_impl(void *out_f16, const void *in, const void *su, const void *sv, const void *hadk, int rows, int n, int hd, int K,
                         int in_mode, int round_mid, float post_scale, float rms_eps, const void *rms_w, void *stream) {
    if (!out_f16 || !in) return QPAL_E_NULL;
    if (out_f16 == in) return QPAL_E_PARAM;  // not in place: the waves of a block read all of it while others already write
    if (K < 1 || K > 256 || (K > 1 && !hadk)) return K > 1 && !hadk ? QPAL_E_NULL : QPAL_E_PARAM;
    if (in_mode < QPAL_IN_F16 || in_mode > QPAL_IN_SWIGLU_F32) return QPAL_E_PARAM;
    if (rows < 1 || n < 1 || hd < 1 || n % hd || hd % K) return QPAL_E_SHAPE;
    if ((long)rows * (n / hd) > 0x7fffffffL || (long)rows * n / 16 > 0x7fffffffL) return QPAL_E_SHAPE;  // 1-D grids: one workgroup per block / per 4 column tiles
    const int P = hd / K;
    if (P & (P - 1)) return QPAL_E_SHAPE;
    if (K > 1 && P < 16) return QPAL_E_SHAPE;
    if (hd < 2 || (hd + (hd >> 5) + 1 + 16) * sizeof(float) > 160 * 1024) return QPAL_E_SHAPE;
    int logP = 0;
    while ((1 << logP) < P) logP++;
    const uintptr_t al = reinterpret_cast<uintptr_t>(out_f16) | reinterpret_cast<uintptr_t>(su) |
                         reinterpret_cast<uintptr_t>(sv) | reinterpret_cast<uintptr_t>(hadk);
    if (al & 1) return QPAL_E_ALIGN;
    // the first pass reads runs of 2^r elements with 16-byte vector loads
    if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(su)) & 15) return QPAL_E_ALIGN;
    if (((size_t)n * (in_mode == QPAL_IN_F16 ? 2 : 4)) % 16 && rows > 1) return QPAL_E_ALIGN;
    HadParams p{static_cast<uint16_t *>(out_f16), in, static_cast<const uint16_t *>(su), static_cast<const uint16_t *>(sv),
                static_cast<const uint16_t *>(hadk), rows, n, hd, K, logP, in_mode, round_mid ? 1 : 0,
                (float)(1.0 / sqrt((double)hd)), post_scale, 0, {0, 0, 0, 0}, nullptr, rms_eps, static_cast<const uint16_t *>(rms_w)};
#ifdef HAD_STAMPS
    p.dbg = g_had_dbg;
#endif
    plan_passes(p);
    if (rms_eps > 0.f && p.npass < 2) return QPAL_E_SHAPE;  // the sum of squares is formed between the first two passes
    return launch_hadamard(p, static_cast<hipStream_t>(stream));
}

extern "C" int qpal_hadamard(void *out_f16, const void *in, const void *su, const void *sv, const void *hadk, int rows,
                             int n, int hd, int K, int in_mode, int round_mid, float post_scale, void *stream) {
    return hadamard_impl(out_f16, in, su, sv, hadk, rows, n, hd, K, in_mode, round_mid, post_scale, 0.f, nullptr, stream);
}

extern "C" int qpal_hadamard_rms(void *out_f16, const float *in_f32, const void *rms_w, float rms_eps, const void *su, const void *hadk,
                                 int rows, int n, int K, float post_scale, void *stream) {
    if (!(rms_eps > 0.f)) return QPAL_E_PARAM;
    if (rms_w && (reinterpret_cast<uintptr_t>(rms_w) & 1)) return QPAL_E_ALIGN;
    return hadamard_impl(out_f16, in_f32, su, nullptr, hadk, rows, n, n, K, QPAL_IN_F32, 1, post_scale, rms_eps, rms_w, stream);
}
