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
};

__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ float round_f16(float v) { return (float)(_Float16)v; }
__device__ __forceinline__ uint16_t f2h(float v) { return __builtin_bit_cast(uint16_t, (_Float16)v); }

template <int NT>
__global__ __launch_bounds__(NT) void had_kernel(HadParams p) {
    extern __shared__ float buf[];  // hd floats
    const int tid = threadIdx.x;
    const int bpr = p.n / p.hd;
    const int row = blockIdx.x / bpr, blk = blockIdx.x - row * bpr;
    const int col0 = blk * p.hd;
    const int hd = p.hd;

    // ---- load (+ SwiGLU, + sign flip); the reference's fp16 rounding points are kept
    for (int i = tid; i < hd; i += NT) {
        const int col = col0 + i;
        float v;
        if (p.in_mode == QPAL_IN_F16) {
            v = h2f(static_cast<const uint16_t *>(p.in)[(long)row * p.n + col]);
        } else if (p.in_mode == QPAL_IN_F32) {
            v = round_f16(static_cast<const float *>(p.in)[(long)row * p.n + col]);
        } else {
            const float *src = static_cast<const float *>(p.in) + (long)row * 2 * p.n;
            const float up = round_f16(src[col]), gate = round_f16(src[p.n + col]);
            const float act = round_f16(gate / (1.0f + __expf(-gate)));
            v = round_f16(act * up);
        }
        if (p.su) v = round_f16(v * h2f(p.su[col]));
        buf[i] = v;
    }
    __syncthreads();

    // ---- WHT over the low logP index bits: radix-4 passes, then one radix-2 pass if logP is odd
    int b = 0;
    for (; b + 2 <= p.logP; b += 2) {
        const int lowmask = (1 << b) - 1;
        for (int g = tid; g < (hd >> 2); g += NT) {
            const int base = ((g >> b) << (b + 2)) | (g & lowmask);
            const float a0 = buf[base], a1 = buf[base + (1 << b)], a2 = buf[base + (2 << b)], a3 = buf[base + (3 << b)];
            const float s0 = a0 + a1, d0 = a0 - a1, s1 = a2 + a3, d1 = a2 - a3;
            buf[base] = s0 + s1;
            buf[base + (1 << b)] = d0 + d1;
            buf[base + (2 << b)] = s0 - s1;
            buf[base + (3 << b)] = d0 - d1;
        }
        __syncthreads();
    }
    if (b < p.logP) {
        const int lowmask = (1 << b) - 1;
        for (int g = tid; g < (hd >> 1); g += NT) {
            const int base = ((g >> b) << (b + 1)) | (g & lowmask);
            const float a0 = buf[base], a1 = buf[base + (1 << b)];
            buf[base] = a0 + a1;
            buf[base + (1 << b)] = a0 - a1;
        }
        __syncthreads();
    }

    uint16_t *orow = p.out + (long)row * p.n + col0;
    if (p.K == 1) {
        for (int i = tid; i < hd; i += NT) {
            float u = round_f16(buf[i] * p.pre_scale) * p.post_scale;
            if (p.sv) u = round_f16(u) * h2f(p.sv[col0 + i]);
            orow[i] = f2h(u);
        }
        return;
    }

    // ---- hadK over the K axis: D[j][c] = sum_i hadK[j][i] * t[i][c], 16 columns per MFMA tile
    const int P = 1 << p.logP;
    const int lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, c16 = lane & 15;
    const int njt = (p.K + 15) >> 4, nkc = (p.K + 31) >> 5;
    for (int ct = wave; ct < (P >> 4); ct += NT / 64) {
        const int c = (ct << 4) + c16;
        for (int jt = 0; jt < njt; jt++) {
            float4_t acc{0.f, 0.f, 0.f, 0.f};
            const int ja = (jt << 4) + c16;  // A operand: row of hadK
            for (int kc = 0; kc < nkc; kc++) {
                half8_t a, bh, bl;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int i = (kc << 5) + (q << 3) + e;
                    const bool in_k = i < p.K;
                    a[e] = (in_k && ja < p.K) ? __builtin_bit_cast(_Float16, p.hadk[ja * p.K + i]) : (_Float16)0.f;
                    const float t = in_k ? buf[i * P + c] * p.pre_scale : 0.f;
                    const _Float16 hi = (_Float16)t;
                    bh[e] = hi;
                    bl[e] = (_Float16)(t - (float)hi);
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bh, acc, 0, 0, 0);
                if (!p.round_mid) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bl, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int j = (jt << 4) + (q << 2) + r;
                if (j < p.K) {
                    float u = round_f16(acc[r]) * p.post_scale;
                    if (p.sv) u = round_f16(u) * h2f(p.sv[col0 + j * P + c]);
                    orow[j * P + c] = f2h(u);
                }
            }
        }
    }
}

int launch_hadamard(const HadParams &p, hipStream_t stream) {
    const int grid = p.rows * (p.n / p.hd);
    const size_t lds = sizeof(float) * (size_t)p.hd;
    if (p.hd <= 1024) {
        hipLaunchKernelGGL((had_kernel<256>), dim3(grid), dim3(256), lds, stream, p);
    } else {
        static bool attr_set = false;
        if (!attr_set) {  // > 64 KiB of dynamic LDS needs the opt-in (idempotent; races are harmless)
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&had_kernel<1024>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        hipLaunchKernelGGL((had_kernel<1024>), dim3(grid), dim3(1024), lds, stream, p);
    }
    return (int)hipGetLastError();
}

}  // namespace qpal

using namespace qpal;

extern "C" int qpal_hadamard(void *out_f16, const void *in, const void *su, const void *sv, const void *hadk, int rows,
                             int n, int hd, int K, int in_mode, int round_mid, float post_scale, void *stream) {
    if (!out_f16 || !in) return QPAL_E_NULL;
    if (K < 1 || K > 256 || (K > 1 && !hadk)) return K > 1 && !hadk ? QPAL_E_NULL : QPAL_E_PARAM;
    if (in_mode < QPAL_IN_F16 || in_mode > QPAL_IN_SWIGLU_F32) return QPAL_E_PARAM;
    if (rows < 1 || rows > 65535 || n < 1 || hd < 1 || n % hd || hd % K) return QPAL_E_SHAPE;
    const int P = hd / K;
    if (P & (P - 1)) return QPAL_E_SHAPE;
    if (K > 1 && P < 16) return QPAL_E_SHAPE;
    if (hd < 4 || hd * sizeof(float) > 160 * 1024) return QPAL_E_SHAPE;
    int logP = 0;
    while ((1 << logP) < P) logP++;
    const uintptr_t al = reinterpret_cast<uintptr_t>(out_f16) | reinterpret_cast<uintptr_t>(su) |
                         reinterpret_cast<uintptr_t>(sv) | reinterpret_cast<uintptr_t>(hadk);
    if ((al & 1) || (reinterpret_cast<uintptr_t>(in) & (in_mode == QPAL_IN_F16 ? 1 : 3))) return QPAL_E_ALIGN;
    HadParams p{static_cast<uint16_t *>(out_f16), in, static_cast<const uint16_t *>(su), static_cast<const uint16_t *>(sv),
                static_cast<const uint16_t *>(hadk), rows, n, hd, K, logP, in_mode, round_mid ? 1 : 0,
                (float)(1.0 / sqrt((double)hd)), post_scale};
    return launch_hadamard(p, static_cast<hipStream_t>(stream));
}
