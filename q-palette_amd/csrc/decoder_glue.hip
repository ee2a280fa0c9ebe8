// Decoder-block glue of a batch-1 decode step that sits between the quantized projections (the harness
// side of the path).  The reference leaves these steps to torch.compile / Triton (eval/measure_latency.py:220-225); as plain
// torch ops they are ~35 tiny launches per layer on MI355X (perf/decode_llama.py: 4.8 ms of a 6.8 ms step).
//
// qpal_rope_kv: one launch = fp32 q|k|v of the new token (the GEMV epilogue's output) -> rotary embedding on q and k
// (HF convention: rotate_half, cos/sin of position * inv_freq rounded to fp16, fp16 arithmetic like the reference's
// model/llama.py apply_rotary_pos_emb) -> q as fp16, k and v written into the static KV cache at `pos`.
#include <hip/hip_runtime.h>

#include "qpal_common.h"

namespace qpal {

struct RopeParams {
    const float *q, *k, *v;
    uint16_t *q_out, *kcache, *vcache;
    const long *pos;
    const float *inv_freq;
    int nq, nkv, hd;
    long max_len;
};

__global__ __launch_bounds__(64) void rope_kv_kernel(const RopeParams p) {
    const int head = blockIdx.x, half = p.hd / 2;
    const long pos = *p.pos;
    for (int i = threadIdx.x; i < half; i += 64) {
        if (head < p.nq + p.nkv) {
            const float *src = head < p.nq ? p.q + (long)head * p.hd : p.k + (long)(head - p.nq) * p.hd;
            const float ang = (float)pos * p.inv_freq[i];
            const _Float16 c = (_Float16)cosf(ang), s = (_Float16)sinf(ang);
            const _Float16 x1 = (_Float16)src[i], x2 = (_Float16)src[i + half];
            const _Float16 o1 = x1 * c + (-x2) * s, o2 = x2 * c + x1 * s;
            uint16_t *dst = head < p.nq ? p.q_out + (long)head * p.hd
                                        : p.kcache + ((long)(head - p.nq) * p.max_len + pos) * p.hd;
            dst[i] = __builtin_bit_cast(uint16_t, o1);
            dst[i + half] = __builtin_bit_cast(uint16_t, o2);
        } else {
            const int kh = head - p.nq - p.nkv;
            const float *src = p.v + (long)kh * p.hd;
            uint16_t *dst = p.vcache + ((long)kh * p.max_len + pos) * p.hd;
            dst[i] = __builtin_bit_cast(uint16_t, (_Float16)src[i]);
            dst[i + half] = __builtin_bit_cast(uint16_t, (_Float16)src[i + half]);
        }
    }
}

// qpal_attn_decode: attention of ONE new token over a static KV cache (batch 1, grouped-query heads), one launch: workgroup =
// one query head; scores q . k[t] / sqrt(hd) for t <= pos (fp32), softmax, out = sum p[t] v[t] (fp32 accumulate, fp16 out).
// torch's scaled_dot_product_attention decomposes this shape into ~10 launches (two batched GEMMs, mask and softmax kernels).
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));

struct AttnParams {
    const uint16_t *q, *kcache, *vcache;
    uint16_t *out;
    const long *pos;
    int nq, nkv, hd;
    long max_len;
    float scale;
};

__global__ __launch_bounds__(256) void attn_decode_kernel(const AttnParams p) {
    extern __shared__ float sh[];  // scores [max_len] | q [hd] | partial out [4][hd] | reduce [8]
    const int head = blockIdx.x, kh = head / (p.nq / p.nkv), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long len = *p.pos + 1;
    float *sc = sh, *qs = sh + p.max_len, *po = qs + p.hd, *red = po + 4 * p.hd;
    for (int d = tid; d < p.hd; d += 256) qs[d] = (float)__builtin_bit_cast(_Float16, p.q[(long)head * p.hd + d]) * p.scale;
    __syncthreads();
    const uint16_t *K = p.kcache + (long)kh * p.max_len * p.hd, *V = p.vcache + (long)kh * p.max_len * p.hd;
    // scores: one position per thread and trip, 16-byte loads along the head dimension
    float mx = -3.0e38f;
    for (long t = tid; t < len; t += 256) {
        const u32x4 *row = reinterpret_cast<const u32x4 *>(K + t * p.hd);
        float acc = 0.f;
        for (int c = 0; c < p.hd / 8; c++) {
            const h8_t h = __builtin_bit_cast(h8_t, row[c]);
            const float *qc = qs + 8 * c;
            acc += (float)h[0] * qc[0] + (float)h[1] * qc[1] + (float)h[2] * qc[2] + (float)h[3] * qc[3] + (float)h[4] * qc[4] +
                   (float)h[5] * qc[5] + (float)h[6] * qc[6] + (float)h[7] * qc[7];
        }
        sc[t] = acc;
        mx = acc > mx ? acc : mx;
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) { const float o = __shfl_xor(mx, sft, 64); mx = o > mx ? o : mx; }
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
    for (int w = 1; w < 4; w++) mx = red[w] > mx ? red[w] : mx;
    float sum = 0.f;
    for (long t = tid; t < len; t += 256) {
        const float e = __expf(sc[t] - mx);
        sc[t] = e;
        sum += e;
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) sum += __shfl_xor(sum, sft, 64);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    sum = red[4] + red[5] + red[6] + red[7];
    // out: wave w takes positions t = w (mod 4); lane owns two adjacent dims (a wave reads 256 contiguous bytes per position)
    for (int d0 = 2 * lane; d0 < p.hd; d0 += 128) {
        float a0 = 0.f, a1 = 0.f;
        for (long t = wave; t < len; t += 4) {
            const h2_t h = __builtin_bit_cast(h2_t, *reinterpret_cast<const uint32_t *>(V + t * p.hd + d0));
            const float w = sc[t];
            a0 += w * (float)h[0];
            a1 += w * (float)h[1];
        }
        po[wave * p.hd + d0] = a0;
        po[wave * p.hd + d0 + 1] = a1;
    }
    __syncthreads();
    for (int d = tid; d < p.hd; d += 256) {
        const float v = (po[d] + po[p.hd + d] + po[2 * p.hd + d] + po[3 * p.hd + d]) / sum;
        p.out[(long)head * p.hd + d] = __builtin_bit_cast(uint16_t, (_Float16)v);
    }
}

// qpal_attn_rope_decode: the two launches above as ONE, and the form the decode harness runs.  Workgroup = one query head,
// 16 waves.  Prologue: rotary embedding of this head's q (kept as packed fp16 in registers) and of its kv head's new k, v
// rounded to fp16 — every workgroup of a kv group computes the new row for itself (128 elements) and reads it from LDS, the
// first one also writes it into the cache for the steps to come, so nothing read in this launch was written by it.
// Scores: HD/8 lanes share one cache row (16 bytes each: a wave instruction reads whole 256-byte rows), v_dot2 against the
// q slice, xor-reduce over the lane group.  Values: a lane owns HD/64 adjacent dims, the waves stride over positions with four
// rows in flight.  q k^T is scaled after the fp32 dot (SDPA semantics).
struct AttnRopeParams {
    const float *q, *k, *v;        // fp32 [nq*hd], [nkv*hd], [nkv*hd]: the new token (GEMV epilogue output)
    uint16_t *kcache, *vcache;     // fp16 [nkv][max_len][hd]
    uint16_t *out;                 // fp16 [nq][hd]
    const long *pos;
    const float *inv_freq;         // fp32 [hd/2]
    int nq, nkv;
    long max_len;
    float scale;
};

template <int HD>
__global__ __launch_bounds__(1024) void attn_rope_decode_kernel(const AttnRopeParams p) {
    constexpr int NT = 1024, NW = 16, HALF = HD / 2;
    constexpr int LPR = HD / 8;        // lanes per cache row in the score loop
    constexpr int RPW = 64 / LPR;      // rows per wave instruction
    constexpr int DPL = HD / 64;       // dims per lane in the value loop
    static_assert(HD == 64 || HD == 128 || HD == 256, "head dims of the Llama family");
    extern __shared__ float sh[];  // scores [max_len] | q, new k (fp16 bits) [HD/2 dwords each] | new v [HD] | partial out [NW][HD] | reduce [2 NW]
    float *sc = sh, *vn = sh + p.max_len + 2 * HALF, *po = vn + HD, *red = po + NW * HD;
    uint32_t *qh = reinterpret_cast<uint32_t *>(sh + p.max_len), *knh = qh + HALF;
    const int head = blockIdx.x, rep = p.nq / p.nkv, kh = head / rep;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long pos = *p.pos;
    const gptr<const uint16_t> K = as_global(p.kcache) + (long)kh * p.max_len * HD, V = as_global(p.vcache) + (long)kh * p.max_len * HD;

    // ---- new token: rope(q) -> LDS as fp16 pairs, rope(k), v -> LDS (+ cache, first head of the group)
    if (tid < 2 * HALF) {
        const bool is_k = tid >= HALF;
        const int i = is_k ? tid - HALF : tid;
        const float *src = is_k ? p.k + (long)kh * HD : p.q + (long)head * HD;
        const float ang = (float)pos * p.inv_freq[i];
        const _Float16 c = (_Float16)cosf(ang), s = (_Float16)sinf(ang);
        const _Float16 x1 = (_Float16)src[i], x2 = (_Float16)src[i + HALF];
        const _Float16 o1 = x1 * c + (-x2) * s, o2 = x2 * c + x1 * s;
        if (is_k) {
            reinterpret_cast<uint16_t *>(knh)[i] = __builtin_bit_cast(uint16_t, o1);
            reinterpret_cast<uint16_t *>(knh)[i + HALF] = __builtin_bit_cast(uint16_t, o2);
            if (head % rep == 0) {
                uint16_t *dst = p.kcache + ((long)kh * p.max_len + pos) * HD;
                dst[i] = __builtin_bit_cast(uint16_t, o1);
                dst[i + HALF] = __builtin_bit_cast(uint16_t, o2);
            }
        } else {
            reinterpret_cast<uint16_t *>(qh)[i] = __builtin_bit_cast(uint16_t, o1);
            reinterpret_cast<uint16_t *>(qh)[i + HALF] = __builtin_bit_cast(uint16_t, o2);
        }
    } else if (tid < 2 * HALF + HD) {
        const int d = tid - 2 * HALF;
        const _Float16 hv = (_Float16)p.v[(long)kh * HD + d];
        vn[d] = (float)hv;
        if (head % rep == 0) p.vcache[((long)kh * p.max_len + pos) * HD + d] = __builtin_bit_cast(uint16_t, hv);
    }
    __syncthreads();

    // ---- scores of the cached positions t < pos
    const int grp = lane / LPR, sl = lane % LPR;  // row inside the wave instruction, 16-byte slice of the row
    const u32x4 qv = *reinterpret_cast<const u32x4 *>(qh + 4 * sl);
    float mx = -3.0e38f;
    for (long t0 = (long)wave * RPW; t0 < pos; t0 += (long)NW * RPW * 4) {
        u32x4 kv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long t = t0 + (long)u * NW * RPW + grp;
            kv[u] = *(gptr<const u32x4>)(K + (t < pos ? t : 0) * HD + 8 * sl);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long t = t0 + (long)u * NW * RPW + grp;
            float a = fdot2(kv[u].x, qv.x, 0.f);
            a = fdot2(kv[u].y, qv.y, a);
            a = fdot2(kv[u].z, qv.z, a);
            a = fdot2(kv[u].w, qv.w, a);
#pragma unroll
            for (int sft = 1; sft < LPR; sft <<= 1) a += __shfl_xor(a, sft, 64);
            a *= p.scale;
            if (t < pos) {
                if (sl == 0) sc[t] = a;
                mx = a > mx ? a : mx;
            }
        }
    }
    if (wave == NW - 1) {  // the new position, from LDS: the first lane group of the last wave
        float a = 0.f;
        if (grp == 0) {
            const u32x4 kn = *reinterpret_cast<const u32x4 *>(knh + 4 * sl);
            a = fdot2(kn.x, qv.x, 0.f);
            a = fdot2(kn.y, qv.y, a);
            a = fdot2(kn.z, qv.z, a);
            a = fdot2(kn.w, qv.w, a);
        }
#pragma unroll
        for (int sft = 1; sft < LPR; sft <<= 1) a += __shfl_xor(a, sft, 64);
        a *= p.scale;
        if (lane == 0) sc[pos] = a;
        if (grp == 0) mx = a > mx ? a : mx;
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) { const float o = __shfl_xor(mx, sft, 64); mx = o > mx ? o : mx; }
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < NW; w++) mx = red[w] > mx ? red[w] : mx;
    float sum = 0.f;
    for (long t = tid; t <= pos; t += NT) {
        const float e = __expf(sc[t] - mx);
        sc[t] = e;
        sum += e;
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) sum += __shfl_xor(sum, sft, 64);
    if (lane == 0) red[NW + wave] = sum;
    __syncthreads();
    sum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) sum += red[NW + w];

    // ---- values: wave w takes positions w, w + 16, ...; four rows in flight
    float acc[DPL];
#pragma unroll
    for (int e = 0; e < DPL; e++) acc[e] = 0.f;
    typedef uint16_t dims_t __attribute__((ext_vector_type(DPL)));
    for (long t0 = wave; t0 < pos; t0 += NW * 4) {
        uint16_t raw[4][DPL];
        float w4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long t = t0 + (long)u * NW;
            const gptr<const uint16_t> row = V + (t < pos ? t : 0) * HD + DPL * lane;
            if constexpr (DPL == 1) raw[u][0] = row[0];
            else if constexpr (DPL == 2) { const uint32_t r = *(gptr<const uint32_t>)row; raw[u][0] = (uint16_t)r; raw[u][1] = (uint16_t)(r >> 16); }
            else { const u32x2 r = *(gptr<const u32x2>)row; raw[u][0] = (uint16_t)r.x; raw[u][1] = (uint16_t)(r.x >> 16); raw[u][2] = (uint16_t)r.y; raw[u][3] = (uint16_t)(r.y >> 16); }
            w4[u] = t < pos ? sc[t] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int e = 0; e < DPL; e++) acc[e] += w4[u] * (float)__builtin_bit_cast(_Float16, raw[u][e]);
    }
    if (wave == 0) {
        const float wn = sc[pos];
#pragma unroll
        for (int e = 0; e < DPL; e++) acc[e] += wn * vn[DPL * lane + e];
    }
#pragma unroll
    for (int e = 0; e < DPL; e++) po[wave * HD + DPL * lane + e] = acc[e];
    __syncthreads();
    if (tid < HD) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w++) v += po[w * HD + tid];
        p.out[(long)head * HD + tid] = __builtin_bit_cast(uint16_t, (_Float16)(v / sum));
    }
}

}  // namespace qpal

using namespace qpal;

extern "C" int qpal_attn_decode(const void *q_f16, const void *kcache_f16, const void *vcache_f16, void *out_f16, const long *pos,
                                int nq, int nkv, int hd, long max_len, float scale, void *stream) {
    if (!q_f16 || !kcache_f16 || !vcache_f16 || !out_f16 || !pos) return QPAL_E_NULL;
    if (nq < 1 || nkv < 1 || nq % nkv || hd < 8 || hd % 8 || max_len < 1) return QPAL_E_SHAPE;
    const size_t lds = sizeof(float) * ((size_t)max_len + 5 * (size_t)hd + 8);
    if (lds > 160 * 1024) return QPAL_E_SHAPE;  // ~40 k positions: longer contexts need the split-context form
    if ((reinterpret_cast<uintptr_t>(kcache_f16) | reinterpret_cast<uintptr_t>(vcache_f16)) & 15) return QPAL_E_ALIGN;
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (lds > 64 * 1024 && (dev < 0 || !attr_set[dev])) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0) attr_set[dev] = true;
    }
    AttnParams p{static_cast<const uint16_t *>(q_f16), static_cast<const uint16_t *>(kcache_f16),
                 static_cast<const uint16_t *>(vcache_f16), static_cast<uint16_t *>(out_f16), pos, nq, nkv, hd, max_len, scale};
    hipLaunchKernelGGL(attn_decode_kernel, dim3(nq), dim3(256), lds, static_cast<hipStream_t>(stream), p);
    return (int)hipGetLastError();
}

extern "C" int qpal_rope_kv(const float *q, const float *k, const float *v, void *q_out_f16, void *kcache_f16, void *vcache_f16,
                            const long *pos, const float *inv_freq, int nq, int nkv, int hd, long max_len, void *stream) {
    if (!q || !k || !v || !q_out_f16 || !kcache_f16 || !vcache_f16 || !pos || !inv_freq) return QPAL_E_NULL;
    if (nq < 1 || nkv < 1 || hd < 2 || hd % 2 || max_len < 1) return QPAL_E_SHAPE;
    RopeParams p{q, k, v, static_cast<uint16_t *>(q_out_f16), static_cast<uint16_t *>(kcache_f16),
                 static_cast<uint16_t *>(vcache_f16), pos, inv_freq, nq, nkv, hd, max_len};
    hipLaunchKernelGGL(rope_kv_kernel, dim3(nq + 2 * nkv), dim3(64), 0, static_cast<hipStream_t>(stream), p);
    return (int)hipGetLastError();
}

extern "C" int qpal_attn_rope_decode(const float *q, const float *k, const float *v, void *kcache_f16, void *vcache_f16, void *out_f16,
                                     const long *pos, const float *inv_freq, int nq, int nkv, int hd, long max_len, float scale,
                                     void *stream) {
    if (!q || !k || !v || !kcache_f16 || !vcache_f16 || !out_f16 || !pos || !inv_freq) return QPAL_E_NULL;
    if (nq < 1 || nkv < 1 || nq % nkv || max_len < 1 || (hd != 64 && hd != 128 && hd != 256)) return QPAL_E_SHAPE;
    const size_t lds = sizeof(float) * ((size_t)max_len + 2 * hd + 16 * (size_t)hd + 32);
    if (lds > 160 * 1024) return QPAL_E_SHAPE;  // ~38 k positions: longer contexts need a split-context form
    if ((reinterpret_cast<uintptr_t>(kcache_f16) | reinterpret_cast<uintptr_t>(vcache_f16)) & 15) return QPAL_E_ALIGN;
    if (max_len % 4) return QPAL_E_ALIGN;  // the fp16 q block behind the scores is read with 16-byte LDS loads
    AttnRopeParams p{q, k, v, static_cast<uint16_t *>(kcache_f16), static_cast<uint16_t *>(vcache_f16),
                     static_cast<uint16_t *>(out_f16), pos, inv_freq, nq, nkv, max_len, scale};
    auto launch = [&](auto kern) -> int {
        static bool attr_set[64] = {};  // one latch per instantiation (the lambda is instantiated per kernel type)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
        if (lds > 64 * 1024 && (dev < 0 || !attr_set[dev])) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            if (dev >= 0) attr_set[dev] = true;
        }
        hipLaunchKernelGGL(kern, dim3(nq), dim3(1024), lds, static_cast<hipStream_t>(stream), p);
        return (int)hipGetLastError();
    };
    if (hd == 64) return launch(attn_rope_decode_kernel<64>);
    if (hd == 128) return launch(attn_rope_decode_kernel<128>);
    return launch(attn_rope_decode_kernel<256>);
}
