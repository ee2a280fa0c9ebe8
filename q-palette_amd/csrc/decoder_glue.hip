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
    if (pos < 0 || pos >= p.max_len) return;  // a position outside the cache: nothing is written (no out-of-bounds access)
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
    if (len < 1 || len > p.max_len) return;  // position outside the cache: out is left untouched
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
    mx = wave_max(mx);
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
    sum = wave_sum(sum);
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
    int log2_rep;                  // log2(nq / nkv) when that is a power of two (host), else -1: kv head of a q head without a division
};

// What a thread of the one-head body reads from global memory that does NOT depend on the position: its element of inv_freq and of
// the new q / k (or v) row, and its first round of cache rows for the score and the value loops (rows past `pos` are masked later;
// any row below max_len is readable).  Round 4: requested at the top of the kernel, BEFORE *pos is waited for — the body was a
// chain of five dependent round trips (arguments -> *pos -> inv_freq -> q / k -> cache rows), ~2 of its 5.6 us.
__device__ __forceinline__ int attn_kv_head(const AttnRopeParams &p, int head) {
    return p.log2_rep >= 0 ? head >> p.log2_rep : head / (p.nq / p.nkv);
}
// the q head that writes its kv head's new cache row: the first of the group
__device__ __forceinline__ bool attn_first_of_group(const AttnRopeParams &p, int head) {
    return p.log2_rep >= 0 ? (head & ((1 << p.log2_rep) - 1)) == 0 : head % (p.nq / p.nkv) == 0;
}

template <int HD>
struct AttnPre {
    uint32_t inv, x1, x2;   // fp32 bit patterns
    u32x4 k4[4];
    u32x2 vraw[4];          // DPL = HD / 64 halves per lane: 2, 4 or 8 bytes (the low ones are used)
};
// (inline asm, in this order: the compiler neither sinks the loads below the wait for *pos nor reorders the cold cache rows in
// front of the three small reads the rotary embedding waits for; attn_landed_* are the matching waits)
template <int HD, int NW>
__device__ __forceinline__ void attn_prefetch(const AttnRopeParams &p, const int head, AttnPre<HD> &pre) {
    constexpr int HALF = HD / 2, LPR = HD / 8, RPW = 64 / LPR, DPL = HD / 64;
    const int kh = attn_kv_head(p, head);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint16_t *K = p.kcache + (long)kh * p.max_len * HD, *V = p.vcache + (long)kh * p.max_len * HD;
    {
        // unconditional loads from clamped addresses: threads [0, HALF) the q row, [HALF, 2 HALF) the k row, [2 HALF, 2 HALF + HD)
        // the v row, the rest re-read element 0
        const bool is_q = tid < HALF, is_k = !is_q && tid < 2 * HALF, is_v = tid >= 2 * HALF && tid < 2 * HALF + HD;
        const float *src = is_q ? p.q + (long)head * HD : is_k ? p.k + (long)kh * HD : p.v + (long)kh * HD;
        const int i = is_q ? tid : is_k ? tid - HALF : is_v ? tid - 2 * HALF : 0;
        const float *a_inv = p.inv_freq + (i < HALF ? i : 0), *a1 = src + i, *a2 = src + ((is_q || is_k) ? i + HALF : i);
        asm volatile("global_load_dword %0, %1, off" : "=v"(pre.inv) : "v"(a_inv));
        asm volatile("global_load_dword %0, %1, off" : "=v"(pre.x1) : "v"(a1));
        asm volatile("global_load_dword %0, %1, off" : "=v"(pre.x2) : "v"(a2));
    }
    const int grp = lane / LPR, sl = lane % LPR;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const long t = (long)wave * RPW + (long)u * NW * RPW + grp;
        const uint16_t *a = K + (t < p.max_len ? t : 0) * HD + 8 * sl;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pre.k4[u]) : "v"(a));
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const long t = wave + (long)u * NW;
        const uint16_t *a = V + (t < p.max_len ? t : 0) * HD + DPL * lane;
        if constexpr (DPL == 1) asm volatile("global_load_ushort %0, %1, off" : "=v"(pre.vraw[u].x) : "v"(a));
        else if constexpr (DPL == 2) asm volatile("global_load_dword %0, %1, off" : "=v"(pre.vraw[u].x) : "v"(a));
        else asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pre.vraw[u]) : "v"(a));
    }
}
// the three small reads have landed (8 younger prefetch loads may be outstanding)
template <int HD>
__device__ __forceinline__ void attn_landed_small(AttnPre<HD> &pre) {
    asm volatile("s_waitcnt vmcnt(8)" : "+v"(pre.inv), "+v"(pre.x1), "+v"(pre.x2)::"memory");
}
// the key rows have landed (the 4 value-row loads are younger; the cache-append stores issued since then only make this stricter)
template <int HD>
__device__ __forceinline__ void attn_landed_keys(AttnPre<HD> &pre) {
    asm volatile("s_waitcnt vmcnt(4)" : "+v"(pre.k4[0]), "+v"(pre.k4[1]), "+v"(pre.k4[2]), "+v"(pre.k4[3])::"memory");
}
template <int HD>
__device__ __forceinline__ void attn_landed_values(AttnPre<HD> &pre) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pre.vraw[0]), "+v"(pre.vraw[1]), "+v"(pre.vraw[2]), "+v"(pre.vraw[3])::"memory");
}

// body: one workgroup of NW waves = one query head; `cap` = positions the score buffer holds (pos < cap)
template <int HD, int NW, bool PRE = false>
__device__ __forceinline__ void attn_rope_head(const AttnRopeParams &p, float *sh, const int head, const long pos, const long cap,
                                               AttnPre<HD> *pre_p = nullptr, const bool ok = true) {
    // ok == false (PRE form only: *pos was outside the cache; the caller passes pos = 0): the body runs — no early return stands
    // between the prefetch and its uses, which is what lets the compiler issue the prefetch before it waits for *pos — and
    // writes nothing
    constexpr int NT = 64 * NW, HALF = HD / 2;
    constexpr int LPR = HD / 8;        // lanes per cache row in the score loop
    constexpr int RPW = 64 / LPR;      // rows per wave instruction
    constexpr int DPL = HD / 64;       // dims per lane in the value loop
    static_assert(HD == 64 || HD == 128 || HD == 256, "head dims of the Llama family");
    // sh: scores [cap] | q, new k (fp16 bits) [HD/2 dwords each] | new v [HD] | partial out [NW][HD] | reduce [2 NW]
    float *sc = sh, *vn = sh + cap + 2 * HALF, *po = vn + HD, *red = po + NW * HD;
    uint32_t *qh = reinterpret_cast<uint32_t *>(sh + cap), *knh = qh + HALF;
    const int kh = attn_kv_head(p, head);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const gptr<const uint16_t> K = as_global(p.kcache) + (long)kh * p.max_len * HD, V = as_global(p.vcache) + (long)kh * p.max_len * HD;

    if constexpr (PRE) attn_landed_small(*pre_p);
    // ---- new token: rope(q) -> LDS as fp16 pairs, rope(k), v -> LDS (+ cache, first head of the group)
    for (int idx = tid; idx < 2 * HALF + HD; idx += NT) {
        if (idx < 2 * HALF) {
            const bool is_k = idx >= HALF;
            const int i = is_k ? idx - HALF : idx;
            const float *src = is_k ? p.k + (long)kh * HD : p.q + (long)head * HD;
            float inv_f, f1, f2;
            if constexpr (PRE) {  // (NT >= 2 HD: one round, idx == tid)
                inv_f = __builtin_bit_cast(float, pre_p->inv); f1 = __builtin_bit_cast(float, pre_p->x1); f2 = __builtin_bit_cast(float, pre_p->x2);
            }
            else { inv_f = p.inv_freq[i]; f1 = src[i]; f2 = src[i + HALF]; }
            const float ang = (float)pos * inv_f;
            const _Float16 c = (_Float16)cosf(ang), s = (_Float16)sinf(ang);
            const _Float16 x1 = (_Float16)f1, x2 = (_Float16)f2;
            const _Float16 o1 = x1 * c + (-x2) * s, o2 = x2 * c + x1 * s;
            uint16_t *dst16 = reinterpret_cast<uint16_t *>(is_k ? knh : qh);
            dst16[i] = __builtin_bit_cast(uint16_t, o1);
            dst16[i + HALF] = __builtin_bit_cast(uint16_t, o2);
            if (is_k && attn_first_of_group(p, head) && ok) {
                uint16_t *dst = p.kcache + ((long)kh * p.max_len + pos) * HD;
                dst[i] = __builtin_bit_cast(uint16_t, o1);
                dst[i + HALF] = __builtin_bit_cast(uint16_t, o2);
            }
        } else {
            const int d = idx - 2 * HALF;
            float fv;
            if constexpr (PRE) fv = __builtin_bit_cast(float, pre_p->x1);
            else fv = p.v[(long)kh * HD + d];
            const _Float16 hv = (_Float16)fv;
            vn[d] = (float)hv;
            if (attn_first_of_group(p, head) && ok) p.vcache[((long)kh * p.max_len + pos) * HD + d] = __builtin_bit_cast(uint16_t, hv);
        }
    }
    __syncthreads();

    // ---- scores of the cached positions t < pos
    const int grp = lane / LPR, sl = lane % LPR;  // row inside the wave instruction, 16-byte slice of the row
    const u32x4 qv = *reinterpret_cast<const u32x4 *>(qh + 4 * sl);
    float mx = -3.0e38f;
    if constexpr (PRE) attn_landed_keys(*pre_p);
    for (long t0 = (long)wave * RPW; t0 < pos; t0 += (long)NW * RPW * 4) {
        u32x4 kv[4];
        bool have = false;
        if constexpr (PRE) have = t0 == (long)wave * RPW;  // the first round was requested at the top of the kernel
        if (have) {
            if constexpr (PRE) {
#pragma unroll
                for (int u = 0; u < 4; u++) kv[u] = pre_p->k4[u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const long t = t0 + (long)u * NW * RPW + grp;
                kv[u] = *(gptr<const u32x4>)(K + (t < pos ? t : 0) * HD + 8 * sl);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long t = t0 + (long)u * NW * RPW + grp;
            float a = fdot2(kv[u].x, qv.x, 0.f);
            a = fdot2(kv[u].y, qv.y, a);
            a = fdot2(kv[u].z, qv.z, a);
            a = fdot2(kv[u].w, qv.w, a);
            a = group_sum<LPR>(a);
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
        a = group_sum<LPR>(a);
        a *= p.scale;
        if (lane == 0) sc[pos] = a;
        if (grp == 0) mx = a > mx ? a : mx;
    }
    mx = wave_max(mx);
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
    sum = wave_sum(sum);
    if (lane == 0) red[NW + wave] = sum;
    __syncthreads();
    sum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) sum += red[NW + w];

    // ---- values: wave w takes positions w, w + 16, ...; four rows in flight
    float acc[DPL];
#pragma unroll
    for (int e = 0; e < DPL; e++) acc[e] = 0.f;
    if constexpr (PRE) attn_landed_values(*pre_p);
    for (long t0 = wave; t0 < pos; t0 += NW * 4) {
        uint16_t raw[4][DPL];
        float w4[4];
        bool havev = false;
        if constexpr (PRE) havev = t0 == wave;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long t = t0 + (long)u * NW;
            if (havev) {
                if constexpr (PRE) {
                    const u32x2 r = pre_p->vraw[u];
                    raw[u][0] = (uint16_t)r.x;
                    if constexpr (DPL >= 2) raw[u][1] = (uint16_t)(r.x >> 16);
                    if constexpr (DPL == 4) { raw[u][2] = (uint16_t)r.y; raw[u][3] = (uint16_t)(r.y >> 16); }
                    if (!(t < pos)) {  // a prefetched row past the context: whatever the cache holds there (0 * NaN is NaN)
#pragma unroll
                        for (int e = 0; e < DPL; e++) raw[u][e] = 0;
                    }
                }
            } else {
                const gptr<const uint16_t> row = V + (t < pos ? t : 0) * HD + DPL * lane;
                if constexpr (DPL == 1) raw[u][0] = row[0];
                else if constexpr (DPL == 2) { const uint32_t r = *(gptr<const uint32_t>)row; raw[u][0] = (uint16_t)r; raw[u][1] = (uint16_t)(r >> 16); }
                else { const u32x2 r = *(gptr<const u32x2>)row; raw[u][0] = (uint16_t)r.x; raw[u][1] = (uint16_t)(r.x >> 16); raw[u][2] = (uint16_t)r.y; raw[u][3] = (uint16_t)(r.y >> 16); }
            }
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
    for (int d = tid; d < HD; d += NT) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w++) v += po[w * HD + d];
        if (ok) p.out[(long)head * HD + d] = __builtin_bit_cast(uint16_t, (_Float16)(v / sum));
    }
}

template <int HD>
__global__ __launch_bounds__(1024) void attn_rope_decode_kernel(const AttnRopeParams p) {
    extern __shared__ float sh[];
    // *pos through the scalar cache (constant address space: an s_load; as an ordinary load behind the inline-asm prefetch it became a
    // VECTOR load, and the wait for it — vmcnt(0) — waited for every prefetched cache row as well)
    const long pos_raw = *(const __attribute__((address_space(4))) long *)p.pos;
    AttnPre<HD> pre;
    attn_prefetch<HD, 16>(p, blockIdx.x, pre);  // everything that does not depend on the position is requested before *pos is waited for
    const bool ok = pos_raw >= 0 && pos_raw < p.max_len;  // position outside the cache: nothing is written
    attn_rope_head<HD, 16, true>(p, sh, blockIdx.x, ok ? pos_raw : 0, p.max_len, &pre, ok);
}

// Split-context form of the above for long caches (flash-decoding shape).  One workgroup per query head streams the whole
// K and V of its kv head through ONE compute unit: 2 MiB at 4 k positions = 56 us per layer (measured), 14x the HBM time.
// Here workgroup (kv head, split) takes one chunk of the context for ALL `REP` query heads of the group (K and V are read once
// per group instead of once per head), leaves a partial (max, sum, unnormalised out) per head in a workspace and takes a
// ticket; the last workgroup of a kv head to arrive merges the partials (agent-scope stores and loads: the L2s of the 8 XCDs
// are not coherent with each other) and resets the ticket for the next launch.
constexpr long kAttnPlainBelow = 512;  // positions below which the split-context launch runs the one-workgroup-per-head body

struct AttnSplitParams {
    AttnRopeParams a;
    float *ws;          // [nkv][nsplit][REP][HD + 2] partials, then nkv tickets (zero-filled once by the caller)
    int nsplit, chunk;  // chunk: positions per split (multiple of 64)
};

__device__ __forceinline__ void st_agent_f(float *p, float v) {
    __hip_atomic_store(as_global(reinterpret_cast<unsigned *>(p)), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent_f(const float *p) {
    return __builtin_bit_cast(float, __hip_atomic_load(as_global(reinterpret_cast<const unsigned *>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

template <int HD, int REP, int NW>
__global__ __launch_bounds__(64 * NW) void attn_rope_split_kernel(const AttnSplitParams sp) {
    constexpr int NT = 64 * NW, HALF = HD / 2;
    constexpr int LPR = HD / 8, DPL = HD / 64;
    static_assert(REP * HD <= 1024, "partial-out buffer: NW x REP x HD floats of LDS");
    const AttnRopeParams &p = sp.a;
    extern __shared__ float sh[];  // scores [REP][chunk] | q [REP][HD/2 dwords] | new k [HD/2 dwords] | new v [HD] | partial out [NW][REP][HD] | reduce [2 NW REP] | new-position scores, maxima, sums [3 REP] | flag
    const int CL = sp.chunk;
    float *sc = sh;
    uint32_t *qh = reinterpret_cast<uint32_t *>(sh + REP * CL), *knh = qh + REP * HALF;
    float *vn = reinterpret_cast<float *>(knh + HALF), *po = vn + HD, *red = po + NW * REP * HD;
    float *park = red + 2 * NW * REP, *mxf = park + REP, *sumf = mxf + REP;
    unsigned *flag = reinterpret_cast<unsigned *>(sumf + REP);
    const int kh = blockIdx.x / sp.nsplit, split = blockIdx.x - kh * sp.nsplit;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long pos = *p.pos;
    if (pos < 0 || pos >= p.max_len) return;  // position outside the cache: nothing is read or written, no ticket taken
    // A short context in a long cache: the first nq workgroups run the one-head body, the others leave.  Measured per launch
    // (32 heads, 8 kv heads, hd 128): one-head body 4.9 us at 40 positions, 9.8 at 500, 28.5 at 2000, 62 at 4000, 457 at 32 k;
    // this kernel's group form 10.0 at 800, 13.5 at 2000, 15.4 at 4000, 18.2 at 8000, 35 at 32 k (134 MB: 3.8 TB/s; 8 heads per
    // group: 20 at 8000) — its floor is the partial / ticket / merge round trips at agent scope.
    if (pos < kAttnPlainBelow) {
        if ((int)blockIdx.x < p.nq) attn_rope_head<HD, NW>(p, sh, blockIdx.x, pos, kAttnPlainBelow);
        return;
    }
    // The context that EXISTS is cut evenly over the splits (chunks of a multiple of 64 positions, at most the static `chunk`
    // the LDS score buffer was sized for): a 500-position context in a 32 k cache still spreads over 8 chunks per kv head
    // instead of sitting in one.  Chunks 0 .. neff-1 hold positions <= pos (the last of them the new one); the workgroups of
    // the others leave at once and take no ticket.
    // (32-bit unsigned: positions are below max_len < 2^31 — the 64-bit divisions that stood here were ~400 instructions)
    long cld = (long)((((unsigned)pos + (unsigned)sp.nsplit) / (unsigned)sp.nsplit + 63u) & ~63u);
    if (cld > CL) cld = CL;
    const long c0 = (long)split * cld;
    const int neff = (int)((unsigned)pos / (unsigned)cld) + 1;
    if (split >= neff) return;
    const bool owner = split == neff - 1;                    // this chunk holds the new position
    const long cend = pos < c0 + cld ? pos : c0 + cld;       // cached positions of the chunk: [c0, cend)
    const int nc = cend > c0 ? (int)(cend - c0) : 0;         // their number
    const gptr<const uint16_t> K = as_global(p.kcache) + (long)kh * p.max_len * HD, V = as_global(p.vcache) + (long)kh * p.max_len * HD;
    float *wsp = sp.ws + ((long)(kh * sp.nsplit + split) * REP) * (HD + 2);
    unsigned *ticket = reinterpret_cast<unsigned *>(sp.ws + (long)p.nkv * sp.nsplit * REP * (HD + 2)) + kh;

    {  // every participating chunk holds at least one position (the earlier ones are full, the last one has the new position)
        // ---- new token: rope of the group's REP query heads (+ k, v in the chunk that owns the new position)
        for (int idx = tid; idx < (REP + 1) * HALF; idx += NT) {
            const int hh = idx / HALF, i = idx - hh * HALF;
            const bool is_k = hh == REP;
            if (is_k && !owner) continue;
            const float *src = is_k ? p.k + (long)kh * HD : p.q + (long)(kh * REP + hh) * HD;
            const float ang = (float)pos * p.inv_freq[i];
            const _Float16 c = (_Float16)cosf(ang), s = (_Float16)sinf(ang);
            const _Float16 x1 = (_Float16)src[i], x2 = (_Float16)src[i + HALF];
            const _Float16 o1 = x1 * c + (-x2) * s, o2 = x2 * c + x1 * s;
            uint16_t *dst16 = reinterpret_cast<uint16_t *>(is_k ? knh : qh + hh * HALF);
            dst16[i] = __builtin_bit_cast(uint16_t, o1);
            dst16[i + HALF] = __builtin_bit_cast(uint16_t, o2);
            if (is_k) {
                uint16_t *dst = p.kcache + ((long)kh * p.max_len + pos) * HD;
                dst[i] = __builtin_bit_cast(uint16_t, o1);
                dst[i + HALF] = __builtin_bit_cast(uint16_t, o2);
            }
        }
        if (owner) {
            for (int d = tid; d < HD; d += NT) {
                const _Float16 hv = (_Float16)p.v[(long)kh * HD + d];
                vn[d] = (float)hv;
                p.vcache[((long)kh * p.max_len + pos) * HD + d] = __builtin_bit_cast(uint16_t, hv);
            }
        }
        __syncthreads();

        // ---- scores
        const int grp = lane / LPR, sl = lane % LPR;
        // Scores on the matrix pipe: D[head][position] = Q[head][:] . K[position][:], v_mfma_f32_16x16x32_f16 with the group's
        // query heads as the (zero-padded) 16 rows of A and 16 cache rows as B.  A B fragment is lane (column j = position,
        // q = lane >> 4): 8 consecutive dims 32 kc + 8 q .. of row j = one 16-byte load; the wave's HD/32 loads cover 16 whole
        // rows.  No cross-lane reduction, no VALU beside the scale: the dot-product form (v_dot2 + a 4-step xor reduction per
        // head and row group) made a chunk VALU-bound — 500 positions for 4 heads: 19.7 us on one CU.
        // Per-head bookkeeping (maxima, sums, the new position's weight) lives in LDS, not in per-thread arrays of REP values:
        // a lane keeps only the four heads its accumulator rows hold (8 heads per group used to spill and ran 2.5x slower).
        constexpr int U = 2;    // 16-row tiles in flight per wave
        constexpr int KC = HD / 32;
        typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
        typedef float float4_t __attribute__((ext_vector_type(4)));
        const int mi = lane & 15, mq = lane >> 4;
        float mx4[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};  // heads 4 mq + r
        {
            half8_t afr[KC];
#pragma unroll
            for (int kc = 0; kc < KC; kc++) {
                u32x4 a{0u, 0u, 0u, 0u};
                if (mi < REP) a = *reinterpret_cast<const u32x4 *>(qh + mi * HALF + 16 * kc + 4 * mq);
                afr[kc] = __builtin_bit_cast(half8_t, a);
            }
            for (int t0 = wave * 16; t0 < nc; t0 += NW * 16 * U) {
                u32x4 kb[U][KC];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int t = t0 + u * NW * 16 + mi;
                    const gptr<const uint16_t> row = K + (c0 + (t < nc ? t : 0)) * HD + 8 * mq;
#pragma unroll
                    for (int kc = 0; kc < KC; kc++) kb[u][kc] = *(gptr<const u32x4>)(row + 32 * kc);
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int t = t0 + u * NW * 16 + mi;
                    float4_t d{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kc = 0; kc < KC; kc++)
                        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[kc], __builtin_bit_cast(half8_t, kb[u][kc]), d, 0, 0, 0);
                    if (4 * mq < REP && t < nc) {  // lane (mq, mi): rows 4 mq + r (heads), column mi (position t)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            if (4 * mq + r < REP) {
                                const float a = d[r] * p.scale;
                                sc[(4 * mq + r) * CL + t] = a;
                                mx4[r] = a > mx4[r] ? a : mx4[r];
                            }
                        }
                    }
                }
            }
        }
        if (owner && wave == NW - 1) {  // the new position, from LDS (its score is parked apart: the chunk's nc may equal CL)
            u32x4 kn = u32x4{0u, 0u, 0u, 0u};
            if (grp == 0) kn = *reinterpret_cast<const u32x4 *>(knh + 4 * sl);
#pragma unroll
            for (int h = 0; h < REP; h++) {
                const u32x4 qv = *reinterpret_cast<const u32x4 *>(qh + h * HALF + 4 * sl);
                float a = fdot2(kn.x, qv.x, 0.f);
                a = fdot2(kn.y, qv.y, a);
                a = fdot2(kn.z, qv.z, a);
                a = fdot2(kn.w, qv.w, a);
                a = group_sum<LPR>(a);
                if (lane == 0) park[h] = a * p.scale;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {  // maximum over the 16 positions (lanes) of this lane's head group, per wave
            float m = mx4[r];
            m = group_max<16>(m);
            if (mi == 0 && 4 * mq + r < REP) red[(4 * mq + r) * NW + wave] = m;
        }
        __syncthreads();
        if (tid < REP) {
            float m = red[tid * NW];
#pragma unroll
            for (int w = 1; w < NW; w++) m = red[tid * NW + w] > m ? red[tid * NW + w] : m;
            if (owner) m = park[tid] > m ? park[tid] : m;
            mxf[tid] = m;
        }
        __syncthreads();
        {   // exponentials and their sums: wave -> (head wave % REP, part wave / REP of the chunk)
            constexpr int NP = NW / REP;
            const int hh = wave % REP, pw = wave / REP;
            const float m = mxf[hh];
            float sm = 0.f;
            for (int t = pw * 64 + lane; t < nc; t += 64 * NP) {
                const float e = __expf(sc[hh * CL + t] - m);
                sc[hh * CL + t] = e;
                sm += e;
            }
            sm = wave_sum(sm);
            if (lane == 0) red[NW * REP + wave] = sm;
        }
        __syncthreads();
        if (tid < REP) {
            constexpr int NP = NW / REP;
            float sm = 0.f;
#pragma unroll
            for (int pw = 0; pw < NP; pw++) sm += red[NW * REP + pw * REP + tid];
            const float wn = owner ? __expf(park[tid] - mxf[tid]) : 0.f;
            park[tid] = wn;  // from here on: the new position's softmax weight
            sumf[tid] = sm + wn;
        }

        // ---- values
        float acc[REP][DPL];
#pragma unroll
        for (int h = 0; h < REP; h++)
#pragma unroll
            for (int e = 0; e < DPL; e++) acc[h][e] = 0.f;
        constexpr int UV = REP * DPL > 16 ? 2 : 4;  // rows in flight per wave in the value loop (latency-bound: one wave = a serial chain of loads)
        for (int t0 = wave; t0 < nc; t0 += NW * UV) {
            uint16_t raw[UV][DPL];
#pragma unroll
            for (int u = 0; u < UV; u++) {
                const int t = t0 + u * NW;
                const gptr<const uint16_t> row = V + (c0 + (t < nc ? t : 0)) * HD + DPL * lane;
                if constexpr (DPL == 1) raw[u][0] = row[0];
                else if constexpr (DPL == 2) { const uint32_t r = *(gptr<const uint32_t>)row; raw[u][0] = (uint16_t)r; raw[u][1] = (uint16_t)(r >> 16); }
                else { const u32x2 r = *(gptr<const u32x2>)row; raw[u][0] = (uint16_t)r.x; raw[u][1] = (uint16_t)(r.x >> 16); raw[u][2] = (uint16_t)r.y; raw[u][3] = (uint16_t)(r.y >> 16); }
            }
#pragma unroll
            for (int u = 0; u < UV; u++) {
                const int t = t0 + u * NW;
#pragma unroll
                for (int h = 0; h < REP; h++) {
                    const float w = t < nc ? sc[h * CL + t] : 0.f;
#pragma unroll
                    for (int e = 0; e < DPL; e++) acc[h][e] += w * (float)__builtin_bit_cast(_Float16, raw[u][e]);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < REP; h++)
#pragma unroll
            for (int e = 0; e < DPL; e++) po[(wave * REP + h) * HD + DPL * lane + e] = acc[h][e];
        __syncthreads();  // (also orders park / sumf, written by the first REP threads above, before the reads below)
        for (int idx = tid; idx < REP * HD; idx += NT) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) v += po[w * REP * HD + idx];
            const int h = idx / HD, d = idx - h * HD;
            if (owner) v += park[h] * vn[d];
            if (neff == 1) p.out[(long)(kh * REP) * HD + idx] = __builtin_bit_cast(uint16_t, (_Float16)(v / sumf[h]));  // the only chunk
            else st_agent_f(wsp + h * (HD + 2) + 2 + d, v);
        }
        if (neff == 1) return;  // no partials, no ticket
        if (tid < REP) {
            st_agent_f(wsp + tid * (HD + 2), mxf[tid]);
            st_agent_f(wsp + tid * (HD + 2) + 1, sumf[tid]);
        }
    }
    // ---- ticket: the stores above are write-through; wait for them, then arrive
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(as_global(ticket), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = t == (unsigned)neff - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (*flag == 0u) return;
    // ---- last arriver of this kv head: merge.  Agent-scope loads are ~1 us each: all of a stage's loads are issued
    // together (max / sum of every partial through LDS first, then the out partials eight at a time), never one per iteration.
    const float *base = sp.ws + (long)kh * sp.nsplit * REP * (HD + 2);
    float *ml = sh;  // [nsplit * REP][2]: the score buffer is free now
    for (int i = tid; i < neff * REP; i += NT) {
        const float *pp = base + (long)i * (HD + 2);
        const float m = ld_agent_f(pp), l = ld_agent_f(pp + 1);
        ml[2 * i] = m;
        ml[2 * i + 1] = l;
    }
    __syncthreads();
    for (int idx = tid; idx < REP * HD; idx += NT) {
        const int h = idx / HD, d = idx - h * HD;
        float M = -3.0e38f;
        for (int s = 0; s < neff; s++) {
            const float m = ml[2 * (s * REP + h)];
            M = m > M ? m : M;
        }
        float L = 0.f, o = 0.f;
        for (int s0 = 0; s0 < neff; s0 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int sx = s0 + u < neff ? s0 + u : neff - 1;
                v[u] = ld_agent_f(base + ((long)sx * REP + h) * (HD + 2) + 2 + d);
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int sx = s0 + u;
                if (sx < neff) {
                    const float f = __expf(ml[2 * (sx * REP + h)] - M);
                    L += ml[2 * (sx * REP + h) + 1] * f;
                    o += v[u] * f;
                }
            }
        }
        p.out[(long)(kh * REP + h) * HD + d] = __builtin_bit_cast(uint16_t, (_Float16)(o / L));
    }
    if (tid == 0) __hip_atomic_store(as_global(ticket), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// qpal_lm_head_argmax: the last launches of a greedy decode step as one — final RMSNorm of the fp32 residual stream, the fp16
// lm_head GEMV (vocab x k, the one unquantized matrix of the model: 1.05 GB for Llama-3.1-8B), optional logits, and the argmax.
// A wave owns rows r = gw, gw + nw, ...: 64 lanes x 16 bytes per load, the normalised x held in registers as packed fp16
// (v_dot2), two rows in flight; per-workgroup (max, index) partials go out with agent-scope stores and the last workgroup to
// take a ticket reduces them (ties: the lowest index, like torch.argmax).  As torch ops the same work is a norm (3 launches),
// a hipBLASLt GEMV (186 us), and a 47 us reduction over 128 k logits.
struct LmHeadParams {
    const float *h;          // fp32 [k] residual stream
    const uint16_t *rms_w;   // fp16 [k] or null
    float rms_eps;           // 0: no norm (x = fp16(h))
    const uint16_t *w;       // fp16 [vocab][k]
    float *logits;           // fp32 [vocab] or null
    long *token;             // argmax
    float *ws;               // [2 * grid] partial (max, index as float bits) + ticket (zero-filled once)
    int vocab, k;
};

template <int KD>  // KD = k / 512: dwords of x per lane = 4 KD
__global__ __launch_bounds__(1024) void lm_head_argmax_kernel(const LmHeadParams p) {
    __shared__ __attribute__((aligned(16))) uint16_t xs[KD * 512];
    __shared__ float red[32];
    __shared__ unsigned flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = KD * 512;
    // ---- x = fp16(rmsnorm(h) * w): every workgroup for itself (16 KiB from L2)
    float ss = 0.f;
    for (int i = tid; i < k; i += 1024) {
        const float v = p.h[i];
        ss += v * v;
    }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; w++) tot += red[w];
    const float inv = p.rms_eps > 0.f ? __builtin_amdgcn_rsqf(tot / (float)k + p.rms_eps) : 1.0f;
    for (int i = tid; i < k; i += 1024) {
        const _Float16 xn = (_Float16)(p.h[i] * inv);  // the norm's output dtype is fp16 (HF LlamaRMSNorm), then * weight
        const _Float16 xw = p.rms_w ? xn * __builtin_bit_cast(_Float16, p.rms_w[i]) : xn;
        xs[i] = __builtin_bit_cast(uint16_t, xw);
    }
    __syncthreads();
    u32x4 xr[KD];
#pragma unroll
    for (int c = 0; c < KD; c++) xr[c] = *reinterpret_cast<const u32x4 *>(xs + c * 512 + 8 * lane);
    // ---- rows
    const int gw = blockIdx.x * 16 + wave, nw = gridDim.x * 16;
    float best = -3.0e38f;
    int besti = 0x7fffffff;
    const gptr<const uint16_t> W = as_global(p.w);
    for (int r0 = gw; r0 < p.vocab; r0 += 2 * nw) {
        u32x4 wv[2][KD];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int r = r0 + u * nw;
            const gptr<const uint16_t> row = W + (long)(r < p.vocab ? r : r0) * k + 8 * lane;
#pragma unroll
            for (int c = 0; c < KD; c++) wv[u][c] = __builtin_nontemporal_load((gptr<const u32x4>)(row + c * 512));
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int r = r0 + u * nw;
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < KD; c++) {
                a = fdot2(wv[u][c].x, xr[c].x, a);
                a = fdot2(wv[u][c].y, xr[c].y, a);
                a = fdot2(wv[u][c].z, xr[c].z, a);
                a = fdot2(wv[u][c].w, xr[c].w, a);
            }
            a = wave_sum(a);
            if (r < p.vocab) {
                if (p.logits && lane == 0) p.logits[r] = a;
                if (a > best) best = a, besti = r;  // rows ascend within a wave: strict > keeps the lowest index
            }
        }
    }
    // ---- workgroup argmax, partial out, ticket, final reduction by the last arriver
    if (lane == 0) {
        red[wave] = best;
        red[16 + wave] = __builtin_bit_cast(float, besti);
    }
    __syncthreads();
    if (tid == 0) {
        float b = red[0];
        int bi = __builtin_bit_cast(int, red[16]);
        for (int w = 1; w < 16; w++) {
            const float v = red[w];
            const int vi = __builtin_bit_cast(int, red[16 + w]);
            if (v > b || (v == b && vi < bi)) b = v, bi = vi;
        }
        __hip_atomic_store(as_global(reinterpret_cast<unsigned *>(p.ws + 2 * blockIdx.x)), __builtin_bit_cast(unsigned, b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(as_global(reinterpret_cast<unsigned *>(p.ws + 2 * blockIdx.x + 1)), (unsigned)bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned *ticket = reinterpret_cast<unsigned *>(p.ws + 2 * gridDim.x);
        const unsigned t = __hip_atomic_fetch_add(as_global(ticket), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag = t == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (flag == 0u) return;
    float b = -3.0e38f;
    int bi = 0x7fffffff;
    for (int i = tid; i < (int)gridDim.x; i += 1024) {
        const float v = __builtin_bit_cast(float, __hip_atomic_load(as_global(reinterpret_cast<const unsigned *>(p.ws + 2 * i)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const int vi = (int)__hip_atomic_load(as_global(reinterpret_cast<const unsigned *>(p.ws + 2 * i + 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v > b || (v == b && vi < bi)) b = v, bi = vi;
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
        const float ob = __shfl_xor(b, sh, 64);
        const int oi = __shfl_xor(bi, sh, 64);
        if (ob > b || (ob == b && oi < bi)) b = ob, bi = oi;
    }
    if (lane == 0) {
        red[wave] = b;
        red[16 + wave] = __builtin_bit_cast(float, bi);
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 0; w < 16; w++) {
            const float v = red[w];
            const int vi = __builtin_bit_cast(int, red[16 + w]);
            if (v > b || (v == b && vi < bi)) b = v, bi = vi;
        }
        // NaN or all -inf logits (a diverged step) leave no row selected: the token must still be a VALID row — the harness
        // feeds it to the next step's embedding lookup on the device (torch.argmax returns an in-range index there too)
        if ((unsigned)bi >= (unsigned)p.vocab) bi = 0;
        *p.token = bi;
        __hip_atomic_store(as_global(reinterpret_cast<unsigned *>(p.ws + 2 * gridDim.x)), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// split geometry for a cache of max_len positions (0 splits: use the one-workgroup-per-head kernel)
static void attn_split_geometry(int nq, int nkv, int hd, long max_len, int &nsplit, int &chunk, int &waves, size_t &lds, size_t &ws_bytes) {
    nsplit = 0, chunk = 0, waves = 0, lds = 0, ws_bytes = 0;
    const int rep = nq / nkv;
    if (max_len < 2048 || (rep != 1 && rep != 2 && rep != 4 && rep != 8) || rep * hd > 1024) return;
    int want = 256 / nkv;
    if (want < 2) want = 2;
    if (want > 64) want = 64;
    long c = (max_len + want - 1) / want;
    c = (c + 63) / 64 * 64;
    const int ns = (int)((max_len + c - 1) / c);
    // 16 waves whatever the chunk: with 4 (tried for short chunks) every wave walks 4x the rows, one load latency at a time
    // (8192-position cache at 500 positions: 30 us against 10)
    const int nw = 16;
    const size_t floats = (size_t)rep * c + (size_t)rep * hd / 2 + hd / 2 + hd + nw * (size_t)rep * hd + 2 * nw * (size_t)rep + 3 * rep + 4;
    const size_t plain = (size_t)kAttnPlainBelow + 2 * hd + nw * (size_t)hd + 2 * nw;  // attn_rope_head's layout at cap = kAttnPlainBelow
    const size_t need = floats > plain ? floats : plain;
    if (ns < 2 || nkv * ns < nq || need * sizeof(float) > 160 * 1024) return;
    nsplit = ns, chunk = (int)c, waves = nw, lds = need * sizeof(float);
    ws_bytes = ((size_t)nkv * ns * rep * (hd + 2) + nkv) * sizeof(float);
}

extern "C" long qpal_attn_ws_bytes(int nq, int nkv, int hd, long max_len) {
    if (nq < 1 || nkv < 1 || nq % nkv || max_len < 1) return 0;
    int ns, ch, nw;
    size_t lds, wsb;
    attn_split_geometry(nq, nkv, hd, max_len, ns, ch, nw, lds, wsb);
    return (long)wsb;
}

template <class Kern, class Params>
static int launch_attn(Kern kern, const Params &p, int grid, int threads, size_t lds, void *stream) {
    static bool attr_set[64] = {};  // one latch per instantiation
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (lds > 64 * 1024 && (dev < 0 || !attr_set[dev])) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, static_cast<hipStream_t>(stream), p);
    return (int)hipGetLastError();
}

extern "C" int qpal_attn_rope_decode(const float *q, const float *k, const float *v, void *kcache_f16, void *vcache_f16, void *out_f16,
                                     const long *pos, const float *inv_freq, int nq, int nkv, int hd, long max_len, float scale,
                                     void *ws, long ws_bytes, void *stream) {
    if (!q || !k || !v || !kcache_f16 || !vcache_f16 || !out_f16 || !pos || !inv_freq) return QPAL_E_NULL;
    if (nq < 1 || nkv < 1 || nq % nkv || max_len < 1 || max_len >= (1L << 30) || (hd != 64 && hd != 128 && hd != 256)) return QPAL_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(kcache_f16) | reinterpret_cast<uintptr_t>(vcache_f16)) & 15) return QPAL_E_ALIGN;
    if (max_len % 4) return QPAL_E_ALIGN;  // the fp16 q block behind the scores is read with 16-byte LDS loads
    AttnRopeParams p{q, k, v, static_cast<uint16_t *>(kcache_f16), static_cast<uint16_t *>(vcache_f16),
                     static_cast<uint16_t *>(out_f16), pos, inv_freq, nq, nkv, max_len, scale, -1};
    {
        const int rep_ = nq / nkv;
        if (rep_ > 0 && (rep_ & (rep_ - 1)) == 0) p.log2_rep = __builtin_ctz((unsigned)rep_);
    }
    int ns, ch, nw;
    size_t slds, wsb;
    attn_split_geometry(nq, nkv, hd, max_len, ns, ch, nw, slds, wsb);
    if (ws && ns >= 2) {  // split-context form
        if ((size_t)ws_bytes < wsb) return QPAL_E_SHAPE;
        if (reinterpret_cast<uintptr_t>(ws) & 3) return QPAL_E_ALIGN;
        AttnSplitParams sp{p, static_cast<float *>(ws), ns, ch};
        const int rep = nq / nkv, grid = nkv * ns;
#define QPAL_SPLIT(HD_, REP_) \
    if (hd == HD_ && rep == REP_) return launch_attn(attn_rope_split_kernel<HD_, REP_, 16>, sp, grid, 64 * nw, slds, stream);
        QPAL_SPLIT(64, 1) QPAL_SPLIT(64, 2) QPAL_SPLIT(64, 4) QPAL_SPLIT(64, 8)
        QPAL_SPLIT(128, 1) QPAL_SPLIT(128, 2) QPAL_SPLIT(128, 4) QPAL_SPLIT(128, 8)
        QPAL_SPLIT(256, 1) QPAL_SPLIT(256, 2) QPAL_SPLIT(256, 4)
#undef QPAL_SPLIT
        return QPAL_E_SHAPE;
    }
    const size_t lds = sizeof(float) * ((size_t)max_len + 2 * hd + 16 * (size_t)hd + 32);
    if (lds > 160 * 1024) return QPAL_E_SHAPE;  // ~38 k positions: give a workspace (qpal_attn_ws_bytes) for the split-context form
    if (hd == 64) return launch_attn(attn_rope_decode_kernel<64>, p, nq, 1024, lds, stream);
    if (hd == 128) return launch_attn(attn_rope_decode_kernel<128>, p, nq, 1024, lds, stream);
    return launch_attn(attn_rope_decode_kernel<256>, p, nq, 1024, lds, stream);
}

// workspace of qpal_lm_head_argmax: one (max, index) pair per workgroup + the ticket
static int lm_head_grid(int vocab) {
    int g = (vocab + 31) / 32;  // >= 2 rows per wave
    return g < 1 ? 1 : g > 1024 ? 1024 : g;
}
extern "C" long qpal_lm_head_ws_bytes(int vocab) { return vocab > 0 ? (2L * lm_head_grid(vocab) + 1) * 4 : 0; }

extern "C" int qpal_lm_head_argmax(const float *h_f32, const void *rms_w_f16, float rms_eps, const void *w_f16, float *logits_f32,
                                   long *token, void *ws, long ws_bytes, int vocab, int k, void *stream) {
    if (!h_f32 || !w_f16 || !token || !ws) return QPAL_E_NULL;
    if (vocab < 1 || (k != 2048 && k != 4096 && k != 8192)) return QPAL_E_SHAPE;
    if (ws_bytes < qpal_lm_head_ws_bytes(vocab)) return QPAL_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(w_f16) & 15) || (reinterpret_cast<uintptr_t>(ws) & 3) || (reinterpret_cast<uintptr_t>(h_f32) & 3) ||
        (rms_w_f16 && (reinterpret_cast<uintptr_t>(rms_w_f16) & 1)))
        return QPAL_E_ALIGN;
    LmHeadParams p{h_f32, static_cast<const uint16_t *>(rms_w_f16), rms_eps > 0.f ? rms_eps : 0.f, static_cast<const uint16_t *>(w_f16),
                   logits_f32, token, static_cast<float *>(ws), vocab, k};
    const int grid = lm_head_grid(vocab);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (k == 2048) hipLaunchKernelGGL(lm_head_argmax_kernel<4>, dim3(grid), dim3(1024), 0, s, p);
    else if (k == 4096) hipLaunchKernelGGL(lm_head_argmax_kernel<8>, dim3(grid), dim3(1024), 0, s, p);
    else hipLaunchKernelGGL(lm_head_argmax_kernel<16>, dim3(grid), dim3(1024), 0, s, p);
    return (int)hipGetLastError();
}
