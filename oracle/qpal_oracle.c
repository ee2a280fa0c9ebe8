/*
 * qpal_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY, never shipped, never on the product path).
 *
 * A plain-C restatement of the reference's (snu-mllab/Q-Palette) packed-weight formats and of its
 * "fake-dequant then matmul" CPU path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.  Parity status: PINNED — every decoder below is checked
 * bit-for-bit against golden vectors produced by importing the reference's own Python
 * (tests/golden/make_golden.py -> tests/golden/*.npz, checked by tests/test_oracle_golden.py).
 *
 * Reference files restated (paths relative to /root/reference):
 *   TCQ bitstream        lib/codebook/bitshift.py:296-329 (pack_trellis), lib/quantizer/tcq_quant.py:47-60
 *                        (nibble permutation), lib/utils/kernel_decompress.py:17-61 (decode_indices)
 *   TCQ codebook         lib/codebook/bitshift.py:71-79 (quantlut_sym), kernels/tcq-kernels/src/inference.cu:582-595
 *   mma tile order       lib/algo/ldlq.py:10-13 (_PERMUTE), lib/utils/kernel_decompress.py:10-14
 *   comb / combt         lib/linear/comb_linear.py:35-48,178-191
 *   VQ/SQ tensor-core    lib/quantizer/quant_op.py:101-162 (pack_qweight_routine), :185-244 (inverses)
 *   SQ SIMT              lib/quantizer/pack_op.py:242-335, kernels/sq-cuda-kernels/gemm_routines.cu:436-461
 *   VQ SIMT              lib/quantizer/quant_op.py:15-78, kernels/vq-cuda-kernels/src/gemm_routines.cu:1943-1968
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ fp16 helpers */
static float h2f_table[65536];
static int h2f_ready = 0;

static float half_bits_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f;
    uint32_t man = h & 0x3ffu;
    uint32_t f;
    if (exp == 0) {
        if (man == 0) {
            f = sign;
        } else { /* subnormal */
            int e = -1;
            do { e++; man <<= 1; } while ((man & 0x400u) == 0);
            man &= 0x3ffu;
            f = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        f = sign | 0x7f800000u | (man << 13);
    } else {
        f = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float out;
    memcpy(&out, &f, 4);
    return out;
}

static void ensure_h2f(void) {
    if (h2f_ready) return;
    for (uint32_t i = 0; i < 65536; i++) h2f_table[i] = half_bits_to_float((uint16_t)i);
    h2f_ready = 1;
}

int qo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void qo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* little-endian bit reader: `nbits` (<= 56) bits starting at absolute LE bit position `pos` */
static inline uint64_t le_bits(const uint8_t *p, uint64_t pos, int nbits) {
    uint64_t v = 0;
    uint64_t byte = pos >> 3;
    int sh = (int)(pos & 7);
    int need = (sh + nbits + 7) >> 3; /* <= 8 */
    for (int i = 0; i < need; i++) v |= (uint64_t)p[byte + i] << (8 * i);
    v >>= sh;
    if (nbits < 64) v &= ((uint64_t)1 << nbits) - 1;
    return v;
}

/* position of (row r, col c) of a 16x16 tile in the mma-ordered sequence (ldlq.py:10-13):
 * s = 8*lane + 2*j + e  with lane = 4*(r%8) + (c%8)/2, j = 2*(c/8) + r/8, e = c%2          */
static inline int tile_seq_pos(int r, int c) {
    int lane = 4 * (r & 7) + ((c & 7) >> 1);
    int j = 2 * (c >> 3) + (r >> 3);
    return 8 * lane + 2 * j + (c & 1);
}

/* ------------------------------------------------------------------ TCQ (a1, a2) */
/* Decode the 128 sixteen-bit trellis states of tile (tr, tc) of an m x k matrix.
 * memory order: [supertile-row][supertile-col][lane 0..31][ksub][msub][KV nibbles]; each
 * (lane, ksub, msub) group is a 4*KV-bit little-endian integer whose value is the lane's 4*KV
 * stream bits MSB-first (tcq_quant.py:52-60).  The tile's stream is 128*KV bits, tail-biting:
 * state t = the 16-bit window starting at stream bit t*KV, wrapping (bitshift.py:296-329).   */
static void tcq_tile_states(const uint8_t *bytes, int k, int KV, int tr, int tc, uint16_t *states) {
    int sr = tr >> 1, msub = tr & 1, sc = tc >> 1, ksub = tc & 1;
    uint8_t bit[128 * 10 + 16];
    int nb = 128 * KV;
    for (int lane = 0; lane < 32; lane++) {
        uint64_t base = ((((uint64_t)sr * (k / 32) + sc) * 32 + lane) * 16 + (uint64_t)(ksub * 2 + msub) * 4) * KV;
        uint64_t v = le_bits(bytes, base, 4 * KV);
        for (int b = 0; b < 4 * KV; b++) bit[lane * 4 * KV + b] = (uint8_t)((v >> (4 * KV - 1 - b)) & 1);
    }
    for (int b = 0; b < 16; b++) bit[nb + b] = bit[b];
    for (int t = 0; t < 128; t++) {
        uint32_t s = 0;
        for (int b = 0; b < 16; b++) s = (s << 1) | bit[t * KV + b];
        states[t] = (uint16_t)s;
    }
}

/* states out: [m/16][k/16][128] (tile row-major, state index t = 4*lane + j) */
void qo_tcq_states(const uint16_t *trellis, int m, int k, int KV, uint16_t *states) {
    const uint8_t *bytes = (const uint8_t *)trellis;
    int ntc = k / 16;
#pragma omp parallel for schedule(static)
    for (int tr = 0; tr < m / 16; tr++)
        for (int tc = 0; tc < ntc; tc++)
            tcq_tile_states(bytes, k, KV, tr, tc, states + ((size_t)tr * ntc + tc) * 128);
}

/* state -> (w0, w1) fp16 bit patterns: h = s*(s+1) mod 2^16, idx = (h >> (15-S)) & (2^S-1),
 * w0 sign-flipped when bit 15 of h is set (bitshift.py:71-79; inference.cu:582-595).          */
static inline void tcq_state_to_pair(uint16_t s, const uint16_t *tlut, int S, uint16_t *w0, uint16_t *w1) {
    uint32_t h = ((uint32_t)s * ((uint32_t)s + 1u)) & 0xffffu;
    uint32_t idx = (h >> (15 - S)) & ((1u << S) - 1u);
    *w0 = (uint16_t)(tlut[2 * idx] ^ (h & 0x8000u));
    *w1 = tlut[2 * idx + 1];
}

/* write fp16 W[m][ldw] columns [col0, col0+k) from one trellis stream */
static void tcq_dequant_block(const uint16_t *trellis, const uint16_t *tlut, int m, int k, int S, int KV,
                              uint16_t *W, size_t ldw, int row0, int col0) {
    const uint8_t *bytes = (const uint8_t *)trellis;
    int ntc = k / 16;
#pragma omp parallel for schedule(static)
    for (int tr = 0; tr < m / 16; tr++) {
        uint16_t st[128];
        for (int tc = 0; tc < ntc; tc++) {
            tcq_tile_states(bytes, k, KV, tr, tc, st);
            for (int r = 0; r < 16; r++)
                for (int c = 0; c < 16; c += 2) {
                    int t = tile_seq_pos(r, c) >> 1;
                    uint16_t w0, w1;
                    tcq_state_to_pair(st[t], tlut, S, &w0, &w1);
                    uint16_t *dst = W + (size_t)(row0 + tr * 16 + r) * ldw + col0 + tc * 16 + c;
                    dst[0] = w0;
                    dst[1] = w1;
                }
        }
    }
}

/* split: 0 = single stream c1 at KV1; 1 = comb (rows [0,m/2) from c1@KV1, rows [m/2,m) from c2@KV2);
 *        2 = combt (cols [0,k/2) from c1@KV1, cols [k/2,k) from c2@KV2).  W: fp16 bits [m][k].       */
int qo_tcq_dequant(const uint16_t *c1, const uint16_t *c2, const uint16_t *tlut, int m, int k, int S,
                   int KV1, int KV2, int split, uint16_t *W) {
    if (split == 0) {
        tcq_dequant_block(c1, tlut, m, k, S, KV1, W, (size_t)k, 0, 0);
    } else if (split == 1) {
        tcq_dequant_block(c1, tlut, m / 2, k, S, KV1, W, (size_t)k, 0, 0);
        tcq_dequant_block(c2, tlut, m / 2, k, S, KV2, W, (size_t)k, m / 2, 0);
    } else if (split == 2) {
        tcq_dequant_block(c1, tlut, m, k / 2, S, KV1, W, (size_t)k, 0, 0);
        tcq_dequant_block(c2, tlut, m, k / 2, S, KV2, W, (size_t)k, 0, k / 2);
    } else {
        return -1;
    }
    return 0;
}

/* ------------------------------------------------------------------ VQ/SQ tensor-core format (a4) */
/* qweight int32 [m][bits*k/32/vec]; per (supertile-row, supertile-col, lane, ksub, msub) group:
 * 8/vec codes of `bits` bits, LSB-first (quant_op.py:134-161).  vec=1: code q = 2j+e; vec=2: code j.
 * idx out: int32 [m][k/vec].                                                                    */
void qo_lut_tc_indices(const uint32_t *qweight, int m, int k, int bits, int vec, int32_t *idx) {
    const uint8_t *bytes = (const uint8_t *)qweight;
    int ncode = 8 / vec;           /* codes per (lane, tile) */
    int gbits = ncode * bits;      /* bits per group */
#pragma omp parallel for schedule(static)
    for (int tr = 0; tr < m / 16; tr++)
        for (int tc = 0; tc < k / 16; tc++) {
            int sr = tr >> 1, msub = tr & 1, sc = tc >> 1, ksub = tc & 1;
            for (int lane = 0; lane < 32; lane++) {
                uint64_t base = ((((uint64_t)sr * (k / 32) + sc) * 32 + lane) * 4 + (uint64_t)(ksub * 2 + msub)) * gbits;
                for (int q = 0; q < ncode; q++) {
                    int32_t code = (int32_t)le_bits(bytes, base + (uint64_t)q * bits, bits);
                    int j = (vec == 1) ? (q >> 1) : q;
                    int e = (vec == 1) ? (q & 1) : 0;
                    int r = (lane >> 2) + 8 * (j & 1);
                    int c = 2 * (lane & 3) + 8 * (j >> 1) + e;
                    size_t row = (size_t)tr * 16 + r, col = (size_t)tc * 16 + c;
                    idx[row * (size_t)(k / vec) + col / vec] = code;
                }
            }
        }
}

/* idx [m][k/vec] + lut fp16 [2^bits][vec] -> W fp16 [m][k] */
void qo_lut_gather(const int32_t *idx, const uint16_t *lut, int m, int k, int vec, uint16_t *W) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < m; r++)
        for (int c = 0; c < k / vec; c++) {
            int32_t code = idx[(size_t)r * (k / vec) + c];
            for (int v = 0; v < vec; v++) W[(size_t)r * k + (size_t)c * vec + v] = lut[(size_t)code * vec + v];
        }
}

void qo_lut_tc_dequant(const uint32_t *qweight, const uint16_t *lut, int m, int k, int bits, int vec, uint16_t *W) {
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)m * (k / vec));
    qo_lut_tc_indices(qweight, m, k, bits, vec, idx);
    qo_lut_gather(idx, lut, m, k, vec, W);
    free(idx);
}

/* ------------------------------------------------------------------ SIMT formats (a5, a6) */
/* Generic SIMT layout: rows are row-major; a row is cut into blocks of 32 "lanes"; lane t of a block
 * owns `ngrp` groups of 8 consecutive weights at elements  blk*B + g*8*W + 8t .. +7  (W = lanes in this
 * block: 32, or the partial-block lane count), B = 256*ngrp.  Its 32 codes (each `vec` weights, in
 * group-major order) are packed LSB-first into `bits` u32 words; word j is stored at
 * blk*bits*32 + t + W*j.   SQ: vec=1, ngrp=4 (pack_op.py:288-335).  VQ: ngrp=4*vec (quant_op.py:15-78). */
void qo_simt_indices(const uint32_t *qweight, int m, int k, int bits, int vec, int32_t *idx) {
    int ngrp = 4 * vec;
    int B = 256 * ngrp;             /* elements per full block */
    int per_lane = 32 * vec;        /* elements per lane */
    size_t row_words = (size_t)k * bits / 32 / vec;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < m; r++) {
        const uint32_t *row = qweight + (size_t)r * row_words;
        int nblk = (k + B - 1) / B;
        for (int blk = 0; blk < nblk; blk++) {
            int W = 32;
            if (blk == k / B) W = (k % B) / per_lane;
            for (int t = 0; t < W; t++) {
                uint32_t words[16];
                for (int j = 0; j < bits; j++) words[j] = row[(size_t)blk * bits * 32 + t + (size_t)W * j];
                for (int c = 0; c < 32; c++) {
                    int32_t code = (int32_t)le_bits((const uint8_t *)words, (uint64_t)c * bits, bits);
                    int e0 = c * vec;          /* element offset within the lane's chunk */
                    int g = e0 / 8, off = e0 % 8;
                    size_t elem = (size_t)blk * B + (size_t)g * 8 * W + 8 * t + off;
                    idx[(size_t)r * (k / vec) + elem / vec] = code;
                }
            }
        }
    }
}

void qo_simt_dequant(const uint32_t *qweight, const uint16_t *lut, int m, int k, int bits, int vec, uint16_t *W) {
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)m * (k / vec));
    qo_simt_indices(qweight, m, k, bits, vec, idx);
    qo_lut_gather(idx, lut, m, k, vec, W);
    free(idx);
}

/* ------------------------------------------------------------------ GEMV on decoded weights */
/* out[n][m] = sum_k W[m][k] * x[n][k]; W, x are fp16 bit patterns.  Accumulates in double (the
 * oracle is the numerically "true" value; the kernels accumulate in fp32).  absout[n][m] (nullable)
 * = sum_k |W*x|, the scale the tests' tolerance is stated against.                              */
void qo_gemv_f16(const uint16_t *W, const uint16_t *x, int m, int n, int k, double *out, double *absout) {
    ensure_h2f();
#pragma omp parallel for schedule(static)
    for (int r = 0; r < m; r++) {
        for (int b = 0; b < n; b++) {
            double acc = 0.0, aacc = 0.0;
            const uint16_t *w = W + (size_t)r * k;
            const uint16_t *xx = x + (size_t)b * k;
            for (int c = 0; c < k; c++) {
                double p = (double)h2f_table[w[c]] * (double)h2f_table[xx[c]];
                acc += p;
                aacc += fabs(p);
            }
            out[(size_t)b * m + r] = acc;
            if (absout) absout[(size_t)b * m + r] = aacc;
        }
    }
}

/* ------------------------------------------------------------------ CPU baseline ("port" kind) */
/* What the reference does on a CPU: fake-dequant to fp16 W, then x @ W.T with fp32 accumulation
 * (lib/quantizer/quant_op.py:185-201 + matmul; lib/utils/kernel_decompress.py:64-88).  One call =
 * one linear at batch n.  Used only by bench.py's cpu_baseline leg and tests.                    */
static void gemv_f32acc(const uint16_t *W, const uint16_t *x, int m, int n, int k, float *out) {
    ensure_h2f();
#pragma omp parallel for schedule(static)
    for (int r = 0; r < m; r++)
        for (int b = 0; b < n; b++) {
            const uint16_t *w = W + (size_t)r * k;
            const uint16_t *xx = x + (size_t)b * k;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int c = 0;
            for (; c + 4 <= k; c += 4) {
                a0 += h2f_table[w[c]] * h2f_table[xx[c]];
                a1 += h2f_table[w[c + 1]] * h2f_table[xx[c + 1]];
                a2 += h2f_table[w[c + 2]] * h2f_table[xx[c + 2]];
                a3 += h2f_table[w[c + 3]] * h2f_table[xx[c + 3]];
            }
            for (; c < k; c++) a0 += h2f_table[w[c]] * h2f_table[xx[c]];
            out[(size_t)b * m + r] = (a0 + a1) + (a2 + a3);
        }
}

int qo_cpu_tcq_linear(const uint16_t *c1, const uint16_t *c2, const uint16_t *tlut, const uint16_t *x,
                      int m, int n, int k, int S, int KV1, int KV2, int split, uint16_t *Wscratch, float *out) {
    int rc = qo_tcq_dequant(c1, c2, tlut, m, k, S, KV1, KV2, split, Wscratch);
    if (rc) return rc;
    gemv_f32acc(Wscratch, x, m, n, k, out);
    return 0;
}

int qo_cpu_lut_tc_linear(const uint32_t *qweight, const uint16_t *lut, const uint16_t *x, int m, int n, int k,
                         int bits, int vec, uint16_t *Wscratch, float *out) {
    qo_lut_tc_dequant(qweight, lut, m, k, bits, vec, Wscratch);
    gemv_f32acc(Wscratch, x, m, n, k, out);
    return 0;
}

/* ------------------------------------------------------------------ CPU baseline, fused variant */
/* The same arithmetic without ever writing W: every 16x16 tile is decoded into registers / L1 and multiplied at once
 * ("plus a fused no-materialise variant").  fp32 accumulation per (batch, row), tile by tile along K.  */
static void tcq_fused_block(const uint16_t *trellis, const uint16_t *tlut, int m, int k, int S, int KV, const uint16_t *x,
                            int n, int ldx, int col0, float *out, int ldo, int row0, int accumulate) {
    const uint8_t *bytes = (const uint8_t *)trellis;
    int ntc = k / 16;
    ensure_h2f();
#pragma omp parallel for schedule(static)
    for (int tr = 0; tr < m / 16; tr++) {
        uint16_t st[128];
        float acc[8][16];
        for (int b = 0; b < n; b++)
            for (int r = 0; r < 16; r++) acc[b][r] = 0.f;
        for (int tc = 0; tc < ntc; tc++) {
            tcq_tile_states(bytes, k, KV, tr, tc, st);
            for (int r = 0; r < 16; r++)
                for (int c = 0; c < 16; c += 2) {
                    uint16_t w0, w1;
                    tcq_state_to_pair(st[tile_seq_pos(r, c) >> 1], tlut, S, &w0, &w1);
                    float f0 = h2f_table[w0], f1 = h2f_table[w1];
                    for (int b = 0; b < n; b++) {
                        const uint16_t *xx = x + (size_t)b * ldx + col0 + tc * 16 + c;
                        acc[b][r] += f0 * h2f_table[xx[0]] + f1 * h2f_table[xx[1]];
                    }
                }
        }
        for (int b = 0; b < n; b++)
            for (int r = 0; r < 16; r++) {
                float *dst = out + (size_t)b * ldo + row0 + tr * 16 + r;
                *dst = accumulate ? *dst + acc[b][r] : acc[b][r];
            }
    }
}

int qo_cpu_tcq_linear_fused(const uint16_t *c1, const uint16_t *c2, const uint16_t *tlut, const uint16_t *x, int m, int n,
                            int k, int S, int KV1, int KV2, int split, float *out) {
    if (n < 1 || n > 8) return -1;
    if (split == 0) {
        tcq_fused_block(c1, tlut, m, k, S, KV1, x, n, k, 0, out, m, 0, 0);
    } else if (split == 1) {
        tcq_fused_block(c1, tlut, m / 2, k, S, KV1, x, n, k, 0, out, m, 0, 0);
        tcq_fused_block(c2, tlut, m / 2, k, S, KV2, x, n, k, 0, out, m, m / 2, 0);
    } else if (split == 2) {
        tcq_fused_block(c1, tlut, m, k / 2, S, KV1, x, n, k, 0, out, m, 0, 0);
        tcq_fused_block(c2, tlut, m, k / 2, S, KV2, x, n, k, k / 2, out, m, 0, 1);
    } else {
        return -1;
    }
    return 0;
}

int qo_cpu_lut_tc_linear_fused(const uint32_t *qweight, const uint16_t *lut, const uint16_t *x, int m, int n, int k,
                               int bits, int vec, float *out) {
    const uint8_t *bytes = (const uint8_t *)qweight;
    int ncode = 8 / vec, gbits = ncode * bits;
    if (n < 1 || n > 8) return -1;
    ensure_h2f();
#pragma omp parallel for schedule(static)
    for (int tr = 0; tr < m / 16; tr++) {
        float acc[8][16];
        for (int b = 0; b < n; b++)
            for (int r = 0; r < 16; r++) acc[b][r] = 0.f;
        for (int tc = 0; tc < k / 16; tc++) {
            int sr = tr >> 1, msub = tr & 1, sc = tc >> 1, ksub = tc & 1;
            for (int lane = 0; lane < 32; lane++) {
                uint64_t base = ((((uint64_t)sr * (k / 32) + sc) * 32 + lane) * 4 + (uint64_t)(ksub * 2 + msub)) * gbits;
                for (int q = 0; q < ncode; q++) {
                    int32_t code = (int32_t)le_bits(bytes, base + (uint64_t)q * bits, bits);
                    int j = (vec == 1) ? (q >> 1) : q, e = (vec == 1) ? (q & 1) : 0;
                    int r = (lane >> 2) + 8 * (j & 1), c = 2 * (lane & 3) + 8 * (j >> 1) + e;
                    for (int v = 0; v < vec; v++) {
                        float w = h2f_table[lut[(size_t)code * vec + v]];
                        for (int b = 0; b < n; b++) acc[b][r] += w * h2f_table[x[(size_t)b * k + tc * 16 + c + v]];
                    }
                }
            }
        }
        for (int b = 0; b < n; b++)
            for (int r = 0; r < 16; r++) out[(size_t)b * m + tr * 16 + r] = acc[b][r];
    }
    return 0;
}

