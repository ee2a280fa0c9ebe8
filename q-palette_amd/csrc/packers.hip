// Host-side encoders of the packed weight formats (CPU code, no GPU involved): what a quantiser or a checkpoint
// converter calls once per layer.  Inverse of the decode the kernels perform; bit for bit the reference's
//   a1  pack_trellis + nibble permutation     lib/codebook/bitshift.py:296-329, lib/quantizer/tcq_quant.py:47-60
//   a4  pack_qweight (tensor-core order)       lib/quantizer/quant_op.py:89-162
//   a5  pack_for_sq_pack_kernel (SQ SIMT)      lib/quantizer/pack_op.py:288-335
//   a6  pack_qweight_vq_simt (VQ SIMT)         lib/quantizer/quant_op.py:15-78
// (there: torch bit-tensor gymnastics on the GPU + numba loops on the CPU; here: one pass of plain C++).
#include <stdint.h>
#include <string.h>

#include "qpal.h"

namespace {

// OR `nbits` (<= 32) bits of `v` into the little-endian bit string `p` at bit position `pos` (p pre-zeroed)
inline void put_le_bits(uint8_t *p, uint64_t pos, int nbits, uint64_t v) {
    v <<= (pos & 7);
    uint64_t byte = pos >> 3;
    const int nbytes = (int)(((pos & 7) + nbits + 7) >> 3);
    for (int i = 0; i < nbytes; i++) p[byte + i] |= (uint8_t)(v >> (8 * i));
}

// 16x16 tile in mma fragment order (lib/algo/ldlq.py:10-13): lane = 4 (r % 8) + (c % 8) / 2, j = 2 (c / 8) + r / 8
inline void tile_rc(int lane, int j, int e, int &r, int &c) {
    r = (lane >> 2) + 8 * (j & 1);
    c = 2 * (lane & 3) + 8 * (j >> 1) + e;
}

// one tile's 128 states -> its 16 KV bytes inside the supertile-interleaved stream
int pack_tile(uint8_t *bytes, const uint16_t *st, int k, int KV, int tr, int tc) {
    const int sr = tr >> 1, msub = tr & 1, sc = tc >> 1, ksub = tc & 1;
    const uint32_t keep = (1u << (16 - KV)) - 1u;
    for (int lane = 0; lane < 32; lane++) {
        // the lane's 4 KV stream bits, MSB first: the KV new bits of its 4 states
        uint64_t v = 0;
        for (int j = 0; j < 4; j++) {
            const int t = 4 * lane + j;
            const uint32_t s = st[t], nxt = st[(t + 1) & 127];
            if ((s & keep) != (nxt >> KV)) return QPAL_E_PARAM;  // not a tail-biting trellis walk
            v = (v << KV) | (s >> (16 - KV));
        }
        const uint64_t base = ((((uint64_t)sr * (k / 32) + sc) * 32 + lane) * 16 + (uint64_t)(ksub * 2 + msub) * 4) * KV;
        put_le_bits(bytes, base, 4 * KV > 32 ? 32 : 4 * KV, v & 0xffffffffu);
        if (4 * KV > 32) put_le_bits(bytes, base + 32, 4 * KV - 32, v >> 32);
    }
    return QPAL_OK;
}

int tcq_args(const void *dst, const void *src, int m, int k, int KV) {
    if (!dst || !src) return QPAL_E_NULL;
    if (m <= 0 || k <= 0 || m % 32 || k % 32) return QPAL_E_SHAPE;
    if (KV < 2 || KV > 10) return QPAL_E_PARAM;
    return QPAL_OK;
}

}  // namespace

extern "C" {

// states: uint16 [m/16 * k/16][128], tile-major (tile = tr * (k/16) + tc), state t = 4 * lane + j of the tile's
// tail-biting sequence.  dst: int16 [(m/16)(k/16)][8 KV] (zeroed here).
int qpal_pack_tcq_states(void *dst, const uint16_t *states, int m, int k, int KV) {
    int rc = tcq_args(dst, states, m, k, KV);
    if (rc) return rc;
    uint8_t *bytes = static_cast<uint8_t *>(dst);
    const int ntc = k / 16;
    memset(bytes, 0, (size_t)(m / 16) * ntc * 16 * KV);
    for (int tr = 0; tr < m / 16; tr++)
        for (int tc = 0; tc < ntc; tc++)
            if ((rc = pack_tile(bytes, states + ((size_t)tr * ntc + tc) * 128, k, KV, tr, tc))) return rc;
    return QPAL_OK;
}

// qidxs: int32 [m][k/2], the quantiser's layout: qidxs[16 tr + t / 8][8 tc + t % 8] = state t of tile (tr, tc)
// (lib/algo/ldlq.py:107-110, lib/quantizer/tcq_quant.py:47-50)
int qpal_pack_tcq(void *dst, const int32_t *qidxs, int m, int k, int KV) {
    int rc = tcq_args(dst, qidxs, m, k, KV);
    if (rc) return rc;
    uint8_t *bytes = static_cast<uint8_t *>(dst);
    const int ntc = k / 16;
    memset(bytes, 0, (size_t)(m / 16) * ntc * 16 * KV);
    uint16_t st[128];
    for (int tr = 0; tr < m / 16; tr++) {
        for (int tc = 0; tc < ntc; tc++) {
            for (int t = 0; t < 128; t++) {
                const int32_t s = qidxs[(size_t)(16 * tr + t / 8) * (k / 2) + 8 * tc + t % 8];
                if (s < 0 || s > 0xffff) return QPAL_E_PARAM;
                st[t] = (uint16_t)s;
            }
            if ((rc = pack_tile(bytes, st, k, KV, tr, tc))) return rc;
        }
    }
    return QPAL_OK;
}

// idx: int32 [m][k/vec] codebook indices (< 2^bits); dst: int32 [m][bits k / 32 / vec], tensor-core order
int qpal_pack_lut_tc(void *dst, const int32_t *idx, int m, int k, int bits, int vec) {
    if (!dst || !idx) return QPAL_E_NULL;
    if (m <= 0 || k <= 0 || m % 32 || k % 32) return QPAL_E_SHAPE;
    if (!((vec == 1 && bits >= 2 && bits <= 8) || (vec == 2 && bits >= 2 && bits <= 12)) || ((long)bits * k) % (32 * vec))
        return QPAL_E_PARAM;
    uint8_t *bytes = static_cast<uint8_t *>(dst);
    memset(bytes, 0, (size_t)m * k / vec * bits / 8);
    const int ncode = 8 / vec, gbits = ncode * bits;
    for (int tr = 0; tr < m / 16; tr++) {
        for (int tc = 0; tc < k / 16; tc++) {
            const int sr = tr >> 1, msub = tr & 1, sc = tc >> 1, ksub = tc & 1;
            for (int lane = 0; lane < 32; lane++) {
                const uint64_t base =
                    ((((uint64_t)sr * (k / 32) + sc) * 32 + lane) * 4 + (uint64_t)(ksub * 2 + msub)) * gbits;
                for (int q = 0; q < ncode; q++) {
                    int r, c;
                    tile_rc(lane, vec == 1 ? q >> 1 : q, vec == 1 ? q & 1 : 0, r, c);
                    const int32_t code = idx[((size_t)tr * 16 + r) * (k / vec) + ((size_t)tc * 16 + c) / vec];
                    if (code < 0 || code >= (1 << bits)) return QPAL_E_PARAM;
                    put_le_bits(bytes, base + (uint64_t)q * bits, bits, (uint32_t)code);
                }
            }
        }
    }
    return QPAL_OK;
}

// idx: int32 [m][k/vec]; dst: uint32 [m][bits k / 32 / vec], SIMT order (vec 1: SQ, vec 2 / 4: VQ).  A row is cut into
// blocks of 32 lanes; a lane owns 4 vec groups of 8 consecutive weights; its 32 codes go LSB-first into `bits` words
// stored lane-interleaved (word j of lane t at block base + t + W j, W = lanes of the block).
int qpal_pack_lut_simt(void *dst, const int32_t *idx, int m, int k, int bits, int vec) {
    if (!dst || !idx) return QPAL_E_NULL;
    if (m <= 0 || k <= 0 || k % (32 * vec)) return QPAL_E_SHAPE;
    if (!((vec == 1 && bits >= 2 && bits <= 8) || (vec == 2 && bits >= 3 && bits <= 12) ||
          (vec == 4 && bits >= 6 && bits <= 12)))
        return QPAL_E_PARAM;
    const int ngrp = 4 * vec, B = 256 * ngrp, per_lane = 32 * vec;
    const size_t row_words = (size_t)k * bits / 32 / vec;
    uint32_t *out = static_cast<uint32_t *>(dst);
    memset(out, 0, sizeof(uint32_t) * row_words * m);
    for (int r = 0; r < m; r++) {
        uint32_t *row = out + (size_t)r * row_words;
        const int nblk = (k + B - 1) / B;
        for (int blk = 0; blk < nblk; blk++) {
            const int W = blk == k / B ? (k % B) / per_lane : 32;
            for (int t = 0; t < W; t++) {
                uint32_t words[16] = {0};
                for (int c = 0; c < 32; c++) {
                    const int e0 = c * vec, g = e0 / 8, off = e0 % 8;
                    const size_t elem = (size_t)blk * B + (size_t)g * 8 * W + 8 * t + off;
                    const int32_t code = idx[(size_t)r * (k / vec) + elem / vec];
                    if (code < 0 || code >= (1 << bits)) return QPAL_E_PARAM;
                    put_le_bits(reinterpret_cast<uint8_t *>(words), (uint64_t)c * bits, bits, (uint32_t)code);
                }
                for (int j = 0; j < bits; j++) row[(size_t)blk * bits * 32 + t + (size_t)W * j] = words[j];
            }
        }
    }
    return QPAL_OK;
}

}  // extern "C"
