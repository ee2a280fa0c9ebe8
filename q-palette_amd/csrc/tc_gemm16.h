// Fused decode + skinny GEMM for batches 9..128, second fragment mapping (round 5): the MFMA's 16 rows are 16 REAL rows of W.
//
// tc_gemm.h inherited the GEMV's mapping: a wave-lane holds the weights of two reference lanes = one tile row x one column HALF
// (4 of every 8 columns), so an MFMA's 16 rows were (8 tile rows x 2 column halves) and its 16 columns (8 batch rows x 2 halves):
// only the products whose halves agree are valid.  The matrix pipe did twice the needed work, the accumulators took 16 VGPRs per 8
// batch rows (128 at batch 64: 256-VGPR kernels, two waves per SIMD), and a pass over the weights could carry 64 batch rows at most
// (65 rows = two passes: the 64 -> 65 cliff).
//
// Here lanes p and p ^ 1 of a DPP row — the two column halves of tile row p >> 1 — EXCHANGE half of their decoded pairs: the even
// lane keeps its pairs of tile row r (jl = 0) and takes the odd lane's, the odd lane keeps its pairs of row r + 8 (jl = 1) and takes
// the even lane's.  Each then holds 8 CONSECUTIVE columns of one row: an A fragment of v_mfma_f32_16x16x32_f16 whose rows are the
// 16 rows (r, r + 8 for r = 0..7) of a 16-row block and whose K is (4 supertiles x 8 columns); the B fragment is 8 consecutive
// columns of one batch row, 16 batch rows per MFMA.  Per step and 16 batch rows: 8 MFMAs (was 16), 8 accumulator VGPRs (was 32);
// the exchange costs one v_cndmask_b32_dpp per decoded pair (32 per step, full-rate VALU).  One pass carries 128 batch rows.
//
// Everything else is tc_gemm.h's: the 8 waves of a workgroup own 8 supertile rows and walk one K range in lockstep, the step's
// [n][128] tile of x is staged once per workgroup (global -> registers -> LDS, two steps ahead where four slots fit), split-K over
// workgroups with atomics only where a layer has too few rows.  Same arithmetic per (row, batch) up to the K split and the order
// of the eight column blocks inside a step.  Replaces, for bs > 8, the reference's decode-to-HBM + cuBLAS path
// (lib/linear/tcq_linear.py:75-84, vq_linear.py:60-66).
#pragma once
#include "tc_gemm.h"

namespace qpal {

// waves = supertile rows per workgroup of this kernel (experiments: -DQPAL_G16_WAVES=16 -DQPAL_G16_PIPE=0 — 16 waves need <= 128 VGPRs)
#ifndef QPAL_G16_WAVES
#define QPAL_G16_WAVES 8
#endif
constexpr int kG16Waves = QPAL_G16_WAVES;
constexpr int kG16Group = 4096;  // bytes of one step's x tile per 16 batch rows: [sup 4][ksub, jh 4][slot 16][16 B]

// x tile layout.  The 16-byte piece (batch row b, columns 8 q .. 8 q + 7 of the step; q = 4 sup + 2 ksub + jh) lives at
//   (b >> 4) * 4096 + q * 256 + ((b + q) & 15) * 16:
// the 16 batch rows' pieces of one q form a 256-byte block — the lanes (sup, c) of a 16-lane group of a ds_read_b128 cover 64
// banks exactly once — and the rotation by q spreads the 8 lanes of a ds_write_b128 group (one batch row, 8 values of q) over the
// banks as well (without it they all hit one 16-byte window: 8-way conflicts on every staging store).
// QPAL_G16_XROT = 1 (round 5, second version): the rotation is g(q) = (q & 3) + 4 (q >> 3) instead of q.  A ds_read_b128 is served in
// four groups of 16 lanes that are NOT 16 consecutive lanes (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...):
// a group mixes batch rows {0-3, 12-15} of column block q with rows {4-11} of block q + 4, so the two blocks must share one rotation
// or four of the sixteen 16-byte windows collide (measured: SQ_LDS_BANK_CONFLICT 54 % of the LDS cycles of a batch-64 step).  The
// staging stores keep their eight lanes of one batch row on eight different windows by taking the column blocks in the order
// 0-3, 8-11 | 4-7, 12-15 (g16_chunk_q).
#ifndef QPAL_G16_XROT
#define QPAL_G16_XROT 1
#endif
__device__ __forceinline__ int g16_rot(int q) { return QPAL_G16_XROT ? (q & 3) + 4 * (q >> 3) : q; }
__device__ __forceinline__ int g16_chunk_q(int j) { return QPAL_G16_XROT ? ((j & 3) | ((j & 4) << 1) | ((j & 8) >> 1)) : j; }  // staging lane j of a batch row -> its column block
__device__ __forceinline__ int g16_x_off(int b, int q) { return (b >> 4) * kG16Group + q * 256 + (((b + g16_rot(q)) & 15) << 4); }

// The four decoded pairs an A fragment is made of — tile group G = ksub * 2 + msub, column block JH.  Pair index I = jl + 2 jh + 4 isB
// (tc_kernels.h gemv_step); a lane's pairs (jl, jh) are columns 4 u + 0..3 (isB = 0: +0, 1; isB = 1: +2, 3) of tile row (p >> 1) + 8 jl.
struct G16Pairs {
    uint32_t p0a, p0b, p1a, p1b;  // p0: tile row r (jl = 0), p1: tile row r + 8 (jl = 1)
};
template <class Codec, int G, int JH>
__device__ __forceinline__ G16Pairs g16_pairs(const uint32_t *lut, uint32_t laneoff, const uint32_t (&w)[Codec::NW]) {
    uint32_t nh = 0u;
    if constexpr (Codec::kNeedsNext) nh = row16_next(Codec::template head<G>(w));  // (one per G: CSE)
    return G16Pairs{Codec::template pair<G, 0 + 2 * JH>(lut, laneoff, w, nh), Codec::template pair<G, 0 + 2 * JH + 4>(lut, laneoff, w, nh),
                    Codec::template pair<G, 1 + 2 * JH>(lut, laneoff, w, nh), Codec::template pair<G, 1 + 2 * JH + 4>(lut, laneoff, w, nh)};
}
// The exchange.  Even lane (u = 0): row r, columns 0..3 own (p0), 4..7 the partner's p0; odd lane: row r + 8, columns 0..3 the
// partner's p1, 4..7 own (p1).  "The partner's" = lane ^ 1 = DPP quad_perm [1, 0, 3, 2], folded into the select: v_cndmask_b32_dpp
// computes vcc ? src1 : dpp(src0) — ONE full-rate instruction per pair.  (Written as a move + a select the compiler emitted both: 64
// more vector instructions per step.)  Hazards the assembler does not see inside an asm: a DPP operand written by the VALU
// instruction right in front needs two wait states (s_nop 1); SALU writes of VCC are interlocked.
__device__ __forceinline__ u32x4 g16_xchg(const G16Pairs &q) {
    if constexpr (QPAL_GEMM_KO & 16) return u32x4{q.p0a, q.p0b, q.p1a, q.p1b};  // (timing experiment: no exchange)
    uint32_t x, y, z, w_;
    asm("s_nop 1\n\t"
        "s_mov_b64 vcc, %[even]\n\t"
        "v_cndmask_b32_dpp %[x], %[p1a], %[p0a], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %[y], %[p1b], %[p0b], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_not_b64 vcc, vcc\n\t"
        "v_cndmask_b32_dpp %[z], %[p0a], %[p1a], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %[w], %[p0b], %[p1b], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        : [x] "=&v"(x), [y] "=&v"(y), [z] "=&v"(z), [w] "=&v"(w_)
        : [p0a] "v"(q.p0a), [p0b] "v"(q.p0b), [p1a] "v"(q.p1a), [p1b] "v"(q.p1b), [even] "s"(0x5555555555555555ull)
        : "vcc", "scc");
    return u32x4{x, y, z, w_};
}

// two fragments' exchanges under one pair of lane masks
__device__ __forceinline__ void g16_xchg2(const G16Pairs &q, const G16Pairs &r, u32x4 &a, u32x4 &b) {
    if constexpr (QPAL_GEMM_KO & 16) {
        a = u32x4{q.p0a, q.p0b, q.p1a, q.p1b};
        b = u32x4{r.p0a, r.p0b, r.p1a, r.p1b};
        return;
    }
    uint32_t x, y, z, w_, x2, y2, z2, w2;
    asm("s_nop 1\n\t"
        "s_mov_b64 vcc, %[even]\n\t"
        "v_cndmask_b32_dpp %[x], %[p1a], %[p0a], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %[y], %[p1b], %[p0b], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %[x2], %[r1a], %[r0a], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %[y2], %[r1b], %[r0b], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_not_b64 vcc, vcc\n\t"
        "v_cndmask_b32_dpp %[z], %[p0a], %[p1a], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %[w], %[p0b], %[p1b], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %[z2], %[r0a], %[r1a], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %[w2], %[r0b], %[r1b], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        : [x] "=&v"(x), [y] "=&v"(y), [z] "=&v"(z), [w] "=&v"(w_), [x2] "=&v"(x2), [y2] "=&v"(y2), [z2] "=&v"(z2), [w2] "=&v"(w2)
        : [p0a] "v"(q.p0a), [p0b] "v"(q.p0b), [p1a] "v"(q.p1a), [p1b] "v"(q.p1b), [r0a] "v"(r.p0a), [r0b] "v"(r.p0b), [r1a] "v"(r.p1a),
          [r1b] "v"(r.p1b), [even] "s"(0x5555555555555555ull)
        : "vcc", "scc");
    a = u32x4{x, y, z, w_};
    b = u32x4{x2, y2, z2, w2};
}

// (QPAL_GEMM_KO, tc_gemm.h: timing experiments — bit 1: no MFMAs, 2: no B-fragment reads, 4: no x staging in the loop, 8: no barrier
// in the loop, 16: no exchange, 64: no output stores / atomics, 128: no codebook image build, 256: no steps at all)
__device__ __forceinline__ u32x4 g16_xread(const unsigned char *at, uint32_t laneoff, int lane) {
    if constexpr (QPAL_GEMM_KO & 2) return u32x4{laneoff, (uint32_t)lane, laneoff, (uint32_t)lane};
    else return *reinterpret_cast<const u32x4 *>(at);
}
__device__ __forceinline__ void g16_mfma(float4_t &acc, const u32x4 &a, const u32x4 &b) {
    if constexpr (QPAL_GEMM_KO & 1) {
        const uint32_t f = (a[0] ^ a[1] ^ a[2] ^ a[3]) ^ (b[0] ^ b[1] ^ b[2] ^ b[3]);
        acc[0] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, acc[0]) ^ f);
    } else {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), acc, 0, 0, 0);
    }
}

// One step: 8 A fragments t = 2 kj + msub (kj = 2 ksub + jh), each feeding NG MFMAs — acc[grp][msub] accumulates the 16 rows
// 16 msub + (r, r + 8) x 16 batch rows; per kj the B fragments of the NG batch groups (8 consecutive columns of batch row 16 grp + c).
// Software-pipelined (QPAL_G16_PIPE, default): a wave issues in order, and left to the compiler a fragment is decode -> wait for the
// gathers -> exchange -> NG MFMAs; here the pairs of fragment t + 1 are decoded and gathered in front of the MFMAs of fragment t and
// exchanged behind them (the MFMAs cover the gather latency), and the B fragments of kj + 1 are read behind the last use of kj's.
#ifndef QPAL_G16_PIPE
#define QPAL_G16_PIPE 2
#endif
template <class Codec, int T>
__device__ __forceinline__ G16Pairs g16_pairs_t(const uint32_t *lut, uint32_t laneoff, const uint32_t (&w)[Codec::NW]) {
    constexpr int kj = T >> 1, msub = T & 1, ksub = kj >> 1, jh = kj & 1;
    return g16_pairs<Codec, ksub * 2 + msub, jh>(lut, laneoff, w);
}
template <class Codec, int NG>
__device__ __forceinline__ void g16_step(const uint32_t *lut, uint32_t laneoff, const uint32_t (&w)[Codec::NW], const unsigned char *xt, int lane,
                                         float4_t (&acc)[NG][2]) {
    const int sup = lane >> 4, c = lane & 15;
    auto xoff = [&](int kj) { return (4 * sup + kj) * 256 + (((c + g16_rot(4 * sup + kj)) & 15) << 4); };
    if constexpr (QPAL_G16_PIPE == 2) {
        // fragment PAIRS (msub 0, 1 of one column block): one VCC flip and one DPP wait per two exchanges, 2 NG MFMAs in a row
        u32x4 xb[NG], xn[NG];
        static_for<0, NG>([&](auto gc) { xb[decltype(gc)::value] = g16_xread(xt + decltype(gc)::value * kG16Group + xoff(0), laneoff, lane); });
        u32x4 a0, a1;
        g16_xchg2(g16_pairs_t<Codec, 0>(lut, laneoff, w), g16_pairs_t<Codec, 1>(lut, laneoff, w), a0, a1);
        static_for<0, 4>([&](auto kc) {
            constexpr int kj = decltype(kc)::value;
            G16Pairs n0{}, n1{};
            if constexpr (kj < 3) {
                n0 = g16_pairs_t<Codec, 2 * kj + 2>(lut, laneoff, w);
                n1 = g16_pairs_t<Codec, 2 * kj + 3>(lut, laneoff, w);
                static_for<0, NG>([&](auto gc) { xn[decltype(gc)::value] = g16_xread(xt + decltype(gc)::value * kG16Group + xoff(kj + 1), laneoff, lane); });
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, NG>([&](auto gc) { g16_mfma(acc[decltype(gc)::value][0], a0, xb[decltype(gc)::value]); });
            static_for<0, NG>([&](auto gc) { g16_mfma(acc[decltype(gc)::value][1], a1, xb[decltype(gc)::value]); });
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (kj < 3) {
                g16_xchg2(n0, n1, a0, a1);
                static_for<0, NG>([&](auto gc) { xb[decltype(gc)::value] = xn[decltype(gc)::value]; });
            }
        });
    } else if constexpr (QPAL_G16_PIPE != 0) {
        u32x4 xb[NG], xn[NG];
        static_for<0, NG>([&](auto gc) { xb[decltype(gc)::value] = g16_xread(xt + decltype(gc)::value * kG16Group + xoff(0), laneoff, lane); });
        u32x4 a = g16_xchg(g16_pairs_t<Codec, 0>(lut, laneoff, w));
        static_for<0, 8>([&](auto tc) {
            constexpr int t = decltype(tc)::value, kj = t >> 1, msub = t & 1;
            G16Pairs nx{};
            if constexpr (t < 7) nx = g16_pairs_t<Codec, t + 1>(lut, laneoff, w);
            if constexpr (msub == 1 && kj < 3)  // the next column block's B fragments, in flight behind this fragment's MFMAs
                static_for<0, NG>([&](auto gc) { xn[decltype(gc)::value] = g16_xread(xt + decltype(gc)::value * kG16Group + xoff(kj + 1), laneoff, lane); });
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, NG>([&](auto gc) {
                constexpr int grp = decltype(gc)::value;
                g16_mfma(acc[grp][msub], a, xb[grp]);
            });
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (t < 7) a = g16_xchg(nx);
            if constexpr (msub == 1 && kj < 3) static_for<0, NG>([&](auto gc) { xb[decltype(gc)::value] = xn[decltype(gc)::value]; });
        });
    } else {
        static_for<0, 4>([&](auto kjc) {
            constexpr int kj = decltype(kjc)::value, ksub = kj >> 1, jh = kj & 1;
            u32x4 xb[NG];
            static_for<0, NG>([&](auto gc) { xb[decltype(gc)::value] = g16_xread(xt + decltype(gc)::value * kG16Group + xoff(kj), laneoff, lane); });
            static_for<0, 2>([&](auto mc) {
                constexpr int msub = decltype(mc)::value;
                const u32x4 a = g16_xchg(g16_pairs<Codec, ksub * 2 + msub, jh>(lut, laneoff, w));
                static_for<0, NG>([&](auto gc) {
                    constexpr int grp = decltype(gc)::value;
                    g16_mfma(acc[grp][msub], a, xb[grp]);
                });
            });
        });
    }
}

template <class C1, class C2, int NG>
__global__ __launch_bounds__(64 * kG16Waves) void tc_gemm16_kernel(const int eie, const TcMultiParams mp) {
    constexpr bool TWO = !std::is_void_v<C2>;
    using CB = std::conditional_t<TWO, C2, C1>;
    constexpr int W = kG16Waves, NT = 64 * W;
    constexpr int XBUF = NG * kG16Group;
    constexpr int NCH = NG * 16 * 16;                   // 16-byte chunks of one step's x tile
    constexpr int CPT = (NCH + NT - 1) / NT;            // chunks per thread
    constexpr int NWMAX = C1::NW > CB::NW ? C1::NW : CB::NW;
    constexpr int NSLOT = (QPAL_GEMM_SLOTS == 4 && C1::LDS_DWORDS * 4 + 4 * XBUF <= 156 * 1024) ? 4 : 2;
    constexpr int AHEAD = NSLOT / 2;
    constexpr int XT = NSLOT * XBUF >= W * 2048 ? NSLOT * XBUF : W * 2048;  // the x buffers double as the epilogue's per-wave [16][32] fp32 scratch
    // ONE block, the codebook image FIRST: at LDS address 0 a gather address is one v_and_or_b32 (hash bits | copy of this lane); as two
    // arrays the compiler put the tiles first and every gather paid a v_and_b32 + v_add_u32 (32 more vector instructions per step)
    __shared__ __attribute__((aligned(16))) unsigned char smem[C1::LDS_DWORDS * 4 + XT];
    uint32_t *const lut = reinterpret_cast<uint32_t *>(smem);
    unsigned char *const xt = smem + C1::LDS_DWORDS * 4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t laneoff = (uint32_t)(lane & (C1::C - 1)) << 2;
    const void *cur_tab = nullptr;
    int ie[kMaxJobs];
#pragma unroll
    for (int i = 0; i < kMaxJobs; i++) ie[i] = mp.item_end[i];
    const int total_items = ie[kMaxJobs - 1];
    int jf = 0;
    if (eie != 0) {
        const int b = (int)blockIdx.x, e0 = eie & 0x3ff, e1 = (eie >> 10) & 0x3ff, e2 = (eie >> 20) & 0x3ff;
        if (b >= e0) {
            jf = 1;
            if (b >= e1) {
                jf = 2;
                if (b >= e2) jf = 3;
            }
        }
    }
    int cur_j = jf;
    TcParams p = mp.job[jf];

    for (int gitem = blockIdx.x; gitem < total_items; gitem += gridDim.x) {
        int j = 0, item_begin = 0;
#pragma unroll
        for (int i = 0; i < kMaxJobs - 1; i++) {
            if (gitem >= ie[i]) {
                j = i + 1;
                item_begin = ie[i];
            }
        }
        if (j != cur_j) {
            p = mp.job[j];
            cur_j = j;
        }
        const int item = gitem - item_begin;
        const int lsk = __builtin_ctz((unsigned)p.sk);  // (a power of two: plan_gemm)
        const int rg = item >> lsk, ks = item & (p.sk - 1);
        const int T = p.st1 + p.st2;
        const int g0 = (T * ks) >> lsk, g1 = (T * (ks + 1)) >> lsk;
        const int sr = rg * W + wave;
        const bool live = sr < p.nrows;
        const int srow = live ? sr : 0;
        const StreamView sv1{p.c1 + (long)srow * p.nsc1 * 16 * C1::NW, p.nsc1, 0};
        const StreamView sv2{TWO ? p.c2 + (long)srow * p.nsc2 * 16 * CB::NW : p.c1, TWO ? p.nsc2 : p.nsc1, p.col2};

        uint32_t wcur[NWMAX], wnext[NWMAX];
        u32x4 xr[CPT];
        // buffer-addressed loads (tc_gemm.h): one 32-bit add per load and step, the descriptors' range checks do the clamping
        const buf_rsrc_t rs_x = gemm_rsrc(p.x, p.n * p.k * 2);
        const buf_rsrc_t rs_w1 = gemm_rsrc(sv1.base, sv1.nsc * 64 * C1::NW);
        const buf_rsrc_t rs_w2 = gemm_rsrc(sv2.base, sv2.nsc * 64 * CB::NW);
        const uint32_t wl1 = gemm_w_lane_off<C1::NW>(lane), wl2 = gemm_w_lane_off<CB::NW>(lane);
        uint32_t xvo[CPT];  // byte offset of this thread's chunk r inside x, without the step's column base
#pragma unroll
        for (int r = 0; r < CPT; r++) {
            const int id = tid + r * NT, b = id >> 4;
            xvo[r] = (id < NCH && b < p.n) ? (uint32_t)(b * p.k + 8 * g16_chunk_q(id & 15)) * 2u : kBufDead;
        }
        auto load_w = [&](int g, uint32_t(&dst)[NWMAX]) {
            const GemmStep st = gemm_where(p, g);
            if (TWO && st.on2) buf_load_words_nt<CB::NW>(rs_w2, wl2 + (uint32_t)st.s * gemm_w_step_bytes<CB::NW>(), reinterpret_cast<uint32_t(&)[CB::NW]>(dst));
            else buf_load_words_nt<C1::NW>(rs_w1, wl1 + (uint32_t)st.s * gemm_w_step_bytes<C1::NW>(), reinterpret_cast<uint32_t(&)[C1::NW]>(dst));
        };
        // chunk id -> (batch row b = id >> 4, 16-byte piece q = id & 15 of the step's 128 columns); unconditional loads from clamped
        // addresses, dead chunks zeroed when stored (tc_gemm.h: a load under a condition makes the compiler's wait counts conservative)
        auto load_x = [&](int g) {  // (columns past the end of the stream are zeroed when the chunk is stored)
            const GemmStep st = gemm_where(p, g);
#pragma unroll
            for (int r = 0; r < CPT; r++) xr[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xvo[r] + (uint32_t)st.col_base * 2u, 0, 0);
        };
        auto store_x = [&](unsigned char *buf, int g) {
            const GemmStep st = gemm_where(p, g);
#pragma unroll
            for (int r = 0; r < CPT; r++) {
                const int id = tid + r * NT;
                const int b = id >> 4, q = g16_chunk_q(id & 15);
                const bool ok = id < NCH && b < p.n && st.col_base + 8 * q < st.col_end;
                if (id < NCH) *reinterpret_cast<u32x4 *>(buf + g16_x_off(b, q)) = ok ? xr[r] : u32x4{0u, 0u, 0u, 0u};
            }
        };

        load_w(g0, wcur);
        load_x(g0);
        [[maybe_unused]] u32x4 xr1[CPT];  // (4 slots: the second step's tile, requested together with the first)
        if constexpr (AHEAD == 2) {
#pragma unroll
            for (int r = 0; r < CPT; r++) xr1[r] = xr[r];
            load_x(g0 + 1 < g1 ? g0 + 1 : g0);
        }
        if (gitem == (int)blockIdx.x && mp.zero_chunks > 0) {  // pre-zero a buffer for a later split-K launch on this stream
            for (int i = blockIdx.x * NT + tid; i < mp.zero_chunks; i += gridDim.x * NT) mp.zero[i] = u32x4{0u, 0u, 0u, 0u};
        }
        if (p.tab != cur_tab && (!(QPAL_GEMM_KO & 128) || p.n == 12345)) {  // workgroup-uniform
            if constexpr (C1::LDS_DWORDS * 4 > 64 * 1024) C1::template build<8>(lut, p.tab, tid, NT);
            else C1::template build<QPAL_GEMM_BUILD_U>(lut, p.tab, tid, NT);
            cur_tab = p.tab;
        }
        if constexpr (AHEAD == 2) {
            store_x(xt + XBUF, g0 + 1 < g1 ? g0 + 1 : g0);
#pragma unroll
            for (int r = 0; r < CPT; r++) xr[r] = xr1[r];
        }
        store_x(xt, g0);
        float4_t acc[NG][2];
        static_for<0, NG>([&](auto gc) {
            acc[decltype(gc)::value][0] = float4_t{0.f, 0.f, 0.f, 0.f};
            acc[decltype(gc)::value][1] = float4_t{0.f, 0.f, 0.f, 0.f};
        });
        __syncthreads();

        // the steps of ONE stream (one codec: no branch inside the loop body)
        auto run = [&](auto codec_c, int ga, int gb) {
            using CC = typename decltype(codec_c)::type;
            constexpr bool SECOND = TWO && std::is_same_v<CC, CB> && !std::is_same_v<C1, CB>;
            auto one_step = [&](int g, uint32_t(&wc)[NWMAX], uint32_t(&wn)[NWMAX]) {
                const int xg = g + AHEAD < g1 ? g + AHEAD : g1 - 1;
                if constexpr (!(QPAL_GEMM_KO & 4)) load_x(xg);
                const int gn = g + 1 < gb ? g + 1 : g;
                if constexpr (SECOND) buf_load_words_nt<CC::NW>(rs_w2, wl2 + (uint32_t)(gn - p.st1) * gemm_w_step_bytes<CC::NW>(), reinterpret_cast<uint32_t(&)[CC::NW]>(wn));
                else buf_load_words_nt<CC::NW>(rs_w1, wl1 + (uint32_t)gn * gemm_w_step_bytes<CC::NW>(), reinterpret_cast<uint32_t(&)[CC::NW]>(wn));
                __builtin_amdgcn_sched_barrier(0);
                const int i = g - g0;
                if (live)  // (wave-uniform: a row past the end of the layer only keeps the staging and the barriers company)
                    g16_step<CC, NG>(lut, laneoff, reinterpret_cast<uint32_t(&)[CC::NW]>(wc), xt + (i & (NSLOT - 1)) * XBUF, lane, acc);
                if constexpr (!(QPAL_GEMM_KO & 4)) store_x(xt + ((i + AHEAD) & (NSLOT - 1)) * XBUF, xg);  // slot of step i + AHEAD: last read before the latest barrier
                if constexpr (!(QPAL_GEMM_KO & 8)) {
                    if (AHEAD == 1 || (i & 1)) __syncthreads();
                }
            };
            int g = ga;  // two steps per trip: the prefetched weights become the current ones by name (tc_gemm.h)
            for (; g + 1 < gb; g += 2) {
                one_step(g, wcur, wnext);
                one_step(g + 1, wnext, wcur);
            }
            if (g < gb) one_step(g, wcur, wnext);
        };
        const int mid = g1 < p.st1 ? g1 : (g0 > p.st1 ? g0 : p.st1);
        if constexpr (!(QPAL_GEMM_KO & 256)) {
            if (g0 < mid) run(std::type_identity<C1>{}, g0, mid);
            if constexpr (TWO) {
                if (mid < g1) {
                    if (g0 < mid) load_w(mid, wcur);  // the switch to stream 2 inside an item: its first step, requested here
                    run(std::type_identity<CB>{}, mid, g1);
                }
            }
        }

        if constexpr (AHEAD == 2) __syncthreads();  // (an odd number of steps ends without one: the scratch below overlays the tiles)
        // ---- epilogue.  acc[grp][msub]: lane (q4 = lane >> 4, c = lane & 15) holds D[i = 4 q4 + e][c], e = 0..3: batch row 16 grp + c,
        // MFMA row i = A lane p = 2 r + u -> row 16 msub + (i >> 1) + 8 (i & 1) of the supertile: e = 0, 1, 2, 3 are rows
        // 2 q4, 2 q4 + 8, 2 q4 + 1, 2 q4 + 9 (+ 16 msub).  One group of 16 batch rows at a time goes through a per-wave [16][32] fp32
        // scratch and leaves as 128-byte runs.
        float *scr = reinterpret_cast<float *>(xt) + wave * 512;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));  // (the epilogue's addresses are computed HERE, not held through the steps)
        const int q4 = lane_e >> 4, c = lane_e & 15, r32 = lane_e & 31;
        float osc = p.oscale;
        if (p.wscale && live) osc *= (float)__builtin_bit_cast(_Float16, p.wscale[sr * 32 + r32]);
        auto epilogue = [&](auto mode_c) {
            constexpr int mode = decltype(mode_c)::value;  // 0: store, 1: out += (residual add), 2: split-K atomics
            static_for<0, NG>([&](auto gc) {
                constexpr int grp = decltype(gc)::value;
                static_for<0, 2>([&](auto mc) {
                    constexpr int msub = decltype(mc)::value;
                    const float4_t d = acc[grp][msub];
                    float *dst = scr + c * 32 + 16 * msub + 2 * q4;
                    *reinterpret_cast<float2 *>(dst) = float2{d[0], d[2]};      // rows 2 q4, 2 q4 + 1
                    *reinterpret_cast<float2 *>(dst + 8) = float2{d[1], d[3]};  // rows 2 q4 + 8, 2 q4 + 9
                });
                // (wave-private scratch: LDS operations of one wave complete in order)
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int bl = 2 * i + (lane_e >> 5), b = 16 * grp + bl;
                    const float v = scr[bl * 32 + r32] * osc;
                    if (live && b < p.n && (!(QPAL_GEMM_KO & 64) || v == 12345.678f)) {
                        float *dst = p.out + (long)b * p.ldo + (long)sr * 32 + r32;
                        if constexpr (mode == 2) atomicAdd(dst, v);
                        else if constexpr (mode == 1) *dst += v;
                        else *dst = v;
                    }
                }
            });
        };
        if (p.sk > 1) epilogue(std::integral_constant<int, 2>{});
        else if (p.accumulate) epilogue(std::integral_constant<int, 1>{});
        else epilogue(std::integral_constant<int, 0>{});
        __syncthreads();  // the scratch is the next item's x buffer
    }
}

}  // namespace qpal
