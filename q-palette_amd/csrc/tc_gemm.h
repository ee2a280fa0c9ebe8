// Fused decode + skinny GEMM for batches 9..64 (and, in passes of 64, beyond): out[n][m] = x[n][k] W^T with W decoded on the
// fly, the activations of a step shared by the whole workgroup through LDS.
//
// Why a second kernel: tc_gemv_kernel's waves each own a K-chunk of a supertile row, so for a batch every wave fetches its own
// [n][128] slice of x per step from L2 (16 KiB per wave-step at n = 64) and waits for it — at batch 64 the step was bound by
// that round trip, not by the decode or the matrix pipe (7.7 ms per Llama-8B step: 0.05 of the MFMA roofline).  Here the 8
// waves of a workgroup own 8 DIFFERENT supertile rows and walk the SAME K range in lockstep: the [n][128] tile of x of a step
// is loaded ONCE per workgroup, two steps ahead (global -> registers -> LDS, four tile slots, one barrier per two steps), in the
// layout the MFMA B fragments are read in (one ds_read_b128 per batch group and ksub).  A wave sees its whole K range, so
// there is no cross-wave reduction; split-K over workgroups (atomics) only where a layer has too few rows to fill the chip.
//
// Decode, MFMA mapping (virtual rows: tile row x column half), codecs and packed formats are tc_kernels.h's — same arithmetic
// order per (row, batch) up to the K split.  Replaces, for bs > 8, the reference's decode-to-HBM + cuBLAS path
// (lib/linear/tcq_linear.py:75-84, vq_linear.py:60-66) up to the batch where that path wins again.
#pragma once
#include "tc_kernels.h"

namespace qpal {

#ifndef QPAL_GEMM_WAVES  // (experiment: -DQPAL_GEMM_WAVES=16 — four waves per SIMD, <= 128 VGPRs)
#define QPAL_GEMM_WAVES 8
#endif
constexpr int kGemmWaves = QPAL_GEMM_WAVES;
constexpr int kGemmXRow = 144;              // bytes per (batch row, column half) row of a step's x tile: 128 + 16 pad (bank spread)
constexpr int kGemmXGroup = 16 * kGemmXRow;  // one batch group (8 rows x 2 column halves)

// host: the packed item table of a launch (<= 4 jobs, boundaries < 1023), 0: none (the kernel scans the table)
inline int gemm_item_table(const TcMultiParams &mp) {
    if (mp.njobs < 2 || mp.njobs > 4) return 0;
    int ie = 1 << 30;
    for (int j = 0; j < 3; j++) {
        const int end = j < mp.njobs - 1 ? mp.item_end[j] : 0x3ff;
        ie |= (end < 0x3ff ? end : 0x3ff) << (10 * j);
    }
    return ie;
}

// geometry of one job (host: plan_gemm): items = ceil(nrows / 8) * sk, item -> (row group, K split)
//   uses TcParams: nrows, nsc1/2, st1/2, col2, sk, out/ldo, wscale/oscale, accumulate, c1/c2, x, tab, n, k

// Software-pipelined step (QPAL_GEMM_PIPE, default): a wave issues in order, and the compiler's own order per A fragment is
// decode -> wait for the gathers -> NBG MFMAs, which leaves the matrix pipe idle during the decode and the VALU / LDS idle during
// the MFMAs unless the SIMD's other wave fills the gaps.  Here fragment t + 1 (t = 4 ksub + 2 msub + jl) is decoded BETWEEN the
// MFMAs of fragment t — the order is pinned with sched_barrier fences, one weight pair in front of each of the first MFMAs, the
// rest of the MFMAs cover the gather latency — and the B fragments of ksub 1 replace those of ksub 0 one batch group at a time
// behind their last use.  Measured: +-0.5 % at every batch (profiles/r03_gemm_knockouts.txt) — the two waves of a SIMD did
// overlap each other's phases; what a step loses is the lockstep wait at the barrier.
#ifndef QPAL_GEMM_PIPE
#define QPAL_GEMM_PIPE 1
#endif
// Timing experiments only (results invalid): QPAL_GEMM_KO bit 1: no MFMAs (operands xor-folded), 2: no B-fragment reads, 4: no x
// staging inside the loop, 8: no per-step barrier, 32: no weight loads inside the loop, 64: no output stores / atomics, 128: no codebook image build, 256: no steps at all (what a launch costs before and after them); -DQPAL_KO_GATHER: no codebook gathers.
#ifndef QPAL_GEMM_SLOTS
#define QPAL_GEMM_SLOTS 4
#endif
#ifndef QPAL_GEMM_BUILD_U
#define QPAL_GEMM_BUILD_U 1
#endif
#ifndef QPAL_GEMM_KO
#define QPAL_GEMM_KO 0
#endif
template <class Codec, int T, int I>
__device__ __forceinline__ uint32_t gemm_pair(const uint32_t *lut, uint32_t laneoff, const uint32_t (&w)[Codec::NW]) {
    constexpr int g = T >> 1, jl = T & 1;
    constexpr int idx[4] = {jl, jl + 4, jl + 2, jl + 6};  // fragment order (jh, isB): i = jl + 2 jh + 4 isB
    uint32_t nh = 0u;
    if constexpr (Codec::kNeedsNext) nh = row16_next(Codec::template head<g>(w));  // (one per g: CSE)
    return Codec::template pair<g, idx[I]>(lut, laneoff, w, nh);
}

template <class Codec, int NBG>
__device__ __forceinline__ void gemm_step_pipe(const uint32_t *lut, uint32_t laneoff, const uint32_t (&w)[Codec::NW],
                                               const unsigned char *xt, int lane, Acc<NBG> &acc) {
    const unsigned char *xl = xt + (lane & 15) * kGemmXRow + (lane >> 4) * 32;
    u32x4 xb[NBG];
    static_for<0, NBG>([&](auto bc) {
        if constexpr (QPAL_GEMM_KO & 2) xb[decltype(bc)::value] = u32x4{laneoff, (uint32_t)lane, laneoff, (uint32_t)lane};
        else xb[decltype(bc)::value] = *reinterpret_cast<const u32x4 *>(xl + decltype(bc)::value * kGemmXGroup);
    });
    uint32_t a[4], an[4];
    static_for<0, 4>([&](auto ic) { a[decltype(ic)::value] = gemm_pair<Codec, 0, decltype(ic)::value>(lut, laneoff, w); });
    // pairs of the next fragment decoded in front of MFMA m of this one: all four in front of the first half of the MFMAs
    constexpr int PER = NBG >= 8 ? 1 : NBG >= 4 ? 2 : 4;  // pairs per slot
    static_for<0, 8>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int msub = (t >> 1) & 1, jl = t & 1;
        static_for<0, NBG>([&](auto bc) {
            constexpr int m = decltype(bc)::value;
            if constexpr (t < 7 && m * PER < 4) {
                static_for<0, PER>([&](auto ic) {
                    constexpr int i = m * PER + decltype(ic)::value;
                    an[i] = gemm_pair<Codec, t + 1, i>(lut, laneoff, w);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (QPAL_GEMM_KO & 1) {
                if constexpr (m == 0 || t == 0 || t == 4) {
                    const uint32_t f = m == 0 ? a[0] ^ a[1] ^ a[2] ^ a[3] : 0u;
                    const uint32_t fx = (t == 0 || t == 4) ? xb[m].x ^ xb[m].y ^ xb[m].z ^ xb[m].w : 0u;
                    acc.v[m][msub * 2 + jl][0] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, acc.v[m][msub * 2 + jl][0]) ^ f ^ fx);
                }
            } else
            acc.v[m][msub * 2 + jl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                __builtin_bit_cast(half8_t, u32x4{a[0], a[1], a[2], a[3]}), __builtin_bit_cast(half8_t, xb[m]), acc.v[m][msub * 2 + jl], 0, 0, 0);
            if constexpr (t == 3 && !(QPAL_GEMM_KO & 2)) xb[m] = *reinterpret_cast<const u32x4 *>(xl + m * kGemmXGroup + 16);  // ksub 1, behind the last use of ksub 0
            __builtin_amdgcn_sched_barrier(0);
        });
        static_for<0, 4>([&](auto ic) { a[decltype(ic)::value] = an[decltype(ic)::value]; });
    });
}

template <class Codec, int NBG>
__device__ __forceinline__ void gemm_step(const uint32_t *lut, uint32_t laneoff, const uint32_t (&w)[Codec::NW],
                                          const unsigned char *xt, int lane, Acc<NBG> &acc) {
    if constexpr (QPAL_GEMM_PIPE != 0) return gemm_step_pipe<Codec, NBG>(lut, laneoff, w, xt, lane, acc);
    const unsigned char *xl = xt + (lane & 15) * kGemmXRow + (lane >> 4) * 32;
    static_for<0, 2>([&](auto kc) {
        constexpr int ksub = decltype(kc)::value;
        __builtin_amdgcn_sched_barrier(0);  // one ksub's B fragments at a time (hoisting the second set spills at 8 batch groups)
        u32x4 xb[NBG];
        static_for<0, NBG>([&](auto bc) {
            constexpr int grp = decltype(bc)::value;
            xb[grp] = *reinterpret_cast<const u32x4 *>(xl + grp * kGemmXGroup + ksub * 16);
        });
        static_for<0, 2>([&](auto mc) {
            constexpr int msub = decltype(mc)::value;
            constexpr int g = ksub * 2 + msub;
            uint32_t nh = 0u;
            if constexpr (Codec::kNeedsNext) nh = row16_next(Codec::template head<g>(w));
            static_for<0, 2>([&](auto jc) {
                constexpr int jl = decltype(jc)::value;
                const u32x4 a{Codec::template pair<g, jl>(lut, laneoff, w, nh), Codec::template pair<g, jl + 4>(lut, laneoff, w, nh),
                              Codec::template pair<g, jl + 2>(lut, laneoff, w, nh), Codec::template pair<g, jl + 6>(lut, laneoff, w, nh)};
                const half8_t afrag = __builtin_bit_cast(half8_t, a);
                static_for<0, NBG>([&](auto bc) {
                    constexpr int grp = decltype(bc)::value;
                    acc.v[grp][msub * 2 + jl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                        afrag, __builtin_bit_cast(half8_t, xb[grp]), acc.v[grp][msub * 2 + jl], 0, 0, 0);
                });
            });
        });
    });
}

// which stream a global step of a row belongs to, and where its columns start
struct GemmStep {
    bool on2;
    int s;         // step inside the stream
    int col_base;  // first x column of the step
    int col_end;   // end of the stream's live columns
};
__device__ __forceinline__ GemmStep gemm_where(const TcParams &p, int g) {
    GemmStep r;
    r.on2 = g >= p.st1;
    r.s = r.on2 ? g - p.st1 : g;
    r.col_base = (r.on2 ? p.col2 : 0) + r.s * 128;
    r.col_end = r.on2 ? p.col2 + p.nsc2 * 32 : p.nsc1 * 32;
    return r;
}

// ---- buffer-addressed loads of the lockstep kernels (round 5).  The step loop is vector-issue bound (SQ counters of the 70B batch-16
// kernel: the issue port 88 % busy, profiles/r05_ab_batched_path.txt §10), and with flat addresses every step paid 64-bit address
// arithmetic for its tile chunk and its weights (v_lshl_add_u64 x 3, v_mad_u64_u32, clamps by v_cndmask: ~14 of 155 vector instructions,
// most of them quarter-rate) plus the copy of the prefetched weights into the current registers.  With buffer descriptors built once
// per item (x: the whole [n][k] block; weights: this wave's row of each stream) a load's address is ONE 32-bit add to a per-thread
// offset, and the range check of the descriptor does the clamping: batch rows >= n, supertile columns past the end of a stream and
// the re-read of dead chunks all return zeros (the per-step part is added to the VECTOR offset: the scalar offset is not range-checked).
// (the helpers — buf_rsrc_t, gemm_rsrc, buf_load_words_nt, gemm_w_lane_off, gemm_w_step_bytes — live in tc_kernels.h: the fused GEMV uses them too)
template <class C1, class C2, int NBG>
// eie: the launch's item table packed into one PRELOADED dword (gemm_item_table; tc_kernels.h early_args does the same for the GEMV
// kernels): a workgroup knows the job of its first item before the kernel-argument block has arrived and fetches THAT job's
// parameters in the first scalar-load round trip (every workgroup used to scan the table after one round trip and fetch its job
// in a second, dependent one)
__global__ __launch_bounds__(64 * kGemmWaves) void tc_gemm_kernel(const int eie, const TcMultiParams mp) {
    constexpr bool TWO = !std::is_void_v<C2>;
    using CB = std::conditional_t<TWO, C2, C1>;
    constexpr int W = kGemmWaves, NT = 64 * W;
    constexpr int XBUF = NBG * kGemmXGroup;
    constexpr int NCH = NBG * 8 * 16;                   // 16-byte chunks of one step's x tile
    constexpr int CPT = (NCH + NT - 1) / NT;            // chunks per thread
    constexpr int NWMAX = C1::NW > CB::NW ? C1::NW : CB::NW;
    // Tile slots: 4 where they fit beside the codebook image (two steps per buffer: the workgroup barrier — 9 % of a batch-64 token
    // as a per-step barrier, knock-out 8 — falls after every SECOND step, tiles are staged two steps ahead), else 2 (one per step)
    // (round 5, measured and not kept: two batch groups on TWO slots — 73 KiB of LDS, < 128 VGPRs — so that two workgroups share a CU as at
    // one batch group: +-0 on Llama-8B at batch 12 / 16, -7 % on the 70B shapes at batch 16; profiles/r05_ab_batched_path.txt)
    constexpr int NSLOT = (QPAL_GEMM_SLOTS == 4 && C1::LDS_DWORDS * 4 + 4 * XBUF <= 156 * 1024) ? 4 : 2;
    constexpr int AHEAD = NSLOT / 2;
    constexpr int XT = NSLOT * XBUF >= W * 1024 ? NSLOT * XBUF : W * 1024;  // the x buffers double as the epilogue's per-wave transposition scratch
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
        // (sk is a power of two — plan_gemm: the three divisions that stood here, two of them 64-bit, were ~400 of the ~560
        // instructions in front of an item's first load, ~1 us of every launch)
        const int lsk = __builtin_ctz((unsigned)p.sk);
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
        const buf_rsrc_t rs_x = gemm_rsrc(p.x, p.n * p.k * 2);
        const buf_rsrc_t rs_w1 = gemm_rsrc(sv1.base, sv1.nsc * 64 * C1::NW);
        const buf_rsrc_t rs_w2 = gemm_rsrc(sv2.base, sv2.nsc * 64 * CB::NW);
        const uint32_t wl1 = gemm_w_lane_off<C1::NW>(lane), wl2 = gemm_w_lane_off<CB::NW>(lane);
        uint32_t xvo[CPT];  // byte offset of this thread's chunk r inside x, without the step's column base
#pragma unroll
        for (int r = 0; r < CPT; r++) {
            const int id = tid + r * NT, b = id >> 4;
            xvo[r] = (id < NCH && b < p.n) ? (uint32_t)(b * p.k + 8 * (id & 15)) * 2u : kBufDead;
        }
        auto load_w = [&](int g, uint32_t(&dst)[NWMAX]) {
            const GemmStep st = gemm_where(p, g);
            if (TWO && st.on2) buf_load_words_nt<CB::NW>(rs_w2, wl2 + (uint32_t)st.s * gemm_w_step_bytes<CB::NW>(), reinterpret_cast<uint32_t(&)[CB::NW]>(dst));
            else buf_load_words_nt<C1::NW>(rs_w1, wl1 + (uint32_t)st.s * gemm_w_step_bytes<C1::NW>(), reinterpret_cast<uint32_t(&)[C1::NW]>(dst));
        };
        // chunk id -> (batch row b = id >> 4, 16-byte piece q = id & 15 of the step's 128 columns)
        // (unconditional loads from clamped addresses — a dead chunk re-reads x[0..7] and is zeroed when it is stored: a load under a
        // condition makes the compiler's wait counts conservative, and the wait for a tile then waits for the weights behind it)
        auto x_live = [&](int g, int r) {
            const GemmStep st = gemm_where(p, g);
            const int id = tid + r * NT;
            return id < NCH && (id >> 4) < p.n && st.col_base + 8 * (id & 15) < st.col_end;
        };
        auto load_x = [&](int g) {  // (columns past the end of the stream are zeroed when the chunk is stored: x_live)
            const GemmStep st = gemm_where(p, g);
#pragma unroll
            for (int r = 0; r < CPT; r++) xr[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xvo[r] + (uint32_t)st.col_base * 2u, 0, 0);
        };
        auto store_x = [&](unsigned char *buf, int g) {
#pragma unroll
            for (int r = 0; r < CPT; r++) {
                const int id = tid + r * NT;
                if (!x_live(g, r)) xr[r] = u32x4{0u, 0u, 0u, 0u};
                if (id < NCH) {
                    const int b = id >> 4, q = id & 15;
                    // piece q = (supertile col sc = q >> 2, ksub = (q >> 1) & 1, jh = q & 1): halves 0..3 belong to column half u = 0, 4..7 to u = 1
                    unsigned char *d = buf + (b >> 3) * kGemmXGroup + (2 * (b & 7)) * kGemmXRow + (q >> 2) * 32 + ((q >> 1) & 1) * 16 + (q & 1) * 8;
                    *reinterpret_cast<u32x2 *>(d) = u32x2{xr[r].x, xr[r].y};
                    *reinterpret_cast<u32x2 *>(d + kGemmXRow) = u32x2{xr[r].z, xr[r].w};
                }
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
            // (the 128 KiB image is 16 chunks per thread: two batches of 8 table reads instead of 16 dependent round trips)
            if constexpr (C1::LDS_DWORDS * 4 > 64 * 1024 && NBG <= 4) C1::template build<8>(lut, p.tab, tid, NT);
            else C1::template build<QPAL_GEMM_BUILD_U>(lut, p.tab, tid, NT);
            cur_tab = p.tab;
        }
        if constexpr (AHEAD == 2) {
            store_x(xt + XBUF, g0 + 1 < g1 ? g0 + 1 : g0);
#pragma unroll
            for (int r = 0; r < CPT; r++) xr[r] = xr1[r];
        }
        store_x(xt, g0);
        Acc<NBG> acc;
        static_for<0, NBG>([&](auto bc) {
            static_for<0, 4>([&](auto ac) { acc.v[decltype(bc)::value][decltype(ac)::value] = float4_t{0.f, 0.f, 0.f, 0.f}; });
        });
        __syncthreads();

        // the steps of ONE stream (one codec: no branch inside the loop body); the prefetch of the step after the last one
        // already belongs to the next stream (or re-requests the last step)
        auto run = [&](auto codec_c, int ga, int gb) {
            using CC = typename decltype(codec_c)::type;
            constexpr bool SECOND = TWO && std::is_same_v<CC, CB> && !std::is_same_v<C1, CB>;
            auto one_step = [&](int g, uint32_t(&wc)[NWMAX], uint32_t(&wn)[NWMAX]) {
                // Round 4: the tile FIRST (it is stored to LDS at the end of this step; the weights are next step's), and the weight
                // prefetch without a branch — the next step of THIS stream, the last one re-requesting itself; the first step of
                // the other stream is requested by the caller at the switch.  With the codec branch of load_w and the predicated
                // tile loads the compiler waited `vmcnt(0)` before the tile's LDS stores: every step waited for the NEXT step's
                // weights, a full HBM round trip where the step is shorter than that (batches 4..32).
                const int xg = g + AHEAD < g1 ? g + AHEAD : g1 - 1;
                if constexpr (!(QPAL_GEMM_KO & 4)) load_x(xg);
                if constexpr (QPAL_GEMM_KO & 32) {
#pragma unroll
                    for (int i = 0; i < NWMAX; i++) wn[i] = wc[i] + 1;
                } else {
                    const int gn = g + 1 < gb ? g + 1 : g;
                    if constexpr (SECOND) buf_load_words_nt<CC::NW>(rs_w2, wl2 + (uint32_t)(gn - p.st1) * gemm_w_step_bytes<CC::NW>(), reinterpret_cast<uint32_t(&)[CC::NW]>(wn));
                    else buf_load_words_nt<CC::NW>(rs_w1, wl1 + (uint32_t)gn * gemm_w_step_bytes<CC::NW>(), reinterpret_cast<uint32_t(&)[CC::NW]>(wn));
                }
                __builtin_amdgcn_sched_barrier(0);
                const int i = g - g0;
                if (live)  // (wave-uniform: a row past the end of the layer only keeps the staging and the barriers company)
                    gemm_step<CC, NBG>(lut, laneoff, reinterpret_cast<uint32_t(&)[CC::NW]>(wc), xt + (i & (NSLOT - 1)) * XBUF, lane, acc);
                // slot of step i + AHEAD: last read AHEAD steps ago, i.e. before the latest barrier
                if constexpr (!(QPAL_GEMM_KO & 4)) store_x(xt + ((i + AHEAD) & (NSLOT - 1)) * XBUF, xg);
                if constexpr (!(QPAL_GEMM_KO & 8)) {
                    if (AHEAD == 1 || (i & 1)) __syncthreads();
                }
            };
            // two steps per trip: the prefetched weights become the current ones by NAME (no register copies); an odd count ends on a
            // single step, and whoever continues (the other stream's run) loads its first step into wcur itself
            int g = ga;
            for (; g + 1 < gb; g += 2) {
                one_step(g, wcur, wnext);
                one_step(g + 1, wnext, wcur);
            }
            if (g < gb) one_step(g, wcur, wnext);
        };
        if constexpr (!(QPAL_GEMM_KO & 256)) {
            const int mid = g1 < p.st1 ? g1 : (g0 > p.st1 ? g0 : p.st1);
            if (g0 < mid) run(std::type_identity<C1>{}, g0, mid);
            if constexpr (TWO) {
                if (mid < g1) {
                    if (g0 < mid) load_w(mid, wcur);  // the switch to stream 2 inside an item: its first step, requested here (one exposed round trip)
                    run(std::type_identity<CB>{}, mid, g1);
                }
            }
        }

        if constexpr (AHEAD == 2) __syncthreads();  // (an odd number of steps ends without one: the scratch below overlays the tiles)
        // ---- epilogue: lane (q = lane >> 4, cidx = lane & 15 = 2 b' + u) holds, for batch row 8 grp + b' and a = msub * 2 + jl,
        // D[4 q + r][cidx]; valid where (r & 1) == u: on even lanes own r = 0 / 2 plus the odd neighbour's r = 1 / 3 are tile rows
        // 8 a + 2 q + {0, 1}.  One batch group at a time goes through a per-wave [8][32] fp32 scratch and leaves as 128-byte runs.
        float *scr = reinterpret_cast<float *>(xt) + wave * 256;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));  // the epilogue's addresses are computed HERE: hoisted out of the item loop they sat in
                                          // registers through the steps (1 spilled VGPR at 8 batch groups = scratch set-up per launch)
        const int q = lane_e >> 4, cidx = lane_e & 15;
        const int r32 = lane_e & 31;
        float osc = p.oscale;
        if (p.wscale && live) osc *= (float)__builtin_bit_cast(_Float16, p.wscale[sr * 32 + r32]);
        auto epilogue = [&](auto mode_c) {
            constexpr int mode = decltype(mode_c)::value;  // 0: store, 1: out += (residual add), 2: split-K atomics
            static_for<0, NBG>([&](auto bc) {
                constexpr int grp = decltype(bc)::value;
                static_for<0, 4>([&](auto ac) {
                    constexpr int a = decltype(ac)::value;
                    const float4_t d = acc.v[grp][a];
                    const float v0 = d[0] + lane_xor<1>(d[1]);
                    const float v1 = d[2] + lane_xor<1>(d[3]);
                    if ((cidx & 1) == 0) *reinterpret_cast<float2 *>(scr + (cidx >> 1) * 32 + 8 * a + 2 * q) = float2{v0, v1};
                });
                // (wave-private scratch: LDS operations of one wave complete in order)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int bl = 2 * i + (lane_e >> 5), b = 8 * grp + bl;
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
