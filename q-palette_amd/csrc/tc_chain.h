// Persistent "chain" form of the fused decode + GEMV (batch <= 8): ONE launch runs a whole sequence of DEPENDENT multi-job
// GEMV phases (q|k|v -> o -> gate|up -> down -> next block's q|k|v ...), with the dependency kept by an in-kernel arrival
// counter instead of a kernel boundary per phase.
//
// Why (profiles/r02_*): at batch 1 a projection is 2-48 MB, i.e. 0.3-7 us of streaming, while a kernel boundary plus the
// cold prologue behind it (kernel-argument fetch, codebook image, first-byte latency of the weights) and the reduction
// tail cost ~5 us — more than half of a decoded token was fixed cost during which HBM and the VALU idled.  Weights never
// depend on activations, so a workgroup that has finished phase p immediately requests phase p+1's first weight steps and
// DECODES them (the decode needs the codebook, not x) while the other workgroups finish, the arrival counter propagates
// and x becomes readable; only the matrix-pipe MACs wait for x.  The codebook image is built once per chain, not per
// launch.  The reference (kernels/tcq-kernels/src/inference.cu:408-634, one cold launch per linear) has no counterpart.
//
// Shape: 512-thread workgroups (8 waves = 2 per SIMD, up to 256 VGPRs each: room for 4 decoded steps = 128 VGPRs of MFMA A
// fragments per lane plus 4 steps of packed weights in flight), one workgroup per CU, grid = number of CUs so that every
// workgroup is resident (the in-kernel dependency needs that).  Steps are processed in groups of 4: group g+1's packed
// words are requested while group g is decoded.
//
// Dependency protocol (MI355X_MICROARCH.md, inter-workgroup visibility): per phase every workgroup — after the
// `s_waitcnt vmcnt(0)` of each of its waves — adds 1 to one of 8 arrival counters (shard = blockIdx % 8, one 128-byte
// line each); before touching x of phase p+1 one wave polls the 8 shards (agent-scope relaxed loads) until all show
// phase p's arrivals, then a workgroup barrier.  Outputs a later phase reads (job.publish) are stored agent-scope
// (write-through `sc1`), zero-fills for later split-K atomics likewise; x produced inside the launch (job.x_fresh /
// x_f32) is loaded agent-scope.  Counters only ever grow (wrap-safe compare); ws->epoch carries the phase count from
// launch to launch so that a captured graph can be replayed without resetting anything.
#pragma once
#include "tc_kernels.h"

namespace qpal {

// Shape knobs (perf experiments override them with -D; the defaults are what measured best):
//   W waves per workgroup (8: 2 per SIMD, 256 VGPRs; 16: 4 per SIMD, 128 VGPRs), G steps decoded ahead of the dependency
//   (32 VGPRs of A fragments each), R ring slots of packed steps (prefetch distance, ~8 VGPRs each), PAIRS: the steady
//   state runs two interleaved steps (more ILP for 2 waves per SIMD, ~60 VGPRs)
#ifndef QPAL_CHAIN_W
#define QPAL_CHAIN_W 16
#endif
#ifndef QPAL_CHAIN_G
#define QPAL_CHAIN_G 0
#endif
#ifndef QPAL_CHAIN_R
#define QPAL_CHAIN_R 2
#endif
#ifndef QPAL_CHAIN_PAIRS
#define QPAL_CHAIN_PAIRS 0
#endif
//   PIPE: software pipeline inside a phase — the decode of step t + G runs beside the MACs of step t (G == R): no wait
//   between a step's LDS gathers and their use, which is what lets two waves per SIMD keep the VALU busy
#ifndef QPAL_CHAIN_PIPE
#define QPAL_CHAIN_PIPE 0
#endif
constexpr int kChainWaves = QPAL_CHAIN_W;
constexpr int kChainLog2W = kChainWaves == 16 ? 4 : 3;
static_assert(kChainWaves == 8 || kChainWaves == 16, "waves per workgroup");
constexpr int kChainThreads = 64 * kChainWaves;
constexpr int kChainG = QPAL_CHAIN_G;
constexpr int kChainR = QPAL_CHAIN_R;
static_assert(kChainG <= kChainR && kChainR >= 2, "ring");
constexpr bool kChainPairs = QPAL_CHAIN_PAIRS != 0 && kChainR % 2 == 0 && kChainG % 2 == 0;
constexpr bool kChainPipe = QPAL_CHAIN_PIPE != 0;
static_assert(!kChainPipe || (kChainR % kChainG == 0 && kChainG >= 2), "pipelined mode: R a multiple of G");
constexpr int kChainShards = 8;
constexpr unsigned kChainGuard = 1u << 21;  // polls before a wait gives up and reports (never hang the GPU)

struct ChainWs {                          // device memory, 2 KiB, zero-filled ONCE by the caller
    unsigned shard[kChainShards][32];     // arrival counters, one 128-byte line each
    unsigned epoch;                       // phases completed by earlier launches
    unsigned error;                       // != 0: 1 + index of the phase whose wait gave up
    unsigned pad[254];
};
static_assert(sizeof(ChainWs) == QPAL_CHAIN_WS_BYTES, "include/qpal.h: QPAL_CHAIN_WS_BYTES");

struct ChainHeader {  // first 64 bytes of a chain blob (host-built, copied to the device by the caller)
    unsigned magic;
    int nphases;
    int grid;
    int family;       // 1: TCQ, 2: LUT
    int a, b, c, d;   // TCQ: S, KV1, KV2, split; LUT: bits, vec
    int n;
    int pad[7];
};
static_assert(sizeof(ChainHeader) == 64, "chain header");
constexpr unsigned kChainMagic = 0x51434831u;  // "QCH1"

// LDS beside the codebook image: reduction buffer [8 waves][n][32] fp32 + x [n][k] fp16 (+ 64-byte zero pad)
template <class C>
constexpr int chain_scratch_bytes() {
    constexpr int avail = 160 * 1024 - C::LDS_DWORDS * 4 - 1024;
    return avail > 72 * 1024 ? 72 * 1024 : avail;
}

template <class Codec>
__device__ __forceinline__ void decode_step(const uint32_t *lut, uint32_t laneoff, const uint32_t (&w)[Codec::NW],
                                            half8_t (&af)[8]) {
    static_for<0, 4>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        uint32_t nh = 0u;
        if constexpr (Codec::kNeedsNext) nh = row16_next(Codec::template head<g>(w));
        static_for<0, 2>([&](auto jc) {
            constexpr int jl = decltype(jc)::value;
            const u32x4 a{Codec::template pair<g, jl>(lut, laneoff, w, nh), Codec::template pair<g, jl + 4>(lut, laneoff, w, nh),
                          Codec::template pair<g, jl + 2>(lut, laneoff, w, nh),
                          Codec::template pair<g, jl + 6>(lut, laneoff, w, nh)};
            af[g * 2 + jl] = __builtin_bit_cast(half8_t, a);
        });
    });
}

__device__ __forceinline__ void mfma_step(const half8_t (&af)[8], const u32x4 (&xb)[1][2], Acc<1> &acc) {
    static_for<0, 4>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int ksub = g >> 1, msub = g & 1;
        static_for<0, 2>([&](auto jc) {
            constexpr int jl = decltype(jc)::value;
            acc.v[0][msub * 2 + jl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                af[g * 2 + jl], __builtin_bit_cast(half8_t, xb[0][ksub]), acc.v[0][msub * 2 + jl], 0, 0, 0);
        });
    });
}

// Ring slots are plain uint32_t[NWMAX] arrays; the codecs see their own uint32_t[NW] copies (register renames once the
// arrays are scalarised — a reinterpret_cast view of a slot kept the whole ring in scratch memory instead).
template <int NWMAX>
using slot_t = uint32_t __attribute__((ext_vector_type(NWMAX)));  // a first-class value: always registers

template <int NW, int NWMAX>
__device__ __forceinline__ void slot_load(const StreamView &sv, int step, int lane, slot_t<NWMAX> &slot) {
    static_assert(NW <= NWMAX, "codec words");
#ifdef QPAL_TEST_NOLOAD  // timing experiment only (results are garbage): what do the steps cost when no weight is loaded?
    {
        slot_t<NWMAX> v = slot;
        static_for<0, NW>([&](auto ic) { v[decltype(ic)::value] = v[decltype(ic)::value] * 1664525u + 1013904223u + (uint32_t)(step + lane); });
        slot = v;
        return;
    }
#endif
    uint32_t t[NW];
    load_step_w<NW>(sv, step, lane, t);
    slot_t<NWMAX> v = slot;
    static_for<0, NW>([&](auto ic) { v[decltype(ic)::value] = t[decltype(ic)::value]; });
    slot = v;
}
template <class Codec, int NWMAX>
__device__ __forceinline__ void slot_decode(const uint32_t *lut, uint32_t laneoff, const slot_t<NWMAX> &slot, half8_t (&af)[8]) {
    uint32_t w[Codec::NW];
    static_for<0, Codec::NW>([&](auto ic) { w[decltype(ic)::value] = slot[decltype(ic)::value]; });
    decode_step<Codec>(lut, laneoff, w, af);
}
template <class Codec, int NWMAX>
__device__ __forceinline__ void slot_gemv(const uint32_t *lut, uint32_t laneoff, const slot_t<NWMAX> &slot,
                                          const u32x4 (&xb)[1][2], Acc<1> &acc) {
    uint32_t w[Codec::NW];
    static_for<0, Codec::NW>([&](auto ic) { w[decltype(ic)::value] = slot[decltype(ic)::value]; });
    gemv_step<Codec, 1>(lut, laneoff, w, xb, acc);
}

// A wave's share of one phase (everything wave-uniform).  The packed words of a phase's first G steps are requested while
// the PREVIOUS phase is still running (ring slots are refilled across the phase boundary), so the context of phase p+1 is
// computed at the top of phase p.
struct ChainCtx {
    const uint32_t *base;  // first dword of the wave's supertile row in its stream
    int nsc, col0;         // supertile columns of the stream, first x column of the stream
    int s0, T;             // first step and number of steps of the wave's chunk (T == 0: nothing to do)
    int on2;               // chunk lies in stream 2 (second codec of a combt layer)
    int j;                 // job of the workgroup's item
    int rg, ks;            // row group and K split index of the item
    int has;               // the workgroup has an item in this phase
};

template <bool TWO, int NW1, int NW2>
__device__ __forceinline__ ChainCtx chain_ctx(const TcMultiParams &mp, int wave) {
    ChainCtx c{};
    const int gitem = blockIdx.x;
    c.has = gitem < mp.total_items;
    int j = 0, item_begin = 0;
#pragma unroll
    for (int i = 0; i < kMaxJobs - 1; i++) {
        if (gitem >= mp.item_end[i] && i + 1 < mp.njobs) {
            j = i + 1;
            item_begin = mp.item_end[i];
        }
    }
    c.j = j;
    const TcParams &p = mp.job[j];
    const int log2_wpr = p.log2_wpr, wpr = 1 << log2_wpr;
    const int rloc = wave >> log2_wpr, wr = wave & (wpr - 1), log2_rpw = kChainLog2W - log2_wpr;
    const int item = gitem - item_begin;
    int rg = item, ks = 0;
    if (p.sk > 1) {
        rg = item / p.sk;
        ks = item - rg * p.sk;
    }
    c.rg = rg;
    c.ks = ks;
    const int sr = (rg << log2_rpw) + rloc;
    const bool live = c.has && sr < p.nrows;
    const int ch = ks * wpr + wr;
    const bool on2 = TWO && ch >= p.nc1;
    const int cc = on2 ? ch - p.nc1 : ch;
    const int base = on2 ? p.base2 : p.base1, rem = on2 ? p.rem2 : p.rem1;
    c.on2 = on2;
    c.s0 = cc * base + (cc < rem ? cc : rem);
    c.T = live ? base + (cc < rem ? 1 : 0) : 0;
    if (on2) {
        c.base = p.c2 + (long)(live ? sr : 0) * p.nsc2 * 16 * NW2;
        c.nsc = p.nsc2;
        c.col0 = p.col2;
    } else {
        c.base = p.c1 + (long)(live ? sr : 0) * p.nsc1 * 16 * NW1;
        c.nsc = p.nsc1;
        c.col0 = 0;
    }
    return c;
}

// packed words of step t (< c.T) of a chunk -> ring slot
template <class C1, class CB, int NWMAX>
__device__ __forceinline__ void ring_load(const ChainCtx &c, int t, int lane, slot_t<NWMAX> &slot) {
    const StreamView sv{c.base, c.nsc, c.col0};
    if (c.on2) slot_load<CB::NW, NWMAX>(sv, c.s0 + t, lane, slot);
    else slot_load<C1::NW, NWMAX>(sv, c.s0 + t, lane, slot);
}

// slot d = t % R has just been consumed as step t of `cur`: its next content is cur's step t + R, or — past the end of the
// chunk — step d of the NEXT phase's chunk (requested a whole phase tail ahead of its use)
template <class Codec, class C1, class CB, int R, int NWMAX>
__device__ __forceinline__ void ring_refill(const ChainCtx &cur, const ChainCtx &nxt, int t, int d, int lane, slot_t<NWMAX> &slot) {
    if (t + R < cur.T) {
        const StreamView sv{cur.base, cur.nsc, cur.col0};
        slot_load<Codec::NW, NWMAX>(sv, cur.s0 + t + R, lane, slot);
    } else if (d < nxt.T) {
        ring_load<C1, CB, NWMAX>(nxt, d, lane, slot);
    }
}

// activations of step `step` from the LDS copy; a step outside the wave's chunk reads the zero pad
__device__ __forceinline__ void chain_x(const ChainCtx &c, const uint16_t *xs, int k, int n, int zero_off, int t, int lane,
                                        u32x4 (&xb)[1][2]) {
    const int sc = (c.s0 + t) * 4 + (lane >> 4);
    const bool live = t < c.T && sc < c.nsc;
    const int cl = lane & 15;
    int b = cl >> 1;
    b = b < n ? b : n - 1;
    const int off = b * k + c.col0 + sc * 32 + 4 * (cl & 1);
    const uint16_t *row = xs + (live ? off : zero_off);
#pragma unroll
    for (int ksub = 0; ksub < 2; ksub++) {
        const u32x2 lo = *reinterpret_cast<const u32x2 *>(row + 16 * ksub);
        const u32x2 hi = *reinterpret_cast<const u32x2 *>(row + 16 * ksub + 8);
        xb[0][ksub] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
}

// before the dependency: decode the chunk's first G steps into MFMA A fragments; every consumed slot is refilled at once
template <class Codec, class C1, class CB, int G, int R, int NWMAX>
__device__ __forceinline__ void chain_ahead(const uint32_t *lut, uint32_t laneoff, const ChainCtx &cur, const ChainCtx &nxt,
                                            int lane, slot_t<NWMAX> (&wq)[R], half8_t (&af)[G > 0 ? G : 1][8]) {
    static_for<0, G>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        // unconditional: a branch around a step that DEFINES 32 fragment registers made the allocator spill hundreds of
        // VGPRs.  A slot the chunk does not use (T < G: the planner avoids it) still holds words that arrived long ago
        // (its request for the next phase goes out after this function), decodes to finite values and meets zero
        // activations.
        slot_decode<Codec, NWMAX>(lut, laneoff, wq[d], af[d]);
        if constexpr (!kChainPipe) {
            if (d < cur.T) ring_refill<Codec, C1, CB, R, NWMAX>(cur, nxt, d, d, lane, wq[d]);
        } else {
            __builtin_amdgcn_sched_barrier(0);  // one step at a time: G steps of gathers in flight at once need no more registers
        }
    });
    if constexpr (kChainPipe) {
        static_for<0, G>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if (d < cur.T) ring_refill<Codec, C1, CB, R, NWMAX>(cur, nxt, d, d, lane, wq[d]);
        });
    }
}

// after the dependency: MACs of the decoded steps, then the rest of the chunk (decode + MACs, R steps in flight)
template <class Codec, class C1, class CB, int G, int R, bool PAIRS, int NWMAX>
__device__ __forceinline__ void chain_run(const uint32_t *lut, uint32_t laneoff, const ChainCtx &cur, const ChainCtx &nxt,
                                          const uint16_t *xs, int k, int n, int zero_off, int lane, slot_t<NWMAX> (&wq)[R],
                                          const half8_t (&af)[G > 0 ? G : 1][8], Acc<1> &acc) {
    static_for<0, G>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        u32x4 xb[1][2];
        chain_x(cur, xs, k, n, zero_off, d, lane, xb);  // zero pad for d >= T
        mfma_step(af[d], xb, acc);
    });
    auto one = [&](auto dc, int t) {
        constexpr int d = decltype(dc)::value;
        if (t < cur.T) {
#ifdef QPAL_CHAIN_PRIO
            // "least progress first": the SIMD arbiter serves the highest priority, then the oldest wave — left alone, the
            // same waves always win and the losers run their steps alone at the end (1-wave issue rate, cold prefetch)
            {
                const int left = ((cur.T - t) * 4 - 1) / cur.T;  // 3 .. 0
                if (left >= 3) __builtin_amdgcn_s_setprio(3);
                else if (left == 2) __builtin_amdgcn_s_setprio(2);
                else if (left == 1) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
#endif
            u32x4 xb[1][2];
            chain_x(cur, xs, k, n, zero_off, t, lane, xb);
            slot_gemv<Codec, NWMAX>(lut, laneoff, wq[d], xb, acc);
            ring_refill<Codec, C1, CB, R, NWMAX>(cur, nxt, t, d, lane, wq[d]);
        }
    };
    // first trip: slots G .. R-1 (slots 0 .. G-1 were decoded ahead)
    static_for<G, R>([&](auto dc) { one(dc, decltype(dc)::value); });
    for (int t0 = R; t0 < cur.T; t0 += R) {
        if (PAIRS && t0 + R <= cur.T) {
            // full trips as straight-line PAIRS of steps: two waves per SIMD do not hide the LDS gather latency of one
            // step, the second step's address arithmetic does
            static_for<0, R / 2>([&](auto dc) {
                constexpr int d = 2 * decltype(dc)::value;
                u32x4 xa[1][2], xb[1][2];
                chain_x(cur, xs, k, n, zero_off, t0 + d, lane, xa);
                chain_x(cur, xs, k, n, zero_off, t0 + d + 1, lane, xb);
                slot_gemv<Codec, NWMAX>(lut, laneoff, wq[d], xa, acc);
                slot_gemv<Codec, NWMAX>(lut, laneoff, wq[d + 1], xb, acc);
                ring_refill<Codec, C1, CB, R, NWMAX>(cur, nxt, t0 + d, d, lane, wq[d]);
                ring_refill<Codec, C1, CB, R, NWMAX>(cur, nxt, t0 + d + 1, d + 1, lane, wq[d + 1]);
            });
        } else {
            static_for<0, R>([&](auto dc) { one(dc, t0 + decltype(dc)::value); });
        }
    }
}

// Pipelined form (R a multiple of G).  Ring slot = step % R, fragment set = step % G.  On entry af[j] holds decoded step j
// (j < G) and ring slot s the packed step s (G <= s < R) resp. R + s (s < G).  A full trip of R sub-steps runs, for
// d = 0 .. R-1: the MACs of step t0 + d (fragments decoded G sub-steps ago, activations read a sub-step ago), then the decode
// of step t0 + d + G into the same fragment registers, then the refill of the slot just decoded; no instruction of a
// sub-step waits for one of its own LDS reads.  The tail (fewer than R + G steps left) runs the same sub-steps with
// uniform branches around the MACs only and, per remaining decode, one switch-free predicate that never DEFINES fragments
// on one path only: a dead decode is skipped together with the MACs that would read it.
template <class Codec, class C1, class CB, int G, int R, int NWMAX>
__device__ __forceinline__ void chain_run_pipe(const uint32_t *lut, uint32_t laneoff, const ChainCtx &cur, const ChainCtx &nxt,
                                               const uint16_t *xs, int k, int n, int zero_off, int lane, slot_t<NWMAX> (&wq)[R],
                                               half8_t (&af)[G][8], Acc<1> &acc) {
    u32x4 xb[1][2];
    chain_x(cur, xs, k, n, zero_off, 0, lane, xb);
    int t0 = 0;
#pragma nounroll
    for (; t0 + R + G <= cur.T; t0 += R) {  // every MAC and every decode of the trip is live
        static_for<0, R>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            constexpr int a = d % G, sl = (d + G) % R;
            u32x4 xn[1][2];
            chain_x(cur, xs, k, n, zero_off, t0 + d + 1, lane, xn);
            mfma_step(af[a], xb, acc);
            // the old fragments are dead before the new ones are born: same registers, no copies on the loop's back edge
            __builtin_amdgcn_sched_barrier(0);
            slot_decode<Codec, NWMAX>(lut, laneoff, wq[sl], af[a]);
            ring_refill<Codec, C1, CB, R, NWMAX>(cur, nxt, t0 + d + G, sl, lane, wq[sl]);
            xb[0][0] = xn[0][0];
            xb[0][1] = xn[0][1];
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    // tail: steps t0 .. T-1 (fewer than R + G), fragments of steps t0 .. t0+G-1 are decoded
    static_for<0, R + G>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        constexpr int a = d % G, sl = (d + G) % R;
        if (t0 + d < cur.T) {
            chain_x(cur, xs, k, n, zero_off, t0 + d, lane, xb);
            mfma_step(af[a], xb, acc);
            if (t0 + d + G < cur.T) {  // af[a] is next read by the MACs of step t0 + d + G, inside this same predicate chain
                slot_decode<Codec, NWMAX>(lut, laneoff, wq[sl], af[a]);
                ring_refill<Codec, C1, CB, R, NWMAX>(cur, nxt, t0 + d + G, sl, lane, wq[sl]);
            }
        }
    });
    if (cur.T < G) {  // fragments the chunk never used were decoded ahead unconditionally: nothing read them
    }
}

__device__ __forceinline__ unsigned ld_agent(const unsigned *p) {
    return __hip_atomic_load(as_global(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned *p, unsigned v) {
    __hip_atomic_store(as_global(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef QPAL_STAMPS
#define QPAL_CSTAMP(i) do { if (dbg && lane == 0) dbg[(((long)ph * gridDim.x + blockIdx.x) * kChainWaves + wave) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define QPAL_CSTAMP(i) do { } while (0)
#endif

template <class C1, class C2>
__global__ __launch_bounds__(kChainThreads) void tc_chain_kernel(const TcMultiParams *__restrict__ phases, int nphases,
                                                                 ChainWs *__restrict__ ws, unsigned long long *dbg) {
    constexpr bool TWO = !std::is_void_v<C2>;
    using CB = std::conditional_t<TWO, C2, C1>;
    constexpr int NWMAX = C1::NW > CB::NW ? C1::NW : CB::NW;
    constexpr int SCR = chain_scratch_bytes<C1>();
    constexpr int G = kChainG, R = kChainR;
    // ONE shared object with the codebook image first: at LDS address 0 the gather address is a single v_and_or_b32
    // (image anywhere else: v_and_b32 + v_add_u32 per weight pair, +32 VALU instructions per step — measured)
    struct Shared {
        uint32_t lut[C1::LDS_DWORDS];
        unsigned char scratch[SCR];
        unsigned wave_ctr;
    };
    __shared__ __attribute__((aligned(16))) Shared sh;
    uint32_t *const lut = sh.lut;
    unsigned char *const scratch = sh.scratch;
    unsigned &wave_ctr = sh.wave_ctr;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t laneoff = (uint32_t)(lane & (C1::C - 1)) << 2;
    if (tid == 0) wave_ctr = 0;  // first used after several workgroup barriers
    // phases completed before this launch (block 0 rewrites it at the very end, when every workgroup has long read it:
    // to get there it has seen every workgroup arrive at phase 0; with a single phase nobody waits, so a late reader of
    // the new value is harmless)
    const unsigned seq0 = (unsigned)__builtin_amdgcn_readfirstlane((int)ld_agent(&ws->epoch));
    const void *cur_tab = nullptr;
#ifdef QPAL_STAMPS
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif

    slot_t<NWMAX> wq[R];  // ring of packed steps: slot d holds steps d, d + R, ... of the current chunk, then step d of the next
    static_for<0, R>([&](auto dc) { wq[decltype(dc)::value] = slot_t<NWMAX>(0u); });
    ChainCtx nxt = chain_ctx<TWO, C1::NW, CB::NW>(phases[0], wave);
    static_for<0, R>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        if (d < nxt.T) ring_load<C1, CB, NWMAX>(nxt, d, lane, wq[d]);
    });

    for (int ph = 0; ph < nphases; ph++) {
        const TcMultiParams &mp = phases[ph];
        const ChainCtx cur = nxt;
        nxt = ChainCtx{};
        if (ph + 1 < nphases) nxt = chain_ctx<TWO, C1::NW, CB::NW>(phases[ph + 1], wave);
        const bool has = cur.has;  // workgroup-uniform
        QPAL_CSTAMP(0);

        // ---------------------------------------------------------------- A: everything that does not need x
        const TcParams &p = mp.job[cur.j];
        const int n = p.n, k = p.k;
        const int log2_wpr = p.log2_wpr;
        const int wpr = 1 << log2_wpr;
        const int log2_rpw = kChainLog2W - log2_wpr;
        const int zero_off = n * k;
        const int rg = cur.rg;
        float *red = reinterpret_cast<float *>(scratch);                                  // [W][n][32]
        uint16_t *xs = reinterpret_cast<uint16_t *>(scratch + kChainWaves * 32 * 4 * n);  // [n][k] + 32 zero halves
        uint32_t wraw = 0;
        if (has && p.wscale && tid < (32 << log2_rpw) && (rg << log2_rpw) + (tid >> 5) < p.nrows)
            wraw = as_global(p.wscale)[((rg << log2_rpw) + (tid >> 5)) * 32 + (tid & 31)];
        if (has && p.tab != cur_tab) {  // workgroup-uniform; every wave is past the previous phase's decode (reduce barrier)
            C1::build(lut, p.tab, tid, kChainThreads);
            cur_tab = p.tab;
            __syncthreads();
        }
        QPAL_CSTAMP(1);
        half8_t af[G > 0 ? G : 1][8];
        if (cur.on2) chain_ahead<CB, C1, CB, G, R, NWMAX>(lut, laneoff, cur, nxt, lane, wq, af);
        else chain_ahead<C1, C1, CB, G, R, NWMAX>(lut, laneoff, cur, nxt, lane, wq, af);
        // ring slots this chunk does not use go to the next phase now
        static_for<0, R>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if (d >= cur.T && d < nxt.T) ring_load<C1, CB, NWMAX>(nxt, d, lane, wq[d]);
        });
        QPAL_CSTAMP(2);

        // ---------------------------------------------------------------- B: the dependency
#ifdef QPAL_TEST_NODEP  // timing experiment only (valid with static activations): what does a chain cost with NO dependency protocol?
        if (false) {
#else
        if (ph > 0 && wave == 0) {
#endif
            const unsigned done = seq0 + (unsigned)ph;  // phases whose arrivals must be in
            unsigned need = 0;
            if (lane < kChainShards) need = ((gridDim.x + (kChainShards - 1) - lane) / kChainShards) * done;
            unsigned guard = 0;
            for (;;) {
                unsigned seen = need;
                if (lane < kChainShards) seen = ld_agent(&ws->shard[lane][0]);
                if (__all((int)(seen - need) >= 0)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++guard > kChainGuard) {
                    if (lane == 0) st_agent(&ws->error, 1u + (unsigned)ph);
                    break;
                }
            }
        }
        __syncthreads();
        QPAL_CSTAMP(3);

        // ---------------------------------------------------------------- C: activations -> LDS
        if (has) {  // the host plans a chain only where x fits the LDS scratch
            const int total = n * k;  // multiple of 8 halves
            if (p.x_f32) {            // x = fp16(src * scale), src written by an earlier phase: agent-scope loads
                const unsigned *src = reinterpret_cast<const unsigned *>(p.x_f32);
                for (int i = tid; i < total + 32; i += kChainThreads) {
                    float v = 0.f;
                    if (i < total) v = __builtin_bit_cast(float, ld_agent(src + i)) * p.x_f32_scale;
                    xs[i] = __builtin_bit_cast(uint16_t, (_Float16)v);
                }
#ifdef QPAL_TEST_FRESHX  // timing experiment: every phase stages x with agent-scope loads (the round trip an embedded-flag poll would pay)
            } else if (true) {
#else
            } else if (p.x_fresh) {
#endif
                const unsigned *src = reinterpret_cast<const unsigned *>(p.x);
                for (int i = tid; i < (total + 32) / 2; i += kChainThreads)
                    reinterpret_cast<uint32_t *>(xs)[i] = i < total / 2 ? ld_agent(src + i) : 0u;
            } else {
                for (int i = tid * 8; i < total + 32; i += kChainThreads * 8) {
                    u32x4 v{0u, 0u, 0u, 0u};
                    if (i < total) v = *(gptr<const u32x4>)as_global(p.x + i);
                    *reinterpret_cast<u32x4 *>(xs + i) = v;
                }
            }
        }
        if (mp.zero_chunks > 0) {  // zero-fill for a LATER phase's split-K atomics (they execute at the memory side: write through)
            unsigned *z = reinterpret_cast<unsigned *>(mp.zero);
            for (int i = blockIdx.x * kChainThreads + tid; i < mp.zero_chunks * 4; i += gridDim.x * kChainThreads) st_agent(z + i, 0u);
        }
        __syncthreads();
        QPAL_CSTAMP(4);

        // ---------------------------------------------------------------- D: MACs (+ the steps not decoded ahead)
        Acc<1> acc;
        static_for<0, 4>([&](auto ac) { acc.v[0][decltype(ac)::value] = float4_t{0.f, 0.f, 0.f, 0.f}; });
        if constexpr (kChainPipe) {
            if (cur.on2) chain_run_pipe<CB, C1, CB, G, R, NWMAX>(lut, laneoff, cur, nxt, xs, k, n, zero_off, lane, wq, af, acc);
            else chain_run_pipe<C1, C1, CB, G, R, NWMAX>(lut, laneoff, cur, nxt, xs, k, n, zero_off, lane, wq, af, acc);
        } else {
            if (cur.on2) chain_run<CB, C1, CB, G, R, kChainPairs, NWMAX>(lut, laneoff, cur, nxt, xs, k, n, zero_off, lane, wq, af, acc);
            else chain_run<C1, C1, CB, G, R, kChainPairs, NWMAX>(lut, laneoff, cur, nxt, xs, k, n, zero_off, lane, wq, af, acc);
        }
#ifdef QPAL_CHAIN_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        QPAL_CSTAMP(5);

        // ---------------------------------------------------------------- E: cross-wave sum, epilogue, arrival
        if (has) {
            const int q = lane >> 4, cidx = lane & 15;
            const int b = cidx >> 1;
            const bool writer = (cidx & 1) == 0 && b < n;
            float *dst = red + ((wave * n + (b < n ? b : 0)) * 32) + 2 * q;
            static_for<0, 4>([&](auto ac) {
                constexpr int a = decltype(ac)::value;
                const float4_t dd = acc.v[0][a];
                const float v0 = dd[0] + lane_xor<1>(dd[1]);
                const float v1 = dd[2] + lane_xor<1>(dd[3]);
                if (writer) {
                    dst[8 * a] = v0;
                    dst[8 * a + 1] = v1;
                }
            });
        }
        __syncthreads();
        QPAL_CSTAMP(6);
        if (has && tid < (32 << log2_rpw)) {
            const int r = tid & 31, rl = tid >> 5;
            const int srow = (rg << log2_rpw) + rl;
            if (srow < p.nrows) {
                const float osc = p.wscale ? p.oscale * (float)__builtin_bit_cast(_Float16, (uint16_t)wraw) : p.oscale;
                for (int b = 0; b < n; b++) {
                    float v = 0.f;
                    for (int qq = 0; qq < wpr; qq++) v += red[(((rl << log2_wpr) + qq) * n + b) * 32 + r];
                    float *dst = p.out + (long)b * p.ldo + (long)srow * 32 + r;
                    v *= osc;
                    if (p.sk > 1) __hip_atomic_fetch_add(as_global(dst), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else if (p.publish) st_agent(reinterpret_cast<unsigned *>(dst), __builtin_bit_cast(unsigned, v));
                    else *as_global(dst) = v;
                }
            }
        }
        // Arrival: every wave waits for its own stores (but not for the packed words it has already requested for the
        // next phase: they are younger, and memory operations complete in issue order), the last of the 8 signals.
#ifndef QPAL_TEST_NODEP
        if (tid < (32 << log2_rpw) || mp.zero_chunks > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const unsigned old = __hip_atomic_fetch_add(&wave_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((old & (kChainWaves - 1)) == kChainWaves - 1)
                __hip_atomic_fetch_add(as_global(&ws->shard[blockIdx.x & (kChainShards - 1)][0]), 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        }
#endif
        QPAL_CSTAMP(7);
    }
    if (blockIdx.x == 0 && tid == 0) st_agent(&ws->epoch, seq0 + (unsigned)nphases);
#ifdef QPAL_STAMPS
    if (dbg && tid == 0) {  // shader clock held by this CU over the launch: (memtime ticks) / (100 MHz realtime ticks)
        unsigned long long *c = dbg + (long)nphases * gridDim.x * kChainWaves * 8 + blockIdx.x * 2;
        c[0] = __builtin_amdgcn_s_memtime() - clk0;
        c[1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#endif
}

}  // namespace qpal
