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

constexpr int kChainWaves = 8;
constexpr int kChainThreads = 64 * kChainWaves;
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

// steps per group = decoded ahead of the dependency = prefetch distance: 4 (128 VGPRs of A fragments); 2 for the codecs
// with more than 10 packed dwords per lane and step (8-bit scalar codes: 16)
template <class C1, class C2>
constexpr int chain_group() {
    int nw = C1::NW;
    if constexpr (!std::is_void_v<C2>) nw = C2::NW > nw ? C2::NW : nw;
#ifdef QPAL_CHAIN_G
    return QPAL_CHAIN_G;
#else
    return nw > 10 ? 2 : 4;
#endif
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

template <int NW, int NWMAX>
__device__ __forceinline__ uint32_t (&words(uint32_t (&w)[NWMAX]))[NW] {
    static_assert(NW <= NWMAX, "codec words");
    return reinterpret_cast<uint32_t(&)[NW]>(w);
}

// Straight-line code per group: every step of a group is decoded whether or not it belongs to the wave's chunk [s0, s1)
// (branches around whole decode steps made the register allocator merge 32-register fragments over every path: 850
// spilled VGPRs).  Packed words start as zeros and loads are predicated, so a dead step decodes codebook entry 0
// (finite) against zero activations; its load is never issued (no extra HBM traffic).

// before the dependency: request the first group of steps
template <class Codec, int G, int NWMAX>
__device__ __forceinline__ void chain_request(const StreamView &sv, int s0, int s1, int lane, uint32_t (&wq)[G][NWMAX]) {
    static_for<0, G>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
#pragma unroll
        for (int i = 0; i < NWMAX; i++) wq[d][i] = 0u;
        if (s0 + d < s1) load_step_w<Codec::NW>(sv, s0 + d, lane, words<Codec::NW>(wq[d]));
    });
}

// before the dependency: decode the first group into MFMA A fragments, request the second group
template <class Codec, int G, int NWMAX>
__device__ __forceinline__ void chain_ahead(const uint32_t *lut, uint32_t laneoff, const StreamView &sv, int s0, int s1,
                                            int lane, uint32_t (&wq)[G][NWMAX], half8_t (&af)[G][8]) {
    static_for<0, G>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        decode_step<Codec>(lut, laneoff, words<Codec::NW>(wq[d]), af[d]);
        if (s0 + G + d < s1) load_step_w<Codec::NW>(sv, s0 + G + d, lane, words<Codec::NW>(wq[d]));
        __builtin_amdgcn_sched_barrier(0);
    });
}

// activations of step `step` from the LDS copy; a step outside the wave's chunk reads the zero pad
__device__ __forceinline__ void chain_x(const StreamView &sv, const uint16_t *xs, int k, int n, int zero_off, int step, bool valid,
                                        int lane, u32x4 (&xb)[1][2]) {
    const int sc = step * 4 + (lane >> 4);
    const bool live = valid && sc < sv.nsc;
    const int c = lane & 15;
    int b = c >> 1;
    b = b < n ? b : n - 1;
    const int off = b * k + sv.col0 + sc * 32 + 4 * (c & 1);
    const uint16_t *row = xs + (live ? off : zero_off);
#pragma unroll
    for (int ksub = 0; ksub < 2; ksub++) {
        const u32x2 lo = *reinterpret_cast<const u32x2 *>(row + 16 * ksub);
        const u32x2 hi = *reinterpret_cast<const u32x2 *>(row + 16 * ksub + 8);
        xb[0][ksub] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
}

// after the dependency: MACs of the decoded group, then the remaining groups (decode + MACs, next group in flight)
template <class Codec, int G, int NWMAX>
__device__ __forceinline__ void chain_run(const uint32_t *lut, uint32_t laneoff, const StreamView &sv, const uint16_t *xs, int k,
                                          int n, int zero_off, int s0, int s1, int lane, uint32_t (&wq)[G][NWMAX],
                                          const half8_t (&af)[G][8], Acc<1> &acc) {
    static_for<0, G>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        u32x4 xb[1][2];
        chain_x(sv, xs, k, n, zero_off, s0 + d, s0 + d < s1, lane, xb);
        mfma_step(af[d], xb, acc);
    });
    for (int g = s0 + G; g < s1; g += G) {
        static_for<0, G>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            u32x4 xb[1][2];
            chain_x(sv, xs, k, n, zero_off, g + d, g + d < s1, lane, xb);
            gemv_step<Codec, 1>(lut, laneoff, words<Codec::NW>(wq[d]), xb, acc);
            if (g + G + d < s1) load_step_w<Codec::NW>(sv, g + G + d, lane, words<Codec::NW>(wq[d]));
            __builtin_amdgcn_sched_barrier(0);
        });
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
    constexpr int G = chain_group<C1, C2>();
    __shared__ __attribute__((aligned(16))) uint32_t lut[C1::LDS_DWORDS];
    __shared__ __attribute__((aligned(16))) unsigned char scratch[SCR];
    __shared__ unsigned wave_ctr;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t laneoff = (uint32_t)(lane & (C1::C - 1)) << 2;
    if (tid == 0) wave_ctr = 0;  // first used after several workgroup barriers
    // phases completed before this launch (block 0 rewrites it at the very end, when every workgroup has long read it:
    // to get there it has seen every workgroup arrive at phase 0; with a single phase nobody waits, so a late reader of
    // the new value is harmless)
    const unsigned seq0 = (unsigned)__builtin_amdgcn_readfirstlane((int)ld_agent(&ws->epoch));
    const void *cur_tab = nullptr;

    for (int ph = 0; ph < nphases; ph++) {
        const TcMultiParams &mp = phases[ph];
        const int gitem = blockIdx.x;
        const int total_items = mp.total_items;
        const bool has = gitem < total_items;  // workgroup-uniform
        QPAL_CSTAMP(0);

        // ---------------------------------------------------------------- A: everything that does not need x
        int j = 0, item_begin = 0;
#pragma unroll
        for (int i = 0; i < kMaxJobs - 1; i++) {
            if (gitem >= mp.item_end[i] && i + 1 < mp.njobs) {
                j = i + 1;
                item_begin = mp.item_end[i];
            }
        }
        const TcParams &p = mp.job[j];
        const int n = p.n, k = p.k;
        const int log2_wpr = p.log2_wpr;
        const int wpr = 1 << log2_wpr;
        const int rloc = wave >> log2_wpr;
        const int wr = wave & (wpr - 1);
        const int log2_rpw = 3 - log2_wpr;
        const int zero_off = n * k;
        float *red = reinterpret_cast<float *>(scratch);                                  // [8][n][32]
        uint16_t *xs = reinterpret_cast<uint16_t *>(scratch + kChainWaves * 32 * 4 * n);  // [n][k] + 32 zero halves
        const int item = gitem - item_begin;
        int rg = item, ks = 0;
        if (p.sk > 1) {
            rg = item / p.sk;
            ks = item - rg * p.sk;
        }
        const int sr = (rg << log2_rpw) + rloc;
        const bool live = has && sr < p.nrows;
        uint32_t wraw = 0;
        if (has && p.wscale && tid < (32 << log2_rpw) && (rg << log2_rpw) + (tid >> 5) < p.nrows)
            wraw = as_global(p.wscale)[((rg << log2_rpw) + (tid >> 5)) * 32 + (tid & 31)];
        const int c = ks * wpr + wr;
        const bool on2 = TWO && c >= p.nc1;
        const int cc = on2 ? c - p.nc1 : c;
        const int base = on2 ? p.base2 : p.base1, rem = on2 ? p.rem2 : p.rem1;
        int s0 = cc * base + (cc < rem ? cc : rem);
        int s1 = s0 + base + (cc < rem ? 1 : 0);
        if (!live) s0 = s1 = 0;
        const StreamView sv1{p.c1 + (long)(live ? sr : 0) * p.nsc1 * 16 * C1::NW, p.nsc1, 0};
        const StreamView sv2{TWO ? p.c2 + (long)(live ? sr : 0) * p.nsc2 * 16 * CB::NW : p.c1, TWO ? p.nsc2 : p.nsc1,
                             p.col2};
        uint32_t wq[G][NWMAX];
        half8_t af[G][8];
        if (on2) chain_request<CB, G, NWMAX>(sv2, s0, s1, lane, wq);
        else chain_request<C1, G, NWMAX>(sv1, s0, s1, lane, wq);
        if (has && p.tab != cur_tab) {  // workgroup-uniform; every wave is past the previous phase's decode (reduce barrier)
            C1::build(lut, p.tab, tid, kChainThreads);
            cur_tab = p.tab;
            __syncthreads();
        }
        QPAL_CSTAMP(1);
        if (on2) chain_ahead<CB, G, NWMAX>(lut, laneoff, sv2, s0, s1, lane, wq, af);
        else chain_ahead<C1, G, NWMAX>(lut, laneoff, sv1, s0, s1, lane, wq, af);
        QPAL_CSTAMP(2);

        // ---------------------------------------------------------------- B: the dependency
        if (ph > 0 && wave == 0) {
            const unsigned done = seq0 + (unsigned)ph;  // phases whose arrivals must be in
            unsigned need = 0;
            if (lane < kChainShards) need = ((gridDim.x + (kChainShards - 1) - lane) / kChainShards) * done;
            unsigned guard = 0;
            for (;;) {
                unsigned cur = need;
                if (lane < kChainShards) cur = ld_agent(&ws->shard[lane][0]);
                if (__all((int)(cur - need) >= 0)) break;
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
            } else if (p.x_fresh) {
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
        if (on2) chain_run<CB, G, NWMAX>(lut, laneoff, sv2, xs, k, n, zero_off, s0, s1, lane, wq, af, acc);
        else chain_run<C1, G, NWMAX>(lut, laneoff, sv1, xs, k, n, zero_off, s0, s1, lane, wq, af, acc);
        QPAL_CSTAMP(5);

        // ---------------------------------------------------------------- E: cross-wave sum, epilogue, arrival
        if (has) {
            const int q = lane >> 4, cidx = lane & 15;
            const int b = cidx >> 1;
            const bool writer = (cidx & 1) == 0 && b < n;
            float *dst = red + ((wave * n + (b < n ? b : 0)) * 32) + 2 * q;
            static_for<0, 4>([&](auto ac) {
                constexpr int a = decltype(ac)::value;
                const float4_t d = acc.v[0][a];
                const float v0 = d[0] + __shfl_xor(d[1], 1, 64);
                const float v1 = d[2] + __shfl_xor(d[3], 1, 64);
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
        // arrival: every wave drains its own stores, the last of the 8 to do so signals for the workgroup
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const unsigned old = __hip_atomic_fetch_add(&wave_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((old & (kChainWaves - 1)) == kChainWaves - 1)
                __hip_atomic_fetch_add(&ws->shard[blockIdx.x & (kChainShards - 1)][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        QPAL_CSTAMP(7);
    }
    if (blockIdx.x == 0 && tid == 0) st_agent(&ws->epoch, seq0 + (unsigned)nphases);
}

}  // namespace qpal
