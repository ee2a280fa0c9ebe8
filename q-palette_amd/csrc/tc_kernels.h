// Fused decode+GEMV and decode-to-fp16 kernels for the "tensor-core tile order" packed formats
// (TCQ trellis streams and VQ/SQ LUT codes), written for gfx950 wave64.
//
// Replaces, for TCQ, kernel_decompress_gemm{,_comb,_combt} / kernel_decompress{,_comb,_combt} and
// load_reg_cs<R> (reference: kernels/tcq-kernels/src/inference.cu:204-1819) and, for VQ/SQ, the
// kernels of kernels/vq-tensor-kernels/src/inference.cu:205-1108.
//
// Mapping chosen for CDNA4 (not the reference's warp32 / mma.m16n8k16 tiling):
//  * Both formats store, per 32x32 "supertile", 32 reference-lanes x R u16 (or u32) words in
//    mma-fragment order.  One wave-lane owns TWO consecutive reference lanes (always a whole number
//    of dwords, 4-byte aligned also for odd R), so a supertile is exactly one 16-lane DPP row and a
//    wave covers 4 supertiles along K per "step": 32 rows x 128 cols, 64 weights per lane, one
//    contiguous 64*NW*4-byte span per wave.
//  * TCQ: the 16 look-ahead bits a lane needs from its successor in the tail-biting stream come from
//    lane+1 of the same DPP row (row_ror:15; the wrap 15->0 is the tail bite): one v_mov_dpp, no LDS.
//    Each 16-bit trellis state is cut straight out of the raw dwords with one v_alignbit_b32 (bits
//    above the window are don't-cares: h = s*(s+1) mod 2^16 only sees the low 16 bits).
//  * Codebooks live in LDS replicated per lane of a 32-lane ds_read_b32 group (bank = lane), so the
//    random gathers are conflict-free; for TCQ the sign flip is folded into the table (index = top
//    S+1 bits of h).  Per 2 weights the VALU work is: alignbit, mad_u24, shift, and_or, dot2.
//  * Integer/bit VALU ops and v_dot2_f32_f16 issue at HALF rate on gfx950 (~4.3 cycles per wave64
//    instruction, measured: perf/valu_rate.hip), which makes this decode VALU-issue-bound, not HBM-bound.
//    The multiply-accumulates therefore go to the otherwise idle matrix pipe: the 8 decoded half2 of a
//    lane are exactly two A fragments of v_mfma_f32_16x16x32_f16; the 16 MFMA columns carry (batch row b,
//    column half u), so batches 1..8 cost the same VALU work.  fp32 accumulation.
#pragma once
#include "qpal_common.h"
#include "wht64.h"
#include "rot_k28.h"

namespace qpal {

struct TcParams {
    // outputs
    float *out;      // gemv: fp32 [n][ldo]
    uint16_t *wout;  // dequant: fp16 [m][ldw]
    long ldo;        // gemv out row stride (floats)
    long ldw;        // dequant row stride (halves)
    // inputs
    const uint32_t *c1;
    const uint32_t *c2;
    const uint16_t *x;  // fp16 [n][k]
    const void *tab;    // codebook: TCQ fp16 pairs [2^S]; LUT fp16 [2^bits][vec]
    int n;              // live batch rows
    int k;              // x row stride (= in_features)
    int nrows;          // supertile rows (m/32)
    int nsc1, nsc2;     // supertile columns of stream 1 / stream 2
    int st1, st2;       // steps (4 supertiles) per supertile row, per stream
    int col2;           // first column of stream 2 (k/2 for combt)
    int sk;             // GEMV: 1, or 2 = "rows are shared between workgroups: the output must start at zero" (plan_launch); lockstep GEMM
                        // and dequant kernels: split factor across workgroups
    int nitems;         // host only: work items of this job (lockstep GEMM / dequant)
    int x_lds;          // 1: stage x[n][k] in LDS (fits beside the codebook image)
    // GEMV launch geometry (round 5): the launch's jobs of one geometry class form ONE row space — job j's rows are the virtual rows
    // [vrow0, vrow0 + nrows) of its class — and the kernel finds a wave's (row, stream, steps) in the launch's table (LaunchPlan)
    int cls;            // geometry class (index into TcMultiParams::plan)
    int vrow0;          // first virtual row
    int kv;                  // TcqAny kernels only: this job's KV (trellis dwords per lane)
    int kv2;                 // TcqAny kernels only: KV of stream 2 of a column-split (combt) job, 0: single stream
    const uint16_t *x_su;    // gemv prologue rotation (x_rot != 0): fp16 [k] sign vector or null
    float x_pre, x_post;     //   staged x = fp16( fp16( H_k (x * su) * x_pre ) * x_post ), x_pre = k^-1/2
    int x_rot;               //   0: x is used as given; else k / 1024 (k in {2048, 4096}), or 28: the 14336-wide rotation of rot_k28.h; needs x_lds
    const uint16_t *x_hadk;  //   x_rot == 28: fp16 [28][28]
    const uint16_t *wscale;  // gemv epilogue: fp16 [m] per-output-row scale or null
    float oscale;            // gemv epilogue: out = acc * wscale[row] * oscale
    unsigned long long *dbg;  // QPAL_STAMPS diagnostic builds only: per-wave s_memtime stamps
    // decoder-block fusion (x_rot jobs of the per-launch kernels): the rotation's input may be the fp32 residual stream
    // with RMSNorm applied on the way in, and the projection may accumulate into its output (the residual add)
    int x_src_f32;           // x points at fp32 [k] (not fp16)
    float x_rms_eps;         // > 0: x <- x * rsqrt(mean(x^2) + eps) (* x_rms_w) before the sign flip (fp32, then ONE fp16 rounding)
    const uint16_t *x_rms_w; // fp16 [k] RMSNorm weight or null
    int accumulate;          // out += result instead of out = result
    const uint16_t *act_su;  // with act_out: fp16 [m / 2] signs (+-1) multiplied into the activation written (the `x * SU_dp` of the NEXT projection's
                             // wrapper, lib/linear/incoherent_linear.py:336: exact, so the consumer's rotation need not read SU)
    uint16_t *act_out;       // ROT kernels, batch 1: the layer's supertile rows alternate up / gate (u0 g0 u1 g1 ...): the epilogue
                             // writes fp16 silu(gate) * up [m / 2] here instead of `out` (rows per workgroup >= 2, no split-K)
};

// One launch's geometry, planned on the host (qpal_capi.hip plan_launch: "tape cut") and read by every wave from the kernel-argument
// block: a GROUP of (1 << lg_g) workgroups owns rg consecutive virtual rows; w[member][wave] says what that wave does.
//   a: [7:0] row inside the group, [8] stream 2, [9] lead = first wave of its row's run of waves in this workgroup (it sums the run's
//      partials and writes the row), [10] the row is shared with another workgroup (atomic add into a zeroed / accumulated output),
//      [15:11] waves in the run (lead only), [16] the wave has steps, [17] lead of an `up` row that also finishes the `gate` row
//      behind it with its upper 32 lanes (SwiGLU epilogue)
//   b: [15:0] first step inside the stream, [31:16] number of steps
constexpr int kPlanMembers = 4;
constexpr int kPlanWaves = 16;
struct WaveEnt {
    uint32_t a, b;
};
struct LaunchPlan {
    int lg_g, rg;
    WaveEnt w[kPlanMembers][kPlanWaves];
};

constexpr int kMaxJobs = 8;

// Several independent GEMVs (same codec, same batch) run by ONE launch: persistent workgroups walk the
// concatenated item lists of the jobs.  Cuts the per-launch fixed cost (kernel-argument fetch, codebook
// image build, launch gap: ~5 us) where the caller has independent projections of one input (q|k|v,
// gate|up).  item_end[j] = items of jobs 0..j.
struct TcMultiParams {
    int njobs;
    int total_items;
    int zero_chunks;   // 16-byte chunks of `zero` this launch fills with zeros (0: none)
    u32x4 *zero;       // buffer pre-zeroed for a later split-K launch on the same stream
    int item_end[kMaxJobs];   // lockstep GEMM kernel (tc_gemm.h): items of jobs 0..j
    // fused GEMV kernel (round 5): table-driven geometry
    int ncls;                 // geometry classes in this launch (1 or 2)
    int items0;               // items (workgroup-sized pieces of work) of class 0; class 1 follows
    int cls_mask;             // bit j: job j belongs to class 1
    int row_end[kMaxJobs];    // end of job j's virtual rows inside its class's row space (padded to whole groups)
    LaunchPlan plan[2];
    TcParams job[kMaxJobs];
};

// Codebook image build shared by the codecs: chunk c (16 bytes = 4 copies of one entry) of the image comes from table entry
// (4 c) >> LOG2C.  U table reads of a thread are issued together, then written.  Round 3 measured the alternatives on one box
// each (perf/ab_wrap.sh; profiles/r03_ab_buildU.txt, r03_ab_image_build_forms.txt): the plain read-then-write loop (U = 1: one L2
// round trip per chunk, 5-8 per launch that builds its image after the argument fetch) looked like 2 us of exposed latency in
// the rotating kernels' prologue — but U = 3 / 8 made the token behind the incoherence wrapper SLOWER (604 -> 597 / 573 tok/s;
// whole-model step 477 -> 474 / 459), delaying the batch by s_sleep made it slower by the delay, and one read per SOURCE entry
// (512 instead of 4 096 reads of the 2 KiB table, its 8 chunk writes in a burst, early staging for every image size) cost the
// wrapper 3.4 % and the plain token 0.6 % (and moved the k / v latency-table entries of the 128 KiB images by -0.9 ... +1.1 us).
// The slow trickle of this loop stays out of the way of the loads and LDS traffic that ARE on the critical path.  U = 1.
#ifndef QPAL_BUILD_U
#define QPAL_BUILD_U 1
#endif
#ifndef QPAL_XPERM_PLAIN  // the permuted x layout in the plain GEMV kernels (see xs_put)
#define QPAL_XPERM_PLAIN 1
#endif
#ifndef QPAL_ROT_BUILD_U  // the image build of the rotating kernels' builder waves (12 of 16 at k = 4096) only
#define QPAL_ROT_BUILD_U 1
#endif
constexpr int kBuildInFlight = QPAL_BUILD_U;
template <int CHUNKS, int LOG2C, int U, class Entry>
__device__ __forceinline__ void build_image(uint32_t *lds, int tid, int nthreads, Entry &&entry) {
    for (int c0 = tid; c0 < CHUNKS; c0 += U * nthreads) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int c = c0 + u * nthreads;
            v[u] = entry(((c < CHUNKS ? c : 0) * 4) >> LOG2C);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int c = c0 + u * nthreads;
            if (c < CHUNKS) reinterpret_cast<u32x4 *>(lds)[c] = u32x4{v[u], v[u], v[u], v[u]};
        }
    }
}

// ================================================================================================
// TCQ codec.  Bit surgery on one lane's KV dwords (reference lanes A = bits [0,16KV), B = [16KV,32KV))
// for tile group G = ksub*2 + msub; state I in 0..7 (0..3 = lane A's j, 4..7 = lane B's j).
// XS = 1 (S = 9 only): the conflict-free image — 32 copies per entry (128 KiB, one bank per lane of a ds_read_b32 group) addressed
// through 2h = h + h, a fourth (full-rate) VALU op per pair.  A loss in the batch-1 GEMV, where the VALU is the scarcer pipe
// (-12 %); used by the lockstep skinny-GEMM kernel up to 4 batch groups, where the LDS is (§4.7).
template <int S, int KV, int XS_ = 0>
struct TcqCodec {
    static_assert(XS_ == 0 || S == 9, "the 32-copy image exists for the 1024-entry codebooks only");
    static constexpr int S_ = S;
    static constexpr int NW = KV;          // dwords per lane per supertile
    static constexpr int L4 = 4 * KV;      // stream bits per reference lane per tile
    static constexpr bool kNeedsNext = true;
    // Codebook image: entry e = top S+1 bits of the hash h (sign flag | S index bits) lives at byte
    // e << (15-S), i.e. exactly where those bits already sit in h, replicated over the 2^(15-S) bytes of
    // its row (16 / 8 / 4 dword copies for S = 9 / 10 / 11; 64 KiB for every S).  The gather address is
    // then ONE v_and_or_b32: (h & mask) | 4*(lane mod copies).  Lanes l and l+16 of a 32-lane ds_read_b32
    // group share a copy, so half of the gathers are 2-way bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.42).  Neither
    // pipe has slack any more: with the pipelined step the VALU and the LDS are each ~73 % busy (545 VALU and 525 LDS
    // cycles of the ~740 a wave-step takes) — the conflict-free 32-copy image (XS = 1) trades a fourth VALU op per pair for them and
    // measured -12 % in round 3 and -18 % with round 5's pipelined step (profiles/r05_ab_conflict_free_image.txt): the steps phase is
    // vector-issue bound at ~80 %, the conflicts overlap it.
    static constexpr int XS = XS_;
    static constexpr int ROWSHIFT = 15 - S + XS;       // log2(bytes per entry row)
    static constexpr int LOG2C = ROWSHIFT - 2;         // copies per entry
    static constexpr int C = 1 << LOG2C;
    static constexpr int LDS_DWORDS = (1 << (S + 1)) * C;  // 16384
    static constexpr uint32_t HMASK = ((1u << (S + 1)) - 1u) << ROWSHIFT;

    template <int G>
    static constexpr int baseA() { return G * L4; }
    template <int G>
    static constexpr int baseB() { return 16 * KV + G * L4; }

    static constexpr int CHUNKS = LDS_DWORDS / 4;  // 16-byte chunks: conflict-free ds_write_b128
    // image entry e (sign flag folded in) from the codebook in memory
    static __device__ __forceinline__ uint32_t entry(const void *tab, int e) {
        return as_global(static_cast<const uint32_t *>(tab))[e & ((1 << S) - 1)] ^ (((uint32_t)e >> S) << 15);
    }
    template <int U = kBuildInFlight>  // table reads in flight per thread
    static __device__ __forceinline__ void build(uint32_t *lds, const void *tab, int tid, int nthreads) {
        build_image<CHUNKS, LOG2C, U>(lds, tid, nthreads, [&](int e) { return entry(tab, e); });
    }
    // Early staging (tc_gemv_kernel): entry(tab, e) == fix(raw(tab, e), e), split so that the loads can be requested at the wave's
    // first instruction with NO arithmetic hanging on them — round 4 found `s_waitcnt vmcnt(0)` + the sign xor right behind the
    // table loads, in front of the kernel-argument loads and the first weight loads: three memory round trips in a row where one
    // was meant (perf/stamps_replay.py: kernel arguments in hand 1.1 us after the wave's entry, first weights requested after that).
    // The loads are issued from inline asm, i.e. OUTSIDE the compiler's wait-count bookkeeping: with loads it knows about it put
    // conservative waits (vmcnt(0) at control-flow joins, register copies at a loop header) in front of the first weight loads
    // whatever the source looked like.  tc_gemv_kernel waits for them by hand (early_landed) where it consumes them.
    static constexpr int RAWN = 1;
    static __device__ __forceinline__ void raw_issue(const void *tab, int e, uint32_t (&r)[RAWN]) {
        const uint32_t off = (uint32_t)(e & ((1 << S) - 1)) * 4u;
        asm volatile("global_load_dword %0, %1, %2" : "=v"(r[0]) : "v"(off), "s"(tab));
    }
    static __device__ __forceinline__ uint32_t fix(const uint32_t (&r)[RAWN], int e) { return r[0] ^ (((uint32_t)e >> S) << 15); }

    // (A << L4) | B in the low 2*L4 bits (KV <= 4 only: 2*L4 <= 32)
    template <int G>
    static __device__ __forceinline__ uint32_t joined(const uint32_t (&w)[NW]) {
        return __builtin_amdgcn_alignbit(ext32<baseA<G>()>(w), ext32<baseB<G>() + L4 - 32>(w), 32 - L4);
    }

    // top-aligned head of this lane's stream for tile G (its first >= 16 bits): what lane-1 needs
    template <int G>
    static __device__ __forceinline__ uint32_t head(const uint32_t (&w)[NW]) {
        if constexpr (KV <= 4) {
            if constexpr (2 * L4 == 32) return joined<G>(w);
            else return joined<G>(w) << (32 - 2 * L4);
        } else if constexpr (baseA<G>() + L4 >= 32) {
            return ext32<baseA<G>() + L4 - 32>(w);
        } else {
            return w[0] << (32 - L4);
        }
    }

    // States that reach into the next reference lane's bits (KV > 4: those with I*KV + 16 > L4) are cut from ONE joined word
    // per chunk: z = [top NMAX bits of the next chunk | low 32-NMAX bits of this chunk], NMAX = 16 - KV = what the LAST state
    // needs from its successor.  The last state is z itself, state I is z >> (3-I)*KV: one half-rate v_alignbit per chunk and
    // full-rate shifts, instead of one v_alignbit per state.
    static constexpr int NMAX = 16 - KV;
    template <int G, bool ISB>
    static __device__ __forceinline__ uint32_t joined_next(const uint32_t (&w)[NW], uint32_t next_head) {
        if constexpr (!ISB) return __builtin_amdgcn_alignbit(extw<baseA<G>(), 32 - NMAX>(w), ext32<baseB<G>() + L4 - 32>(w), 32 - NMAX);
        else return __builtin_amdgcn_alignbit(extw<baseB<G>(), 32 - NMAX>(w), next_head, 32 - NMAX);
    }

    // trellis state in the LOW 16 bits, garbage above
    template <int G, int I>
    static __device__ __forceinline__ uint32_t window(const uint32_t (&w)[NW], uint32_t next_head) {
        if constexpr (KV <= 4) {
            constexpr int n = I * KV + 16 - 2 * L4;  // bits needed from the next lane
            const uint32_t y = joined<G>(w);
            if constexpr (n <= 0) return y >> (-n);
            else return __builtin_amdgcn_alignbit(y, next_head, 32 - n);
        } else {
            constexpr bool ISB = I >= 4;
            constexpr int t0 = (I & 3) * KV;
            constexpr int base = ISB ? baseB<G>() : baseA<G>();
            if constexpr (t0 + 16 <= L4) {
                return extw<base + L4 - 16 - t0, 16>(w);
            } else {
#ifdef QPAL_NO_NARROW_EXT
                constexpr int n = t0 + 16 - L4;
                if constexpr (!ISB) return __builtin_amdgcn_alignbit(ext32<baseA<G>()>(w), ext32<baseB<G>() + L4 - 32>(w), 32 - n);
                else return __builtin_amdgcn_alignbit(ext32<baseB<G>()>(w), next_head, 32 - n);
#else
                const uint32_t z = joined_next<G, ISB>(w, next_head);  // (CSE: one per chunk)
                constexpr int sh = (3 - (I & 3)) * KV;
                static_assert(sh + 16 <= 32, "state window inside the joined word");
                if constexpr (sh == 0) return z;
                else return z >> sh;
#endif
            }
        }
    }

    // byte address, inside the codebook image, of the half2 weight pair of vector (G, I): the VALU half of the decode
    template <int G, int I>
    static __device__ __forceinline__ uint32_t addr(uint32_t laneoff, const uint32_t (&w)[NW], uint32_t next_head) {
        const uint32_t s = window<G, I>(w, next_head);
        uint32_t h;
#if defined(QPAL_KO_HASH)  // power experiment: no multiplier at all (results meaningless)
        asm("v_add_u32 %0, %1, %1" : "=v"(h) : "v"(s));
#elif defined(QPAL_HASH_PK16)  // power experiment (perf/power_probe.hip): two 16 x 16 multipliers instead of one 24 x 24 — the low half is the hash
        asm("v_pk_mad_u16 %0, %1, %1, %1" : "=v"(h) : "v"(s));
#else
        asm("v_mad_u32_u24 %0, %1, %1, %1" : "=v"(h) : "v"(s));  // s*s + s: low 16 bits exact
#endif
        if constexpr (XS == 1) asm("v_add_u32 %0, %1, %1" : "=v"(h) : "v"(h));  // (h << 1 as a shift is a half-rate op)
        return (h & HMASK) | laneoff;
    }
    // half2 weight pair of vector (G, I)
    template <int G, int I>
    static __device__ __forceinline__ uint32_t pair(const uint32_t *lds, uint32_t laneoff, const uint32_t (&w)[NW],
                                                    uint32_t next_head) {
        const uint32_t a = addr<G, I>(laneoff, w, next_head);
#ifdef QPAL_KO_GATHER  // timing experiment (tc_gemm.h): the address instead of the gathered entry
        return a;
#else
        return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lds) + a);
#endif
    }
};

// ================================================================================================
// VQ/SQ LUT codec (tensor-core packed format, lib/quantizer/quant_op.py:101-162).
//   VEC == 2 : `BITS` dwords per lane; code j of reference lane r, tile G at LE bit r*16B + G*4B + j*B
//   VEC == 1 : 2*BITS dwords per lane; code q = 2j+e at LE bit r*32B + G*8B + q*B.  For BITS <= 6 the
//              two codes of a half2 are adjacent, so one gather in a pair table of 2^(2B) half2
//              entries decodes both (the reference does this for BITS <= 4 only: LOAD_TYPE::DUP).
template <int BITS, int VEC>
struct LutCodec {
    static_assert(VEC == 1 || VEC == 2, "tensor-core format has vec_sz 1 or 2");
    static constexpr bool PAIR = (VEC == 1 && BITS <= 6);
    static constexpr int NW = (VEC == 2) ? BITS : 2 * BITS;
    static constexpr bool kNeedsNext = false;
    static constexpr int IDXBITS = (VEC == 2) ? BITS : (PAIR ? 2 * BITS : BITS);
    static constexpr int LOG2C = (15 - IDXBITS) < 5 ? (15 - IDXBITS) : 5;  // <= 128 KiB
    static constexpr int C = 1 << LOG2C;
    static constexpr int LDS_DWORDS = (1 << IDXBITS) * C;
    static_assert(LOG2C >= 2, "table build writes 4 copies per 16-byte chunk");

    static constexpr int CHUNKS = LDS_DWORDS / 4;
    static __device__ __forceinline__ uint32_t entry(const void *tab, int e) {
        const gptr<const uint16_t> l16 = as_global(static_cast<const uint16_t *>(tab));
        const gptr<const uint32_t> l32 = as_global(static_cast<const uint32_t *>(tab));
        if constexpr (VEC == 2) return l32[e];
        else if constexpr (PAIR) return (uint32_t)l16[e & ((1 << BITS) - 1)] | ((uint32_t)l16[e >> BITS] << 16);
        else return l16[e];
    }
    template <int U = kBuildInFlight>  // table reads in flight per thread
    static __device__ __forceinline__ void build(uint32_t *lds, const void *tab, int tid, int nthreads) {
        build_image<CHUNKS, LOG2C, U>(lds, tid, nthreads, [&](int e) { return entry(tab, e); });
    }
    // early staging: loads and the arithmetic on them apart (see TcqCodec::raw)
    static constexpr int RAWN = PAIR ? 2 : 1;
    static __device__ __forceinline__ void raw_issue(const void *tab, int e, uint32_t (&r)[RAWN]) {
        if constexpr (VEC == 2) {
            asm volatile("global_load_dword %0, %1, %2" : "=v"(r[0]) : "v"((uint32_t)e * 4u), "s"(tab));
        } else if constexpr (PAIR) {
            asm volatile("global_load_ushort %0, %1, %2" : "=v"(r[0]) : "v"((uint32_t)(e & ((1 << BITS) - 1)) * 2u), "s"(tab));
            asm volatile("global_load_ushort %0, %1, %2" : "=v"(r[1]) : "v"((uint32_t)(e >> BITS) * 2u), "s"(tab));
        } else {
            asm volatile("global_load_ushort %0, %1, %2" : "=v"(r[0]) : "v"((uint32_t)e * 2u), "s"(tab));
        }
    }
    static __device__ __forceinline__ uint32_t fix(const uint32_t (&r)[RAWN], int) {
        if constexpr (PAIR) return r[0] | (r[1] << 16);
        else return r[0];
    }

    template <int G>
    static __device__ __forceinline__ uint32_t head(const uint32_t (&)[NW]) { return 0u; }

    template <int POS, int NB_>
    static __device__ __forceinline__ uint32_t gather_addr(uint32_t laneoff, const uint32_t (&w)[NW]) {
        // NB_ index bits at LE bit POS; stay inside one dword when possible (single v_bfe_u32)
        uint32_t e;
        if constexpr ((POS & 31) + NB_ <= 32) e = __builtin_amdgcn_ubfe(w[POS >> 5], POS & 31, NB_);
        else e = __builtin_amdgcn_ubfe(ext32<POS>(w), 0, NB_);
        return (e << (LOG2C + 2)) | laneoff;
    }
    template <int POS, int NB_>
    static __device__ __forceinline__ uint32_t gather(const uint32_t *lds, uint32_t laneoff, const uint32_t (&w)[NW]) {
        return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lds) + gather_addr<POS, NB_>(laneoff, w));
    }
    // byte address of the pair (G, I) inside the image, for the codes that decode a pair with ONE gather (gemv_step_pipe)
    template <int G, int I>
    static __device__ __forceinline__ uint32_t addr(uint32_t laneoff, const uint32_t (&w)[NW], uint32_t)
        requires(VEC == 2 || PAIR)
    {
        constexpr int r = I >> 2, j = I & 3;
        if constexpr (VEC == 2) return gather_addr<r * 16 * BITS + G * 4 * BITS + j * BITS, BITS>(laneoff, w);
        else return gather_addr<r * 32 * BITS + G * 8 * BITS + 2 * j * BITS, 2 * BITS>(laneoff, w);
    }

    template <int G, int I>
    static __device__ __forceinline__ uint32_t pair(const uint32_t *lds, uint32_t laneoff, const uint32_t (&w)[NW],
                                                    uint32_t) {
        constexpr int r = I >> 2, j = I & 3;
        if constexpr (VEC == 2) {
            return gather<r * 16 * BITS + G * 4 * BITS + j * BITS, BITS>(lds, laneoff, w);
        } else if constexpr (PAIR) {
            return gather<r * 32 * BITS + G * 8 * BITS + 2 * j * BITS, 2 * BITS>(lds, laneoff, w);
        } else {
            constexpr int pos = r * 32 * BITS + G * 8 * BITS + 2 * j * BITS;
            const uint32_t lo = gather<pos, BITS>(lds, laneoff, w);
            const uint32_t hi = gather<pos + BITS, BITS>(lds, laneoff, w);
            return lo | (hi << 16);
        }
    }
};

// ================================================================================================
// One stream of one supertile row as seen by one wave.
struct StreamView {
    const uint32_t *base;  // first dword of this supertile row of the stream
    int nsc;               // supertile columns in the stream
    int col0;              // first x column of the stream
};

#ifdef QPAL_STAMPS
#define QPAL_STAMP(i) do { if (p.dbg && lane == 0 && gitem == (int)blockIdx.x) p.dbg[((long)blockIdx.x * 16 + wave) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define QPAL_STAMP(i) do { } while (0)
#endif

// Experiment (make VARIANT=w8 EXTRA=-DQPAL_W8): TWO 8-wave workgroups per CU instead of one 16-wave one for the
// batch <= 16 kernels — 64 KiB image + 15 KiB scratch each (x fits only for n * k <= ~7000: the k = 4096 launches at batch 1),
// 128 VGPRs as before (4 waves per SIMD).  No fused rotation in this build.
#if defined(QPAL_W8) && !defined(QPAL_W8_ONE)  // (QPAL_W8_ONE: ONE 8-wave workgroup per CU — half the waves to dispatch — keeps the full scratch)
constexpr int kScratchBytes = 15 * 1024;
#else
constexpr int kScratchBytes = 31 * 1024;  // LDS left beside the 128 KiB codebook image: reduction buffer + x
#endif

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));

// Weights of step `step` (4 supertiles).  Supertile columns past the end of the stream re-read the last
// valid one (uniform load count; neutralised by zero activations).
template <int NW>
__device__ __forceinline__ void load_step_w(const StreamView &sv, int step, int lane, uint32_t (&w)[NW]) {
    int sc = step * 4 + (lane >> 4);
    sc = sc < sv.nsc ? sc : sv.nsc - 1;
    // 32-bit dword offset from the (wave-uniform) row base: scalar base + vector offset addressing, no 64-bit math
    const uint32_t off = (uint32_t)sc * (16u * NW) + (uint32_t)(lane & 15) * NW;
    load_words_nt<NW>(sv.base + off, w);
}

// ---- buffer-addressed weight / tile loads (round 5; first in the lockstep kernels, tc_gemm.h, then here): a descriptor per wave-item,
// one 32-bit add per load and step instead of 64-bit address arithmetic, the descriptor's range check instead of clamps.
using buf_rsrc_t = __amdgpu_buffer_rsrc_t;
template <class T>
__device__ __forceinline__ buf_rsrc_t gemm_rsrc(const T *base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(base), (short)0, bytes, 0x00020000);
}
constexpr uint32_t kBufDead = 0x7fffff00u;  // a vector offset beyond every descriptor's range (num_records < 2^31)
// NW consecutive dwords at byte offset voff of the descriptor, non-temporal (streamed-once weights)
template <int NW>
__device__ __forceinline__ void buf_load_words_nt(buf_rsrc_t rs, uint32_t voff, uint32_t (&w)[NW]) {
    constexpr int Q = NW / 4, R = NW % 4;
    static_for<0, Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16 * q, 0, 2);
        w[4 * q + 0] = v.x;
        w[4 * q + 1] = v.y;
        w[4 * q + 2] = v.z;
        w[4 * q + 3] = v.w;
    });
    if constexpr (R == 3) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b96(rs, voff + 16 * Q, 0, 2);
        w[4 * Q + 0] = v[0];
        w[4 * Q + 1] = v[1];
        w[4 * Q + 2] = v[2];
    } else if constexpr (R == 2) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + 16 * Q, 0, 2);
        w[4 * Q + 0] = v[0];
        w[4 * Q + 1] = v[1];
    } else if constexpr (R == 1) {
        w[4 * Q] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff + 16 * Q, 0, 2);
    }
}
// this lane's byte offset inside its row of a stream, for step 0: supertile lane >> 4, 16 x NW dwords per supertile
template <int NW>
__device__ __forceinline__ uint32_t gemm_w_lane_off(int lane) { return ((uint32_t)(lane >> 4) * 16u * NW + (uint32_t)(lane & 15) * NW) * 4u; }
template <int NW>
constexpr uint32_t gemm_w_step_bytes() { return 4u * 16u * NW * 4u; }  // four supertiles per step

// weights of step `step` through a descriptor of the wave's row of the stream (num_records = nsc * 64 * NW bytes: supertile columns
// past the end read zeros — neutralised by zero activations like the re-read of load_step_w)
template <int NW>
__device__ __forceinline__ void load_step_w_buf(buf_rsrc_t rs, int step, int lane, uint32_t (&w)[NW]) {
    buf_load_words_nt<NW>(rs, gemm_w_lane_off<NW>(lane) + (uint32_t)step * gemm_w_step_bytes<NW>(), w);
}

// Accumulators of one wave: NBG batch groups (8 batch rows each) x 4 row groups (msub*2 + jl).
template <int NBG>
struct Acc {
    float4_t v[NBG][4];
};

// Layout of the staged activations (batch <= 8, XLDS): inside every block of 32 halves (one supertile column) the eight
// 4-half pieces p0..p7 are stored as [p0 p2 p4 p6 | p1 p3 p5 p7] — the lane of column half u reads 16 contiguous halves at
// 16 u.  xs_put stores the 16-byte chunk of halves [i, i + 8) (i a multiple of 8); xs_index maps one half.
// XP (the plain kernels: ROT 0 / 3): the permuted layout; the rotating kernels keep the plain one (their second stage writes single
// halves through xs_index: the permutation's index arithmetic costs them more than the wider reads save — round 4,
// profiles/r04_ab_xperm.txt: plain token +0.4…+0.9 %, wrapper token -1.4 % with the permutation everywhere).
template <bool XP>
__device__ __forceinline__ void xs_put(uint16_t *xs, int i, u32x4 v) {
    if constexpr (XP) {
        uint16_t *d = xs + (i & ~31) + ((i >> 3) & 3) * 4;
        *reinterpret_cast<u32x2 *>(d) = u32x2{v.x, v.y};
        *reinterpret_cast<u32x2 *>(d + 16) = u32x2{v.z, v.w};
    } else {
        *reinterpret_cast<u32x4 *>(xs + i) = v;
    }
}
template <bool XP>
__device__ __forceinline__ int xs_index(int i) {
    if constexpr (XP) return (i & ~31) | (((i >> 2) & 1) << 4) | (((i >> 3) & 3) << 2) | (i & 3);
    else return i;
}

// The MFMA's 16 columns are 8 batch rows x 2 column halves; at batch n < 8 the columns of the batch rows that do not exist used to
// carry COPIES of row n - 1 (results never stored).  The matrix pipe multiplies whatever it is given: with zeros in those columns
// (the staged x's zero pad, one broadcast LDS read) its switching power goes down — measured with perf/power_probe.hip: the 8 matrix
// instructions of a step are ~200 W of the ~1 100 W the decode draws, and the token runs into the card's power limit (shader clock 2.15
// instead of 2.4 GHz while tokens replay back to back, profiles/r05_power_and_clock.txt).
#ifndef QPAL_ZERO_B
#define QPAL_ZERO_B 1
#endif
// MFMA B operand of a step.  Lane (kb = lane>>4, c = lane&15) supplies, for batch row b = 8*grp + (c>>1) and
// column half u = c&1, the 8 activations  x[b][col0 + 32*sc + 16*ksub + 8*jh + 4*u + 0..3], jh = 0,1
// (order matches the A fragment: jh-major, then the reference lanes A|B, then the element of the pair).
// XLDS (batch <= 8 only): x was staged in LDS as [n][k] followed by a 64-byte zero pad (dead supertile
// columns of a partial last step read the pad); otherwise x is read from global memory (L2) and dead lanes
// are zeroed by select.
// XLDS: 0 x from global memory, 1 staged in LDS, 2 staged in LDS in the permuted layout (xs_put<true>)
template <int XLDS, int NBG>
__device__ __forceinline__ void load_step_x(const StreamView &sv, const uint16_t *xg, const uint16_t *xs, int k, int n,
                                            int zero_off, int step, int lane, u32x4 (&xb)[NBG][2]) {
    const int sc = step * 4 + (lane >> 4);
    const bool live = sc < sv.nsc;
    const int c = lane & 15;
    static_for<0, NBG>([&](auto gc) {
        constexpr int grp = decltype(gc)::value;
        int b = 8 * grp + (c >> 1);
        b = b < n ? b : n - 1;
        const int off = b * k + sv.col0 + sc * 32 + 4 * (c & 1);
        [[maybe_unused]] const bool real = QPAL_ZERO_B == 0 || 8 * grp + (c >> 1) < n;  // (see QPAL_ZERO_B)
        if constexpr (XLDS == 2) {
            // staged x is PERMUTED inside every 32-half block (xs_put<true>): the four 8-byte pieces a lane needs are
            // contiguous, so the B operand of a step is two ds_read_b128 (4 LDS cycles each) instead of two ds_read2_b64 (8)
            const uint16_t *row = xs + (live && real ? off + 12 * (c & 1) : zero_off);
            xb[grp][0] = *reinterpret_cast<const u32x4 *>(row);
            xb[grp][1] = *reinterpret_cast<const u32x4 *>(row + 8);
        } else if constexpr (XLDS == 1) {
            const uint16_t *row = xs + (live && real ? off : zero_off);
#pragma unroll
            for (int ksub = 0; ksub < 2; ksub++) {
                const u32x2 lo = *reinterpret_cast<const u32x2 *>(row + 16 * ksub);
                const u32x2 hi = *reinterpret_cast<const u32x2 *>(row + 16 * ksub + 8);
                xb[grp][ksub] = u32x4{lo.x, lo.y, hi.x, hi.y};
            }
        } else {
            const uint16_t *row = xg + (live ? off : 0);
#pragma unroll
            for (int ksub = 0; ksub < 2; ksub++) {
                const u32x2 lo = *reinterpret_cast<const u32x2 *>(row + 16 * ksub);
                const u32x2 hi = *reinterpret_cast<const u32x2 *>(row + 16 * ksub + 8);
                xb[grp][ksub] = live ? u32x4{lo.x, lo.y, hi.x, hi.y} : u32x4{0u, 0u, 0u, 0u};
            }
        }
    });
}

// One step = 8 MFMAs (16x16x32 f16) per batch group: per tile group (ksub, msub) the 8 decoded half2 of a
// lane form the A fragments of two MFMAs (jl = 0, 1: tile rows p>>1 and (p>>1)+8).  MFMA row i = lane&15 = p
// is the VIRTUAL row (tile row p>>1, column half u = p&1); MFMA column j = 2b+u; D[i][j] is a valid partial
// product only where the two u agree.  acc.v[grp][msub*2+jl] accumulates over ksub, steps and supertiles.
// The decode (all of the VALU work) is shared by the batch groups: batches 9..16 cost 8 more MFMAs per step.
template <class Codec, int NBG>
__device__ __forceinline__ void gemv_step(const uint32_t *lut, uint32_t laneoff, const uint32_t (&w)[Codec::NW],
                                          const u32x4 (&xb)[NBG][2], Acc<NBG> &acc) {
    static_for<0, 4>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int ksub = g >> 1, msub = g & 1;
        uint32_t nh = 0u;
        if constexpr (Codec::kNeedsNext) nh = row16_next(Codec::template head<g>(w));
        static_for<0, 2>([&](auto jc) {
            constexpr int jl = decltype(jc)::value;
            // fragment order (jh, isB): i = jl + 2*jh + 4*isB
            const u32x4 a{Codec::template pair<g, jl>(lut, laneoff, w, nh),
                          Codec::template pair<g, jl + 4>(lut, laneoff, w, nh),
                          Codec::template pair<g, jl + 2>(lut, laneoff, w, nh),
                          Codec::template pair<g, jl + 6>(lut, laneoff, w, nh)};
            const half8_t afrag = __builtin_bit_cast(half8_t, a);
            static_for<0, NBG>([&](auto bc) {
                constexpr int grp = decltype(bc)::value;
                acc.v[grp][msub * 2 + jl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                    afrag, __builtin_bit_cast(half8_t, xb[grp][ksub]), acc.v[grp][msub * 2 + jl], 0, 0, 0);
            });
        });
    });
}

// Software-pipelined form of the step (QPAL_GEMV_PIPE, batch group count 1, codecs with `addr`): the compiler's own order per
// tile group is  8 addresses (VALU) -> 8 gathers -> wait -> 2 MFMAs,  i.e. a wave has no VALU work while its gathers are in flight
// and no gathers in flight while it computes addresses; here the addresses of group g + 1 are computed between the gathers of
// group g and the MFMAs that consume them (order pinned by sched_barrier fences).
#ifndef QPAL_GEMV_PIPE
#define QPAL_GEMV_PIPE 1
#endif
template <class Codec>
constexpr bool has_addr_v = requires(const uint32_t (&w)[Codec::NW]) { Codec::template addr<0, 0>(0u, w, 0u); };
template <class Codec>
__device__ __forceinline__ void gemv_step_pipe(const uint32_t *lut, uint32_t laneoff, const uint32_t (&w)[Codec::NW],
                                               const u32x4 (&xb)[1][2], Acc<1> &acc) {
    const char *l8 = reinterpret_cast<const char *>(lut);
    uint32_t ac[8], an[8];
    auto addrs = [&](auto gc, uint32_t(&a)[8]) {
        constexpr int g = decltype(gc)::value;
        uint32_t nh = 0u;
        if constexpr (Codec::kNeedsNext) nh = row16_next(Codec::template head<g>(w));
        static_for<0, 8>([&](auto ic) { a[decltype(ic)::value] = Codec::template addr<g, decltype(ic)::value>(laneoff, w, nh); });
    };
    addrs(std::integral_constant<int, 0>{}, ac);
    static_for<0, 4>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int ksub = g >> 1, msub = g & 1;
        uint32_t d[8];
#ifdef QPAL_KO_GATHER  // power / timing experiments (perf/power_probe.hip): the address instead of the gathered entry
        static_for<0, 8>([&](auto ic) { d[decltype(ic)::value] = ac[decltype(ic)::value]; });
#else
        static_for<0, 8>([&](auto ic) { d[decltype(ic)::value] = *reinterpret_cast<const uint32_t *>(l8 + ac[decltype(ic)::value]); });
#endif
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g < 3) addrs(std::integral_constant<int, g + 1>{}, an);
        __builtin_amdgcn_sched_barrier(0);
#ifdef QPAL_PROBE_MFMA444  // power / timing probe only (perf/power_probe.hip; the results are NOT the GEMV's): the step's MACs as 16 x 4x4x4
        {                      // (16 blocks) matrix instructions — a quarter of the products of the 16 x 16 x 32 shape, twice the instructions
            typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
            typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
            const u32x4 xq = xb[0][ksub];
            const half4_t b0 = __builtin_bit_cast(half4_t, u32x2_t{xq.x, xq.y}), b1 = __builtin_bit_cast(half4_t, u32x2_t{xq.z, xq.w});
            acc.v[0][msub * 2 + 0] = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(half4_t, u32x2_t{d[0], d[4]}), b0, acc.v[0][msub * 2 + 0], 0, 0, 0);
            acc.v[0][msub * 2 + 0] = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(half4_t, u32x2_t{d[2], d[6]}), b1, acc.v[0][msub * 2 + 0], 0, 0, 0);
            acc.v[0][msub * 2 + 1] = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(half4_t, u32x2_t{d[1], d[5]}), b0, acc.v[0][msub * 2 + 1], 0, 0, 0);
            acc.v[0][msub * 2 + 1] = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(half4_t, u32x2_t{d[3], d[7]}), b1, acc.v[0][msub * 2 + 1], 0, 0, 0);
        }
#elif defined(QPAL_KO_MFMA)  // (the decoded pairs consumed by nothing: no matrix instruction, no other instruction in their place)
        asm volatile("" ::"v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(xb[0][ksub]));
#else
        // fragment order (jh, isB): i = jl + 2*jh + 4*isB
        acc.v[0][msub * 2 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, u32x4{d[0], d[4], d[2], d[6]}),
                                                                        __builtin_bit_cast(half8_t, xb[0][ksub]), acc.v[0][msub * 2 + 0], 0, 0, 0);
        acc.v[0][msub * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, u32x4{d[1], d[5], d[3], d[7]}),
                                                                        __builtin_bit_cast(half8_t, xb[0][ksub]), acc.v[0][msub * 2 + 1], 0, 0, 0);
#endif
        if constexpr (g < 3) static_for<0, 8>([&](auto ic) { ac[decltype(ic)::value] = an[decltype(ic)::value]; });
    });
}
template <class Codec, int NBG>
__device__ __forceinline__ void gemv_step_any(const uint32_t *lut, uint32_t laneoff, const uint32_t (&w)[Codec::NW],
                                              const u32x4 (&xb)[NBG][2], Acc<NBG> &acc) {
    if constexpr (QPAL_GEMV_PIPE != 0 && NBG == 1 && has_addr_v<Codec>) gemv_step_pipe<Codec>(lut, laneoff, w, xb, acc);
    else gemv_step<Codec, NBG>(lut, laneoff, w, xb, acc);
}

// steps [s0, s1) of one stream; `w` already holds step s0 (loaded before the codebook image was built).
// Two register sets ping-pong (loop unrolled by 2) so the one-step-ahead prefetch costs no copies.
// Deeper prefetch, measured and not kept.  Round 3: two or three steps in flight with every wave's steps padded to whole rounds of
// three: 10-19 % slower (profiles/r03_ab_ring.txt).  Round 4 (profiles/r04_ab_prologue.txt (4), r04_ab_depth2_long_only.txt): three
// register sets, TWO steps ahead, nothing padded, steps past the end requesting a 16-byte dummy so that no load sits under a
// branch — every launch kind 0.4-0.7 us slower (124 instead of 91 VGPRs, 1.5 x the code, the compiler's wait in the first of
// the three bodies still covers the newer step); restricted to chunks of >= 3 (>= 5) steps, i.e. gate | up and down only:
// gate | up +-0, down +0.9 us (+0.1).  The stamps' 0.36 us per step inside gate | up against 0.31 on register-resident words
// is therefore not load latency a deeper register prefetch recovers.
template <class Codec, int XLDS, int NBG>
__device__ __forceinline__ void gemv_run(uint32_t (&w)[Codec::NW], const uint32_t *lut, uint32_t laneoff,
                                         const StreamView &sv, buf_rsrc_t rs, const uint16_t *xg, const uint16_t *xs, int k, int n,
                                         int zero_off, int s0, int s1, int lane, Acc<NBG> &acc) {
    uint32_t wb[Codec::NW];
    // Permuted staged x (XLDS == 2, batch 1..8 in one group): the lane's B operand of step s sits 256 bytes behind that of step s - 1, and
    // the lane's supertile column is past the end of the stream from ONE step on — a per-lane address and a per-lane step bound
    // worked out once per piece leave three vector instructions per step (add, compare, select) where the general form below spent
    // eight (round 5: the steps are vector-issue bound).
    [[maybe_unused]] const char *xl = nullptr, *xz = nullptr;
    [[maybe_unused]] int dead = 0;
    if constexpr (XLDS == 2 && NBG == 1) {
        const int c = lane & 15;
        int b = c >> 1;
        b = b < n ? b : n - 1;
        xl = reinterpret_cast<const char *>(xs) + (b * k + sv.col0 + (lane >> 4) * 32 + 16 * (c & 1)) * 2;
        xz = reinterpret_cast<const char *>(xs) + zero_off * 2;
        dead = (sv.nsc - (lane >> 4) + 3) >> 2;  // first step whose supertile column 4 s + (lane >> 4) is >= nsc
        if (QPAL_ZERO_B != 0 && (c >> 1) >= n) dead = 0;  // a batch row that does not exist: zeros from the first step on (see QPAL_ZERO_B)
    }
    auto step_x = [&](int s, u32x4(&xb)[NBG][2]) {
        if constexpr (XLDS == 2 && NBG == 1) {
            const char *row = s < dead ? xl + s * 256 : xz;
            xb[0][0] = *reinterpret_cast<const u32x4 *>(row);
            xb[0][1] = *reinterpret_cast<const u32x4 *>(row + 16);
        } else {
            load_step_x<XLDS, NBG>(sv, xg, xs, k, n, zero_off, s, lane, xb);
        }
    };
    for (int s = s0; s < s1; s += 2) {
        {
            const int sn = s + 1 < s1 ? s + 1 : s;  // last step re-reads itself (L1 hit, unused)
            load_step_w_buf<Codec::NW>(rs, sn, lane, wb);
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this step's decode (hipcc sinks it otherwise)
            u32x4 xb[NBG][2];
            step_x(s, xb);
            gemv_step_any<Codec, NBG>(lut, laneoff, w, xb, acc);
        }
        if (s + 1 < s1) {
            const int sn = s + 2 < s1 ? s + 2 : s + 1;
            load_step_w_buf<Codec::NW>(rs, sn, lane, w);
            __builtin_amdgcn_sched_barrier(0);
            u32x4 xb[NBG][2];
            step_x(s + 1, xb);
            gemv_step_any<Codec, NBG>(lut, laneoff, wb, xb, acc);
        }
    }
}

// Waves per workgroup of the fused kernel: 16 (4 per SIMD, 128 VGPRs) up to batch 16; 8 (2 per SIMD, 256 VGPRs: room for
// the accumulators and activations of 4 / 8 batch groups) for batches 17..64, where every decoded step feeds 32 / 64 MFMAs
#ifdef QPAL_W8
template <int NBG>
constexpr int gemv_waves() { return 8; }
#define QPAL_GEMV_BOUNDS(NBG) __launch_bounds__(512, (NBG) <= 2 ? 4 : 2)
#else
template <int NBG>
constexpr int gemv_waves() { return NBG >= 4 ? 8 : 16; }
#define QPAL_GEMV_BOUNDS(NBG) __launch_bounds__(64 * gemv_waves<NBG>())
#endif
// LDS beside the 64 KiB codebook image: reduction buffer [waves][n][32] fp32 (+ x for batch <= 8)
template <int NBG>
constexpr int scratch_bytes() { return NBG == 1 ? kScratchBytes : gemv_waves<NBG>() * 32 * 4 * 8 * NBG; }

// Mixed-precision launches: everything about a TCQ codec except the bit surgery depends on the codebook size S only (the
// LDS image, its address mask, the hash), so single-stream layers of one S but DIFFERENT KV — q, k and v of a mixed-scheme
// model each have their own bit width — can share one launch and one image.  TcqAny<S> stands for "KV is a per-job
// runtime value": the kernel switches to the matching decode loop once per work item (workgroup-uniform).
template <int S>
struct TcqAny : TcqCodec<S, (S == 9 ? 8 : 10)> {
    static constexpr bool kAny = true;
    // KV range per S as the quantizer strings produce it: tlut_bits = 9 for KV <= 8, KV + 1 above (mem_op.py:274)
    // (the reference's op table also lists (9, 9) and (9, 10); those stay with the per-KV kernels)
    static constexpr int kMinKV = S == 9 ? 2 : S == 10 ? 8 : 9;
    static constexpr int kMaxKV = S == 9 ? 8 : 10;
};
template <class C>
constexpr bool is_any_v = requires { C::kAny; };

// f(std::integral_constant<int, KV>) for the runtime kv (uniform across the workgroup)
template <int S, class F>
__device__ __forceinline__ void dispatch_kv(int kv, F &&f) {
    static_for<TcqAny<S>::kMinKV, TcqAny<S>::kMaxKV + 1>([&](auto kc) {
        if (kv == decltype(kc)::value) f(kc);
    });
}

// ------------------------------------------------------------------------------------------------
// Fused decode + GEMV / skinny GEMM, 1 <= n <= 8*NBG.  1024 threads (16 waves, 4 per SIMD), one workgroup per
// CU (LDS-bound).  C2 == void: single stream.  Otherwise combt (columns [0,col2) from c1 via C1, the rest from
// c2 via C2); both codecs share one codebook image.  A wave's chunk of steps never straddles the two streams.
// When they fit beside the codebook image (batch <= 8) the activations are staged once per workgroup in LDS.
// All index arithmetic is shift/compare: the host passes the chunk partition (base/rem) precomputed.
// ROT: instantiation that can rotate x while staging it (x_rot jobs); plain launches run the ROT = false kernels,
// which do not carry that code (it costs ~3 % of a plain token when merely present: measured)
// The leading scalar arguments are PRELOADED into SGPRs by the dispatcher (-mllvm -amdgpu-kernarg-preload-count, Makefile):
// when every job of the launch reads the same activations and codebook (`early`), their loads are issued before the
// kernel-argument block `mp` has even arrived (its fetch is a memory round trip on the critical path of a short kernel).
#ifdef QPAL_W8
constexpr int kEarlyXChunks = 1;
#else
constexpr int kEarlyXChunks = 2;
#endif
struct TcEarly {
    const uint16_t *x;  // shared by all jobs, staged in LDS
    const void *tab;    // shared codebook
    int n, k;           // x is [n][k]
    int on;             // 0: jobs differ (or x does not fit LDS): everything comes from `mp` as before.  Bit 0: plain early staging.
                        // Rotating launches (round 4): bit 1: every job rotates the same x (x_rot in {2, 4}: k = 2048 / 4096) with the
                        // same sign vector / RMSNorm weight and shares the codebook — the rotation's inputs and the image's table
                        // entries are requested at the wave's first instruction; bit 2: x is fp32; bit 3: RMSNorm in front of the rotation; bits 4..6: x_rot
    const uint16_t *su;  // rotating launches: sign vector (or null), RMSNorm weight (or null; the 14336-wide rotation: its hadK factor)
    const uint16_t *rw;
    float pre, post;     // the 14336-wide rotation: x_pre, x_post
    int ie;              // != 0 (bit 30): row_end[0..2] of the launch's jobs, 10 bits each (0x3ff: no such boundary), and `on` carries the
                         // plan's header (bits 16..17 lg_g, 18..24 rg) — a wave knows its table entry's address and its group's job before
                         // the kernel-argument block has arrived, and requests both in the first scalar-load round trip (one geometry
                         // class, <= 4 jobs; 0: it reads the plan header and scans the row table first)
};

// host: the launch qualifies when x is staged in LDS and every job has the same x, codebook and batch
inline TcEarly early_args(const TcMultiParams &mp, int grid) {
    const TcParams &a = mp.job[0];
    // (x and its 32-half zero pad must fit the chunks the threads hold)
    TcEarly e{a.x, a.tab, a.n, a.k, a.x_lds && !a.x_rot && a.n <= 8 && a.n * a.k + 32 <= kEarlyXChunks * 64 * gemv_waves<1>() * 8 ? 1 : 0,
              a.x_su, a.x_rms_w, a.x_pre, a.x_post};
    if (a.x_lds && a.x_rot == kK28 && a.n == 1 && a.x_hadk && !a.x_src_f32 && !(a.x_rms_eps > 0.f)) {  // bit 1 with x_rot = 28 in bits 4..9
        e.on = 2 | (kK28 << 4);
        e.rw = a.x_hadk;
    }
    if (a.x_lds && (a.x_rot == 2 || a.x_rot == 4) && a.n == 1) e.on = 2 | (a.x_src_f32 ? 4 : 0) | (a.x_rms_eps > 0.f ? 8 : 0) | (a.x_rot << 4);
    for (int j = 1; j < mp.njobs; j++) {
        const TcParams &b = mp.job[j];
        if (b.x != a.x || b.tab != a.tab || b.n != a.n || b.k != a.k || !b.x_lds || b.x_rot != a.x_rot || b.x_su != a.x_su ||
            b.x_rms_w != a.x_rms_w || b.x_src_f32 != a.x_src_f32 || (b.x_rms_eps > 0.f) != (a.x_rms_eps > 0.f) || b.x_hadk != a.x_hadk ||
            b.x_pre != a.x_pre || b.x_post != a.x_post)
            e.on = 0;
    }
    // Fast path of the kernel's row lookup: one geometry class, <= 4 jobs, row ends below 1023 — the plan's header and the jobs' row
    // ends travel as preloaded arguments, so the first kernel-argument round trip can fetch the wave's table entry AND its job.
    e.ie = 0;
    if (mp.ncls == 1 && mp.njobs <= 4 && mp.row_end[mp.njobs - 1] < 0x3ff && mp.plan[0].rg < 128) {
        e.ie = 1 << 30;
        for (int j = 0; j < 3; j++) e.ie |= (j < mp.njobs - 1 ? mp.row_end[j] : 0x3ff) << (10 * j);
        e.on |= (mp.plan[0].lg_g << 16) | (mp.plan[0].rg << 18);
    }
    // bit 30: ONE round — every workgroup has exactly one item (grid == total_items, the usual case).  The kernel then neither reads
    // gridDim.x (a hidden kernel argument: one more scalar-load round trip behind the last barrier of EVERY workgroup) nor waits at
    // the barrier that only protects the reduction buffer against a next item.
    if (grid >= mp.total_items) e.on |= 1 << 30;
    return e;
}

// Stage 1 of the rotation fused into the GEMV staging, on inputs that are already in registers (rq[kc]: x — two chunks when fp32 —,
// RMSNorm weight, sign vector of this wave's row tile and 32-column half kc): conversion (RMSNorm fused: the transform is linear,
// so the norm's scalar 1/rms is applied AFTER it — stage 1 rotates x * w * 2^-6, the power of two keeps fp16 clear of overflow on
// residual-stream outliers — while the same lanes sum the squares of what they hold), sign flip, the four column tiles of
// x_tile . H_64 (8 MFMAs) into d1buf, the wave's sum of squares into part[wave].
constexpr float kRmsPre = 0.015625f;
__device__ __forceinline__ void rot_stage1_regs(const u32x4 (&rq)[2][4], bool f32, bool rms, bool has_w, bool has_su, int wave, int lane,
                                                wht_float4 *d1buf, float *part) {
    float ss = 0.f;
    wht_half8 a[2];
#pragma unroll
    for (int kc = 0; kc < 2; kc++) {
        wht_half8 h;
        if (f32 || rms) {
            float f[8];
            if (f32) {
                const float4_t v0 = __builtin_bit_cast(float4_t, rq[kc][0]), v1 = __builtin_bit_cast(float4_t, rq[kc][1]);
#pragma unroll
                for (int e = 0; e < 4; e++) { f[e] = v0[e]; f[4 + e] = v1[e]; }
            } else {
                const wht_half8 hx = __builtin_bit_cast(wht_half8, rq[kc][0]);
#pragma unroll
                for (int e = 0; e < 8; e++) f[e] = (float)hx[e];
            }
            const wht_half8 wgt = __builtin_bit_cast(wht_half8, rq[kc][2]);
            const float pre = rms ? kRmsPre : 1.0f;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                ss += f[e] * f[e];
                h[e] = (_Float16)(f[e] * pre * (has_w ? (float)wgt[e] : 1.0f));
            }
        } else {
            h = __builtin_bit_cast(wht_half8, rq[kc][0]);
        }
        if (has_su) h = h * __builtin_bit_cast(wht_half8, rq[kc][3]);
        a[kc] = h;
    }
    wht64_wg_stage1_pre(wave, lane, d1buf, a[0], a[1]);
    if (rms) {
        ss = wave_sum(ss);
        if (lane == 0) part[wave] = ss;
    }
}

// ROT: 0 plain; 1: can rotate x while staging it (k = 2048 / 4096, wht64.h); 2: the 14336-wide rotation of rot_k28.h (its own
// instantiation: its registers would make the other rotating launches spill); 3: plain + pair mode (TcParams: sk == -1)
template <class C1, class C2, int NBG, int ROT = 0>
__global__ QPAL_GEMV_BOUNDS(NBG) void tc_gemv_kernel(const uint16_t *ex, const void *etab, int en, int ek, int eon, int eie,
                                                       const uint16_t *esu, const uint16_t *erw, const TcMultiParams mp) {
    constexpr bool TWO = !std::is_void_v<C2>;
    using CB = std::conditional_t<TWO, C2, C1>;
    constexpr int W = gemv_waves<NBG>(), NT = 64 * W;
    static_assert(!ROT || NBG == 1, "fused rotation: batch 1 kernels");
    __shared__ __attribute__((aligned(16))) uint32_t lut[C1::LDS_DWORDS];
    __shared__ __attribute__((aligned(16))) unsigned char scratch[scratch_bytes<NBG>()];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t laneoff = (uint32_t)(lane & (C1::C - 1)) << 2;
    // (measured and not kept, round 4: `s_setprio` tiers — one wave of every SIMD ahead of the next through the scalar prologue, so that a
    // SIMD's first weights are requested after a quarter of it: -0.5 % tokens/s, profiles/r04_ab_prologue2.txt — the staging barrier
    // waits for the slowest wave either way)
#ifdef QPAL_STAMPS
    const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();  // the wave's first instruction (slot 2 of an early-staging launch)
#endif

    const void *cur_tab = nullptr;      // codebook whose image is in LDS
    const uint16_t *cur_x = nullptr;    // activations staged in LDS
    int cur_j = 0;  // (set to the first item's job below)
    // early staging, part 1: request x and the codebook entries from the preloaded arguments (held in registers until
    // the first weight loads have been issued)
    constexpr int NV = (C1::CHUNKS + NT - 1) / NT;
    // (the 128 KiB images of the wide VQ/SQ codebooks would hold 8 entries per thread: spills, measured 20-30 % slower)
    // <= 4 entries per thread of a 16-wave workgroup.  NOT in the any-KV kernels in round 3: holding the early values
    // across their seven decode loops' set-up spilled 16-28 VGPRs, and a kernel that touches scratch at all pays ~1 us per
    // launch — without early staging they have no spills and the q | k | v launch of a mixed-scheme model takes 7.1 instead of
    // 8.3 us (llama3.1-8b_figure1c 651 -> 707 tok/s, mem3p25 678 -> 722, one box: profiles/r03_ab_any_spills.txt).
    constexpr bool kXPerm = QPAL_XPERM_PLAIN != 0 && (ROT == 0 || ROT == 3);  // staged x in the permuted layout (xs_put)
    constexpr int kXL = kXPerm ? 2 : 1;
#ifndef QPAL_ANY_EARLY
#define QPAL_ANY_EARLY 1  // round 4: with the early loads outside the compiler's bookkeeping (inline asm) and the first item as its own
#endif                    // body, the any-KV kernels hold them without spilling (round 3: 16-28 spilled VGPRs, early staging off)
    constexpr bool kEarly = NBG == 1 && (ROT == 0 || ROT == 3) && NV * NT <= 4096 && (QPAL_ANY_EARLY || !is_any_v<C1>);
    constexpr int EV = NV < 4 ? NV : 4;  // image entries a thread holds across the argument fetch; the rest are built after it
    constexpr int XR = kEarlyXChunks;    // 16-byte chunks of x a thread holds likewise
    [[maybe_unused]] u32x4 exr[XR];
    [[maybe_unused]] uint32_t etv[EV][C1::RAWN];
    const bool early = kEarly && (eon & 1) != 0;
    // Rotating launches (ROT == 1), early part: wave t < x_rot requests its row tile of the rotation's inputs, the other waves the
    // table entries of the codebook image they will build — all of it at the wave's first instruction from preloaded arguments,
    // from inline asm (see part 2a below).  Everything is in by the time the kernel arguments are (~0.3 us), i.e. BEFORE the
    // first weights can be requested at all: the weight stream — 6.8 MB for the first step of 4 096 waves, a microsecond of the
    // chip's bandwidth and the launch's critical path — starts against an idle memory system and nothing later competes with it.
    // (What round 3 and the first half of round 4 found again and again: every way of making the rotation's or the image's loads
    // "faster" that put them beside or in front of the weight stream made the token slower.)
    constexpr int kRotTab = 6;  // table entries a builder thread holds: 4 096 chunks over 1 024 - 64 x_rot threads
#ifndef QPAL_ROT_TAB_EARLY
#define QPAL_ROT_TAB_EARLY 1
#endif
    constexpr bool kRotTabEarly = QPAL_ROT_TAB_EARLY != 0;  // (6 more registers held across the prologue: spills in most instantiations)
    [[maybe_unused]] u32x4 rq_e[2][4];  // per 32-column half kc: x (two chunks when fp32), RMSNorm weight, sign vector
    [[maybe_unused]] uint32_t rtv[kRotTab][C1::RAWN];
    // (not for the widest two-stream codecs — 8 | 9 and 9 | 10 dwords per lane: their first weight step and the early registers
    // together spill)
    constexpr bool kRotEarly = ROT == 1 && NBG == 1 && C1::CHUNKS <= kRotTab * (1024 - 64 * 4) && C1::NW + (TWO ? CB::NW : 0) <= 15;
    [[maybe_unused]] const bool rot_early = kRotEarly && (eon & 2) != 0;
    [[maybe_unused]] const int e_rot = (eon >> 4) & 63;
    if constexpr (kRotEarly) {
        if (rot_early) {
            if (wave < e_rot) {
#pragma unroll
                for (int kc = 0; kc < 2; kc++) {
                    const uint32_t el = (uint32_t)((16 * wave + (lane & 15)) * 64 + 32 * kc + 8 * (lane >> 4));
                    if (eon & 4) {
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rq_e[kc][0]) : "v"(el * 4u), "s"(ex));
                        asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(rq_e[kc][1]) : "v"(el * 4u), "s"(ex));
                    } else {
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rq_e[kc][0]) : "v"(el * 2u), "s"(ex));
                    }
                    if (erw) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rq_e[kc][2]) : "v"(el * 2u), "s"(erw));
                    if (esu) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rq_e[kc][3]) : "v"(el * 2u), "s"(esu));
                }
                // ... and stage 1 of the rotation at once, BEFORE the kernel arguments are looked at: everything it needs is
                // preloaded (n, k: where the staging area lies; the flags), its inputs land ~0.25 us from now, and its ~150
                // instructions are then off the path between "kernel arguments in hand" and "first weights requested" — the
                // three builder waves of this SIMD request their weights meanwhile, this wave a little later (its weights are
                // not needed before stage 2 and the barrier behind it are done anyway)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int kc = 0; kc < 2; kc++) {
#pragma unroll
                    for (int q = 0; q < 4; q++) asm volatile("" : "+v"(rq_e[kc][q]));
                }
                uint16_t *xs_e = reinterpret_cast<uint16_t *>(scratch + W * 32 * 4 * en);
                wht_float4 *d1buf_e = reinterpret_cast<wht_float4 *>(xs_e + ((en * ek + 32 + 7) & ~7));
                rot_stage1_regs(rq_e, (eon & 4) != 0, (eon & 8) != 0, erw != nullptr, esu != nullptr, wave, lane, d1buf_e,
                                reinterpret_cast<float *>(d1buf_e + 4 * 4 * 64));
            } else if constexpr (kRotTabEarly) {
                const int bt = tid - 64 * e_rot, nb = 1024 - 64 * e_rot;
#pragma unroll
                for (int r = 0; r < kRotTab; r++) {
                    const int c = bt + r * nb;
                    C1::raw_issue(etab, ((c < C1::CHUNKS ? c : 0) * 4) >> C1::LOG2C, rtv[r]);
                }
            }
        }
    }
    if constexpr (kEarly) {
        // Nothing in this block may USE what it loads (round 4): a use makes the compiler wait for the loads right here, in front of
        // the kernel-argument fetch below and of the first weight loads behind that — the round trips this block exists to overlap.
        if (early) {
            const int total = en * ek;
#pragma unroll
            for (int r = 0; r < XR; r++) {
                const int i = tid * 8 + r * (NT * 8);
                // (clamped instead of predicated: no zero-fill to merge with the load; part 2 stores only i < total + 32 and
                // overwrites the pad chunk with zeros)
                const uint32_t off = (uint32_t)(i < total ? i : 0) * 2u;
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(exr[r]) : "v"(off), "s"(ex));
            }
#pragma unroll
            for (int r = 0; r < EV; r++) {
                const int c = tid + r * NT;
                C1::raw_issue(etab, ((c < C1::CHUNKS ? c : 0) * 4) >> C1::LOG2C, etv[r]);
            }
        }
    }
    // The 14336-wide rotation (ROT == 2, down_proj behind the wrapper) likewise: this wave's pieces of hadK, of x and of the sign
    // vector, and the image's table entries (the rotation's scratch aliases the image: they are written after stage B), all
    // requested now.  Round 3's version requested them inside the item, BEHIND the first weights: vector-memory operations complete
    // in issue order, so the rotation could not start before the first weight step of the wave had landed (+2.8 us per launch
    // against the plain kernel).
    constexpr bool kRot28Early = ROT == 2 && NBG == 1 && C1::LDS_DWORDS * 4 >= kP28 * kTbRow && C1::CHUNKS <= 4 * 1024;
    [[maybe_unused]] const bool rot28_early = kRot28Early && (eon & 2) != 0 && e_rot == kK28;
    [[maybe_unused]] RotK28Regs r28;
    [[maybe_unused]] uint32_t ev28[4][C1::RAWN];
    if constexpr (kRot28Early) {
        if (rot28_early) {
            rot_k28_issue(r28, ex, esu, erw, wave, lane);
#pragma unroll
            for (int r = 0; r < C1::CHUNKS / 1024; r++) C1::raw_issue(etab, ((tid + r * 1024) * 4) >> C1::LOG2C, ev28[r]);
        }
    }
    // ---- Which rows?  (Round 5: the launch's geometry is a host-planned table, TcMultiParams::plan — see LaunchPlan.)
    // Workgroup `item` of geometry class c is member (item mod G) of group (item / G); the group owns rg consecutive virtual rows of
    // the class's row space, and w[member][wave] names this wave's row inside the group, its stream and its steps.  The row's JOB
    // follows from the jobs' row ends.  FAST PATH (eie != 0: one class, <= 4 jobs): lg_g, rg and the row ends are preloaded, so the
    // addresses of the table entry and of the job of the group's FIRST row are known before any kernel argument has arrived — one
    // scalar-load round trip fetches both.
    // Everything on the way to the first weight loads is in this one batch (round 4: a second, dependent round trip there cost
    // 0.3-0.4 us per launch).
    const bool fast = eie != 0;
    const bool one_round = ((eon >> 30) & 1) != 0;  // (early_args: grid == total_items — no second item, no look at gridDim.x)
    const int total_items = mp.total_items;  // (requested with the first batch: read behind the first item, it is a whole scalar-load round trip in every workgroup's tail)
    struct Where {  // a wave's place in the launch
        WaveEnt ent;
        int row0;   // first virtual row of its group
        int jA;     // job of that row
    };
    // generic lookup (launches the fast path does not cover, and the later items of a workgroup in launches of more than one round):
    // plan header, table entry and the row table come from the kernel-argument block, the job in a second, dependent round trip
    auto lookup = [&](int gitem) {
        int c = 0, item = gitem;
        if (mp.ncls > 1 && item >= mp.items0) {
            c = 1;
            item -= mp.items0;
        }
        const int lgg = mp.plan[c].lg_g;
        Where wh;
        wh.row0 = (item >> lgg) * mp.plan[c].rg;
        wh.ent = mp.plan[c].w[item & ((1 << lgg) - 1)][wave];
        int jA = -1, last = 0;
#pragma unroll 1  // (a scalar loop: unrolled, its eight row ends and masks cost the kernel's other paths their scalar registers)
        for (int i = 0; i < mp.njobs; i++) {
            const bool mine = ((mp.cls_mask >> i) & 1) == c;
            if (mine) last = i;
            if (mine && jA < 0 && wh.row0 < mp.row_end[i]) jA = i;
        }
        wh.jA = jA < 0 ? last : jA;  // (rows behind the class's last job: dead rows of the last group)
        return wh;
    };
    // the first item on the fast path: everything from preloaded arguments, its loads form the first batch
    Where f_wh{};
    {
        const int lgg = (eon >> 16) & 3, item = blockIdx.x;
        f_wh.row0 = (item >> lgg) * ((eon >> 18) & 127);
        f_wh.ent = mp.plan[0].w[item & ((1 << lgg) - 1)][wave];
        const int e0 = eie & 0x3ff, e1 = (eie >> 10) & 0x3ff, e2 = (eie >> 20) & 0x3ff;
        f_wh.jA = (f_wh.row0 >= e0 ? 1 : 0) + (f_wh.row0 >= e1 ? 1 : 0) + (f_wh.row0 >= e2 ? 1 : 0);
    }
    if (!fast) f_wh = lookup(blockIdx.x);
    TcParams p = mp.job[f_wh.jA];
    cur_j = f_wh.jA;
    // Pin the batch: left alone, the compiler requests the job only inside the item, after it has waited for the table entry.
    // (round 5: what only the epilogue reads — out, ldo, wscale, oscale, accumulate — is requested behind the first weight loads
    // instead, see there: held from here it pushed the kernel over its scalar-register budget, and every spill is a wait)
    asm volatile("" ::"s"(p.c1), "s"(p.c2), "s"(p.x), "s"(p.tab), "s"(p.nrows), "s"(p.nsc1), "s"(p.vrow0), "s"(f_wh.ent.a), "s"(f_wh.ent.b),
                 "s"(p.nsc2), "s"(p.x_lds), "s"(p.n), "s"(p.k), "s"(mp.zero_chunks), "s"(total_items));
    // One work item.  FIRST (this workgroup's first item, compile-time): the only one that consumes the early-staged registers —
    // a body of its own, so that those registers are plain straight-line values.  (Measured against ONE body with the early wait
    // in front of the item loop, profiles/r04_ab_prologue.txt: the single body is 0.3-0.4 us slower on every short launch —
    // 900 vs 932 tok/s — although it is a third of the code: loop-carried copies of the staged registers, 120 instead of 91 VGPRs.)
    auto run_item = [&](const int gitem, auto first_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value;
        // (declared per item: as a capture of this lambda it ended up in scratch memory.  Two arrays, not a union: `on2` is
        // wave-uniform but the compiler lays the two typed loads out as branches that may BOTH run, and with shared registers
        // the second branch waited — vmcnt(0) — for the first one's loads, i.e. for the early x / table loads, before it issued
        // its own: the weights of half of the waves were requested a memory round trip late)
        struct {
            uint32_t a[C1::NW];
            uint32_t b[TWO ? CB::NW : 1];
        } w;
        // this wave's table entry and its group's job
        Where wh = f_wh;
        if constexpr (!FIRST) {
            wh = lookup(gitem);
            if (wh.jA != cur_j) {
                p = mp.job[wh.jA];
                cur_j = wh.jA;
            }
        }
        const WaveEnt ent = wh.ent;
        const int row0 = wh.row0;
        float *red = reinterpret_cast<float *>(scratch);                           // [W][n][32]
        uint16_t *xs = reinterpret_cast<uint16_t *>(scratch + W * 32 * 4 * p.n);   // [n][k] + 32 zero halves
        const int zero_off = p.n * p.k;
        const bool x_lds = NBG == 1 && p.x_lds;
#ifdef QPAL_STAMPS  // stamp 0 (kernel arguments in hand) carries the time since the wave's first instruction in its bits 52..63
        if (p.dbg && lane == 0 && FIRST) {
            const unsigned long long t_now = __builtin_amdgcn_s_memrealtime();
            p.dbg[((long)blockIdx.x * 16 + wave) * 8] = (t_now & ((1ull << 52) - 1)) | ((t_now - t_entry) << 52);
        }
#endif
        // the wave's row: virtual row -> row inside its job (a group never leaves its job: the jobs' virtual rows are padded to whole
        // groups.  Groups that run across job boundaries — q | k | v of Llama-8B as ONE row space of 192 rows, 64 groups of 4 x 3 rows,
        // 6 instead of 8 steps on the busiest SIMD — were built and measured in round 5: 6.43 us against 6.30, the launch is not bound
        // by its steps; profiles/r05_ab_tape_planner.txt)
        const int vrow = row0 + (int)(ent.a & 255u);
        const int sr = vrow - p.vrow0;            // supertile row inside its job
        const int nrows_j = p.nrows;
        const bool live = sr < nrows_j && ((ent.a >> 16) & 1u) != 0;
        // (the entry's epilogue bits — lead, shared row, run length — are cut out of `ea` again BEHIND the steps: one scalar register
        // held across them instead of four)
        uint32_t ea = ent.a;
        // (stream, [s0, s1)) of the wave's piece (any-KV kernels take column-split jobs too: the two streams are two KV of the same
        // codebook size, and a wave picks its decode loop by the stream its piece lies in — a wave-uniform choice)
        constexpr bool ANY = is_any_v<C1>;
        const int kv1_j = p.kv, kv2_j = p.kv2;
        const bool two_rt = TWO || (ANY && kv2_j != 0);
        const bool on2 = two_rt && ((ent.a >> 8) & 1u) != 0;
        int s0 = (int)(ent.b & 0xffffu);
        int s1 = s0 + (int)(ent.b >> 16);
        if (!live) s0 = s1 = 0;
        const int nw1 = ANY ? kv1_j : C1::NW;  // dwords per lane per supertile of stream 1
        const int nw2 = ANY ? kv2_j : CB::NW;
        const uint32_t *c1_j = p.c1, *c2_j = p.c2;
        const StreamView sv1{c1_j + (long)(live ? sr : 0) * p.nsc1 * 16 * nw1, p.nsc1, 0};
        const StreamView sv2{two_rt ? c2_j + (long)(live ? sr : 0) * p.nsc2 * 16 * nw2 : c1_j, two_rt ? p.nsc2 : p.nsc1,
                             p.col2};
        // the ONE stream this wave's piece lies in, as a buffer descriptor (weight loads: load_step_w_buf)
        const buf_rsrc_t rs_w = gemm_rsrc(on2 ? sv2.base : sv1.base, (on2 ? sv2.nsc * nw2 : sv1.nsc * nw1) * 64);
        // Early staging, part 2a: the early loads were issued from inline asm, i.e. outside the compiler's wait-count bookkeeping —
        // wait for them by hand, HERE: ~0.2-0.3 us after the wave's entry they and the kernel arguments have arrived together
        // (perf/first_touch.hip), and from here on every load in flight is one the compiler knows about.  (Requesting the weights
        // before this wait — they need nothing but the kernel arguments — was tried first: the hardware counter cannot tell the
        // hidden loads from the compiler's own, and the hazard waits the compiler puts between the two typed weight-load branches
        // then wait for the early loads anyway, at points no source change moved.)  Every early value then passes through an
        // (empty) volatile asm, which orders its uses behind the wait.
        // (The waits themselves are UNCONDITIONAL in a first item — with nothing in flight they cost nothing — so that EVERY control-flow
        // path from an early load to a use of its register passes one: perf/check_early_loads.py walks the disassembly of every build
        // for exactly that, tests/test_capi_and_host.py runs it.)
        if constexpr (kRotEarly) {
            if constexpr (FIRST) {
#ifdef QPAL_STAMPS
                if (p.dbg) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (FIRST && rot_early) {  // the rotation's inputs / the image's table entries, requested at the wave's first instruction
                if constexpr (kRotTabEarly) {
#pragma unroll
                    for (int r = 0; r < kRotTab; r++) {
#pragma unroll
                        for (int q = 0; q < C1::RAWN; q++) asm volatile("" : "+v"(rtv[r][q]));
                    }
                }
            }
        }
        if constexpr (kRot28Early) {
            if constexpr (FIRST) {
#ifdef QPAL_STAMPS
                if (p.dbg) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (FIRST && rot28_early) {
                rot_k28_landed(r28);
#pragma unroll
                for (int r = 0; r < C1::CHUNKS / 1024; r++) {
#pragma unroll
                    for (int q = 0; q < C1::RAWN; q++) asm volatile("" : "+v"(ev28[r][q]));
                }
            }
        }
        if constexpr (kEarly) {
            if constexpr (FIRST) {
#ifdef QPAL_STAMPS  // (the stamp-0 store above is younger than the early loads and takes its time)
                if (p.dbg) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (FIRST && early) {
#pragma unroll
                for (int r = 0; r < XR; r++) asm volatile("" : "+v"(exr[r]));
#pragma unroll
                for (int r = 0; r < EV; r++) {
#pragma unroll
                    for (int q = 0; q < C1::RAWN; q++) asm volatile("" : "+v"(etv[r][q]));
                }
            }
        }
        // first step's weights are in flight while x and the codebook image are (re)staged
        if constexpr (ANY) {
            dispatch_kv<C1::S_>(on2 ? kv2_j : kv1_j, [&](auto kc) {
                constexpr int KVr = decltype(kc)::value;
                load_step_w_buf<KVr>(rs_w, s0, lane, reinterpret_cast<uint32_t(&)[KVr]>(w.a));
            });
        } else {
            if constexpr (TWO) {
                if (on2) load_step_w_buf<CB::NW>(rs_w, s0, lane, w.b);
                else load_step_w_buf<C1::NW>(rs_w, s0, lane, w.a);
            } else {
                load_step_w_buf<C1::NW>(rs_w, s0, lane, w.a);
            }
        }
        QPAL_STAMP(1);  // (the first weights are requested)
        if constexpr (FIRST)  // the epilogue's job fields: one more scalar-load batch, in the shadow of the weight loads
            asm volatile("" ::"s"(p.out), "s"(p.wscale), "s"(p.ldo), "s"(p.accumulate), "s"(__builtin_bit_cast(uint32_t, p.oscale)));
        if constexpr (kEarly) {
            if (FIRST && early) {  // early staging, part 2b: registers -> LDS
                const int total = en * ek;
#pragma unroll
                for (int r = 0; r < XR; r++) {
                    const int i = tid * 8 + r * (NT * 8);
                    if (i < total + 32) xs_put<kXPerm>(xs, i, i < total ? exr[r] : u32x4{0u, 0u, 0u, 0u});
                }
#pragma unroll
                for (int r = 0; r < EV; r++) {
                    const int c = tid + r * NT;
                    const uint32_t v = C1::fix(etv[r], ((c < C1::CHUNKS ? c : 0) * 4) >> C1::LOG2C);
                    if (c < C1::CHUNKS) reinterpret_cast<u32x4 *>(lut)[c] = u32x4{v, v, v, v};
                }
                if constexpr (EV < NV) {  // (8-wave experiment build only)
                    for (int c = tid + EV * NT; c < C1::CHUNKS; c += NT) {
                        const uint32_t v = C1::entry(etab, (c * 4) >> C1::LOG2C);
                        reinterpret_cast<u32x4 *>(lut)[c] = u32x4{v, v, v, v};
                    }
                }
                cur_x = ex;
                cur_tab = etab;
                QPAL_STAMP(2);  // (staged; stamp 3 is behind the barrier)
                __syncthreads();
            }
        }
        if (FIRST && mp.zero_chunks > 0) {  // pre-zero a buffer for a later split-K launch on this stream
            for (int i = blockIdx.x * NT + tid; i < mp.zero_chunks; i += gridDim.x * NT) mp.zero[i] = u32x4{0u, 0u, 0u, 0u};
        }
        if (p.tab != cur_tab || (x_lds && p.x != cur_x)) {  // workgroup-uniform
            if (x_lds && p.x != cur_x) {
                const int total = p.n * p.k;  // multiple of 8 halves
                if (ROT == 2 && p.x_rot == 28) {
                  if constexpr (ROT == 2 && C1::LDS_DWORDS * 4 >= kP28 * kTbRow) {
                    // down_proj of Llama-3.1-8B: (hadK(28) (x) H_512) of the 14336-vector on all 16 waves (rot_k28.h); its 40 KiB of
                    // scratch alias the codebook image, whose table entries are requested first and written afterwards
                    constexpr int NVR = C1::CHUNKS / 1024;
                    uint32_t ev[NVR];
                    bool from_regs = false;
                    if constexpr (kRot28Early) from_regs = FIRST && rot28_early;
                    if (from_regs) {
                        if constexpr (kRot28Early) {
                            rot_k28_regs(r28, p.x_su != nullptr, p.x_pre, p.x_post, xs, reinterpret_cast<unsigned char *>(lut), wave, lane,
                                         [](int i) { return xs_index<kXPerm>(i); });
#pragma unroll
                            for (int r = 0; r < NVR; r++) ev[r] = C1::fix(ev28[r], ((tid + r * 1024) * 4) >> C1::LOG2C);
                        }
                    } else
                    rot_k28(p.x, p.x_su, p.x_hadk, p.x_pre, p.x_post, xs, reinterpret_cast<unsigned char *>(lut), wave, lane,
                            [](int i) { return xs_index<kXPerm>(i); }, [&] {
#pragma unroll
                                for (int r = 0; r < NVR; r++) ev[r] = C1::entry(p.tab, ((tid + r * 1024) * 4) >> C1::LOG2C);
                            });
#pragma unroll
                    for (int r = 0; r < NVR; r++) reinterpret_cast<u32x4 *>(lut)[tid + r * 1024] = u32x4{ev[r], ev[r], ev[r], ev[r]};
                    if (tid < 32) xs[total + tid] = 0;
                    cur_tab = p.tab;
                  }
                } else if (ROT == 1 && p.x_rot) {
                  if constexpr (ROT == 1) {
                    // Incoherence rotation fused into the staging (wht64.h: Walsh-Hadamard transform on the matrix pipe).
                    wht_float4 *d1buf = reinterpret_cast<wht_float4 *>(xs + ((total + 32 + 7) & ~7));  // <= 16 KiB
                    // RMSNorm fused into the rotation (decoder-block fusion).  The transform is linear, so the norm's scalar
                    // 1/rms is applied AFTER it: stage 1 rotates x * w * 2^-6 (the power of two keeps fp16 clear of overflow
                    // on residual-stream outliers; the rounding is relative, as on the normalised value) while the same lanes
                    // sum the squares of what they load; the sums meet at the barrier the transform has anyway.  x is read
                    // once and no wave waits for the norm before the matrix pipe starts.
                    const bool rms = p.x_rms_eps > 0.f;
                    float *part = reinterpret_cast<float *>(d1buf + 4 * 4 * 64);
                    // Batch 1 only (the host refuses x_rot otherwise): the transform is spread over all 16 waves in two
                    // stages around one extra barrier — x is read once per workgroup and no wave runs more than ~100
                    // instructions; the codebook image is built by the waves that have no stage-1 tile.
                    // (Alternatives measured slower: one wave quad rotating straight from global memory, +2.0 us per
                    // launch; x * su staged in LDS first, two more barriers.)
                    // Stage 1: done at the wave's first instructions when the launch qualified (rot_early: kernel top); otherwise
                    // here, with loads on demand (every load a round trip of its own: the compiler waits on the spot — the slow path)
                    bool stage1_done = false;
                    if constexpr (kRotEarly) stage1_done = FIRST && rot_early;
                    if (!stage1_done) {
                        float ss = 0.f;
                        auto load_row = [&](const uint16_t *xrow) {
                            return [=, &ss](int t, int kc) {
                                const int off = (16 * t + (lane & 15)) * 64 + 32 * kc + 8 * (lane >> 4);
                                wht_half8 h;
                                if (p.x_src_f32 || rms) {
                                    float f[8];
                                    if (p.x_src_f32) {
                                        const float *xf = reinterpret_cast<const float *>(xrow) + off;
                                        const float4_t v0 = *reinterpret_cast<const float4_t *>(xf), v1 = *reinterpret_cast<const float4_t *>(xf + 4);
#pragma unroll
                                        for (int e = 0; e < 4; e++) { f[e] = v0[e]; f[4 + e] = v1[e]; }
                                    } else {
                                        const wht_half8 hx = *reinterpret_cast<const wht_half8 *>(xrow + off);
#pragma unroll
                                        for (int e = 0; e < 8; e++) f[e] = (float)hx[e];
                                    }
                                    wht_half8 wgt;
                                    if (p.x_rms_w) wgt = *reinterpret_cast<const wht_half8 *>(p.x_rms_w + off);
                                    const float pre = rms ? kRmsPre : 1.0f;
#pragma unroll
                                    for (int e = 0; e < 8; e++) {
                                        ss += f[e] * f[e];
                                        h[e] = (_Float16)(f[e] * pre * (p.x_rms_w ? (float)wgt[e] : 1.0f));
                                    }
                                } else {
                                    h = *reinterpret_cast<const wht_half8 *>(xrow + off);
                                }
                                if (p.x_su) h = h * *reinterpret_cast<const wht_half8 *>(p.x_su + off);
                                return h;
                            };
                        };
                        if (p.x_rot == 4) wht64_wg_stage1<4>(wave, lane, d1buf, load_row(p.x));
                        else wht64_wg_stage1<2>(wave, lane, d1buf, load_row(p.x));
                        if (rms && wave < p.x_rot) {
                            ss = wave_sum(ss);
                            if (lane == 0) part[wave] = ss;
                        }
                    }
                    if (p.tab != cur_tab) {
                        bool built = false;
                        if constexpr (kRotEarly && kRotTabEarly) {
                            if (FIRST && rot_early) {  // (host: every job of the launch has job 0's codebook)
                                if (wave >= p.x_rot) {
                                    const int bt = tid - 64 * p.x_rot, nb = 1024 - 64 * p.x_rot;
#pragma unroll
                                    for (int r = 0; r < kRotTab; r++) {
                                        const int c = bt + r * nb;
                                        const uint32_t v = C1::fix(rtv[r], ((c < C1::CHUNKS ? c : 0) * 4) >> C1::LOG2C);
                                        if (c < C1::CHUNKS) reinterpret_cast<u32x4 *>(lut)[c] = u32x4{v, v, v, v};
                                    }
                                }
                                built = true;
                            }
                        }
                        if (!built && wave >= p.x_rot) C1::template build<QPAL_ROT_BUILD_U>(lut, p.tab, tid - 64 * p.x_rot, 1024 - 64 * p.x_rot);
                        cur_tab = p.tab;
                    }
                    if (tid < 32) xs[total + tid] = 0;
                    __syncthreads();
                    float post = 1.0f;
                    if (rms) {
                        float tot = part[0] + part[1];
                        if (p.x_rot == 4) tot += part[2] + part[3];
                        post = __builtin_amdgcn_rsqf(tot / (float)total + p.x_rms_eps) * (1.0f / kRmsPre);
                    }
                    auto store_row = [&](uint16_t *dst) {
                        return [=](int, int, int i, float v) {
                            dst[xs_index<kXPerm>(i)] = __builtin_bit_cast(uint16_t, (_Float16)((float)(_Float16)(v * post) * p.x_post));
                        };
                    };
                    if (p.x_rot == 4) wht64_wg_stage2<4>(wave, lane, p.x_pre, d1buf, store_row(xs));
                    else wht64_wg_stage2<2>(wave, lane, p.x_pre, d1buf, store_row(xs));
                  }
                } else {
                    for (int i = tid * 8; i < total + 32; i += NT * 8) {
                        u32x4 v{0u, 0u, 0u, 0u};
                        if (i < total) v = *reinterpret_cast<const u32x4 *>(p.x + i);
                        xs_put<kXPerm>(xs, i, v);
                    }
                }
                cur_x = p.x;
            }
            if (p.tab != cur_tab) {
                C1::build(lut, p.tab, tid, NT);
                cur_tab = p.tab;
            }
            QPAL_STAMP(2);
            __syncthreads();
        }
        QPAL_STAMP(3);
        Acc<NBG> acc;  // (zeroed here, not in front of the staging: 16 registers the rotating prologue has other uses for)
        static_for<0, NBG>([&](auto bc) {
            static_for<0, 4>([&](auto ac) {
                acc.v[decltype(bc)::value][decltype(ac)::value] = float4_t{0.f, 0.f, 0.f, 0.f};
            });
        });
        // per-row output scale of the epilogue: requested HERE — behind the first weights and the staging (round 4: its ~25
        // instructions stood in front of the first weight loads of every wave, and the path to those is bound by instruction issue) —
        // and consumed after the steps (a load issued there
        // would put a whole memory round trip at the end of the kernel)
        uint32_t wraw;
        asm volatile("" : "=v"(wraw));  // "no value yet": a constant here would be merged with the load at the join
                                        // below, and the merge waits for the load on the spot
        // The lead wave of a run finishes its row with lanes 0..31 (lanes 32..63: the gate row behind an up row, SwiGLU epilogue only)
        const int fr = lane & 31, fhi = lane >> 5;
        bool fin = ((ea >> 9) & 1u) != 0 && live && fhi == 0;   // lead: first wave of its row's run — it sums the run and writes the row
        if constexpr (ROT == 1) fin = ((ea >> 9) & 1u) != 0 && live && (fhi == 0 || (((ea >> 17) & 1u) != 0 && sr + 1 < nrows_j));
        const int fsrow = sr + fhi;
        if (p.wscale && fin) wraw = p.wscale[fsrow * 32 + fr];
        if constexpr (ANY) {
            dispatch_kv<C1::S_>(on2 ? kv2_j : kv1_j, [&](auto kc) {
                constexpr int KVr = decltype(kc)::value;
                using CK = TcqCodec<C1::S_, KVr>;
                auto &wk = reinterpret_cast<uint32_t(&)[KVr]>(w.a);
                const StreamView &sv = on2 ? sv2 : sv1;
                if (NBG == 1 && x_lds) gemv_run<CK, kXL, NBG>(wk, lut, laneoff, sv, rs_w, p.x, xs, p.k, p.n, zero_off, s0, s1, lane, acc);
                else gemv_run<CK, 0, NBG>(wk, lut, laneoff, sv, rs_w, p.x, xs, p.k, p.n, zero_off, s0, s1, lane, acc);
            });
        } else
        if constexpr (NBG == 1) {
            if (x_lds) {
                if constexpr (!TWO) gemv_run<C1, kXL, 1>(w.a, lut, laneoff, sv1, rs_w, p.x, xs, p.k, p.n, zero_off, s0, s1, lane, acc);
                else if (on2) gemv_run<CB, kXL, 1>(w.b, lut, laneoff, sv2, rs_w, p.x, xs, p.k, p.n, zero_off, s0, s1, lane, acc);
                else gemv_run<C1, kXL, 1>(w.a, lut, laneoff, sv1, rs_w, p.x, xs, p.k, p.n, zero_off, s0, s1, lane, acc);
            } else {
                if constexpr (!TWO) gemv_run<C1, 0, 1>(w.a, lut, laneoff, sv1, rs_w, p.x, xs, p.k, p.n, zero_off, s0, s1, lane, acc);
                else if (on2) gemv_run<CB, 0, 1>(w.b, lut, laneoff, sv2, rs_w, p.x, xs, p.k, p.n, zero_off, s0, s1, lane, acc);
                else gemv_run<C1, 0, 1>(w.a, lut, laneoff, sv1, rs_w, p.x, xs, p.k, p.n, zero_off, s0, s1, lane, acc);
            }
        } else {
            if constexpr (!TWO) gemv_run<C1, 0, NBG>(w.a, lut, laneoff, sv1, rs_w, p.x, xs, p.k, p.n, zero_off, s0, s1, lane, acc);
                else if (on2) gemv_run<CB, 0, NBG>(w.b, lut, laneoff, sv2, rs_w, p.x, xs, p.k, p.n, zero_off, s0, s1, lane, acc);
            else gemv_run<C1, 0, NBG>(w.a, lut, laneoff, sv1, rs_w, p.x, xs, p.k, p.n, zero_off, s0, s1, lane, acc);
        }
        QPAL_STAMP(4);

        // lane (q = lane>>4, c = lane&15 = 2b+u), register r of acc[msub*2+jl] = D[4q+r][c]; valid where
        // r&1 == u.  On even lanes (u = 0): own r = 0 / 2 plus the odd neighbour's r = 1 / 3 are the two
        // column halves of tile rows 2q and 2q+1.  (Odd lanes compute garbage that is never stored.)
        {
            const int q = lane >> 4, cidx = lane & 15;
            static_for<0, NBG>([&](auto bc) {
                constexpr int grp = decltype(bc)::value;
                const int b = 8 * grp + (cidx >> 1);
                const bool writer = (cidx & 1) == 0 && b < p.n;
                float *dst = red + ((wave * p.n + (b < p.n ? b : 0)) * 32) + 2 * q;
                static_for<0, 4>([&](auto ac) {  // rows 16*msub + 8*jl + 2q + {0,1}
                    constexpr int a = decltype(ac)::value;
                    const float4_t d = acc.v[grp][a];
                    const float v0 = d[0] + lane_xor<1>(d[1]);
                    const float v1 = d[2] + lane_xor<1>(d[3]);
                    if (writer) {
                        dst[8 * a] = v0;
                        dst[8 * a + 1] = v1;
                    }
                });
            });
        }
        // (what the final sum needs besides the partials — row, scale, destination — is worked out in FRONT of the barrier: behind it
        // only the lead waves run, and everything there is the launch's tail)
        asm volatile("" : "+s"(ea));
        const bool lead = ((ea >> 9) & 1u) != 0;                       // (wave-uniform)
        const bool row_shared = ((ea >> 10) & 1u) != 0;                 // another workgroup adds to the same row: atomics
        const int run_waves = (int)((ea >> 11) & 31u);
        [[maybe_unused]] const bool lead2 = ((ea >> 17) & 1u) != 0;     // (SwiGLU) the lead of an up row also finishes the gate row behind it
        float fosc = (fin && p.wscale) ? p.oscale * (float)__builtin_bit_cast(_Float16, (uint16_t)wraw) : p.oscale;
        int fdoff = fsrow * 32 + fr;                          // (offsets, not pointers: a pointer through an asm loses its address space)
        int froff = (wave + fhi * run_waves) * p.n * 32 + fr;  // the run's first partial (lanes 32..63: the next run's)
        asm volatile("" : "+v"(fosc), "+v"(fdoff), "+v"(froff));
        float *fdst = p.out + fdoff;
        const float *fred = red + froff;
        QPAL_STAMP(5);
        __syncthreads();
        QPAL_STAMP(6);
        if (lead) {  // (wave-uniform)
            [[maybe_unused]] bool continue_item = false;
            // the incoherent wrappers' `* Wscale * scale`, fused
            const float osc = fosc;
            if constexpr (ROT == 1) {
                if (p.act_out) {  // SwiGLU of an interleaved up | gate layer: up row in lanes r, the gate row behind it in lanes r + 32
                    float v = 0.f;
                    for (int qq = 0; qq < run_waves; qq++) v += fred[qq * 32];
                    const float mine = (float)(_Float16)(v * osc);           // the reference's fp16 up / gate
                    const float gate = lane_xor<32>(mine);  // (lanes r < 32 read lane r + 32: a permlane swap, not an LDS permute)
                    if (fin && fhi == 0 && lead2) {  // (the gate row's own lead wave has nothing to do here)
                        const float sg = (float)(_Float16)(gate * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gate * -1.44269504f)));
                        uint16_t av = __builtin_bit_cast(uint16_t, (_Float16)(sg * mine));
                        if (p.act_su) av ^= p.act_su[(long)(fsrow >> 1) * 32 + fr] & 0x8000u;  // * (+-1): flip the sign bit
                        p.act_out[(long)(fsrow >> 1) * 32 + fr] = av;
                    }
                    continue_item = true;
                }
            }
            if (!continue_item && fin)
            for (int b = 0; b < p.n; b++) {
                float v = 0.f;
                for (int qq = 0; qq < run_waves; qq++) v += fred[(qq * p.n + b) * 32];
                float *dst = fdst + (long)b * p.ldo;
                v *= osc;
                if (row_shared) atomicAdd(dst, v);
                else if (p.accumulate) *dst += v;  // the residual add of a decoder block: out is the fp32 residual stream
                else *dst = v;
            }
        }
        QPAL_STAMP(7);
        if (!one_round) __syncthreads();  // (the next item reuses `red`)
    };
    // (the grid never exceeds the item count: host, plan_launch)
    int gitem = blockIdx.x;
    run_item(gitem, std::true_type{});
    if (!one_round)
        for (gitem += gridDim.x; gitem < total_items; gitem += gridDim.x) run_item(gitem, std::false_type{});
}

// ------------------------------------------------------------------------------------------------
// Decode to fp16 W (bit-exact integer + LUT path).  Each lane stores 8-byte pieces: lane A's and
// lane B's vector of the same j are adjacent columns.
template <class Codec>
__device__ __forceinline__ void dequant_step(const uint32_t *lut, uint32_t laneoff, const uint32_t (&w)[Codec::NW],
                                             uint16_t *__restrict__ wrow0, long ldw) {
    // wrow0 -> W[32*sr + (p>>1)][col0 + 32*sc + 4*u]
    static_for<0, 4>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int ksub = g >> 1, msub = g & 1;
        uint32_t nh = 0u;
        if constexpr (Codec::kNeedsNext) nh = row16_next(Codec::template head<g>(w));
        static_for<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int jl = j & 1, jh = j >> 1;
            const uint32_t va = Codec::template pair<g, j>(lut, laneoff, w, nh);
            const uint32_t vb = Codec::template pair<g, j + 4>(lut, laneoff, w, nh);
            uint16_t *dst = wrow0 + (long)(8 * jl + 16 * msub) * ldw + 16 * ksub + 8 * jh;
            *reinterpret_cast<u32x2 *>(dst) = u32x2{va, vb};
        });
    });
}

template <class Codec>
__device__ __forceinline__ void dequant_stream(const uint32_t *lut, uint32_t laneoff, const uint32_t *__restrict__ c,
                                               int nsc, int nst, int col0, uint16_t *__restrict__ wout, long ldw,
                                               int sr, int wave, int lane) {
    constexpr int NW = Codec::NW;
    const StreamView sv{c + (long)sr * nsc * 16 * NW, nsc, col0};
    for (int s = wave; s < nst; s += 16) {
        uint32_t w[NW];
        load_step_w<NW>(sv, s, lane, w);
        const int sc = s * 4 + (lane >> 4);
        if (sc < nsc) {  // whole 16-lane DPP rows are live or idle together
            uint16_t *wrow0 = wout + ((long)sr * 32 + ((lane & 15) >> 1)) * ldw + col0 + (long)sc * 32 + 4 * (lane & 1);
            dequant_step<Codec>(lut, laneoff, w, wrow0, ldw);
        }
    }
}

// Staged form of the same step: the wave's 32 x 128 tile goes through a private LDS buffer (16 rows x 256 B + 16 B pad, one
// msub half at a time) and leaves as 16-byte stores that cover whole 256-byte row segments — four full 128-byte lines per
// row and instruction instead of 16-byte runs in 32 different rows (the direct form above: 2.1-2.8 TB/s written).
constexpr int kDqStageRowHalves = 136;                       // 272 bytes: conflict-free ds_write_b64 / ds_read_b128
constexpr int kDqStageBytes = 16 * kDqStageRowHalves * 2;    // per wave
template <class Codec>
__device__ __forceinline__ void dequant_step_staged(const uint32_t *lut, uint32_t laneoff, const uint32_t (&w)[Codec::NW],
                                                    uint16_t *stage, uint16_t *__restrict__ wtile, long ldw, int lane,
                                                    int cols_valid) {
    // wtile -> W[32*sr][col0 + 128*step]; cols_valid: live columns of this step (multiple of 32, <= 128)
    const int p = lane & 15, sc = lane >> 4, u = p & 1;
    static_for<0, 2>([&](auto mc) {
        constexpr int msub = decltype(mc)::value;
        static_for<0, 2>([&](auto kc) {
            constexpr int ksub = decltype(kc)::value;
            constexpr int g = ksub * 2 + msub;
            uint32_t nh = 0u;
            if constexpr (Codec::kNeedsNext) nh = row16_next(Codec::template head<g>(w));
            static_for<0, 4>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int jl = j & 1, jh = j >> 1;
                const uint32_t va = Codec::template pair<g, j>(lut, laneoff, w, nh);
                const uint32_t vb = Codec::template pair<g, j + 4>(lut, laneoff, w, nh);
                uint16_t *dst = stage + (8 * jl + (p >> 1)) * kDqStageRowHalves + 32 * sc + 16 * ksub + 8 * jh + 4 * u;
                *reinterpret_cast<u32x2 *>(dst) = u32x2{va, vb};
            });
        });
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = (lane >> 4) + 4 * i, col = 8 * (lane & 15);
            const u32x4 v = *reinterpret_cast<const u32x4 *>(stage + row * kDqStageRowHalves + col);
            if (col < cols_valid) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(wtile + (long)(16 * msub + row) * ldw + col));
        }
    });
}

template <class Codec>
__device__ __forceinline__ void dequant_stream_staged(const uint32_t *lut, uint32_t laneoff, const uint32_t *__restrict__ c,
                                                      int nsc, int nst, int col0, uint16_t *__restrict__ wout, long ldw,
                                                      int sr, int wave, int lane, uint16_t *stage, int ch, int nch) {
    constexpr int NW = Codec::NW;
    const StreamView sv{c + (long)sr * nsc * 16 * NW, nsc, col0};
    const int s_lo = (int)((long)nst * ch / nch), s_hi = (int)((long)nst * (ch + 1) / nch);  // this workgroup's share of the row
    for (int s = s_lo + wave; s < s_hi; s += 16) {
        uint32_t w[NW];
        load_step_w<NW>(sv, s, lane, w);
        int valid = (nsc - 4 * s) * 32;
        valid = valid > 128 ? 128 : valid;
        dequant_step_staged<Codec>(lut, laneoff, w, stage, wout + (long)sr * 32 * ldw + col0 + (long)s * 128, ldw, lane, valid);
    }
}

template <class C1, class C2>
__global__ __launch_bounds__(1024) void tc_dequant_kernel(const TcParams p) {
    __shared__ __attribute__((aligned(16))) uint32_t lut[C1::LDS_DWORDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t laneoff = (uint32_t)(lane & (C1::C - 1)) << 2;
    C1::build(lut, p.tab, tid, 1024);
    __syncthreads();
    // 16-byte row-segment stores through a per-wave LDS transpose where the buffers fit beside the image (64 KiB images)
    // and the output allows 16-byte stores; the direct 8-byte form otherwise
    constexpr bool kStage = C1::LDS_DWORDS * 4 + 16 * kDqStageBytes <= 160 * 1024;
    if constexpr (kStage) {
        __shared__ __attribute__((aligned(16))) uint16_t stage_all[16 * kDqStageBytes / 2];
        if (p.x_lds) {  // host: wout and ldw are 16-byte friendly
            uint16_t *stage = stage_all + wave * (kDqStageBytes / 2);
            const int nch = p.sk < 1 ? 1 : p.sk;  // column chunks per supertile row: enough items to balance the CUs
            for (int item = blockIdx.x; item < p.nrows * nch; item += gridDim.x) {
                const int sr = item / nch, ch = item - sr * nch;
                dequant_stream_staged<C1>(lut, laneoff, p.c1, p.nsc1, p.st1, 0, p.wout, p.ldw, sr, wave, lane, stage, ch, nch);
                if constexpr (!std::is_void_v<C2>)
                    dequant_stream_staged<C2>(lut, laneoff, p.c2, p.nsc2, p.st2, p.col2, p.wout, p.ldw, sr, wave, lane, stage, ch, nch);
            }
            return;
        }
    }
    for (int sr = blockIdx.x; sr < p.nrows; sr += gridDim.x) {
        dequant_stream<C1>(lut, laneoff, p.c1, p.nsc1, p.st1, 0, p.wout, p.ldw, sr, wave, lane);
        if constexpr (!std::is_void_v<C2>)
            dequant_stream<C2>(lut, laneoff, p.c2, p.nsc2, p.st2, p.col2, p.wout, p.ldw, sr, wave, lane);
    }
}

}  // namespace qpal
