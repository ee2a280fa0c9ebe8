// Calibration kernels of the measurement contract (bench.py `roofline` block; SURVEY.md §8d asks for a MEASURED stream-read
// ceiling beside the nominal 8 TB/s).  Not on the data path: nothing in the modules calls them.
//
//   qpal_calib_stream_read  a pure streaming read of caller-given buffers — 16-byte non-temporal loads, every load of a thread in
//                           flight before the first is consumed, one workgroup of 1024 threads per CU: what the chip delivers when
//                           a launch does nothing but fetch its bytes.  bench.py runs it (a) over > 1 GB in one launch (the
//                           stream ceiling in GB/s) and (b) as a "stream token": one launch per GEMV launch of the token over
//                           exactly that launch's packed buffers, in the same dependent order inside one HIP graph — the time a
//                           token of zero-cost decodes would take through the same launch structure.
//   qpal_calib_decode_rate  the decode + MFMA step of tc_gemv_kernel (gemv_step, the very function) in a loop on register-resident
//                           packed words: no HBM traffic, no launch structure — wave-steps per microsecond a CU sustains, from which
//                           bench.py prices the token's decode floor.
#include "tc_kernels.h"

namespace qpal {

constexpr int kCalibSegs = 16;
struct CalibSegs {
    const u32x4 *p[kCalibSegs];
    unsigned n[kCalibSegs];  // 16-byte chunks
    int nseg;
};

__global__ __launch_bounds__(1024) void calib_stream_kernel(const CalibSegs s, uint32_t *sink) {
    const unsigned gt = blockIdx.x * 1024u + threadIdx.x, T = gridDim.x * 1024u;
    u32x4 acc{0u, 0u, 0u, 0u};
    for (int g = 0; g < s.nseg; g++) {
        const gptr<const u32x4> p = as_global(s.p[g]);
        const unsigned n = s.n[g];
        unsigned i = gt;
        for (; i + 3 * T < n; i += 4 * T) {  // four loads in flight per thread and round
            const u32x4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + T),
                        c = __builtin_nontemporal_load(p + i + 2 * T), d = __builtin_nontemporal_load(p + i + 3 * T);
            acc ^= a ^ b ^ c ^ d;
        }
        for (; i < n; i += T) acc ^= __builtin_nontemporal_load(p + i);
    }
    const uint32_t f = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (f == 0x9e3779b9u) sink[gt & 1023] = f;  // keeps the loads alive; (practically) never taken
}

template <class Codec>
__global__ __launch_bounds__(1024) void calib_decode_kernel(const void *tab, uint32_t *sink, int iters) {
    __shared__ __attribute__((aligned(16))) uint32_t lut[Codec::LDS_DWORDS];
    __shared__ __attribute__((aligned(16))) uint16_t xs[4096 + 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t laneoff = (uint32_t)(lane & (Codec::C - 1)) << 2;
    Codec::build(lut, tab, tid, 1024);
    for (int i = tid; i < 4096 + 64; i += 1024) xs[i] = (uint16_t)(0x3c00 + (i & 63));
    __syncthreads();
    uint32_t w[Codec::NW];
#pragma unroll
    for (int i = 0; i < Codec::NW; i++) w[i] = tid * 2654435761u + i * 40503u + blockIdx.x * 97u;
    Acc<1> acc;
    static_for<0, 4>([&](auto ac) { acc.v[0][decltype(ac)::value] = float4_t{0.f, 0.f, 0.f, 0.f}; });
    const StreamView sv{nullptr, 32, 0};
#pragma nounroll
    for (int it = 0; it < iters; it++) {
        u32x4 xb[1][2];
        load_step_x<1, 1>(sv, xs, xs, 4096, 1, 4096, it & 7, lane, xb);
        gemv_step_any<Codec, 1>(lut, laneoff, w, xb, acc);
#pragma unroll
        for (int i = 0; i < Codec::NW; i++) asm volatile("" : "+v"(w[i]));  // "new packed words": nothing is hoisted, no instruction
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
    static_for<0, 4>([&](auto ac) { s += acc.v[0][decltype(ac)::value][0] + acc.v[0][decltype(ac)::value][3]; });
    if (s == 12345.678f) sink[tid] = 1;
}

}  // namespace qpal

using namespace qpal;

extern "C" int qpal_calib_stream_read(const void *const *srcs, const long *bytes, int nseg, void *sink, int grid, void *stream) {
    if (!srcs || !bytes || !sink) return QPAL_E_NULL;
    if (nseg < 1 || nseg > kCalibSegs || grid < 1 || grid > 4096) return QPAL_E_SHAPE;
    CalibSegs s{};
    s.nseg = nseg;
    for (int g = 0; g < nseg; g++) {
        if (!srcs[g]) return QPAL_E_NULL;
        if (((uintptr_t)srcs[g] & 15) != 0) return QPAL_E_ALIGN;
        if (bytes[g] < 0 || (bytes[g] & 15) != 0 || bytes[g] / 16 > 0xfffffff0L - 4L * grid * 1024) return QPAL_E_SHAPE;
        s.p[g] = static_cast<const u32x4 *>(srcs[g]);
        s.n[g] = (unsigned)(bytes[g] / 16);
    }
    hipLaunchKernelGGL(calib_stream_kernel, dim3(grid), dim3(1024), 0, static_cast<hipStream_t>(stream), s, static_cast<uint32_t *>(sink));
    return (int)hipGetLastError();
}

extern "C" int qpal_calib_decode_rate(const void *tlut, void *sink, int iters, int S, int KV, int grid, void *stream) {
    if (!tlut || !sink) return QPAL_E_NULL;
    if (iters < 1 || grid < 1 || grid > 4096) return QPAL_E_SHAPE;
#define QPAL_TCQ(S_, A_, B_)                                                                                                  \
    if constexpr (B_ == 0)                                                                                                    \
        if (S == S_ && KV == A_) {                                                                                            \
            hipLaunchKernelGGL((calib_decode_kernel<TcqCodec<S_, A_>>), dim3(grid), dim3(1024), 0, static_cast<hipStream_t>(stream), \
                               tlut, static_cast<uint32_t *>(sink), iters);                                                   \
            return (int)hipGetLastError();                                                                                    \
        }
#include "tcq_table.inc"
#undef QPAL_TCQ
    return QPAL_E_PARAM;
}
