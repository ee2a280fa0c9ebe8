// Launch sequences (round 5): a recorded list of dependent launches that qpal_seq_launch re-issues on a stream — the fused GEMV
// launches among them WITHOUT the barrier bit of their AQL packet (hipExtAnyOrderLaunch), the dependency kept by in-kernel arrival
// counters instead (TcMultiParams::seq_*; tc_gemv_kernel).  What a HIP graph cannot express: hipGraph drops the flag, and two graph
// branches are serialised (perf/dlo_probe.hip, perf/anyorder_probe.hip).
#pragma once
#include <functional>

#include "tc_kernels.h"

namespace qpal {

struct SeqRecorder;
// the sequence being recorded on the calling thread (between qpal_seq_begin and qpal_seq_end), or null
SeqRecorder *seq_recording();
// record a fused-GEMV launch instead of issuing it.  can_wait: the kernel has the sequence wait and this launch qualifies for it
// (early staging: x staged in LDS from the preloaded arguments); signals: everything the kernel writes is written through
int seq_record_gemv(SeqRecorder *r, const void *fn, int grid, int block, const TcEarly &e, const TcMultiParams &mp, bool can_wait, bool signals);
int seq_record_memset(SeqRecorder *r, void *p, size_t pitch, size_t width, size_t height);
// anything else, re-issued in stream order by calling `call(stream)` at replay
int seq_record_call(SeqRecorder *r, std::function<int(hipStream_t)> call);
// zero fill on the stream, or its record
inline hipError_t zero_async(void *p, size_t pitch, size_t width, size_t height, hipStream_t s) {
    if (SeqRecorder *r = seq_recording()) {
        seq_record_memset(r, p, pitch, width, height);
        return hipSuccess;
    }
    return height == 1 ? hipMemsetAsync(p, 0, width, s) : hipMemset2DAsync(p, pitch, 0, width, height, s);
}

// ONE place that issues (or records) a tc_gemv_kernel launch
template <class C1, class C2, int NBG, int ROT>
inline int launch_gemv_kernel(const TcEarly &e, const TcMultiParams &mp, int grid, hipStream_t stream) {
    auto *fn = &tc_gemv_kernel<C1, C2, NBG, ROT>;
    if (SeqRecorder *r = seq_recording())
        return seq_record_gemv(r, reinterpret_cast<const void *>(fn), grid, 64 * gemv_waves<NBG>(), e, mp,
                               gemv_early_kernel_v<C1, NBG, ROT> && (e.on & 1) != 0, ROT == 0);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * gemv_waves<NBG>()), 0, stream, e.x, e.tab, e.n, e.k, e.on, e.ie, e.su, e.rw, mp);
    return (int)hipGetLastError();
}

}  // namespace qpal
