// Launch sequences (include/qpal.h qpal_seq_*; seq.h): record a list of dependent launches once, re-issue it per token.
//
// Why: a decoded token is ~130 dependent launches of 5-15 us, and ~2-3 us of every launch boundary is the command processor's barrier
// (wait for the previous kernel's last wave, release / acquire, dispatch), the new workgroups' start-up and their prologue (kernel
// arguments, codebook image, first weights).  An AQL packet without the barrier bit (hipExtAnyOrderLaunch) is dispatched as soon as the
// packet before it has been DISPATCHED: the next launch's workgroups take the CUs the previous launch's workgroups leave, run their
// prologue, and wait on the previous launch's arrival counters before they read x or write anything (tc_gemv_kernel, `seqw`).  The
// queue dispatches packets in order, so a waiting workgroup never holds a CU that a workgroup it waits for still needs.
// hipGraph drops the flag (perf/anyorder_probe.hip: a captured any-order chain runs at the stream-ordered rate), hence the recorder:
// the C-ABI's launchers RECORD into the sequence between qpal_seq_begin / qpal_seq_end instead of launching, qpal_seq_launch replays.
// Only the fused GEMV launches that stage x from preloaded arguments overlap their predecessor; everything else that is recorded
// (memsets, rotating launches, glue kernels) is re-issued in stream order, and the launch behind it starts ordered as well.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <functional>
#include <vector>

#include "seq.h"

namespace qpal {

struct SeqRecorder {
    enum Kind { kGemv, kMemset, kCall };
    struct Rec {
        Kind kind;
        // kGemv
        const void *fn;
        int grid, block;
        TcEarly e;
        TcMultiParams mp;
        bool can_wait, signals, waits;
        // kMemset
        void *ptr;
        size_t pitch, width, height;
        // kCall
        std::function<int(hipStream_t)> call;
    };
    std::vector<Rec> recs;
    unsigned *ctr = nullptr;   // [launch][kSeqShards][32] arrival counters + one error word behind them
    unsigned epoch = 0;        // replays issued
    bool open = false, sealed = false;
    int overlap = 1;           // 0: replay everything in stream order (A/B switch, QPAL_SEQ_OVERLAP=0)
    int need0 = 0;             // timing experiment (QPAL_SEQ_NEED0=1, results INVALID): every wait is satisfied at once — the bound of any hand-over protocol
};

namespace {
thread_local SeqRecorder *g_rec = nullptr;
}

SeqRecorder *seq_recording() { return g_rec; }

int seq_record_gemv(SeqRecorder *r, const void *fn, int grid, int block, const TcEarly &e, const TcMultiParams &mp, bool can_wait, bool signals) {
    SeqRecorder::Rec rec{};
    rec.kind = SeqRecorder::kGemv;
    rec.fn = fn;
    rec.grid = grid;
    rec.block = block;
    rec.e = e;
    rec.mp = mp;
    rec.can_wait = can_wait;
    rec.signals = signals;
    r->recs.push_back(std::move(rec));
    return QPAL_OK;
}

int seq_record_memset(SeqRecorder *r, void *p, size_t pitch, size_t width, size_t height) {
    SeqRecorder::Rec rec{};
    rec.kind = SeqRecorder::kMemset;
    rec.ptr = p;
    rec.pitch = pitch;
    rec.width = width;
    rec.height = height;
    r->recs.push_back(std::move(rec));
    return QPAL_OK;
}

int seq_record_call(SeqRecorder *r, std::function<int(hipStream_t)> call) {
    SeqRecorder::Rec rec{};
    rec.kind = SeqRecorder::kCall;
    rec.call = std::move(call);
    r->recs.push_back(std::move(rec));
    return QPAL_OK;
}

}  // namespace qpal

using namespace qpal;

extern "C" {

int qpal_seq_create(void **seq) {
    if (!seq) return QPAL_E_NULL;
    *seq = new SeqRecorder();
    return QPAL_OK;
}

int qpal_seq_destroy(void *seq) {
    SeqRecorder *r = static_cast<SeqRecorder *>(seq);
    if (!r) return QPAL_E_NULL;
    if (g_rec == r) g_rec = nullptr;
    if (r->ctr) (void)hipFree(r->ctr);
    delete r;
    return QPAL_OK;
}

int qpal_seq_begin(void *seq) {
    SeqRecorder *r = static_cast<SeqRecorder *>(seq);
    if (!r) return QPAL_E_NULL;
    if (g_rec || r->sealed) return QPAL_E_SEQ;
    r->open = true;
    g_rec = r;
    return QPAL_OK;
}

int qpal_seq_end(void *seq) {
    SeqRecorder *r = static_cast<SeqRecorder *>(seq);
    if (!r) return QPAL_E_NULL;
    if (g_rec != r) return QPAL_E_SEQ;
    g_rec = nullptr;
    r->open = false;
    r->sealed = true;
    if (const char *v = getenv("QPAL_SEQ_OVERLAP")) r->overlap = atoi(v);
    if (const char *v = getenv("QPAL_SEQ_NEED0")) r->need0 = atoi(v);
    const size_t n = r->recs.size();
    if (n == 0) return QPAL_OK;
    const size_t words = n * kSeqShards * 32 + 32;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&r->ctr), words * sizeof(unsigned));
    if (e != hipSuccess) return (int)e;
    e = hipMemset(r->ctr, 0, words * sizeof(unsigned));
    if (e != hipSuccess) return (int)e;
    unsigned *err = r->ctr + n * kSeqShards * 32;
    for (size_t i = 0; i < n; i++) {
        SeqRecorder::Rec &c = r->recs[i];
        if (c.kind != SeqRecorder::kGemv) continue;
        const bool next_waits = r->overlap && i + 1 < n && r->recs[i + 1].kind == SeqRecorder::kGemv && r->recs[i + 1].can_wait && c.signals;
        c.mp.seq_signal = next_waits ? r->ctr + i * kSeqShards * 32 : nullptr;
        c.mp.seq_err = err;
        c.waits = r->overlap && i > 0 && c.can_wait && r->recs[i - 1].kind == SeqRecorder::kGemv && r->recs[i - 1].signals;
        if (c.waits) {
            c.mp.seq_wait = r->ctr + (i - 1) * kSeqShards * 32;
            c.mp.seq_wait_grid = r->recs[i - 1].grid;
            c.e.on |= 1 << kSeqWaitBit;
        }
    }
    return QPAL_OK;
}

int qpal_seq_launch(void *seq, void *stream) {
    SeqRecorder *r = static_cast<SeqRecorder *>(seq);
    if (!r) return QPAL_E_NULL;
    if (!r->sealed || g_rec) return QPAL_E_SEQ;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return QPAL_E_SEQ;  // (a graph drops the launch flag and bakes the epoch in)
    r->epoch++;
    for (SeqRecorder::Rec &c : r->recs) {
        if (c.kind == SeqRecorder::kMemset) {
            hipError_t e = c.height == 1 ? hipMemsetAsync(c.ptr, 0, c.width, s) : hipMemset2DAsync(c.ptr, c.pitch, 0, c.width, c.height, s);
            if (e != hipSuccess) return (int)e;
        } else if (c.kind == SeqRecorder::kCall) {
            int rc = c.call(s);
            if (rc) return rc;
        } else {
            c.mp.seq_need = r->need0 ? 0u : r->epoch;
            void *args[9] = {&c.e.x, &c.e.tab, &c.e.n, &c.e.k, &c.e.on, &c.e.ie, &c.e.su, &c.e.rw, &c.mp};
            hipError_t e = hipExtLaunchKernel(c.fn, dim3(c.grid), dim3(c.block), args, 0, s, nullptr, nullptr, c.waits ? hipExtAnyOrderLaunch : 0);
            if (e != hipSuccess) return (int)e;
        }
    }
    return QPAL_OK;
}

// -> launches recorded; *overlapped: those that start under their predecessor; *err (after a stream synchronisation by the caller):
// 0, or 1 + the workgroup whose bounded wait ran out
int qpal_seq_info(void *seq, int *launches, int *overlapped, unsigned *err) {
    SeqRecorder *r = static_cast<SeqRecorder *>(seq);
    if (!r) return QPAL_E_NULL;
    int n = 0, w = 0;
    for (const SeqRecorder::Rec &c : r->recs) {
        n++;
        if (c.kind == SeqRecorder::kGemv && c.waits) w++;
    }
    if (launches) *launches = n;
    if (overlapped) *overlapped = w;
    if (err) {
        *err = 0;
        if (r->ctr) {
            hipError_t e = hipMemcpy(err, r->ctr + r->recs.size() * kSeqShards * 32, sizeof(unsigned), hipMemcpyDeviceToHost);
            if (e != hipSuccess) return (int)e;
        }
    }
    return QPAL_OK;
}

}  // extern "C"
