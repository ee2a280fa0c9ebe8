// One-shot all-gather of a small activation slice over xGMI by direct peer writes: every rank stores its
// slice into EVERY rank's gather buffer over its own point-to-point link and raises a flag there; nobody forwards anything.
// At batch 1 the slices are 2-57 KB: a ring collective pays (world - 1) link latencies, a library call ~10-15 us; this is one
// kernel that fits a captured graph.  The reference has no multi-GPU code at all.
//
// Protocol per call site ("slot": buffers and flags are never shared between the call sites of a token, so a fast rank can
// never overwrite what a slow peer still reads — see qpalette_amd/shard.py PeerGatherer).  Block p of rank r:
//   e = ++epoch[slot][p]                         (own word; identical on all ranks by construction: one increment per call)
//   copy src -> peer p's buffer + r * bytes      (16-byte stores; p == r: local copy)
//   system-scope release, then flag[slot][r] of peer p = e (system-scope atomic store)
//   wait until the LOCAL flag[slot][p] >= e      (system-scope atomic loads, bounded spin)
// When the kernel has ended, every slice has arrived; the kernel boundary makes the peers' writes visible to what follows.
#include <hip/hip_runtime.h>
#include <string.h>

#include "qpal_common.h"

namespace qpal {

constexpr int kPeerMaxWorld = 16;
constexpr int kPeerSlotWords = 64;  // per slot: 16 flags + 16 per-block epochs + error word, 256 bytes

struct PeerParams {
    const u32x4 *src;
    long chunks;  // 16-byte chunks of the slice
    int slot, rank, world;
    u32x4 *buf[kPeerMaxWorld];       // each rank's gather buffer of this slot: [world][chunks]
    unsigned *ws[kPeerMaxWorld];     // each rank's flag block (all slots)
};

__global__ __launch_bounds__(256) void peer_gather_kernel(const PeerParams p) {
    const int peer = blockIdx.x, tid = threadIdx.x;
    __shared__ unsigned e_sh;
    unsigned *mine = p.ws[p.rank] + (long)p.slot * kPeerSlotWords;
    if (tid == 0) {
        const unsigned e = mine[16 + peer] + 1u;
        mine[16 + peer] = e;
        e_sh = e;
    }
    __syncthreads();
    const unsigned e = e_sh;
    u32x4 *dst = p.buf[peer] + (long)p.rank * p.chunks;
    for (long i = tid; i < p.chunks; i += 256) dst[i] = p.src[i];
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        unsigned *flag = p.ws[peer] + (long)p.slot * kPeerSlotWords + p.rank;
        __hip_atomic_store(flag, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        unsigned guard = 0;
        while ((int)(__hip_atomic_load(mine + peer, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++guard > (1u << 24)) {  // a peer that never arrives must not hang the GPU: report and go on
                mine[32] = e;
                break;
            }
        }
    }
}

}  // namespace qpal

using namespace qpal;

extern "C" int qpal_peer_gather(const void *src, long bytes, int slot, void *const *peer_bufs, void *const *peer_ws, int rank,
                                int world, void *stream) {
    if (!src || !peer_bufs || !peer_ws) return QPAL_E_NULL;
    if (world < 1 || world > kPeerMaxWorld || rank < 0 || rank >= world || slot < 0 || bytes <= 0) return QPAL_E_SHAPE;
    if (bytes % 16 || (reinterpret_cast<uintptr_t>(src) & 15)) return QPAL_E_ALIGN;
    PeerParams p{};
    p.src = static_cast<const u32x4 *>(src);
    p.chunks = bytes / 16;
    p.slot = slot;
    p.rank = rank;
    p.world = world;
    for (int r = 0; r < world; r++) {
        if (!peer_bufs[r] || !peer_ws[r]) return QPAL_E_NULL;
        if (reinterpret_cast<uintptr_t>(peer_bufs[r]) & 15) return QPAL_E_ALIGN;
        p.buf[r] = static_cast<u32x4 *>(peer_bufs[r]);
        p.ws[r] = static_cast<unsigned *>(peer_ws[r]);
    }
    hipLaunchKernelGGL(peer_gather_kernel, dim3(world), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return (int)hipGetLastError();
}

// ---- set-up helpers (include/qpal.h): fine-grained / uncached flag memory and IPC mapping ------------------------------
extern "C" int qpal_peer_alloc(void **ptr, long bytes, int kind) {
    if (!ptr) return QPAL_E_NULL;
    if (bytes <= 0 || kind < 0 || kind > 2) return QPAL_E_PARAM;
    void *p = nullptr;
    hipError_t e = kind == 0 ? hipMalloc(&p, (size_t)bytes)
                             : hipExtMallocWithFlags(&p, (size_t)bytes, kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
    if (e != hipSuccess) return (int)e;
    e = hipMemset(p, 0, (size_t)bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(p);
        return (int)e;
    }
    *ptr = p;
    return QPAL_OK;
}

extern "C" int qpal_peer_free(void *ptr) { return ptr ? (int)hipFree(ptr) : QPAL_OK; }

static_assert(sizeof(hipIpcMemHandle_t) == QPAL_IPC_HANDLE_BYTES, "IPC handle size");

extern "C" int qpal_ipc_export(void *ptr, void *handle64) {
    if (!ptr || !handle64) return QPAL_E_NULL;
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, ptr);
    if (e != hipSuccess) return (int)e;
    memcpy(handle64, &h, sizeof(h));
    return QPAL_OK;
}

extern "C" int qpal_ipc_open(const void *handle64, void **ptr) {
    if (!handle64 || !ptr) return QPAL_E_NULL;
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    return (int)hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
}

extern "C" int qpal_ipc_close(void *ptr) { return ptr ? (int)hipIpcCloseMemHandle(ptr) : QPAL_OK; }
