// Device side of perf/aql_fence_probe.cpp (built as a bare code object, loaded through the HSA runtime).
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ __launch_bounds__(1024) void k_empty(const uint32_t *src, uint32_t *dst, uint32_t nb) {
    if (src == nullptr) *dst = 1;
}

// dst[b] = src[(b + 1) mod grid] + 1, the neighbour block runs on another XCD (blocks go round-robin over the 8 XCDs): after N dispatches in a
// ping-pong chain every word is N only if each dispatch saw what the one before it wrote — through another XCD's L2.
// This form carries the data itself across the boundary: the load bypasses this XCD's L2 (sc0 sc1), the store is written through (sc0 sc1).
extern "C" __global__ __launch_bounds__(1024) void k_chain(const uint32_t *src, uint32_t *dst, uint32_t nb) {
    if (threadIdx.x != 0) return;
    const uint32_t b = blockIdx.x;   // (gridDim.x would be a hidden kernel argument the raw dispatch does not fill in)
    const uint32_t *p = src + (b + 1 == nb ? 0 : b + 1);
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    v += 1;
    uint32_t *o = dst + b;
    asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(o), "v"(v) : "memory");
}

// The same chain with ordinary loads and stores: correct only if the packet's fences do the cache maintenance.
extern "C" __global__ __launch_bounds__(1024) void k_chain_plain(const uint32_t *src, uint32_t *dst, uint32_t nb) {
    if (threadIdx.x != 0) return;
    const uint32_t b = blockIdx.x;   // (gridDim.x would be a hidden kernel argument the raw dispatch does not fill in)
    dst[b] = src[b + 1 == nb ? 0 : b + 1] + 1;
}
