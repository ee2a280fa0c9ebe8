// Dependent-launch overlap probe (MI355X): can launch N+1 of a dependent chain start, run its prologue and wait on a
// device flag while launch N is still draining, when the two go out on two graph branches / streams?
// A chain of 128 short kernels (512 WGs x 512 threads, 79 KiB LDS each: two per CU): [pre-work] [wait for the predecessor's
// completion counter] [post-work] [signal].  Modes: one stream in stream order (today's structure), one stream + flags,
// two alternating streams + flags.  s_memrealtime stamps (100 MHz) show when each kernel's workgroups start, see the flag
// and end.
// hipcc --offload-arch=gfx950 -O3 perf/dlo_probe.hip -o perf/dlo_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t spin_work(uint32_t a, int iters) {
  uint32_t b = a * 3u + 1u, c = a ^ 0x9e3779b9u, d = a + 77u;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      asm volatile("v_mad_u32_u24 %0, %0, %0, %0" : "+v"(a));
      asm volatile("v_mad_u32_u24 %0, %0, %0, %0" : "+v"(b));
      asm volatile("v_mad_u32_u24 %0, %0, %0, %0" : "+v"(c));
      asm volatile("v_mad_u32_u24 %0, %0, %0, %0" : "+v"(d));
    }
  }
  return a + b + c + d;
}

__global__ __launch_bounds__(512) void phase(unsigned* flags, int idx, unsigned need, int use_flags, int pre_iters, int post_iters,
                                             uint32_t* sink, unsigned long long* stamps, int* err) {
  extern __shared__ uint32_t lds[];
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  uint32_t v = spin_work(threadIdx.x + idx, pre_iters);
  lds[threadIdx.x] = v;
  if (use_flags && idx > 0) {
    if (threadIdx.x == 0) {
      int guard = 0;
      while (__hip_atomic_load(&flags[idx - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(1);
        if (++guard > (1 << 14)) { *err = idx; break; }
      }
    }
  }
  __syncthreads();
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  v = spin_work(v + lds[(threadIdx.x + 64) & 511], post_iters);
  if (v == 0x12345) sink[0] = v;
  __syncthreads();
  unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    if (use_flags) __hip_atomic_fetch_add(&flags[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (stamps) {
      unsigned long long* s = stamps + ((size_t)idx * gridDim.x + blockIdx.x) * 3;
      s[0] = t0; s[1] = t1; s[2] = t2;
    }
  }
}


// Persistent alternative: ONE launch, 256 WGs (one per CU), phases separated by a sharded arrival counter (8 shards on lines of
// their own, shard = blockIdx % 8) that every workgroup polls (8 lanes, one shard each).
__global__ __launch_bounds__(1024) void persistent(unsigned* ctr /*[8][32]*/, int nphase, int work_iters, unsigned base, uint32_t* sink, int* err,
                                                   unsigned long long* stamps) {
  extern __shared__ uint32_t lds[];
  uint32_t v = threadIdx.x;
  for (int p = 0; p < nphase; p++) {
    v = spin_work(v, work_iters);
    lds[threadIdx.x] = v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&ctr[(blockIdx.x & 7) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < 64) {
      const unsigned need = base + (unsigned)(p + 1) * (gridDim.x / 8);
      int guard = 0;
      for (;;) {
        unsigned c = need;
        if (threadIdx.x < 8) c = __hip_atomic_load(&ctr[threadIdx.x * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all((int)(c - need) >= 0)) break;
        __builtin_amdgcn_s_sleep(1);
        if (++guard > (1 << 14)) { *err = 1000 + p; break; }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0 && stamps) stamps[(size_t)p * gridDim.x + blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
  }
  if (v == 0x12345) sink[0] = v;
}
struct Result { float us_per_kernel; };

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int N = 128, GRID = 512, LDS = 79 * 1024;
  int pre = argc > 1 ? atoi(argv[1]) : 1, post = argc > 2 ? atoi(argv[2]) : 3;
  unsigned* flags; CK(hipMalloc(&flags, N * 4));
  uint32_t* sink; CK(hipMalloc(&sink, 64));
  int* err; CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  unsigned long long* stamps; CK(hipMalloc(&stamps, (size_t)N * GRID * 3 * 8));
  CK(hipFuncSetAttribute((const void*)phase, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  hipEvent_t ef, ej; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));

  for (int mode = 0; mode < 4; mode++) {
    if (mode == 2 && !getenv("DLO_UNSAFE")) continue;  // 512 WGs per kernel can fill every slot with waiters: deadlock until the guard fires
    // 0: one stream, stream order only; 1: one stream + flags; 2: two streams + flags; 3: two streams + flags, grid 256
    const int use_flags = mode >= 1, two = mode >= 2, grid = mode == 3 ? 256 : GRID;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
    CK(hipMemsetAsync(flags, 0, N * 4, s1));
    if (two) { CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0)); }
    for (int i = 0; i < N; i++) {
      hipStream_t s = (two && (i & 1)) ? s2 : s1;
      hipLaunchKernelGGL(phase, dim3(grid), dim3(512), LDS, s, flags, i, (unsigned)grid, use_flags, pre, post, sink, stamps, err);
    }
    if (two) { CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0)); }
    CK(hipStreamEndCapture(s1, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; r++) CK(hipGraphLaunch(ge, s1));
    CK(hipStreamSynchronize(s1));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int R = 20;
    CK(hipEventRecord(e0, s1));
    for (int r = 0; r < R; r++) CK(hipGraphLaunch(ge, s1));
    CK(hipEventRecord(e1, s1));
    CK(hipStreamSynchronize(s1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> h((size_t)N * grid * 3);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    // per kernel: first start, last flag-seen, last end; relative to predecessor's last end
    double early = 0, flaglat = 0, span = 0;
    for (int i = 1; i < N; i++) {
      unsigned long long s_min = ~0ull, f_max = 0, f_min = ~0ull, e_max = 0, pe_max = 0;
      for (int b = 0; b < grid; b++) {
        const unsigned long long* s = &h[((size_t)i * grid + b) * 3];
        const unsigned long long* p = &h[((size_t)(i - 1) * grid + b) * 3];
        s_min = std::min(s_min, s[0]); f_max = std::max(f_max, s[1]); f_min = std::min(f_min, s[1]); e_max = std::max(e_max, s[2]);
        pe_max = std::max(pe_max, p[2]);
      }
      early += ((double)pe_max - (double)s_min) / 100.0;     // >0: this kernel started before its predecessor ended
      flaglat += ((double)f_max - (double)pe_max) / 100.0;   // predecessor's last end -> our last workgroup released
      span += ((double)e_max - (double)pe_max) / 100.0;      // period
    }
    printf("mode %d (%s, %s, grid %d): %.2f us per kernel | start-before-pred-end %.2f us, pred-end->all-released %.2f us, period %.2f us, err %d\n",
           mode, two ? "two streams" : "one stream", use_flags ? "flags" : "stream order", grid, ms * 1e3f / (R * N),
           early / (N - 1), flaglat / (N - 1), span / (N - 1), herr);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  // persistent kernel: sync cost per phase
  {
    unsigned* ctr; CK(hipMalloc(&ctr, 8 * 32 * 4)); CK(hipMemset(ctr, 0, 8 * 32 * 4));
    CK(hipFuncSetAttribute((const void*)persistent, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    unsigned base = 0;
    for (int work : {0, 1, 4}) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const int P = 128, R = 10;
      hipLaunchKernelGGL(persistent, dim3(256), dim3(1024), 100 * 1024, s1, ctr, P, work, base, sink, err, stamps); base += P * 32;
      CK(hipStreamSynchronize(s1));
      CK(hipEventRecord(e0, s1));
      for (int r = 0; r < R; r++) { hipLaunchKernelGGL(persistent, dim3(256), dim3(1024), 100 * 1024, s1, ctr, P, work, base, sink, err, stamps); base += P * 32; }
      CK(hipEventRecord(e1, s1));
      CK(hipStreamSynchronize(s1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      std::vector<unsigned long long> h((size_t)P * 256);
      CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
      double mean = 0, mx = 0; for (auto t : h) { mean += t / 100.0; mx = std::max(mx, t / 100.0); }
      printf("persistent 256x1024, work %d: %.2f us per phase; arrive->released per WG mean %.2f us max %.2f us, err %d\n", work, ms * 1e3f / (R * P), mean / h.size(), mx, herr);
    }
  }
  return 0;
}
