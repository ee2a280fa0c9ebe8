// Any-order launch probe (MI355X): does hipExtAnyOrderLaunch (AQL packet without the barrier bit) let launch N+1 of a
// dependent chain start its workgroups, run a prologue and wait on a device flag while launch N is still draining?
// (hip_ext.h says the flag is "not supported on GFX9xx boards"; perf/dlo_probe.hip found two graph branches serialised.)
// A chain of 128 short kernels (256 WGs x 512 threads, 79 KiB LDS each: two fit on a CU, so waiters can never starve the
// kernel they wait for): [pre-work] [wait for the predecessor's completion counter] [post-work] [signal].
// Modes: 0 eager, stream order; 1 eager + any-order + flags; 2 graph capture of mode 1; 3 graph, stream order (reference).
// hipcc --offload-arch=gfx950 -O3 perf/anyorder_probe.hip -o perf/anyorder_probe.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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

// flags[idx] counts the workgroups of kernel idx that have finished, cumulatively over repetitions: need = (rep + 1) * grid.
__global__ __launch_bounds__(512) void phase(unsigned* flags, int idx, unsigned need, int use_flags, int pre_iters, int post_iters,
                                             uint32_t* sink, unsigned long long* stamps, int* err) {
  extern __shared__ uint32_t lds[];
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  uint32_t v = spin_work(threadIdx.x + idx, pre_iters);
  lds[threadIdx.x] = v;
  if (use_flags && idx > 0) {
    if (threadIdx.x == 0) {
      int guard = 0;
      while ((int)(__hip_atomic_load(&flags[idx - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - need) < 0) {
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
    __hip_atomic_fetch_add(&flags[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (stamps) {
      unsigned long long* s = stamps + ((size_t)idx * gridDim.x + blockIdx.x) * 3;
      s[0] = t0; s[1] = t1; s[2] = t2;
    }
  }
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int N = 128, GRID = 256, LDS = 79 * 1024;
  int pre = argc > 1 ? atoi(argv[1]) : 1, post = argc > 2 ? atoi(argv[2]) : 3;
  unsigned* flags; CK(hipMalloc(&flags, N * 4)); CK(hipMemset(flags, 0, N * 4));
  uint32_t* sink; CK(hipMalloc(&sink, 64));
  int* err; CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  unsigned long long* stamps; CK(hipMalloc(&stamps, (size_t)N * GRID * 3 * 8));
  CK(hipFuncSetAttribute((const void*)phase, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  hipStream_t s1; CK(hipStreamCreate(&s1));
  unsigned reps_done = 0;  // every kernel of every repetition adds GRID to its counter

  auto chain = [&](int use_flags, int anyorder) {
    for (int i = 0; i < N; i++) {
      const unsigned need = (reps_done + 1) * GRID;
      // the first kernel of a repetition keeps the barrier: it must not pass the previous repetition's tail
      hipExtLaunchKernelGGL(phase, dim3(GRID), dim3(512), LDS, s1, nullptr, nullptr, (anyorder && i > 0) ? hipExtAnyOrderLaunch : 0,
                            flags, i, need, use_flags, pre, post, sink, stamps, err);
    }
    reps_done++;
  };

  for (int mode = 0; mode < 4; mode++) {
    const int use_flags = (mode == 1 || mode == 2), anyorder = use_flags, graph = mode >= 2;
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    const int R = 20;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    if (!graph) {
      for (int r = 0; r < 3; r++) chain(use_flags, anyorder);
      CK(hipStreamSynchronize(s1));
      CK(hipEventRecord(e0, s1));
      for (int r = 0; r < R; r++) chain(use_flags, anyorder);
      CK(hipEventRecord(e1, s1));
      CK(hipStreamSynchronize(s1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    } else {
      // a graph bakes `need` in: one graph per repetition count is not an option, so the graph form resets the counters itself
      CK(hipStreamSynchronize(s1));
      CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
      CK(hipMemsetAsync(flags, 0, N * 4, s1));
      const unsigned keep = reps_done; reps_done = 0;
      chain(use_flags, anyorder);
      reps_done = keep;
      CK(hipStreamEndCapture(s1, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int r = 0; r < 3; r++) CK(hipGraphLaunch(ge, s1));
      CK(hipStreamSynchronize(s1));
      CK(hipEventRecord(e0, s1));
      for (int r = 0; r < R; r++) CK(hipGraphLaunch(ge, s1));
      CK(hipEventRecord(e1, s1));
      CK(hipStreamSynchronize(s1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemset(flags, 0, N * 4)); reps_done = 0;
    }
    int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> h((size_t)N * GRID * 3);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    double early = 0, flaglat = 0, span = 0;
    for (int i = 1; i < N; i++) {
      unsigned long long s_min = ~0ull, f_max = 0, e_max = 0, pe_max = 0;
      for (int b = 0; b < GRID; b++) {
        const unsigned long long* s = &h[((size_t)i * GRID + b) * 3];
        const unsigned long long* p = &h[((size_t)(i - 1) * GRID + b) * 3];
        s_min = std::min(s_min, s[0]); f_max = std::max(f_max, s[1]); e_max = std::max(e_max, s[2]);
        pe_max = std::max(pe_max, p[2]);
      }
      early += ((double)pe_max - (double)s_min) / 100.0;
      flaglat += ((double)f_max - (double)pe_max) / 100.0;
      span += ((double)e_max - (double)pe_max) / 100.0;
    }
    static const char* names[] = {"eager, stream order", "eager, any-order + flags", "graph of any-order + flags", "graph, stream order"};
    printf("mode %d (%s, pre %d post %d): %.2f us per kernel | start-before-pred-end %.2f us, pred-end->all-released %.2f us, period %.2f us, err %d\n",
           mode, names[mode], pre, post, ms * 1e3f / (R * N), early / (N - 1), flaglat / (N - 1), span / (N - 1), herr);
    CK(hipMemset(err, 0, 4));
    if (ge) { CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); }
  }
  return 0;
}
