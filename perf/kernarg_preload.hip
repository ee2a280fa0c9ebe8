// Does kernel-argument preloading (user SGPRs filled by the dispatcher, -mllvm -amdgpu-kernarg-preload-count=N) take the
// kernarg fetch off a short kernel's critical path on MI355X?  Same source built twice:
//   hipcc --offload-arch=gfx950 -O3 perf/kernarg_preload.hip -o perf/kp0.bin
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=14 perf/kernarg_preload.hip -o perf/kp1.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Big { int v[320]; };  // like TcMultiParams: lives behind the scalar arguments, never preloaded

// what a GEMV launch does first: a cold weight load + a 64 KB LDS image from a table, all addresses from arguments
__global__ __launch_bounds__(1024) void k_work(const uint32_t* tab, const uint32_t* w, int* out, int stride, const Big b) {
  __shared__ u32x4 lut[4096];
  uint32_t a = w[blockIdx.x * stride + threadIdx.x * 7];
  for (int c = threadIdx.x; c < 4096; c += 1024) { uint32_t v = tab[(c >> 3) & 511]; lut[c] = u32x4{v, v, v, v}; }
  __syncthreads();
  if (lut[threadIdx.x].x + a + b.v[threadIdx.x & 255] == 0x12345678) *out = 1;
}
__global__ __launch_bounds__(1024) void k_tiny(const uint32_t* w, int* out, const Big b) {
  if (w[threadIdx.x] + b.v[threadIdx.x & 255] == 0x12345678) *out = 1;
}

template <class F>
float time_graph(F launch, int n) {
  hipStream_t s; hipStreamCreate(&s);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < n; i++) launch(s, i);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, s); hipStreamSynchronize(s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, s);
  for (int r = 0; r < 10; r++) hipGraphLaunch(ge, s);
  hipEventRecord(e1, s); hipStreamSynchronize(s);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / (10 * n);
}

int main() {
  int* out; hipMalloc(&out, 64);
  uint32_t* tab; hipMalloc(&tab, 2048 * 256); hipMemset(tab, 1, 2048 * 256);
  uint32_t* w; hipMalloc(&w, (size_t)256 * 7168 * 4 * 64); hipMemset(w, 1, (size_t)256 * 7168 * 4 * 64);
  Big b{};
  const int N = 200;
  for (int grid : {1, 32, 256}) {
    printf("grid %3d: tiny (one load) %.2f us", grid, time_graph([&](hipStream_t s, int i) { k_tiny<<<grid, 1024, 0, s>>>(w + (size_t)(i % 64) * 4096, out, b); }, N));
    printf("   cold load + 64KB image %.2f us\n", time_graph([&](hipStream_t s, int i) { k_work<<<grid, 1024, 0, s>>>(tab + (i % 256) * 512, w + (size_t)(i % 64) * 256 * 7168, out, 7168, b); }, N));
  }
  return 0;
}
