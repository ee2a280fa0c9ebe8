// What does straight-line code cost when it is executed ONCE per launch (cold instruction cache), compared with the
// same number of instructions executed from a loop?  hipcc --offload-arch=gfx950 -O3 perf/icache_cost.hip -o perf/icache.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
#define ADD "v_add_f32 %0, %0, %1\n"

template <int KB4>  // straight line: KB4 * 1024 v_add_f32 (4 bytes each) = KB4 * 4 KiB of code
__global__ __launch_bounds__(1024) void k_line(float* out, float b) {
  float a = threadIdx.x;
#pragma unroll
  for (int i = 0; i < KB4 * 4; i++) asm volatile(R256(ADD) : "+v"(a) : "v"(b));
  if (a == 123.f) *out = a;
}
__global__ __launch_bounds__(1024) void k_loop(float* out, float b, int n256) {
  float a = threadIdx.x;
  for (int i = 0; i < n256; i++) asm volatile(R256(ADD) : "+v"(a) : "v"(b));
  if (a == 123.f) *out = a;
}

template <class F>
float time_graph(F launch, int n) {
  hipStream_t s; (void)hipStreamCreate(&s);
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < n; i++) launch(s, i);
  (void)hipStreamEndCapture(s, &g);
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, s);
  for (int r = 0; r < 10; r++) (void)hipGraphLaunch(ge, s);
  (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / (10 * n);
}

int main() {
  float* out; (void)hipMalloc(&out, 64);
  const int N = 100;
  for (int grid : {1, 256}) {
    for (int wg : {64, 1024}) {
      printf("grid %3d x %4d threads:", grid, wg);
      printf("  line 4KB %.2f", time_graph([&](hipStream_t s, int) { k_line<1><<<grid, wg, 0, s>>>(out, 1.f); }, N));
      printf("  loop same %.2f |", time_graph([&](hipStream_t s, int) { k_loop<<<grid, wg, 0, s>>>(out, 1.f, 4); }, N));
      printf("  line 16KB %.2f", time_graph([&](hipStream_t s, int) { k_line<4><<<grid, wg, 0, s>>>(out, 1.f); }, N));
      printf("  loop same %.2f |", time_graph([&](hipStream_t s, int) { k_loop<<<grid, wg, 0, s>>>(out, 1.f, 16); }, N));
      printf("  line 64KB %.2f", time_graph([&](hipStream_t s, int) { k_line<16><<<grid, wg, 0, s>>>(out, 1.f); }, N));
      printf("  loop same %.2f us\n", time_graph([&](hipStream_t s, int) { k_loop<<<grid, wg, 0, s>>>(out, 1.f, 64); }, N));
    }
  }
  return 0;
}
