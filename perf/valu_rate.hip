// Measures sustained wave64 VALU issue rate on gfx950 for the integer/bit ops the decode loop uses.
// hipcc --offload-arch=gfx950 -O3 perf/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(1024) void k(uint32_t* out, int iters) {
  uint32_t a[8];
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i;
  float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t m = out[0];
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (KIND == 0) a[i] = __builtin_amdgcn_alignbit(a[i], m, 7);
        if (KIND == 1) asm volatile("v_mad_u32_u24 %0, %1, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(m));
        if (KIND == 2) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "s"(0xffc0), "v"(m));
        if (KIND == 3) f[i] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, a[i]), __builtin_bit_cast(h2_t, m), f[i], false);
        if (KIND == 4) asm volatile("v_lshlrev_b32 %0, 1, %1" : "=v"(a[i]) : "v"(a[i]));
        if (KIND == 5) f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f);
        if (KIND == 6) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(m), "s"(0x05040100));
        if (KIND == 7) asm volatile("v_bfe_u32 %0, %1, 6, 10" : "=v"(a[i]) : "v"(a[i]));
      }
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s += a[i] + (uint32_t)f[i];
  if (s == 0x12345) out[1] = s;
}

template <int KIND>
void run(const char* name, int wg_threads, int grid) {
  uint32_t* d;
  hipMalloc(&d, 64);
  hipMemset(d, 0, 64);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<KIND><<<grid, wg_threads>>>(d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND><<<grid, wg_threads>>>(d, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double winstr = (double)grid * (wg_threads / 64) * iters * 64.0;
  double per_simd = winstr / (256 * 4) / (ms * 1e-3);  // wave-instr / s / SIMD
  printf("%-14s wg=%4d grid=%4d  %.3f ms  %.2f G wave-instr/s/SIMD  => %.2f cycles/instr @2.4GHz\n", name, wg_threads, grid,
         ms, per_simd / 1e9, 2.4e9 / per_simd);
  hipFree(d);
}

int main() {
  for (int wg : {256, 512, 1024}) {
    run<0>("v_alignbit", wg, 256);
    run<1>("v_mad_u32_u24", wg, 256);
    run<2>("v_and_or", wg, 256);
    run<3>("v_dot2_f32_f16", wg, 256);
    run<4>("v_lshlrev", wg, 256);
    run<5>("v_fma_f32", wg, 256);
    run<6>("v_perm", wg, 256);
    run<7>("v_bfe_u32", wg, 256);
  }
  run<0>("v_alignbit x2wg", 1024, 512);
  return 0;
}
