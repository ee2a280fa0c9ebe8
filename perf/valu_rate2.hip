// Sustained wave64 issue rate of candidate decode instructions on gfx950 (round 2: which forms are not half rate?).
// hipcc --offload-arch=gfx950 -O3 perf/valu_rate2.hip -o perf/valu_rate2.bin && perf/valu_rate2.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define OP1(s) asm volatile(s : "+v"(a[i]) : "v"(m))
template <int KIND>
__global__ __launch_bounds__(1024) void k(uint32_t* out, int iters) {
  uint32_t a[8];
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i;
  uint32_t m = out[0] + threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (KIND == 0) OP1("v_pk_mad_u16 %0, %0, %0, %1");
        if (KIND == 1) OP1("v_pk_mul_lo_u16 %0, %0, %1");
        if (KIND == 2) OP1("v_mad_u16 %0, %0, %0, %1");
        if (KIND == 3) OP1("v_mad_u32_u16 %0, %0, %0, %1");
        if (KIND == 4) OP1("v_mul_u32_u24 %0, %0, %1");
        if (KIND == 5) OP1("v_and_b32 %0, %0, %1");
        if (KIND == 6) OP1("v_xor_b32 %0, %0, %1");
        if (KIND == 7) OP1("v_add_u32 %0, %0, %1");
        if (KIND == 8) OP1("v_lshrrev_b32 %0, 3, %0");
        if (KIND == 9) OP1("v_bfi_b32 %0, %1, %0, %1");
        if (KIND == 10) OP1("v_lshl_or_b32 %0, %0, 1, %1");
        if (KIND == 11) OP1("v_mov_b32 %0, %1");
        if (KIND == 12) OP1("v_mov_b32_dpp %0, %0 row_ror:15 row_mask:0xf bank_mask:0xf");
        if (KIND == 13) OP1("v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0");
        if (KIND == 14) OP1("v_mul_f32 %0, %0, %1");
        if (KIND == 15) OP1("v_pk_fma_f16 %0, %0, %1, %0");
        if (KIND == 17) OP1("v_mul_lo_u32 %0, %0, %1");
        if (KIND == 18) OP1("v_cvt_f32_ubyte0 %0, %0");
        if (KIND == 19) OP1("v_pk_lshrrev_b16 %0, %1, %0");
        if (KIND == 20) OP1("v_alignbit_b32 %0, %0, %1, 7");
        if (KIND == 21) OP1("v_and_or_b32 %0, %0, %1, %1");
        if (KIND == 22) OP1("v_mad_u32_u24 %0, %0, %0, %0");
        if (KIND == 23) OP1("v_pk_add_u16 %0, %0, %1");
        if (KIND == 24) OP1("v_add_f32 %0, %0, %1");
        if (KIND == 25) OP1("v_fma_f32 %0, %0, %1, %0");
        if (KIND == 26) OP1("v_perm_b32 %0, %0, %1, %1");
        if (KIND == 27) OP1("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD");
        if (KIND == 28) OP1("v_fma_f16 %0, %0, %1, %0");
        if (KIND == 29) OP1("v_lshlrev_b32 %0, 1, %0");
        if (KIND == 30) OP1("v_sub_u32 %0, %0, %1");
        if (KIND == 31) OP1("v_or_b32 %0, %0, %1");
      }
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s += a[i];
  if (s == 0x12345) out[1] = s;
}

template <int KIND>
void run(const char* name, int wg_threads) {
  uint32_t* d;
  hipMalloc(&d, 64);
  hipMemset(d, 0, 64);
  const int iters = 1000, grid = 256;
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
  double per_simd = winstr / (256 * 4) / (ms * 1e-3);
  printf("%-22s wg=%4d  %.3f ms  %.3f G wave-instr/s/SIMD  => %.2f cyc/instr @2.4GHz\n", name, wg_threads, ms, per_simd / 1e9,
         2.4e9 / per_simd);
  hipFree(d);
}

int main() {
  for (int wg : {1024, 512}) {
    run<25>("v_fma_f32", wg);
    run<24>("v_add_f32", wg);
    run<14>("v_mul_f32", wg);
    run<15>("v_pk_fma_f16", wg);
    run<28>("v_fma_f16", wg);
    run<0>("v_pk_mad_u16", wg);
    run<1>("v_pk_mul_lo_u16", wg);
    run<23>("v_pk_add_u16", wg);
    run<19>("v_pk_lshrrev_b16", wg);
    run<2>("v_mad_u16", wg);
    run<3>("v_mad_u32_u16", wg);
    run<22>("v_mad_u32_u24", wg);
    run<4>("v_mul_u32_u24", wg);
    run<13>("v_mul_u32_u24_sdwa", wg);
    run<17>("v_mul_lo_u32", wg);
    run<5>("v_and_b32", wg);
    run<31>("v_or_b32", wg);
    run<6>("v_xor_b32", wg);
    run<27>("v_and_b32_sdwa", wg);
    run<7>("v_add_u32", wg);
    run<30>("v_sub_u32", wg);
    run<8>("v_lshrrev_b32", wg);
    run<29>("v_lshlrev_b32", wg);
    run<9>("v_bfi_b32", wg);
    run<10>("v_lshl_or_b32", wg);
    run<21>("v_and_or_b32", wg);
    run<20>("v_alignbit_b32", wg);
    run<26>("v_perm_b32", wg);
    run<11>("v_mov_b32", wg);
    run<12>("v_mov_b32_dpp", wg);
    run<18>("v_cvt_f32_ubyte0", wg);
  }
  return 0;
}
