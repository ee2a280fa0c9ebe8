// How long does the dispatcher take to get all waves of a one-workgroup-per-CU launch running, by workgroup shape?
// Round 4 found (perf/stamps_by_block.py on the GEMV kernels) that the ~1.1 us "wave start spread" of a launch is INSIDE every
// workgroup — all 256 workgroups start together, each CU then receives its 16 waves over ~1 us — so it is a wave-launch RATE,
// and what it depends on (waves per workgroup, registers to initialise, preloaded kernel arguments, LDS size) is worth knowing.
//   hipcc --offload-arch=gfx950 -O3 perf/wave_launch.hip -o perf/wave_launch0.bin
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=8 -DPRELOAD=8 perf/wave_launch.hip -o perf/wave_launch8.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#ifndef PRELOAD
#define PRELOAD 0
#endif

struct Big { int v[320]; };

// T threads, REGS: touch a high VGPR so that the wave is allocated that many; LDSB: static LDS bytes
template <int T, int REGS, int LDSB>
__global__ __launch_bounds__(T) void k(const uint16_t *ex, const void *etab, int en, int ek, int eon, unsigned long long *st, const Big b) {
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    __shared__ char lds[LDSB > 0 ? LDSB : 4];
    if (LDSB > 0) lds[threadIdx.x] = (char)en;
    if constexpr (REGS == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if constexpr (REGS == 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    if constexpr (REGS == 512) asm volatile("v_mov_b32 v255, 0\n v_accvgpr_write_b32 a255, 0" ::: "v255", "a255");
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * (T / 64) + (threadIdx.x >> 6)] = t;
    if (ek == 0x7fffffff && eon == 3 && ex && etab) st[0] = b.v[threadIdx.x & 255] + lds[0];
}

template <int T, int REGS, int LDSB>
void run(const char *name, unsigned long long *d_st, int grid) {
    const int W = T / 64;
    Big b{};
    std::vector<unsigned long long> h((size_t)grid * W);
    double spread = 0, inwg = 0, wgstart = 0;
    const int reps = 20;
    // steady state: the stamps of the LAST of 50 dependent launches of a replayed graph (an isolated launch after a device
    // synchronisation measures the chip waking up: 2-4 us spreads)
    {
        hipStream_t s; (void)hipStreamCreate(&s);
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL((k<T, REGS, LDSB>), dim3(grid), dim3(T), 0, s, (const uint16_t *)d_st, (const void *)d_st, 1, 4096, 1, d_st, b);
        (void)hipStreamEndCapture(s, &g);
        (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int r = 0; r < reps + 3; r++) {
            (void)hipGraphLaunch(ge, s); (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
            if (r < 3) continue;
            (void)hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
            unsigned long long lo = ~0ull, hi = 0;
            double in = 0;
            unsigned long long wlo_min = ~0ull, wlo_max = 0;
            for (int g_ = 0; g_ < grid; g_++) {
                unsigned long long a = ~0ull, z = 0;
                for (int w = 0; w < W; w++) { a = std::min(a, h[g_ * W + w]); z = std::max(z, h[g_ * W + w]); }
                lo = std::min(lo, a); hi = std::max(hi, z);
                in += (double)(z - a);
                wlo_min = std::min(wlo_min, a); wlo_max = std::max(wlo_max, a);
            }
            spread += (double)(hi - lo);
            inwg += in / grid;
            wgstart += (double)(wlo_max - wlo_min);
        }
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(s);
    }
    // dependent launches under graph replay: the floor of a launch of this shape
    hipStream_t s; (void)hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 100; i++) hipLaunchKernelGGL((k<T, REGS, LDSB>), dim3(grid), dim3(T), 0, s, (const uint16_t *)d_st, (const void *)d_st, 1, 4096, 1, d_st, b);
    (void)hipStreamEndCapture(s, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    for (int r = 0; r < 10; r++) (void)hipGraphLaunch(ge, s);
    (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("preload %d  %-34s grid %3d: all waves started within %5.2f us (first waves of the workgroups within %5.2f, the waves of one workgroup within %5.2f); "
           "dependent launch of this empty kernel %5.2f us\n", PRELOAD, name, grid, spread / reps / 100.0, wgstart / reps / 100.0, inwg / reps / 100.0, ms * 1e3 / 1000);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(s);
}

int main() {
    unsigned long long *d_st;
    (void)hipMalloc(&d_st, 1 << 20);
    (void)hipMemset(d_st, 0, 1 << 20);
    for (int grid : {256, 192}) {
        run<1024, 128, 94 * 1024>("1024 thr, 128 VGPR, 94 KB LDS", d_st, grid);
        run<1024, 128, 0>("1024 thr, 128 VGPR, no LDS", d_st, grid);
        run<1024, 64, 0>("1024 thr, <=64 VGPR, no LDS", d_st, grid);
        run<512, 128, 94 * 1024>("512 thr, 128 VGPR, 94 KB LDS", d_st, grid);
        run<512, 256, 94 * 1024>("512 thr, 256 VGPR, 94 KB LDS", d_st, grid);
        run<256, 256, 94 * 1024>("256 thr, 256 VGPR, 94 KB LDS", d_st, grid);
        run<256, 512, 94 * 1024>("256 thr, 512 VGPR, 94 KB LDS", d_st, grid);
    }
    run<512, 128, 64 * 1024>("512 thr, 128 VGPR, 64 KB LDS", d_st, 512);
    return 0;
}
