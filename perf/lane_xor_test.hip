// lane_xor<MASK> (qpal_common.h: DPP / permlane-swap forms) against __shfl_xor for every mask it specialises.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -I include -I q-palette_amd/csrc perf/lane_xor_test.hip -o perf/lane_xor_test.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "qpal_common.h"
using namespace qpal;
template <int M>
__device__ void one(float v, int *bad, int slot) {
    const float a = lane_xor<M>(v), b = __shfl_xor(v, (M == 7 || M == 15) ? M : M, 64);
    if (a != b) atomicAdd(bad + slot, 1);
}
__global__ void k(int *bad, float *sums) {
    const float v = (float)(threadIdx.x * 3 + 1);
    one<1>(v, bad, 0); one<2>(v, bad, 1); one<7>(v, bad, 2); one<15>(v, bad, 3); one<16>(v, bad, 4); one<32>(v, bad, 5);
    float t = v;
    for (int sh = 32; sh >= 1; sh >>= 1) t += __shfl_xor(t, sh, 64);
    if (wave_sum(v) != t) atomicAdd(bad + 6, 1);
    float m = -v;
    for (int sh = 32; sh >= 1; sh >>= 1) m = fmaxf(m, __shfl_xor(m, sh, 64));
    if (wave_max(-v) != m) atomicAdd(bad + 7, 1);
    float g = v;
    for (int sh = 8; sh >= 1; sh >>= 1) g += __shfl_xor(g, sh, 64);
    if (group_sum<16>(v) != g) atomicAdd(bad + 8, 1);
    float g32 = v;
    for (int sh = 16; sh >= 1; sh >>= 1) g32 += __shfl_xor(g32, sh, 64);
    if (group_sum<32>(v) != g32) atomicAdd(bad + 9, 1);
    if (threadIdx.x == 0) sums[0] = t;
}
int main() {
    int *bad; float *sums; (void)hipMalloc(&bad, 64); (void)hipMalloc(&sums, 64); (void)hipMemset(bad, 0, 64);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, bad, sums);
    int h[16]; float hs; (void)hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost); (void)hipMemcpy(&hs, sums, 4, hipMemcpyDeviceToHost);
    const char *names[] = {"xor1", "xor2", "half_mirror", "mirror", "xor16", "xor32", "wave_sum", "wave_max", "group_sum16", "group_sum32"};
    int rc = 0;
    for (int i = 0; i < 10; i++) { printf("%-12s %s (%d lanes differ)\n", names[i], h[i] ? "MISMATCH" : "ok", h[i]); rc |= h[i] != 0; }
    printf("wave sum %.1f (expected %.1f)\n", hs, 64.0 * 1 + 3.0 * 63 * 64 / 2);
    return rc;
}
