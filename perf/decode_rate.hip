// How fast can a CU decode (window -> hash -> LDS gather) when nothing else is in the way?  Wave-steps per microsecond per CU
// for the tcomb_6_7 codecs, by waves per workgroup (8 = 2 per SIMD, 16 = 4 per SIMD), with the fragments consumed by the
// MFMAs of a GEMV step (MAC = 1) or xor-folded (MAC = 0), packed words kept in registers (no HBM).
// hipcc --offload-arch=gfx950 -O3 -std=c++20 -Iinclude -Iq-palette_amd/csrc perf/decode_rate.hip -o perf/decode_rate.bin
#include <stdio.h>
#include "tc_chain.h"
using namespace qpal;

// (MAC = 0: every decoded pair must reach the result, or the compiler drops its gather — the round-2 version folded only .x
// of each fragment, i.e. one pair in four, and its "56 wave-steps/us" was not a decode rate)
__device__ __forceinline__ uint32_t xor4(u32x4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <class Codec, int W, int MAC, int PIPE>
__global__ __launch_bounds__(64 * W) void k(const uint32_t *tab, uint32_t *out, int iters) {
    __shared__ __attribute__((aligned(16))) uint32_t lut[Codec::LDS_DWORDS];
    __shared__ __attribute__((aligned(16))) uint16_t xs[4096 + 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t laneoff = (uint32_t)(lane & (Codec::C - 1)) << 2;
    Codec::build(lut, tab, tid, 64 * W);
    for (int i = tid; i < 4096 + 64; i += 64 * W) xs[i] = (uint16_t)(0x3c00 + (i & 63));
    __syncthreads();
    uint32_t w[Codec::NW];
#pragma unroll
    for (int i = 0; i < Codec::NW; i++) w[i] = tid * 2654435761u + i * 40503u + blockIdx.x;
    Acc<1> acc;
    static_for<0, 4>([&](auto ac) { acc.v[0][decltype(ac)::value] = float4_t{0.f, 0.f, 0.f, 0.f}; });
    uint32_t fold = 0;
    half8_t af[2][8];
    if (PIPE) { decode_step<Codec>(lut, laneoff, w, af[0]); decode_step<Codec>(lut, laneoff, w, af[1]); }
#pragma nounroll
    for (int it = 0; it < iters; it++) {
        static_for<0, 2>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            u32x4 xb[1][2];
            const uint16_t *row = xs + ((it * 2 + d) & 31) * 128 + 4 * (lane & 1) + 32 * (lane >> 4);
            xb[0][0] = u32x4{*(const uint32_t *)(row), *(const uint32_t *)(row + 2), *(const uint32_t *)(row + 8), *(const uint32_t *)(row + 10)};
            xb[0][1] = u32x4{*(const uint32_t *)(row + 16), *(const uint32_t *)(row + 18), *(const uint32_t *)(row + 24), *(const uint32_t *)(row + 26)};
            if (PIPE) {
                if (MAC) mfma_step(af[d], xb, acc);
                else static_for<0, 8>([&](auto ic) { fold ^= xor4(__builtin_bit_cast(u32x4, af[d][decltype(ic)::value])); });
                __builtin_amdgcn_sched_barrier(0);
                decode_step<Codec>(lut, laneoff, w, af[d]);
            } else {
                half8_t a[8];
                decode_step<Codec>(lut, laneoff, w, a);
                if (MAC) mfma_step(a, xb, acc);
                else static_for<0, 8>([&](auto ic) { fold ^= xor4(__builtin_bit_cast(u32x4, a[decltype(ic)::value])); });
            }
#pragma unroll
            for (int i = 0; i < Codec::NW; i++) w[i] = w[i] * 1664525u + 1013904223u;  // 2 VALU per packed word: new bits every step
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    float s = 0;
    static_for<0, 4>([&](auto ac) { s += acc.v[0][decltype(ac)::value][0] + acc.v[0][decltype(ac)::value][3]; });
    if (PIPE) static_for<0, 2>([&](auto dc) { fold ^= __builtin_bit_cast(u32x4, af[decltype(dc)::value][0]).x; });
    out[blockIdx.x * 64 * W + tid] = fold + (uint32_t)s;
}

template <class Codec, int W, int MAC, int PIPE>
void run(const char *name, const uint32_t *tab, uint32_t *out) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<Codec, W, MAC, PIPE>), dim3(256), dim3(64 * W), 0, 0, tab, out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<Codec, W, MAC, PIPE>), dim3(256), dim3(64 * W), 0, 0, tab, out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double steps = 2.0 * iters * W;  // wave-steps per CU
    printf("%-10s W=%2d MAC=%d PIPE=%d: %.3f ms, %.2f wave-steps/us/CU, %.0f ns per wave-step per SIMD slot (%s)\n", name, W, MAC, PIPE, ms,
           steps / (ms * 1e3), ms * 1e6 / steps * 4, hipGetErrorString(hipGetLastError()));
}

int main() {
    uint32_t *tab, *out;
    (void)hipMalloc(&tab, 1 << 16);
    (void)hipMemset(tab, 0x3c, 1 << 16);
    (void)hipMalloc(&out, 256 * 1024 * 4);
    using A = TcqCodec<9, 6>;
    using B = TcqCodec<9, 7>;
    run<A, 16, 1, 0>("tcq_9_6", tab, out);
    run<A, 8, 1, 0>("tcq_9_6", tab, out);
    run<A, 16, 1, 1>("tcq_9_6", tab, out);
    run<A, 8, 1, 1>("tcq_9_6", tab, out);
    run<A, 4, 1, 1>("tcq_9_6", tab, out);
    run<A, 16, 0, 0>("tcq_9_6", tab, out);
    run<A, 8, 0, 0>("tcq_9_6", tab, out);
    run<B, 16, 1, 0>("tcq_9_7", tab, out);
    run<B, 8, 1, 1>("tcq_9_7", tab, out);
    return 0;
}
