// What does the decode draw?  The step of the fused GEMV (gemv_step_any, the very function) in a loop on register-resident words — no HBM
// traffic — and a pure HBM read, each held for a few seconds while this process samples the card's power and shader clock from sysfs.
// Knock-out builds of the same source price the step's parts:  -DQPAL_KO_MFMA (no matrix instructions), -DQPAL_KO_GATHER (no LDS gathers).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -I include -I q-palette_amd/csrc [-DQPAL_KO_...] perf/power_probe.hip -o perf/power_probe_<name>.bin
//   perf/power_probe_<name>.bin [seconds per leg]
#include <hip/hip_runtime.h>
#include <glob.h>
#include <chrono>
#include <cstdio>
#include <cctype>
#include <cstdlib>
#include <string>
#include <vector>
#include "tc_kernels.h"
using namespace qpal;

template <class Codec>
__global__ __launch_bounds__(1024) void decode_loop(const void *tab, uint32_t *sink, int iters) {
    __shared__ __attribute__((aligned(16))) uint32_t lut[Codec::LDS_DWORDS];
    __shared__ __attribute__((aligned(16))) uint16_t xs[4096 + 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t laneoff = (uint32_t)(lane & (Codec::C - 1)) << 2;
    Codec::build(lut, tab, tid, 1024);
    for (int i = tid; i < 4096 + 64; i += 1024) xs[i] = i < 4096 ? (uint16_t)(0x3c00 + (i & 63)) : (uint16_t)0;  // (the pad behind x is zero, as in the kernel)
    __syncthreads();
    uint32_t w[Codec::NW];
#pragma unroll
    for (int i = 0; i < Codec::NW; i++) w[i] = tid * 2654435761u + i * 40503u + blockIdx.x * 97u;
    Acc<1> acc;
    static_for<0, 4>([&](auto ac) { acc.v[0][decltype(ac)::value] = float4_t{0.f, 0.f, 0.f, 0.f}; });
    const StreamView sv{nullptr, 32, 0};
#pragma nounroll
    for (int it = 0; it < iters; it++) {
        u32x4 xb[1][2];
        load_step_x<1, 1>(sv, xs, xs, 4096, 1, 4096, it & 7, lane, xb);
        gemv_step_any<Codec, 1>(lut, laneoff, w, xb, acc);
#pragma unroll
        for (int i = 0; i < Codec::NW; i++) asm volatile("" : "+v"(w[i]));
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
    static_for<0, 4>([&](auto ac) { s += acc.v[0][decltype(ac)::value][0] + acc.v[0][decltype(ac)::value][3]; });
    if (s == 12345.678f) sink[tid] = 1;
}

__global__ __launch_bounds__(1024) void stream_read(const u32x4 *p, unsigned n, uint32_t *sink) {
    const unsigned gt = blockIdx.x * 1024u + threadIdx.x, T = gridDim.x * 1024u;
    u32x4 acc{0u, 0u, 0u, 0u};
    unsigned i = gt;
    for (; i + 3 * T < n; i += 4 * T) {
        const u32x4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + T), c = __builtin_nontemporal_load(p + i + 2 * T),
                    d = __builtin_nontemporal_load(p + i + 3 * T);
        acc ^= a ^ b ^ c ^ d;
    }
    const uint32_t f = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (f == 0x9e3779b9u) sink[gt & 1023] = f;
}

static std::string find_one(const char *pat) {
    glob_t g; std::string r;
    if (glob(pat, 0, nullptr, &g) == 0 && g.gl_pathc) r = g.gl_pathv[0];
    globfree(&g);
    return r;
}
static double read_num(const std::string &path) {
    if (path.empty()) return -1;
    FILE *f = fopen(path.c_str(), "r"); if (!f) return -1;
    double v = -1; if (fscanf(f, "%lf", &v) != 1) v = -1; fclose(f);
    return v;
}

int main(int argc, char **argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    // the card this process runs on (the host has several): its PCI address names its sysfs directory
    char bdf[64] = {0}; hipDeviceGetPCIBusId(bdf, sizeof(bdf), 0);
    for (char *c = bdf; *c; c++) *c = (char)tolower(*c);
    const std::string dev = std::string("/sys/bus/pci/devices/") + bdf + "/hwmon/hwmon*/";
    std::string pw = find_one((dev + "power1_average").c_str());
    if (pw.empty()) pw = find_one((dev + "power1_input").c_str());
    const std::string fq = find_one((dev + "freq1_input").c_str());
    fprintf(stderr, "power file: %s\nclock file: %s\n", pw.c_str(), fq.c_str());
    uint32_t *sink, *tab; hipMalloc(&sink, 4096 * 4); hipMalloc(&tab, 1 << 16); hipMemset(tab, 0x3c, 1 << 16);
    const size_t big_bytes = (size_t)2 << 30; u32x4 *big; hipMalloc(&big, big_bytes); hipMemset(big, 1, big_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *variant =
#if defined(QPAL_KO_HASH)
        "no hash multiply (v_add in its place)";
#elif defined(QPAL_HASH_PK16)
        "hash as v_pk_mad_u16";
#elif defined(QPAL_PROBE_MFMA444)
        "MACs as 16 x 4x4x4 matrix instructions (probe: a quarter of the products)";
#elif defined(QPAL_KO_MFMA) && defined(QPAL_KO_GATHER)
        "address arithmetic only (no gathers, no matrix instructions)";
#elif defined(QPAL_KO_MFMA)
        "no matrix instructions";
#elif defined(QPAL_KO_GATHER)
        "no LDS gathers";
#else
        "full step";
#endif
    auto leg = [&](const char *name, auto launch, double units_per_launch, const char *unit) {
        launch(); hipDeviceSynchronize();
        std::vector<double> P, F; double ms_sum = 0; int n = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0);
            // sample while the launch runs
            P.push_back(read_num(pw) * 1e-6); F.push_back(read_num(fq) * 1e-6);
            hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms_sum += ms; n++;
        }
        // steady state = last half of the samples
        double p = 0, f = 0; int m = 0;
        for (size_t i = P.size() / 2; i < P.size(); i++) { p += P[i]; f += F[i]; m++; }
        printf("%-58s %7.1f W  %6.0f MHz  %8.3f ms per launch  %9.2f %s\n", name, p / m, f / m, ms_sum / n, units_per_launch / (ms_sum / n * 1e-3), unit);
        fflush(stdout);
    };
    printf("== %s build (%.0f s per leg; power and shader clock: mean of the second half of the samples)\n", variant, secs);
    { const auto t0 = std::chrono::steady_clock::now(); std::vector<double> P;
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 1.0) { P.push_back(read_num(pw) * 1e-6); hipDeviceSynchronize(); }
      printf("%-58s %7.1f W\n", "idle (context up, nothing queued)", P.back()); }
    const int iters = 40000;  // ~12 ms per launch
    leg("decode step, TCQ S 9 KV 6, 256 x 1024 threads", [&] { decode_loop<TcqCodec<9, 6>><<<256, 1024>>>(tab, sink, iters); }, 256.0 * 16 * iters * 1e-6, "M wave-steps/s");
    leg("decode step, TCQ S 9 KV 7", [&] { decode_loop<TcqCodec<9, 7>><<<256, 1024>>>(tab, sink, iters); }, 256.0 * 16 * iters * 1e-6, "M wave-steps/s");
#if !defined(QPAL_KO_MFMA) && !defined(QPAL_KO_GATHER)
    leg("pure HBM read of 2 GiB, 256 x 1024 threads", [&] { for (int r = 0; r < 8; r++) stream_read<<<256, 1024>>>(big, (unsigned)(big_bytes / 16), sink); }, 8.0 * big_bytes * 1e-9, "GB/s");
#endif
    return 0;
}
