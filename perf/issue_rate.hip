// How fast does ONE wave get through straight-line code when 1 / 4 / 8 / 16 waves of its workgroup run the same code on the CU?
// (Round 4: the ~140 scalar instructions between a GEMV wave's entry and its first weight loads take ~1 us with 16 waves per CU.)
//   hipcc --offload-arch=gfx950 -O3 perf/issue_rate.hip -o perf/issue_rate.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))

// mode 0: 256 dependent s_add_u32; 1: 256 independent-ish s_add (4 chains); 2: 256 dependent v_add_u32; 3: mix s, v alternating (independent of each other)
template <int MODE>
__global__ void probe(unsigned long long *out, int seed) {
    uint32_t s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
    uint32_t v0 = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {
        asm volatile(REP256("s_add_u32 %0, %0, 1\n") : "+s"(s0));
    } else if (MODE == 1) {
        asm volatile(REP16(REP16("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n")) : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
    } else if (MODE == 2) {
        asm volatile(REP256("v_add_u32 %0, %0, %0\n") : "+v"(v0));
    } else {
        asm volatile(REP256("s_add_u32 %0, %0, 1\n v_add_u32 %1, %1, %1\n") : "+s"(s0), "+v"(v0));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (s0 + s1 + s2 + s3 + v0 == 0x12345) out[0] = 1;
}

template <int MODE>
void run(const char *what, int ninstr, unsigned long long *d) {
    for (int waves : {1, 4, 8, 16}) {
        std::vector<unsigned long long> h(256 * 16);
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(64 * waves), 0, 0, d, r);
        (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0, mx = 0; int n = 0;
        for (int b = 0; b < 256; b++) for (int w = 0; w < waves; w++) { const double c = (double)h[b * 16 + w]; sum += c; if (c > mx) mx = c; n++; }
        printf("%-52s %2d waves per workgroup: %.2f counter ticks per instruction (mean), %.2f (slowest wave)\n", what, waves, sum / n / ninstr, mx / ninstr);
    }
}

int main() {
    unsigned long long *d; (void)hipMalloc(&d, 256 * 16 * 8);
    printf("(s_memtime ticks; the counter runs at the shader clock on gfx9)\n");
    run<0>("256 dependent s_add_u32", 256, d);
    run<1>("1024 s_add_u32, four independent chains", 1024, d);
    run<2>("256 dependent v_add_u32", 256, d);
    run<3>("256 x (s_add_u32 ; v_add_u32), two chains", 512, d);
    return 0;
}
