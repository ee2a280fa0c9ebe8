// One TCQ GEMV launch (+ the QPAL_STAMPS diagnostic: per-phase in-kernel time stamps of the first launch per shape).
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include "tcq_kernels_api.h"

namespace qpal {

template <int S, int KV1, int KV2, int NBG, int ROT = 0>
static int launch_one(const TcMultiParams &mp, int grid, hipStream_t stream) {
    [[maybe_unused]] const TcParams &p = mp.job[0];
    const TcEarly e = early_args(mp, grid);
#ifndef QPAL_TCQ_XS  // (experiment: 1 = the conflict-free 32-copy codebook image, 128 KiB, for the 1024-entry codebooks)
#define QPAL_TCQ_XS 0
#endif
    constexpr int XS = (QPAL_TCQ_XS != 0 && S == 9) ? 1 : 0;
    using C1 = TcqCodec<S, KV1, XS>;
    using C2 = std::conditional_t<KV2 == 0, void, TcqCodec<S, KV2 == 0 ? KV1 : KV2, XS>>;
#ifdef QPAL_STAMPS
    // QPAL_STAMPS_BUF=<device address>: EVERY launch stamps into a caller-owned buffer (slot per launch shape: 8 slots of
    // 256 x 16 x 8 u64), also inside a captured graph — the steady-state timeline of a replayed token (perf/stamps_replay.py)
    if (const char *sb = getenv("QPAL_STAMPS_BUF")) {
        static int keys[8] = {0};
        static int nkeys = 0;
        const int key = grid * 131 + p.k + ROT * 7 + 1 + p.nrows * 1009;
        int slot = -1;
        for (int i = 0; i < nkeys; i++)
            if (keys[i] == key) slot = i;
        if (slot < 0 && nkeys < 8) {
            slot = nkeys;
            keys[nkeys++] = key;
            fprintf(stderr, "[stamps-slot] %d %s grid %d m %d k %d groups of %d x %d rows\n", slot, ROT ? "rot" : "plain", grid, p.nrows * 32, p.k,
                    1 << mp.plan[0].lg_g, mp.plan[0].rg);
        }
        if (slot >= 0 && grid <= 256) {
            TcMultiParams mq = mp;
            mq.job[0].dbg = reinterpret_cast<unsigned long long *>(strtoull(sb, nullptr, 0)) + (size_t)slot * 256 * 16 * 8;
            hipLaunchKernelGGL((tc_gemv_kernel<C1, C2, NBG, ROT>), dim3(grid), dim3(64 * gemv_waves<NBG>()), 0, stream, e.x, e.tab, e.n, e.k, e.on, e.ie, e.su, e.rw, mq);
            return (int)hipGetLastError();
        }
    }
    // diagnostic build: stamp the first launch of every grid size and print the per-phase shares
    static int seen[8] = {0};
    static int nseen = 0;
    bool fresh = true;
    for (int i = 0; i < nseen; i++) fresh = fresh && seen[i] != grid * 131 + p.k + p.nrows * 1009;
    if (fresh && nseen < 8) {
        seen[nseen++] = grid * 131 + p.k + p.nrows * 1009;
        TcParams q = p;
        unsigned long long *d = nullptr;
        const size_t nb = (size_t)grid * 16 * 8 * sizeof(unsigned long long);
        hipMalloc(&d, nb);
        hipMemset(d, 0, nb);
        q.dbg = d;
        for (int rep = 0; rep < 3; rep++) { TcMultiParams mq = mp; mq.job[0] = q; hipLaunchKernelGGL((tc_gemv_kernel<C1, C2, NBG, ROT>), dim3(grid), dim3(64 * gemv_waves<1>()), 0, stream, e.x, e.tab, e.n, e.k, e.on, e.ie, e.su, e.rw, mq); }
        hipDeviceSynchronize();
        unsigned long long *h = (unsigned long long *)malloc(nb);
        hipMemcpy(h, d, nb, hipMemcpyDeviceToHost);
        // s_memrealtime: 100 MHz, one clock for the whole device (10 ns steps) — spans across workgroups are meaningful
        unsigned long long t0 = ~0ull, t7 = 0, t0max = 0, t7min = ~0ull;
        double ph[8] = {0}, phmax[8] = {0};
        int nw = 0;
        for (int w = 0; w < grid * 16; w++) h[w * 8] &= (1ull << 52) - 1;  // (bits 52..63: time since the wave's entry, perf/stamps_replay.py)
        for (int w = 0; w < grid * 16; w++) {
            if (!h[w * 8] || !h[w * 8 + 7]) continue;  // wave of a narrower workgroup
            nw++;
            if (h[w * 8] < t0) t0 = h[w * 8];
            if (h[w * 8 + 7] > t7) t7 = h[w * 8 + 7];
            if (h[w * 8] > t0max) t0max = h[w * 8];
            if (h[w * 8 + 7] < t7min) t7min = h[w * 8 + 7];
            unsigned long long prev = h[w * 8];
            for (int i = 1; i < 8; i++) {  // a stamp the path did not pass (early staging skips 2 and 3) reads 0: zero-length phase
                const unsigned long long cur = h[w * 8 + i] ? h[w * 8 + i] : prev;
                const double d = (double)(cur - prev);
                ph[i] += d;
                if (d > phmax[i]) phmax[i] = d;
                prev = cur;
            }
        }
        printf("[stamps] %s grid %d m %d k %d groups of %d x %d rows: first stamp -> last stamp %.2f us (wave starts spread over %.2f us, wave ends over %.2f us); mean / max per-wave phase (us):", ROT ? "ROT" : "plain", grid,
               p.nrows * 32, p.k, 1 << mp.plan[0].lg_g, mp.plan[0].rg, (t7 - t0) / 100.0, (t0max - t0) / 100.0, (t7 - t7min) / 100.0);
        const char *nm[8] = {"", "issue-w", "x+lut", "barrier", "steps", "xor-red", "barrier2", "final"};
        for (int i = 1; i < 8; i++) printf(" %s %.2f/%.2f", nm[i], ph[i] / (nw ? nw : 1) / 100.0, phmax[i] / 100.0);
        printf("\n");
        {   // by wave index inside the workgroup (waves w, w + 4, ... share a SIMD; lower index = older): steps of the wave, length of its steps phase, when it ended
            double len[16] = {0}, end[16] = {0};
            int cnt[16] = {0};
            for (int w = 0; w < grid * 16; w++) {
                if (!h[w * 8] || !h[w * 8 + 7] || !h[w * 8 + 4] || !h[w * 8 + 3]) continue;
                len[w & 15] += (double)(h[w * 8 + 4] - h[w * 8 + 3]);
                end[w & 15] += (double)(h[w * 8 + 4] - t0);
                cnt[w & 15]++;
            }
            printf("[stamps]   wave: steps / steps phase us / ended at us:");
            for (int w = 0; w < 16; w++)
                if (cnt[w]) printf(" %d: %d / %.2f / %.2f |", w, (int)(mp.plan[0].w[0][w].b >> 16), len[w] / cnt[w] / 100.0, end[w] / cnt[w] / 100.0);
            printf("\n");
        }
        if (const char *dump = getenv("QPAL_STAMPS_DUMP")) {  // raw stamps for offline analysis: one binary file per stamped launch
            char path[512];
            snprintf(path, sizeof path, "%s_%s_grid%d_m%d_k%d.bin", dump, ROT ? "rot" : "plain", grid, p.nrows * 32, p.k);
            if (FILE *f = fopen(path, "wb")) {
                fwrite(h, 1, nb, f);
                fclose(f);
            }
        }
        free(h);
        hipFree(d);
    }
#endif
    hipLaunchKernelGGL((tc_gemv_kernel<C1, C2, NBG, ROT>), dim3(grid), dim3(64 * gemv_waves<NBG>()), 0, stream, e.x, e.tab, e.n, e.k, e.on, e.ie, e.su, e.rw, mp);
    return (int)hipGetLastError();
}


}  // namespace qpal
