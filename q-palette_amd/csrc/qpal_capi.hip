// C-ABI entry points (include/qpal.h): argument checks, launch geometry, dispatch.  No allocation, no
// synchronisation, everything on the caller's stream (graph-capturable).
#include <stdio.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>

#include "lut_kernels_api.h"
#include "tcq_kernels_api.h"
#include "tc_gemm16.h"
#include <string.h>

using namespace qpal;

namespace {

constexpr int kMaxBatch = 128;  // fused decode + skinny GEMM: up to 8 groups of 16 batch rows (tc_gemm16.h); the per-wave-K-chunk kernels: 64

// batch -> 8-row units the kernels are instantiated for: 1, 2 (16 waves per workgroup), 4, 8 (8 waves: tc_kernels.h gemv_waves);
// 16 (batch 65..128): the lockstep kernel of tc_gemm16.h only
int nbg_of(int n) { return n <= 8 ? 1 : n <= 16 ? 2 : n <= 32 ? 4 : n <= 64 ? 8 : n <= 80 ? 10 : 16; }
int waves_of(int nbg) { return gemv_waves<1>() == 8 ? 8 : (nbg >= 4 ? 8 : 16); }
// workgroups one launch round can hold: one per CU; two with the 8-wave experiment build (tc_kernels.h QPAL_W8), whose
// batch <= 16 kernels leave room for a second workgroup's LDS and registers
int round_capacity(int waves) {
#if defined(QPAL_W8) && !defined(QPAL_W8_ONE)
    return waves == 8 ? 2 * kNumCU : kNumCU;   // (the wide-batch kernels use 256 VGPRs: they would also fit twice, LDS allowing)
#else
    (void)waves;
    return kNumCU;
#endif
}

inline bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// reference's S/KV table (lib/linear/__init__.py:166-172)
bool tcq_pair_ok(int S, int KV) {
    if (S == 9) return KV >= 2 && KV <= 10;
    if (S == 10) return KV >= 8 && KV <= 10;
    if (S == 11) return KV >= 9 && KV <= 10;
    return false;
}

bool lut_tc_ok(int bits, int vec) {
    if (vec == 1) return bits >= 2 && bits <= 8;
    if (vec == 2) return bits >= 2 && bits <= 12;
    return false;
}

bool simt_ok(int bits, int vec) {
    if (vec == 1) return bits >= 2 && bits <= 8;
    if (vec == 2) return bits >= 3 && bits <= 12;
    if (vec == 4) return bits >= 6 && bits <= 12;
    return false;
}

// reduction buffer [16 waves][n][32] fp32 + x [n][k] fp16 must fit the kernel's LDS scratch
int x_fits_lds(int n, int k) {
    static const int no_xlds = getenv("QPAL_NO_XLDS") != nullptr;
    return !no_xlds && n <= 8 && gemv_waves<1>() * 32 * 4 * n + 2 * n * k + 64 <= kScratchBytes && (n * k) % 8 == 0;
}

// (k = 8192 would need the 8-row-tile transform: its registers do not fit the GEMV kernel's 128-VGPR budget)
// and batch 1 only: the workgroup-wide transform needs 16 KiB of LDS beside x (larger batches: qpal_hadamard first)
bool rot_ok(int n, int k) {
#ifdef QPAL_W8
    return false;  // the workgroup-wide transform is written for 16 waves
#endif
    return n == 1 && (k == 2048 || k == 4096) && x_fits_lds(n, k);
}

// the 14336-wide rotation inside the staging (rot_k28.h): batch 1, x in LDS, a codebook image that can lend 40 KiB
bool rot_k28_ok(int n, int k, int K, int image_bytes) {
#ifdef QPAL_W8
    return false;
#endif
    return n == 1 && K == kK28 && k == kN28 && x_fits_lds(n, k) && image_bytes >= kP28 * kTbRow;
}

// x_had of a job -> kernel parameters; QPAL_E_SHAPE where the fused rotation is not available
int set_rotation(TcParams &p, int x_had, const void *x_su, float x_post, int n, int k, const void *x_f32 = nullptr,
                 float x_rms_eps = 0.f, const void *x_rms_w = nullptr, const void *x_hadk = nullptr, int x_K = 0,
                 int image_bytes = 64 * 1024) {
    if (!x_had) return (x_f32 || x_rms_eps > 0.f) ? QPAL_E_PARAM : QPAL_OK;  // fp32 / normalised input: rotation staging only
    if (x_K > 1) {
        if (!rot_k28_ok(n, k, x_K, image_bytes)) return QPAL_E_SHAPE;
        if (!x_hadk || !p.x) return QPAL_E_NULL;
        if (x_f32 || x_rms_eps > 0.f) return QPAL_E_PARAM;
        if (!aligned(p.x, 16) || (x_su && !aligned(x_su, 16)) || !aligned(x_hadk, 2)) return QPAL_E_ALIGN;
        p.x_rot = kK28;
        p.x_hadk = static_cast<const uint16_t *>(x_hadk);
        p.x_su = static_cast<const uint16_t *>(x_su);
        p.x_pre = (float)(1.0 / sqrt((double)k));
        p.x_post = x_post;
        return QPAL_OK;
    }
    if (!rot_ok(n, k)) return QPAL_E_SHAPE;
    if (x_su && !aligned(x_su, 16)) return QPAL_E_ALIGN;
    if (x_f32) {
        p.x = static_cast<const uint16_t *>(x_f32);  // the staging reads it as fp32 (x_src_f32)
        p.x_src_f32 = 1;
    }
    if (!p.x) return QPAL_E_NULL;
    if (x_rms_w && !aligned(x_rms_w, 16)) return QPAL_E_ALIGN;
    p.x_rms_eps = x_rms_eps > 0.f ? x_rms_eps : 0.f;
    p.x_rms_w = static_cast<const uint16_t *>(x_rms_w);
    if (!aligned(p.x, 16)) return QPAL_E_ALIGN;
    p.x_rot = k / 1024;
    p.x_su = static_cast<const uint16_t *>(x_su);
    p.x_pre = (float)(1.0 / sqrt((double)k));
    p.x_post = x_post;
    return QPAL_OK;
}

int nb_of(int n) { return n <= 1 ? 1 : n <= 2 ? 2 : n <= 4 ? 4 : 8; }

// Launch geometry of the fused GEMV (round 5: "tape cut").  16 (8) waves per workgroup, one workgroup per CU.
//
// A GROUP of G workgroups (G = 1, 2, 4) owns RG consecutive supertile rows (32 output rows each).  The group's work is a TAPE:
// row 0 stream 1, row 0 stream 2, row 1 stream 1, ... — RG * (st1 + st2) steps — cut into G equal member ranges, and a member's
// range into one piece per wave (a wave never straddles a row or a stream of a column-split layer; pieces of one row are
// consecutive waves, so the cross-wave reduction sums a run of waves).  A row whose steps fall into two members is SHARED: both
// add their partial sums with float atomics into an output the caller declared zeroed (or accumulates onto) — at most two adders
// per element when a member range holds at least one row's steps, so the sums stay order-independent.  Every earlier mode is a
// tape cut: whole rows (G = 1), split-K (RG = 1), pair mode (G = 2, RG = 2 R - S) — and so is what they could not express, e.g.
// q|k|v of Llama-8B (192 rows of 32 steps) on 256 workgroups: G = 4, RG = 3, 24 steps per workgroup, 6 per SIMD instead of 8.
// The planner enumerates (G, RG) per geometry class (jobs of one launch differ in geometry at most by single- vs two-stream),
// builds the per-wave table of every candidate and keeps the cheapest: rounds of <= 256 workgroups x the busiest SIMD's steps
// (waves w, w + 4, w + 8, w + 12 of a workgroup share a SIMD).  The kernel reads ITS table entry — 8 bytes per (member, wave) —
// with the first kernel-argument round trip: no chunk arithmetic is left in its prologue.
// experiment knobs (perf/ scripts only): QPAL_FORCE_G=<1|2|4>, QPAL_FORCE_RG=<rows per group>, QPAL_SHARE=0 (QPAL_PAIR=0): no
// row sharing on zeroed outputs beyond what the memset rule below allows
static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

struct TapePlan {
    LaunchPlan lp;
    int cost;      // steps of the busiest SIMD of any member
    bool shared;   // some row is summed by two workgroups
    bool ok;
};

// table of one candidate: waves W (16 or 8), group of (1 << lg_g) workgroups x rg rows, st1 + st2 steps per row
TapePlan tape_plan(int lg_g, int rg, int st1, int st2, int W) {
    TapePlan tp{};
    const int G = 1 << lg_g, st = st1 + st2;
    tp.lp.lg_g = lg_g;
    tp.lp.rg = rg;
    const long T = (long)rg * st;
    if (st <= 0 || rg < 1 || rg > 255 || T < G) return tp;
    struct Piece { int row, stream, s0, len; };
    for (int m = 0; m < G; m++) {
        const long t0 = T * m / G, t1 = T * (m + 1) / G;
        Piece pc[kPlanWaves];
        int np = 0;
        for (int r = (int)(t0 / st); r < rg && (long)r * st < t1; r++)
            for (int sidx = 0; sidx < 2; sidx++) {
                const long seg0 = (long)r * st + (sidx ? st1 : 0), seg1 = seg0 + (sidx ? st2 : st1);
                const long a = seg0 > t0 ? seg0 : t0, b = seg1 < t1 ? seg1 : t1;
                if (b <= a) continue;
                if (np == W) return tp;  // more pieces than waves
                pc[np++] = Piece{r, sidx, (int)(a - seg0), (int)(b - a)};
            }
        // Waves per piece.  What counts is the busiest SIMD (waves w, w + 4, ... share one), not the longest wave: gate | up of
        // Llama-8B — a member holds three whole rows and half of a fourth, seven pieces of 16 steps — is balanced by 2 waves x 8
        // steps on the whole pieces and 4 x 4 on the half row (28 steps on every SIMD), not by 7 steps everywhere (21 waves).
        // Try every quantum q: ceil(len / q) waves per piece, the waves left over given to the piece(s) with the longest waves or
        // all to one piece — and keep the assignment whose busiest SIMD is least loaded.
        // Wave ORDER inside a member: waves of one row stay together (the reduction sums a run of consecutive waves), but rows — and
        // the pieces of a row — may come in any order, and the order matters: the hardware serves a SIMD's oldest wave first, so a
        // short piece on waves 0..3 is finished early and leaves its SIMD with three waves for the rest of the launch, while the
        // same piece on the YOUNGEST waves runs in the gaps of the long ones and ends with them (measured, round 5: gate | up of
        // Llama-8B 14.5 us with the half row's 4-step waves first, 13.6 with them last — what rounds 3-4's pair mode did).
        // Short waves last: rows by their shortest wave, descending; inside a row the pieces by wave length, descending.
        const long L = t1 - t0;
        struct Slot { int piece, part, ns; };
        Slot best_slots[kPlanWaves];
        int best_nslots = 0, best_cost = 1 << 30, best_long = 1 << 30;
        const int q0 = (int)((L + W - 1) / W);
        auto consider = [&](const int *nw) {
            int order[kPlanWaves], per[kPlanWaves], rowmin[256];
            for (int i = 0; i < np; i++) rowmin[pc[i].row] = 1 << 30;
            for (int i = 0; i < np; i++) {
                order[i] = i;
                per[i] = pc[i].len / nw[i];  // (the shortest wave of the piece)
                if (per[i] < rowmin[pc[i].row]) rowmin[pc[i].row] = per[i];
            }
            for (int i = 1; i < np; i++)  // insertion sort: (row's SHORTEST wave desc — rows with short waves last —, row asc, piece's wave length desc)
                for (int k = i; k > 0; k--) {
                    const int x = order[k - 1], y = order[k];
                    const bool swap = rowmin[pc[y].row] != rowmin[pc[x].row] ? rowmin[pc[y].row] > rowmin[pc[x].row]
                                      : pc[y].row != pc[x].row             ? pc[y].row < pc[x].row
                                                                           : per[y] > per[x];
                    if (!swap) break;
                    order[k - 1] = y;
                    order[k] = x;
                }
            Slot slots[kPlanWaves];
            int simd[4] = {0, 0, 0, 0}, w = 0, longest = 0;
            for (int oi = 0; oi < np; oi++) {
                const int i = order[oi];
                for (int part = 0; part < nw[i]; part++, w++) {
                    const int ns = pc[i].len / nw[i] + (part < pc[i].len % nw[i] ? 1 : 0);
                    slots[w] = Slot{i, part, ns};
                    simd[w & 3] += ns;
                    if (ns > longest) longest = ns;
                }
            }
            int c = simd[0];
            for (int sd = 1; sd < 4; sd++) c = simd[sd] > c ? simd[sd] : c;
            if (c < best_cost || (c == best_cost && longest < best_long)) {
                best_cost = c;
                best_long = longest;
                best_nslots = w;
                for (int k = 0; k < w; k++) best_slots[k] = slots[k];
            }
        };
        for (int q = q0 > 0 ? q0 : 1; q <= 4 * q0 + 4; q++) {
            int nw[kPlanWaves], used = 0;
            for (int i = 0; i < np; i++) {
                nw[i] = (pc[i].len + q - 1) / q;
                used += nw[i];
            }
            if (used > W) continue;
            consider(nw);
            const int spare = W - used;
            if (spare == 0) continue;
            {   // spare waves, one at a time, to the piece with the longest waves
                int g[kPlanWaves];
                for (int i = 0; i < np; i++) g[i] = nw[i];
                for (int k = 0; k < spare; k++) {
                    int at = -1, worst = 0;
                    for (int i = 0; i < np; i++) {
                        const int pr = (pc[i].len + g[i] - 1) / g[i];
                        if (g[i] < pc[i].len && pr > worst) { worst = pr; at = i; }
                    }
                    if (at < 0) break;
                    g[at]++;
                }
                consider(g);
            }
            for (int t = 0; t < np; t++) {  // ... or all to one piece
                int g[kPlanWaves];
                for (int i = 0; i < np; i++) g[i] = nw[i];
                g[t] = g[t] + spare < pc[t].len ? g[t] + spare : pc[t].len;
                consider(g);
            }
        }
        if (best_cost == 1 << 30) return tp;
        int wave = best_nslots;
        for (int k = 0; k < best_nslots; k++) {
            const Piece &pp = pc[best_slots[k].piece];
            int nparts = 0;  // (waves of this piece: its slots are consecutive)
            for (int k2 = 0; k2 < best_nslots; k2++) nparts += best_slots[k2].piece == best_slots[k].piece;
            const int base = pp.len / nparts, rem = pp.len % nparts, part = best_slots[k].part;
            const int s0 = pp.s0 + part * base + (part < rem ? part : rem);
            if (best_slots[k].ns > 0xffff || s0 > 0xffff) return tp;
            WaveEnt &e = tp.lp.w[m][k];
            e.a = (uint32_t)pp.row | ((uint32_t)pp.stream << 8) | (1u << 16);
            e.b = (uint32_t)s0 | ((uint32_t)best_slots[k].ns << 16);
        }
        // runs: consecutive waves of one row; the first is the run's lead and knows its length
        for (int w0 = 0; w0 < wave;) {
            int w1 = w0 + 1;
            while (w1 < wave && (tp.lp.w[m][w1].a & 255u) == (tp.lp.w[m][w0].a & 255u)) w1++;
            tp.lp.w[m][w0].a |= (1u << 9) | ((uint32_t)(w1 - w0) << 11);
            w0 = w1;
        }
        // busiest SIMD of this member
        const int per = W / 4;
        for (int sd = 0; sd < 4; sd++) {
            int c = 0;
            for (int q = 0; q < per; q++) c += (int)(tp.lp.w[m][sd + 4 * q].b >> 16);
            if (c > tp.cost) tp.cost = c;
        }
    }
    // shared rows: a row whose steps lie in two member ranges
    for (int r = 0; r < rg; r++) {
        int owners = 0;
        for (int m = 0; m < G; m++) {
            bool has = false;
            for (int w = 0; w < W; w++) has = has || ((tp.lp.w[m][w].a >> 16 & 1u) && (int)(tp.lp.w[m][w].a & 255u) == r);
            owners += has;
        }
        if (owners > 2) return tp;  // at most two adders per output element: (0 + a) + b == (0 + b) + a, results stay reproducible to the bit
        if (owners > 1) {
            tp.shared = true;
            for (int m = 0; m < G; m++)
                for (int w = 0; w < W; w++)
                    if ((tp.lp.w[m][w].a >> 16 & 1u) && (int)(tp.lp.w[m][w].a & 255u) == r) tp.lp.w[m][w].a |= 1u << 10;
        }
    }
    tp.ok = true;
    return tp;
}

// virtual rows of class c under a candidate: every job padded to whole groups
int class_rows(const TcMultiParams &mp, int c, int rg) {
    int rows = 0;
    for (int j = 0; j < mp.njobs; j++)
        if (mp.job[j].cls == c) rows += (mp.job[j].nrows + rg - 1) / rg * rg;
    return rows;
}

// out_zeroed[j]: the caller pre-zeroed job j's output (row sharing is then free of a memset node); nullptr: nobody did
void plan_launch(TcMultiParams &mp, const int *out_zeroed, int &grid, int waves = 16) {
    static const int force_g = env_int("QPAL_FORCE_G", 0);
    static const int force_rg = env_int("QPAL_FORCE_RG", 0);
    static const int share_on = env_int("QPAL_SHARE", env_int("QPAL_PAIR", 1));
    const int cap = round_capacity(waves);
    // geometry classes: (st1, st2) — at most two per launch (the entry points split a launch that has more)
    int ncls = 0, cst1[2] = {0, 0}, cst2[2] = {0, 0};
    for (int j = 0; j < mp.njobs; j++) {
        TcParams &p = mp.job[j];
        int c = 0;
        while (c < ncls && (cst1[c] != p.st1 || cst2[c] != p.st2)) c++;
        if (c == ncls) {
            if (ncls == 2) c = 1;  // (never reached through the entry points: gemv_classes_ok)
            else { cst1[c] = p.st1; cst2[c] = p.st2; ncls++; }
        }
        p.cls = c;
    }
    // what the jobs allow: sharing a row needs a zeroed (or accumulated-onto) output — or a memset node (~3 us: worth it only for
    // a large gain); the SwiGLU epilogue needs both rows of an up / gate pair whole inside one workgroup, in runs of equal length
    bool free_share = share_on != 0, swiglu = false;
    for (int j = 0; j < mp.njobs; j++) {
        const TcParams &p = mp.job[j];
        if (!((out_zeroed && out_zeroed[j]) || p.accumulate)) free_share = false;
        swiglu = swiglu || p.act_out != nullptr;
    }
    constexpr int kMemsetCost = 10;  // a memset node in steps of the busiest SIMD (~3 us / 0.3 us)
    constexpr int kRoundCost = 4;    // a workgroup's fixed cost per item, in steps
    // candidates per class
    struct Cand { TapePlan tp; int lg_g, rg; };
    static thread_local Cand cands[2][3 * 64];
    int ncand[2] = {0, 0};
    for (int c = 0; c < ncls; c++)
        for (int lg = 0; lg <= 2; lg++) {
            if (force_g > 0 && (1 << lg) != force_g) continue;
            if (swiglu && lg != 0) continue;
            for (int rg = 1; rg <= 16 << lg && rg <= 64; rg++) {
                if (force_rg > 0 && rg != force_rg) continue;
                if (swiglu && (rg & (rg - 1) || rg < 2 || rg > waves)) continue;
                TapePlan tp = tape_plan(lg, rg, cst1[c], cst2[c], waves);
                if (!tp.ok) continue;
                if (tp.shared && swiglu) continue;
                cands[c][ncand[c]++] = Cand{tp, lg, rg};
            }
        }
    // cheapest combination: rounds x (busiest SIMD over the classes) [+ memset]; ties: no sharing, fewer workgroups per group,
    // more workgroups inside the round (the largest number of waves per row that still fits ONE round: the rule of rounds 1-4)
    long best = -1;
    int bi[2] = {0, 0};
    for (int i0 = 0; i0 < ncand[0]; i0++)
        for (int i1 = 0; i1 < (ncls > 1 ? ncand[1] : 1); i1++) {
            const Cand *cc[2] = {&cands[0][i0], ncls > 1 ? &cands[1][i1] : nullptr};
            int items = 0, cost = 0;
            bool shared = false;
            for (int c = 0; c < ncls; c++) {
                items += (class_rows(mp, c, cc[c]->rg) + cc[c]->rg - 1) / cc[c]->rg << cc[c]->lg_g;
                if (cc[c]->tp.cost > cost) cost = cc[c]->tp.cost;
                shared = shared || cc[c]->tp.shared;
            }
            const int rounds = (items + cap - 1) / cap;
            // (every round pays a workgroup's prologue again: image and x staged, two barriers — about four steps' worth)
            long score = (long)rounds * (cost + kRoundCost) + (shared && !free_share ? kMemsetCost : 0);
            score = score * 4 + (shared ? 1 : 0);
            score = score * 8 + cc[0]->lg_g + (cc[1] ? cc[1]->lg_g : 0);
            score = score * 1024 + (1023 - (items < 1023 ? items : 1023));
            if (best < 0 || score < best) {
                best = score;
                bi[0] = i0;
                bi[1] = i1;
            }
        }
    // the launch's table
    mp.ncls = ncls;
    mp.cls_mask = 0;
    int items_c[2] = {0, 0};
    for (int c = 0; c < ncls; c++) {
        const Cand &cd = cands[c][bi[c]];
        mp.plan[c] = cd.tp.lp;
        int vrow = 0;
        for (int j = 0; j < mp.njobs; j++) {
            TcParams &p = mp.job[j];
            if (p.cls != c) continue;
            if (c) mp.cls_mask |= 1 << j;
            p.vrow0 = vrow;
            vrow += (p.nrows + cd.rg - 1) / cd.rg * cd.rg;
            mp.row_end[j] = vrow;
            p.sk = cd.tp.shared ? 2 : 1;  // (> 1: the output must start at zero — zero_split_jobs() adds the memset the caller did not)
            p.nitems = 0;
            if (p.act_out) {              // the lead wave of an up row also finishes the gate row behind it
                for (int w = 0; w < waves; w++) {
                    WaveEnt &e = mp.plan[c].w[0][w];
                    if ((e.a >> 9 & 1u) && !((e.a & 255u) & 1u)) e.a |= 1u << 17;
                }
            }
        }
        items_c[c] = (vrow + cd.rg - 1) / cd.rg << cd.lg_g;
    }
    for (int j = mp.njobs; j < kMaxJobs; j++) mp.row_end[j] = 0x7fffffff;
    mp.items0 = items_c[0];
    mp.total_items = items_c[0] + items_c[1];
    for (int j = 0; j < kMaxJobs; j++) mp.item_end[j] = mp.total_items;
    grid = mp.total_items < cap ? mp.total_items : cap;
    static const int plan_log = env_int("QPAL_PLAN_LOG", 0);  // one line per planned GEMV launch on stderr (tests, debugging)
    if (plan_log) {
        fprintf(stderr, "[qpal plan] gemv: %d jobs, grid %d:", mp.njobs, grid);
        for (int j = 0; j < mp.njobs; j++) {
            const Cand &cd = cands[mp.job[j].cls][bi[mp.job[j].cls]];
            fprintf(stderr, " [rows %d steps %d+%d groups of %d x %d rows, %s, busiest SIMD %d steps]", mp.job[j].nrows, mp.job[j].st1,
                    mp.job[j].st2, 1 << cd.lg_g, cd.rg, cd.tp.shared ? "shared rows" : "whole rows", cd.tp.cost);
        }
        fprintf(stderr, "\n");
    }
}

// one GEMV launch holds at most two geometry classes (single- and two-stream jobs of an any-KV launch, say)
bool gemv_classes_ok(const TcMultiParams &mp) {
    int ncls = 0, a1[2], a2[2];
    for (int j = 0; j < mp.njobs; j++) {
        int c = 0;
        while (c < ncls && (a1[c] != mp.job[j].st1 || a2[c] != mp.job[j].st2)) c++;
        if (c == ncls) {
            if (ncls == 2) return false;
            a1[c] = mp.job[j].st1;
            a2[c] = mp.job[j].st2;
            ncls++;
        }
    }
    return true;
}

// Geometry of the lockstep skinny-GEMM kernel (tc_gemm.h, batch > 8): a workgroup = 8 waves = 8 supertile rows that walk one K
// range together; K is split over workgroups (float atomics into a zeroed output) only while the launch has too few row groups
// to occupy the chip and every split keeps >= kGemmMinSteps steps.  QPAL_GEMM_SK=<n> forces the split (experiments).
constexpr int kGemmMinSteps = 4;
// two_per_cu: one batch group beside a 64 KiB codebook image (75 KiB of LDS, < 128 VGPRs): two workgroups share a CU — 4 waves per
// SIMD hide each other's barrier waits — and the launch is planned for twice the items (batch 4 / 8: +3 % tokens/s, measured).
void plan_gemm(TcMultiParams &mp, int &grid, bool two_per_cu = false, int wg_rows = kGemmWaves) {
    static const int force_sk = env_int("QPAL_GEMM_SK", 0);
    static const int items_env = env_int("QPAL_GEMM_ITEMS", 0);  // measured: 96 / 192 / 256 / 512 -> 4.9 / 4.1 / 4.5 / 5.3 ms per Llama-8B step at batch 64
    static const int mult_env = env_int("QPAL_GEMM_GRID_MULT", 0);
    const int grid_mult = mult_env > 0 ? mult_env : (two_per_cu ? 2 : 1);
    const int want_items = items_env > 0 ? items_env : 192 * grid_mult;
    static const int min_steps = env_int("QPAL_GEMM_MINSTEPS", kGemmMinSteps);
    int groups = 0;
    for (int j = 0; j < mp.njobs; j++) groups += (mp.job[j].nrows + wg_rows - 1) / wg_rows;
    int total = 0;
    for (int j = 0; j < kMaxJobs; j++) {
        if (j < mp.njobs) {
            TcParams &p = mp.job[j];
            const int T = p.st1 + p.st2;
            int sk = 1;
            while (groups * sk < want_items && T / (sk * 2) >= min_steps && sk < 32) sk *= 2;
            if (force_sk > 0) {  // (a power of two: the kernel shifts)
                sk = 1;
                while (sk * 2 <= force_sk) sk *= 2;
            }
            while (sk > 1 && sk > T) sk >>= 1;
            p.sk = sk;
            p.nitems = ((p.nrows + wg_rows - 1) / wg_rows) * sk;
            total += p.nitems;
        }
        mp.item_end[j] = total;
    }
    mp.total_items = total;
    grid = total < kNumCU * grid_mult ? total : kNumCU * grid_mult;
}

// Few-row launches at a wide batch (lockstep kernel): the chip is filled by splitting K, and every split adds m * n * 4 bytes of
// float atomics (o / down of Llama-8B at batch 64: 16 row groups -> K split 4 / 8, 64 / 128 of 256 CUs busy, the atomics 13 % of a
// token).  Slices of the BATCH are independent instead — no reduction at all, the decode (cheap beside 8 batch groups of MFMAs
// and B-fragment reads) repeated per slice: the launch runs them as jobs of their own (same weights, x and out moved by the
// slice's first row), then K is split as before.  Returns the rows of the widest slice.  QPAL_GEMM_BS=<n> forces n slices.
int slice_gemm_batch(TcMultiParams &mp, int *ms, int *zeroed) {
    static const int force = env_int("QPAL_GEMM_BS", 0);
    static const int min_rows = env_int("QPAL_GEMM_BS_MIN_ROWS", 16);
    static const int want_items = env_int("QPAL_GEMM_ITEMS", 192);
    const int n = mp.job[0].n;
    int groups = 0;
    for (int j = 0; j < mp.njobs; j++) groups += (mp.job[j].nrows + kGemmWaves - 1) / kGemmWaves;
    // measured (Llama-8B tcomb_6_7, tokens/s without -> with, profiles/r03_gemm_batch_slices.txt): two slices of the few-row launches
    // (q|k|v, o, down) 64: 17 500 -> 18 870, 32: 13 270 -> 13 440; FOUR slices (16 rows each at 64) 17 650, slicing gate|up as well
    // 18 490, slices of 8 rows -9...-14 %: at most two, never below 16 rows
    int bs = 1;
    while (bs < (force > 0 ? force : 2) && mp.njobs * bs * 2 <= kMaxJobs && n / (bs * 2) >= min_rows &&
           (force > 0 || groups * bs * 2 <= want_items))
        bs *= 2;
    if (bs == 1) return n;
    const int rows = ((n + bs - 1) / bs + 15) & ~15;  // whole groups of 16 batch rows per slice
    const TcMultiParams src = mp;
    int ms0[kMaxJobs], z0[kMaxJobs];
    for (int j = 0; j < kMaxJobs; j++) {
        ms0[j] = ms[j];
        z0[j] = zeroed[j];
    }
    int nj = 0;
    for (int j = 0; j < src.njobs; j++)
        for (int b0 = 0; b0 < n; b0 += rows) {
            TcParams q = src.job[j];
            q.n = n - b0 < rows ? n - b0 : rows;
            q.x = q.x + (long)b0 * q.k;
            q.out = q.out + (long)b0 * q.ldo;
            mp.job[nj] = q;
            ms[nj] = ms0[j];
            zeroed[nj] = z0[j];
            nj++;
        }
    mp.njobs = nj;
    return rows;
}

// batches 9..64: which kernel.  QPAL_GEMM=0: the per-wave-K-chunk kernel (tc_gemv_kernel, x from L2 per wave);
// 1 (default): the lockstep kernel with the step's activations shared through LDS (tc_gemm.h)
bool use_gemm(int nbg, const TcMultiParams &mp) {
    static const int on = env_int("QPAL_GEMM", 1);
    // the lockstep kernel from this batch on (measured, Llama-8B tcomb_6_7, ms per step old / new: batch 2 1.38 / 1.50, 3 1.48 / 1.51,
    // 4 1.68 / 1.51, 6 1.89 / 1.53, 8 2.09 / 1.55: profiles/r03_sweep_minbatch.txt)
    static const int min_n = env_int("QPAL_GEMM_MIN_BATCH", 4);
    if (!on || mp.job[0].n < min_n) return false;
    (void)nbg;
    for (int j = 0; j < mp.njobs; j++) {
        const TcParams &p = mp.job[j];
        if (!aligned(p.x, 16) || (p.k % 8) || p.x_rot || p.act_out) return false;
    }
    return true;
}

int lut_image_bytes(int bits, int vec) {
    const int idx = vec == 2 ? bits : (bits <= 6 ? 2 * bits : bits);
    const int log2c = (15 - idx) < 5 ? (15 - idx) : 5;
    return 4 << (idx + log2c);
}

// fills one job's parameters + geometry; returns its item count through p.nitems
void tcq_fill(TcParams &p, float *out, long ldo, const void *c1, const void *c2, const void *x, const void *tlut, int m,
              int n, int k, int k1, int k2, const void *wscale = nullptr, float oscale = 1.0f) {
    p = TcParams{};
    p.wscale = static_cast<const uint16_t *>(wscale);
    p.oscale = oscale == 0.0f ? 1.0f : oscale;
    p.out = out;
    p.ldo = ldo;
    p.c1 = static_cast<const uint32_t *>(c1);
    p.c2 = static_cast<const uint32_t *>(c2);
    p.x = static_cast<const uint16_t *>(x);
    p.tab = tlut;
    p.n = n;
    p.k = k;
    p.nrows = m / 32;
    p.nsc1 = k1 / 32;
    p.nsc2 = k2 / 32;
    p.st1 = (p.nsc1 + 3) / 4;
    p.st2 = (p.nsc2 + 3) / 4;
    p.col2 = k1;
    p.x_lds = x_fits_lds(n, k);
}

// Decode-to-fp16 work items: supertile rows cut into column chunks until there are >= 8 items per CU (a row-only grid left
// half of the chip idle for m = 4096 and 12 % at m = 14336); every wave still needs a step of its own: chunk >= 16 steps
int dequant_chunks(int nrows, int st1, int st2) {
    int st = st1;
    if (st2 > 0 && st2 < st) st = st2;
    int nch = 1;
    while (nrows * nch < 8 * kNumCU && st / (nch * 2) >= 16) nch *= 2;
    return nch;
}

// An accumulating job (out += y: the residual add of a decoder block) is never zeroed: its split-K atomics add onto what
// `out` already holds, which IS the accumulation (a memset here would silently turn h + y into y).
int zero_if_split(const TcParams &p, int m, hipStream_t stream, int out_zeroed = 0) {
    if (p.sk > 1 && !out_zeroed && !p.accumulate) {
        // ONE node for the whole [n][m] block (a batch of 64 used to cost 64 memset nodes per layer)
        hipError_t e = p.ldo == m ? hipMemsetAsync(p.out, 0, sizeof(float) * (size_t)m * p.n, stream)
                                  : hipMemset2DAsync(p.out, sizeof(float) * (size_t)p.ldo, 0, sizeof(float) * (size_t)m, (size_t)p.n, stream);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

int launch_tcq_gemm(const TcMultiParams &mp, int S, int KV1, int KV2, int nbg, int grid, hipStream_t stream) {
    return nbg == 1 ? launch_tcq_gemm_nbg1(mp, S, KV1, KV2, grid, stream)
         : nbg == 2 ? launch_tcq_gemm_nbg2(mp, S, KV1, KV2, grid, stream)
         : nbg == 4 ? launch_tcq_gemm_nbg4(mp, S, KV1, KV2, grid, stream)
         : nbg == 8 ? launch_tcq_gemm_nbg8(mp, S, KV1, KV2, grid, stream)
         : nbg == 10 ? launch_tcq_gemm_nbg10(mp, S, KV1, KV2, grid, stream)
                    : launch_tcq_gemm_nbg16(mp, S, KV1, KV2, grid, stream);
}

// Zeroing of the split-K outputs of a multi-job launch: contiguous [n][m] outputs that follow each other in memory (how
// qpalette_amd.ops allocates the outputs of one launch) are zeroed by ONE memset node — a node costs ~5 us inside a graph, more
// than filling a megabyte.
int zero_split_jobs(const TcMultiParams &mp, const int *ms, const int *zeroed, hipStream_t stream) {
    struct Span { char *p; size_t bytes; };
    Span spans[kMaxJobs];
    int ns = 0;
    for (int j = 0; j < mp.njobs; j++) {
        const TcParams &p = mp.job[j];
        if (p.act_out || p.sk <= 1 || zeroed[j] || p.accumulate) continue;
        if (p.ldo != ms[j]) {  // a column block of a wider buffer: its own 2-D fill
            int rc = zero_if_split(p, ms[j], stream, 0);
            if (rc) return rc;
            continue;
        }
        spans[ns++] = Span{reinterpret_cast<char *>(p.out), sizeof(float) * (size_t)ms[j] * p.n};
    }
    for (int i = 1; i < ns; i++)  // insertion sort by address
        for (int k = i; k > 0 && spans[k].p < spans[k - 1].p; k--) { Span t = spans[k]; spans[k] = spans[k - 1]; spans[k - 1] = t; }
    for (int i = 0; i < ns;) {
        char *p0 = spans[i].p;
        size_t bytes = spans[i].bytes;
        int k = i + 1;
        while (k < ns && spans[k].p == p0 + bytes) bytes += spans[k++].bytes;
        hipError_t e = hipMemsetAsync(p0, 0, bytes, stream);
        if (e != hipSuccess) return (int)e;
        i = k;
    }
    return 0;
}

int tcq_gemv_one(float *out, long ldo, const void *c1, const void *c2, const void *x, const void *tlut, int m, int n,
                 int k, int k1, int k2, int S, int KV1, int KV2, hipStream_t stream) {
    TcMultiParams mp{};
    mp.njobs = 1;
    tcq_fill(mp.job[0], out, ldo, c1, c2, x, tlut, m, n, k, k1, k2);
    int grid;
    const int nbg = nbg_of(n);
    if (use_gemm(nbg, mp)) {
        int ms[kMaxJobs] = {m}, zeroed[kMaxJobs] = {0};
        const int rows = slice_gemm_batch(mp, ms, zeroed);
        plan_gemm(mp, grid, nbg_of(rows) <= env_int("QPAL_GEMM_TWO_NBG", 1) && S == 9, nbg_of(rows) >= 4 ? kG16Waves : kGemmWaves);
        int rc = zero_split_jobs(mp, ms, zeroed, stream);
        if (rc) return rc;
        return launch_tcq_gemm(mp, S, KV1, KV2, nbg_of(rows), grid, stream);
    }
    if (nbg > 8) return QPAL_E_SHAPE;  // (batches 65..128: the lockstep kernel only — it needs 16-byte aligned x and k % 8 == 0)
    plan_launch(mp, nullptr, grid, waves_of(nbg));
    int rc = zero_if_split(mp.job[0], m, stream);
    if (rc) return rc;
    return launch_tcq_gemv(mp, S, KV1, KV2, nbg, grid, stream);
}

int tcq_check(const void *c1, const void *c2, const void *tlut, int m, int k, int S, int KV1, int KV2, int split) {
    if (!c1 || !tlut) return QPAL_E_NULL;
    if (split != QPAL_SPLIT_NONE && !c2) return QPAL_E_NULL;
    if (m <= 0 || k <= 0 || m % 32 || k % 32) return QPAL_E_SHAPE;
    if (split == QPAL_SPLIT_ROWS && (m % 64)) return QPAL_E_SHAPE;
    if (split == QPAL_SPLIT_COLS && (k % 64)) return QPAL_E_SHAPE;
    if (split < 0 || split > 2) return QPAL_E_PARAM;
    if (!tcq_pair_ok(S, KV1)) return QPAL_E_PARAM;
    if (split != QPAL_SPLIT_NONE && (!tcq_pair_ok(S, KV2) || KV2 != KV1 + 1)) return QPAL_E_PARAM;
    if (!aligned(c1, 4) || (c2 && !aligned(c2, 4)) || !aligned(tlut, 4)) return QPAL_E_ALIGN;
    return QPAL_OK;
}

}  // namespace

extern "C" {

int qpal_tcq_gemv(float *out, const void *c1, const void *c2, const void *x, const void *tlut, int m, int n, int k,
                  int S, int KV1, int KV2, int split, void *stream) {
    int rc = tcq_check(c1, c2, tlut, m, k, S, KV1, KV2, split);
    if (rc) return rc;
    if (!out || !x) return QPAL_E_NULL;
    if (n < 1 || n > kMaxBatch) return QPAL_E_SHAPE;
    if (!aligned(x, 8) || !aligned(out, 4) || (k % 4)) return QPAL_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (split == QPAL_SPLIT_NONE) return tcq_gemv_one(out, m, c1, nullptr, x, tlut, m, n, k, k, 0, S, KV1, 0, s);
    if (split == QPAL_SPLIT_COLS)
        return tcq_gemv_one(out, m, c1, c2, x, tlut, m, n, k, k / 2, k / 2, S, KV1, KV2, s);
    // row halves: two independent single-stream problems writing the two halves of each output row
    rc = tcq_gemv_one(out, m, c1, nullptr, x, tlut, m / 2, n, k, k, 0, S, KV1, 0, s);
    if (rc) return rc;
    return tcq_gemv_one(out + m / 2, m, c2, nullptr, x, tlut, m / 2, n, k, k, 0, S, KV2, 0, s);
}

int qpal_tcq_gemv_multi(const qpal_tcq_job *jobs, int njobs, int n, int S, int KV1, int KV2, int split, void *prezero,
                        long prezero_bytes, void *stream) {
    if (!jobs) return QPAL_E_NULL;
    if (njobs < 1 || njobs > kMaxJobs) return QPAL_E_SHAPE;
    if (prezero_bytes < 0 || prezero_bytes % 16 || (prezero_bytes && (!prezero || !aligned(prezero, 16)))) return QPAL_E_ALIGN;
    if (split == QPAL_SPLIT_ROWS) return QPAL_E_PARAM;  // the two row halves use different codecs: one call each
    if (n < 1 || n > kMaxBatch) return QPAL_E_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    {   // a launch holds at most two row geometries (k, single / two streams): more -> one launch per geometry
        long keys[kMaxJobs];
        int nkeys = 0;
        for (int j = 0; j < njobs; j++) {
            const long key = (long)jobs[j].k * 2 + ((split == QPAL_SPLIT_COLS || (jobs[j].kv && jobs[j].kv2 && jobs[j].c2)) ? 1 : 0);
            int i = 0;
            while (i < nkeys && keys[i] != key) i++;
            if (i == nkeys) keys[nkeys++] = key;
        }
        if (nkeys > 2) {
            for (int i = 0; i < nkeys; i++) {
                qpal_tcq_job sub[kMaxJobs];
                int ns = 0;
                for (int j = 0; j < njobs; j++)
                    if ((long)jobs[j].k * 2 + ((split == QPAL_SPLIT_COLS || (jobs[j].kv && jobs[j].kv2 && jobs[j].c2)) ? 1 : 0) == keys[i]) sub[ns++] = jobs[j];
                int rc = qpal_tcq_gemv_multi(sub, ns, n, S, KV1, KV2, split, i == 0 ? prezero : nullptr, i == 0 ? prezero_bytes : 0, stream);
                if (rc) return rc;
            }
            return QPAL_OK;
        }
    }
    TcMultiParams mp{};
    mp.njobs = njobs;
    int zeroed[kMaxJobs] = {0};
    bool mixed = false;  // jobs with their own KV: the any-KV kernel of this S
    for (int j = 0; j < njobs; j++) {
        const qpal_tcq_job &jb = jobs[j];
        const int kv = jb.kv ? jb.kv : KV1;
        const bool job_two = split == QPAL_SPLIT_NONE && jb.kv && jb.kv2 && jb.c2;  // column-split layer in an any-KV launch
        mixed = mixed || kv != KV1 || job_two;
        if ((kv != KV1 || job_two) && (split != QPAL_SPLIT_NONE || jb.x_had)) return QPAL_E_PARAM;
        if (jb.kv2 && !job_two) return QPAL_E_PARAM;
        int rc = job_two ? tcq_check(jb.c1, jb.c2, jb.tlut, jb.m, jb.k, S, kv, jb.kv2, QPAL_SPLIT_COLS)
                         : tcq_check(jb.c1, jb.c2, jb.tlut, jb.m, jb.k, S, kv, KV2, split);
        if (rc) return rc;
        if ((!jb.out && !jb.act_out) || (!jb.x && !(jb.x_f32 && jb.x_had))) return QPAL_E_NULL;
        if (jb.x_f32 && !jb.x_had) return QPAL_E_PARAM;  // fp32 input: rotation staging only
        if (jb.act_out && (!jb.x_had || n != 1 || jb.accumulate || (jb.m % 64) || !aligned(jb.act_out, 2))) return QPAL_E_PARAM;
        if ((jb.x && !aligned(jb.x, 8)) || !aligned(jb.out, 4) || (jb.k % 4)) return QPAL_E_ALIGN;
        if (jb.wscale && !aligned(jb.wscale, 2)) return QPAL_E_ALIGN;
        if (jb.ldo != 0 && jb.ldo < jb.m) return QPAL_E_SHAPE;
        const long ldo = jb.ldo ? jb.ldo : jb.m;
        if (split == QPAL_SPLIT_NONE && !job_two)
            tcq_fill(mp.job[j], jb.out, ldo, jb.c1, nullptr, jb.x, jb.tlut, jb.m, n, jb.k, jb.k, 0, jb.wscale, jb.oscale);
        else
            tcq_fill(mp.job[j], jb.out, ldo, jb.c1, jb.c2, jb.x, jb.tlut, jb.m, n, jb.k, jb.k / 2, jb.k / 2, jb.wscale,
                     jb.oscale);
        mp.job[j].kv = kv;
        mp.job[j].kv2 = job_two ? jb.kv2 : 0;
        rc = set_rotation(mp.job[j], jb.x_had, jb.x_su, jb.x_post, n, jb.k, jb.x_f32, jb.x_rms_eps, jb.x_rms_w, jb.x_hadk, jb.x_K);
        if (rc) return rc;
        mp.job[j].accumulate = jb.accumulate ? 1 : 0;
        mp.job[j].act_out = static_cast<uint16_t *>(jb.act_out);
        mp.job[j].act_su = jb.act_out ? static_cast<const uint16_t *>(jb.act_su) : nullptr;
        zeroed[j] = jb.act_out ? 1 : jb.out_zeroed;  // (nothing to zero: `out` is not written; split-K is refused below)
    }
    // the kernel stages (or rotates) x once per distinct x pointer: jobs that share x must agree on how
    for (int j = 1; j < njobs; j++)
        for (int i = 0; i < j; i++)
            if (mp.job[i].x == mp.job[j].x &&
                (jobs[i].x_had != jobs[j].x_had || jobs[i].x_su != jobs[j].x_su || jobs[i].x_post != jobs[j].x_post ||
                 jobs[i].k != jobs[j].k || jobs[i].x_rms_eps != jobs[j].x_rms_eps || jobs[i].x_rms_w != jobs[j].x_rms_w ||
                 jobs[i].x_K != jobs[j].x_K || jobs[i].x_hadk != jobs[j].x_hadk))
                return QPAL_E_PARAM;
    for (int j = 1; j < njobs; j++)  // one kernel per launch: the 14336-wide rotation has its own instantiation
        if ((mp.job[j].x_rot == kK28) != (mp.job[0].x_rot == kK28)) return QPAL_E_PARAM;
    if (mixed) {
        if (n > 8) return QPAL_E_SHAPE;
        for (int j = 0; j < njobs; j++) {
            for (const int kv : {mp.job[j].kv, mp.job[j].kv2})
                if (kv && ((S == 9 && kv > 8) || (S == 10 && kv < 8) || (S == 11 && kv < 9))) return QPAL_E_PARAM;
        }
    }
    mp.zero = static_cast<u32x4 *>(prezero);
    mp.zero_chunks = (int)(prezero_bytes / 16);
    int grid;
    const int nbg = nbg_of(n);
    if (nbg >= 4)
        for (int j = 0; j < njobs; j++)
            if (jobs[j].x_had) return QPAL_E_SHAPE;  // fused rotation: batch 1
    const bool gemm = !mixed && use_gemm(nbg, mp);
    int ms[kMaxJobs] = {0};
    if (gemm) {
        for (int j = 0; j < njobs; j++) {
            ms[j] = jobs[j].m;
            zeroed[j] = jobs[j].out_zeroed;
        }
        const int rows = slice_gemm_batch(mp, ms, zeroed);
        plan_gemm(mp, grid, nbg_of(rows) <= env_int("QPAL_GEMM_TWO_NBG", 1) && S == 9, nbg_of(rows) >= 4 ? kG16Waves : kGemmWaves);
        int rc = zero_split_jobs(mp, ms, zeroed, s);
        if (rc) return rc;
        return launch_tcq_gemm(mp, S, KV1, split == QPAL_SPLIT_NONE ? 0 : KV2, nbg_of(rows), grid, s);
    }
    if (nbg > 8) return QPAL_E_SHAPE;  // (batches 65..128: the lockstep kernel only)
    plan_launch(mp, zeroed, grid, waves_of(nbg));
    for (int j = 0; j < njobs; j++) {
        ms[j] = jobs[j].m;
        zeroed[j] = jobs[j].out_zeroed;
        // the SwiGLU epilogue pairs two supertile rows of one workgroup and has no shared-row form (the planner keeps to that)
        if (mp.job[j].act_out && (mp.job[j].sk != 1 || mp.plan[mp.job[j].cls].rg < 2)) return QPAL_E_SHAPE;
    }
    {
        int rc = zero_split_jobs(mp, ms, zeroed, s);
        if (rc) return rc;
    }
    if (mixed) return launch_tcq_gemv_any(mp, S, grid, s);
    return launch_tcq_gemv(mp, S, KV1, split == QPAL_SPLIT_NONE ? 0 : KV2, nbg, grid, s);
}

int qpal_tcq_dequant(void *out_f16, const void *c1, const void *c2, const void *tlut, int m, int k, int S, int KV1,
                     int KV2, int split, void *stream) {
    int rc = tcq_check(c1, c2, tlut, m, k, S, KV1, KV2, split);
    if (rc) return rc;
    if (!out_f16) return QPAL_E_NULL;
    if (!aligned(out_f16, 8)) return QPAL_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto one = [&](uint16_t *w, const void *a, const void *b, int mm, int k1, int k2, int kva, int kvb) -> int {
        TcParams p{};
        p.wout = w;
        p.ldw = k;
        p.c1 = static_cast<const uint32_t *>(a);
        p.c2 = static_cast<const uint32_t *>(b);
        p.tab = tlut;
        p.nrows = mm / 32;
        p.nsc1 = k1 / 32;
        p.nsc2 = k2 / 32;
        p.st1 = (p.nsc1 + 3) / 4;
        p.st2 = (p.nsc2 + 3) / 4;
        p.col2 = k1;
        p.x_lds = aligned(w, 16) && (k % 8) == 0 && (k1 % 8) == 0 && !getenv("QPAL_DEQUANT_DIRECT");  // staged 16-byte stores
        p.sk = dequant_chunks(p.nrows, p.st1, p.st2);
        const int items = p.x_lds ? p.nrows * p.sk : p.nrows;
        const int grid = items < kNumCU ? items : kNumCU;
        return launch_tcq_dequant(p, S, kva, kvb, grid, s);
    };
    uint16_t *w = static_cast<uint16_t *>(out_f16);
    if (split == QPAL_SPLIT_NONE) return one(w, c1, nullptr, m, k, 0, KV1, 0);
    if (split == QPAL_SPLIT_COLS) return one(w, c1, c2, m, k / 2, k / 2, KV1, KV2);
    rc = one(w, c1, nullptr, m / 2, k, 0, KV1, 0);
    if (rc) return rc;
    return one(w + (size_t)(m / 2) * k, c2, nullptr, m / 2, k, 0, KV2, 0);
}

static void lut_fill(TcParams &p, float *out, long ldo, const void *qweight, const void *x, const void *lut, int m, int n,
                     int k, const void *wscale, float oscale) {
    p = TcParams{};
    p.wscale = static_cast<const uint16_t *>(wscale);
    p.oscale = oscale == 0.0f ? 1.0f : oscale;
    p.out = out;
    p.ldo = ldo;
    p.c1 = static_cast<const uint32_t *>(qweight);
    p.x = static_cast<const uint16_t *>(x);
    p.tab = lut;
    p.n = n;
    p.k = k;
    p.nrows = m / 32;
    p.nsc1 = k / 32;
    p.st1 = (p.nsc1 + 3) / 4;
    p.x_lds = x_fits_lds(n, k);
}

static int lut_args_ok(const void *out, const void *qweight, const void *x, const void *lut, int m, int n, int k, int bits,
                       int vec) {
    if (!out || !qweight || !x || !lut) return QPAL_E_NULL;
    if (m <= 0 || k <= 0 || m % 32 || k % 32 || n < 1 || n > kMaxBatch) return QPAL_E_SHAPE;
    if (!lut_tc_ok(bits, vec) || ((long)bits * k) % (32 * vec)) return QPAL_E_PARAM;
    if (!aligned(qweight, 4) || !aligned(x, 8) || !aligned(lut, 4) || !aligned(out, 4)) return QPAL_E_ALIGN;
    return QPAL_OK;
}

int qpal_lut_tc_gemv(float *out, const void *qweight, const void *x, const void *lut, int m, int n, int k, int bits,
                     int vec, void *stream) {
    qpal_lut_job job{};
    job.out = out; job.qweight = qweight; job.x = x; job.lut = lut; job.m = m; job.k = k; job.oscale = 1.0f; job.x_post = 1.0f;
    return qpal_lut_tc_gemv_multi(&job, 1, n, bits, vec, nullptr, 0, stream);
}

int qpal_lut_tc_gemv_multi(const qpal_lut_job *jobs, int njobs, int n, int bits, int vec, void *prezero,
                           long prezero_bytes, void *stream) {
    if (!jobs) return QPAL_E_NULL;
    if (njobs < 1 || njobs > kMaxJobs) return QPAL_E_SHAPE;
    if (prezero_bytes < 0 || prezero_bytes % 16 || (prezero_bytes && (!prezero || !aligned(prezero, 16)))) return QPAL_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    {   // a launch holds at most two row geometries (here: k): more -> one launch per k
        int keys[kMaxJobs], nkeys = 0;
        for (int j = 0; j < njobs; j++) {
            int i = 0;
            while (i < nkeys && keys[i] != jobs[j].k) i++;
            if (i == nkeys) keys[nkeys++] = jobs[j].k;
        }
        if (nkeys > 2) {
            for (int i = 0; i < nkeys; i++) {
                qpal_lut_job sub[kMaxJobs];
                int ns = 0;
                for (int j = 0; j < njobs; j++)
                    if (jobs[j].k == keys[i]) sub[ns++] = jobs[j];
                int rc = qpal_lut_tc_gemv_multi(sub, ns, n, bits, vec, i == 0 ? prezero : nullptr, i == 0 ? prezero_bytes : 0, stream);
                if (rc) return rc;
            }
            return QPAL_OK;
        }
    }
    TcMultiParams mp{};
    mp.njobs = njobs;
    int zeroed[kMaxJobs] = {0};
    for (int j = 0; j < njobs; j++) {
        const qpal_lut_job &jb = jobs[j];
        int rc = lut_args_ok(jb.out ? jb.out : jb.act_out, jb.qweight, jb.x ? jb.x : (jb.x_had ? jb.x_f32 : nullptr), jb.lut, jb.m, n, jb.k, bits, vec);
        if (rc) return rc;
        if (jb.x_f32 && !jb.x_had) return QPAL_E_PARAM;  // fp32 input: rotation staging only
        if (jb.act_out && (!jb.x_had || n != 1 || jb.accumulate || (jb.m % 64) || !aligned(jb.act_out, 2))) return QPAL_E_PARAM;
        if (jb.wscale && !aligned(jb.wscale, 2)) return QPAL_E_ALIGN;
        if (jb.ldo != 0 && jb.ldo < jb.m) return QPAL_E_SHAPE;
        lut_fill(mp.job[j], jb.out, jb.ldo ? jb.ldo : jb.m, jb.qweight, jb.x, jb.lut, jb.m, n, jb.k, jb.wscale, jb.oscale);
        rc = set_rotation(mp.job[j], jb.x_had, jb.x_su, jb.x_post, n, jb.k, jb.x_f32, jb.x_rms_eps, jb.x_rms_w, jb.x_hadk, jb.x_K,
                          lut_image_bytes(bits, vec));
        if (rc) return rc;
        mp.job[j].accumulate = jb.accumulate ? 1 : 0;
        mp.job[j].act_out = static_cast<uint16_t *>(jb.act_out);
        mp.job[j].act_su = jb.act_out ? static_cast<const uint16_t *>(jb.act_su) : nullptr;
        zeroed[j] = jb.act_out ? 1 : jb.out_zeroed;
    }
    for (int j = 1; j < njobs; j++)
        for (int i = 0; i < j; i++)
            if (mp.job[i].x == mp.job[j].x &&
                (jobs[i].x_had != jobs[j].x_had || jobs[i].x_su != jobs[j].x_su || jobs[i].x_post != jobs[j].x_post ||
                 jobs[i].k != jobs[j].k || jobs[i].x_rms_eps != jobs[j].x_rms_eps || jobs[i].x_rms_w != jobs[j].x_rms_w ||
                 jobs[i].x_K != jobs[j].x_K || jobs[i].x_hadk != jobs[j].x_hadk))
                return QPAL_E_PARAM;
    for (int j = 1; j < njobs; j++)
        if ((mp.job[j].x_rot == kK28) != (mp.job[0].x_rot == kK28)) return QPAL_E_PARAM;
    mp.zero = static_cast<u32x4 *>(prezero);
    mp.zero_chunks = (int)(prezero_bytes / 16);
    int grid;
    const int nbg = nbg_of(n);
    // the reduction buffer of 8 batch groups (64 KiB) does not fit beside a 128 KiB codebook image
    if (nbg >= 10 && lut_image_bytes(bits, vec) > 64 * 1024) return QPAL_E_SHAPE;  // (128 KiB image: the LDS holds the x tiles of 64 batch rows)
    const bool gemm = use_gemm(nbg, mp);
    int ms[kMaxJobs] = {0};
    if (gemm) {
        for (int j = 0; j < njobs; j++) {
            ms[j] = jobs[j].m;
            zeroed[j] = jobs[j].out_zeroed;
        }
        const int rows = slice_gemm_batch(mp, ms, zeroed);
        plan_gemm(mp, grid, nbg_of(rows) <= env_int("QPAL_GEMM_TWO_NBG", 1) && lut_image_bytes(bits, vec) <= 64 * 1024, nbg_of(rows) >= 4 ? kG16Waves : kGemmWaves);
        int rc = zero_split_jobs(mp, ms, zeroed, s);
        if (rc) return rc;
        return launch_lut_tc_gemm(mp, bits, vec, nbg_of(rows), grid, s);
    }
    if (nbg > 8) return QPAL_E_SHAPE;  // (batches 65..128: the lockstep kernel only)
    plan_launch(mp, zeroed, grid, waves_of(nbg));
    for (int j = 0; j < njobs; j++) {
        ms[j] = jobs[j].m;
        zeroed[j] = jobs[j].out_zeroed;
        if (mp.job[j].act_out && (mp.job[j].sk != 1 || mp.plan[mp.job[j].cls].rg < 2)) return QPAL_E_SHAPE;
    }
    {
        int rc = zero_split_jobs(mp, ms, zeroed, s);
        if (rc) return rc;
    }
    return launch_lut_tc_gemv(mp, bits, vec, nbg, grid, s);
}

int qpal_lut_tc_dequant(void *out_f16, const void *qweight, const void *lut, int m, int k, int bits, int vec,
                        void *stream) {
    if (!out_f16 || !qweight || !lut) return QPAL_E_NULL;
    if (m <= 0 || k <= 0 || m % 32 || k % 32) return QPAL_E_SHAPE;
    if (!lut_tc_ok(bits, vec) || ((long)bits * k) % (32 * vec)) return QPAL_E_PARAM;
    if (!aligned(qweight, 4) || !aligned(lut, 4) || !aligned(out_f16, 8)) return QPAL_E_ALIGN;
    TcParams p{};
    p.wout = static_cast<uint16_t *>(out_f16);
    p.ldw = k;
    p.c1 = static_cast<const uint32_t *>(qweight);
    p.tab = lut;
    p.nrows = m / 32;
    p.nsc1 = k / 32;
    p.st1 = (p.nsc1 + 3) / 4;
    p.x_lds = aligned(out_f16, 16) && (k % 8) == 0 && !getenv("QPAL_DEQUANT_DIRECT");  // staged 16-byte stores
    p.sk = dequant_chunks(p.nrows, p.st1, 0);
    const int items = p.x_lds ? p.nrows * p.sk : p.nrows;
    const int grid = items < kNumCU ? items : kNumCU;
    return launch_lut_tc_dequant(p, bits, vec, grid, static_cast<hipStream_t>(stream));
}

int qpal_lut_simt_gemv(void *out_f16, const void *qweight, const void *x, const void *lut, int m, int n, int k,
                       int bits, int vec, void *stream) {
    if (!out_f16 || !qweight || !x || !lut) return QPAL_E_NULL;
    if (m <= 0 || k <= 0 || n < 1 || n > 8 || k % (32 * vec)) return QPAL_E_SHAPE;
    if (!simt_ok(bits, vec)) return QPAL_E_PARAM;
    if (!aligned(qweight, 4) || !aligned(x, 16) || !aligned(lut, 2 * vec) || !aligned(out_f16, 2) || (k % 8))
        return QPAL_E_ALIGN;
    SimtParams p{static_cast<uint16_t *>(out_f16), static_cast<const uint32_t *>(qweight),
                 static_cast<const uint16_t *>(x), lut, n, m, k};
    return launch_simt_gemv(p, bits, vec, nb_of(n), simt_gemv_geometry(m, n, k, bits, vec, kNumCU), static_cast<hipStream_t>(stream));
}

int qpal_lut_simt_dequant(void *out_f16, const void *qweight, const void *lut, int m, int k, int bits, int vec,
                          void *stream) {
    if (!out_f16 || !qweight || !lut) return QPAL_E_NULL;
    if (m <= 0 || k <= 0 || k % (32 * vec)) return QPAL_E_SHAPE;
    if (!simt_ok(bits, vec)) return QPAL_E_PARAM;
    if (!aligned(qweight, 4) || !aligned(lut, 2 * vec) || !aligned(out_f16, 16) || (k % 8)) return QPAL_E_ALIGN;
    SimtParams p{static_cast<uint16_t *>(out_f16), static_cast<const uint32_t *>(qweight), nullptr, lut, 1, m, k};
    const int pairs = (m + 1) / 2;
    int grid = (pairs + 15) / 16;
    if (grid > kNumCU) grid = kNumCU;
    return launch_simt_dequant(p, bits, vec, grid, static_cast<hipStream_t>(stream));
}

int qpal_tc_to_simt(void *dst_simt, const void *src_tc, int m, int k, int bits, int vec, void *stream) {
    if (!dst_simt || !src_tc) return QPAL_E_NULL;
    if (m <= 0 || k <= 0 || m % 32 || k % 32 || k % (32 * vec)) return QPAL_E_SHAPE;
    if (!(vec == 1 || vec == 2) || !lut_tc_ok(bits, vec) || ((long)bits * k) % (32 * vec)) return QPAL_E_PARAM;
    if (!aligned(dst_simt, 4) || !aligned(src_tc, 4)) return QPAL_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(dst_simt, 0, (size_t)m * k / vec * bits / 8, s);
    if (e != hipSuccess) return (int)e;
    return launch_tc_to_simt(static_cast<uint32_t *>(dst_simt), static_cast<const uint32_t *>(src_tc), m, k, bits, vec, s);
}

// Launch planner of the fused GEMV, callable without a GPU (tests/test_capi_and_host.py checks its tables: every step of every
// row covered exactly once, at most two workgroups per row, the busiest SIMD).  See include/qpal.h.
int qpal_plan_gemv(const int *rows, const int *steps1, const int *steps2, const int *flags, int njobs, int waves, int shared_staging,
                   int *out, int out_len) {
    if (!rows || !steps1 || !steps2 || !flags || !out) return QPAL_E_NULL;
    if (njobs < 1 || njobs > kMaxJobs || (waves != 16 && waves != 8)) return QPAL_E_SHAPE;
    TcMultiParams mp{};
    mp.njobs = njobs;
    int zeroed[kMaxJobs] = {0};
    static const uint16_t dummy_x[8] = {0};
    for (int j = 0; j < njobs; j++) {
        TcParams &p = mp.job[j];
        p.nrows = rows[j];
        p.st1 = steps1[j];
        p.st2 = steps2[j];
        p.nsc1 = steps1[j] * 4;
        p.nsc2 = steps2[j] * 4;
        p.n = 1;
        p.k = (steps1[j] + steps2[j]) * 128;
        p.x = dummy_x;
        (void)shared_staging;
        zeroed[j] = flags[j] & 1;
        p.accumulate = (flags[j] >> 1) & 1;
        p.act_out = (flags[j] >> 2) & 1 ? reinterpret_cast<uint16_t *>(8) : nullptr;
        if (rows[j] < 1 || steps1[j] < 1 || steps2[j] < 0) return QPAL_E_SHAPE;
    }
    if (!gemv_classes_ok(mp)) return QPAL_E_PARAM;
    int grid = 0;
    plan_launch(mp, zeroed, grid, waves);
    const int need = 8 + 2 * (2 + 2 * kPlanMembers * kPlanWaves) + 4 * njobs;
    if (out_len < need) return QPAL_E_SHAPE;
    int o = 0;
    out[o++] = grid;
    out[o++] = mp.total_items;
    out[o++] = mp.ncls;
    out[o++] = mp.items0;
    out[o++] = 0;  // (groups never run across job boundaries: measured slower, round 5)
    out[o++] = mp.cls_mask;
    out[o++] = kPlanMembers;
    out[o++] = kPlanWaves;
    for (int c = 0; c < 2; c++) {
        out[o++] = mp.plan[c].lg_g;
        out[o++] = mp.plan[c].rg;
        for (int m = 0; m < kPlanMembers; m++)
            for (int w = 0; w < kPlanWaves; w++) {
                out[o++] = (int)mp.plan[c].w[m][w].a;
                out[o++] = (int)mp.plan[c].w[m][w].b;
            }
    }
    for (int j = 0; j < njobs; j++) {
        out[o++] = mp.job[j].cls;
        out[o++] = mp.job[j].vrow0;
        out[o++] = mp.row_end[j];
        out[o++] = mp.job[j].sk;
    }
    return QPAL_OK;
}

int qpal_can_fuse_rotation(int n, int k) { return n >= 1 && k > 0 && rot_ok(n, k) ? 1 : 0; }
int qpal_can_fuse_rotation_k(int n, int k, int K) {
    if (K <= 1) return qpal_can_fuse_rotation(n, k);
    return rot_k28_ok(n, k, K, 64 * 1024) ? 1 : 0;
}

const char *qpal_error_string(int code) {
    switch (code) {
        case QPAL_OK: return "ok";
        case QPAL_E_SHAPE: return "unsupported shape (need m%32==0, k%32==0, 1<=n<=128; see include/qpal.h)";
        case QPAL_E_PARAM: return "unsupported quantizer parameters (S/KV/bits/vec/split)";
        case QPAL_E_NULL: return "null pointer";
        case QPAL_E_ALIGN: return "misaligned pointer";
        default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "unknown qpal error";
    }
}

int qpal_version(void) { return QPAL_VERSION; }

}  // extern "C"
