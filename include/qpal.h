/*
 * qpal.h — C-ABI of the MI355X (gfx950) dequant-matmul hot path of Q-Palette.
 *
 * One runtime-shaped entry point per kernel family replaces the reference's ~8k generated,
 * shape-templated pybind functions.  All pointers are DEVICE pointers owned by the caller (the
 * Python op layer allocates them with the torch caching allocator); the library never allocates,
 * frees or synchronises, launches on the stream it is given (graph-capturable) and reports errors
 * by return code instead of exit() (reference: gpuErrchk -> exit, kernels/tcq-kernels/src/inference.h:10-18).
 *
 * Return value: 0 = ok; < 0 = argument error (QPAL_E_*); > 0 = hipError_t of the failed launch.
 *
 * Data formats (bit for bit the reference's; see DESIGN.md §2 and SURVEY.md §8a):
 *   TCQ trellis   int16  [(m/16)*(k/16)][8*KV]      lib/linear/tcq_linear.py:31-35
 *   TCQ codebook  fp16   [2^S][2]                   lib/linear/tcq_linear.py:37-40
 *   LUT-TC        int32  [m][bits*k/32/vec], fp16 lut [2^bits][vec]   lib/linear/vq_linear.py:15-23
 *   LUT-SIMT      uint32 [m][bits*k/32/vec]         lib/quantizer/pack_op.py:288-335, quant_op.py:69-78
 *   x             fp16   [n][k] row-major, 1 <= n <= 128 (tensor-core-order families), 1 <= n <= 8 (SIMT)
 */
#ifndef QPAL_H
#define QPAL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QPAL_VERSION 300

#define QPAL_OK 0
#define QPAL_E_SHAPE (-1)   /* m, k, n outside the supported set (m%32, k%32, 1<=n<=128 ...) */
#define QPAL_E_PARAM (-2)   /* S / KV / bits / vec / split combination not supported        */
#define QPAL_E_NULL (-3)    /* required pointer is NULL                                      */
#define QPAL_E_ALIGN (-4)   /* pointer not aligned to the format's natural alignment         */

/* split modes of the TCQ family (lib/linear/comb_linear.py) */
#define QPAL_SPLIT_NONE 0   /* QTIPLinearTCQ:  one stream c1 @ KV1                                   */
#define QPAL_SPLIT_ROWS 1   /* CombLinearTCQ:  rows [0,m/2) from c1 @ KV1, rows [m/2,m) from c2 @ KV2 */
#define QPAL_SPLIT_COLS 2   /* CombtLinearTCQ: cols [0,k/2) from c1 @ KV1, cols [k/2,k) from c2 @ KV2 */

/* Fused trellis decode + GEMV:  out[n][m] (fp32) = sum_k W[m][k] * x[n][k].
 * Replaces decompress_gemm_ptr / _comb_ptr / _combt_ptr, kernels/tcq-kernels/src/inference.cu:1826-1968
 * (bindings kernels/tcq-kernels/src/qtip_torch.cu:14-279).  `out` needs no initialisation.
 * S in {9,10,11}; KV in 2..10 per the reference's S/KV table (lib/linear/__init__.py:166-172).   */
int qpal_tcq_gemv(float *out, const void *c1, const void *c2, const void *x, const void *tlut,
                  int m, int n, int k, int S, int KV1, int KV2, int split, void *stream);

/* Several independent TCQ GEMVs of ONE codec (same S, KV1, KV2, split in {NONE, COLS}, same batch n) in a
 * single launch: out/c1/c2/x/tlut/m/k per job.  No counterpart in the reference (it launches one kernel
 * per linear); exists because on MI355X a launch costs ~5 us of fixed time, more than the q/k/v/o GEMVs
 * themselves.  Intended for projections of one input (q|k|v, gate|up); njobs <= 8.                  */
typedef struct qpal_tcq_job {
    float *out;        /* fp32 [n][m] */
    const void *c1;    /* stream 1 */
    const void *c2;    /* stream 2 (split COLS) or NULL */
    const void *x;     /* fp16 [n][k] */
    const void *tlut;  /* fp16 [2^S][2] */
    int m, k;
    int out_zeroed;    /* 1: the caller guarantees out is all zeros (e.g. pre-zeroed by an earlier launch, below):
                          a split-K job then needs no memset node of its own, and the launch planner may let two workgroups share
                          a row (csrc/qpal_capi.hip plan_launch, DESIGN.md §4.2: +3 % tokens/s on a Llama-8B token) — both use
                          float atomics into the zeroed buffer; with at most two adders per element of a ZEROED output the
                          result is order-independent (0 + a + b).  A job that accumulates (`accumulate`: out += ...) may be
                          split or paired as well: its adders land on the live value h, and (h + a) + b != (h + b) + a in
                          fp32 — such an output is reproducible to rounding, not to the bit */
    const void *wscale; /* fp16 [m] or NULL: fused epilogue out[b][r] = acc * wscale[r] * oscale — the
                          `* Wscale * scale` that follows every quantized linear in the reference's incoherent
                          wrappers (lib/linear/incoherent_linear.py:83-99, 107, 327-337, 496-503) */
    float oscale;      /* 0 is read as 1 */
    long ldo;          /* row stride of out in floats, >= m; 0 is read as m.  Lets q|k|v or up|gate write the
                          column blocks of one [n][sum m] buffer */
    int x_had;         /* 1: x is the UN-rotated input; the kernel stages fp16(fp16(H_k (x * x_su) / sqrt(k)) * x_post)
                          instead — the left rotation of the incoherent wrappers (qpal_hadamard with K = 1, hd = k)
                          without a launch of its own.  Needs qpal_can_fuse_rotation(n, k); all jobs of a launch
                          that share x must share x_su / x_post */
    float x_post;      /* e.g. 1 / scale */
    const void *x_su;  /* fp16 [k] or NULL */
    int kv;            /* 0: the call's KV1.  Otherwise this job's own KV (split NONE only): jobs of one S but different
                          bit widths — q, k, v of a mixed-scheme model — then share ONE launch and one codebook image
                          (KV 2..8 for S = 9, 8..10 for S = 10, 9..10 for S = 11; batch <= 8; no x_had) */
    const void *x_f32; /* fp32 [k] or NULL (see below: only together with x_had) */
    /* decoder-block fusion (x_had jobs): the rotation reads the fp32 residual stream and applies
     * the RMSNorm in front of it (lib/linear/incoherent_linear.py:76-108 is called on `input_layernorm(h)` by the model,
     * model/llama.py), and o_proj / down_proj add their result to it.  With x_had = 1: x_f32 (fp32 [k], 16-byte aligned) may
     * replace x; x_rms_eps > 0 normalises x <- x * rsqrt(mean(x^2) + eps) * x_rms_w (fp16 [k] or NULL) in fp32 before the
     * fp16 rounding and the sign flip.  accumulate = 1 (any job): out += result (out is the residual stream).  The launch
     * may split K on its own: the split-K atomics add onto the live contents of out, no memset is ever issued for an
     * accumulating job, and out_zeroed is ignored for it (leave it 0: the buffer is NOT zero).                           */
    float x_rms_eps;
    const void *x_rms_w;
    int accumulate;
    int kv2;           /* with kv != 0 and c2 != NULL: a COLUMN-SPLIT (combt) layer inside an any-KV launch — stream 1 (columns
                          [0, k/2)) at kv, stream 2 at kv2 bits, both of the call's S; the call's split stays NONE.  tcomb and
                          tcq projections of a mixed-scheme model then share one launch.  0: single stream */
    void *act_out;     /* fp16 [m / 2] or NULL.  Non-NULL (needs x_had, batch 1): the layer is an up | gate pair whose supertile
                          rows (32 output rows) ALTERNATE up, gate, up, gate ... (qpalette_amd.linear.interleave_up_gate builds
                          it from the two layers); the epilogue then writes fp16(silu(fp16 gate)) * fp16 up — the
                          `act_fn(gate) * up` of lib/linear/incoherent_linear.py:333 — here and `out` is not written
                          (may be NULL).  The following rotation reads 2 bytes per element instead of 8 and evaluates no SwiGLU */
    /* x_had with a NON-power-of-two width (round 3): x_K = 28 and k = 14336 = 28 * 512 (the down_proj input of Llama-3.1-8B):
     * the staging applies (hadK (x) H_512) / sqrt(k) with the reference's fp16 rounding between the two factors
     * (lib/utils/matmul_had.py:137-148); x_hadk: fp16 [28][28], y[j] = sum_i x_hadk[j][i] t[i] (what qpal_hadamard takes).
     * Batch 1, fp16 x, no RMSNorm; qpal_can_fuse_rotation_k(n, k, K) says where.  x_K = 0 / 1: power-of-two widths as before. */
    const void *x_hadk;
    int x_K;
    const void *act_su; /* with act_out: fp16 [m / 2] of +-1, multiplied into the activation written — the `* SU` in front of the NEXT
                           projection's rotation (exact: a sign flip), so that rotation reads one vector instead of two; or NULL */
} qpal_tcq_job;
/* prezero/prezero_bytes (may be NULL/0): a buffer this launch also fills with zeros, for a LATER launch on the
 * same stream that accumulates into it with atomics (split-K of a few-rows x long-K layer such as down_proj).
 * Saves that launch's memset node and the two extra graph boundaries around it.  bytes % 16 == 0.           */
int qpal_tcq_gemv_multi(const qpal_tcq_job *jobs, int njobs, int n, int S, int KV1, int KV2, int split,
                        void *prezero, long prezero_bytes, void *stream);

/* Trellis decode to fp16 W[m][k] row-major (bit-exact).  Replaces decompress_ptr / _comb_ptr /
 * _combt_ptr, kernels/tcq-kernels/src/inference.cu:1862-1891, 1970-2035.                         */
int qpal_tcq_dequant(void *out_f16, const void *c1, const void *c2, const void *tlut,
                     int m, int k, int S, int KV1, int KV2, int split, void *stream);

/* VQ/SQ "tensor-core" packed format: fused decode + GEMV, fp32 out[n][m].  vec in {1,2};
 * vec=1: bits 2..8 (sq_dup / sq), vec=2: bits 2..12 (vq2).
 * Replaces decompress_gemm_ptr, kernels/vq-tensor-kernels/src/inference.cu:1112-1180.            */
int qpal_lut_tc_gemv(float *out, const void *qweight, const void *x, const void *lut,
                     int m, int n, int k, int bits, int vec, void *stream);

typedef struct qpal_lut_job {
    float *out;           /* fp32 [n][m] */
    const void *qweight;  /* int32 [m][bits*k/32/vec] */
    const void *x;        /* fp16 [n][k] */
    const void *lut;      /* fp16 [2^bits][vec] */
    int m, k;
    int out_zeroed;       /* as in qpal_tcq_job */
    const void *wscale;   /* as in qpal_tcq_job */
    float oscale;
    long ldo;
    int x_had;            /* as in qpal_tcq_job */
    float x_post;
    const void *x_su;
    const void *x_f32;    /* as in qpal_tcq_job */
    float x_rms_eps;      /* as in qpal_tcq_job */
    const void *x_rms_w;
    int accumulate;
    void *act_out;     /* as in qpal_tcq_job */
    const void *x_hadk; /* as in qpal_tcq_job */
    int x_K;
    const void *act_su;
} qpal_lut_job;
int qpal_lut_tc_gemv_multi(const qpal_lut_job *jobs, int njobs, int n, int bits, int vec, void *prezero,
                           long prezero_bytes, void *stream);

/* Same format decoded to fp16 W[m][k].  Replaces decompress_ptr, vq-tensor inference.cu:1182-1226. */
int qpal_lut_tc_dequant(void *out_f16, const void *qweight, const void *lut,
                        int m, int k, int bits, int vec, void *stream);

/* SIMT packed formats (vec=1: sq_pack_gemm.pack_gemm, kernels/sq-cuda-kernels/gemm.cu:40-85;
 * vec in {2,4}: vq_pack_gemm_*, kernels/vq-cuda-kernels/src/gemm.cu:25-100).  fp16 out[n][m].
 * Accumulates in fp32 (the reference accumulates in fp16) and rounds once at the end.           */
int qpal_lut_simt_gemv(void *out_f16, const void *qweight, const void *x, const void *lut,
                       int m, int n, int k, int bits, int vec, void *stream);

int qpal_lut_simt_dequant(void *out_f16, const void *qweight, const void *lut,
                          int m, int k, int bits, int vec, void *stream);

/* Re-pack a tensor-core-format qweight into the SIMT format on the device (load-time step of
 * VQLinearPackSIMT.gen_layer_from_info, lib/linear/vq_linear.py:175-188 ->
 * lib/quantizer/quant_op.py:246-257).  vec in {1,2}.  dst: uint32 [m][bits*k/32/vec], zeroed by the call. */
int qpal_tc_to_simt(void *dst_simt, const void *src_tc, int m, int k, int bits, int vec, void *stream);

/* Incoherence rotation either side of a quantized linear, one launch:
 *   out[r][blk] = fp16( hadK (x) H_P applied to (f(in)[r][blk] * su) / sqrt(hd) * post_scale [* sv] )
 * for every block of hd = K * 2^p consecutive elements of every row (hd == n: whole-vector transform).
 * Replaces matmul_hadU_cuda (lib/utils/matmul_had.py:137-148: third-party fast_hadamard_transform + hadK
 * matmul) and matmul_hadU_head_cuda (:95-110) plus the elementwise ops around them in
 * lib/linear/incoherent_linear.py:81, 106, 325-337 (incl. act_fn(gate) * up), 488-503.
 *   in_mode   QPAL_IN_F16: fp16 [rows][n];  QPAL_IN_F32: fp32 [rows][n] (rounded to fp16 first, the reference's
 *             .half());  QPAL_IN_SWIGLU_F32: fp32 [rows][2n] = up | gate, f = silu(gate) * up
 *   su, sv    fp16 [n] element-wise pre / post multipliers or NULL (SU sign vector; SV * scale)
 *   hadk      fp16 [K][K] row-major, entries +-1, applied as given (pass the transpose for had_left_T); NULL if K == 1
 *   round_mid 1: fp16 between the butterflies and the hadK product (matmul_hadU_cuda's fp16 pipeline);
 *             0: fp32-grade throughout (matmul_hadU_head_cuda's float path)
 * hd * 4 bytes (+ 1/32 padding) must fit the 160 KiB LDS (hd <= 39 k); K > 1 needs hd / K >= 16; `in` and `su`
 * 16-byte aligned; `out` must not be `in`.                          */
#define QPAL_IN_F16 0
#define QPAL_IN_F32 1
#define QPAL_IN_SWIGLU_F32 2
int qpal_hadamard(void *out_f16, const void *in, const void *su, const void *sv, const void *hadk,
                  int rows, int n, int hd, int K, int in_mode, int round_mid, float post_scale, void *stream);

/* RMSNorm + rotation of a whole row in one launch: out = fp16( fp16( H_n (su * w * x / rms(x)) ) * post_scale ), x fp32 [rows][n]
 * (the residual stream), rms(x) = sqrt(mean(x^2) + rms_eps), w = rms_w (fp16 [n]) or 1 — input_layernorm /
 * post_attention_layernorm (model/llama.py:119) followed by the left rotation of the incoherent wrappers, for the widths the
 * GEMV staging cannot rotate itself (qpal_can_fuse_rotation == 0: k = 8192, 5120, ...).  n = K * 2^p as for qpal_hadamard
 * (hd = n); the norm's scalar is applied after the transform (linear), one fp16 rounding of x * w * 2^-6 on the way in.   */
int qpal_hadamard_rms(void *out_f16, const float *in_f32, const void *rms_w, float rms_eps, const void *su, const void *hadk,
                      int rows, int n, int K, float post_scale, void *stream);

/* Host-side encoders of the packed formats (plain CPU code; HOST pointers; no GPU involved): what a quantiser or a
 * checkpoint converter calls once per layer.  Bit for bit the reference's packers:
 *   qpal_pack_tcq         Qidxs int32 [m][k/2] (state t of tile (tr, tc) at [16 tr + t/8][8 tc + t%8]) -> int16
 *                         [(m/16)(k/16)][8 KV]: pack_trellis + nibble permutation, lib/codebook/bitshift.py:296-329,
 *                         lib/quantizer/tcq_quant.py:47-60.  QPAL_E_PARAM if the states are not a tail-biting walk.
 *   qpal_pack_tcq_states  the same from uint16 [tiles][128] tile-major states
 *   qpal_pack_lut_tc      indices int32 [m][k/vec] -> int32 [m][bits k/32/vec]: pack_qweight, lib/quantizer/quant_op.py:89-162
 *   qpal_pack_lut_simt    indices -> uint32 [m][bits k/32/vec]: pack_qweight_sq_simt / pack_qweight_vq_simt,
 *                         lib/quantizer/quant_op.py:69-87 (numba loops of lib/quantizer/pack_op.py:288-335)            */
int qpal_pack_tcq(void *dst, const int32_t *qidxs, int m, int k, int KV);
int qpal_pack_tcq_states(void *dst, const uint16_t *states, int m, int k, int KV);
int qpal_pack_lut_tc(void *dst, const int32_t *idx, int m, int k, int bits, int vec);
int qpal_pack_lut_simt(void *dst, const int32_t *idx, int m, int k, int bits, int vec);

/* One-shot all-gather of a small activation slice across the GPUs of a node by direct peer writes over xGMI (SURVEY.md §8e;
 * no counterpart in the reference, which has no multi-GPU code): rank `rank` stores `bytes` bytes from src into
 * peer_bufs[p] + rank * bytes for every p and raises a flag in peer_ws[p]; the call returns (in stream order) when all
 * `world` slices have arrived in THIS rank's buffer peer_bufs[rank].  peer_bufs / peer_ws: HOST arrays of `world` device
 * pointers — each rank's gather buffer of this call site (world * bytes bytes, 16-byte aligned) and flag block
 * (QPAL_PEER_WS_BYTES_PER_SLOT * number of slots, zero-filled once), opened in every process through IPC handles.
 * slot: index of the call site inside a token (buffers and flags are per call site).  world <= 16; bytes % 16 == 0.
 * Graph-capturable (one kernel, epochs kept in the flag blocks).                                                         */
#define QPAL_PEER_WS_BYTES_PER_SLOT 256
int qpal_peer_gather(const void *src, long bytes, int slot, void *const *peer_bufs, void *const *peer_ws, int rank,
                     int world, void *stream);
/* Set-up helpers of the peer gather (the ONLY entry points of this library that allocate; not on the data path).  The flag
 * blocks are written by a remote GPU while a kernel of this GPU spins on them, so they must not live in ordinary (coarse-
 * grained) device memory, which is only guaranteed coherent at kernel boundaries: qpal_peer_alloc returns `bytes` bytes of
 * zero-filled device memory on the current device — kind 1: fine-grained (hipDeviceMallocFinegrained), 2: uncached
 * (hipDeviceMallocUncached), 0: plain hipMalloc (the gather buffers themselves: their contents are consumed after the
 * kernel boundary).  qpal_ipc_export writes the 64-byte IPC handle of an allocation; qpal_ipc_open maps another process's
 * allocation into this one (peer access enabled lazily); qpal_ipc_close / qpal_peer_free undo them.                      */
#define QPAL_IPC_HANDLE_BYTES 64
int qpal_peer_alloc(void **ptr, long bytes, int kind);
int qpal_peer_free(void *ptr);
int qpal_ipc_export(void *ptr, void *handle64);
int qpal_ipc_open(const void *handle64, void **ptr);
int qpal_ipc_close(void *ptr);

/* Decoder-block glue of a batch-1 decode step, one launch: rotary embedding of the new token's q and k (HF rotate_half
 * convention, cos / sin of pos * inv_freq rounded to fp16, fp16 arithmetic: model/llama.py apply_rotary_pos_emb), q as fp16,
 * k and v written into a static KV cache fp16 [nkv][max_len][hd] at position *pos (device int64).  q / k / v: fp32 (the GEMV
 * epilogue's output) [nq * hd] / [nkv * hd]; inv_freq: fp32 [hd / 2].                                                    */
int qpal_rope_kv(const float *q, const float *k, const float *v, void *q_out_f16, void *kcache_f16, void *vcache_f16,
                 const long *pos, const float *inv_freq, int nq, int nkv, int hd, long max_len, void *stream);

/* Attention of ONE new token over a static KV cache (batch 1, grouped-query heads: nq % nkv == 0), one launch: softmax(q k^T *
 * scale) v over positions 0 .. *pos, fp32 accumulation, fp16 out [nq][hd].  Caches fp16 [nkv][max_len][hd], 16-byte aligned.
 * (torch SDPA runs this shape as ~10 launches.)  max_len up to ~40 k positions (scores live in LDS).                      */
int qpal_attn_decode(const void *q_f16, const void *kcache_f16, const void *vcache_f16, void *out_f16, const long *pos,
                     int nq, int nkv, int hd, long max_len, float scale, void *stream);

/* The tail of a greedy decode step as one launch: x = fp16(rmsnorm(h) [* rms_w]) (rms_eps = 0: x = fp16(h)), logits = W x with
 * W the fp16 lm_head [vocab][k] (k in {2048, 4096, 8192}, 16-byte aligned), *token = argmax (lowest index on ties); logits
 * (fp32 [vocab]) are also written when the pointer is not NULL.  ws: qpal_lm_head_ws_bytes(vocab) bytes of device memory,
 * zero-filled once, kept across launches.  Replaces model.norm + lm_head + argmax of the reference's decode loop
 * (eval/measure_latency.py: logits[:, -1].argmax).                                                                          */
long qpal_lm_head_ws_bytes(int vocab);
int qpal_lm_head_argmax(const float *h_f32, const void *rms_w_f16, float rms_eps, const void *w_f16, float *logits_f32, long *token,
                        void *ws, long ws_bytes, int vocab, int k, void *stream);

/* The two launches above as one (what a decode step runs): rotary embedding of the new token's q / k, k and v appended to the
 * cache at *pos, attention over positions 0 .. *pos, fp16 out [nq][hd].  q / k / v fp32 as for qpal_rope_kv.  hd in {64, 128,
 * 256}; max_len % 4 == 0.  The new row is read from on-chip memory by every head of its group: nothing this launch reads was
 * written by it.  ws == NULL: one workgroup per query head (scores of the whole context in LDS: max_len up to ~38 k).
 * ws != NULL (qpal_attn_ws_bytes(...) > 0 bytes of device memory, 4-byte aligned, zero-filled ONCE, kept across launches):
 * split-context form for long caches — workgroup (kv head, chunk of the context) serves all nq / nkv query heads of its group,
 * the last workgroup of a kv head to arrive merges the partial softmaxes.  qpal_attn_ws_bytes returns 0 where the split form
 * does not apply (max_len < 2048, nq / nkv not in {1, 2, 4, 8}, (nq / nkv) * hd > 1024): pass ws = NULL then.
 * *pos outside [0, max_len) (it lives on the device: the host cannot check it): the launch does nothing — no cache row is
 * written, out is left as it was; the same holds for qpal_rope_kv / qpal_attn_decode.                                   */
long qpal_attn_ws_bytes(int nq, int nkv, int hd, long max_len);
int qpal_attn_rope_decode(const float *q, const float *k, const float *v, void *kcache_f16, void *vcache_f16, void *out_f16,
                          const long *pos, const float *inv_freq, int nq, int nkv, int hd, long max_len, float scale,
                          void *ws, long ws_bytes, void *stream);

/* The launch planner of the fused GEMV entry points, on its own (host code, no GPU call; what tests and tools inspect).
 * A launch of njobs jobs — rows[j] supertile rows (m / 32) of steps1[j] + steps2[j] steps (a step = 128 columns; steps2 = 0: one
 * stream) — is cut into workgroup-sized pieces: a GROUP of G = 1 << lg_g workgroups (`waves` = 16 or 8 waves each) owns rg
 * consecutive rows; its work, laid out as a tape (row 0 stream 1, row 0 stream 2, row 1 ...), is cut into G equal ranges and every
 * range into one piece per wave.  flags[j]: bit 0 the output is zeroed, bit 1 the job accumulates, bit 2 SwiGLU epilogue;
 * shared_staging: the jobs read one x / codebook (groups may then run across job boundaries).  out (ints):
 *   [0] grid  [1] items  [2] geometry classes  [3] items of class 0  [4] groups span jobs  [5] class-1 job mask  [6] M  [7] W
 *   then per class c < 2: lg_g, rg, M * W entries (a, b) — a: bits 0..7 row inside the group, 8 stream 2, 9 lead of its row's run,
 *   10 row shared with another workgroup, 11..15 waves in the run, 16 has steps, 17 SwiGLU lead; b: first step | steps << 16 —
 *   then per job: class, first virtual row, end of its virtual rows, 2 if its output must start at zero (shared rows) else 1.
 * out_len >= 8 + 2 * (2 + 2 * M * W) + 4 * njobs (M = 4, W = 16).                                                              */
int qpal_plan_gemv(const int *rows, const int *steps1, const int *steps2, const int *flags, int njobs, int waves, int shared_staging,
                   int *out, int out_len);

/* 1 if the GEMV entry points can apply the rotation themselves (x_had): k in {2048, 4096} at batch 1 (the
 * decode case); 0 otherwise (then call qpal_hadamard first). */
int qpal_can_fuse_rotation(int n, int k);
/* the same for x_K > 1 (K = 28, k = 14336, batch 1; codecs whose codebook image is >= 40 KiB: every TCQ codec) */
int qpal_can_fuse_rotation_k(int n, int k, int K);

/* Calibration of the measurement contract (bench.py `roofline` block; SURVEY.md §8d: "report a measured stream-read ceiling").
 * NOT on the data path — no module calls them; csrc/calib.hip.
 *   qpal_calib_stream_read  one launch (`grid` workgroups of 1024 threads) that does nothing but read srcs[i] (bytes[i] bytes each,
 *                           16-byte aligned, bytes % 16 == 0, nseg <= 16) with 16-byte non-temporal loads; sink: >= 4 KiB of
 *                           device memory (never written in practice).  Over > 1 GB: the stream ceiling; over the packed buffers
 *                           of one GEMV launch: what that launch would take if the decode were free.
 *   qpal_calib_decode_rate  `grid` workgroups of 16 waves run `iters` decode + MFMA steps of the TCQ codec (S, KV) — the step
 *                           function of the fused GEMV kernel itself — on register-resident packed words: no HBM traffic.
 *                           wave-steps executed = grid * 16 * iters (one step = 32 rows x 128 columns of W).            */
int qpal_calib_stream_read(const void *const *srcs, const long *bytes, int nseg, void *sink, int grid, void *stream);
int qpal_calib_decode_rate(const void *tlut, void *sink, int iters, int S, int KV, int grid, void *stream);

const char *qpal_error_string(int code);
int qpal_version(void);

#ifdef __cplusplus
}
#endif
#endif /* QPAL_H */
