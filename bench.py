#!/usr/bin/env python3
"""Decode-throughput bench of the dequant-matmul hot path (BASELINE.json metric).

A "step" is ONE decoded token at batch 1 = one pass over all 224 quantized linears of a
Llama-3.1-8B-shaped model (32 layers x {q,k,v,o,gate,up,down}), every layer with its own packed
buffers (2.8 GB working set >> the 256 MB Infinity Cache), captured once into a HIP graph and
replayed.  Inputs are resident in HBM before the timed region.  Default workload = BASELINE.json
configs[1]: uniform `tcomb_6_7_0.5_none_0.9` (TCQ 3.25 b/w, CombtLinearTCQ).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU; default `--parallel dp` = N independent replicas decoding N token streams
(weak scaling, no data-path collective — how an 8B model that fits one GPU is served); `--parallel tp`
row-shards every linear over the ranks and all-gathers the activations over RCCL/xGMI (strong scaling).

Prints ONE JSON line (rank 0).  roofline: HBM-bound; achieved = algorithmic bytes per token (packed
weights + codebooks + x + out of every linear) / measured time per token, per GPU,
timed with HIP events on the launch stream; per-launch figures = per-token / 224.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TP_ABANDONED_RC = 75  # (EX_TEMPFAIL) the multi-GPU leg was abandoned / raised / lost a rank; the headline line was still printed
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable

WORKLOADS = {
    # name -> (model_key, quantizer for every linear, fused layer list or None)
    "llama3.1-8b_tcomb_6_7": ("3_8b", "tcomb_6_7_0.5_none_0.9"),
    "llama3.1-8b_tcq_6": ("3_8b", "tcq_6_none_0.9"),
    "llama3.1-8b_ldlq_1_4": ("3_8b", "ldlq_1_4_none_1.0"),
    "llama3.1-70b_tcq_6": ("3_70b", "tcq_6_none_0.9"),
    # the reference's published mixed-scheme results (perf/qdicts/*.json, exported from msq_results/):
    "llama3.1-8b_figure1c": ("3_8b", "qdict:figure1c"),   # latency-aware MSQ, no fusion, avg 2.86 b/w
    "llama3.1-8b_figure1d": ("3_8b", "qdict:figure1d"),   # fusion-aware MSQ (merged qkv/kv + up|gate), avg 2.96 b/w
    # BASELINE.json configs[2]: memory-constrained MSQ at 3.25 avg bits, mixed TCQ / VQ / SQ (perf/make_mem3p25.py)
    "llama3.1-8b_mem3p25": ("3_8b", "qdict:mem3p25"),
}
LINEAR_ORDER = ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
                "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="llama3.1-8b_tcomb_6_7", choices=sorted(WORKLOADS))
    ap.add_argument("--parallel", default="dp", choices=["dp", "tp"])
    ap.add_argument("--gather", default="peer", choices=["peer", "rccl"],
                    help="--parallel tp: how the row shards of o_proj / down_proj outputs are all-gathered. peer (default): one-shot "
                         "direct peer writes over xGMI, one kernel inside the captured graph (csrc/peer_gather.hip); rccl: "
                         "dist.all_gather_into_tensor, eager (the correctness baseline)")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--streams", type=int, default=1,
                    help="1: every GEMV on one stream; 3: q|k|v and gate|up fork onto side streams inside the graph")
    ap.add_argument("--launch", default="multi", choices=["single", "multi"],
                    help="single: one launch per linear (7/layer); multi (default, the fastest measured): q|k|v and gate|up "
                         "(projections of one input) go out as one multi-job launch each (4 launches/layer, same arithmetic, "
                         "same buffers)")
    ap.add_argument("--distinct-codebooks", action="store_true",
                    help="give every linear its own random codebook (default: one codebook per model, as in real "
                         "Q-Palette checkpoints where every layer stores a copy of the same k-means codebook)")
    ap.add_argument("--no-prezero", action="store_true",
                    help="A/B switch: let down_proj's split-K zero its output with its own memset node")
    ap.add_argument("--incoherent", action="store_true",
                    help="run every projection group inside the reference's incoherence wrapper: sign flip + Hadamard + "
                         "1/scale before (one qpal_hadamard launch, SwiGLU fused for down_proj), Wscale*scale fused into "
                         "the GEMV epilogue (SURVEY §8 f-1)")
    ap.add_argument("--packing", default="qdict", choices=["qdict", "mi355x"],
                    help="qdict workloads: tensor-core-order vs SIMT packing of the VQ/SQ layers as published (chosen from the "
                         "reference's RTX 4090 latency table) or re-chosen from this GPU's table (perf/latency/)")
    ap.add_argument("--no-kind-breakdown", action="store_true", help="skip the per-launch-kind timing of the default run")
    ap.add_argument("--no-calibration", action="store_true",
                    help="skip the calibration block of the roofline (measured stream ceiling, stream token, decode floor)")
    ap.add_argument("--no-swiglu-epilogue", action="store_true",
                    help="--incoherent: up|gate as fp32 outputs and SwiGLU inside the rotation launch (default: interleaved up|gate layer, "
                         "SwiGLU in the GEMV epilogue)")
    ap.add_argument("--no-incoherent-extra", action="store_true",
                    help="skip the second figure (token with the incoherence wrapper) of the default run")
    ap.add_argument("--no-fuse-rotation", action="store_true",
                    help="--incoherent: always rotate in a launch of its own (default: inside the GEMV where k allows)")
    ap.add_argument("--layers", type=int, default=0, help="override the number of layers (0 = model's)")
    ap.add_argument("--strict-exit", action="store_true", help="(the default since round 5; accepted for older command lines)")
    ap.add_argument("--lenient-exit", action="store_true",
                    help=f"exit with code 0 instead of {TP_ABANDONED_RC} when the tp_70b leg of an N > 1 run was abandoned, raised, or a rank "
                         "died in it (default: a multi-GPU leg that did not complete FAILS the run — the headline line, measured before "
                         "the leg, is printed either way and carries \"tp_70b_abandoned\": true)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) for real runs; gloo to rehearse ranks on one GPU")
    ap.add_argument("--force-device", type=int, default=-1, help="rehearsal only: put every rank on this GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the `other_configs` key of the default N = 1 run (BASELINE configs[2..4]: mem3p25, figure1d, figure1c, the 70B "
                         "shapes at batch 1 and 16 — <= 10 steps each, after the headline's timed region)")
    ap.add_argument("--no-whole-model", action="store_true",
                    help="skip the `whole_model_decode` figure of the default N = 1 run (the fused decode step of perf/decode_llama.py)")
    ap.add_argument("--no-tp-leg", action="store_true",
                    help="N > 1: skip the `tp_70b` figures (Llama-3.1-70B shapes row-sharded over the N GPUs, batch 1 and 16)")
    ap.add_argument("--tp-layers", type=int, default=0, help="layers of the tp_70b leg (0 = the model's 80)")
    ap.add_argument("--tp-steps", type=int, default=0, help="timed steps of each tp_70b figure (0 = min(--steps, 30))")
    ap.add_argument("--tp-timeout", type=float, default=float(os.environ.get("QPAL_TP_TIMEOUT", "300")),
                    help="seconds after which the tp_70b leg is abandoned (the headline line is printed without it)")
    return ap.parse_args()


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher: start the N ranks as child processes (one per GPU, the same command line,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) BEFORE anything in this process touches the GPU, wait for
    them and pass rank 0's output (the ONE JSON line) through.  Nothing is re-exec'ed: the parent stays a plain Python process."""
    import socket
    import subprocess
    import tempfile

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    # rank 0 parks the measured headline line here before it enters the tp_70b leg and removes the file once it has printed:
    # should the leg take the process down (a fault the in-process watchdog cannot catch), the parent prints the parked line
    park = os.path.join(tempfile.gettempdir(), f"qpal_bench_headline_{os.getpid()}.json")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), QPAL_BENCH_HEADLINE_FILE=park,
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # wait for all; a rank that dies takes the others down with it (they would sit in a rendezvous or a collective until its
    # timeout): terminate exactly the processes started here
    rc, live = 0, list(procs)
    while live:
        for p_ in list(live):
            r = p_.poll()
            if r is None:
                continue
            live.remove(p_)
            rc = max(rc, abs(r))
            if r != 0:
                for q_ in live:
                    q_.terminate()
        if live:
            time.sleep(0.2)
    if os.path.exists(park):
        try:
            with open(park) as f:
                out = json.load(f)
            out["tp_70b"] = {"error": f"a rank died inside the leg (exit code {rc}); the headline above was measured before it"}
            out["tp_70b_abandoned"] = True
            print(json.dumps(out), flush=True)
            print(f"[bench] a rank died inside the tp_70b leg (exit code {rc}); the headline line above was measured before it",
                  file=sys.stderr, flush=True)
            # the measured headline is out; the run still FAILS (a multi-GPU leg that lost a rank must be visible in the exit code)
            rc = 0 if "--lenient-exit" in sys.argv else TP_ABANDONED_RC
        finally:
            os.remove(park)
    return rc


_KIND = {"self_attn.q_proj": "q", "self_attn.k_proj": "k", "self_attn.v_proj": "v", "self_attn.o_proj": "o",
         "mlp.gate_proj": "g", "mlp.up_proj": "u", "mlp.down_proj": "d"}


def _mi355x_prefers_simt(keys, qstr):
    """Packing choice from THIS GPU's latency table (perf/latency/3_8b_latency_coeffs_mi355x.json, the MI355X counterpart of
    the table the reference's solver reads): the published qdicts carry the choice made from the RTX 4090 table."""
    global _LAT
    try:
        _LAT
    except NameError:
        with open(os.path.join(ROOT, "perf", "latency", "3_8b_latency_coeffs_mi355x.json")) as f:
            _LAT = json.load(f)
    kind = "".join(_KIND[k] for k in keys)
    kind = {"gu": "ug", "ug": "ug"}.get(kind, kind)
    tc, simt = _LAT.get(f"{kind}_{qstr}_False"), _LAT.get(f"{kind}_{qstr}_True")
    return tc is not None and simt is not None and simt < tc


def build_model(qp, torch, model_key, qstr, nlayers, device, shard=None, distinct_codebooks=False, packing="qdict",
                keep_infos=False):
    """-> list over layers of 4 groups [attention inputs (q,k,v), o, mlp inputs (gate,up), down]; a group is a
    list of (module, in_features, linear_info-or-None) that share one input vector.  Every layer has
    distinct packed buffers.  qstr "qdict:<name>": quantizer per linear (+ fused layers) from perf/qdicts."""
    li = qp.mem_op.get_layer_info(model_key)
    qdict, merge_info = None, None
    if qstr.startswith("qdict:"):
        with open(os.path.join(ROOT, "perf", "qdicts", qstr[6:] + ".json")) as f:
            data = json.load(f)
        qdict, merge_info = data["qdict"], data["merge_info"]
    cseed = None if distinct_codebooks else 777

    def make(layer, keys):
        """one (possibly fused) linear from the row-concatenation of `keys`"""
        infos, q0, simt0 = [], None, None
        for key in keys:
            q, simt = (qdict[f"{layer}_{key}"] if qdict is not None else (qstr, "0"))
            q0, simt0 = (q, simt) if q0 is None else (q0, simt0)
            assert (q, simt) == (q0, simt0), "fused layers must share the quantizer"
            k, m = li[key]["in_features"], li[key]["out_features"]
            info = qp.mem_op.dummy_linear_info(k, m, q, seed=layer * 16 + LINEAR_ORDER.index(key), device=device,
                                               codebook_seed=cseed)
            if shard is not None:  # this rank's rows of the SAME full layer every rank generates (q|k|v by heads, ...)
                info = qp.shard.shard_linear_info(info, shard[0], shard[1])
            infos.append(info)
        cls = qp.linear.linear_class_for(q0, use_simt=False)
        info = infos[0]
        for other in infos[1:]:
            info = cls.merge_infos(info, other)
        use_simt = simt0 == "1" if packing == "qdict" else _mi355x_prefers_simt(keys, q0)
        if use_simt and "ldlq" in q0:
            mod = qp.VQLinearPackSIMT.gen_layer_from_info(info, device=device)
        else:
            mod = cls.gen_layer_from_info(info).to(device)
        return (mod, info["in_features"], info if (layer == 0 or keep_infos) else None)

    layers = []
    for layer in range(nlayers):
        merges = set(merge_info[layer]) if merge_info is not None else set()
        q, k, v, o, g, u, d = LINEAR_ORDER
        if "merge_qkv" in merges:
            attn = [make(layer, [q, k, v])]
        elif "merge_kv" in merges:
            attn = [make(layer, [q]), make(layer, [k, v])]
        elif "merge_qk" in merges:
            attn = [make(layer, [q, k]), make(layer, [v])]
        elif "merge_qv" in merges:
            attn = [make(layer, [q, v]), make(layer, [k])]
        else:
            attn = [make(layer, [q]), make(layer, [k]), make(layer, [v])]
        mlp = [make(layer, [u, g])] if "merge_ug" in merges else [make(layer, [g]), make(layer, [u])]
        layers.append([attn, [make(layer, [o])], mlp, [make(layer, [d])]])
    if not distinct_codebooks:  # what a checkpoint loader does: identical codebooks share one tensor
        qp.share_codebooks([m for groups in layers for grp in groups for m, _, _ in grp])
    if os.environ.get("QPAL_BENCH_SHARED_WEIGHTS"):  # experiment knob, NOT a benchmark mode: every layer reads layer 0's buffers (88 MB:
        layers = [layers[0]] * len(layers)          # resident in the 256 MB Infinity Cache) — what would cache-resident weights buy?
    return layers


KINDS = ["q|k|v", "o", "gate|up", "down"]  # the four dependent multi-job launches of a decoder block


def _kind_on(only_kind, gi):
    """only_kind: None (every launch kind), one kind index, or a collection of them"""
    return only_kind is None or (gi in only_kind if hasattr(only_kind, "__contains__") else only_kind == gi)


def make_token(qp, torch, layers, xs, n, device, launch="multi", no_prezero=False, gather=None, side=(), main_stream=None,
               only_kind=None):
    """-> (token, owned): token() runs every quantized linear of `layers` once (one decoded token at batch n, plain inputs
    xs[in_features]) and returns the outputs in model order.  Also what tests/test_bench_workloads.py drives for the published
    mixed-scheme workloads.
    only_kind (multi-job launches only): run just launch kind 0..3 of every block, with exactly the arguments it has inside
    the token (timing of one launch kind; o / down then accumulate into whatever their pre-zeroed buffer holds)."""
    # Batched token (4 <= n <= fused batch, multi-job launches): the lockstep GEMM kernel splits K of every launch kind, so every
    # output must start at zero.  The harness owns the buffers, as a decode loop would: the outputs of a launch are consecutive
    # blocks of one allocation, zeroed by the launch BEFORE it (prezero) — one memset node per token instead of five per layer.
    # Up to max_chunked_batch the batch goes through each launch kind in passes of <= 64 rows (the second pass re-reads weights the
    # first has just pulled through L2 / the Infinity Cache).
    batched = (gather is None and launch == "multi" and only_kind is None
               and n >= int(os.environ.get("QPAL_GEMM_MIN_BATCH", "4"))
               and all(n <= max(m.max_fused_batch, m.max_chunked_batch) and type(m) in qp.linear._PACKED_KEYS
                       for groups in layers for grp in groups for m, _, _ in grp))
    blocks = None
    if batched:
        blocks = []
        for groups in layers:
            per = []
            for grp in groups:
                ms = [m.out_features for m, _, _ in grp]
                flat = torch.empty(n * sum(ms), dtype=torch.float32, device=device)
                views, off = [], 0
                for m_ in ms:
                    views.append(flat[off: off + n * m_].view(n, m_))
                    off += n * m_
                per.append((flat, views))
            blocks.append(per)

    def token_batched():
        outs = []
        seq = [(li, gi) for li in range(len(layers)) for gi in range(4)]
        for idx, (li, gi) in enumerate(seq):
            grp = layers[li][gi]
            flat, views = blocks[li][gi]
            nxt = blocks[seq[idx + 1][0]][seq[idx + 1][1]][0] if idx + 1 < len(seq) else None
            mods = [m for m, _, _ in grp]
            step = min(m.max_fused_batch for m in mods)
            for i in range(0, n, step):
                rows = [v[i:i + step] for v in views]
                if no_prezero:
                    qp.multi_gemv(mods, xs[grp[0][1]][i:i + step], outs=rows)
                else:
                    qp.multi_gemv(mods, xs[grp[0][1]][i:i + step], outs=rows, outs_zeroed=idx > 0, prezero=nxt if i == 0 else None)
            outs += views
        return outs

    # Batch 1..3, multi-job launches: the harness owns the outputs here too — one block per launch, zeroed by the launch before it
    # (prezero), so that the library's launch planner may share rows between workgroups or split K wherever that fills the chip,
    # without a memset node of its own.  The first launch of a token has no predecessor: its outputs are not declared zeroed.
    # (Round 4: per GROUP — a mixed-scheme model's groups of tensor-core-order layers get owned, pre-zeroed outputs although other
    # groups of the model are SIMT-packed: their single-codec gate | up launches then pair like the uniform model's.)
    owned = None
    if launch == "multi" and not batched and not no_prezero and \
            all(n <= m.max_fused_batch for groups in layers for grp in groups for m, _, _ in grp):
        owned = []
        for groups in layers:
            per = []
            for grp in groups:
                if not all(type(m) in qp.linear._PACKED_KEYS for m, _, _ in grp):
                    per.append(None)  # a SIMT-packed layer in the group: its op allocates its own fp16 output
                    continue
                ms = [m.out_features for m, _, _ in grp]
                flat = torch.empty(n * sum(ms), dtype=torch.float32, device=device)
                views, off = [], 0
                for m_ in ms:
                    views.append(flat[off: off + n * m_].view(n, m_))
                    off += n * m_
                per.append((flat, views))
            owned.append(per)
        if all(g is None for per in owned for g in per):
            owned = None

    def token():
        if batched:
            return token_batched()
        outs = []
        if hasattr(gather, "new_token"):
            gather.new_token()  # the same call sites take the same peer-gather slots in every (captured) token
        for li, groups in enumerate(layers):
            pre = {}  # group index -> output buffer zeroed by an earlier launch of this block
            for gi, grp in enumerate(groups):
                if owned is not None:
                    nli, ngi = (li, gi + 1) if gi < 3 else (li + 1, 0)
                    nxt = owned[nli][ngi][0] if nli < len(layers) and owned[nli][ngi] is not None else None
                    pli, pgi = (li, gi - 1) if gi > 0 else (li - 1, 3)
                    # this group's block was zeroed by the launch before it iff that launch was a multi-job launch of packed layers
                    prev_zeroed = pli >= 0 and owned[pli][pgi] is not None
                    if owned[li][gi] is None:  # a SIMT-packed layer in the group: plain launches (nothing zeroes the next block)
                        if _kind_on(only_kind, gi):
                            outs += qp.multi_gemv([m for m, _, _ in grp], xs[grp[0][1]])
                        continue
                    flat, views = owned[li][gi]
                    if _kind_on(only_kind, gi):
                        ys = qp.multi_gemv([m for m, _, _ in grp], xs[grp[0][1]], outs=views, outs_zeroed=prev_zeroed, prezero=nxt)
                        # row-sharded model (--parallel tp, tp_70b): the outputs of o_proj / down_proj feed full-width consumers
                        # (the next block's rotation): all-gather them.  q|k|v stay head-sharded through attention, gate|up
                        # channel-sharded into down_proj.  A shard has 1 / N of the rows: with its outputs owned
                        # and zeroed the planner can split K until the chip is busy, without memset nodes.
                        outs += [gather(y) for y in ys] if gather is not None and gi in (1, 3) else ys
                    continue
                x = xs[grp[0][1]]
                mods = [m for m, _, _ in grp]
                if gather is not None and launch == "multi" and n <= min(m.max_fused_batch for m in mods):
                    # row-sharded model (--parallel tp): the same multi-job launches on every rank's shard; the outputs of
                    # o_proj / down_proj feed full-width consumers (the next block's rotation): all-gather them.  q|k|v stay
                    # head-sharded through attention, gate|up channel-sharded into down_proj's rotation input, which is
                    # gathered as part of down's own input in a real model — here as its output slice.
                    ys = qp.multi_gemv(mods, x)
                    outs += [gather(y) for y in ys] if gi in (1, 3) else ys
                    continue
                if n > min(m.max_fused_batch for m in mods) and gather is None:
                    # beyond the fused batch: decode to fp16 W (staged 16-byte stores) + fp16 GEMM, as the reference does for
                    # bs > 8 (lib/linear/tcq_linear.py:75-84); the MFMA roofline of this path is reported by --batch
                    if _kind_on(only_kind, gi):
                        outs += qp.multi_gemv(mods, x)   # passes of the fused launches up to max_chunked_batch, decode + GEMM above
                    continue
                if launch == "multi" and gather is None:
                    # projections of one input: one multi-job launch per codec.  The attention-input / mlp-input
                    # launches also zero the output of o_proj / down_proj, so that a split-K there (few rows:
                    # half of the CUs would idle) needs no memset node of its own.
                    nxt = gi + 1
                    if gi in (0, 2) and not no_prezero and len(groups[nxt]) == 1:
                        pre[nxt] = torch.empty((n, groups[nxt][0][0].out_features), dtype=torch.float32, device=device)
                        if _kind_on(only_kind, gi):
                            outs += qp.multi_gemv(mods, x, prezero=pre[nxt])
                    elif not _kind_on(only_kind, gi):
                        pass
                    elif gi in pre:
                        outs += qp.multi_gemv(mods, x, outs=[pre[gi]], outs_zeroed=True)
                    else:
                        outs += qp.multi_gemv(mods, x)
                    continue
                if len(mods) > 1 and side:                # fork/join onto side streams inside the graph
                    ev = torch.cuda.Event()
                    ev.record(main_stream)
                    evs = []
                    for j, mod in enumerate(mods):
                        if j == 0:
                            outs.append(mod._gemv(x, n))
                            continue
                        s_ = side[(j - 1) % len(side)]
                        s_.wait_event(ev)
                        with torch.cuda.stream(s_):
                            outs.append(mod._gemv(x, n))
                            e2 = torch.cuda.Event()
                            e2.record(s_)
                            evs.append(e2)
                    for e2 in evs:
                        main_stream.wait_event(e2)
                    continue
                for mod in mods:
                    y = mod._gemv(x, n)
                    if gather is not None and gi in (1, 3):  # o_proj / down_proj feed full-width consumers
                        y = gather(y)
                    outs.append(y)
        return outs

    # (second value: the harness-owned output blocks [block][launch kind] -> flat fp32 tensor or None; None when the library owns them)
    return token, ([[g[0] if g is not None else None for g in per] for per in owned] if owned is not None else None)


def algorithmic_bytes(qp, layers, batch):
    """Bytes every token must read/write at least once: packed weights + codebook + x (fp16) + out."""
    total = 0
    for groups in layers:
        for mod, k, _ in (u for grp in groups for u in grp):
            for name in ("trellis", "trellis1", "trellis2", "qweight", "tlut", "lut"):
                t = getattr(mod, name, None)
                if t is not None:
                    total += t.numel() * t.element_size()
            total += batch * k * 2 + batch * mod.out_features * 4
    return total


_PACKED_NAMES = ("trellis", "trellis1", "trellis2", "qweight")


def calibrate(qp, torch, layers, n, device, stream, steps):
    """Calibration of the roofline block, measured in the SAME run on the SAME box (boxes differ by +-5 %):
      measured_stream_GBps  one launch of csrc/calib.hip's pure read (16-byte non-temporal loads, 256 x 1024 threads) over 2 GiB
      stream_token_ms       the token's launch structure with every GEMV launch replaced by that pure read over exactly the packed
                            buffers the launch decodes (same dependent order, one HIP graph): what the token would take if decode,
                            staging, reduction and epilogue were free — the boundary, dispatch and first-byte cost of `launches`
                            dependent launches plus their bytes at the chip's streaming rate
      decode_floor_ms       the token's wave-steps at the rate the decode + MFMA step of tc_gemv_kernel sustains on register-resident
                            words (qpal_calib_decode_rate, per TCQ codec of the workload; null when the workload has VQ/SQ layers)."""
    import ctypes

    lib = qp._native.lib()
    P, L = ctypes.c_void_p, ctypes.c_long
    sink = torch.zeros(1024, dtype=torch.int32, device=device)
    ncu = torch.cuda.get_device_properties(device).multi_processor_count

    def read_launch(tensors):
        segs = [(t.data_ptr(), t.numel() * t.element_size()) for t in tensors]
        segs = [(p_, b - b % 16) for p_, b in segs if b >= 16]
        for i in range(0, len(segs), 16):
            part = segs[i:i + 16]
            ptrs = (P * len(part))(*[p_ for p_, _ in part])
            byts = (L * len(part))(*[b for _, b in part])
            qp._native.check(lib.qpal_calib_stream_read(ptrs, byts, len(part), sink.data_ptr(), ncu, stream.cuda_stream), "qpal_calib_stream_read")

    def timed(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    out = {}
    with torch.cuda.stream(stream):
        big = torch.empty(2 << 30, dtype=torch.uint8, device=device)
        read_launch([big])
        torch.cuda.synchronize()
        t = min(timed(lambda: read_launch([big]), 3) for _ in range(3))
        out["measured_stream_GBps"] = big.numel() / t / 1e9
        out["measured_stream_what"] = "pure read of 2 GiB in one launch (csrc/calib.hip), best of 3 x 3 launches"
        del big
        # the stream token: per GEMV launch of the token, a pure read of its packed buffers
        mixed_kv = n <= 8
        launches = []
        for groups in layers:
            for grp in groups:
                mods = [m for m, _, _ in grp]
                for idxs in qp.linear.launch_groups(mods, mixed_kv=mixed_kv):
                    launches.append([t_ for i in idxs for nm in _PACKED_NAMES if (t_ := getattr(mods[i], nm, None)) is not None])

        def stream_token():
            for ts in launches:
                read_launch(ts)
        stream_token()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            stream_token()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        out["stream_token_ms"] = timed(g.replay, max(5, min(steps, 50))) * 1e3
        out["stream_token_launches"] = len(launches)
        # decode floor
        steps_by_codec, ok = {}, True
        for groups in layers:
            for m, k, _ in (u for grp in groups for u in grp):
                rows = m.out_features // 32
                if isinstance(m, qp.QTIPLinearTCQ):
                    key = (m.tlut_bits, m.KV)
                    steps_by_codec[key] = steps_by_codec.get(key, 0) + rows * -(-k // 128)
                elif isinstance(m, qp.CombtLinearTCQ):
                    for kv, kp in zip(m.KV, m.in_part):
                        key = (m.tlut_bits, kv)
                        steps_by_codec[key] = steps_by_codec.get(key, 0) + rows * -(-kp // 128)
                else:
                    ok = False
        if ok and steps_by_codec:
            iters = 400
            floor, rates = 0.0, {}
            for (S, kv), wsteps in sorted(steps_by_codec.items()):
                tl = torch.randn(2 ** S, 2, device=device).half()
                run = lambda: qp._native.check(lib.qpal_calib_decode_rate(tl.data_ptr(), sink.data_ptr(), iters, S, kv, ncu, stream.cuda_stream),
                                               "qpal_calib_decode_rate")
                run()
                torch.cuda.synchronize()
                t = min(timed(run, 2) for _ in range(2))
                rate = ncu * 16 * iters / t  # wave-steps per second, whole chip
                rates[f"tcq_{S}_{kv}"] = {"wave_steps_per_us_per_cu": rate / ncu / 1e6, "ns_per_wave_step_per_simd_slot": 4e9 * ncu / rate,
                                         "wave_steps_per_token": wsteps}
                floor += wsteps / rate
            out["decode_floor_ms"] = floor * 1e3
            out["decode_rate"] = rates
        else:
            out["decode_floor_ms"] = None
    return out


def usable_cpus():
    """-> (CPUs visible, CPUs this process can actually use, how that was found).  A GPU box gives a job a SHARE of its host
    (cgroup CPU quota): 128 visible cores with a quota of 16 run 128 OpenMP threads 8-deep — round 3's "128 cores, 5.7 x one
    thread" was that.  The baseline uses, and reports, the share."""
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    quota, why = None, "affinity mask"
    for path in ("/sys/fs/cgroup/cpu.max", ):
        try:
            with open(path) as f:
                q, per = f.read().split()[:2]
            if q != "max":
                quota = max(1, int(math.ceil(int(q) / int(per))))
                why = f"cgroup v2 cpu.max = {q} {per}"
        except (OSError, ValueError):
            pass
    if quota is None:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0:
                quota = max(1, int(math.ceil(q / per)))
                why = f"cgroup v1 cfs quota {q} / {per}"
        except (OSError, ValueError):
            pass
    env = os.environ.get("QPAL_CPU_THREADS")
    if env:
        return visible, int(env), "QPAL_CPU_THREADS"
    return visible, (min(visible, quota) if quota else visible), why


def cpu_baseline(qp, layers, batch, seconds, nl=None):
    """CPU baseline (kind "port": the oracle's restatement of what the reference does without a GPU — fake-dequant to fp16 W,
    then fp32-accumulate x @ W.T; lib/quantizer/quant_op.py:185-201, lib/utils/kernel_decompress.py:64-88) on ONE layer's
    linears of the same workload: all host cores (OpenMP) and one thread, W materialised and fused (decode a tile, multiply,
    never write W).  All-core figures: median of >= 10 full-layer runs; one-thread figures: median of 3 runs of the layer's
    smallest linear, scaled by weight count (a full layer on one thread takes ~20 s).  value = materialised, all cores."""
    import numpy as np
    from oracle import oracle

    mods = [u for grp in layers[0] for u in grp]
    host = []
    for mod, k, info in mods:
        m = mod.out_features
        x = np.random.default_rng(0).standard_normal((batch, k)).astype(np.float16)
        host.append(({kk: (v.cpu().numpy() if hasattr(v, "cpu") else v) for kk, v in info.items()}, m, k, x,
                     np.empty((m, k), dtype=np.uint16)))

    def run_one(entry, fused):
        info, m, k, x, scratch = entry
        if "trellis1" in info:
            args_ = (info["trellis1"], info["trellis2"], info["tlut"], x, m, batch, k, info["tlut_bits"], info["KV"][0], info["KV"][1], 2)
            return oracle.cpu_tcq_linear_fused(*args_) if fused else oracle.cpu_tcq_linear(*args_, scratch)
        if "trellis" in info:
            args_ = (info["trellis"], None, info["tlut"], x, m, batch, k, info["tlut_bits"], info["KV"], 0, 0)
            return oracle.cpu_tcq_linear_fused(*args_) if fused else oracle.cpu_tcq_linear(*args_, scratch)
        args_ = (info["qweight"], info["lut"], x, m, batch, k, info["lut_bits"], info["vec_sz"])
        return oracle.cpu_lut_tc_linear_fused(*args_) if fused else oracle.cpu_lut_tc_linear(*args_, scratch)

    def timed(fn, reps):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), len(ts)

    nl = nl or len(layers)
    visible, ncores, why = usable_cpus()
    ncores = max(1, min(ncores, oracle.num_threads()))
    oracle.set_num_threads(ncores)  # threads actually used = the CPUs this process may really run on (cgroup quota), not nproc
    variants = {}
    run_one(host[0], False)  # warm-up: page faults, table init
    budget = max(2.0, seconds) / 3.0
    for fused in (False, True):
        def layer_run(f=fused):
            for e in host:
                run_one(e, f)
        t1, _ = timed(layer_run, 1)
        reps = max(10, min(50, int(budget / max(t1, 1e-3))))
        t_layer, reps = timed(layer_run, reps)
        variants[("fused" if fused else "materialise") + "_all_cores"] = {
            "value": 1.0 / (t_layer * nl), "unit": "tokens/s", "cores": ncores, "median_layer_s": t_layer, "layer_runs": reps}
    small = min(host, key=lambda e: e[1] * e[2])
    frac = small[1] * small[2] / sum(e[1] * e[2] for e in host)
    oracle.set_num_threads(1)
    try:
        for fused in (False, True):
            t_small, reps = timed(lambda f=fused: run_one(small, f), 3)
            variants[("fused" if fused else "materialise") + "_1_thread"] = {
                "value": frac / (t_small * nl), "unit": "tokens/s", "cores": 1, "median_linear_s": t_small, "runs": reps,
                "extrapolation": f"{small[1]}x{small[2]} linear = {frac:.4f} of a layer's weights"}
    finally:
        oracle.set_num_threads(ncores)
    best = variants["materialise_all_cores"]
    one = variants["materialise_1_thread"]["value"]
    speedup = best["value"] / one if one > 0 else None
    return {"value": best["value"], "unit": "tokens/s", "cores": ncores, "kind": "port",
            "host_cpus_visible": visible, "cores_source": why,
            # how well the all-core figure scales: the decode is table-driven scalar code, the GEMV a stream over W — beyond the
            # host's memory channels more threads buy little (round 3: 5.7x on 128 threads).  The one-thread figure is beside it.
            "one_thread_value": one, "speedup_over_one_thread": speedup,
            "parallel_efficiency": (speedup / ncores) if speedup else None,
            "sample": f"1 of {nl} layers ({len(host)} linears, batch {batch}), median of {best['layer_runs']} full-layer runs; "
                      f"per-token time = {nl} x per-layer time",
            "variants": variants}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path is the product; there is no CPU fallback)")
    if args.force_device >= 0:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    # Who is in this run: backend, world size and every rank's device as the RANK sees it (index, PCI address, name) — so that a
    # multi-GPU record shows N ranks on N distinct devices (and a rehearsal with every rank on one card shows that too).
    prop = torch.cuda.get_device_properties(device)
    me = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device": torch.cuda.current_device(),
          "pci_bus_id": "%04x:%02x:%02x.0" % (getattr(prop, "pci_domain_id", 0), getattr(prop, "pci_bus_id", 0), getattr(prop, "pci_device_id", 0)),
          "uuid": str(getattr(prop, "uuid", "")), "name": prop.name, "pid": os.getpid()}
    seen = [me]
    if world > 1:
        seen = [None] * world
        dist.all_gather_object(seen, me)
    ranks_seen = {"backend": (dist.get_backend() if world > 1 else None), "world_size": world, "ranks": seen,
                  "distinct_devices": len({(r_["pci_bus_id"], r_["uuid"]) for r_ in seen})}

    import qpalette_amd as qp
    qp._native.lib()

    model_key, qstr = WORKLOADS[args.workload]
    nlayers = args.layers or qp.mem_op.get_layer_info(model_key)["nlayers"]
    tp = args.parallel == "tp" and world > 1
    torch.manual_seed(1234)
    layers = build_model(qp, torch, model_key, qstr, nlayers, device, shard=(rank, world) if tp else None,
                         distinct_codebooks=args.distinct_codebooks, packing=args.packing)
    n = args.batch
    xs = {}
    for groups in layers:
        for mod, k, _ in (u for grp in groups for u in grp):
            if k not in xs:
                xs[k] = torch.randn(n, k, device=device).half()
    gather = None
    if tp:
        if args.gather == "peer":
            hidden = max(m.out_features for groups in layers for grp in (groups[1], groups[3]) for m, _, _ in grp)
            gather = qp.shard.PeerGatherer(world, rank, device, max_bytes=n * hidden * 4, slots=2 * len(layers) + 2)
        else:
            gather = qp.shard.make_gatherer(world, device)

    main_stream = torch.cuda.Stream(device)
    side = [torch.cuda.Stream(device) for _ in range(2)] if args.streams >= 3 else []

    had = qp.hadamard
    inc = []

    def build_incoherent_state():
        assert args.launch == "multi" and gather is None, "--incoherent runs on the multi-job launch path"
        gen = torch.Generator(device=device).manual_seed(4321)
        for groups in layers:
            per = []
            for grp in groups:
                k = grp[0][1]
                hadK, K = had.get_hadK(k)
                per.append({
                    "su": (torch.randint(0, 2, (k,), device=device, generator=gen) * 2 - 1).half(),
                    "hadK": None if hadK is None else hadK.T.contiguous().half().to(device), "K": K,
                    "wscale": [(0.01 + 0.02 * torch.rand(m.out_features, device=device, generator=gen)).half()
                               for m, _, _ in grp]})
                mods = [m for m, _, _ in grp]
                if (len(per) == 3 and n == 1 and K == 1 and not args.no_fuse_rotation and not args.no_swiglu_epilogue
                        and len(mods) == 2 and type(mods[0]) is type(mods[1]) and qp.linear.rotation_fusable(mods, n)
                        and type(mods[0]) in qp.linear._PACKED_KEYS and mods[0].out_features == mods[1].out_features
                        and qp.linear._codec_key(mods[0]) == qp.linear._codec_key(mods[1])):
                    gate, up = mods  # the group is (gate, up)
                    per[-1]["il"] = qp.linear.interleave_up_gate(up, gate)
                    per[-1]["il_w"] = qp.linear.interleave_rows(per[-1]["wscale"][1], per[-1]["wscale"][0])
                    qp.share_codebooks([per[-1]["il"], up, gate])
            inc.append(per)

    if args.incoherent:
        build_incoherent_state()

    nrot = [0]  # rotation launches of one token (launches of their own: qpal_hadamard)

    def k28_ok(mod, k, K):
        if n != 1 or not qp.ops.can_fuse_rotation(1, k, K) or type(mod) not in qp.linear._PACKED_KEYS:
            return False
        if isinstance(mod, qp.VQLinearPackTensorCore):
            idx = mod.lut_bits if mod.vec_sz == 2 else (2 * mod.lut_bits if mod.lut_bits <= 6 else mod.lut_bits)
            return (4 << (idx + min(15 - idx, 5))) >= 40 * 1024
        return qp.linear._codec_key(mod)[0] != "single"

    def token_incoherent():
        """[rotate -> one multi-job GEMV with fused Wscale*scale] x 4 per layer; down_proj's rotation also applies
        SwiGLU to the up|gate buffer the previous launch wrote."""
        outs = []
        scale = 64.0
        nrot[0] = 0
        for groups, per in zip(layers, inc):
            pre, ug, act_signed = {}, None, False
            for gi, (grp, pi) in enumerate(zip(groups, per)):
                mods = [m for m, _, _ in grp]
                wsc = pi["wscale"]
                if gi == 2 and len(mods) == 2:  # the group is (gate, up); the SwiGLU rotation reads up | gate
                    mods, wsc = mods[::-1], wsc[::-1]
                kw = dict(wscales=wsc, oscale=scale)
                if gi == 2 and "il" in pi:
                    # up | gate as one layer with interleaved supertile rows: the launch writes fp16 silu(gate) * up itself
                    act = torch.empty((n, pi["il"].out_features // 2), dtype=torch.float16, device=device)
                    nxt = groups[3][0][0].out_features
                    if not args.no_prezero:
                        pre[3] = torch.empty((n, nxt), dtype=torch.float32, device=device)
                    dmod, dpi = groups[3][0][0], per[3]
                    fuse28 = (dpi["K"] > 1 and not args.no_fuse_rotation and len(groups[3]) == 1 and k28_ok(dmod, groups[3][0][1], dpi["K"]))
                    # (with the fused 14336-wide rotation the epilogue also applies down_proj's sign vector: exact, and the rotation
                    # inside every down_proj workgroup then reads one vector instead of two)
                    qp.multi_gemv([pi["il"]], xs[grp[0][1]], wscales=[pi["il_w"]], oscale=scale, x_rot=(pi["su"], 1 / scale),
                                  prezero=pre.get(3), act_out=act, act_su=dpi["su"] if fuse28 else None)
                    act_signed = fuse28
                    ug = act
                    outs.append(act)
                    continue
                if (gi == 3 and ug.dtype == torch.float16 and pi["K"] > 1 and not args.no_fuse_rotation and len(mods) == 1
                        and k28_ok(mods[0], grp[0][1], pi["K"])):
                    # the 28 x 512 rotation of the activation inside down_proj's own x staging (csrc/rot_k28.h)
                    xr = ug
                    kw["x_rot"] = (None if act_signed else pi["su"], 1 / scale, pi["hadK"], pi["K"])
                elif gi == 3:
                    xr = had.rotate(ug, hadK=pi["hadK"], K=pi["K"], su=pi["su"], post_scale=1 / scale,
                                    in_mode=had.IN_F16 if ug.dtype == torch.float16 else had.IN_SWIGLU_F32)
                    nrot[0] += 1
                elif pi["K"] == 1 and not args.no_fuse_rotation and qp.linear.rotation_fusable(mods, n):
                    xr = xs[grp[0][1]]
                    kw["x_rot"] = (pi["su"], 1 / scale)
                else:
                    xr = had.rotate(xs[grp[0][1]], hadK=pi["hadK"], K=pi["K"], su=pi["su"], post_scale=1 / scale)
                    nrot[0] += 1
                widths = [m.out_features for m in mods]
                if gi in (0, 2):
                    buf = torch.empty((n, sum(widths)), dtype=torch.float32, device=device)
                    nxt = groups[gi + 1][0][0].out_features
                    if not args.no_prezero:
                        pre[gi + 1] = torch.empty((n, nxt), dtype=torch.float32, device=device)
                        kw["prezero"] = pre[gi + 1]
                    qp.multi_gemv(mods, xr, outs=list(buf.split(widths, dim=1)), **kw)
                    if gi == 2:
                        ug = buf
                    outs.append(buf)
                elif gi in pre:
                    outs += qp.multi_gemv(mods, xr, outs=[pre[gi]], outs_zeroed=True, **kw)
                else:
                    outs += qp.multi_gemv(mods, xr, **kw)
        return outs

    plain_token, parts = make_token(qp, torch, layers, xs, n, device, launch=args.launch if not args.incoherent else "multi",
                                    no_prezero=args.no_prezero, gather=gather, side=side, main_stream=main_stream)

    def token():
        return token_incoherent() if args.incoherent else plain_token()

    graph = None
    with torch.cuda.stream(main_stream):
        token()  # registers ops, sizes the allocator
        torch.cuda.synchronize()
        if not args.no_graph and (not tp or isinstance(gather, qp.shard.PeerGatherer)):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=main_stream):
                token()
        run = graph.replay if graph is not None else token

        def sync_all():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize()

        for _ in range(args.warmup):
            run()
        sync_all()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(main_stream)
        for _ in range(args.steps):
            run()
        e1.record(main_stream)
        sync_all()
        wall = time.perf_counter() - t0
        dev_s = e0.elapsed_time(e1) / 1e3

    # Per launch kind (N = 1, multi-job launches, after the timed region): the four dependent launches of a block timed one
    # kind at a time — 32 launches of that kind, each with its in-token arguments, as one graph — so that the launch
    # furthest from the roofline can be read from the driver's record.
    by_kind = None
    if world == 1 and not args.incoherent and args.launch == "multi" and graph is not None and not args.no_kind_breakdown:
        try:
            by_kind = {}
            with torch.cuda.stream(main_stream):
                kinds = list(enumerate(KINDS)) + ([((0, 1), "q|k|v+o"), ((2, 3), "gate|up+down")] if os.environ.get("QPAL_KIND_PAIRS") else [])
                for gi, kname in kinds:
                    ktoken, kowned = make_token(qp, torch, layers, xs, n, device, launch="multi", no_prezero=args.no_prezero,
                                                only_kind=gi)
                    ktoken()
                    torch.cuda.synchronize()
                    kg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(kg, stream=main_stream):
                        ktoken()
                    for _ in range(3):
                        kg.replay()
                    torch.cuda.synchronize()
                    # Inside the token every output block is zeroed by the launch BEFORE it; a graph of one kind alone has no such
                    # launch, and the pair-mode / split-K atomics of o / down / gate|up would pile replay upon replay onto stale sums.
                    # The harness zeroes the kind's blocks between the replays, OUTSIDE the timed spans (one event pair per replay),
                    # so the launches timed here run on exactly the inputs they have inside the token.
                    gis = gi if isinstance(gi, tuple) else (gi,)
                    kblocks = [per[g_] for per in (kowned or []) for g_ in gis if per[g_] is not None]
                    reps = max(5, min(args.steps, 50))
                    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
                    for k0, k1 in evs:
                        for blk in kblocks:
                            blk.zero_()
                        k0.record(main_stream)
                        kg.replay()
                        k1.record(main_stream)
                    torch.cuda.synchronize()
                    kbytes = algorithmic_bytes(qp, [[groups[g_] for g_ in gis] for groups in layers], n)
                    nk = sum(len(qp.linear.launch_groups([m for m, _, _ in groups[g_]], mixed_kv=n <= 8)) for groups in layers for g_ in gis)
                    t_k = sum(k0.elapsed_time(k1) for k0, k1 in evs) / 1e3 / reps
                    by_kind[kname] = {"launches": nk, "bytes_per_launch": kbytes / nk, "us_per_launch": t_k / nk * 1e6,
                                      "achieved_GBps": kbytes / t_k / 1e9, "frac": kbytes / t_k / 1e9 / HBM_PEAK_GBS}
        except Exception as exc:  # the headline line must not depend on this leg
            by_kind = {"error": repr(exc)}

    calib = None
    if world == 1 and not args.incoherent and not args.no_calibration and graph is not None:
        try:
            calib = calibrate(qp, torch, layers, n, device, main_stream, args.steps)
        except Exception as exc:  # the headline line must not depend on this leg
            calib = {"error": repr(exc)}

    # Second figure (N = 1 only, after the timed region of the headline): the same token with every projection group
    # inside the reference's incoherence wrapper (rotation + scales), i.e. what an IncoherentMLP / attention forward costs.
    extra = None
    if world == 1 and not args.incoherent and not args.no_incoherent_extra and args.launch == "multi" and graph is not None:
        try:
            build_incoherent_state()
            with torch.cuda.stream(main_stream):
                token_incoherent()
                torch.cuda.synchronize()
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, stream=main_stream):
                    token_incoherent()
                for _ in range(args.warmup):
                    g2.replay()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    g2.replay()
                torch.cuda.synchronize()
                w2 = time.perf_counter() - t1
            extra = {"value": args.steps * n / w2, "unit": "tokens/s", "ms_per_step": w2 / args.steps * 1e3,
                     "rotation_launches_per_token": nrot[0],
                     "what": "same token, every projection group behind sign flip + Hadamard + scales (rotation fused into the GEMV x "
                             "staging: k = 4096 on the matrix pipe, the 28 x 512 transform of down_proj's 14336-wide input likewise; SwiGLU "
                             "in the gate|up epilogue)"}
        except Exception as exc:  # the headline line must not depend on this leg
            extra = {"error": repr(exc)}

    if world > 1:
        t = torch.tensor([wall, dev_s], device=device if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_s = float(t[0]), float(t[1])

    tokens = args.steps * n * (1 if tp else world)
    value = tokens / wall
    nlinear = sum(len(grp) for groups in layers for grp in groups)
    multi = args.launch == "multi" and not tp
    mixed_kv = n <= 8 and not args.incoherent
    nphase = sum(len(qp.linear.launch_groups([m for m, _, _ in grp], mixed_kv=mixed_kv)) if multi else len(grp)
                 for groups in layers for grp in groups)  # dependent multi-job GEMVs per token
    nlaunch = nphase  # GEMV kernel launches per token
    abytes = algorithmic_bytes(qp, layers, n) * (world if tp else 1)  # per token
    t_token = dev_s / args.steps
    achieved = abytes / (world if tp else 1) / t_token / 1e9  # per GPU

    out = {
        "metric": "decode tokens/s bs=1 Llama-3.1-8B @3.25b; achieved HBM GB/s vs peak",
        "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if tp else "weak", "vs_baseline": None, "dtype": "f16 weights x f16 activations, f32 accumulate",
        "data": "synthetic (random packed bits, random activations, " + ("one random codebook per linear" if
                args.distinct_codebooks else "one random codebook shared by all layers as in real checkpoints") + ")",
        "config": {"workload": f"{args.workload}: {nlayers} layers, {nlinear} quantized linears ({qstr}), batch {n}, "
                               f"{'HIP-graph replay' if graph is not None else 'eager'}, {args.streams} stream(s)",
                   "parallelism": (f"tp{world} row-sharded + " + ("one-shot peer-write gather (xGMI)" if isinstance(gather, qp.shard.PeerGatherer)
                                                                   else "all-gather (torch.distributed)") if tp else f"dp{world} replicas"),
                   "linears_per_token": nlinear, "launches_per_token": nlaunch, "phases_per_token": nphase,
                   "launch_mode": args.launch,
                   "outputs": ("fresh per launch (--no-prezero): the library adds a memset node where it splits K" if args.no_prezero else
                               "owned by the harness, every launch zeroes the next launch's block (the library may split K / share rows "
                               "between workgroups without memset nodes; QPAL_SHARE=" + os.environ.get("QPAL_SHARE", os.environ.get("QPAL_PAIR", "1")) + ")"),
                   "incoherent": bool(args.incoherent),
                   "rotation_launches_per_token": nrot[0] if args.incoherent else 0,
                   "ranks_seen": ranks_seen},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": load_traffic(args.workload),
                     "traffic_source": traffic_source(),
                     "kernel": "qpal::tc_gemv_kernel (every GEMV launch of a token)",
                     "algorithmic_bytes_per_launch": abytes / (world if tp else 1) / nlaunch,
                     "avg_launch_us": t_token / nlaunch * 1e6,
                     "avg_phase_us": t_token / nphase * 1e6},
    }
    if by_kind is not None:
        out["roofline"]["by_launch_kind"] = by_kind
    if calib is not None:
        out["roofline"].update(calib)
        if calib.get("measured_stream_GBps"):
            out["roofline"]["frac_of_measured"] = achieved / calib["measured_stream_GBps"]
        if calib.get("stream_token_ms"):
            out["roofline"]["frac_of_stream_token"] = calib["stream_token_ms"] / (t_token * 1e3)
    if n > 8:  # skinny GEMM / decode + fp16 GEMM: a dense contraction, priced against the matrix pipe as well
        flops = 2.0 * n * sum(m.out_features * k for groups in layers for grp in groups for m, k, _ in grp)
        out["roofline_mfma"] = {"bound": "mfma", "achieved": flops / t_token / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                                "frac": flops / t_token / 1e12 / 2500.0,
                                "note": "batch <= 64: fused decode + MFMA skinny GEMM (one decode feeds every batch group); above: decode "
                                        "to fp16 W in HBM (staged 16-byte stores) + fp16 GEMM (hipBLASLt)"}
    if extra is not None:
        out["with_incoherence_wrapper"] = extra
    if isinstance(gather, qp.shard.PeerGatherer):
        assert gather.error() == 0, "a peer gather gave up waiting for a flag: the figures above are invalid"
        out["config"]["peer_gather"] = {"flag_memory": gather.flag_memory, "buffer_memory": gather.buffer_memory, "world": world}
    if os.environ.get("QPAL_BENCH_SHARED_WEIGHTS"):
        # experiment knob (cache-resident weights): NOT the benchmark — say so where the number is read
        out["config"]["INVALID_experiment_knob"] = "QPAL_BENCH_SHARED_WEIGHTS: every layer aliases layer 0's buffers"
        out["metric"] = "EXPERIMENT (shared weights), not the headline metric"
    printed = [False]
    tp_failed = [False]
    watchdog = None
    if world > 1 and not args.no_tp_leg and not tp:
        # BASELINE configs[4]: Llama-3.1-70B shapes @3.0 b row-sharded over the N GPUs, batch 1 and batch 16 (strong scaling),
        # after the headline's timed region; every rank takes part (collectives), rank 0 reports.
        # The headline line must not depend on this leg: a rank that raises leaves the others inside a collective, and a
        # collective that never completes would take the (already measured) headline with it — so a watchdog bounds the leg
        # and everything after it: on expiry rank 0 prints the line with the leg marked abandoned, and every rank leaves.
        import threading

        fail_rc = 0 if args.lenient_exit else TP_ABANDONED_RC

        def abandon():
            # every rank says where it stands (stderr): the stage of the leg and, if a peer gather is alive, its slot / epoch view
            try:
                where = _TP_STATE.get("stage", "?")
                peer_ = _TP_STATE.get("peer")
                print(f"[bench rank {rank}] tp_70b leg abandoned after {args.tp_timeout} s at stage '{where}'"
                      + (f"; peer gather: {peer_.describe_wait()}" if peer_ is not None else "; no peer gather alive"),
                      file=sys.stderr, flush=True)
            except Exception as exc:  # noqa: BLE001 (diagnostics must not keep the process alive)
                print(f"[bench rank {rank}] tp_70b leg abandoned (no state: {exc!r})", file=sys.stderr, flush=True)
            if rank == 0 and not printed[0]:
                printed[0] = True
                out["tp_70b"] = {"error": f"abandoned after {args.tp_timeout} s (a rank failed or a collective did not complete); "
                                          "the headline above was measured before this leg"}
                out["tp_70b_abandoned"] = True   # (top level: a run record can flag it without parsing the leg)
                print(json.dumps(out), flush=True)
                print(f"[bench] tp_70b leg ABANDONED after {args.tp_timeout} s; exit code {fail_rc} (--lenient-exit: 0)", file=sys.stderr, flush=True)
                if os.environ.get("QPAL_BENCH_HEADLINE_FILE") and os.path.exists(os.environ["QPAL_BENCH_HEADLINE_FILE"]):
                    os.remove(os.environ["QPAL_BENCH_HEADLINE_FILE"])
            os._exit(fail_rc)

        park = os.environ.get("QPAL_BENCH_HEADLINE_FILE") if rank == 0 else None
        if park:
            with open(park, "w") as f:
                json.dump(out, f)
        watchdog = threading.Timer(args.tp_timeout, abandon)
        watchdog.daemon = True
        watchdog.start()
        try:
            out["tp_70b"] = tp_leg(qp, torch, dist, args, rank, world, device)
        except Exception as exc:  # the leg raised on this rank: the headline still goes out, the run fails
            out["tp_70b"] = {"error": repr(exc)}
            out["tp_70b_abandoned"] = True
            tp_failed[0] = True
            print(f"[bench rank {rank}] tp_70b leg raised at stage '{_TP_STATE.get('stage', '?')}': {exc!r}", file=sys.stderr, flush=True)
    if rank == 0:
        if world == 1 and not args.no_whole_model and not args.incoherent and n == 1:
            # the reference's own metric (eval/measure_latency.py:236-272 times whole-model generate()): the fused decode step of
            # perf/decode_llama.py — embedding, 32 decoder blocks (6 launches each), final norm + lm_head + argmax — short context
            try:
                out["whole_model_decode"] = whole_model_leg(args)
            except Exception as exc:
                out["whole_model_decode"] = {"error": repr(exc)}
        if (world == 1 and not args.no_other_configs and not args.incoherent and n == 1 and args.launch == "multi"
                and args.workload == "llama3.1-8b_tcomb_6_7" and not args.no_graph):
            try:
                out["other_configs"] = other_configs_leg(qp, torch, device, args, main_stream)
            except Exception as exc:
                out["other_configs"] = {"error": repr(exc)}
        if not args.no_cpu_baseline and world == 1:
            layers = build_model(qp, torch, model_key, qstr, 1, device, distinct_codebooks=args.distinct_codebooks,
                                 packing=args.packing, keep_infos=True)
            out["cpu_baseline"] = cpu_baseline(qp, layers, n, args.cpu_seconds, nlayers)
        if not printed[0]:
            printed[0] = True
            print(json.dumps(out), flush=True)
            park = os.environ.get("QPAL_BENCH_HEADLINE_FILE")
            if park and os.path.exists(park):
                os.remove(park)
    if world > 1:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # (a rank that failed in the leg above is not at this barrier; the line is out)
            pass
    if watchdog is not None:
        watchdog.cancel()
    if tp_failed[0] and not args.lenient_exit:
        sys.stdout.flush()
        raise SystemExit(TP_ABANDONED_RC)


OTHER_CONFIGS = [  # (key, workload, batch): BASELINE.json configs[2], [3] (both published qdicts) and [4] on ONE GPU
    ("llama3.1-8b_mem3p25", "llama3.1-8b_mem3p25", 1),
    ("llama3.1-8b_figure1d", "llama3.1-8b_figure1d", 1),
    ("llama3.1-8b_figure1c", "llama3.1-8b_figure1c", 1),
    ("llama3.1-70b_tcq_6_bs1", "llama3.1-70b_tcq_6", 1),
    ("llama3.1-70b_tcq_6_bs16", "llama3.1-70b_tcq_6", 16),
    # the batched fused path (north_star "MFMA ... batched dequant-then-GEMM") on the headline model: one launch per projection group
    ("llama3.1-8b_tcomb_6_7_bs64", "llama3.1-8b_tcomb_6_7", 64),
    ("llama3.1-8b_tcomb_6_7_bs128", "llama3.1-8b_tcomb_6_7", 128),
]


def other_configs_leg(qp, torch, device, args, stream):
    """The other BASELINE configs in the driver-timed line: the same token harness as the headline (multi-job
    launches, harness-owned pre-zeroed outputs, one HIP graph per token), <= 10 replays each between two HIP events, after the
    headline's timed region.  value = tokens/s (batch x steps / time); frac = algorithmic bytes per token / time / 8 TB/s."""
    res, cache = {}, {}
    steps = max(3, min(args.steps, 10))
    for key, wl, nb in OTHER_CONFIGS:
        t_leg = time.perf_counter()
        try:
            model_key, qstr = WORKLOADS[wl]
            nl = args.layers or qp.mem_op.get_layer_info(model_key)["nlayers"]
            if wl not in cache:
                cache.clear()  # (one model alive at a time beside the headline's)
                torch.cuda.empty_cache()
                torch.manual_seed(1234)
                cache[wl] = build_model(qp, torch, model_key, qstr, nl, device, packing=args.packing)
            layers = cache[wl]
            xs = {}
            for groups in layers:
                for mod, k, _ in (u for grp in groups for u in grp):
                    if k not in xs:
                        xs[k] = torch.randn(nb, k, device=device).half()
            token, _ = make_token(qp, torch, layers, xs, nb, device, launch="multi")
            with torch.cuda.stream(stream):
                token()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    token()
                for _ in range(2):
                    g.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(steps):
                    g.replay()
                e1.record(stream)
                torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 1e3 / steps
            abytes = algorithmic_bytes(qp, layers, nb)
            nlaunch = sum(len(qp.linear.launch_groups([m for m, _, _ in grp], mixed_kv=nb <= 8)) for groups in layers for grp in groups)
            res[key] = {"value": nb / t, "unit": "tokens/s", "ms_per_step": t * 1e3, "batch": nb, "steps": steps, "layers": nl,
                        "frac": abytes / t / 1e9 / HBM_PEAK_GBS, "launches_per_token": nlaunch,
                        "leg_s": round(time.perf_counter() - t_leg, 1)}
            del g, token
        except Exception as exc:  # the headline line must not depend on this leg
            res[key] = {"error": repr(exc)}
    cache.clear()
    torch.cuda.empty_cache()
    return res


_TP_STATE = {}  # where the tp_70b leg stands on this rank (stage text, live PeerGatherer): read by the watchdog's post-mortem


def whole_model_leg(args):
    """tokens/s of a whole Llama-3.1-8B-shaped greedy decode step on this library's fused decoder-block glue (the same quantizer
    as the headline workload where it is a single scheme; the published fusion-aware qdict otherwise), KV cache of 1024 positions."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("_qpal_decode_llama", os.path.join(ROOT, "perf", "decode_llama.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    model_key, qstr = WORKLOADS[args.workload]
    argv = ["--model", model_key, "--no-modular", "--tokens", str(max(64, min(args.steps, 256))), "--context", "1024"]
    argv += ["--qdict", qstr[6:]] if qstr.startswith("qdict:") else ["--quantizer", qstr]
    if args.layers:
        argv += ["--layers", str(args.layers)]
    r = mod.main(argv, quiet=True)
    return {"value": 1e3 / r["ms_fused_glue"], "unit": "tokens/s", "ms_per_step": r["ms_fused_glue"],
            "launches_per_token": r["launches_per_token"], "context": 1024,
            "what": "whole-model greedy decode step, batch 1: embedding + every decoder block on the fused glue (RMSNorm + rotation "
                    "inside the q|k|v and up|gate launches, rope + cache append + attention as one launch, the 28 x 512 rotation and "
                    "the residual adds inside o / down) + final norm, fp16 lm_head and argmax as one launch; HIP-graph replay; "
                    "the reference times the same thing as generate() (eval/measure_latency.py:236-272)"}


def tp_leg(qp, torch, dist, args, rank, world, device):
    """Llama-3.1-70B-shaped token (uniform tcq_6 = 3.0 b/w, BASELINE configs[4]) with every linear row-sharded over the `world`
    ranks: multi-job launches on each rank's shard, the outputs of o_proj / down_proj all-gathered.  Figures at
    batch 1 and batch 16, each with the one-shot peer-write gather inside a HIP graph — but only after PeerGatherer.validate()
    has agreed with the library collective on this node — and with dist.all_gather_into_tensor (RCCL) between eager launches
    as the checked baseline.  value = the faster VALID one."""
    t_leg = time.perf_counter()

    def log(msg):  # progress on stderr (rank 0): where a slow or abandoned leg spent its time; every rank remembers its stage
        _TP_STATE["stage"] = msg
        if rank == 0:
            print(f"[bench tp_70b +{time.perf_counter() - t_leg:6.1f} s] {msg}", file=sys.stderr, flush=True)

    model_key, qstr = WORKLOADS["llama3.1-70b_tcq_6"]
    nl = args.tp_layers or qp.mem_op.get_layer_info(model_key)["nlayers"]
    steps = args.tp_steps or max(5, min(args.steps, 30))
    layers = build_model(qp, torch, model_key, qstr, nl, device, shard=(rank, world))
    hidden = max(m.out_features for groups in layers for grp in (groups[1], groups[3]) for m, _, _ in grp) * world
    res = {"workload": f"llama3.1-70b_tcq_6: {nl} layers, row-sharded over {world} ranks ({qstr})", "world": world,
           "backend": args.dist_backend}
    def all_ranks_ok(ok):
        """every rank must take the same branch (the branches contain collectives)"""
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t[0]))

    log(f"{nl} layers of this rank's shard built")
    peer, err = None, None
    try:
        peer = qp.shard.PeerGatherer(world, rank, device, max_bytes=16 * hidden * 4 // world + 4096, slots=2 * nl + 2)
    except Exception as exc:  # (PeerGatherer's set-up collectives have completed or failed on every rank alike)
        err = repr(exc)
    if all_ranks_ok(peer is not None):
        res["peer_gather"] = {"flag_memory": peer.flag_memory, "buffer_memory": peer.buffer_memory,
                              "validated_against_collective": bool(peer.validate())}
    else:
        res["peer_gather"] = {"error": err or "set-up failed on another rank", "validated_against_collective": False}
        peer = None
    _TP_STATE["peer"] = peer
    log(f"peer gather: {res['peer_gather']}")
    stream = torch.cuda.Stream(device)

    def timed(token, graphable):
        with torch.cuda.stream(stream):
            token()
            torch.cuda.synchronize()
            run = token
            if graphable:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    token()
                run = g.replay
            # (the eager collective token is the checked baseline, 160 library collectives per step: a few steps are enough,
            # and the whole leg has to fit its watchdog also where the ranks share one card)
            nsteps = steps if graphable else min(steps, 5)
            for _ in range(3 if graphable else 1):
                run()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nsteps):
                run()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t[0]) / nsteps

    try:  # (the peers' mapped allocations are released also when a leg raises)
        for nb in (1, 16):
            xs = {}
            for groups in layers:
                for mod, k, _ in (u for grp in groups for u in grp):
                    if k not in xs:
                        xs[k] = torch.randn(nb, k, device=device).half()
            abytes = algorithmic_bytes(qp, layers, nb)  # per rank
            fig = {}
            if peer is not None and res["peer_gather"].get("validated_against_collective"):
                token, _ = make_token(qp, torch, layers, xs, nb, device, launch="multi", gather=peer)
                token()
                peer.finish_token()  # (>= 2 call sites per token: checked on the token that is about to be captured and replayed)
                t = timed(token, True)
                torch.cuda.synchronize()
                log(f"batch {nb}: peer gather in graph {t * 1e3:.2f} ms per step")
                if all_ranks_ok(peer.error() == 0):
                    fig["peer_gather_in_graph"] = {"tokens_per_s": nb / t, "ms_per_step": t * 1e3}
                else:
                    fig["peer_gather_in_graph"] = {"error": "a peer gather gave up waiting for a flag"}
                    res["peer_gather"]["validated_against_collective"] = False
            coll = qp.shard.make_gatherer(world, device)
            token, _ = make_token(qp, torch, layers, xs, nb, device, launch="multi", gather=coll)
            t = timed(token, False)
            log(f"batch {nb}: collective between eager launches {t * 1e3:.2f} ms per step")
            fig["collective_eager"] = {"tokens_per_s": nb / t, "ms_per_step": t * 1e3}
            valid = [v for v in fig.values() if "tokens_per_s" in v]
            best = max(valid, key=lambda v: v["tokens_per_s"])
            fig["value"] = best["tokens_per_s"]
            fig["unit"] = "tokens/s"
            fig["ms_per_step"] = best["ms_per_step"]
            fig["roofline_frac_per_gpu"] = abytes / (best["ms_per_step"] / 1e3) / 1e9 / HBM_PEAK_GBS
            res[f"bs{nb}"] = fig
    finally:
        _TP_STATE["peer"] = None
        if peer is not None:
            peer.close()
    return res


def load_traffic(workload):
    """HBM bytes per launch from the rocprofv3 PMC pass committed under profiles/ (None if absent).  NOT measured by this run:
    the JSON line names the file it was read from (roofline.traffic_source)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            data = json.load(f)
        # the committed figure belongs to ONE version of the kernels: another library (QPAL_VERSION bumped with the kernels) -> null,
        # not a stale number
        import qpalette_amd as qp
        if data.get("_qpal_version") is not None and data["_qpal_version"] != qp._native.lib().qpal_version():
            return None
        return data.get(workload)
    except (OSError, ValueError):
        return None


def traffic_source():
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get("_source", "profiles/traffic.json (committed rocprofv3 --pmc FETCH_SIZE pass; not measured by this run)")
    except (OSError, ValueError):
        return None


if __name__ == "__main__":
    main()
