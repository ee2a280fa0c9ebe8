"""N > 1 path on CPU: world_size 2, gloo.  Each rank holds a row shard of a packed linear, produces its slice
of y (the oracle stands in for the GPU GEMV here — this test covers sharding + the collective, not the
kernel) and the slices are all-gathered with the product's gather helpers."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, m, k, qstr, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import qpalette_amd as qp
        from oracle import oracle

        info = qp.mem_op.dummy_linear_info(k, m, qstr, seed=11)  # same seed on every rank = same full layer
        x = torch.randn(2, k, generator=torch.Generator().manual_seed(5)).half()
        mine = qp.shard.shard_linear_info(info, rank, world)

        def weight(i):
            mm = i["out_features"]
            if "trellis1" in i:
                return oracle.tcq_dequant(i["trellis1"].numpy(), i["tlut"].numpy(), mm, k, 9, 6, c2=i["trellis2"].numpy(),
                                          KV2=7, split=2)
            return oracle.lut_tc_dequant(i["qweight"].numpy(), i["lut"].numpy(), mm, k, i["lut_bits"], i["vec_sz"])

        y_local = torch.from_numpy(oracle.gemv(weight(mine), x.numpy())[0]).float()
        sizes = qp.shard.shard_rows(m, world)
        if len(set(sizes)) == 1:
            y = qp.shard.make_gatherer(world)(y_local)
        else:
            y = qp.shard.gather_ragged(y_local, sizes)
        y_full = torch.from_numpy(oracle.gemv(weight(info), x.numpy())[0]).float()
        ok = torch.equal(y, y_full)
        if rank == 0:
            ret.put(bool(ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("m,qstr", [(256, "tcomb_6_7_0.5_none_0.9"), (32 * 5, "ldlq_1_4_none_1.0")])
def test_row_sharded_linear_all_gather(m, qstr):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, m, 256, qstr, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True


# ------------------------------------------------------------------------------------------------------------------------
# The row-sharded (--parallel tp) token of bench.py at world_size 2 and 4: bench.build_model(shard=...) + bench.make_token with
# the product's gatherer over gloo.  The GPU GEMV is replaced by a CPU stand-in (the oracle) — this covers sharding of every
# layer kind of a decoder block, ragged shard widths and the collective, not the kernel.  Shapes: a model with the Llama-70B
# proportions (hidden : kv : intermediate = 8 : 1 : 28) scaled down so that four CPU processes finish in seconds, chosen so
# that hidden (10 units of 32 rows) and intermediate (35 units) do NOT divide by 4 ranks: ragged shard_rows.
def _tp_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        import bench
        import qpalette_amd as qp
        from oracle import oracle

        qp.mem_op.LAYER_INFO["tiny_70b"] = qp.mem_op._llama(2, 320, 128, 1120)
        qstr = "tcq_6_none_0.9"
        n = 1

        def weight(info):
            return oracle.tcq_dequant(info["trellis"].numpy(), info["tlut"].numpy(), info["out_features"], info["in_features"],
                                      info["tlut_bits"], info["KV"])

        infos = {}

        def cpu_multi_gemv(layers, x, **kw):  # stand-in for the GPU launch: float64 GEMV over the oracle's decode, as fp32
            return [torch.from_numpy(oracle.gemv(weight(infos[id(l)]), x.half().numpy())[0]).float() for l in layers]

        qp.multi_gemv = cpu_multi_gemv
        dev = torch.device("cpu")
        torch.manual_seed(1234)
        sharded = bench.build_model(qp, torch, "tiny_70b", qstr, 2, dev, shard=(rank, world), keep_infos=True)
        full = bench.build_model(qp, torch, "tiny_70b", qstr, 2, dev, keep_infos=True)
        for model in (sharded, full):
            for groups in model:
                for grp in groups:
                    for mod, _, info in grp:
                        infos[id(mod)] = info
        gen = torch.Generator().manual_seed(5)
        xs = {k: torch.randn(n, k, generator=gen).half() for k in (320, 1120)}
        gather = qp.shard.make_gatherer(world, dev)
        token, _ = bench.make_token(qp, torch, sharded, xs, n, dev, launch="multi", gather=gather)
        outs = token()
        outs2 = token()  # second token: the call sites' cached widths are reused
        ref_token, _ = bench.make_token(qp, torch, full, xs, n, dev, launch="multi")
        ref = ref_token()
        ok = len(outs) == len(ref)
        i = 0
        for groups_s, groups_f in zip(sharded, full):
            for gi, (gs, gf) in enumerate(zip(groups_s, groups_f)):
                for (ms, _, _), (mf, _, _) in zip(gs, gf):
                    y, r = outs[i], ref[i]
                    if gi in (1, 3):   # o_proj / down_proj: gathered to full width, identical on every rank
                        ok = ok and tuple(y.shape) == (n, mf.out_features) and torch.equal(y, r) and torch.equal(outs2[i], r)
                    else:              # q|k|v, gate|up stay sharded: this rank's rows of the full result
                        r0, r1 = qp.shard.shard_bounds(mf.out_features, world, rank)
                        ok = ok and tuple(y.shape) == (n, r1 - r0) and torch.equal(y, r[:, r0:r1])
                    i += 1
        widths = qp.shard.shard_rows(320, world)
        ret.put((rank, bool(ok), widths))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_tp_token_of_bench_over_gloo(world):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tp_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = sorted(ret.get(timeout=5) for _ in range(world))
    assert [g[1] for g in got] == [True] * world, got
    if world == 4:
        assert len(set(got[0][2])) > 1  # the shard widths of the hidden dimension really are ragged


# ------------------------------------------------------------------------------------------------------------------------
# One-shot peer-write gather (csrc/peer_gather.hip, qpalette_amd.shard.PeerGatherer) rehearsed with TWO processes on ONE GPU:
# the IPC handle exchange, the per-call-site slots, the flag protocol and HIP-graph replay are exactly what an 8-GPU node
# runs; what one GPU cannot show is the xGMI link itself.
def _peer_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import qpalette_amd as qp
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        ml, nsites = 2048, 3
        g = qp.shard.PeerGatherer(world, rank, dev, max_bytes=ml * 4, slots=nsites)
        srcs = [torch.zeros(1, ml, dtype=torch.float32, device=dev) for _ in range(nsites)]

        def token():
            g.new_token()
            return [g(s) for s in srcs]

        def fill(step):
            for i, s in enumerate(srcs):
                s.copy_(torch.arange(ml, dtype=torch.float32, device=dev) + 1000.0 * rank + 10000.0 * i + 100000.0 * step)

        def expect(i, step):
            return torch.cat([torch.arange(ml, dtype=torch.float32) + 1000.0 * r + 10000.0 * i + 100000.0 * step for r in range(world)])[None]

        ok = True
        stream = torch.cuda.Stream(dev)
        with torch.cuda.stream(stream):
            fill(0)
            outs = token()
            torch.cuda.synchronize()
            ok = ok and all(torch.equal(o.cpu(), expect(i, 0)) for i, o in enumerate(outs))
            dist.barrier()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                outs = token()
            for step in range(1, 6):
                fill(step)
                graph.replay()
                torch.cuda.synchronize()
                ok = ok and all(torch.equal(o.cpu(), expect(i, step)) for i, o in enumerate(outs))
        ok = ok and g.error() == 0
        ok = ok and g.flag_memory in ("fine-grained", "uncached", "coarse-grained")
        ok = ok and g.validate(n=2, width=1024)   # against the collective (gloo here, RCCL on a real node), eager + graph replay
        ok = ok and g.error() == 0
        dist.barrier()
        g.close()
        ret.put((rank, bool(ok), g.flag_memory))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_peer_gather_two_ranks_on_one_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_peer_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = sorted(ret.get(timeout=5) for _ in range(world))
    assert [g[1] for g in got] == [True] * world, got
    print("peer gather flag memory:", got[0][2])
