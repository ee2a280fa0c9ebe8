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
