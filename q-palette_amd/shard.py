"""Row-sharding of packed linears across GPUs (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The path shards by output rows: packed rows are independent in units of one supertile row (32 rows),
exactly the unit `merge_infos` concatenates.  Every rank keeps 1/world of every packed
buffer plus the whole (tiny) codebook, computes its slice of y, and the slices are all-gathered where the
next consumer needs the full vector (after o_proj / down_proj: the Hadamard pre-rotation of the next
linear consumes complete vectors — lib/linear/incoherent_linear.py:81, 106, 325, 336)."""
import torch
import torch.distributed as dist

ROW_UNIT = 32  # supertile rows


def shard_rows(m, world):
    """Split m output rows into `world` contiguous shards, each a multiple of 32 rows (earlier ranks take
    the remainder units)."""
    assert m % ROW_UNIT == 0, "out_features must be a multiple of 32"
    units = m // ROW_UNIT
    base, rem = divmod(units, world)
    return [(base + (1 if r < rem else 0)) * ROW_UNIT for r in range(world)]


def shard_bounds(m, world, rank):
    sizes = shard_rows(m, world)
    start = sum(sizes[:rank])
    return start, start + sizes[rank]


def shard_linear_info(info, rank, world):
    """Row shard [r0, r1) of a `linear_info` dict (any of the five module kinds).  Packed buffers are
    sliced along dim 0, which is contiguous per supertile row in every format (tensor-core order:
    [supertile row][supertile col][lane]...; SIMT: plain rows)."""
    m, k = info["out_features"], info["in_features"]
    r0, r1 = shard_bounds(m, world, rank)
    out = dict(info)
    out["out_features"] = r1 - r0
    if "trellis" in info:  # [(m/16)*(k/16), 8*KV]
        per_row16 = k // 16
        out["trellis"] = info["trellis"][r0 // 16 * per_row16: r1 // 16 * per_row16].contiguous()
    elif "in_part" in info:  # combt: two column halves, same rows
        for key, kk in (("trellis1", info["in_part"][0]), ("trellis2", info["in_part"][1])):
            per_row16 = kk // 16
            out[key] = info[key][r0 // 16 * per_row16: r1 // 16 * per_row16].contiguous()
    elif "out_part" in info:
        raise NotImplementedError("row-split comb layers shard per half; shard the two halves separately")
    elif "qweight" in info:  # [m, bits*k/32/vec]
        out["qweight"] = info["qweight"][r0:r1].contiguous()
    else:
        raise ValueError("unknown linear_info kind")
    return out


def make_gatherer(world, device=None, group=None):
    """-> f(y_local [n, m_local]) = y [n, m] (concatenation over ranks in rank order; equal shard sizes use
    one all_gather_into_tensor, ragged sizes fall back to all_gather of padded slices)."""
    if world == 1:
        return lambda y: y

    def gather(y):
        n, ml = y.shape
        buf = torch.empty((world * n, ml), dtype=y.dtype, device=y.device)  # rank-major concatenation
        dist.all_gather_into_tensor(buf, y.contiguous(), group=group)
        return buf.view(world, n, ml).permute(1, 0, 2).reshape(n, world * ml)

    return gather


def gather_ragged(y, sizes, group=None):
    """All-gather row shards of unequal width (m not divisible by 32*world)."""
    world = len(sizes)
    n = y.shape[0]
    mx = max(sizes)
    pad = torch.zeros((n, mx), dtype=y.dtype, device=y.device)
    pad[:, : y.shape[1]] = y
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:, :s] for b, s in zip(bufs, sizes)], dim=1)
