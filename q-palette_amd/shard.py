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
        # CombLinearTCQ: rows [0, out_part[0]) at KV[0] in trellis1, the rest at KV[1] in trellis2.  The shard [r0, r1) takes
        # its rows from whichever halves it overlaps; the result is again a comb layer (possibly with an empty half, which
        # the module handles as two single-stream ops — lib/linear/comb_linear.py:91-102)
        per_row16 = k // 16
        p0 = info["out_part"][0]
        a0, a1 = min(r0, p0), min(r1, p0)            # rows of half 1
        b0, b1 = max(r0, p0) - p0, max(r1, p0) - p0  # rows of half 2 (relative)
        out["trellis1"] = info["trellis1"][a0 // 16 * per_row16: a1 // 16 * per_row16].contiguous()
        out["trellis2"] = info["trellis2"][b0 // 16 * per_row16: b1 // 16 * per_row16].contiguous()
        out["out_part"] = (a1 - a0, b1 - b0)
    elif "qweight" in info:  # [m, bits*k/32/vec]
        out["qweight"] = info["qweight"][r0:r1].contiguous()
    else:
        raise ValueError("unknown linear_info kind")
    return out


def make_gatherer(world, device=None, group=None):
    """-> f(y_local [n, m_local]) = y [n, m] (concatenation over ranks in rank order) on torch.distributed collectives: equal
    shard widths use one all_gather_into_tensor, ragged widths (out_features not a multiple of 32 * world) an all_gather of
    padded slices.  The widths of a call site are exchanged once (first token) and cached; call f.new_token() at the start
    of every token so that call sites keep their index."""
    if world == 1:
        return lambda y: y
    sites, state = {}, {"i": 0}

    def gather(y):
        i = state["i"]
        state["i"] += 1
        n, ml = y.shape
        if i not in sites:
            widths = [None] * world
            dist.all_gather_object(widths, int(ml), group=group)
            sites[i] = widths
        widths = sites[i]
        if len(set(widths)) == 1:
            buf = torch.empty((world * n, ml), dtype=y.dtype, device=y.device)  # rank-major concatenation
            dist.all_gather_into_tensor(buf, y.contiguous(), group=group)
            return buf.view(world, n, ml).permute(1, 0, 2).reshape(n, world * ml)
        return gather_ragged(y, widths, group=group)

    def new_token():
        state["i"] = 0

    gather.new_token = new_token
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


class PeerGatherer:
    """All-gather by direct peer writes over xGMI (C-ABI qpal_peer_gather, csrc/peer_gather.hip): one kernel per call, inside
    the captured graph, no library collective.  Every rank allocates `slots` gather buffers and one flag block and opens
    every peer's through IPC handles exchanged with torch.distributed (any backend: the data path never touches it again).

    Memory: the flag blocks are polled by a running kernel while a REMOTE GPU writes them, so they live in fine-grained
    device memory (qpal_peer_alloc kind 1; uncached as second choice) — ordinary hipMalloc memory is only guaranteed coherent
    at kernel boundaries.  `flag_memory` says which kind was obtained ("fine-grained" / "uncached" / "coarse-grained": the
    last only if the platform refused both, and then `validated` matters all the more).  The gather buffers are fine-grained
    as well (`buffer_memory`; round 5): their visibility to the consumer launch does not rest on what a kernel boundary does
    to coarse-grained memory that a REMOTE agent wrote.

    Call sites: each call inside a token takes the next slot (buffers and flags are never shared between call sites).  A fast
    rank can start token T+1 while a slow peer still runs token T, but it cannot pass the token's LAST call site before that
    peer has launched its own last call — which is stream-ordered behind the peer's consumers of every EARLIER site.  So the
    view returned for call site s may be overwritten only once its consumers are done, PROVIDED s is not the only site:
    a token needs >= 2 call sites (checked by finish_token() on the token just issued — call it at the end of a token that is
    captured and replayed — and by the next new_token()); with a single site per token use two gatherers alternately.
    Call `new_token()` before re-running or capturing a token so that the same call sites map to the same slots.
    RCCL's all_gather (`make_gatherer`) stays the correctness baseline: `validate()` compares the two on this node."""

    def __init__(self, world, rank, device, max_bytes, slots=64, group=None):
        import ctypes
        from . import _native as nat
        self.nat = nat
        self.world, self.rank, self.device = world, rank, torch.device(device)
        self.max_bytes = (max_bytes + 15) // 16 * 16
        self.slots = slots
        self._next = 0
        self._sites_last_token = None
        self._group = group
        self._widths = {}  # call site -> per-rank slice widths (exchanged once, during the first, un-captured token)
        self.validated = None
        lib = nat.lib()
        nbuf = slots * world * self.max_bytes
        nws = slots * nat.PEER_WS_BYTES_PER_SLOT
        self._own, self._opened = [], []

        def alloc(nbytes, kinds):
            with torch.cuda.device(self.device):
                for kind in kinds:
                    ptr = ctypes.c_void_p()
                    if lib.qpal_peer_alloc(ctypes.byref(ptr), nbytes, kind) == 0:
                        h = ctypes.create_string_buffer(nat.IPC_HANDLE_BYTES)
                        if lib.qpal_ipc_export(ptr, h) == 0:
                            self._own.append(ptr.value)
                            return ptr.value, h.raw, kind
                        lib.qpal_peer_free(ptr)
            raise nat.QpalError("PeerGatherer: no shareable device memory (qpal_peer_alloc / qpal_ipc_export failed)")

        # Round 5: the gather buffers are fine-grained too.  Remote GPUs write them over xGMI while kernels of THIS GPU run, and the
        # consumer reads a slot that the previous token's consumer may have left in this GPU's L2: ordinary (coarse-grained) device
        # memory is only guaranteed coherent with a remote writer at a SYSTEM-scope acquire, and a kernel boundary inside a
        # stream / graph is an agent-scope one.  Fine-grained memory takes the question away (remote writes are visible to loads
        # issued after the flag's system-scope acquire, whatever this GPU cached before); the slices are 2-57 KB, so what the
        # uncached reads cost is not measurable beside the launch.  `buffer_memory` says which kind was obtained.
        self._bufs_ptr, hb, bkind = alloc(nbuf, (1, 2, 0))
        self._ws_ptr, hw, kind = alloc(nws, (1, 2, 0))
        names = {1: "fine-grained", 2: "uncached", 0: "coarse-grained"}
        self.flag_memory, self.buffer_memory = names[kind], names[bkind]
        self._calls = 0   # host-side count of gather launches ISSUED (a replayed graph re-runs captured ones without the host)
        everyone = [None] * world
        dist.all_gather_object(everyone, (hb, hw), group=group)
        self.peer_bufs, self.peer_ws = [], []
        with torch.cuda.device(self.device):
            for r, (pb, pw) in enumerate(everyone):
                if r == rank:
                    self.peer_bufs.append(self._bufs_ptr)
                    self.peer_ws.append(self._ws_ptr)
                    continue
                ptrs = []
                for h in (pb, pw):
                    ptr = ctypes.c_void_p()
                    nat.check(lib.qpal_ipc_open(h, ctypes.byref(ptr)), "qpal_ipc_open")
                    self._opened.append(ptr.value)
                    ptrs.append(ptr.value)
                self.peer_bufs.append(ptrs[0])
                self.peer_ws.append(ptrs[1])
        # torch views of this rank's own allocations (results are views into bufs; error() reads ws)
        self.bufs = _tensor_from_ptr(self._bufs_ptr, nbuf, self.device, self)
        self.ws = _tensor_from_ptr(self._ws_ptr, nws, self.device, self)
        # (a view, not a copy: as_tensor on another device than the one it deduces from the pointer would copy silently, and the
        # gather's results and its error word would then be read from the copy)
        if self.bufs.data_ptr() != self._bufs_ptr or self.ws.data_ptr() != self._ws_ptr:
            raise nat.QpalError("PeerGatherer: the torch views of the shared allocations are copies, not views")
        dist.barrier(group=group)

    def __del__(self):
        # best effort: unmap the peers' allocations of a gatherer nobody closed (no barrier here: close() is the orderly way)
        try:
            lib = self.nat.lib()
            for p in getattr(self, "_opened", []):
                lib.qpal_ipc_close(p)
            self._opened = []
        except Exception:
            pass

    def close(self):
        """Unmap the peers' allocations and free this rank's (after a barrier: nobody may still be writing here)."""
        lib = self.nat.lib()
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            for p in self._opened:
                lib.qpal_ipc_close(p)
            self._opened = []
            dist.barrier(group=self._group)
            for p in self._own:
                lib.qpal_peer_free(p)
            self._own = []
        self.bufs = self.ws = None  # views of freed memory

    _ONE_SITE = ("PeerGatherer: a token with ONE call site may overwrite a slice a slow peer still reads; use >= 2 "
                 "call sites per token (or two gatherers alternately)")

    def new_token(self):
        """Start of a token's call sites.  A token that ended with ONE call site is refused HERE, i.e. before a second such
        token can be issued (and finish_token() refuses it at once — call it at the end of a token that is captured once and
        replayed: new_token() runs only at capture)."""
        if self._next == 1:
            raise RuntimeError(self._ONE_SITE)
        self._sites_last_token = self._next if self._next else self._sites_last_token
        self._next = 0

    def finish_token(self):
        """End of a token: the >= 2 call sites rule checked on the token just issued (a captured graph replays exactly this)."""
        if self._next == 1:
            raise RuntimeError(self._ONE_SITE)

    def __call__(self, y):
        import ctypes
        n, ml = y.shape
        y = y.contiguous()
        nbytes = y.numel() * y.element_size()
        assert nbytes % 16 == 0 and nbytes <= self.max_bytes and self._next < self.slots
        slot = self._next
        self._next += 1
        self._calls += 1
        if slot not in self._widths:
            widths = [None] * self.world
            dist.all_gather_object(widths, int(ml), group=self._group)
            self._widths[slot] = widths
        widths = self._widths[slot]
        # every rank writes at rank * seg inside the slot's buffer; seg = the widest slice, so equal widths land contiguously
        seg = max(widths) * n * y.element_size()
        assert seg % 16 == 0 and seg <= self.max_bytes, "padded slice: 16-byte multiple that fits the slot"
        base = slot * self.world * self.max_bytes
        arr_b = (ctypes.c_void_p * self.world)(*[b + base for b in self.peer_bufs])
        arr_w = (ctypes.c_void_p * self.world)(*self.peer_ws)
        with torch.cuda.device(self.device):
            lib = self.nat.lib()
            if len(set(widths)) == 1:
                rc = lib.qpal_peer_gather(y.data_ptr(), nbytes, slot, arr_b, arr_w, self.rank, self.world,
                                          torch.cuda.current_stream(self.device).cuda_stream)
            else:  # ragged: pad the slice to the widest (the kernel's layout is rank * bytes)
                pad = torch.zeros((n, max(widths)), dtype=y.dtype, device=y.device)
                pad[:, :ml] = y
                rc = lib.qpal_peer_gather(pad.data_ptr(), seg, slot, arr_b, arr_w, self.rank, self.world,
                                          torch.cuda.current_stream(self.device).cuda_stream)
        self.nat.check(rc, "qpal_peer_gather")
        out = self.bufs[base: base + self.world * seg].view(y.dtype).view(self.world, n, max(widths))
        if len(set(widths)) == 1:
            return out.reshape(1, self.world * ml) if n == 1 else out.permute(1, 0, 2).reshape(n, self.world * ml)
        return torch.cat([out[r, :, :w] for r, w in enumerate(widths)], dim=1)

    def describe_wait(self, timeout_s=2.0):
        """One line for a post-mortem (bench.py prints it from every rank when it abandons a multi-GPU leg): the host's view (call
        sites of the current token, launches issued) and — read from the device on a helper thread, given up after `timeout_s`, the
        card may be wedged — per slot the epoch this rank has reached and the peers whose flag is still behind it."""
        import threading
        host = (f"rank {self.rank}/{self.world}: next call site {self._next} of {self.slots} slots, sites in the last token "
                f"{self._sites_last_token}, gather launches issued by the host {self._calls}")
        box = {}

        def read():
            try:
                w = self.ws.view(torch.int32).view(self.slots, -1).cpu()
                behind = []
                for s_ in range(self.slots):
                    ep = [int(w[s_, 16 + p]) for p in range(self.world)]
                    late = [p for p in range(self.world) if int(w[s_, p]) - ep[p] < 0]
                    if late:
                        behind.append(f"slot {s_}: epoch {max(ep)}, flags behind from ranks {late}")
                    if int(w[s_, 32]) != 0:
                        behind.append(f"slot {s_}: a bounded wait gave up at epoch {int(w[s_, 32])}")
                box["dev"] = "; ".join(behind[:6]) if behind else "no slot is waiting for a peer"
            except Exception as exc:  # noqa: BLE001 (diagnostics only)
                box["dev"] = f"device state unreadable ({exc!r})"

        t = threading.Thread(target=read, daemon=True)
        t.start()
        t.join(timeout_s)
        return host + " | device: " + box.get("dev", f"no answer within {timeout_s} s (a kernel is still running or the card is wedged)")

    def error(self):
        """!= 0 after a synchronisation: a wait for a peer's flag gave up."""
        return int(self.ws.view(torch.int32).view(self.slots, -1)[:, 32].abs().sum().item())

    def validate(self, n=1, width=2048, rounds=8, dtype=torch.float32):
        """Compare the peer gather with the library collective (dist.all_gather_into_tensor: RCCL on a real node) on THIS node's
        links: `rounds` tokens of two call sites each, eager and from a replayed HIP graph, every word checked, plus the
        bounded-wait error word.  Uses (and then releases) the first two slots.  -> True / False, the same on every rank."""
        assert self.slots >= 2 and n * width * torch.empty((), dtype=dtype).element_size() <= self.max_bytes
        dev = self.device
        saved = (self._next, self._sites_last_token, dict(self._widths))
        self._widths = {}  # the validation's own slice widths for the two slots it borrows
        self._next = 0
        srcs = [torch.empty(n, width, dtype=dtype, device=dev) for _ in range(2)]
        ok = True

        def fill(step):
            for i, s_ in enumerate(srcs):
                s_.copy_((torch.arange(n * width, device=dev, dtype=torch.float32).view(n, width) * 0.5
                          + 1000.0 * self.rank + 10000.0 * i + 100000.0 * step).to(dtype))

        def reference():
            outs = []
            for s_ in srcs:
                buf = torch.empty((self.world * n, width), dtype=dtype, device=dev)
                dist.all_gather_into_tensor(buf, s_.contiguous(), group=self._group)
                outs.append(buf.view(self.world, n, width).permute(1, 0, 2).reshape(n, self.world * width))
            return outs

        def token():
            self.new_token()
            return [self(s_).clone() for s_ in srcs]

        stream = torch.cuda.Stream(dev)
        with torch.cuda.stream(stream):
            for step in range(rounds // 2):
                fill(step)
                torch.cuda.synchronize(dev)
                got = token()
                torch.cuda.synchronize(dev)
                ok = ok and all(torch.equal(g, r) for g, r in zip(got, reference()))
            dist.barrier(group=self._group)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                got = token()
            for step in range(rounds // 2, rounds):
                fill(step)
                torch.cuda.synchronize(dev)
                graph.replay()
                torch.cuda.synchronize(dev)
                ok = ok and all(torch.equal(g, r) for g, r in zip(got, reference()))
        ok = ok and self.error() == 0
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        gathered = [None] * self.world
        dist.all_gather_object(gathered, int(flag.item()), group=self._group)
        self._next, self._sites_last_token, self._widths = saved[0], saved[1], saved[2]
        self.validated = all(gathered)
        return self.validated


def _tensor_from_ptr(ptr, nbytes, device, owner):
    """uint8 torch view of a raw device allocation (kept alive by `owner`)."""
    class _Mem:
        pass
    m = _Mem()
    m.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
    m._owner = owner
    t = torch.as_tensor(m, device=device)
    t._qpal_owner = m
    return t
