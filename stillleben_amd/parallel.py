"""Multi-GPU plumbing of the path: scenes are independent units, so each rank
works on its own shard (no collective inside the path); the single exchange step is the
all-gather of the rendered batches, one collective per dtype buffer (RCCL over xGMI on GPUs,
gloo in the CPU tests)."""
import torch


def shard_seeds(rank, world, n_items, batch):
    """Seeds of the scenes rank `rank` processes: item k of rank r covers
    [(r * n_items + k) * batch, ... + batch) -- disjoint across ranks and items."""
    return [[(rank * n_items + k) * batch + i for i in range(batch)] for k in range(n_items)]


def shard_scenes(n_scenes, rank, world):
    """Static partition of a fixed scene list (strong-scaling use): scene s -> rank s % world."""
    return list(range(rank, n_scenes, world))


def cu_partition_streams(settle_cus, n_settle_streams=1, device=None):
    """Streams for a generation loop that settles batch k+1 while it renders batch k: `n_settle_streams`
    streams confined to the first `settle_cus` compute units and one render stream confined to the rest
    (slhip_stream_create_cu_range, include/slhip.h).  Returns (settle_streams, render_stream) as
    torch ExternalStreams; the caller keeps them alive."""
    import ctypes as C

    from . import _abi

    L = _abi.lib()
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    total = torch.cuda.get_device_properties(dev).multi_processor_count
    if not 0 < settle_cus < total:
        raise ValueError("settle_cus must be in (0, %d)" % total)

    def make(first, count):
        h = C.c_void_p()
        with torch.cuda.device(dev):
            _abi.check(L.slhip_stream_create_cu_range(first, count, C.byref(h)), "slhip_stream_create_cu_range")
        return torch.cuda.ExternalStream(h.value, device=dev)

    settle = [make(0, settle_cus) for _ in range(max(1, n_settle_streams))]
    return settle, make(settle_cus, total - settle_cus)


class _EventWork:
    """What `all_gather(..., async_op=True)` returns for the C-ABI transport: `wait()` orders the CURRENT
    stream after the collective (same meaning as torch's Work.wait() for RCCL)."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class SlhipComm:
    """slhip_comm (include/slhip.h): an RCCL communicator owned by libslhip.so, one rank per GPU.  The
    128-byte id is drawn on rank 0 and distributed through torch.distributed (any backend) -- or pass
    `unique_id` when the ranks exchange it some other way; `world == 1` needs no peer at all."""

    def __init__(self, rank, world, dist=None, unique_id=None, device=None):
        import ctypes as C

        from . import _abi

        self.L = _abi.lib()
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if unique_id is None:
            buf = (C.c_uint8 * _abi.COMM_ID_BYTES)()
            if world > 1 and dist is None:
                raise ValueError("SlhipComm: world > 1 needs `dist` (or a shared `unique_id`)")
            box = [None]
            if rank == 0:
                # a failure here (librccl not loadable through the C-ABI) must still reach the broadcast below: the other ranks
                # are already waiting in it, and all ranks have to raise together
                st = self.L.slhip_comm_unique_id(buf)
                box = [bytes(buf) if st == 0 else None]
                err = self.L.slhip_last_error() if st != 0 else None
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            if box[0] is None:
                raise _abi.SlhipError("slhip_comm_unique_id failed on rank 0%s" % ((": " + err.decode()) if rank == 0 and err else ""))
            unique_id = box[0]
        idbuf = (C.c_uint8 * _abi.COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _abi.check(self.L.slhip_comm_create(idbuf, world, rank, C.byref(h)), "slhip_comm_create")
        self.handle = h
        self.stream = torch.cuda.Stream(device=self.device)   # collectives run beside the render stream

    def all_gather(self, tensors, outs):
        """outs[i] ([world * B, ...]) <- every rank's tensors[i], one fused RCCL group, enqueued on the
        communicator's stream AFTER the work already queued on the current stream; returns an _EventWork."""
        import ctypes as C

        from . import _abi

        n = len(tensors)
        send = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
        recv = (C.c_void_p * n)(*[t.data_ptr() for t in outs])
        nbytes = (C.c_uint64 * n)(*[t.numel() * t.element_size() for t in tensors])
        produced = torch.cuda.Event()
        produced.record()
        self.stream.wait_event(produced)
        with torch.cuda.device(self.device):
            st = self.L.slhip_allgather_group(self.handle, n, send, recv, nbytes, C.c_void_p(self.stream.cuda_stream))
        _abi.check(st, "slhip_allgather_group")
        done = torch.cuda.Event()
        done.record(self.stream)
        for t in list(tensors) + list(outs):
            t.record_stream(self.stream)
        return _EventWork(done)

    def close(self):
        if self.handle is not None:
            self.stream.synchronize()
            self.L.slhip_comm_destroy(self.handle)
            self.handle = None


class BatchGatherer:
    """all_gather_into_tensor of a list of per-rank tensors into a small ring of persistent
    [world, ...] staging buffers (`depth` sets per distinct shape signature).

    Synchronous use: ``views = gather(tensors)``.
    Overlapped use (bench.py): ``views, works = gather(tensors, async_op=True)`` right after the
    producer kernels were enqueued on the current stream -- the collective is ordered after them on
    RCCL's own stream; call ``w.wait()`` on the stream that will overwrite `tensors` (or read the
    views) to order that stream after the collective.  Collectives are issued in program order, which
    is the same on every rank.  A staging set is reused after `depth` further calls with the same
    shapes; consumers read it before that (the collectives themselves are serialised on the RCCL
    stream, so a reuse never races with an earlier gather into the same set)."""

    def __init__(self, dist, world, depth=2, comm=None):
        """`comm`: a SlhipComm -- the collectives then go through the C-ABI (slhip_allgather_group, RCCL owned by
        libslhip.so); without it torch.distributed moves the bytes (gloo in the CPU tests)."""
        self.dist, self.world, self.depth = dist, world, max(1, int(depth))
        self.comm = comm
        self.rings = {}

    def _staging(self, tensors):
        key = tuple((tuple(t.shape), t.dtype, str(t.device)) for t in tensors)
        ring = self.rings.get(key)
        if ring is None:
            # concatenated layout [world * B, ...] (accepted by both RCCL and gloo), viewed as [world, B, ...]
            ring = {"next": 0, "sets": [[torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype,
                                                     device=t.device) for t in tensors] for _ in range(self.depth)]}
            self.rings[key] = ring
        bufs = ring["sets"][ring["next"]]
        ring["next"] = (ring["next"] + 1) % self.depth
        return bufs

    def __call__(self, tensors, async_op=False):
        if self.comm is not None:
            bufs = self._staging(tensors)
            work = self.comm.all_gather([t.contiguous() for t in tensors], bufs)
            views = [g.view((self.world, -1) + tuple(g.shape[1:])) for g in bufs]
            if async_op:
                return views, [work]
            work.wait()
            return views
        if self.dist is None or self.world == 1:
            views = [t.unsqueeze(0) for t in tensors]
            return (views, []) if async_op else views
        bufs = self._staging(tensors)
        works = []
        for t, g in zip(tensors, bufs):
            # an all-gather is type-agnostic: move bytes (RCCL/gloo have no int16 datatype)
            w = self.dist.all_gather_into_tensor(g.view(torch.uint8), t.contiguous().view(torch.uint8), async_op=async_op)
            if async_op:
                works.append(w)
        views = [g.view((self.world, -1) + tuple(g.shape[1:])) for g in bufs]
        return (views, works) if async_op else views


class ChunkedGatherer:
    """Streams a rank's WHOLE output through the exchange step, `piece` scenes at a time: every piece is one fused all-gather
    (BatchGatherer: slhip_allgather_group over RCCL, or torch.distributed) into a double-buffered staging set, issued right after
    the kernels that produced the chunk -- the pieces of chunk k travel while chunk k + 1 renders.  Staging costs
    2 x world x piece scenes whatever the step's size (the full ground truth of 16384 scenes per rank would be 1.6 TB gathered).
    `on_piece(first_scene, views, works)` consumes a gathered piece (the views are valid until two further pieces of the same
    shape have been issued); without it the pieces are only moved (bench.py measures the exchange, the trainer reads them)."""

    def __init__(self, gatherer, piece):
        self.g, self.piece = gatherer, max(1, int(piece))
        self.bytes_sent = 0          # bytes this rank handed to the exchange since construction
        self.pieces = 0

    def __call__(self, tensors, on_piece=None):
        n = int(tensors[0].shape[0])
        works = []
        for p0 in range(0, n, self.piece):
            part = [t[p0:p0 + self.piece] for t in tensors]
            views, w = self.g(part, async_op=True)
            works.extend(w)
            self.bytes_sent += sum(t.numel() * t.element_size() for t in part)
            self.pieces += 1
            if on_piece is not None:
                on_piece(p0, views, w)
        return works
