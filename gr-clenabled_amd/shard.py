"""Multi-GPU sharding of the hot path (one process per GPU, torch.distributed over RCCL/xGMI).

Only two things shard:

* block replicas (MathOp / FFT / filter / PFB instances): independent instances, one or more per
  GPU, NO data-path collective -- the reference's own model of one OpenCL context per block pinned
  to a device id (lib/GRCLBase.cpp:115-134).  Ranks meet only in barriers and in the
  max-over-ranks timing reduction.
* the X-engine: antenna groups are ingested one group per GPU and exchanged with ONE all-to-all
  (the FX-correlator corner turn): afterwards every rank holds all antennas for its slice of the
  channels and correlates it locally with no further traffic.

torch is plumbing here (device buffers, the RCCL collective); the arithmetic stays in the C ABI.
"""
import os


def rank_env():
    """(rank, world, local_rank) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")))


def replica_assignment(n_instances, world):
    """Independent block instances -> ranks, round robin; returns a list of lists."""
    if n_instances < 0 or world < 1:
        raise ValueError("bad arguments")
    return [[i for i in range(n_instances) if i % world == r] for r in range(world)]


def channel_slices(num_channels, world):
    """Contiguous channel slab of every rank after the corner turn; num_channels % world == 0."""
    if num_channels % world:
        raise ValueError("num_channels (%d) must be a multiple of the number of ranks (%d)" % (num_channels, world))
    per = num_channels // world
    return [(r * per, (r + 1) * per) for r in range(world)]


def antenna_groups(num_inputs, world):
    if num_inputs % world:
        raise ValueError("num_inputs (%d) must be a multiple of the number of ranks (%d)" % (num_inputs, world))
    per = num_inputs // world
    return [(r * per, (r + 1) * per) for r in range(world)]


def max_over_ranks(value, group=None):
    """Max of a Python float over all ranks (timing reduction of bench.py)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


class XEngineCornerTurn:
    """All-to-all of antenna-group frames into per-rank channel slabs.

    Rank g holds its group's integration window in the reference's frame layout
    [T][Ng][F][npol][ncomp] (lib/clXEngine_impl.cc:987-1061 with num_inputs = Ng).  The send buffer is
    [peer][T][Ng][F/W].. (block r = my frames restricted to rank r's channels), packed by ONE strided device copy
    (mi355_pack3d_dev) straight from the frame buffer; the receive buffer is [group][T][Ng][F/W].. and is what
    clXEngine.xcorrelate_device(..., stations_per_group=Ng) reads IN PLACE -- no re-layout pass on either side.
    Per rank and integration this moves (W-1)/W of its frame buffer once over xGMI.

    Two buffer pairs and a side stream: start(i+1) packs and exchanges the next integration while the caller's
    stream correlates integration i.  exchange() is the blocking one-shot form (and, for CPU tensors
    under gloo -- the world-size-2 tests -- the packing is the same index arithmetic done with torch views).

    windows > 1: one exchange carries that many integration windows -- local frames [window][T][Ng][F].., receive buffer
    [group][window][T][Ng][F/W].. -- which clXEngine.xcorrelate_n_device(windows, recv, out, stations_per_group=Ng) correlates
    in ONE launch: after the corner turn a rank holds only F/W channels, too few to fill a device one window at a time (64
    antennas x 128 channels: 28 us per window alone, 7 us in a batch of 8), and the all-to-all messages grow from 2 to 16 MiB.
    """

    def __init__(self, num_inputs, num_channels, integration, npol, ncomp=2, group=None, block=None, windows=1):
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.N, self.F, self.T, self.npol, self.ncomp = num_inputs, num_channels, integration, npol, ncomp
        self.groups = antenna_groups(num_inputs, self.world)
        self.slices = channel_slices(num_channels, self.world)
        self.Ng = num_inputs // self.world
        self.Fw = num_channels // self.world
        self.windows = int(windows)
        if self.windows < 1:
            raise ValueError("windows must be >= 1")
        self.block = block      # any block of this rank's context (its pack3d_device runs the packing kernel)
        self._bufs = {}
        self._side = None

    def _w(self):
        return (self.windows,) if self.windows > 1 else ()

    def local_shape(self):
        return self._w() + (self.T, self.Ng, self.F, self.npol, self.ncomp)

    def grouped_shape(self):
        """Receive buffer: [group]([window])[T][Ng][F/W][npol][ncomp]."""
        return (self.world,) + self._w() + (self.T, self.Ng, self.Fw, self.npol, self.ncomp)

    def slab_shape(self):
        return self._w() + (self.T, self.N, self.Fw, self.npol, self.ncomp)

    # ---- packing -------------------------------------------------------------------------------------------
    def pack(self, local_frames, send):
        """send[r] = local_frames[:, :, rank r's channels] for every peer r."""
        x = local_frames.reshape(self.local_shape())
        esz = x.element_size() * self.npol * self.ncomp
        if x.is_cuda:
            if self.block is None:
                raise RuntimeError("XEngineCornerTurn on GPU tensors needs block= (a gr-clenabled block of this rank) for the packing kernel")
            rows = self.windows * self.T * self.Ng
            self.block.pack3d_device(send, x, self.Fw * esz, rows, self.world, self.F * esz, self.Fw * esz, self.Fw * esz, rows * self.Fw * esz)
        else:  # gloo / CPU tensors (tests/test_multi_gpu_cpu.py): same index arithmetic through a strided view
            rows = self.windows * self.T * self.Ng  # (window, t, station) rows: the packing does not look inside
            send.reshape(self.world, rows, self.Fw, self.npol, self.ncomp).copy_(
                x.reshape(rows, self.world, self.Fw, self.npol, self.ncomp).permute(1, 0, 2, 3, 4))
        return send

    def to_slab(self, grouped):
        """[group]([window])[T][Ng][Fw].. -> the reference layout ([window])[T][N][Fw].. (a copy; only the parity tests and non-fused geometries need it)."""
        g = grouped.reshape(self.world, self.windows, self.T, self.Ng, self.Fw, self.npol, self.ncomp)
        return g.permute(1, 2, 0, 3, 4, 5, 6).reshape(self.slab_shape()).contiguous()

    def _buffers(self, like, slot):
        import torch
        key = (slot, like.device, like.dtype)
        if key not in self._bufs:
            n = self.windows * self.T * self.Ng * self.F * self.npol * self.ncomp
            self._bufs[key] = (torch.empty(n, dtype=like.dtype, device=like.device), torch.empty(n, dtype=like.dtype, device=like.device))
        return self._bufs[key]

    # ---- overlapped form -----------------------------------------------------------------------------------
    def start(self, local_frames, slot=0):
        """Enqueue pack + all-to-all of one exchange (`windows` integration windows) on the side stream for GPU tensors; returns a
        handle for finish().  A slot's receive buffer is overwritten by the next start() on that slot: the caller's consumer of the
        previous result must have been enqueued on the current stream before that start() (it waits for the current stream)."""
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return (None, local_frames, None)
        send, recv = self._buffers(local_frames, slot)
        if local_frames.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream()
            cur = torch.cuda.current_stream()
            self._side.wait_stream(cur)  # the frames were produced on, and the slot's buffers last read by, the caller's stream
            # the side stream reads the caller's tensor and the slot's buffers after this call returns: tell the caching allocator,
            # or a tensor the caller drops right after start() could be handed out again while the packing kernel still reads it
            local_frames.record_stream(self._side)
            send.record_stream(self._side)
            recv.record_stream(self._side)
            with torch.cuda.stream(self._side):
                self.pack(local_frames, send)
                if dist.get_backend(self.group) != "nccl":
                    # device tensors under a backend without a device all-to-all (gloo: tests/test_multi_rank_gpu.py and
                    # `MI355_BENCH_BACKEND=gloo`, several ranks sharing ONE GPU): the packed blocks go through the host.  A test
                    # vehicle for everything around the collective (packing kernel, in-place group-major read, batched windows);
                    # RCCL takes the branch below.
                    self._side.synchronize()
                    hs, hr = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
                    dist.all_to_all_single(hr, hs, group=self.group)
                    recv.copy_(hr)
                    return (None, recv, self._side)
                work = dist.all_to_all_single(recv, send, group=self.group, async_op=True)
            return (work, recv, self._side)
        self.pack(local_frames, send)
        work = dist.all_to_all_single(recv, send, group=self.group, async_op=True)
        return (work, recv, None)

    def finish(self, handle):
        """The caller's stream waits for the exchange; returns the group-major receive buffer (flat)."""
        import torch
        work, recv, side = handle
        if work is not None:
            work.wait()
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        return recv

    # ---- blocking one-shot forms ---------------------------------------------------------------------------
    def exchange_grouped(self, local_frames):
        return self.finish(self.start(local_frames, 0))

    def exchange(self, local_frames):
        """Reference-layout slab [T][N][F/W].. of this rank (one extra copy; the overlapped in-place form is start/finish)."""
        return self.to_slab(self.exchange_grouped(local_frames))

    def output_slice(self, rank=None):
        """(first, last) channel of the rank's rows in the full [F][baseline][pol^2] result."""
        return self.slices[self.rank if rank is None else rank]
