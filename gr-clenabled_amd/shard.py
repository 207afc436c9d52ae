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
    [T][Ng][F][npol][ncomp] (lib/clXEngine_impl.cc:987-1061 with num_inputs = Ng); after
    exchange() every rank holds [T][N][F/W][npol][ncomp] -- the same layout with all N antennas
    and its own F/W channels -- ready for clXEngine(num_inputs=N, num_channels=F/W).xcorrelate().
    Per rank and integration this moves (W-1)/W of its frame buffer once over xGMI.
    """

    def __init__(self, num_inputs, num_channels, integration, npol, ncomp=2, group=None):
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.N, self.F, self.T, self.npol, self.ncomp = num_inputs, num_channels, integration, npol, ncomp
        self.groups = antenna_groups(num_inputs, self.world)
        self.slices = channel_slices(num_channels, self.world)
        self.Ng = num_inputs // self.world
        self.Fw = num_channels // self.world

    def local_shape(self):
        return (self.T, self.Ng, self.F, self.npol, self.ncomp)

    def slab_shape(self):
        return (self.T, self.N, self.Fw, self.npol, self.ncomp)

    def exchange(self, local_frames):
        import torch
        import torch.distributed as dist
        x = local_frames.reshape(self.local_shape())
        if self.world == 1:
            return x.reshape(self.slab_shape()).contiguous()
        W = self.world
        # send block r = my group's frames restricted to rank r's channels
        send = x.reshape(self.T, self.Ng, W, self.Fw, self.npol, self.ncomp).permute(2, 0, 1, 3, 4, 5).contiguous()
        recv = torch.empty_like(send)  # block g = group g's frames restricted to my channels
        dist.all_to_all_single(recv, send, group=self.group)
        # [g][T][Ng][Fw].. -> [T][g*Ng + s][Fw]..
        return recv.permute(1, 0, 2, 3, 4, 5).reshape(self.slab_shape()).contiguous()

    def output_slice(self, rank=None):
        """(first, last) channel of the rank's rows in the full [F][baseline][pol^2] result."""
        return self.slices[self.rank if rank is None else rank]
