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


class BatchGatherer:
    """all_gather_into_tensor of a list of per-rank tensors into persistent [world, ...] buffers."""

    def __init__(self, dist, world):
        self.dist, self.world = dist, world
        self.buffers = None

    def __call__(self, tensors):
        if self.dist is None or self.world == 1:
            return [t.unsqueeze(0) for t in tensors]
        if self.buffers is None:
            # concatenated layout [world * B, ...] (accepted by both RCCL and gloo), viewed as [world, B, ...]
            self.buffers = [torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                            for t in tensors]
        for t, g in zip(tensors, self.buffers):
            # an all-gather is type-agnostic: move bytes (RCCL/gloo have no int16 datatype)
            self.dist.all_gather_into_tensor(g.view(torch.uint8), t.contiguous().view(torch.uint8))
        return [g.view((self.world, -1) + tuple(g.shape[1:])) for g in self.buffers]
