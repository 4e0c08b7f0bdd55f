"""Multi-GPU execution of the bi-level forward: one process per GPU, RCCL over xGMI.

The lower level is communication-free: molecules are independent connected
components of the block-diagonal merged graph (src/merged_graph.py:51-66), so rank r
owns a contiguous range of drugs (balanced by ATOM count, prefix-sum split) and runs
the lower stack + readout on them.  The one exchange of the path is an all-gather of
the pooled drug embeddings -- the distributed form of the per-gid ``init_x`` row
writes (model/layers_aggregation.py:68-74) -- after which every rank holds the full
``init_x`` and runs the (small, replicated) interaction-level pass and scores ITS
shard of the edge-pair batch.  Shards have unequal drug counts, so rows are padded to
the largest shard for one fixed-size ``all_gather_into_tensor`` (AllGatherV by padding:
one collective, no per-rank message-size negotiation).
"""

import numpy as np
import torch
import torch.distributed as dist

__all__ = ["shard_ranges", "all_gather_rows", "shard_pairs", "ShardedLower"]


def shard_ranges(sizes, world_size):
    """Contiguous molecule ranges [(start, end)] with near-equal atom counts."""
    sizes = np.asarray(sizes, dtype=np.int64)
    csum = np.concatenate([[0], np.cumsum(sizes)])
    total = csum[-1]
    cuts = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        c = int(np.searchsorted(csum, target, side="left"))
        c = min(max(c, cuts[-1]), len(sizes))
        cuts.append(c)
    cuts.append(len(sizes))
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def shard_pairs(num_pairs, world_size, rank):
    """Contiguous, near-equal shard of the edge-pair batch for ``rank``."""
    base, extra = divmod(num_pairs, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_rows(local, counts, group=None, out=None):
    """Concatenate per-rank row blocks ``local`` [counts[rank], D] in rank order.
    One collective; with a single rank it is the identity (no copy)."""
    world = len(counts)
    if world == 1:
        if out is not None:
            out.copy_(local)
            return out
        return local
    D = local.shape[1]
    cmax = int(max(counts))
    if all(int(c) == cmax for c in counts):
        buf = out if out is not None else torch.empty((cmax * world, D), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(buf, local.contiguous(), group=group)
        return buf
    padded = torch.zeros((cmax, D), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]].copy_(local)
    buf = torch.empty((world * cmax, D), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    buf = buf.view(world, cmax, D)
    res = out if out is not None else torch.empty((int(sum(counts)), D), dtype=local.dtype, device=local.device)
    o = 0
    for r, c in enumerate(counts):
        res[o:o + int(c)].copy_(buf[r, :int(c)])
        o += int(c)
    return res


class ShardedLower(object):
    """Rank-local lower-level sweep + the all-gather.  ``lower_fn(gids) -> [len(gids), D]``
    is the single-GPU lower pass (Model with use_layers='lower_layers' on a BatchData of
    those molecules); it is injected so that the collective logic can be exercised
    under ``gloo`` on CPU."""

    def __init__(self, gids, sizes, rank, world_size, group=None):
        self.rank, self.world, self.group = rank, world_size, group
        self.ranges = shard_ranges(sizes, world_size)
        self.counts = [e - s for s, e in self.ranges]
        s, e = self.ranges[rank]
        self.local_gids = list(gids[s:e])

    def gather_init_x(self, pooled_local, out=None):
        assert pooled_local.shape[0] == self.counts[self.rank]
        return all_gather_rows(pooled_local, self.counts, self.group, out=out)
