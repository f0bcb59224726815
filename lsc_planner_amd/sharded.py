"""Agent-sharded stepping: one process per GPU, the trajectory table replicated, one all-gather per tick.

Within a tick every agent's problem depends only on LAST tick's trajectories of all agents
(src/multi_sync_simulator.cpp:249-304 freezes the inputs before anybody plans), so contiguous blocks of agents
are planned independently per rank and the only exchange is an all-gather of the new trajectories
(N * 360 B per tick) -- torch.distributed over RCCL/xGMI on GPUs, gloo in the CPU tests.
"""
import numpy as np


def shard_bounds(n_agents, world_size, rank):
    """Contiguous blocks, the first (n % world) ranks get one extra agent."""
    base, extra = divmod(n_agents, world_size)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return first, count


def all_gather_rows(dist, full, first, count, counts=None):
    """In-place exchange: `full` is [N, ...]; this rank has written rows [first, first+count).
    Equal shards use all_gather_into_tensor (one fused collective); ragged shards fall back to all_gather."""
    import torch
    if dist is None or not dist.is_initialized():
        return
    world = dist.get_world_size()
    if world == 1:
        return
    mine = full[first:first + count].clone()      # own rows, detached from the receive buffer
    if counts is None or len(set(counts)) == 1:
        dist.all_gather_into_tensor(full.view(-1), mine.view(-1))
        return
    # ragged shards: pad every shard to the largest one so that a single equal-sized collective still does the job
    maxc = max(counts)
    row = int(np.prod(full.shape[1:]))
    send = torch.zeros((maxc, row), dtype=full.dtype, device=full.device)
    send[:count] = mine.view(count, row)
    recv = torch.empty((world, maxc, row), dtype=full.dtype, device=full.device)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1))
    off = 0
    for r, c in enumerate(counts):
        full[off:off + c] = recv[r, :c].view((c,) + tuple(full.shape[1:]))
        off += c


class ShardedSwarm:
    """Generic sharded tick loop.  `tick_fn(state, goal, traj_prev, planner_seq, first, count) -> traj rows of the
    shard` is the HIP path on GPUs and the oracle in the gloo CPU tests; everything else is identical."""

    def __init__(self, dist, n_agents, tick_fn, propagate_fn, device="cpu"):
        import torch
        self.torch = torch
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.N = n_agents
        self.first, self.count = shard_bounds(n_agents, self.world, self.rank)
        self.counts = [shard_bounds(n_agents, self.world, r)[1] for r in range(self.world)]
        self.tick_fn, self.propagate_fn = tick_fn, propagate_fn
        self.device = device
        self.planner_seq = 0

    def step(self, state, goal, traj_prev, traj_next):
        """One synchronous tick: plan own shard into traj_next, exchange, propagate all states."""
        self.planner_seq += 1
        self.tick_fn(state, goal, traj_prev, traj_next, self.planner_seq, self.first, self.count)
        if self.world > 1:
            all_gather_rows(self.dist, traj_next, self.first, self.count, self.counts)
        self.propagate_fn(traj_next, state)
        return traj_next
