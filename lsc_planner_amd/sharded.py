"""Agent-sharded stepping, host-language mirror of the native path (lsc_comm_init / lsc_tick_device_sharded).

Within a tick every agent's problem depends only on LAST tick's trajectories of all agents
(src/multi_sync_simulator.cpp:249-304 freezes the inputs before anybody plans; :297-303 is where every agent receives
the others' trajectories), so blocks of agents are planned independently per rank and the only exchange is one
all-gather of the new trajectory rows.  On GPUs the product does that exchange natively (RCCL, in place, enqueued on
the tick's stream: csrc/lsc_abi.cpp); this module is the same partitioning driven through torch.distributed, used by
the world-size-2 gloo tests on CPU and usable as a torch-only alternative.

Partitioning (identical to lsc_comm_info): shard_rows = ceil(N / world); rank r owns agents
[r*shard_rows, min((r+1)*shard_rows, N)); the table is padded ONCE, at allocation, to shard_rows*world rows so that
the exchange is a single equal-sized collective with no per-tick packing.
"""


def shard_rows(n_agents, world_size):
    return -(-n_agents // world_size)


def table_rows(n_agents, world_size):
    return shard_rows(n_agents, world_size) * world_size


def shard_bounds(n_agents, world_size, rank):
    """(first, count) of rank's block; trailing ranks of a ragged split own fewer (possibly zero) agents."""
    s = shard_rows(n_agents, world_size)
    first = min(rank * s, n_agents)
    return first, min(s, n_agents - first)


def all_gather_rows(dist, table, rank, rows):
    """In-place exchange: `table` is [rows*world, ...] (padded); this rank has written its block [rank*rows, (rank+1)*rows).
    One collective, no staging beyond what the backend needs (NCCL/RCCL gathers in place; gloo wants a detached input)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    mine = table[rank * rows:(rank + 1) * rows]
    if dist.get_backend() != "nccl":
        mine = mine.clone()
    dist.all_gather_into_tensor(table.view(-1), mine.reshape(-1))


class ShardedSwarm:
    """Generic sharded tick loop.  `tick_fn(state, goal, traj_prev, traj_next, planner_seq, first, count)` writes the
    shard's rows of traj_next: the HIP path on GPUs, the oracle in the gloo CPU tests; everything else is identical.
    Trajectory tables are [table_rows][90]; state / goal are [N][...]."""

    def __init__(self, dist, n_agents, tick_fn, propagate_fn, device="cpu"):
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.N = n_agents
        self.rows = shard_rows(n_agents, self.world)
        self.table_rows = self.rows * self.world
        self.first, self.count = shard_bounds(n_agents, self.world, self.rank)
        self.tick_fn, self.propagate_fn = tick_fn, propagate_fn
        self.device = device
        self.planner_seq = 0

    def step(self, state, goal, traj_prev, traj_next):
        """One synchronous tick: plan own shard into traj_next, exchange, propagate all states."""
        self.planner_seq += 1
        self.tick_fn(state, goal, traj_prev[:self.N], traj_next, self.planner_seq, self.first, self.count)
        all_gather_rows(self.dist, traj_next, self.rank, self.rows)
        self.propagate_fn(traj_next[:self.N], state)
        return traj_next
