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


def mission_list_ids(world_size, rank, missions_per_rank):
    """The reference's node flies a LIST of missions (src/multi_sync_simulator_node.cpp:43-70, src/param.cpp:106-122).  Independent
    missions need no exchange at all, so a list of world_size * missions_per_rank missions is cut into contiguous blocks like the agents of
    a swarm are: rank r flies missions [r K, (r + 1) K) -- K of them in flight together, one launch per tick (lsc_tick_device_fused_batch)."""
    return list(range(rank * missions_per_rank, (rank + 1) * missions_per_rank))


def mission_list_summary(dist, rank, world_size, agents_planned, steps, elapsed_s, tick_p99_ms, device_index, device_uuid="", failed=0):
    """Rank bookkeeping of `bench.py --gpus N --mission-list`: every rank reports what it flew; rank 0 returns the whole-job line
    (sum of the agent-replans over the ranks divided by the SLOWEST rank's time, the per-rank values, the largest per-rank tick p99) and
    checks that world_size processes each drove a device of their own.  No data-path collective: the only communication is this
    report (and the barriers around the timed region).  Ranks other than 0 return None."""
    mine = {"rank": rank, "agents": int(agents_planned), "steps": int(steps), "elapsed_s": float(elapsed_s), "tick_p99_ms": float(tick_p99_ms),
            "device_index": int(device_index), "device_uuid": str(device_uuid), "failed_agents_last_tick": int(failed)}
    if dist is not None and dist.is_initialized() and world_size > 1:
        box = [None] * world_size
        dist.all_gather_object(box, mine)
    else:
        box = [mine]
    if rank != 0:
        return None
    if len(box) != world_size or sorted(b["rank"] for b in box) != list(range(world_size)):
        raise RuntimeError(f"mission list: {world_size} ranks expected, reports from {[b['rank'] for b in box]}")
    if len({b["device_index"] for b in box}) != world_size:
        raise RuntimeError(f"mission list: ranks share a device: {[(b['rank'], b['device_index']) for b in box]}")
    uu = [b["device_uuid"] for b in box if b["device_uuid"]]
    if len(set(uu)) != len(uu):
        raise RuntimeError(f"mission list: two ranks report the same device uuid: {uu}")
    slowest = max(b["elapsed_s"] for b in box)
    total = sum(b["agents"] * b["steps"] for b in box)
    return {"value": total / slowest, "elapsed_s_max_over_ranks": slowest,
            "per_rank_values": [round(b["agents"] * b["steps"] / b["elapsed_s"], 1) for b in sorted(box, key=lambda b: b["rank"])],
            "per_rank_devices": [b["device_index"] for b in sorted(box, key=lambda b: b["rank"])],
            "tick_p99_ms_max_over_ranks": max(b["tick_p99_ms"] for b in box),
            "per_rank_tick_p99_ms": [round(b["tick_p99_ms"], 4) for b in sorted(box, key=lambda b: b["rank"])],
            "agents_in_flight": sum(b["agents"] for b in box),
            "failed_agents_last_tick": sum(b["failed_agents_last_tick"] for b in box)}


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
