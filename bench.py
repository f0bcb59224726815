#!/usr/bin/env python
"""bench.py -- agent-replans/sec of the MI355X replanning path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one synchronous replan tick of the whole swarm: prediction shift -> LSC construction -> QP solve for
every agent (the reference's MultiSyncSimulator::plan loop), followed by the ideal-state propagation that feeds the
next tick.  Everything stays resident in HBM during the timed region (no host round trip inside a tick).

The timed window (config.tick_window): the mission is fast-forwarded untimed to --start-tick (default 60) so that any --steps
starts inside the crossing of the swarm (ticks ~40-140 of ~216), where the ticks are longest; config.mission_ticks is the tick at
which the last agent arrives (found by flying on, untimed, after the measurement).

Workload (config.workload): BASELINE.json configs[2], the 64-agent circle swap on the empty map
(matlab/mission_generator.m geometry, R = 8 m, z = 1 m, testall_empty.launch parameters incl. mode/goal =
prior_based: goalPlanningWithPriority runs on the device; on maps without a distance field the reference's grid A*
has no observable effect on the planned goal, see DESIGN.md -- and multisim/reset_threshold = 0.15: the disturbance checks
of the reference run every tick, lsc_plan_alt_kernel + the hand-over launch of lsc_general_kernel).
With --gpus G the swarm is G such circles side by side in one world (64*G agents, 30 m pitch; --single-circle: one circle of
radius 8*G), agent-sharded one circle per GPU with one in-place RCCL all-gather of the new trajectories per tick (native:
lsc_tick_device_sharded enqueues plan kernel, ncclAllGather and state propagation on one stream; if that communicator
cannot be created, the torch.distributed all-gather around the same kernels): weak scaling with fixed difficulty per GPU -- a BEST
case, every row between agents of different ranks is culled (config.scaling_note says so; the strong-scaling workloads below keep them).
`python bench.py --gpus G` starts the G ranks itself
(re-executes under torch.distributed.run when WORLD_SIZE is not set); torch.distributed is only the control plane
(rendezvous token, barrier, max-over-ranks of the elapsed time).

--workload random1024 / forest256 time BASELINE configs[4] / configs[3] as SURVEY 8(d) writes them (seeded samplers, the forest
from the committed leaf fixture) as ONE swarm sharded over the G ranks: strong scaling, every cross-rank LSC row active; the line
then carries the per-rank plan-kernel time and the all-gather time, which is where the latency floor shows.

One JSON line on rank 0 with `roofline` (plan kernel), `roofline_sweep{,_large}` (dense LSC sweep: HBM view, bound by the fp64 GJK),
`cpu_baseline` (the oracle = CPU restatement of the reference path, timed on this box's host cores), and -- single GPU, circle workload --
`interior_point_leg` (the timed ticks once more under --solver interior_point, priced with SURVEY 8(d)'s flop model: continuity with
rounds 1-4), `concurrent_missions` (--missions K independent swarms planned by ONE launch per tick, lsc_tick_device_fused_batch) and
`latency_host_abi_ms` (PCIe-inclusive per-tick latency through lsc_replan_tick).

--solver: the QP solver of the fast path (lsc_config.solver).  Default active_set: a dual active-set solve with the interior point as its
fallback; a tick of the headline is TWO launches (the plan kernel and the -- normally empty -- hand-over launch of the alternate-mode
kernel that multisim/reset_threshold > 0 asks for).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak (guide: half the 157.3 TF fp32 vector rate)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec


PMC_FILES = ("profiles/r06_pmc_summary.json", "profiles/r05_pmc_summary.json", "profiles/r04_pmc_summary.json", "profiles/r03_pmc_summary.json")


def pmc_kernel(key):
    """Hardware counters of one kernel (per launch).  NOT measured by this run: counters need rocprofv3, so the values are read
    from the committed PMC passes of this same command (profiles/rNN_pmc_summary.json, made by tools/profile_round.sh and
    profiles/summarize_rocpd.py); the bench line names the file.  ({}, None) when absent."""
    for f in PMC_FILES:
        try:
            return json.load(open(os.path.join(ROOT, f)))["kernels"][key], f
        except Exception:
            continue
    return {}, None


def pmc_traffic(key):
    """HBM-side traffic per launch (bytes): FETCH_SIZE + WRITE_SIZE, raw counter values in KB."""
    d, f = pmc_kernel(key)
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        return int((d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024), f
    return None, None


def pmc_utilisation(key):
    """VALU / LDS utilisation of a kernel from the committed counters (north_star: "LDS/VALU utilisation for the QP solve"):
    wave-cycle shares (SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES: how much of a resident wave's life is spent on that instruction class)
    and the LDS array's own view (SQ_LDS_IDX_ACTIVE = LDS-array cycles, SQ_LDS_BANK_CONFLICT = the extra cycles of conflicts)."""
    d, f = pmc_kernel(key)
    if not d or not d.get("SQ_WAVE_CYCLES"):
        return None
    wc = d["SQ_WAVE_CYCLES"]
    out = {"source": f, "valu_active_share_of_wave_cycles": round(d.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4)}
    if "SQ_ACTIVE_INST_LDS" in d:
        out["lds_active_share_of_wave_cycles"] = round(d["SQ_ACTIVE_INST_LDS"] / wc, 4)
    if "SQ_WAIT_INST_LDS" in d:
        out["lds_wait_share_of_wave_cycles"] = round(d["SQ_WAIT_INST_LDS"] / wc, 4)
    if d.get("SQC_ICACHE_REQ"):
        # the instruction stream (round 6): requests of the instruction cache two CUs share, those that missed (a miss some other wave already
        # has in flight counted apart), and -- an upper bound for everything on the instruction side -- wave-cycles waiting for an instruction to issue
        out["icache_hit_rate"] = round(d.get("SQC_ICACHE_HITS", 0.0) / d["SQC_ICACHE_REQ"], 4)
        out["icache_misses_per_wave"] = round(d.get("SQC_ICACHE_MISSES", 0.0) / max(1.0, d.get("SQ_WAVES", 1.0)), 2)
        out["icache_duplicate_misses_per_wave"] = round(d.get("SQC_ICACHE_MISSES_DUPLICATE", 0.0) / max(1.0, d.get("SQ_WAVES", 1.0)), 2)
        if "SQ_WAIT_INST_ANY" in d:
            out["wait_inst_any_share_of_wave_cycles"] = round(d["SQ_WAIT_INST_ANY"] / wc, 4)
    if d.get("SQ_LDS_IDX_ACTIVE"):
        out["lds_bank_conflict_share_of_lds_cycles"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 4)
        if d.get("SQ_BUSY_CYCLES"):
            out["lds_array_busy_share_of_sq_busy_cycles"] = round(d["SQ_LDS_IDX_ACTIVE"] / d["SQ_BUSY_CYCLES"], 4)
    return out


def weak_scaling_mission(L, G, per_gpu=64, single_circle=False):
    """Workload.  One GPU: BASELINE configs[2], the 64-agent circle swap.  Several GPUs, weak scaling = fixed work per GPU: G
    such circles side by side in ONE world (30 m pitch), rank r owning circle r -- every rank's agents are exactly as hard
    to plan as the single-GPU case, every agent still sees all 64 G agents of the swarm (the far circles' rows are redundant
    and culled, as far agents of any large swarm are), and the exchange moves all of them.  single_circle: one circle of
    64 G agents with R = 8 G instead -- the agents' difficulty then changes with G.
    Returns (mission, description)."""
    n_agents = per_gpu * G
    R = 8.0 * per_gpu / 64.0
    if G == 1 or single_circle:
        R = 8.0 * n_agents / 64.0
        ms = L.circle_swap(n_agents, circle_radius=R, z=1.0, world=(-R - 2, -R - 2, 0, R + 2, R + 2, 2.5))
        return ms, f"{n_agents}-agent generated circle swap (R={R:g} m, z=1 m)"
    from lsc_planner_amd.mission import Mission
    cell = L.circle_swap(per_gpu, circle_radius=R, z=1.0)
    cols = int(np.ceil(np.sqrt(G)))
    rows = (G + cols - 1) // cols
    pitch = 2.0 * R + 14.0
    off = np.array([[(c % cols) * pitch, (c // cols) * pitch, 0.0] for c in range(G)], np.float32)
    start = np.concatenate([cell.start + off[c] for c in range(G)])
    goal = np.concatenate([cell.goal + off[c] for c in range(G)])
    ms = Mission(start, goal, np.array([-R - 2, -R - 2, 0], np.float32),
                 np.array([(cols - 1) * pitch + R + 2, (rows - 1) * pitch + R + 2, 2.5], np.float32),
                 np.tile(cell.radius, G), np.tile(cell.downwash, G), np.tile(cell.max_vel, (G, 1)), np.tile(cell.max_acc, (G, 1)),
                 np.tile(cell.nominal_velocity, G), name=f"circle_swap{per_gpu}x{G}")
    return ms, (f"{G} x {per_gpu}-agent generated circle swaps (R={R:g} m, z=1 m, {pitch:g} m pitch) in one world = "
                f"{n_agents} agents, one circle per GPU")


def forest256_mission(L):
    """BASELINE configs[3] as SURVEY 8(d)#4 writes it.  The occupancy is the committed leaf fixture of the reference's data
    file world/simple_forest.bt (lsc_planner_amd/data/simple_forest_leaves.npz), written out as a .bt and read back by the product's
    own reader."""
    import tempfile
    from lsc_planner_amd.maps import forest_leaves, write_bt
    leaves, res = forest_leaves()
    bt = os.path.join(tempfile.mkdtemp(prefix="lsc_forest_"), "simple_forest.bt")
    write_bt(bt, leaves, res)
    world = (-5.0, -5.0, 0.0, 5.0, 5.0, 2.5)
    dist, kmin, r = L.edt_from_bt(bt, np.asarray(world[:3], np.float32), np.asarray(world[3:], np.float32))
    return L.random_swarm(256, world=world, seed=20260928, edt=dist, edt_key_min=kmin, edt_res=r), bt


def algorithmic_flops(n_agents, iters_total):
    """per interior-point iteration per agent ~ (N-1)*1.0 kflop + 0.3 Mflop (fp64)."""
    return float(iters_total) * ((n_agents - 1) * 1.0e3 + 0.3e6)


# The active-set solve does other -- and much less -- arithmetic than the interior point SURVEY 8(d) prices, so it gets a model of its own
# (DESIGN section 4.1): per change of the working set one pass over the rows (violation of every row: 3 multiply-adds, the scale,
# the compare ~ 8 flop per row; the 414 bound / velocity / acceleration rows are always carried) plus ~2.6 kflop of small dense algebra
# (the row's normal and H^-1 times it from the per-variable tables: 2 x 39 x 9; G_W H^-1 n: 12 x 21; the multiplier rates from the kept
# inverse of the Gram matrix and its update: 2 x 12 x 12 x 2; the direction: 39 x 12 x 2; x from y: 90 x 6 -- the loops run over all 12
# working-set slots, ~3.2 kflop as executed, ~1.9 kflop at the typical four rows); per solve the start (gradient, unconstrained optimum:
# ~2.5 kflop) and two more passes over the rows (the last search, the verification).
GI_ROW_FLOP, GI_AXIS_ROWS, GI_DENSE_PER_CHANGE, GI_PER_SOLVE = 8.0, 414.0, 2.6e3, 2.5e3


def active_set_flops(n_agents, stats, row_changes_total, rows_mean, all_rows):
    """fp64 flops of the timed ticks under the model above.  row_changes_total = sum over agents of changes x LSC rows carried (device
    accumulator), or changes x 27 (N - 1) when all_rows (the reference's row count); handed-over agents add SURVEY's interior-point model."""
    solves = float(stats["solved"] + stats["handed_over"])
    changes = float(stats["changes"])
    rows = 27.0 * (n_agents - 1) if all_rows else rows_mean
    rc = changes * 27.0 * (n_agents - 1) if all_rows else float(row_changes_total)
    f = GI_ROW_FLOP * (rc + changes * GI_AXIS_ROWS) + changes * GI_DENSE_PER_CHANGE
    f += solves * (GI_PER_SOLVE + 2.0 * GI_ROW_FLOP * (rows + GI_AXIS_ROWS))
    f += float(stats["ip_iterations"]) * ((n_agents - 1) * 1.0e3 + 0.3e6)
    return f


def cpu_baseline(ms, seconds_target=12.0, max_ticks=120, static_goal=False):
    """Oracle (CPU restatement of the reference path) on the same mission from its start, sequential over agents
    like the reference, then once more with OpenMP over agents on all cores."""
    from oracle import oracle as O
    from lsc_planner_amd.planner import next_state_host
    N = ms.qn
    out = {}
    # what is timed is the restatement of the REFERENCE's path (its interior point standing in for CPLEX); the exact finish the oracle adds
    # for the parity tests (orc_gi_polish, +20 % of a solve) is test infrastructure and is switched off here
    os.environ["ORC_NO_POLISH"] = "1"
    for label, threads in (("seq", 1), ("omp", min(os.cpu_count() or 1, N, 32))):
        prm = O.make_params(world_min=ms.world_min, world_max=ms.world_max, obs_f32=True)
        sw = O.Swarm(prm, ms.radius, ms.downwash, ms.max_vel, ms.max_acc, ms.nominal_velocity)
        state = np.zeros((N, 9), np.float32)
        state[:, :3] = ms.start
        traj = np.zeros((N, 3, 30), np.float32)
        t0 = time.perf_counter()
        ticks = 0
        budget = seconds_target if threads == 1 else seconds_target / 3
        while ticks < max_ticks and (time.perf_counter() - t0) < budget:
            goal = ms.goal if static_goal else O.goal_prior_based(state, ms.goal, traj, ticks + 1)
            r = sw.tick(state, goal, traj, ticks + 1, nthreads=threads)
            traj = r["traj"]
            state = next_state_host(traj)
            ticks += 1
        dt = time.perf_counter() - t0
        out[label] = (N * ticks / dt, ticks, dt, threads)
    os.environ.pop("ORC_NO_POLISH", None)
    seq, omp = out["seq"], out["omp"]
    return {
        "value": round(seq[0], 2), "unit": "agent-replans/s", "cores": 1, "kind": "port",
        "sample": f"first {seq[1]} ticks of the same {N}-agent mission ({seq[2]:.1f} s), sequential over agents like "
                  f"MultiSyncSimulator::plan; CPLEX absent -> oracle's exact fp64 QP solve",
        "all_cores_value": round(omp[0], 2), "all_cores": omp[3],
        "reference_published": "9.47 ms per agent-plan = ~106 replans/s (16 agents, forest map, hardware unknown, "
                               "log/summary_LSC_16agents.csv:2)",
    }


class MissionRun:
    """One mission flown device-resident on a stream of its own: a context (lsc_ctx), its buffers, one fused launch pair per tick."""

    def __init__(self, L, torch, ms, cfg, dev, stream):
        self.torch, self.ms = torch, ms
        self.pl = L.SwarmPlanner(ms, cfg)
        n = ms.qn
        f32 = dict(dtype=torch.float32, device=dev)
        s0 = torch.zeros((n, 9), **f32)
        s0[:, :3] = torch.from_numpy(ms.start).to(dev)
        self.states = [s0, torch.zeros_like(s0)]
        self.goal = torch.from_numpy(ms.goal).to(dev).contiguous()
        self.prev, self.nxt = torch.zeros((n, 90), **f32), torch.zeros((n, 90), **f32)
        self.cost = torch.zeros(n, dtype=torch.float64, device=dev)
        self.status = torch.zeros(n, dtype=torch.int32, device=dev)
        self.iters = torch.zeros(n, dtype=torch.int32, device=dev)
        self.stream = stream.cuda_stream
        self.seq = 0

    def tick(self):
        self.seq += 1
        self.pl.tick_device_fused(self.states[0], self.goal, self.prev, self.nxt, self.states[1], self.cost, self.status, self.iters, self.seq,
                                  self.stream)
        self.states.reverse()
        self.prev, self.nxt = self.nxt, self.prev

    def close(self):
        self.pl.close()


def rotated_mission(L, ms, angle, name):
    """The same swarm turned about the vertical axis through the world's centre: a congruent mission with other numbers."""
    from lsc_planner_amd.mission import Mission
    c, s = np.cos(angle), np.sin(angle)
    ctr = 0.5 * (np.asarray(ms.world_min, np.float64) + np.asarray(ms.world_max, np.float64))

    def rot(p):
        q = np.asarray(p, np.float64) - ctr
        out = q.copy()
        out[:, 0] = c * q[:, 0] - s * q[:, 1]
        out[:, 1] = s * q[:, 0] + c * q[:, 1]
        return (out + ctr).astype(np.float32)
    return Mission(rot(ms.start), rot(ms.goal), ms.world_min, ms.world_max, ms.radius, ms.downwash, ms.max_vel, ms.max_acc, ms.nominal_velocity,
                   name=name)


def tick_runs(L, runs, batch):
    """One tick of several independent missions: one launch for all of them (lsc_tick_device_fused_batch) or one launch each."""
    if not batch:
        for r in runs:
            r.tick()
        return
    for r in runs:
        r.seq += 1
    L.tick_device_fused_batch([r.pl for r in runs], [r.states[0] for r in runs], [r.goal for r in runs], [r.prev for r in runs],
                              [r.nxt for r in runs], [r.states[1] for r in runs], [r.cost for r in runs], [r.status for r in runs],
                              [r.iters for r in runs], [r.seq for r in runs], runs[0].stream)
    for r in runs:
        r.states.reverse()
        r.prev, r.nxt = r.nxt, r.prev


def mission_list_missions(L, ms, ids, total):
    """Mission j of the list: the headline swarm turned about the vertical axis (j = 0: the headline swarm itself); congruent missions with
    other numbers, the same family concurrent_missions flies on one GPU."""
    return [ms if j == 0 else rotated_mission(L, ms, 2.0 * np.pi * (j / (7.0 * total) + 0.013 * j), f"{getattr(ms, 'name', 'mission')}_rot{j}") for j in ids]


def mission_list_leg(L, torch, dist, ms, cfg_of, dev, K, start_tick, steps, G, rank, local_rank):
    """The headline configuration on G GPUs WITHOUT a collective: the reference's outer loop is a mission list
    (src/multi_sync_simulator_node.cpp:43-70, src/param.cpp:106-122; testall_*.launch: 30 missions per swarm size), independent missions
    share nothing, so rank r flies missions [r K, (r + 1) K) of a list of G K -- K in flight together, ONE launch per tick
    (lsc_tick_device_fused_batch).  Timed like the headline: fast-forward to start_tick untimed, barrier + synchronize, `steps` ticks,
    barrier + synchronize, the slowest rank's time.  Returns (summary on rank 0 / None, this rank's device-side numbers)."""
    from lsc_planner_amd.sharded import mission_list_ids, mission_list_summary
    missions = mission_list_missions(L, ms, mission_list_ids(G, rank, K), G * K)
    runs = [MissionRun(L, torch, m, cfg_of(), dev, torch.cuda.current_stream()) for m in missions]

    def sync():
        if G > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(start_tick - 1):
        tick_runs(L, runs, True)
    sync()
    for r in runs:
        r.pl.iterations_total(reset=True)
    runs[0].pl.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        tick_runs(L, runs, True)
    sync()
    elapsed = time.perf_counter() - t0
    kb = runs[0].pl.kernel_times_ms(0)               # the batch launch is clocked on the first context
    runs[0].pl.set_timing(False)
    stats = {"solved": 0.0, "handed_over": 0.0, "changes": 0.0, "ip_iterations": 0.0}
    it_total = rowit_total = 0.0
    for r in runs:
        st = r.pl.solver_stats()
        for k in stats:
            stats[k] += float(st[k])
        it_total += float(r.pl.iterations_total(reset=False))
        rowit_total += float(r.pl.row_iterations_total())
    failed = sum(int((r.status != 0).sum().item()) for r in runs)
    agents = sum(m.qn for m in missions)
    props = torch.cuda.get_device_properties(dev)
    uuid = str(getattr(props, "uuid", "")) + "/" + str(getattr(props, "pci_bus_id", ""))
    summary = mission_list_summary(dist if G > 1 else None, rank, G, agents, steps, elapsed, float(np.percentile(kb, 99)) if len(kb) else 0.0,
                                   torch.cuda.current_device(), uuid if uuid != "/" else "", failed)
    mine = {"kernel_ms": kb, "stats": stats, "iters_total": it_total, "rowit_total": rowit_total, "agents": agents,
            "missions": [getattr(m, "name", "mission") for m in missions], "traj": [r.prev.clone() for r in runs]}
    for r in runs:
        r.close()
    return summary, mine


def concurrent_missions_leg(L, torch, ms, cfg_of, dev, K, start_tick, steps, n_cu):
    """The reference's outer loop -- a directory of missions flown back to back (src/multi_sync_simulator_node.cpp:43-70,
    src/param.cpp:106-122; testall_*.launch: 30 missions per swarm size) -- as a batch axis: K independent 64-agent missions, one
    context each, planned by ONE launch per tick (lsc_tick_device_fused_batch: blockIdx.y = mission).  A 64-agent tick is 64 workgroups
    on a 256-CU chip; K = 4 fills it.  Every mission is first flown ALONE over the same tick window (the yardstick), then all K together
    -- through the batch launch, and, for comparison, as K launches on K streams --; the plans of all three runs must be the same bits."""
    missions = [ms] + [rotated_mission(L, ms, 2.0 * np.pi * (m / (7.0 * K) + 0.013 * m), f"{getattr(ms, 'name', 'mission')}_rot{m}") for m in range(1, K)]

    tick_all = lambda runs, batch: tick_runs(L, runs, batch)

    def fly(runs, batch=False):
        for _ in range(start_tick - 1):
            tick_all(runs, batch)
        torch.cuda.synchronize()
        for r in runs:
            r.pl.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            tick_all(runs, batch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ks = [r.pl.kernel_times_ms(0) for r in (runs[:1] if batch else runs)]
        for r in runs:
            r.pl.set_timing(False)
        return dt, ks

    p = lambda a, q: round(float(np.percentile(a, q)), 4)
    alone, alone_k, alone_traj = [], [], []
    for m in range(K):
        r = MissionRun(L, torch, missions[m], cfg_of(), dev, torch.cuda.current_stream())
        dt, ks = fly([r])
        alone.append(missions[m].qn * steps / dt)
        alone_k.append(ks[0])
        alone_traj.append(r.prev.clone())
        r.close()
    agents = sum(mm.qn for mm in missions)
    # (a) one launch per tick for all K missions
    runs = [MissionRun(L, torch, missions[m], cfg_of(), dev, torch.cuda.current_stream()) for m in range(K)]
    dt_b, ks_b = fly(runs, batch=True)
    same_b = all(bool(torch.equal(runs[m].prev, alone_traj[m])) for m in range(K))
    for r in runs:
        r.close()
    # (b) K launches on K streams
    streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
    runs = [MissionRun(L, torch, missions[m], cfg_of(), dev, streams[m]) for m in range(K)]
    torch.cuda.synchronize()
    dt_s, ks_s = fly(runs)
    same_s = all(bool(torch.equal(runs[m].prev, alone_traj[m])) for m in range(K))
    for r in runs:
        r.close()
    value = agents * steps / dt_b
    kb = ks_b[0]
    worst_alone_p99 = max(float(np.percentile(k, 99)) for k in alone_k)
    return {
        "missions": K, "agents_per_mission": ms.qn, "contexts": K, "launches_per_tick": 1,
        "entry_point": "lsc_tick_device_fused_batch (lsc_plan_batch_kernel: blockIdx.y = mission, every mission's argument block in the kernarg segment)",
        "value": round(value, 1), "unit": "agent-replans/s (aggregate over the concurrent missions)",
        "ms_per_step_all_missions": round(1e3 * dt_b / steps, 4),
        "alone_value_mean": round(float(np.mean(alone)), 1), "alone_values": [round(v, 1) for v in alone],
        "speedup_vs_one_mission_alone": round(value / float(np.mean(alone)), 3),
        "tick_ms": {"batch_p50": p(kb, 50), "batch_p99": p(kb, 99),
                    "alone_p50": [p(k, 50) for k in alone_k], "alone_p99": [p(k, 99) for k in alone_k],
                    "p99_ratio_batch_over_slowest_mission_alone": round(float(np.percentile(kb, 99)) / worst_alone_p99, 3),
                    "note": "a batch tick ends with the slowest agent of the K missions: its p99 is compared with the largest of the K missions' own p99"},
        "workgroups_in_flight": agents, "cus": n_cu, "cus_occupied": min(agents, n_cu),
        "plans_bit_identical_to_the_missions_flown_alone": same_b,
        "tick_window": [start_tick, start_tick + steps - 1],
        "streams_variant": {"launches_per_tick": K, "value": round(agents * steps / dt_s, 1),
                            "speedup_vs_one_mission_alone": round(agents * steps / dt_s / float(np.mean(alone)), 3),
                            "per_mission_p50": [p(k, 50) for k in ks_s], "per_mission_p99": [p(k, 99) for k in ks_s],
                            "plans_bit_identical_to_the_missions_flown_alone": same_s,
                            "note": "the same K missions as K launches on K streams: independent dispatches are not placed one workgroup per CU "
                                    "(two of four missions wait for CUs each tick), and streams beyond the hardware queues serialise"},
        "note": "K independent missions (mission 0 = the headline swarm, the others the same swarm turned about the vertical axis), one context each; "
                "device time per tick = HIP events around the launch; one 512-lane workgroup with ~150 KB of LDS per agent = one workgroup per CU",
    }


def mission_list_main(args, L, torch, dist, dev, G, rank, local_rank, json_fd):
    """`bench.py --gpus N --mission-list`: the headline configuration (BASELINE configs[2]) on N GPUs as a mission list -- no collective
    on the data path.  One JSON line on rank 0 with the contract's keys, `mission_list` (per-rank values, devices, tick p99),
    `roofline` of the batch launch and -- one GPU only -- `cpu_baseline`."""
    K = max(args.missions, 1)
    ms, layout = weak_scaling_mission(L, 1, args.agents_per_gpu)
    goal_mode = "static" if args.static_goal else "prior_based"
    cfg_of = lambda: L.PlannerConfig(device=local_rank, prune=not args.no_prune, goal_mode=goal_mode, reset_threshold=args.reset_threshold,
                                     solver=args.solver)
    start_tick = max(args.start_tick if args.start_tick is not None else 60, args.warmup + 1)
    summary, mine = mission_list_leg(L, torch, dist, ms, cfg_of, dev, K, start_tick, args.steps, G, rank, local_rank)
    if rank == 0:
        kb = mine["kernel_ms"]
        st = mine["stats"]
        gi = args.solver == "active_set" and (st["solved"] + st["handed_over"]) > 0
        n = ms.qn
        rows_mean = mine["rowit_total"] / max(mine["iters_total"], 1.0)
        if gi:
            flops = active_set_flops(n, st, mine["rowit_total"], rows_mean, True) / max(len(kb), 1)
            flops_exec = active_set_flops(n, st, mine["rowit_total"], rows_mean, False) / max(len(kb), 1)
        else:
            flops = algorithmic_flops(n, mine["iters_total"]) / max(len(kb), 1)
            flops_exec = (mine["rowit_total"] * (1.0e3 / 27.0) + mine["iters_total"] * 0.3e6) / max(len(kb), 1)
        k_ms = float(kb.mean()) if len(kb) else 0.0
        ach = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        ach_exec = flops_exec / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        p = lambda a, q: round(float(np.percentile(a, q)), 4) if len(a) else None
        elapsed = summary["elapsed_s_max_over_ranks"]
        result = {
            "metric": "agent-replans/sec (whole node)", "value": round(summary["value"], 1), "unit": "agent-replans/s",
            "n_gpus": G, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "mission-list", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "tick_solve_ms": {"p50": p(kb, 50), "p99": p(kb, 99), "p99_max_over_ranks": round(summary["tick_p99_ms_max_over_ranks"], 4),
                              "note": "device time of the ONE launch per tick that plans this rank's K missions (HIP events, rank 0); a batch tick ends with "
                                      "the slowest agent of the K missions"},
            "p99_tick_ms": {"device_resident": round(summary["tick_p99_ms_max_over_ranks"], 4), "device_resident_samples": args.steps,
                            "note": "largest per-rank p99 of the batch launch over the timed steps"},
            "config": {"tick_window": [start_tick, start_tick + args.steps - 1],
                       "workload": f"mission list of {G * K} independent missions: {layout} (mission 0) and the same swarm turned about the vertical axis, empty map, "
                                   f"LSC mode, dt 0.2 s, M=5 n=5, mode/goal={goal_mode}, {K} missions in flight per GPU, ONE launch per tick and GPU "
                                   "(lsc_tick_device_fused_batch: goal planning + LSC + QP + state propagation of all K missions"
                                   + (" + the hand-over launch of the alternate-mode kernel" if args.reset_threshold > 0 else "") + "), device-resident; "
                                   "no collective: independent missions share nothing (the reference's node flies them back to back, "
                                   "src/multi_sync_simulator_node.cpp:43-70)",
                       "agents": summary["agents_in_flight"], "agents_per_mission": n, "missions_per_gpu": K, "missions": G * K,
                       "parallelism": f"mission-list x{G} (rank r flies missions [r K, (r + 1) K))", "reset_threshold": args.reset_threshold,
                       "scaling_note": "fixed work per GPU (K missions each); the single-mission headline of the plain `bench.py` line is the latency figure, "
                                       "this line the throughput one"},
            "mission_list": summary,
            "roofline": {"kernel": "lsc_plan_batch_kernel", "bound": "valu_fp64", "achieved": round(ach, 5), "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / FP64_VALU_PEAK_TFLOPS, 7), "frac_executed": round(ach_exec / FP64_VALU_PEAK_TFLOPS, 7),
                         "avg_launch_ms": round(k_ms, 5), "launches": int(len(kb)), "traffic": None,
                         "note": "rank 0's batch launch (K missions x 64 workgroups on 256 CUs), flop model of the single-mission line; latency-bound like it"},
        }
        result["cpu_baseline"] = cpu_baseline(ms, static_goal=args.static_goal) if (G == 1 and not args.no_cpu_baseline) else None
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    os.close(json_fd)
    if G > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--start-tick", type=int, default=None,
                    help="mission tick of the first TIMED step; the ticks before it run untimed (fast-forward, the last --warmup of "
                         "them are the warm-up).  Default: 60 for the circle workload -- the 64 agents meet in ticks ~40-140 of the "
                         "~226-tick mission, so any --steps starts in the crossing, where the ticks are longest -- and warmup + 1 for "
                         "the strong-scaling workloads")
    ap.add_argument("--agents-per-gpu", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency-leg", action="store_true")
    ap.add_argument("--no-ip-leg", action="store_true", help="skip the interior_point_leg (the timed ticks once more under --solver interior_point)")
    ap.add_argument("--no-prune", action="store_true")
    ap.add_argument("--static-goal", action="store_true", help="mode/goal=static instead of the reference default prior_based")
    ap.add_argument("--reset-threshold", type=float, default=0.15,
                    help="multisim/reset_threshold of testall_empty.launch: the reference's disturbance checks run every tick "
                         "(they never fire on this mission; the tick pays the scan and the hand-over launch, ~2.4 %%); 0 = off")
    ap.add_argument("--planner", default="lsc", choices=["lsc", "bvc"], help="mode/planner (bvc: the general dense kernel)")
    ap.add_argument("--slack", default="none", choices=["none", "dynamical_limit", "collision_constraint"])
    ap.add_argument("--solver", default="active_set", choices=["active_set", "interior_point"],
                    help="QP solver of the fast path (lsc_config.solver): active_set = the dual active-set solve first, the interior point as its "
                         "fallback (the library's default); interior_point = the interior point alone (rounds 1-4)")
    ap.add_argument("--unfused", action="store_true",
                    help="single GPU only: use the multi-GPU tick sequence (plan shard, exchange, propagate) instead of the fused launch")
    ap.add_argument("--single-circle", action="store_true",
                    help="several GPUs: one circle of 64 G agents (R = 8 G) instead of G circles of 64")
    ap.add_argument("--torch-exchange", action="store_true",
                    help="several GPUs (or --unfused): exchange through torch.distributed instead of the library's own communicator")
    ap.add_argument("--workload", default="circle64", choices=["circle64", "random1024", "forest256"],
                    help="circle64: BASELINE configs[2], weak scaling (64 agents per GPU); random1024: configs[4], one 1024-agent "
                         "swarm (seed 20260929) sharded over the GPUs, strong scaling; forest256: configs[3] as written (256 agents, "
                         "simple_forest, world [-5,5]^2 x [0,2.5], seed 20260928), strong scaling")
    ap.add_argument("--agents", type=int, default=0,
                    help="--workload random1024 only: this many agents instead of 1024, in a world of the same density (40 m x sqrt(N / 1024) square, "
                         "5 m high): e.g. 8192 agents, on one GPU or sharded over --gpus 8 at 1024 per rank")
    ap.add_argument("--missions", type=int, default=4,
                    help="extra leg (single GPU, circle64 workload): this many independent 64-agent missions in flight together, one context and "
                         "stream each -- the reference's mission-list outer loop as a batch axis; reported as `concurrent_missions` NEXT TO the "
                         "single-mission headline, never instead of it; 0 or 1 = skip")
    ap.add_argument("--mission-list", action="store_true",
                    help="the headline configuration on --gpus N WITHOUT a collective: every rank flies --missions K independent 64-agent "
                         "missions of a list of N K (one launch per tick, lsc_tick_device_fused_batch); value = sum over the ranks / the slowest "
                         "rank's time, \"scaling\": \"mission-list\".  Without this flag a --gpus N > 1 line still carries the same measurement as its "
                         "`mission_list` object next to the all-gather workload")
    ap.add_argument("--sweep-agents", type=int, default=1024,
                    help="extra leg: dense LSC sweep at this swarm size (HBM-meaningful working set); 0 = skip")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start the N ranks ourselves, one process per GPU (same command line the driver uses)
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner through C
    # stdio when a communicator is created), so file descriptor 1 is pointed at stderr for the whole run and the JSON line
    # goes to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if args.missions > 1:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, args.missions + 1)))   # (read when the HIP runtime starts: one hardware queue per mission stream)
    import torch
    import torch.distributed as dist
    import lsc_planner_amd as L

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank} but this node has {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    G = max(world, 1)
    if args.gpus != G:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={G}: launch with --nproc-per-node {args.gpus} "
                         "(or run `python bench.py --gpus N`, which starts the ranks itself)")
    token = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        box = [L.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)           # control plane only: the rendezvous token of the native communicator
        token = box[0]
    elif args.unfused:
        token = L.comm_unique_id()                       # world-size-1 communicator: the same code path on one GPU
    sharded = token is not None

    if args.mission_list:
        if args.workload != "circle64" or args.planner != "lsc" or args.slack != "none":
            raise SystemExit("bench.py --mission-list flies the headline configuration (circle64, LSC mode)")
        mission_list_main(args, L, torch, dist, dev, G, rank, local_rank, json_fd)
        return

    strong = args.workload != "circle64"
    bt_path = None
    if args.workload == "random1024":
        if args.agents in (0, 1024):
            ms = L.random_swarm(1024, seed=20260929)
            layout = "1024-agent random swarm (world [-20,20]^2 x [0,5], seed 20260929: BASELINE configs[4])"
        else:
            # configs[4]'s density with another number of agents (round 6: swarms beyond one GPU's 1024 agents; tools/shard_emulation.py --agents)
            half = 20.0 * (args.agents / 1024.0) ** 0.5
            ms = L.random_swarm(args.agents, world=(-half, -half, 0, half, half, 5), seed=20260929)
            layout = f"{args.agents}-agent random swarm of BASELINE configs[4]'s density (world [-{half:.1f},{half:.1f}]^2 x [0,5], seed 20260929)"
    elif args.workload == "forest256":
        ms, bt_path = forest256_mission(L)
        layout = ("256-agent random swarm in world/simple_forest.bt (world [-5,5]^2 x [0,2.5], seed 20260928: BASELINE configs[3] as "
                  "written), EDT + SFC + grid-search goals")
    else:
        ms, layout = weak_scaling_mission(L, G, args.agents_per_gpu, args.single_circle)
    n_agents = ms.qn
    goal_mode = "static" if args.static_goal else "prior_based"
    def make_planner(comm):
        p = L.SwarmPlanner(ms, L.PlannerConfig(device=local_rank, prune=not args.no_prune, goal_mode=goal_mode,
                                               reset_threshold=args.reset_threshold, planner_mode=args.planner,
                                               slack_mode=args.slack, use_octomap=bt_path is not None, comm=comm, solver=args.solver))
        if bt_path is not None:
            p.load_octomap(bt_path)
        return p

    # Several GPUs: the exchange is the library's own RCCL all-gather (lsc_comm_init).  Should that communicator not come
    # up on this node, the ranks agree on it and the run falls back to the torch.distributed all-gather of
    # lsc_planner_amd/sharded.py around the same shard kernels (reported as "native": false) instead of dying without a line.
    from lsc_planner_amd.sharded import shard_bounds, shard_rows as _shard_rows, table_rows as _table_rows, all_gather_rows
    native, why, pl = sharded and not args.torch_exchange, "--torch-exchange" if args.torch_exchange else None, None
    try:
        pl = make_planner((G, rank, token) if native else None)
    except L.LscError as e:
        if world == 1:
            raise
        native, why = False, str(e)
    if world > 1:
        agree = torch.tensor([1 if native else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if int(agree.item()) == 0 and not args.torch_exchange:
            native = False
            why = why or "another rank could not create the native communicator"
            if pl is not None:
                pl.close()
            pl = make_planner(None)
    if world > 1 and not native and not args.torch_exchange:
        # A multi-GPU line must say what ran: without --torch-exchange the exchange is the library's own RCCL all-gather or nothing.
        # (Round 4 fell back silently to the torch.distributed exchange and reported "native": false inside the line.)
        sys.stderr.write(f"bench.py: rank {rank}: the native RCCL communicator did not come up ({why}); refusing to fall back silently -- "
                         "pass --torch-exchange to measure the torch.distributed exchange instead\n")
        if pl is not None:
            pl.close()
        dist.barrier()
        dist.destroy_process_group()
        raise SystemExit(3)
    if sharded and native and pl.world != G:
        raise SystemExit(f"bench.py: the native communicator has {pl.world} ranks, --gpus says {G}")
    if sharded and not native:
        pl.set_shard(*shard_bounds(n_agents, G, rank))
        pl.shard_rows, pl.table_rows = _shard_rows(n_agents, G), _table_rows(n_agents, G)
    first, count, rows = pl.first, pl.count, pl.table_rows      # rank's block of the (padded) trajectory table

    f32 = dict(dtype=torch.float32, device=dev)
    state = torch.zeros((n_agents, 9), **f32)
    state[:, :3] = torch.from_numpy(ms.start).to(dev)
    goal = torch.from_numpy(ms.goal).to(dev).contiguous()
    traj_a = torch.zeros((rows, 90), **f32)
    traj_b = torch.zeros((rows, 90), **f32)
    cost = torch.zeros(n_agents, dtype=torch.float64, device=dev)
    status = torch.zeros(n_agents, dtype=torch.int32, device=dev)
    iters = torch.zeros(n_agents, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    seq = 0

    state_b = torch.zeros_like(state)
    states = [state, state_b]

    def tick(prev, nxt):
        # one GPU: a tick is ONE launch (goal planning, LSC, QP and the next ideal states fused);
        # several GPUs: plan the shard, all-gather the new trajectories, then every rank propagates all states
        nonlocal seq
        seq += 1
        if not sharded:
            pl.tick_device_fused(states[0], goal, prev, nxt, states[1], cost, status, iters, seq, stream)
            states.reverse()
        elif native:
            pl.tick_device_sharded(states[0], goal, prev, nxt, cost, status, iters, seq, stream)
        else:
            pl.tick_device(states[0], goal, prev, nxt, cost, status, iters, seq, stream)
            all_gather_rows(dist, nxt, rank, pl.shard_rows)
            pl.propagate_device(nxt, states[0], stream)

    def sync():
        if G > 1:
            dist.barrier()
        torch.cuda.synchronize()

    prev, nxt = traj_a, traj_b
    start_tick = args.start_tick if args.start_tick is not None else (60 if args.workload == "circle64" else args.warmup + 1)
    start_tick = max(start_tick, args.warmup + 1)
    pl.set_timing(True)                      # (HIP events only: the launch times of the fast-forward feed tick_solve_ms.p99_whole_mission)
    for _ in range(start_tick - 1):          # not in the measurement: fast-forward into the mission; its last --warmup ticks are the warm-up
        tick(prev, nxt)
        prev, nxt = nxt, prev
    sync()
    k_before = pl.kernel_times_ms(0) if start_tick > 1 else np.zeros(0)
    pl.iterations_total(reset=True)
    pl.set_timing(True)
    bad = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tick(prev, nxt)
        prev, nxt = nxt, prev
    sync()
    elapsed = time.perf_counter() - t0
    k_ms, k_n = pl.kernel_time_ms(0)
    k_all = pl.kernel_times_ms(0)
    x_all = pl.kernel_times_ms(2) if native else np.zeros(0)
    g_all = pl.kernel_times_ms(3) if bt_path is not None and goal_mode == "prior_based" else np.zeros(0)
    c_all = pl.kernel_times_ms(4) if bt_path is not None else np.zeros(0)
    sstats = pl.solver_stats()               # (counters of the active-set solve over the timed ticks; zeros under --solver interior_point)
    iters_total = pl.iterations_total(reset=False)
    rowit_total = pl.row_iterations_total()
    bad = int((status[first:first + count] != 0).sum().item())
    lrows = pl.row_counts()[first:first + count]
    pl.set_timing(False)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    it_t = torch.tensor([float(iters_total), float(rowit_total), float(sstats["solved"]), float(sstats["handed_over"]), float(sstats["changes"]),
                         float(sstats["ip_iterations"])], dtype=torch.float64, device=dev)
    if G > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(it_t, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    iters_total, rowit_total = float(it_t[0].item()), float(it_t[1].item())
    sstats = dict(solved=float(it_t[2].item()), handed_over=float(it_t[3].item()), changes=float(it_t[4].item()), ip_iterations=float(it_t[5].item()))
    gi = args.solver == "active_set" and (sstats["solved"] + sstats["handed_over"]) > 0
    # how long the mission is (untimed, after the measurement): ticks until every agent is within plan/goal_threshold of its goal
    mission_ticks = None
    k_mission = None
    if args.workload == "circle64" and G == 1 and args.planner == "lsc":
        gl = torch.from_numpy(ms.goal).to(dev)
        n_done = seq
        pl.set_timing(True)
        while n_done < 600 and float((states[0][:, :3] - gl).norm(dim=1).max().item()) >= 0.1:
            tick(prev, nxt)
            prev, nxt = nxt, prev
            n_done += 1
        torch.cuda.synchronize()
        k_after = pl.kernel_times_ms(0)
        pl.set_timing(False)
        mission_ticks = n_done if n_done < 600 else None
        # launch times of the WHOLE mission: fast-forward (without its first five launches: first-touch effects) + timed window + the tail
        k_mission = np.concatenate([k_before[5:], k_all, k_after])
    # per-rank device times of the tick's launches (HIP events): where a sharded tick's time goes
    mine = torch.tensor([float(k_all.mean()) if len(k_all) else 0.0, float(np.percentile(k_all, 99)) if len(k_all) else 0.0,
                         float(g_all.mean()) if len(g_all) else 0.0, float(c_all.mean()) if len(c_all) else 0.0,
                         1e3 * float(x_all.mean()) if len(x_all) else 0.0, float(count)], dtype=torch.float64, device=dev)
    per_rank = [mine.clone() for _ in range(G)]
    if G > 1:
        dist.all_gather(per_rank, mine)
    per_rank = [[round(float(v), 4) for v in r.tolist()] for r in per_rank]

    ml_summary = None
    if G > 1 and args.workload == "circle64" and args.missions > 1 and args.planner == "lsc" and args.slack == "none":
        # every rank takes part (barriers around its timed region); the headline above is unaffected
        ms1, _ = weak_scaling_mission(L, 1, args.agents_per_gpu)
        ml_cfg = lambda: L.PlannerConfig(device=local_rank, prune=not args.no_prune, goal_mode=goal_mode, reset_threshold=args.reset_threshold,
                                         solver=args.solver)
        ml_summary, _ = mission_list_leg(L, torch, dist, ms1, ml_cfg, dev, args.missions, start_tick, args.steps, G, rank, local_rank)

    result = None
    if rank == 0:
        value = n_agents * args.steps / elapsed
        if gi:
            # the active-set kernel: its own flop model (active_set_flops), with the reference's 27 (N - 1) rows (frac) and with the rows carried (frac_executed)
            rows_mean = rowit_total / max(iters_total, 1.0)
            flops = active_set_flops(n_agents, sstats, rowit_total, rows_mean, True) / G / max(k_n, 1)
            flops_exec = active_set_flops(n_agents, sstats, rowit_total, rows_mean, False) / G / max(k_n, 1)
        else:
            flops = algorithmic_flops(n_agents, iters_total / G) / max(k_n, 1)   # per launch of this rank's kernel
            # the same model charged with the rows the kernel really carried: (N-1) kflop per iteration = 27 (N-1) rows x ~37 flop
            flops_exec = (rowit_total / G * (1.0e3 / 27.0) + iters_total / G * 0.3e6) / max(k_n, 1)
        ach = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        ach_exec = flops_exec / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        kname = "lsc_plan_kernel" if (args.reset_threshold <= 0 and args.planner == "lsc" and args.slack == "none") else "lsc_plan_alt_kernel"
        if count > torch.cuda.get_device_properties(dev).multi_processor_count:
            kname = kname.replace("_kernel", "_tp_kernel")       # more agents in the shard than CUs: the throughput build"
        traffic, traffic_src = pmc_traffic(kname + "@grid32768") if n_agents == 64 else (None, None)
        result = {
            "metric": "agent-replans/sec (whole node)", "value": round(value, 1), "unit": "agent-replans/s",
            "n_gpus": G, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "tick_solve_ms": {"p50": round(float(np.percentile(k_all, 50)), 4), "p99": round(float(np.percentile(k_all, 99)), 4),
                              "max": round(float(k_all.max()), 4),
                              "p99_whole_mission": round(float(np.percentile(k_mission, 99)), 4) if k_mission is not None and len(k_mission) else None,
                              "p50_whole_mission": round(float(np.percentile(k_mission, 50)), 4) if k_mission is not None and len(k_mission) else None,
                              "whole_mission_samples": int(len(k_mission)) if k_mission is not None else 0,
                              "mean_by_20_ticks": [round(float(k_all[i:i + 20].mean()), 4) for i in range(0, len(k_all), 20)],
                              "note": "device time of the per-tick launch (HIP events, rank 0) over the timed steps; mean_by_20_ticks: "
                                      "consecutive blocks of the timed window (how much the answer depends on where 20 steps land); "
                                      "p99_whole_mission: the same HIP events over every tick of the mission from tick 6 to the last agent's arrival -- the "
                                      "fast-forward, the timed window and the tail the bench flies anyway -- so that a 20-step run still carries a >= 200-sample p99"},
            "config": {"tick_window": [start_tick, start_tick + args.steps - 1], "mission_ticks": mission_ticks,
                       "workload": f"{layout}, " + ("" if bt_path is not None else "empty map, ") + "LSC mode, "
                                   f"dt 0.2 s, M=5 n=5, mode/goal={goal_mode}, "
                                   + (f"one swarm sharded over {G} GPU(s), {-(-n_agents // G)} agents per rank, " if strong else f"{args.agents_per_gpu} agents per GPU, ")
                                   + "device-resident ticks ("
                                   + ("plan kernel -> in-place RCCL all-gather of the new trajectories -> state propagation, "
                                      "one stream, per tick)" if sharded else
                                      ("goal search, corridor and plan launches per tick, states propagated in the plan launch)" if bt_path is not None else
                                       "one fused launch per tick: goal planning + LSC + QP + state propagation"
                                       + (" + the hand-over launch of the alternate-mode kernel (empty unless an agent is disturbed)" if args.reset_threshold > 0 else "") + ")")),
                       "agents": n_agents, "parallelism": f"agent-shard x{G}", "prune_redundant_rows": not args.no_prune,
                       "planner_mode": args.planner, "slack_mode": args.slack, "reset_threshold": args.reset_threshold,
                       "scaling_note": ("one swarm: every agent's rows against agents of other ranks are live" if strong else
                                        ("weak scaling over disjoint circles: the rows between agents of different ranks are all redundant and culled, so this "
                                         "curve is a best case (what it shows is the all-gather); --workload random1024 / forest256 shard ONE swarm" if G > 1 else
                                         "single GPU"))},
            "qp": {"solver": ("active set (Goldfarb-Idnani on the 39-unknown reduced problem from the unconstrained optimum), interior point as fallback" if gi
                              else "interior point (reduced-space Mehrotra predictor-corrector, warm-started)"),
                   "mean_working_set_changes": round(sstats["changes"] / max(sstats["solved"] + sstats["handed_over"], 1.0), 2) if gi else None,
                   "handed_to_the_interior_point": int(sstats["handed_over"]) if gi else None,
                   "mean_ip_iterations": round((sstats["ip_iterations"] / max(sstats["handed_over"], 1.0)) if gi else iters_total / (n_agents * args.steps), 2),
                   "failed_agents_last_tick": bad,
                   "active_lsc_rows_last_tick_mean": float(np.mean(lrows)), "active_lsc_rows_last_tick_max": int(np.max(lrows)),
                   "reference_rows_per_agent": 27 * (n_agents - 1)},
            "roofline": {"kernel": kname, "bound": "valu_fp64", "achieved": round(ach_exec if n_agents > 256 else ach, 5),
                         "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         # N > 256: the reference's 27 (N - 1) rows per agent are never touched (~70 of 27 621 survive the culls at N = 1024), so the
                         # line leads with the rows the kernel carried and keeps the reference-row figure under its own name
                         "frac": round((ach_exec if n_agents > 256 else ach) / FP64_VALU_PEAK_TFLOPS, 7),
                         "frac_basis": "rows carried (frac_executed)" if n_agents > 256 else "the reference's 27 (N - 1) rows per agent",
                         "frac_reference_rows": round(ach / FP64_VALU_PEAK_TFLOPS, 7),
                         "frac_executed": round(ach_exec / FP64_VALU_PEAK_TFLOPS, 7),
                         "executed_rows_per_iteration_mean": round(rowit_total / max(iters_total, 1.0), 1),
                         "traffic": traffic,
                         "traffic_source": (f"not measured in this run: rocprofv3 --pmc passes of this command, {traffic_src}"
                                            if traffic is not None else None),
                         "avg_launch_ms": round(k_ms, 5), "launches": k_n,
                         "executed_rows_mean": float(np.mean(lrows)),
                         "utilisation": pmc_utilisation(kname + "@grid32768") if n_agents == 64 else (pmc_utilisation(kname + "@grid262144#random1024") if n_agents == 1024 else None),
                         "flop_model": ("active-set solve: changes x (8 flop x (rows + 414) + 2.6 kflop) + solves x (2.5 kflop + two more row passes) [+ SURVEY's "
                                        "interior-point model for handed-over agents]; frac with the reference's 27 (N-1) rows, frac_executed with the rows carried" if gi else
                                        "SURVEY 8(d): IP iterations x ((N-1) kflop + 0.3 Mflop); frac_executed with the rows carried (~37 flop per row and iteration)"),
                         "note": ("latency-bound: one 512-lane workgroup per agent, a tick ends with its slowest agent. The active-set solve needs ~20x fewer flops than "
                                  "the interior point SURVEY 8(d) prices (a handful of rank-one steps instead of ~7 factorisations of a 39 x 39 system with "
                                  "all rows reduced into it), so its fraction of the fp64 peak is LOWER than rounds 1-4's 1 % while the tick is 2.5-3x shorter: "
                                  "the fraction measures arithmetic density, and this kernel's time is dependent chains and barriers, not arithmetic; "
                                  "interior_point_leg prices the same ticks under the old solver and SURVEY's model for continuity; "
                                  "neither HBM nor MFMA bounds this kernel" if gi else
                                  "latency-bound: one 512-lane workgroup per agent; algorithmic flops = IP iterations x "
                                  "((N-1) kflop + 0.3 Mflop) per SURVEY 8(d), which charges all 27(N-1) LSC rows; frac_executed "
                                  "charges the rows the kernel carried after pruning the provably redundant ones (iterations x rows "
                                  "accumulated on the device over the timed ticks, ~37 flop per row and iteration + 0.3 Mflop); "
                                  "neither HBM nor MFMA bounds this kernel")},
        }
        result["per_rank"] = {"columns": ["plan_kernel_ms_mean", "plan_kernel_ms_p99", "goal_kernel_ms_mean", "corridor_kernel_ms_mean",
                                          "allgather_us_mean", "agents"], "ranks": per_rank,
                              "note": "device time per tick of each rank's launches (HIP events on the tick's stream); the tick of a sharded swarm "
                                      "ends with its slowest rank plus the all-gather"}
        if ml_summary is not None:
            ml_summary["value"] = round(ml_summary["value"], 1)
            ml_summary["note"] = (f"the SAME headline configuration without a collective: every rank flies {args.missions} independent {args.agents_per_gpu}-agent missions "
                                  "of one mission list, one launch per tick (bench.py --mission-list makes this the headline of the line); value = agent-replans "
                                  "of all ranks / the slowest rank's time over the same tick window")
            result["mission_list"] = ml_summary
        if sharded:
            result["rccl"] = {"world_size": G, "native": native,
                              "collective": "ncclAllGather, in place, on the tick's stream" if native else
                                            f"torch.distributed all_gather_into_tensor (fallback: {why})",
                              "bytes_per_rank_per_tick": pl.shard_rows * 360,
                              "exchange_us_per_tick": {"mean": round(1e3 * float(x_all.mean()), 2) if len(x_all) else None,
                                                       "p99": round(1e3 * float(np.percentile(x_all, 99)), 2) if len(x_all) else None},
                              "note": "HIP events around the all-gather on rank 0 (includes waiting for the slowest rank's plan kernel)"}

    # ---- the goal planner's grid search in the octomap workload: what bounds it, measured over a few more (untimed) ticks.  A launch
    # lasts as long as its longest search (one wave per agent), so shader cycles per expanded node = launch time x clock / the
    # largest expansion count of the tick.
    if rank == 0 and bt_path is not None and goal_mode == "prior_based":
        pl.set_timing(True)
        exp_max, exp_sum = [], []
        for _ in range(10):
            tick(prev, nxt)
            prev, nxt = nxt, prev
            torch.cuda.synchronize()
            e = pl.goal_trace()["expansions"]
            exp_max.append(int(e.max())); exp_sum.append(int(e.sum()))
        gk = pl.kernel_times_ms(3)
        pl.set_timing(False)
        if len(gk) == len(exp_max) and len(gk):
            clock_mhz = 2400.0
            cyc = [1e3 * t * clock_mhz / max(m, 1) for t, m in zip(gk, exp_max)]
            result["roofline_goal"] = {
                "kernel": "lsc_goal_kernel", "bound": "latency (instruction issue of one wave per search; neither HBM nor MFMA: the search state is in LDS and registers)",
                "avg_launch_ms": round(float(np.mean(gk)), 4), "share_of_tick": round(float(np.mean(g_all) / (1e3 * elapsed / args.steps)), 3) if len(g_all) else None,
                "longest_search_nodes_mean": round(float(np.mean(exp_max)), 1), "nodes_per_tick_all_agents_mean": round(float(np.mean(exp_sum)), 1),
                "achieved": round(float(np.mean(exp_max) / (np.mean(gk) * 1e-3)), 1), "unit": "expanded nodes/s of the longest search", "peak": None, "frac": None,
                "shader_cycles_per_node": round(float(np.mean(cyc)), 1),
                "note": "one wave per agent alone on its SIMD issues one instruction per 5.7-8.5 cycles (profiles/r03_microbench.log): the cost of a node is its "
                        "instruction count; ticks measured after the timed region, " + str(len(gk)) + " launches"}

    # ---- the same ticks under the interior point alone (the solver of rounds 1-4), priced with SURVEY 8(d)'s model: continuity of the roofline
    if rank == 0 and G == 1 and gi and args.workload == "circle64" and not sharded and not args.no_ip_leg:
        ipr = MissionRun(L, torch, ms, L.PlannerConfig(device=local_rank, prune=not args.no_prune, goal_mode=goal_mode, reset_threshold=args.reset_threshold,
                                                       planner_mode=args.planner, slack_mode=args.slack, solver="interior_point"), dev, torch.cuda.current_stream())
        for _ in range(start_tick - 1):
            ipr.tick()
        torch.cuda.synchronize()
        ipr.pl.iterations_total(reset=True)
        ipr.pl.set_timing(True)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            ipr.tick()
        torch.cuda.synchronize()
        dt_ip = time.perf_counter() - t1
        kip = ipr.pl.kernel_times_ms(0)
        it_ip = ipr.pl.iterations_total(reset=False)
        fl = algorithmic_flops(n_agents, it_ip) / max(len(kip), 1)
        ach_ip = fl / (float(kip.mean()) * 1e-3) / 1e12
        result["interior_point_leg"] = {"value": round(n_agents * args.steps / dt_ip, 1), "ms_per_step": round(1e3 * dt_ip / args.steps, 4),
                                        "tick_solve_ms": {"p50": round(float(np.percentile(kip, 50)), 4), "p99": round(float(np.percentile(kip, 99)), 4)},
                                        "mean_ip_iterations": round(it_ip / (n_agents * args.steps), 2),
                                        "roofline_frac_survey_model": round(ach_ip / FP64_VALU_PEAK_TFLOPS, 7),
                                        "speedup_of_the_default_solver": round(value / (n_agents * args.steps / dt_ip), 3),
                                        "note": "--solver interior_point on the same mission and tick window (its own closed loop: the plans agree within the parity tolerances, the missions drift apart)"}
        ipr.close()

    # ---- several independent missions in flight together (the idle three quarters of the chip at 64 agents)
    if rank == 0 and G == 1 and args.missions > 1 and args.workload == "circle64" and not sharded:
        def cfg_of():
            return L.PlannerConfig(device=local_rank, prune=not args.no_prune, goal_mode=goal_mode, reset_threshold=args.reset_threshold,
                                   planner_mode=args.planner, slack_mode=args.slack, solver=args.solver)
        result["concurrent_missions"] = concurrent_missions_leg(L, torch, ms, cfg_of, dev, args.missions, start_tick, args.steps,
                                                                torch.cuda.get_device_properties(dev).multi_processor_count)
        result["concurrent_missions"]["headline_value_single_mission"] = result["value"]

    # ---- dense LSC sweep kernel (the HBM-class stage of SURVEY 8(d)): N(N-1)*180 B written per launch
    nobs = n_agents - 1
    nrm = torch.empty((count, nobs, 5, 3), **f32)
    dd = torch.empty((count, nobs, 5, 6), dtype=torch.float32, device=dev)       # float32 margins (lsc_sweep_device_f32)
    for _ in range(3):
        pl.sweep_device(states[0], prev, seq + 1, nrm, dd, stream)
    torch.cuda.synchronize()
    pl.set_timing(True)
    for _ in range(20):
        pl.sweep_device(states[0], prev, seq + 1, nrm, dd, stream)
    torch.cuda.synchronize()
    s_ms, s_n = pl.kernel_time_ms(1)
    pl.set_timing(False)
    if rank == 0:
        # bytes as materialised: fp32 normal x3 + fp32 margins x6 per (pair, segment)
        wr = count * nobs * 5 * (3 * 4 + 6 * 4)
        alg = count * nobs * 180 + n_agents * 404
        result["roofline_sweep"] = {"kernel": "lsc_sweep_kernel", "bound": "launch latency at this N (cache-resident); valu_fp64 (GJK) once the grid fills the chip",
                                    "hbm_side": "achieved / peak / frac below are the HBM view north_star asks to be reported; it is not the binding roof",
                                    "achieved": round(alg / (s_ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(alg / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                                    "traffic": pmc_traffic("lsc_sweep_kernel@grid20224")[0] if n_agents == 64 else None,
                                    "avg_launch_ms": round(s_ms, 5), "bytes_written_per_launch": wr,
                                    "algorithmic_bytes_per_launch": alg,
                                    "note": "N(N-1)*180 B + N*404 B per SURVEY 8(d); at this N the working set is "
                                            "L2/Infinity-Cache resident and the launch is latency-bound"}

    # ---- the same sweep at a swarm size whose output does not fit the caches (only N = 1024 is a
    # meaningful HBM measurement): 1024-agent seeded random swarm, tick-2 inputs (shifted previous plans)
    if args.sweep_agents > 1 and rank == 0 and G == 1 and not strong:
        from lsc_planner_amd.planner import next_state_host
        n2 = args.sweep_agents
        ms2 = L.random_swarm(n2, seed=20260929) if n2 >= 512 else L.circle_swap(n2, 8.0 * n2 / 64)
        p2 = L.SwarmPlanner(ms2, L.PlannerConfig(device=local_rank))
        st2 = np.zeros((n2, 9), np.float32)
        st2[:, :3] = ms2.start
        g2 = p2.plan(st2, ms2.goal, np.zeros((n2, 3, 30), np.float32))
        tj2 = torch.from_numpy(g2["traj"].reshape(n2, 90)).to(dev)
        s2 = torch.from_numpy(next_state_host(g2["traj"])).to(dev)
        nrm2 = torch.empty((n2, n2 - 1, 5, 3), **f32)
        dd2 = torch.empty((n2, n2 - 1, 5, 6), dtype=torch.float32, device=dev)      # float32 margins: the 180 B per pair of SURVEY 8(d)
        for _ in range(2):
            p2.sweep_device(s2, tj2, 2, nrm2, dd2, stream)
        torch.cuda.synchronize()
        p2.set_timing(True)
        for _ in range(10):
            p2.sweep_device(s2, tj2, 2, nrm2, dd2, stream)
        torch.cuda.synchronize()
        ms_l, _ = p2.kernel_time_ms(1)
        p2.set_timing(False)
        alg2 = n2 * (n2 - 1) * 180 + n2 * 404
        wr2 = n2 * (n2 - 1) * 5 * 36
        sw_pmc, sw_src = pmc_kernel("lsc_sweep_kernel@grid524288") if n2 == 1024 else ({}, None)
        # VALU busy share of the SIMDs: SQ_ACTIVE_INST_VALU is in quad-cycles per wave, SQ_BUSY_CYCLES per SE (x 4 SIMDs x 256 CUs / 32 SEs);
        # the simpler, unit-free statement is the one the counters give directly: wave-cycles on VALU instructions over all wave-cycles
        valu_share = round(sw_pmc["SQ_ACTIVE_INST_VALU"] / sw_pmc["SQ_WAVE_CYCLES"], 4) if sw_pmc.get("SQ_WAVE_CYCLES") else None
        gjk_rate = n2 * (n2 - 1) * 5 / (ms_l * 1e-3)
        result["roofline_sweep_large"] = {"kernel": "lsc_sweep_kernel", "agents": n2,
                                          "bound": "valu_fp64 (GJK)",
                                          "valu_busy": {"valu_active_share_of_wave_cycles": valu_share,
                                                        "valu_instructions_per_launch": sw_pmc.get("SQ_INSTS_VALU"),
                                                        "valu_instructions_per_hull": round(sw_pmc["SQ_INSTS_VALU"] * 64 / (n2 * (n2 - 1) * 5), 1) if sw_pmc.get("SQ_INSTS_VALU") else None,
                                                        "source": sw_src,
                                                        "note": "the launch is bound by the fp64 GJK arithmetic (divergent simplex cases), not by HBM: counter traffic "
                                                                "equals the algorithmic bytes (nothing is re-read), so the HBM fraction can only rise with fewer GJK "
                                                                "instructions per hull; the sweep is a dump API, not on the tick's path"},
                                          "hbm_side": "achieved / peak / frac are the HBM view north_star asks to be reported (secondary: not the binding roof)",
                                          "achieved": round(alg2 / (ms_l * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": round(alg2 / (ms_l * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                          "traffic": pmc_traffic("lsc_sweep_kernel@grid524288")[0] if n2 == 1024 else None,
                                          "avg_launch_ms": round(ms_l, 4), "algorithmic_bytes_per_launch": alg2,
                                          "bytes_written_per_launch": wr2,
                                          "written_GBps": round(wr2 / (ms_l * 1e-3) / 1e9, 2),
                                          "gjk_per_s": round(gjk_rate, 0)}
        p2.close()
        del nrm2, dd2

    # ---- per-tick latency through the host-buffer ABI (H2D + kernel + D2H, PCIe-inclusive): p50 / p99
    if not args.no_latency_leg and rank == 0 and G == 1 and not strong:
        from lsc_planner_amd.planner import next_state_host
        pl2 = L.SwarmPlanner(ms, L.PlannerConfig(device=local_rank, prune=not args.no_prune, goal_mode=goal_mode, solver=args.solver))
        st = np.zeros((n_agents, 9), np.float32)
        st[:, :3] = ms.start
        tj = np.zeros((n_agents, 3, 30), np.float32)
        lat = []
        pl2.set_timing(True)
        for i in range(120):
            t1 = time.perf_counter()
            r = pl2.plan(st, ms.goal, tj)
            lat.append(time.perf_counter() - t1)
            tj = r["traj"]
            st = next_state_host(tj)
        lat = np.asarray(lat[10:]) * 1e3
        abi = pl2.kernel_times_ms(5)[10:]                      # the same calls, clocked inside the library (no Python wrapper)
        result["latency_host_abi_ms"] = {"p50": round(float(np.percentile(abi, 50)), 4), "p99": round(float(np.percentile(abi, 99)), 4),
                                         "ticks": len(abi), "agent_replans_per_s": round(n_agents / (np.median(abi) * 1e-3), 1),
                                         "through_python_wrapper": {"p50": round(float(np.percentile(lat, 50)), 4),
                                                                    "p99": round(float(np.percentile(lat, 99)), 4)},
                                         "note": "lsc_replan_tick, entry to return, clocked inside the C ABI: host buffers in/out, "
                                                 "PCIe-inclusive, synchronous (through_python_wrapper adds the ctypes harness)"}
        # SURVEY 8(d) defines the per-tick solve time PCIe-inclusive: promote it next to the device-resident numbers
        ts = result["tick_solve_ms"]
        whole = ts["p99_whole_mission"] is not None and ts["whole_mission_samples"] >= 200
        result["p99_tick_ms"] = {"host_abi_pcie_inclusive": result["latency_host_abi_ms"]["p99"],
                                 "device_resident": ts["p99_whole_mission"] if whole else ts["p99"],
                                 "device_resident_samples": ts["whole_mission_samples"] if whole else args.steps,
                                 "p99_timed_window": ts["p99"],
                                 "note": "device_resident: HIP events around the per-tick launch over EVERY tick of the mission when that gives >= 200 samples "
                                         "(a 20-step window's p99 is its maximum); p99_timed_window: the same over the timed steps only"}
        pl2.close()

    if rank == 0 and G == 1 and not args.no_cpu_baseline and not strong:
        result["cpu_baseline"] = cpu_baseline(ms, static_goal=args.static_goal)
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    os.close(json_fd)
    pl.close()
    if G > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
