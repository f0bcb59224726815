"""Hand-overs of the active-set solve on BASELINE configs[3] (256 agents in the forest, static and grid goals): lsc_solver_stats, status
histogram and plan-kernel time over 110 ticks.  Needs a GPU; nothing here touches oracle/ or /root/reference."""
import sys, json, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
import lsc_planner_amd as L
from lsc_planner_amd.planner import PlannerConfig
import bench
ms, bt = bench.forest256_mission(L)
for mode in ("static", "prior_based"):
    pl = L.SwarmPlanner(ms, PlannerConfig(goal_mode=mode, use_octomap=True, reset_threshold=0.15))
    pl.load_octomap(bt)
    dev = torch.device("cuda", 0); N = ms.qn
    f32 = dict(dtype=torch.float32, device=dev)
    states = [torch.zeros((N, 9), **f32), torch.zeros((N, 9), **f32)]
    states[0][:, :3] = torch.from_numpy(ms.start).to(dev)
    goal = torch.from_numpy(ms.goal).to(dev).contiguous()
    trajs = [torch.zeros((N, 90), **f32), torch.zeros((N, 90), **f32)]
    cost = torch.zeros(N, dtype=torch.float64, device=dev); status = torch.zeros(N, dtype=torch.int32, device=dev); iters = torch.zeros(N, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    pl.iterations_total(reset=True); pl.set_timing(True)
    hist = {}
    for seq in range(1, 111):
        pl.tick_device_fused(states[0], goal, trajs[0], trajs[1], states[1], cost, status, iters, seq, stream)
        states.reverse(); trajs.reverse()
        st = status.cpu().numpy()
        for v in np.unique(st): hist[int(v)] = hist.get(int(v), 0) + int((st == v).sum())
    torch.cuda.synchronize()
    k = pl.kernel_times_ms(0)
    print(mode, json.dumps(pl.solver_stats()), "status histogram", hist, "plan kernel ms mean", round(float(k.mean()), 4), "p50", round(float(np.percentile(k, 50)), 4), "max iters last tick", int(iters.max().item()))
    pl.close()
