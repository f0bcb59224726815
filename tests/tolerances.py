"""The one table of parity tolerances.  Integer / byte / index work -- LSC normals and margins, corridor
boxes, distance fields, grid paths, planned goals, propagated states, statuses -- is compared bit for bit and has no entry
here.  The QP solve is floating point: north_star allows 1e-3 relative cost; the tests hold the kernel to

    cost          |gpu - oracle| <= COST_RTOL |oracle| + COST_ATOL
    control point |gpu - oracle| <= TRAJ_ATOL   (metres; float32 storage + flat directions of the cost; measured <= 1.5e-5)

Exceptions, each used by name where it applies:
"""
import os

COST_RTOL = 1e-6
COST_ATOL = 1e-8
TRAJ_ATOL = 2e-5
# LSC_SOLVER=interior_point (tests/test_gpu_round5.py runs the parity files once more through the interior point alone): that solver stops
# at a duality gap of 1e-9 (1 + |f|), which along flat directions of the cost leaves a plan up to ~1e-4 m from the optimum the oracle now
# returns exactly (measured 8.9e-5 m at tick 28 of the 20-agent circle, HiGHS arbitrating); rounds 1-4 compared two interior points, whose
# errors were similar.  The cost tolerance does not move.
if os.environ.get("LSC_SOLVER") == "interior_point":
    TRAJ_ATOL = 1e-4

# Seeded fuzzing of tiny swarms with extreme parameters (vmax 0.2..3, amax 0.5..6, radii 0.05..0.4, goals outside the world,
# coincident agents): optima with nearly flat directions -- the plan may move 3-4e-5 m at 1e-9 relative cost.
FUZZ_TRAJ_ATOL = 5e-5
# The same with dt = 0.5 (the M = 4 build): 2e-4 in round 4, for ONE genuinely flat optimum on which the oracle's and the kernel's interior
# points stopped on opposite sides of HiGHS's plan (tests/golden/fuzz_found_planar_m4_7500378.npz).  Since round 5 the oracle finishes its
# optimum exactly (oracle/lsc_oracle.c: orc_gi_polish) and the product's default solver is exact too: the exception is gone.
#   What is left at dt = 0.5 is the interior point's own tolerance: with half-second segments working sets beyond the active-set solve's
#   capacity (12 rows) are common, those agents are handed to the interior point, and an interior point that stops at a gap of 1e-9 (1 + |f|)
#   is up to ~1e-4 m off along flat directions of the cost (the TRAJ_ATOL of the interior-point-only runs above).  Found by the fuzzers in
#   round 5's last round: tests/golden/fuzz_found_m4_handover_8500018.npz -- a handed-over agent 5.8e-5 m from the oracle's plan at 4.2e-10
#   relative HIGHER cost, bit-identical to solver = interior_point (tests/test_gpu_round5.py).  Cost tolerance unchanged.
FUZZ_TRAJ_ATOL_HALF_SECOND = 1e-4
# ... and with the 1e5 slack penalty a grossly violated limit makes |f| ~ 1e7; both solvers stop on criteria relative to |f|
# and their plans may then differ by centimetres at 4e-8 relative cost: plans are compared below this objective only.
# (1e4 until round 4.  Slack-mode QPs can have a face of optima -- tests/golden/fuzz_found_7301082.npz: one value, no one plan -- and they
#  are the ones with a violated limit at penalty 1e5, i.e. with a large objective.  The cost comparison pins the optimum above the limit.)
FUZZ_PLAN_COMPARED_BELOW_COST = 1e3
# 1024-agent swarms near their goals: costs approach 0 (1e-6..1e-5), so the absolute floor is what binds.
LARGE_SWARM_COST_ATOL = 1e-7
# The same at the end of a mission flown to completion (tick 193 of the 64-agent bench mission: cost 2.4e-5, kernel and oracle
# 1.08e-8 apart): both solvers stop on criteria relative to 1 + |f|, i.e. absolute ones once f -> 0.
NEAR_GOAL_COST_ATOL = 1e-7
