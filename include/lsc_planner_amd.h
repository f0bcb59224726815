/*
 * lsc_planner_amd.h -- C ABI of the MI355X-native replanning path for qwerty35/lsc_planner.
 *
 * One call per replan tick for the WHOLE swarm replaces the reference's sequential
 *     for (qi...) agents[qi]->plan(sim_current_time);            src/multi_sync_simulator.cpp:320-328
 * i.e. for every agent: prediction shift -> LSC construction -> (SFC) -> QP solve
 *     TrajPlanner::planImpl / planLSC                            src/traj_planner.cpp:344-425
 *     TrajPlanner::generateLSC (GJK normals + margins)           src/traj_planner.cpp:1310-1407
 *     TrajPlanner::generateFeasibleSFC + CorridorConstructor     src/traj_planner.cpp:1451-1491,
 *                                                                include/corridor_constructor.hpp:18-245
 *     TrajOptimizer::solve / getTrajectory / getQPcost (CPLEX)   src/traj_optimizer.cpp:31-166
 * The inputs of all agents are frozen before any agent plans (multi_sync_simulator.cpp:249-304),
 * so one batched launch is semantically identical to the reference's loop.
 *
 * Conventions: plain C, caller owns every host buffer, the context owns device memory and the
 * per-agent persistent state (optimiser's last trajectory, SFC history).  Functions return 0 on
 * success or a negative LSC_E* code; nothing throws across the boundary.  A context is not
 * thread-safe (the reference's caller is single-threaded).  There is NO CPU fallback: every entry
 * point that computes fails with LSC_ENODEV when no gfx950 device is usable.
 *
 * Layouts
 *   traj   float  [N][3][M*(n+1)]   axis-major control points, index k*M*6 + m*6 + i  (n=5; M=5: k*30 + m*6 + i)
 *                                   == TrajOptimizer's variable order (src/traj_optimizer.cpp:277)
 *   state  float  [N][9]            position, velocity, acceleration (octomap::point3d = float32)
 *   goal   float  [N][3]            agent.current_goal_position (output of goal planning)
 */
#ifndef LSC_PLANNER_AMD_H
#define LSC_PLANNER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Segments of a plan: M = (int)((horizon + 1e-9) / dt), src/traj_optimizer.cpp:9.  A build parameter of the library: liblsc_hip.so
 * plans M = 5 (dt 0.2, horizon 1.0: launch/simulation.launch:60-61 and every other shipped launch file), liblsc_hip_m4.so -- the same
 * sources compiled with -DLSC_SEGMENTS=4 -- plans M = 4 (dt 0.5, horizon 2.0: the C++ defaults of src/param.cpp:66-67).  A caller
 * built against the M = 4 library defines LSC_SEGMENTS=4 before this header; lsc_segments() says what a loaded library plans, and
 * lsc_create refuses a configuration whose horizon / dt is another number. */
#ifndef LSC_SEGMENTS
#define LSC_SEGMENTS 5
#endif
#define LSC_M LSC_SEGMENTS
#define LSC_DEG 5        /* Bernstein degree n (only n=5, phi=3 exist: traj_optimizer.cpp:190-207) */
#define LSC_NC 6
#define LSC_SEGV (LSC_M * LSC_NC)
#define LSC_NV (3 * LSC_SEGV)

#define LSC_OK 0
#define LSC_EINVAL (-1)   /* bad argument                                              */
#define LSC_ENODEV (-2)   /* no usable HIP device / kernel image (no CPU fallback)     */
#define LSC_ENOMEM (-3)
#define LSC_EHIP (-4)     /* HIP runtime error, see lsc_last_error()                   */
#define LSC_ESTATE (-5)   /* call order (e.g. tick before set_agents)                  */
#define LSC_ECOMM (-6)    /* RCCL error / librccl.so not loadable, see lsc_last_error() */

/* per-agent status written by a tick (PlanningReport analogue, include/sp_const.hpp) */
#define LSC_STATUS_OK 0          /* QP solved, trajectory replaced                                       */
#define LSC_STATUS_INFEASIBLE 1  /* solver failed: optimiser's previous trajectory kept, like the reference
                                    (exception swallowed, src/traj_planner.cpp:1553-1584)                */
/* (3 was "more LSC rows than the LDS row capacity" in round 1: such agents are now solved by a second pass with their
 *  rows in HBM, so the code no longer reaches the caller)                                                  */
#define LSC_STATUS_SFC_BLOCKED 4 /* seed box of the corridor touches an obstacle (the reference throws
                                    std::invalid_argument, corridor_constructor.hpp:35-38): stale trajectory kept */
#define LSC_STATUS_GOAL_CAPACITY 5 /* the goal planner's search outgrew its LDS capacity (OPEN row / path / ray stack):
                                    nothing is guessed, the stale trajectory is kept                                */

typedef struct lsc_ctx lsc_ctx;

/* Param fields that reach the hot path (src/param.cpp:4-144, launch/testall_empty.launch:36-101) */
typedef struct {
    double dt;                 /* traj/dt                     0.2  */
    double control_weight;     /* opt/control_input_weight    0.01 */
    double terminal_weight;    /* opt/terminal_weight         1.0  */
    float  world_min[3];       /* mission world box (Mission::world_min/max, float32) */
    float  world_max[3];
    int    use_octomap;        /* world/use_octomap: adds the SFC rows                 */
    double world_resolution;   /* world/resolution (0.1)                               */
    int    device;             /* HIP device ordinal                                   */
    int    max_rows_per_cp;    /* LDS row capacity per control point of the first pass; 0 = min(N-1, 64) (lowered further if
                                  the 160 KB LDS of a workgroup demands it).  Not a limit of the problem: an agent with more
                                  rows is re-planned by the second pass with all its 27(N-1) rows in HBM */
    int    max_iters;          /* interior-point iteration cap; 0 = 50                 */
    int    prune;              /* 1 (default via lsc_default_config): drop LSC rows that are provably redundant inside
                                  the box each control point can reach under the velocity AND acceleration rows;
                                  2: the same with the velocity rows only; 0: keep all 27(N-1) rows;
                                  3: like 1 without the spatial pre-cull of far obstacles that large swarms use
                                  (identical rows and plans by construction; the switch exists for the test of that claim) */
    int    goal_mode;          /* mode/goal: 0 static (the goal input IS current_goal_position), 1 prior_based: the goal
                                  input is the desired goal and TrajPlanner::goalPlanningWithPriority runs on the
                                  device: fused into the plan kernel on maps without a distance field (where the grid
                                  search has no observable effect), a separate launch with use_octomap (priority rule,
                                  GridBasedPlanner's A* with the reference's tie-breaking, line-of-sight goal) */
    double goal_threshold;     /* plan/goal_threshold          0.1 */
    double priority_dist_threshold; /* plan/priority_dist_threshold 0.4 */
    double goal_radius;        /* plan/goal_radius             2.0 */
    double warm_start_mu;      /* interior-point start: > 0 warm start from the shifted previous plan, every row
                                  centred on this complementarity value (default 0.03), with the cold (Mehrotra)
                                  start as fallback; 0 = always cold start */
    double grid_resolution;    /* grid/resolution 0.3: cell of the goal planner's search grid (goal_mode 1 + use_octomap) */
    double grid_margin;        /* grid/margin     0.2: a cell is occupied when EDT(centre) < radius + grid_margin        */
    double horizon;            /* traj/horizon 1.0: M = (int)((horizon + 1e-9) / dt) must be the library's lsc_segments() (5, or 4
                                  in liblsc_hip_m4.so); anything else is refused by lsc_create instead of silently planning
                                  a different horizon                                                                       */
    int    goal_row_cap;       /* 0 = as large as LDS allows; > 0 lowers the goal search's OPEN-row capacity (tests)       */
    /* ---- alternate planner modes.  Any of them changes the shape of the QP; such agents are solved by a
     * general dense kernel (csrc/lsc_general.hip) instead of the banded fast path -- same results, several times slower. */
    int    planner_mode;       /* mode/planner: 0 lsc (default), 1 bvc -- Buffered Voronoi Cells: TrajPlanner::generateBVC
                                  (src/traj_planner.cpp:1409-1440), prediction / initial trajectory = current position, no
                                  stop-at-horizon rows; empty maps only (generateSFC throws in BVC mode)                    */
    int    slack_mode;         /* SlackMode: 0 none (default), 1 dynamical_limit, 2 collision_constraint
                                  (src/traj_optimizer.cpp:306-326, 375-390, 455-457, 476-510)                               */
    double slack_collision_weight; /* opt/slack_collision_weight, 100000 in every launch file                               */
    int    n_constraint_segments;  /* opt/N_constraint_segments; -1 = all M segments carry collision / corridor rows        */
    double reset_threshold;    /* multisim/reset_threshold (0.15 in the launch files).  > 0 switches the reference's disturbance
                                  checks on (obstaclePredictionCheck / initialTrajPlanningCheck, src/traj_planner.cpp:866-878,
                                  1047-1061): an agent found farther than this from where its plan puts it has its prediction
                                  reset to its position, and -- for the rest of the mission, the reference never clears
                                  obs_slack_indices -- every row against it carries a slack variable.  0 (default) = off:
                                  identical results as long as the caller's states follow the plans, and no extra launch on
                                  the device-resident ticks                                                                 */
    double gap_tolerance;      /* interior point: duality gap <= gap_tolerance (1 + |objective|) at the optimum; 0 = 1e-9
                                  (CPLEX's barrier default, CPX_PARAM_BAREPCOMP, is 1e-8)                                   */
    int    world_dimension;    /* world/dimension (src/param.cpp:12; 3).  2 = planar world: the goal planner's grid is the single
                                  layer z = world_z_2d (src/grid_based_planner.cpp:82-85, 127-133, 199-215) and the QP has the
                                  x and y variables only -- dim = param.world_dimension, src/traj_optimizer.cpp:8: 60 variables
                                  (:264-266), cost, equalities, velocity / acceleration rows over k < dim (:330, 394, 469), no z
                                  term in the terminal cost, the corridor rows (Box::convertToLSCs(dim): 4 half-spaces,
                                  src/collision_constraints.cpp:37-59) and the collision rows (:367, 423, 450) -- while every
                                  planned control point gets z = world_z_2d (:87-90).  The whole swarm must sit in that plane
                                  (states and previous plans at z = (float)world_z_2d, which is what the reference's simulator
                                  produces: src/mission.cpp:88-112, src/traj_planner.cpp:304-314); the host-buffer ticks return
                                  LSC_EINVAL otherwise.  An agent whose solve fails keeps the optimiser's previous trajectory, which
                                  starts as zeros in the reference (src/traj_optimizer.cpp:16-19) and has its own z overridden on the
                                  next state callback; here that stale plan's z block starts at world_z_2d, so a failed first solve
                                  leaves the swarm in the plane.  0 is read as 3                                              */
    double world_z_2d;         /* world/z_2d (src/param.cpp:15; 1.0)                                                        */
    int    goal_search;        /* goal planner's grid search: 0 (default) the register-resident search whenever the grid admits it
                                  (at most 128 rows, (j, z) of a cell in 17 bits), with 32-bit search keys when their table fits
                                  LDS (else the double itself as key); 1 always the general search with the row bookkeeping in
                                  LDS; 2 the register-resident search with 64-bit keys.  Same paths in every case (tests)      */
    int    solver;             /* QP solver of the LSC fast path (the reference: CPLEX with RootAlgorithm Dual, src/traj_optimizer.cpp:42-56).
                                  1 (default via lsc_default_config): a dual active-set solve first -- Goldfarb-Idnani on the 39-unknown
                                  reduced problem from the unconstrained optimum; measured: at most 9 of the ~2 000 rows are active at an
                                  optimum and the slowest agent of a tick needs ~8-13 changes of the working set -- and the interior point
                                  only when that gives up (more than 12 active rows, dependent rows, an infeasible QP: the interior point
                                  then decides the status as before).  0: the interior point alone (rounds 1-4).  Same optimum within the
                                  parity tolerances either way; the second pass (rows in HBM) keeps solver 0.  2: a test mode -- the
                                  active-set solve runs and then hands EVERY agent to the interior point (exercises the hand-over path,
                                  where everything only the interior point needs is set up); results = the interior point's           */
} lsc_config;

void lsc_default_config(lsc_config *cfg);

/* TrajPlanner ctor x N + TrajOptimizer ctor (src/traj_planner.cpp:4-73, src/traj_optimizer.cpp:4-25) */
lsc_ctx *lsc_create(const lsc_config *cfg);
void     lsc_destroy(lsc_ctx *ctx);
const char *lsc_last_error(const lsc_ctx *ctx);
/* Informational remark of lsc_create, "" when there is none -- e.g. that a slack mode was configured with the LSC planner and
 * fixed to none like TrajPlanner::checkPlannerMode does (src/traj_planner.cpp:445-448).  Never an error. */
const char *lsc_last_note(const lsc_ctx *ctx);
/* Segments M this library was built for (see LSC_SEGMENTS above). */
int lsc_segments(void);

/* Mission::agents (src/mission.cpp:60-130): radius, downwash, max_vel[3], max_acc[3], nominal_velocity.
 * Doubles, as in struct Agent (include/sp_const.hpp:153-165); an agent seen as somebody else's
 * obstacle has its radius/downwash rounded through float32 (dynamic_msgs::Obstacle), which the
 * reference fixture log/QPmodel.lp pins.  Resets the per-agent persistent state. */
int lsc_set_agents(lsc_ctx *ctx, int N, const double *radius, const double *downwash,
                   const double *max_vel /*[N][3]*/, const double *max_acc /*[N][3]*/, const double *nominal_vel);

/* Shard of agents [first, first+count) planned by this context (agent-sharded multi-GPU); default all.
 * count may be 0 (a rank without agents: its ticks are no-ops). */
int lsc_set_shard(lsc_ctx *ctx, int first, int count);

/* TrajPlanner::setDistMap (src/traj_planner.cpp:168): dense EDT, metres, [nx][ny][nz], copied to HBM.
 * key_min = octomap key of cell (0,0,0).  Only used when use_octomap. */
int lsc_set_distmap(lsc_ctx *ctx, const float *edt, int nx, int ny, int nz, const int key_min[3], double res);

/* Map input formats: reads an octomap binary tree (.bt) and builds the distance field that
 * MultiSyncSimulator::setOctomap creates with DynamicEDTOctomap(maxdist, tree, world_min, world_max, false)
 * (src/multi_sync_simulator.cpp:153-167).  *edt is malloc'ed [dims0][dims1][dims2] float metres: release it with
 * lsc_free_host.  Pure host code (no GPU needed). */
int lsc_edt_from_bt(const char *path, const float world_min[3], const float world_max[3], double maxdist, float **edt,
                    int dims[3], int key_min[3], double *res);
void lsc_free_host(void *p);

/* One replan tick, host buffers in and out (H2D + kernels + D2H inside).
 *   planner_seq : TrajPlanner::planner_seq AFTER its increment in plan() (1 on the first tick):
 *                 < 2 selects the current-velocity prediction (src/traj_planner.cpp:830-833, 998-1000)
 *   state/goal/prev_traj : all N agents (prev_traj = every agent's getTraj() of the previous tick)
 *   out_traj/out_cost/out_status/out_iters : the shard's agents only, [count]...
 *   out_lsc_normal [count][N-1][M][3] float, out_lsc_d [count][N-1][M][n+1] double, out_sfc [count][M][6]
 *                 : optional (NULL to skip) dense constraint dumps (CollisionConstraints::getLSC/getSFC). */
int lsc_replan_tick(lsc_ctx *ctx, const float *state, const float *goal, const float *prev_traj, int planner_seq,
                    float *out_traj, double *out_cost, int *out_status, int *out_iters,
                    float *out_lsc_normal, double *out_lsc_d, float *out_sfc);

/* ---- device-resident stepping (no host round trip inside a tick) -------------------------------
 * The caller (e.g. a torch tensor) owns device buffers and the stream; pointers are device pointers.
 *   d_state [N][9], d_goal [N][3], d_traj_prev [N][3][30] : read
 *   d_traj_next [N][3][30] : the shard's rows [first,first+count) are written (full table so that an
 *                            all-gather can run in place across ranks)
 *   d_cost [N] double, d_status [N] int, d_iters [N] int : shard's entries written                */
int lsc_tick_device(lsc_ctx *ctx, const float *d_state, const float *d_goal, const float *d_traj_prev,
                    int planner_seq, float *d_traj_next, double *d_cost, int *d_status, int *d_iters,
                    void *hip_stream);

/* Same tick with MultiSyncSimulator::update()'s ideal-state step fused into the launch: additionally writes
 * d_state_next [N][9] rows of the shard = the agent's state at t = dt on its new plan (what lsc_propagate_device
 * computes).  With one GPU a whole tick is then a single kernel launch; d_state and d_state_next must be different
 * buffers (every workgroup reads all current states). */
int lsc_tick_device_fused(lsc_ctx *ctx, const float *d_state, const float *d_goal, const float *d_traj_prev,
                          int planner_seq, float *d_traj_next, float *d_state_next, double *d_cost, int *d_status,
                          int *d_iters, void *hip_stream);

/* ---- several independent swarms on one GPU: the mission list as a batch axis ------------------------------
 * The reference's node flies a directory of missions back to back (src/multi_sync_simulator_node.cpp:43-70, src/param.cpp:106-122;
 * launch/testall_*.launch: 30 missions per swarm size).  A 64-agent swarm is 64 workgroups on a 256-CU chip, so up to LSC_BATCH_MAX
 * swarms -- one context each, all on one device -- are planned by ONE launch: lsc_tick_device_fused for every ctx[i] with its own
 * buffers and its own planner_seq[i], bit-identical results.  Every array argument has n entries (arrays of device pointers, held on
 * the host).  The contexts must be of one kind: empty maps (no distance field), not sharded, at most one agent per CU each, all with or
 * all without the alternate-mode hooks (reset_threshold > 0 / BVC / slack), all planar or all 3-D; LSC_EINVAL otherwise, with the
 * reason in lsc_last_error(ctx[0]).  With lsc_set_timing the launch is timed on ctx[0]. */
#define LSC_BATCH_MAX 8
int lsc_tick_device_fused_batch(lsc_ctx *const *ctx, int n, const float *const *d_state, const float *const *d_goal,
                                const float *const *d_traj_prev, const int *planner_seq, float *const *d_traj_next,
                                float *const *d_state_next, double *const *d_cost, int *const *d_status, int *const *d_iters,
                                void *hip_stream);

/* ---- agent-sharded multi-GPU: one context (= one process, one GPU) per rank ---------------------
 * The reference's exchange point is MultiSyncSimulator::update (src/multi_sync_simulator.cpp:297-303), where every
 * agent receives every other agent's previous trajectory.  With the swarm partitioned over `world_size` ranks that
 * hand-over is ONE RCCL all-gather per tick, in place on the trajectory table (xGMI on a node).  The table is padded to
 * table_rows = shard_rows * world_size rows, shard_rows = ceil(N / world_size); rank r plans agents
 * [r*shard_rows, min((r+1)*shard_rows, N)) -- the trailing ranks of a ragged split own fewer (possibly zero) agents.
 *
 *   lsc_comm_unique_id : rank 0 creates the rendezvous token (ncclGetUniqueId) and ships it to the other ranks by any
 *                        means (a file, MPI, torch.distributed's store ...)
 *   lsc_comm_init      : collective over all ranks, between lsc_create and lsc_set_agents (which then sets the shard
 *                        and pads the context's tables); librccl.so is bound at run time (the copy already mapped into
 *                        the process, else the system's)
 *   lsc_comm_info      : world size, rank, shard_rows, table_rows                                                    */
#define LSC_COMM_ID_BYTES 128
int lsc_comm_unique_id(unsigned char id[LSC_COMM_ID_BYTES]);
int lsc_comm_init(lsc_ctx *ctx, int world_size, int rank, const unsigned char id[LSC_COMM_ID_BYTES]);
int lsc_comm_info(const lsc_ctx *ctx, int *world_size, int *rank, int *shard_rows, int *table_rows);

/* One tick of a sharded swarm, device-resident: lsc_tick_device for the rank's agents, the in-place all-gather of
 * d_traj_next (float [table_rows][90]: padded!), then the ideal next state of ALL N agents into d_state
 * (lsc_propagate_device) -- i.e. d_state is read by the tick and overwritten for the next one.  d_cost / d_status /
 * d_iters: the rank's entries only.  Everything is enqueued on hip_stream; no host synchronisation. */
int lsc_tick_device_sharded(lsc_ctx *ctx, float *d_state, const float *d_goal, const float *d_traj_prev, int planner_seq,
                            float *d_traj_next, double *d_cost, int *d_status, int *d_iters, void *hip_stream);

/* Multi-GPU form of lsc_replan_tick (host buffers): every rank passes the same inputs for all N agents, plans its
 * shard, and receives the outputs of ALL N agents (trajectories, costs, statuses, iterations and -- optional --
 * current_goal_position are all-gathered in one RCCL group).  What a replicated MultiSyncSimulator per rank calls. */
int lsc_replan_tick_all(lsc_ctx *ctx, const float *state, const float *goal, const float *prev_traj, int planner_seq,
                        float *out_traj /*[N][3][30]*/, double *out_cost /*[N]*/, int *out_status /*[N]*/,
                        int *out_iters /*[N] or NULL*/, float *out_goal /*[N][3] or NULL*/);

/* MultiSyncSimulator::update()'s ideal-state step on device: state[qi] = traj[qi] evaluated at t = dt
 * (getFutureStateMsg -> getStateFromControlPoints, include/polynomial.hpp:63-97).  All N agents. */
int lsc_propagate_device(lsc_ctx *ctx, const float *d_traj, float *d_state, void *hip_stream);

/* MultiSyncSimulator::savePlanningResult's agent-agent accounting (src/multi_sync_simulator.cpp:446-503) on the device, for
 * the plans of the last lsc_replan_tick / lsc_replan_tick_all.  times[n_times] (n_times <= 64): seconds into the plan at which
 * the swarm is sampled (the reference: 0, multisim_record_time_step, ... below multisim_time_step).  The context must hold the new
 * plans of ALL N agents: the last tick was lsc_replan_tick_all, or lsc_replan_tick of a context whose shard is the whole swarm
 * (LSC_ESTATE otherwise -- the partners' rows would be stale).  For every sample and
 * every agent of this context's shard, the downwash-scaled distance to every other agent over the sum of the two radii
 * (distBetweenAgents, include/util.hpp:225-229): out_ratio[n_times][count] = the minimum over the partners,
 * out_partner[n_times][count] = the first partner attaining it (what the reference's collision message names); either may be
 * NULL.  *out_min = the minimum over everything -- over ALL ranks when the context has a communicator (one ncclAllReduce(min);
 * collective: every rank must call it).  The reference walks all N^2 pairs on the host after every tick. */
int lsc_safety_ratio(lsc_ctx *ctx, const double *times, int n_times, double *out_ratio, int *out_partner, double *out_min);

/* Dense LSC sweep only (TrajPlanner::generateLSC for every ordered pair), device buffers:
 *   d_normal [count][N-1][M][3] float, d_d [count][N-1][M][n+1] double. */
int lsc_sweep_device(lsc_ctx *ctx, const float *d_state, const float *d_traj_prev, int planner_seq,
                     float *d_normal, double *d_d, void *hip_stream);
/* The same sweep with the margins rounded to float32 (d_d32 [count][N-1][M][n+1] float): 180 B per ordered pair instead of
 * 300 -- the dump format for swarms whose table is large; the QP itself always reads the doubles. */
int lsc_sweep_device_f32(lsc_ctx *ctx, const float *d_state, const float *d_traj_prev, int planner_seq,
                         float *d_normal, float *d_d32, void *hip_stream);

/* GJK distance origin <-> conv(points) for `count` independent 6-point hulls (device kernel; test hook
 * for src/openGJK/openGJK.cpp:674-780).  pts [count][6][3] double (host), v [count][3], dist [count]. */
int lsc_gjk_batch(lsc_ctx *ctx, const double *pts, int count, double *v, double *dist);

/* Introspection used by bench.py: name / average device time (ms, HIP events) of the kernels timed since
 * the last reset.  which: 0 = plan kernel (all passes: LSC generation + QP), 1 = dense sweep kernel, 2 = the trajectory
 * all-gather of the sharded ticks, 3 = goal kernel (octomap worlds, mode/goal prior_based), 4 = corridor (SFC) kernel --
 * the per-phase columns of PlanningTimeStatistics (include/sp_const.hpp:89-128) that exist as separate launches; 5 = host
 * wall clock of lsc_replan_tick itself, entry to return (PCIe-inclusive: what the reference-side caller waits for). */
int lsc_kernel_time_ms(lsc_ctx *ctx, int which, double *avg_ms, long *launches);
int lsc_set_timing(lsc_ctx *ctx, int enabled);
/* Per-launch device times (ms) of the launches timed since lsc_set_timing(ctx, 1): up to `capacity` values in launch
 * order; *launches receives how many were timed (bench.py's p99 per-tick solve time). */
int lsc_kernel_times_ms(lsc_ctx *ctx, int which, double *out_ms, long capacity, long *launches);

/* current_goal_position of every agent as used by the last tick, float [N][3] (goal_mode 1: planned on the device). */
int lsc_last_goals(lsc_ctx *ctx, float *goals);

/* Goal-planner introspection (goal_mode 1 + use_octomap; parity tests): lsc_set_goal_trace(ctx, path_cap > 0) makes
 * the following ticks keep each agent's grid path; lsc_get_goal_trace returns, for the shard's agents, the path as
 * grid cells (i, j, k) int [count][path_cap][3], its length [count] (GridBasedPlanner::plan_result.grid_path,
 * src/grid_based_planner.cpp:66), flags [count] (bit 0 retreat rule fired, bit 1 the search without priorities was
 * used) and the number of nodes the search popped [count].  grid_dims / grid_min (optional) describe the grid of
 * GridBasedPlanner::updateGridInfo (src/grid_based_planner.cpp:72-93). */
int lsc_set_goal_trace(lsc_ctx *ctx, int path_cap);
int lsc_get_goal_trace(lsc_ctx *ctx, int *path_cells, int *path_len, int *flags, int *expansions, int grid_dims[3],
                       double grid_min[3]);

/* Active (non-redundant) LSC rows each agent's QP carried in the last tick, [N] (diagnostics). */
int lsc_last_row_counts(lsc_ctx *ctx, int *rows);
/* Neighbour lists of the last tick (diagnostics).  Swarms of >= 512 agents do not walk all N - 1 other agents per agent (the loops of
 * TrajPlanner::generateLSC, src/traj_planner.cpp:1335-1407, and goalPlanningWithPriority, :540-608): a uniform grid built in front of
 * the tick hands every agent of the shard
 *   units [N]               how many (obstacle, segment) units its LSC build looks at instead of all 5 (N - 1); -1: no list (capacity
 *                           overflow), the agent culled by itself;
 *   priority_candidates [N] (optional, may be NULL) how many agents lie within priority_dist_threshold of it -- all the priority rule can
 *                           act on; -1: no list, the agent scanned everybody; 0 when goals are not planned in the plan kernel.
 * Agents outside the shard keep what an earlier tick left.  Returns LSC_ESTATE when the context builds no lists (small swarm, prune != 1,
 * LSC_NO_NEIGHBOUR_LISTS set when lsc_set_agents ran).  Results never depend on the lists. */
int lsc_neighbour_counts(lsc_ctx *ctx, int *units, int *priority_candidates);
/* LSC rows one agent may carry in LDS before the second pass (rows in HBM) takes it over: *lds_rows for the 512-lane latency
 * build (shards of at most one agent per CU), *throughput_rows for the 256-lane build that larger shards use (two workgroups
 * per CU, half the LDS each; 0 = not available, e.g. when max_rows_per_cp was set explicitly). */
int lsc_row_capacity(const lsc_ctx *ctx, int *lds_rows, int *throughput_rows);
/* Rows of each agent's fullest control-point bucket in the last tick, [N]: what the first pass's LDS capacity
 * (max_rows_per_cp) has to hold for the agent not to take the second pass (diagnostics). */
int lsc_last_bucket_max(lsc_ctx *ctx, int *rows);

/* Sum over agents of interior-point iterations since the last reset (bench flop accounting). Synchronises. */
int lsc_iterations_total(lsc_ctx *ctx, long long *total, int reset);
/* Sum over agents of iterations x LSC rows carried (the rows that survived the redundancy pruning), since the last reset of
 * lsc_iterations_total: what the kernels executed, next to the 27 (N - 1) rows per iteration of the reference's model
 * (bench.py: roofline.frac_executed). */
int lsc_row_iterations_total(lsc_ctx *ctx, long long *total);

/* Counters of the active-set solve (lsc_config.solver 1) since the last reset of lsc_iterations_total: out[0] agent-replans it finished,
 * [1] agent-replans it handed to the interior point (working set beyond its capacity, dependent rows, no admissible step: the
 * interior point then decides, infeasible verdicts included), [2] changes of the working set, [3] interior-point iterations of the
 * handed-over agents.  Synchronises. */
int lsc_solver_stats(lsc_ctx *ctx, long long out[4]);

/* Diagnostics.  lsc_phase_profile: enable=1 selects the instrumented plan kernel and clears its counters, 0 goes
 * back to the production kernel, -1 only reads; out (may be NULL) gets [N][16] counts (100 MHz wall clock)
 * per phase: setup, LSC build, IP init, residual pass, row reduction, Hessian assembly, Cholesky, triangular
 * solves, affine pass, corrector pass, step+update, output; then two parts of the row reduction (the LSC-bucket sums as wave 0
 * sees them, the axis-row gather as the last lane sees it) and two spare slots.  lsc_solver_residuals: [N][4] last duality gap,
 * primal residual, stationarity residual, objective. */
int lsc_phase_profile(lsc_ctx *ctx, int enable, long long *out);
/* The same for the goal planner's register-resident grid search: out gets [N][16] counters, shader cycles unless noted:
 * prologue (priority / retreat rule), grid set-up, search, path + line-of-sight goal; of the search: findMin, deleteMin,
 * neighbour screening, insertions; of those: cycles and count of the pops and of the insertions that took the general LDS
 * routines (rows beyond 64 entries, rehashes); the rest is reserved. */
int lsc_goal_profile(lsc_ctx *ctx, int enable, long long *out);
/* Test hook, host only (no device needed): the 32-bit key table of the grid search for squared cell distances 0 .. words-1 --
 * out[d] = floor(sqrt d) << *rank_bits | rank of frac(sqrt d); a search key is (steps << rank_bits) + out[d2].  LSC_ESTATE when the
 * table's exactness conditions do not hold for that range (the search then keeps 64-bit keys). */
int lsc_goal_key_table(int words, unsigned int *out, int *rank_bits);
/* lsc_general_profile: sections of the alternate-mode kernel -- BVC / slack modes / disturbed agents -- collected while
 * lsc_phase_profile is enabled and cleared with it; out[N][16] shader cycles: set-up, start, residual pass, row reduction,
 * assembly, factorization, solves, affine pass, corrector right-hand side, its reduction and assembly, step; [12] the
 * iterations and [13] the solves of the agent. */
int lsc_general_profile(lsc_ctx *ctx, long long *out);
int lsc_solver_residuals(lsc_ctx *ctx, double *out);
/* QP failure forensics (TrajOptimizer::solve exports the model it could not solve: log/QPmodel.lp, src/traj_optimizer.cpp:99-153).
 * Writes the QP of `agent` as the LAST host-buffer tick (lsc_replan_tick / lsc_replan_tick_all) posed it, in CPLEX LP format with
 * the reference's variable names (x_m_i, y_m_i, z_m_i) and populatebyrow's row order (c1, c2, ...), so that it can be diffed against a
 * reference dump or fed to any LP/QP solver.  LSC mode, rows without slack variables. */
int lsc_dump_qp(lsc_ctx *ctx, int agent, const char *path);
/* Reads the [64][8] per-iteration trace recorded for the agent selected by the PREVIOUS call (out may be NULL),
 * then selects `agent` (-1: off): gap, |rp|, objective, affine step, sigma, step, |dx_aff|, mu.  Agents solved by the
 * alternate-mode kernel record the same eight values per iteration (the matrices behind the [64][8] block are the fast path's only). */
int lsc_solver_trace(lsc_ctx *ctx, int agent, double *out);

#ifdef __cplusplus
}
#endif
#endif
