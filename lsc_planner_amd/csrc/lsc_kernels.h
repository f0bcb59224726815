// lsc_kernels.h -- argument blocks shared by the kernels (lsc_kernels.hip) and the C ABI (lsc_abi.cpp)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "lsc_model.hpp"

#define LSC_STATUS_OK_K 0
#define LSC_STATUS_INFEASIBLE_K 1
#define LSC_STATUS_CAPACITY_K 3
#define LSC_STATUS_SFC_K 4
#define LSC_STATUS_GOAL_K 5
#define LSC_STATUS_GENERAL_K 6   /* internal: the agent's QP has an alternate-mode shape -> lsc_general_kernel solves it */

namespace lsc {

struct PlanArgs;
struct NeighArgs;
struct NeighView;
// an argument block read where it lies, in the kernarg segment: constant address space => scalar loads, no private copy
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) PlanArgs KArgs;
#else
typedef const PlanArgs KArgs;
#endif

struct PlanArgs {
    const Model *model;
    const uint32_t *terms;     // packed Hessian assembly terms
    const uint32_t *entries;   // [n_entries+1][2] : (gi<<16|gj, first term)
    int N, first, count, planner_seq;
    int dim2;                  // world/dimension == 2: the launch takes the planar instantiation of the plan kernels (a compile-time switch)
    int cap;                   // LSC rows the LDS pass holds (total over the 27 control points; compact layout)
    int solver;                // 0 interior point; 1 dual active set first, interior point as fallback (lsc_config.solver)
    long long *solver_stats;   // optional [N][4] running counters of the active-set solve, per agent (lsc_solver_stats sums them)
    int *order;                // throughput build: launch order of the shard's agents (longest first), or null
    float *obs_bound;          // throughput build: [N][4] bounding sphere of every agent's predicted control points, or null
    int cap_tp;                // > 0: use the 256-lane throughput build with this row capacity and smem_tp bytes of LDS
    size_t smem_tp;
    NeighArgs *neigh;          // HOST pointer (never read on the device), or null: the context's neighbour-list buffers; run_plan (lsc_abi.cpp) launches
                               // the two kernels of lsc_neigh.hip in front of the tick of a large swarm and sets nv
    const NeighView *nv;       // device: the lists those kernels left for this tick, or null (every agent walks all the others, as small swarms do)
    const float *state;        // [N][9]
    const float *goal;         // [N][3] current goal (mode/goal static) or desired goal (prior_based)
    int goal_mode;             // 0 static, 1 prior_based: goalPlanningWithPriority runs in phase A of the plan kernel
    double goal_threshold, priority_dist_threshold, goal_radius;
    float *goal_out;           // [N][3] current_goal_position actually used (diagnostics / getCurrentGoalPosition)
    float *state_next;         // optional [N][9]: ideal state of the agent at t = dt on its NEW plan (fused propagation)
    float finv;                // (float)pow(dt, -1)
    const float *traj_prev;    // [N][90]
    const double *radius, *radius_obs, *downwash, *downwash_obs;  // [N]; *_obs = value rounded through float32
    const double *vmax, *amax; // [N][3]
    const double *vnom;        // [N]
    float *traj_next;          // [N][90]
    double *cost;              // [N]
    int *status, *iters, *nrows;
    int *bucket_max;           // optional [N]: rows of the fullest control-point bucket (diagnostics)
    long long *iters_acc;      // [N] running sum of interior-point iterations (bench accounting), may be null; entries [N, 2N) = running
                               // sum of iterations x LSC rows the agent carried (the row passes the kernel really executed)
    float *stale;              // [N][90] optimiser's last good trajectory (persistent)
    const float *sfc;          // [N][M][6] or null
    const int *sfc_err;        // [N] or null: seed box blocked -> status 4
    const int *goal_err;       // [N] or null: goal planner capacity overflow -> status 5
    float *out_normal;         // optional dense dump [count][N-1][M][3]
    double *out_d;             // optional dense dump [count][N-1][M][6]
    double *dbg;               // optional [N][4]: last (gap, |rp|, |rd|, objective) seen by the solver
    long long *prof;           // optional [N][PROF_PHASES] phase counters (selects the instrumented kernel)
    double *trace;             // optional [64][8] per-iteration trace of agent trace_agent (diagnostics)
    int trace_agent;
    unsigned char *spill_ws;   // second pass only: HBM row workspaces, one of spill_stride bytes per workgroup
    size_t spill_stride;
    // alternate planner modes (lsc_general.hip)
    const GModel *gmodel;
    int planner_mode;          // 0 LSC, 1 BVC
    int slack_mode;            // 0 none, 1 dynamical_limit, 2 collision_constraint
    int ncs;                   // opt/N_constraint_segments (-1 -> M)
    int general_all;           // every agent takes the general kernel (BVC or a slack mode is configured)
    double slack_w;            // opt/slack_collision_weight
    double reset_thr;          // multisim/reset_threshold; <= 0: no disturbance checks
    unsigned char *ever;       // [N] persistent: agent was seen off its plan at some tick (its rows keep slack variables)
    unsigned char *gen_ws;     // HBM workspaces of lsc_general_kernel, gen_stride bytes per workgroup
    size_t gen_stride;
};
constexpr int PROF_PHASES = 16;

// Neighbour lists of a large swarm (lsc_neigh.hip): which (obstacle, segment) units can carry a row that survives the pruning of phase B,
// found through a uniform grid instead of a walk over all N - 1 obstacles per agent (the loop being replaced: `for oi < N_obs`,
// src/traj_planner.cpp:1335-1407).  Two launches in front of the tick: build (bounds of every agent + one grid insertion each), query
// (one workgroup per agent of the shard: the cells its reach overlaps -> sphere tests -> a sorted unit list in HBM).
constexpr int NEIGH_SLOTS = 12;                  // agents a grid bucket holds (one 32-byte sector: a tagged counter + 12 indices); the rest overflows
constexpr int NEIGH_MIN_AGENTS = 512;            // swarms below this keep the in-kernel cull
constexpr int NEIGH_PRIO_CAP = 64;               // candidates of the priority rule an agent's list holds
// what phase A / B of plan_agent read of the lists (device memory, written once per lsc_set_agents: the pointers do not change)
struct NeighView {
    const unsigned short *list;                  // [N][cap] per agent: the units (obstacle * M + segment) phase B has to look at, ascending
    const int *cnt;                              // [N] entries of the agent's list; < 0: no list (capacity overflow), the agent culls by itself
    const unsigned short *plist;                 // [N][pcap] agents within priority_dist_threshold of the agent's position (goalPlanningWithPriority's candidates)
    const int *pcnt;                             // [N] -1: no information (the agent scans everybody); else entries of plist | (1 << 30 when ANY agent of the
                                                 // swarm is off its plan or was: the disturbance checks of phase A, made once by the build kernel)
    int cap, pcap;
};
struct NeighArgs {
    int N, first, count, planner_seq;
    float dtf;
    int dim2;
    double hv_scale, ha_scale, z2d;              // Model::hv_scale, ha_scale, z2d
    const float *state, *traj_prev;
    const double *radius, *radius_obs, *downwash, *downwash_obs, *vmax, *amax;
    int *order;                                  // launch order of the throughput build (longest agent first), or null
    const int *iters, *nrows;
    float *obs_bound;                            // [N][4] bounding sphere of all predicted control points (the in-kernel cull's, kept for agents without a list)
    float *seg_bound;                            // [N][M][4] bounding sphere of the predicted control points of each segment
    float *reach;                                // [N][M] per segment max_i(|c_{0,2} - p_{m,i}| + reach radius of c_{m,i}), rounded up
    unsigned long long *cells;                   // [hmask + 1][4] buckets: tag << 32 | count, then NEIGH_SLOTS agent indices (16 bit)
    unsigned long long *glob;                    // [16] tagged maxima: obstacle-side radius, overflow count, cell bounding box (6), somebody is off its plan
    unsigned short *ovf;                         // [ovf_cap] agents whose bucket was full
    int ovf_cap;
    unsigned tag, hmask;                         // tag of this tick (buckets of older ticks count as empty: nothing is ever cleared)
    double inv_cell, inv_cell_z;                 // 1 / cell size in x, y and in z (z cells are downwash times taller)
    double sc_max, zscale;                       // max(1, 1 / smallest downwash), max(1, largest downwash)
    unsigned short *list;                        // [N][list_cap]
    int *cnt;                                    // [N]
    int list_cap;
    // phase A's walks over all agents (see NeighView)
    int goal_mode;                               // 1: goalPlanningWithPriority runs in phase A -> candidates by position
    double prio_thr;                             // priority_dist_threshold
    unsigned short *plist;                       // [N][plist_cap]
    int *pcnt;                                   // [N]
    int plist_cap;
    int checks;                                  // the plan kernel's disturbance checks are on (reset_threshold > 0, LSC mode, planner_seq >= 2)
    double reset_thr;
    unsigned char *ever;                         // [N] persistent "was seen off its plan" flags
    const NeighView *view;                       // device copy of the view over list / cnt / plist / pcnt
    long long *prof;                             // optional [count][8]: 100 MHz wall-clock stamps of the query kernel's stages (LSC_NEIGH_PROFILE)
};
hipError_t launch_neigh(const NeighArgs &a, hipStream_t st);

// Several independent swarms -- one argument block each -- planned by ONE launch (blockIdx.y = swarm): the reference flies a list of
// missions back to back (src/multi_sync_simulator_node.cpp:43-70, src/param.cpp:106-122), and a 64-agent swarm is 64 workgroups on a
// 256-CU chip.  The blocks travel in the kernarg segment itself (no copy to the device; kernarg segments hold 4 KB).
constexpr int PLAN_BATCH_MAX = 8;
struct PlanBatch {
    PlanArgs a[PLAN_BATCH_MAX];
};
static_assert(sizeof(PlanBatch) <= 4096, "a batch of argument blocks must fit the kernarg segment");

struct SweepArgs {
    int N, first, count, planner_seq;
    float dtf;
    const float *state, *traj_prev;
    const double *radius, *radius_obs, *downwash, *downwash_obs;
    float *out_normal = nullptr;
    double *out_d = nullptr;
    float *out_d32 = nullptr;  // margins as float32 instead (out_d unused)
};

// Safe Flight Corridor update (TrajPlanner::generateFeasibleSFC), one lane per agent
struct SfcArgs {
    int N, first, count;
    const float *state, *goal, *traj_prev;
    const double *radius;
    const int *img_of_agent;    // [N] index of the blocked-cell integral image matching the agent's margin
    const int *integral;        // [n_img][(nx+1)(ny+1)(nz+1)] inclusive 3-D prefix sums of "EDT < r + res/2 - 1e-5"
    int nx, ny, nz;
    int key_min[3];
    double rf;                  // 1 / tree resolution (OcTreeBaseImpl::resolution_factor)
    double wres;                // world/resolution
    float world_min[3], world_max[3];
    float *sfc;                 // [N][M][6] persistent boxes (box_min, box_max as float)
    int *init_flag;             // [N] flag_initialize_sfc
    int *err;                   // [N] 1 when the seed box already touches an obstacle
    int table_len;              // entries per face table in LDS: steps a face can move inside the world + slack
    int planner_seq;
    double reset_thr;           // initialTrajPlanningCheck: an agent found off its plan re-initialises its corridor
};
hipError_t launch_sfc(const SfcArgs &a, hipStream_t st);

// Goal planning with a distance field (lsc_goal.hip): priority rule, grid A*, line-of-sight goal.  One wave per agent.
struct GoalArgs {
    int N, first, count, planner_seq;
    float dtf;
    const float *state, *goal, *traj_prev;     // goal = desired goals [N][3]
    const double *radius, *downwash, *radius_obs, *downwash_obs;
    double goal_threshold, priority_dist_threshold, goal_radius;
    // planning grid (GridBasedPlanner::updateGridInfo): dims, origin, resolution; cells addressed by key = H*W*z + W*i + j
    int H, W, A;
    int dim2;                                   // world/dimension == 2: start, goal and stamped agents sit in layer 0
    double gmin[3], gres;
    const unsigned char *occ_static;            // [n_img][H*W*A] by key: EDT(cell centre) < radius + grid_margin
    const int *img_of_agent;                    // [N]
    // distance field (DynamicEDTOctomap::getDistance)
    const float *edt;
    int nx, ny, nz, key_min[3];
    double rf, wres;
    // libstdc++ unordered_map bucket-count sequence (13, 29, 59, ...), read from the real container on the host
    int n_nb;
    int nb_seq[16];
    uint32_t nb_magic[16];                      // floor(2^32 / nb_seq[k])
    int row_cap;                                // LDS capacity of one OPEN row (entries)
    int variant, jbits;                         // search variant (see launch_goal: slots | 8 = Key32) and the bits of j in a register-search entry
    const uint32_t *fcode;                      // Key32 table [fcode_n]: floor(sqrt d2) << fcode_rb | rank of frac(sqrt d2), or null
    int fcode_n, fcode_rb;
    float *goal_out;                            // [N][3] current_goal_position
    int *err;                                   // [N] 0 ok, 1 capacity (row / path / g overflow), 2 ray stack overflow
    int *flags;                                 // optional [N]: bit 0 retreat rule, bit 1 search without priorities used
    int *expansions;                            // optional [N]: nodes popped
    int *path_out;                              // optional [count][path_cap] keys of the grid path (start -> goal)
    int path_cap;
    int *path_len;                              // optional [count]
    float *ray_stack;                           // [count][64][24][6] bisection stacks of castRay
    double reset_thr;                           // disturbance checks (multisim/reset_threshold; <= 0 off)
    const unsigned char *ever;                  // [N] persistent "was seen off its plan" flags
    long long *prof;                            // optional [N][8] section cycle counters (selects the instrumented kernel)
    int smem_bytes;                             // dynamic LDS of the launch (set by launch_goal; the poison build fills it)
};
size_t goal_smem_bytes(int H, int W, int A, int cap, int key_words = 0);
int goal_fast_slots(int H, int W, int A, int *jbits);
hipError_t launch_goal(const GoalArgs &a, hipStream_t st);

size_t general_ws_bytes(int N);
hipError_t init_device_general_kernel();
hipError_t launch_general(const PlanArgs &a, int slots, hipStream_t st);
size_t plan_smem_bytes(int n_terms, int n_entries, int rows, bool tables_in_lds = true);
size_t plan_spill_bytes(int N);
hipError_t init_device_kernels();
hipError_t init_device_goal_kernel();
hipError_t launch_plan(const PlanArgs &a, size_t smem, hipStream_t st);
hipError_t launch_plan_batch(const PlanArgs *a, int n, size_t smem, hipStream_t st);
hipError_t launch_general_batch(const PlanArgs *a, int n, int slots, hipStream_t st);
hipError_t launch_plan_spill(const PlanArgs &a, int slots, size_t smem, hipStream_t st);
hipError_t launch_sweep(const SweepArgs &a, hipStream_t st);
hipError_t launch_propagate(const float *traj, float *state, int N, double dt, hipStream_t st);
hipError_t launch_safety(const float *traj, const double *weights, const int *seg, int n_times, int N, int first, int count,
                         const double *radius, const double *downwash, float *pos, double *out_ratio, int *out_partner, hipStream_t st);
hipError_t launch_gjk(const double *pts, int count, double *v, double *dist, hipStream_t st);

}  // namespace lsc
