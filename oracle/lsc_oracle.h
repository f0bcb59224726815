/*
 * lsc_oracle.h -- CPU restatement ("oracle") of the per-agent replanning tick of
 * qwerty35/lsc_planner.  TEST INFRASTRUCTURE ONLY: nothing under lsc_planner_amd/
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and there only as the checker.
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   - GJK           : pinned against the reference's own openGJK built in-container
 *                     (oracle/_ref, tests/golden/gjk_vectors.npz)
 *   - QP assembly   : pinned against the reference fixture log/QPmodel.lp
 *                     (tests/golden/qpmodel_lp.json)
 *   - QP optimum    : CPLEX 20.1 is proprietary and absent and no reference fixture holds an
 *                     optimum; pinned instead to an independent solver, HiGHS (tests/highs_qp.py,
 *                     tests/test_oracle_pins.py): the reference's log/QPmodel.lp read verbatim
 *                     (infeasible), all golden-tick QPs and >1300 soak QPs to <= 1e-7 relative
 *                     cost, every infeasible verdict certified by a phase-1 LP.
 *   - EDT / SFC     : octomap and dynamicEDT3D are external and absent -> PARITY UNPINNED at
 *                     the library level (file format and distance semantics restated from
 *                     their documentation); the box growth follows corridor_constructor.hpp
 *                     line by line; the reference's simple_forest.bt is the map fixture.
 *   - goal planning : Astar-3D needs tinyxml2 (absent) -> PARITY UNPINNED; restated in C++ on
 *                     the same std::unordered_map so that the reference's hash-order
 *                     tie-breaking is inherited from libstdc++ (lsc_oracle_goal.cpp).
 *
 * Layouts (shared with the product C-ABI so that buffers compare element-wise):
 *   traj   : float  [N][3][M*(n+1)]   axis-major, x[k*30 + m*6 + i]   (M=5, n=5)
 *   state  : float  [N][9]            pos(3), vel(3), acc(3)
 *   normal : float  [N][N-1][M][3]    LSC normal (already de-scaled: z / downwash)
 *   d      : double [N][N-1][M][n+1]  LSC margins
 */
#ifndef LSC_ORACLE_H
#define LSC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Segments M = (int)((horizon + 1e-9) / dt), src/traj_optimizer.cpp:9: a build parameter here as in the product.  liblsc_oracle.so
 * is built for M = 5 (every shipped launch file), liblsc_oracle_m4.so from the same files with -DORC_M=4 (the C++ defaults of
 * src/param.cpp:66-67: dt 0.5, horizon 2.0). */
#ifndef ORC_M
#define ORC_M 5
#endif
#define ORC_N 5
#define ORC_PHI 3
#define ORC_NC (ORC_N + 1)            /* control points per segment   */
#define ORC_SEGV (ORC_M * ORC_NC)     /* 30 variables per axis        */
#define ORC_NV (3 * ORC_SEGV)         /* 90 QP variables              */
#define ORC_NEQ (3 * (ORC_PHI * ORC_M) + 3 * (ORC_PHI - 1)) /* 45 + 6 = 51 */

typedef struct {
    double dt;               /* segment time (0.2)                         */
    double w_control;        /* opt/control_input_weight (0.01)            */
    double w_terminal;       /* opt/terminal_weight (1)                    */
    float  world_min[3];     /* mission world box (float32 as point3d)     */
    float  world_max[3];
    double reset_threshold;  /* multisim/reset_threshold (0.15)            */
    int    use_sfc;          /* world/use_octomap                          */
    int    obs_f32;          /* 1: obstacle radius/downwash pass through a float32 message field */
    int    world_dimension;  /* world/dimension (src/param.cpp:12): 2 = planar world: goal grid at z = world_z_2d, QP in x and y only
                                (src/traj_optimizer.cpp:8-90, 264-536), own position read at z = world_z_2d
                                (src/traj_planner.cpp:304-314); anything else = 3 */
    double world_z_2d;       /* world/z_2d (src/param.cpp:15)                                                  */
} orc_params;

/* one sparse QP row:  sum val[j]*x[idx[j]]  (sense)  rhs ; sense: 0 '=', 1 '>=', 2 '<=' */
typedef struct {
    int    nnz;
    int    idx[9];
    double val[9];
    double rhs;
    int    sense;
} orc_row;

int orc_segments(void);   /* ORC_M of this build */

/* ---- constants (src/traj_optimizer.cpp:169-236, include/polynomial.hpp:415-428) ---- */
void orc_bernstein_basis(double B[ORC_NC * ORC_NC]);
void orc_qbase(double dt, double Q[ORC_NC * ORC_NC]);
void orc_aeq_base(double dt, double A[(ORC_PHI * ORC_M) * ORC_SEGV]);

/* ---- GJK: distance origin <-> conv(pts)  (src/openGJK/openGJK.cpp:633-780) ---- */
double orc_gjk_origin(const double (*pts)[3], int npts, double v[3], int *nvrtx, int *iters);

/* ---- closest point + float32 normalise (include/geometry.hpp:364-394, src/traj_planner.cpp:2030-2043) ---- */
void orc_normal_between_polys(const float (*pa)[3], const float (*po)[3], int npts, float normal[3]);

/* ---- prediction / initial trajectory (src/traj_planner.cpp:699-712, 829-878, 997-1061) ---- */
void orc_shift_traj(const float *prev /*[3][30]*/, float *out /*[3][30]*/);
void orc_const_vel_traj(const float pos[3], const float vel[3], double dt, float *out /*[3][30]*/);

/* ---- LSC for one (agent, obstacle) pair (src/traj_planner.cpp:1310-1407) ---- */
void orc_lsc_pair(const float *init_traj /*[3][30]*/, const float *obs_traj /*[3][30]*/,
                  double r_a, double r_o, double dw_a, double dw_o,
                  float normal[ORC_M][3], double d[ORC_M][ORC_NC]);

/* ---- QP assembly in the reference's row order (src/traj_optimizer.cpp:261-548) ---- */
int  orc_terminal_segments(const float goal[3], const float pos[3], double v_nom, double dt);
/* rows must hold at least 51 + 27*n_obs + 252 (+162 with SFC) entries; returns #rows.
 * P is the dense nv x nv Hessian of (1/2)x'Px (nv = orc_qp_nvars(prm): 90, or 60 in a planar world -- dim = world/dimension,
 * src/traj_optimizer.cpp:8, 264-266), c the linear term, cst the constant. */
int  orc_qp_nvars(const orc_params *prm);
int  orc_qp_assemble(const orc_params *prm, const float state[9], const float goal[3], double v_nom,
                     const double vmax[3], const double amax[3],
                     int n_obs, const float *obs_traj /*[n_obs][3][30]*/,
                     const float *normal /*[n_obs][M][3]*/, const double *d /*[n_obs][M][6]*/,
                     const float *sfc /*[M][6] or NULL*/,
                     double *P, double *c, double *cst, double *lo, double *hi, orc_row *rows);

/* ---- exact fp64 QP solve (replaces CPLEX, src/traj_optimizer.cpp:31-154) ----
 * status: 0 optimal, 1 infeasible / not converged (caller keeps stale trajectory)
 * kkt[4] (optional): stationarity, primal infeasibility, dual infeasibility, complementarity */
int  orc_qp_solve(const double *P, const double *c, double cst, const double *lo, const double *hi,
                  const orc_row *rows, int nrows, double *x, double *cost, int *iters, double *kkt);

int  orc_qp_solve_n(int nv, const double *P, const double *c, double cst, const double *lo, const double *hi,
                    const orc_row *rows, int nrows, double *x, double *cost, int *iters, double *kkt);

/* ---- alternate planner modes (lsc_oracle_modes.c) ---- */
typedef struct {
    int    planner_mode;          /* 0 LSC, 1 BVC (mode/planner, src/param.cpp:33-48)                              */
    int    slack_mode;            /* 0 none, 1 dynamical_limit, 2 collision_constraint (SlackMode, sp_const.hpp)   */
    double slack_weight;          /* opt/slack_collision_weight (100000 in every launch file)                      */
    int    n_constraint_segments; /* opt/N_constraint_segments, -1 -> M                                            */
    double reset_threshold;       /* multisim/reset_threshold (0.15); <= 0 switches the disturbance checks off     */
} orc_modes;
int  orc_slack_count(const orc_modes *md, int n_obs, const unsigned char *slack_flags);
void orc_bvc_pair(const float *init_traj, const float *obs_traj, double r_a, double r_o, double dw_a, double dw_o,
                  float normal[ORC_M][3], double d[ORC_M][ORC_NC]);
int  orc_qp_assemble_ex(const orc_params *prm, const orc_modes *md, const float state[9], const float goal[3], double v_nom,
                        const double vmax[3], const double amax[3], int n_obs, const float *obs_traj, const float *normal,
                        const double *d, const float *sfc, const unsigned char *slack_flags, int *nv_out, double *P, double *c,
                        double *cst, double *lo, double *hi, orc_row *rows);
void orc_disturbance_update(const orc_params *prm, const orc_modes *md, int N, const float *state, const float *prev_traj,
                            int planner_seq, unsigned char *slack_set, int *sfc_init, unsigned char *own_reset);
void orc_goal_prior_based_ex(int N, int qi, const float *state, const float *desired_goal, const float *prev_traj, int planner_seq,
                             double dt, double goal_threshold, double priority_dist_threshold, double goal_radius,
                             const unsigned char *slack_row, int own_reset, float out_goal[3]);

/* ---- goal planning, prior_based, empty map (src/traj_planner.cpp:540-608; see the .c file) ---- */
void orc_goal_prior_based(int N, int qi, const float *state, const float *desired_goal, const float *prev_traj,
                          int planner_seq, double dt, double goal_threshold, double priority_dist_threshold,
                          double goal_radius, float out_goal[3]);

/* ---- state propagation (include/polynomial.hpp:63-97 at t = dt; multi_sync_simulator.cpp:190-247) ---- */
void orc_next_state(const float *traj /*[3][30]*/, double dt, float state[9]);

/* ---- one synchronous replan tick for the whole swarm (multi_sync_simulator.cpp:320-337) ----
 * planner_seq is the value AFTER TrajPlanner::plan's increment (1 on the first tick).
 * stale_traj: optimiser's previous `trajectory` member per agent (kept on failure); updated in place.
 * sfc_io: [N][M][6] persistent SFC boxes (NULL when !use_sfc).
 * out_normal / out_d may be NULL.  nthreads<=1 -> sequential like the reference. */
int  orc_tick(const orc_params *prm, int N, const float *state, const float *goal, const float *prev_traj,
              int planner_seq, const double *radius, const double *downwash,
              const double *vmax, const double *amax, const double *vnom,
              float *stale_traj, float *sfc_io,
              float *out_traj, double *out_cost, int *out_status, int *out_iters,
              float *out_normal, double *out_d, int nthreads);
/* SFC inputs of the next orc_tick calls (use_sfc): distance field, world/resolution, per-agent flag_initialize_sfc [N] */
void orc_tick_set_map(const void *edt, double world_res, int *sfc_init_flags);

/* ---- EDT + SFC (include/corridor_constructor.hpp:18-245; dynamicEDT3D restated) ---- */
typedef struct {
    const float *dist;   /* dense [nx][ny][nz], metres, truncated */
    int nx, ny, nz;
    int key_min[3];      /* octomap key of cell (0,0,0)            */
    double res;
} orc_edt;
int  orc_bt_read(const char *path, double *res, int **leaves /*[n][4] min key + cube edge*/, int *n);
int  orc_edt_build(const int *leaves, int n, double res, const float world_min[3], const float world_max[3],
                   double maxdist, orc_edt *edt);
/* the same field by dynamicEDT3D's published propagation (26-neighbour "lower" wavefront): see lsc_oracle_sfc.c */
int  orc_edt_brushfire(const int *leaves, int n, double res, const float world_min[3], const float world_max[3],
                       double maxdist, orc_edt *edt, int *sq_out);
int  orc_expand_box(const orc_params *prm, const orc_edt *edt, double world_res,
                    const float point[3], const float goal[3], double radius, double box[6]);
int  orc_update_sfc(const orc_params *prm, const orc_edt *edt, double world_res, const float pos[3], const float goal[3],
                    const float *prev_traj, double radius, float *sfc /*[M][6]*/, int *init_flag);

int  orc_tick_ex(const orc_params *prm, const orc_modes *md, int N, const float *state, const float *goal, const float *prev_traj,
                 int planner_seq, const double *radius, const double *downwash, const double *vmax, const double *amax,
                 const double *vnom, float *stale_traj, unsigned char *slack_set, const orc_edt *edt, double world_res,
                 float *sfc_io, int *sfc_init, float *out_traj, double *out_cost, int *out_status, int *out_iters,
                 float *out_normal, double *out_d, int nthreads);

/* ---- goal planning with a distance field: grid A* + line-of-sight goal (lsc_oracle_goal.cpp) ---- */
long orc_astar_last_expansions(void);
void orc_grid_dims(const orc_params *prm, double grid_res, int dims[3], double gmin[3]);
int  orc_astar(const unsigned char *occ, const int dims[3], const int start[3], const int goal[3], int *path_out, int max_len);
void orc_goal_prior_based_map(const orc_params *prm, const orc_edt *edt, double world_res, double grid_res, double grid_margin,
                              int N, int qi, const float *state, const float *desired_goal, const float *prev_traj,
                              int planner_seq, double goal_threshold, double priority_dist_threshold, double goal_radius,
                              const double *radius, const double *downwash, const unsigned char *slack_row /* [N] or NULL */,
                              int own_reset, float out_goal[3], int *path_out, int max_path, int *path_len, int *flags);

#ifdef __cplusplus
}
#endif
#endif
