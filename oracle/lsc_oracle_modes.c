/*
 * lsc_oracle_modes.c -- CPU restatement of the reference's ALTERNATE planner modes.
 * TEST INFRASTRUCTURE ONLY (see lsc_oracle.h): nothing under lsc_planner_amd/ may include, link or call this.
 *
 *   BVC planner mode          TrajPlanner::generateBVC                     src/traj_planner.cpp:1409-1440
 *                             prediction / initial trajectory = current position (:796-807, :1039-1045; param.cpp:40-45)
 *                             no stop-at-horizon rows (src/traj_optimizer.cpp:527-536 is LSC only)
 *                             opt/N_constraint_segments                     src/traj_optimizer.cpp:410, 438
 *   slack variables           SlackMode::DYNAMICALLIMIT / COLLISIONCONSTRAINT in populatebyrow
 *                                                                           src/traj_optimizer.cpp:306-326, 375-390, 455-457, 476-510
 *   disturbance reset         obstaclePredictionCheck / initialTrajPlanningCheck, obs_slack_indices (a set that only ever
 *                             grows), flag_initialize_sfc                   src/traj_planner.cpp:866-878, 1047-1061
 *                             slack obstacles count as "higher priority" in goal planning   :547-551
 *
 * Parity pinning: CPLEX is absent, so the optimum of these QPs is pinned like the LSC one, against HiGHS
 * (tests/test_oracle_pins.py); the assembly follows populatebyrow row by row.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "lsc_oracle.h"

static inline int vidx(int k, int m, int i) { return k * ORC_SEGV + m * ORC_NC + i; }

static double f32_dist3(const float *a, const float *b)
{
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float n2 = dx * dx + dy * dy + dz * dz;
    return sqrt((double)n2);
}

static void f32_normalize3(float *a)
{
    float n2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    double len = sqrt((double)n2);
    if (len > 0) { float l = (float)len; a[0] /= l; a[1] /= l; a[2] /= l; }   /* octomath: a zero vector stays zero */
}

/* number of slack variables the reference creates (src/traj_optimizer.cpp:306-326) */
int orc_slack_count(const orc_modes *md, int n_obs, const unsigned char *slack_flags)
{
    if (md->slack_mode == 1) return 2 * ORC_M;
    int any = 0;
    for (int oi = 0; oi < n_obs && slack_flags; oi++) any |= slack_flags[oi];
    if (md->slack_mode == 2 || any) return n_obs * ORC_M;
    return 0;
}

/* TrajPlanner::generateBVC (:1409-1440): one normal / margin per obstacle, shared by every segment */
void orc_bvc_pair(const float *init_traj, const float *obs_traj, double r_a, double r_o, double dw_a, double dw_o,
                  float normal[ORC_M][3], double d[ORC_M][ORC_NC])
{
    const double downwash = (dw_a * r_a + dw_o * r_o) / (r_a + r_o);
    float p[3] = {init_traj[0], init_traj[ORC_SEGV], (float)((double)init_traj[2 * ORC_SEGV] / downwash)};
    float q[3] = {obs_traj[0], obs_traj[ORC_SEGV], (float)((double)obs_traj[2 * ORC_SEGV] / downwash)};
    float rel[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
    float nv[3] = {rel[0], rel[1], rel[2]};
    f32_normalize3(nv);
    const float dot = rel[0] * nv[0] + rel[1] * nv[1] + rel[2] * nv[2];
    const double dd = 0.5 * ((r_o + r_a) + (double)dot);
    nv[2] = (float)((double)nv[2] / downwash);
    for (int m = 0; m < ORC_M; m++) {
        normal[m][0] = nv[0]; normal[m][1] = nv[1]; normal[m][2] = nv[2];
        for (int i = 0; i < ORC_NC; i++) d[m][i] = dd;
    }
}

/* populatebyrow with every mode switch (src/traj_optimizer.cpp:261-539).  Variables: the 90 control-point coordinates, then
 * the slack variables (offset_slack = 90).  P is (nv x nv), c / lo / hi are nv long; rows needs
 * 51 + 27 n_obs + 252 + 162 entries.  Returns the number of rows; *nv_out the number of variables. */
int orc_qp_assemble_ex(const orc_params *prm, const orc_modes *md, const float state[9], const float goal[3], double v_nom,
                       const double vmax[3], const double amax[3], int n_obs, const float *obs_traj, const float *normal,
                       const double *d, const float *sfc, const unsigned char *slack_flags, int *nv_out, double *P, double *c,
                       double *cst, double *lo, double *hi, orc_row *rows)
{
    const double dt = prm->dt;
    const int nslack = orc_slack_count(md, n_obs, slack_flags);
    /* dim = param.world_dimension (src/traj_optimizer.cpp:8): a planar world has no z variables, no z rows, and its
     * collision / corridor rows drop their z term (:264-266, 330, 367, 394, 411, 423, 450, 469, 529) */
    const int dim = prm->world_dimension == 2 ? 2 : 3;
    const int nv = dim * ORC_SEGV + nslack, off = dim * ORC_SEGV;
    const int ncs = md->n_constraint_segments < 0 ? ORC_M : md->n_constraint_segments;
    double Q[ORC_NC * ORC_NC], Aeq[(ORC_PHI * ORC_M) * ORC_SEGV];
    orc_qbase(dt, Q);
    orc_aeq_base(dt, Aeq);
    *nv_out = nv;

    for (int k = 0; k < dim; k++)
        for (int m = 0; m < ORC_M; m++)
            for (int i = 0; i < ORC_NC; i++) {
                int r = vidx(k, m, i);
                if (m == 0 && i < 3) { lo[r] = -INFINITY; hi[r] = INFINITY; }
                else { lo[r] = (double)prm->world_min[k]; hi[r] = (double)prm->world_max[k]; }
            }
    for (int j = 0; j < nslack; j++) { lo[off + j] = -INFINITY; hi[off + j] = 0.0; }      /* IloNumVar(env, -inf, 0) :310, 320 */

    memset(P, 0, sizeof(double) * (size_t)nv * nv);
    memset(c, 0, sizeof(double) * (size_t)nv);
    *cst = 0;
    for (int k = 0; k < dim; k++)
        for (int m = 0; m < ORC_M; m++)
            for (int i = 0; i < ORC_NC; i++)
                for (int j = 0; j < ORC_NC; j++)
                    if (Q[i * ORC_NC + j] != 0 && prm->w_control != 0)
                        P[(size_t)vidx(k, m, i) * nv + vidx(k, m, j)] += 2.0 * prm->w_control * Q[i * ORC_NC + j];
    int T = orc_terminal_segments(goal, state, v_nom, dt);       /* the 3-D norm in either case (:541-548) */
    for (int m = ORC_M - T; m < ORC_M; m++)
        for (int k = 0; k < dim; k++) {
            int r = vidx(k, m, ORC_N);
            double g = (double)goal[k];
            P[(size_t)r * nv + r] += 2.0 * prm->w_terminal;
            c[r] += -2.0 * prm->w_terminal * g;
            *cst += prm->w_terminal * g * g;
        }
    /* slack cost :375-390: slack_collision_weight (M - m)/M eps^2, for every slack variable that exists */
    if (md->slack_mode == 1) {
        for (int i = 0; i < 2; i++)
            for (int m = 0; m < ORC_M; m++) {
                int r = off + ORC_M * i + m;
                P[(size_t)r * nv + r] += 2.0 * md->slack_weight * ((double)(ORC_M - m) / ORC_M);
            }
    } else if (nslack) {
        for (int oi = 0; oi < n_obs; oi++)
            for (int m = 0; m < ORC_M; m++) {
                int r = off + ORC_M * oi + m;
                P[(size_t)r * nv + r] += 2.0 * md->slack_weight * ((double)(ORC_M - m) / ORC_M);
            }
    }

    int nr = 0;
    for (int k = 0; k < dim; k++)
        for (int r = 0; r < ORC_PHI * ORC_M; r++) {
            orc_row *R = &rows[nr++];
            R->nnz = 0; R->sense = 0;
            for (int j = 0; j < ORC_SEGV; j++)
                if (Aeq[r * ORC_SEGV + j] != 0) {
                    R->idx[R->nnz] = k * ORC_SEGV + j;
                    R->val[R->nnz] = Aeq[r * ORC_SEGV + j];
                    R->nnz++;
                }
            R->rhs = (r < 3) ? (double)state[3 * r + k] : 0.0;
        }
    if (prm->use_sfc && sfc) {
        for (int m = 0; m < ncs; m++)
            for (int f = 0; f < 2 * dim; f++) {                   /* Box::convertToLSCs(world_dimension): 2 dim half-spaces */
                int ax = f / 2;
                double sgn = (f & 1) ? -1.0 : 1.0;
                double dd = (f & 1) ? -(double)sfc[m * 6 + 3 + ax] : (double)sfc[m * 6 + ax];
                for (int j = 0; j < ORC_NC; j++) {
                    if (m == 0 && j < ORC_PHI) continue;
                    orc_row *R = &rows[nr++];
                    R->nnz = 1; R->sense = 1;
                    R->idx[0] = vidx(ax, m, j); R->val[0] = sgn; R->rhs = dd;
                }
            }
    }
    /* LSC or BVC :437-466 */
    for (int oi = 0; oi < n_obs; oi++)
        for (int m = 0; m < ncs; m++)
            for (int i = 0; i < ORC_NC; i++) {
                if (m == 0 && i < ORC_PHI) continue;
                const float *nv3 = normal + (oi * ORC_M + m) * 3;
                orc_row *R = &rows[nr++];
                R->nnz = dim; R->sense = 1;
                double rhs = d[(oi * ORC_M + m) * ORC_NC + i];
                for (int k = 0; k < dim; k++) {                   /* :446-453: the z term only `if (dim == 3)` */
                    double q = (double)obs_traj[(oi * 3 + k) * ORC_SEGV + m * ORC_NC + i];
                    R->idx[k] = vidx(k, m, i);
                    R->val[k] = (double)nv3[k];
                    rhs += (double)nv3[k] * q;
                }
                if (md->slack_mode == 2 || (md->slack_mode != 1 && slack_flags && slack_flags[oi])) {
                    /* expr += -(d + eps) :455-457 */
                    R->idx[dim] = off + ORC_M * oi + m; R->val[dim] = -1.0; R->nnz = dim + 1;
                }
                R->rhs = rhs;
            }
    for (int k = 0; k < dim; k++)
        for (int m = 0; m < ORC_M; m++) {
            for (int i = 0; i < ORC_N; i++) {
                if (m == 0 && (i == 0 || i == 1)) continue;
                for (int sg = 0; sg < 2; sg++) {
                    double f = (sg ? -1.0 : 1.0) * pow(dt, -1) * ORC_N;
                    orc_row *R = &rows[nr++];
                    R->nnz = 2; R->sense = 2;
                    R->idx[0] = vidx(k, m, i + 1); R->val[0] = f;
                    R->idx[1] = vidx(k, m, i); R->val[1] = -f;
                    if (md->slack_mode == 1) { R->idx[2] = off + m; R->val[2] = 1.0; R->nnz = 3; }      /* :476-483 */
                    R->rhs = vmax[k];
                }
            }
            for (int i = 0; i < ORC_N - 1; i++) {
                if (m == 0 && i == 0) continue;
                for (int sg = 0; sg < 2; sg++) {
                    double f = (sg ? -1.0 : 1.0) * pow(dt, -2) * ORC_N * (ORC_N - 1);
                    orc_row *R = &rows[nr++];
                    R->nnz = 3; R->sense = 2;
                    R->idx[0] = vidx(k, m, i + 2); R->val[0] = f;
                    R->idx[1] = vidx(k, m, i + 1); R->val[1] = -2 * f;
                    R->idx[2] = vidx(k, m, i); R->val[2] = f;
                    if (md->slack_mode == 1) { R->idx[3] = off + ORC_M + m; R->val[3] = 1.0; R->nnz = 4; }   /* :498-510 */
                    R->rhs = amax[k];
                }
            }
        }
    if (md->planner_mode == 0) {                          /* stop at the horizon: LSC only :527-536 */
        for (int k = 0; k < dim; k++)
            for (int i = 1; i < ORC_PHI; i++) {
                orc_row *R = &rows[nr++];
                R->nnz = 2; R->sense = 0;
                R->idx[0] = vidx(k, ORC_M - 1, ORC_N); R->val[0] = 1.0;
                R->idx[1] = vidx(k, ORC_M - 1, ORC_N - i); R->val[1] = -1.0;
                R->rhs = 0.0;
            }
    }
    return nr;
}

/* One tick of the whole swarm with the alternate-mode switches.  On top of orc_tick:
 *   slack_set  [N][N] persistent bytes, slack_set[qi][qj] = 1 once agent qi has put agent qj into obs_slack_indices
 *              (the reference never removes an index); updated in place
 *   sfc_init   [N] flag_initialize_sfc (set again by initialTrajPlanningCheck), may be NULL without a map
 * goal is the current goal of every agent (goal planning stays outside, like orc_tick; use orc_goal_prior_based_ex for the
 * priority rule with slack obstacles).  Empty maps only when planner_mode == 1 (generateSFC throws in BVC mode). */
int orc_tick_ex(const orc_params *prm, const orc_modes *md, int N, const float *state, const float *goal, const float *prev_traj,
                int planner_seq, const double *radius, const double *downwash, const double *vmax, const double *amax,
                const double *vnom, float *stale_traj, unsigned char *slack_set, const orc_edt *edt, double world_res,
                float *sfc_io, int *sfc_init, float *out_traj, double *out_cost, int *out_status, int *out_iters,
                float *out_normal, double *out_d, int nthreads)
{
    const int n_obs = N - 1;
    int rc = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 1 ? nthreads : 1)
#endif
    for (int qi = 0; qi < N; qi++) {
        const size_t no = (size_t)(n_obs > 0 ? n_obs : 1);
        float init_traj[ORC_NV];
        float *obs_traj = (float *)malloc(sizeof(float) * ORC_NV * no);
        float *nrm = (float *)malloc(sizeof(float) * 3 * ORC_M * no);
        double *dd = (double *)malloc(sizeof(double) * ORC_NC * ORC_M * no);
        unsigned char *flags = (unsigned char *)calloc(no, 1);
        orc_row *rows = (orc_row *)malloc(sizeof(orc_row) * (size_t)(51 + 27 * n_obs + 252 + 162));
        unsigned char *myset = slack_set + (size_t)qi * N;
        /* currentStateCallback (src/traj_planner.cpp:304-314): planar world -> the agent's own position sits at z = world/z_2d */
        float own[9];
        memcpy(own, state + 9 * qi, sizeof own);
        if (prm->world_dimension == 2) own[2] = (float)prm->world_z_2d;
        const float *pos = own;

        /* ---- obstaclePrediction (+Check): :610-637, 866-878 */
        int oi = 0;
        for (int qj = 0; qj < N; qj++) {
            if (qj == qi) continue;
            float *ot = obs_traj + (size_t)oi * ORC_NV;
            const float *op = state + 9 * qj;
            if (md->planner_mode == 1) {                                   /* PredictionMode::POSITION */
                for (int k = 0; k < 3; k++) for (int t = 0; t < ORC_SEGV; t++) ot[k * ORC_SEGV + t] = op[k];
            } else if (planner_seq < 2) orc_const_vel_traj(op, op + 3, prm->dt, ot);
            else orc_shift_traj(prev_traj + (size_t)qj * ORC_NV, ot);
            if (md->reset_threshold > 0) {
                float p0[3] = {ot[0], ot[ORC_SEGV], ot[2 * ORC_SEGV]};
                if (f32_dist3(p0, op) > md->reset_threshold) {
                    myset[qj] = 1;
                    for (int k = 0; k < 3; k++) for (int t = 0; t < ORC_SEGV; t++) ot[k * ORC_SEGV + t] = op[k];
                }
            }
            oi++;
        }
        /* ---- initialTrajPlanning (+Check): :930-957, 1047-1061 */
        if (md->planner_mode == 1) {
            for (int k = 0; k < 3; k++) for (int t = 0; t < ORC_SEGV; t++) init_traj[k * ORC_SEGV + t] = pos[k];
        } else if (planner_seq < 2) orc_const_vel_traj(pos, pos + 3, prm->dt, init_traj);
        else orc_shift_traj(prev_traj + (size_t)qi * ORC_NV, init_traj);
        if (md->reset_threshold > 0) {
            float p0[3] = {init_traj[0], init_traj[ORC_SEGV], init_traj[2 * ORC_SEGV]};
            if (f32_dist3(p0, pos) > md->reset_threshold) {
                for (int qj = 0; qj < N; qj++) if (qj != qi) myset[qj] = 1;
                for (int k = 0; k < 3; k++) for (int t = 0; t < ORC_SEGV; t++) init_traj[k * ORC_SEGV + t] = pos[k];
                if (sfc_init) sfc_init[qi] = 1;
            }
        }
        oi = 0;
        for (int qj = 0; qj < N; qj++) { if (qj == qi) continue; flags[oi++] = myset[qj]; }

        /* ---- generateCollisionConstraints :1225-1250 */
        oi = 0;
        for (int qj = 0; qj < N; qj++) {
            if (qj == qi) continue;
            float *ot = obs_traj + (size_t)oi * ORC_NV;
            double r_o = prm->obs_f32 ? (double)(float)radius[qj] : radius[qj];
            double dw_o = prm->obs_f32 ? (double)(float)downwash[qj] : downwash[qj];
            if (md->planner_mode == 1)
                orc_bvc_pair(init_traj, ot, radius[qi], r_o, downwash[qi], dw_o, (float(*)[3])(nrm + (size_t)oi * ORC_M * 3),
                             (double(*)[ORC_NC])(dd + (size_t)oi * ORC_M * ORC_NC));
            else
                orc_lsc_pair(init_traj, ot, radius[qi], r_o, downwash[qi], dw_o, (float(*)[3])(nrm + (size_t)oi * ORC_M * 3),
                             (double(*)[ORC_NC])(dd + (size_t)oi * ORC_M * ORC_NC));
            oi++;
        }
        if (out_normal) memcpy(out_normal + (size_t)qi * n_obs * ORC_M * 3, nrm, sizeof(float) * 3 * ORC_M * (size_t)n_obs);
        if (out_d) memcpy(out_d + (size_t)qi * n_obs * ORC_M * ORC_NC, dd, sizeof(double) * ORC_NC * ORC_M * (size_t)n_obs);
        const float *sfc = (prm->use_sfc && sfc_io) ? sfc_io + (size_t)qi * ORC_M * 6 : NULL;
        int sfc_rc = 0;
        if (sfc && edt && sfc_init)
            sfc_rc = orc_update_sfc(prm, edt, world_res, pos, goal + 3 * qi, prev_traj + (size_t)qi * ORC_NV, radius[qi],
                                    sfc_io + (size_t)qi * ORC_M * 6, &sfc_init[qi]);

        /* ---- trajOptimization */
        const int nslack = orc_slack_count(md, n_obs, flags);
        const int nxy = (prm->world_dimension == 2 ? 2 : 3) * ORC_SEGV;
        const int nv = nxy + nslack;
        double *P = (double *)malloc(sizeof(double) * (size_t)nv * nv);
        double *c = (double *)malloc(sizeof(double) * nv), *lo = (double *)malloc(sizeof(double) * nv);
        double *hi = (double *)malloc(sizeof(double) * nv), *x = (double *)malloc(sizeof(double) * nv);
        double cst, cost;
        int nvv, iters = 0;
        int nr = orc_qp_assemble_ex(prm, md, pos, goal + 3 * qi, vnom[qi], vmax + 3 * qi, amax + 3 * qi, n_obs, obs_traj, nrm, dd,
                                    sfc, flags, &nvv, P, c, &cst, lo, hi, rows);
        int st = orc_qp_solve_n(nv, P, c, cst, lo, hi, rows, nr, x, &cost, &iters, NULL);
        if (sfc_rc) st = 4;
        float *o = out_traj + (size_t)qi * ORC_NV;
        float *stale = stale_traj + (size_t)qi * ORC_NV;
        if (st == 0) {
            /* planar world: the height of every control point is world/z_2d (src/traj_optimizer.cpp:87-90) */
            for (int j = 0; j < ORC_NV; j++) { o[j] = j < nxy ? (float)x[j] : (float)prm->world_z_2d; stale[j] = o[j]; }
            out_cost[qi] = cost;
        } else {
            for (int j = 0; j < ORC_NV; j++) o[j] = stale[j];
#ifdef _OPENMP
#pragma omp atomic write
#endif
            rc = 1;
        }
        out_status[qi] = st;
        if (out_iters) out_iters[qi] = iters;
        free(obs_traj); free(nrm); free(dd); free(flags); free(rows); free(P); free(c); free(lo); free(hi); free(x);
    }
    return rc;
}

/* The two disturbance checks alone (obstaclePredictionCheck + initialTrajPlanningCheck), for callers that need the slack
 * set before the tick itself -- the reference runs them before goalPlanning().  Idempotent: orc_tick_ex repeats them on the
 * same inputs.  own_reset[qi] (optional) = 1 when the agent's own initial trajectory was reset to its current position. */
void orc_disturbance_update(const orc_params *prm, const orc_modes *md, int N, const float *state, const float *prev_traj,
                            int planner_seq, unsigned char *slack_set, int *sfc_init, unsigned char *own_reset)
{
    for (int qi = 0; qi < N; qi++) {
        if (own_reset) own_reset[qi] = 0;
        if (!(md->reset_threshold > 0) || md->planner_mode == 1) continue;
        unsigned char *myset = slack_set + (size_t)qi * N;
        for (int qj = 0; qj < N; qj++) {
            float tr[ORC_NV];
            const float *p = state + 9 * qj;
            if (planner_seq < 2) orc_const_vel_traj(p, p + 3, prm->dt, tr);
            else orc_shift_traj(prev_traj + (size_t)qj * ORC_NV, tr);
            float p0[3] = {tr[0], tr[ORC_SEGV], tr[2 * ORC_SEGV]};
            if (f32_dist3(p0, p) > md->reset_threshold) {
                if (qj != qi) myset[qj] = 1;
                else {
                    for (int q = 0; q < N; q++) if (q != qi) myset[q] = 1;
                    if (sfc_init) sfc_init[qi] = 1;
                    if (own_reset) own_reset[qi] = 1;
                }
            }
        }
    }
}

/* goalPlanningWithPriority on an empty map with a slack set (:547-551: a slack obstacle is "higher priority" by decree and
 * takes no part in the closest-obstacle / retreat rule) and with the agent's own initial trajectory possibly reset
 * (findLOSFreeGoal starts from initial_traj[M-1][n], which is then the current position). */
void orc_goal_prior_based_ex(int N, int qi, const float *state, const float *desired_goal, const float *prev_traj, int planner_seq,
                             double dt, double goal_threshold, double priority_dist_threshold, double goal_radius,
                             const unsigned char *slack_row, int own_reset, float out_goal[3])
{
    const float *pos = state + 9 * qi;
    const float *goal_i = desired_goal + 3 * qi;
    const double dist_to_goal = f32_dist3(pos, goal_i);
    double min_dist_to_obs = 1e9;
    int closest = -1;
    for (int qj = 0; qj < N; qj++) {
        if (qj == qi) continue;
        if (slack_row && slack_row[qj]) continue;
        const float *opos = state + 9 * qj, *ogoal = desired_goal + 3 * qj;
        const double obs_dist_to_goal = f32_dist3(opos, ogoal);
        const double dist_to_obs = f32_dist3(opos, pos);
        if (obs_dist_to_goal < goal_threshold) continue;
        const float *pt = prev_traj + (size_t)qj * ORC_NV;
        float a[3], b[3];
        for (int k = 0; k < 3; k++) {
            float last = pt[k * ORC_SEGV + (ORC_M - 1) * ORC_NC + ORC_N], first = pt[k * ORC_SEGV + ORC_N];
            a[k] = last - first;
            b[k] = first - pos[k];
        }
        const float dp = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
        if (dist_to_goal > goal_threshold && (double)dp > 0) continue;
        if (dist_to_goal < goal_threshold || obs_dist_to_goal < dist_to_goal)
            if (dist_to_obs < min_dist_to_obs) { min_dist_to_obs = dist_to_obs; closest = qj; }
    }
    if (min_dist_to_obs < priority_dist_threshold) {
        const float *opos = state + 9 * closest;
        float dir[3] = {opos[0] - pos[0], opos[1] - pos[1], opos[2] - pos[2]};
        f32_normalize3(dir);
        const double dist_keep = priority_dist_threshold + 0.1;
        for (int k = 0; k < 3; k++) { float s = dir[k] * (float)dist_keep; out_goal[k] = pos[k] - s; }
        return;
    }
    float end[3];
    if (own_reset) { for (int k = 0; k < 3; k++) end[k] = pos[k]; }
    else if (planner_seq < 2) {
        float tmp[ORC_NV];
        orc_const_vel_traj(pos, pos + 3, dt, tmp);
        for (int k = 0; k < 3; k++) end[k] = tmp[k * ORC_SEGV + (ORC_M - 1) * ORC_NC + ORC_N];
    } else {
        const float *pt = prev_traj + (size_t)qi * ORC_NV;
        for (int k = 0; k < 3; k++) end[k] = pt[k * ORC_SEGV + (ORC_M - 1) * ORC_NC + ORC_N];
    }
    float delta[3] = {goal_i[0] - end[0], goal_i[1] - end[1], goal_i[2] - end[2]};
    float n2 = delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2];
    if (sqrt((double)n2) > goal_radius) {
        f32_normalize3(delta);
        for (int k = 0; k < 3; k++) { float s = delta[k] * (float)goal_radius; out_goal[k] = end[k] + s; }
    } else {
        for (int k = 0; k < 3; k++) out_goal[k] = goal_i[k];
    }
}
