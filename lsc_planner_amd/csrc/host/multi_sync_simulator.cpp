// multi_sync_simulator.cpp -- headless MultiSyncSimulator (src/multi_sync_simulator.cpp) over the C ABI.
//   lsc_sim --mission m.json [--mission m2.json ...] [--mission-dir DIR] [--world map.bt] [--max-iter 300] [--csv DIR] [--device 0] [--quiet]
//           [--ranks W --rank R --comm-file PATH]     one process per GPU; or RANK / WORLD_SIZE / LOCAL_RANK from the env
//           [--solver active_set|interior_point]      QP solver of the fast path (lsc_config.solver).  The active-set solve returns the exact
//                                                     optimum: a PERFECTLY symmetric mission (multi_simple4, an unperturbed circle) then stays
//                                                     symmetric and can tie in the priority rule for good -- the reference's remedy is
//                                                     multisim/max_noise (0.02 in launch/simulation.launch): --max-noise 0.02
//           [--on-deadlock noise|report|ignore]       what to do when no agent's horizon end point has moved for 20 ticks while goals
//                                                     are unmet (the reference's own, unused, bookkeeping: src/traj_planner.cpp:396-409).
//                                                     noise (default): say so and apply the reference's remedy once -- goal noise of
//                                                     max(--max-noise, 0.02), Mission::addNoise -- so that a run with the launch files'
//                                                     max_noise 0 still ends; report: say so and fly on; ignore: the reference's silence
// Loop: isFinished -> doStep -> update (ideal next state of every agent) -> plan (one lsc_replan_tick for the
// swarm) -> savePlanningResult (safety ratio / collision accounting) -> optional result / summary CSV in the
// reference's column layout, so that its replayer can read our runs.
#include <chrono>
#include <cstring>
#include <iostream>
#include <thread>

#include <dirent.h>
#include <algorithm>
#include "lsc_host.hpp"

namespace DynamicPlanning {

class MultiSyncSimulator {
  public:
    MultiSyncSimulator(const Param &p, const Mission &m, int mission_index = 0) : param(p), mission(m), mission_index(mission_index) {
        for (int qi = 0; qi < mission.qn; qi++) agents.emplace_back(new TrajPlanner(qi, param, mission));
        lsc_config cfg;
        lsc_default_config(&cfg);
        cfg.dt = param.dt; cfg.control_weight = param.control_input_weight; cfg.terminal_weight = param.terminal_weight;
        cfg.horizon = param.horizon;
        if (param.planner_mode == 0 && param.slack_mode != 0) {   // TrajPlanner::checkPlannerMode, src/traj_planner.cpp:445-448
            std::fprintf(stderr, "[TrajPlanner] LSC does not need slack variables, fix to none\n");
            param.slack_mode = 0;
        }
        cfg.planner_mode = param.planner_mode; cfg.slack_mode = param.slack_mode; cfg.solver = param.solver;
        cfg.slack_collision_weight = param.slack_collision_weight; cfg.n_constraint_segments = param.N_constraint_segments;
        cfg.reset_threshold = param.multisim_reset_threshold;   // the disturbance checks of every shipped launch file (0.15)
        cfg.world_dimension = param.world_dimension; cfg.world_z_2d = param.world_z_2d;
        for (int k = 0; k < 3; k++) { cfg.world_min[k] = mission.world_min(k); cfg.world_max[k] = mission.world_max(k); }
        cfg.use_octomap = param.world_use_octomap; cfg.world_resolution = param.world_resolution; cfg.device = param.device;
        // mode/goal: prior_based like every shipped launch file (on octomap worlds that includes the grid search)
        cfg.goal_mode = param.goal_mode_prior_based ? 1 : 0;
        cfg.goal_threshold = param.goal_threshold;
        cfg.grid_resolution = param.grid_resolution; cfg.grid_margin = param.grid_margin;
        ctx = lsc_create(&cfg);
        if (!ctx) throw std::runtime_error("[MultiSyncSimulator] lsc_create failed: no usable MI355X (there is no CPU path)");
        if (param.phase_stats) {
            if (param.multisim_reset_threshold > 0 || param.planner_mode != 0)
                std::fprintf(stderr, "[MultiSyncSimulator] --phase-stats: the instrumented plan kernel has no alternate-mode hooks; use it with "
                                     "--reset-threshold 0 --planner lsc (the phase columns stay 0 otherwise)\n");
            phase_stats_on = true;                 // switched on after lsc_set_agents (the counters belong to the swarm)
        }
        if (param.world > 1 || !param.comm_file.empty()) initComm();
        const int N = mission.qn;
        std::vector<double> r(N), dw(N), vm(3 * N), am(3 * N), vn(N);
        for (int qi = 0; qi < N; qi++) {
            const Agent &a = mission.agents[qi];
            r[qi] = a.radius; dw[qi] = a.downwash; vn[qi] = a.nominal_velocity;
            for (int k = 0; k < 3; k++) { vm[3 * qi + k] = a.max_vel[k]; am[3 * qi + k] = a.max_acc[k]; }
        }
        check(lsc_set_agents(ctx, N, r.data(), dw.data(), vm.data(), am.data(), vn.data()));
        check(lsc_set_timing(ctx, 1));   // per-kernel device times -> the per-phase columns of the summary
        if (phase_stats_on) check(lsc_phase_profile(ctx, 1, nullptr));
        if (param.world_use_octomap) setOctomap(mission.world_file_name);
        h_state.resize(9 * N); h_goal.resize(3 * N); h_prev.assign((size_t)LSC_NV * N, 0.f); h_next.resize((size_t)LSC_NV * N);
        h_cost.assign(N, 0.0); h_status.assign(N, 0); h_iters.assign(N, 0);
        points.resize(N);
        file_name_param = param.getPlannerModeStr() + "_" + std::to_string(N) + "agents";
    }
    ~MultiSyncSimulator() { lsc_destroy(ctx); }

    // Rendezvous of the native RCCL communicator: rank 0 makes the token and publishes it through a file (one node, one
    // file system); lsc_comm_init is collective and must precede lsc_set_agents.
    void initComm() {
        if (param.comm_file.empty()) throw std::invalid_argument("[MultiSyncSimulator] --ranks needs --comm-file");
        // One token per mission of a mission list (every mission builds its own context and communicator): a rank that is ahead must
        // never read the token of the mission before.  The first mission keeps the plain name.
        const std::string token = mission_index == 0 ? param.comm_file : param.comm_file + "." + std::to_string(mission_index);
        unsigned char id[LSC_COMM_ID_BYTES];
        if (param.rank == 0) {
            check(lsc_comm_unique_id(id));
            const std::string tmp = token + ".tmp";
            { std::ofstream f(tmp, std::ios::binary); f.write(reinterpret_cast<const char *>(id), sizeof(id)); }
            std::rename(tmp.c_str(), token.c_str());
        } else {
            for (int tries = 0;; tries++) {
                std::ifstream f(token, std::ios::binary);
                if (f && f.read(reinterpret_cast<char *>(id), sizeof(id))) break;
                if (tries > 600) throw std::runtime_error("[MultiSyncSimulator] no rendezvous token in " + token);
                std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
        }
        check(lsc_comm_init(ctx, param.world, param.rank, id));
        // lsc_comm_init is collective: when it returns on rank 0 every rank has read the token.  Remove it, so that a later run with
        // the same --comm-file (or the next mission of a list) cannot pick up a stale one.
        if (param.rank == 0) std::remove(token.c_str());
        sharded = true;
    }

    // src/multi_sync_simulator.cpp:153-167
    void setOctomap(const std::string &file) {
        float *edt = nullptr; int dims[3], kmin[3]; double res;
        const float wmin[3] = {mission.world_min(0), mission.world_min(1), mission.world_min(2)};
        const float wmax[3] = {mission.world_max(0), mission.world_max(1), mission.world_max(2)};
        if (lsc_edt_from_bt(file.c_str(), wmin, wmax, 1.0, &edt, dims, kmin, &res) != LSC_OK)
            throw std::invalid_argument("[MultiSyncSimulator] Fail to read octomap file " + file);
        check(lsc_set_distmap(ctx, edt, dims[0], dims[1], dims[2], kmin, res));
        lsc_free_host(edt);
    }

    // :83-147 (simulation branch)
    void run(bool quiet) {
        for (int iter = 0; iter < param.multisim_max_planner_iteration; iter++) {
            if (isFinished() || iter == param.multisim_max_planner_iteration - 1) { summarizeResult(); break; }
            if (initial_update) { sim_start_time = sim_current_time = param.multisim_time_step; }
            else sim_current_time += param.multisim_time_step;
            update();
            if (!plan()) break;
            if (!quiet && param.rank == 0 && iter % 10 == 0) {
                double worst = 0; int failed = 0;
                for (int qi = 0; qi < mission.qn; qi++) {
                    worst = std::max(worst, (agents[qi]->getCurrentPosition() - mission.agents[qi].desired_goal_position).norm());
                    failed += agents[qi]->last_status != 0;
                }
                std::printf("[MultiSyncSimulator] iter %d t=%.1f max dist to goal %.3f safety ratio %.4f qp failures %d tick %.3f ms\n",
                            iter, sim_current_time - sim_start_time, worst, safety_ratio_agent, failed, last_tick_ms);
            }
        }
    }

    // :190-318 (no tf: ideal states; no dynamic obstacles)
    void update() {
        const int N = mission.qn;
        std::vector<State> next(N);
        for (int qi = 0; qi < N; qi++)
            next[qi] = initial_update ? agents[qi]->getCurrentStateMsg() : agents[qi]->getFutureStateMsg(param.multisim_time_step);
        for (int qi = 0; qi < N; qi++) {
            next[qi].planner_seq = agents[qi]->getPlannerSeq();
            agents[qi]->setCurrentState(next[qi]);
        }
        // obstacle list of agent qi = every other agent's next state + previous trajectory: one shared table
        for (int qi = 0; qi < N; qi++) {
            const traj_t t = agents[qi]->getTraj();
            for (int m = 0; m < LSC_M; m++) for (int i = 0; i < LSC_NC; i++) for (int k = 0; k < 3; k++) h_prev[LSC_NV * qi + LSC_SEGV * k + LSC_NC * m + i] = t[m][i](k);
            agents[qi]->setObsPrevTrajs({});
        }
        initial_update = false;
    }

    // :320-337
    bool plan() {
        const int N = mission.qn;
        for (int qi = 0; qi < N; qi++) {
            if (!agents[qi]->inputsFresh()) return false;
            agents[qi]->goalPlanning();
            const State &s = agents[qi]->agent.current_state;
            for (int k = 0; k < 3; k++) {
                h_state[9 * qi + k] = s.position(k); h_state[9 * qi + 3 + k] = s.velocity(k); h_state[9 * qi + 6 + k] = s.acceleration(k);
                h_goal[3 * qi + k] = agents[qi]->getDesiredGoalPosition()(k);   // goalPlanning() runs behind the ABI (cfg.goal_mode)
            }
        }
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<float> goals(3 * N);
        if (sharded) {
            // every rank plans its block of agents; one RCCL all-gather group brings everybody's results to every rank
            check(lsc_replan_tick_all(ctx, h_state.data(), h_goal.data(), h_prev.data(), agents[0]->getPlannerSeq() + 1, h_next.data(),
                                      h_cost.data(), h_status.data(), h_iters.data(), goals.data()));
        } else {
            check(lsc_replan_tick(ctx, h_state.data(), h_goal.data(), h_prev.data(), agents[0]->getPlannerSeq() + 1, h_next.data(),
                                  h_cost.data(), h_status.data(), h_iters.data(), nullptr, nullptr, nullptr));
            check(lsc_last_goals(ctx, goals.data()));
        }
        last_tick_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        for (int qi = 0; qi < N; qi++) {
            agents[qi]->agent.current_goal_position = point3d(goals[3 * qi], goals[3 * qi + 1], goals[3 * qi + 2]);
            agents[qi]->acceptPlan(h_next.data() + (size_t)LSC_NV * qi, h_cost[qi], h_status[qi], last_tick_ms * 1e-3 / N);
        }
        total_ticks++; total_tick_ms += last_tick_ms;
        checkDeadlock();
        // TrajOptimizer::solve exports the model of a failed solve (log/QPmodel.lp, src/traj_optimizer.cpp:99-102); like there the
        // file is overwritten by every failure, so it holds the last one.  Best effort: a swarm with slack rows is not dumped.
        // The rank that OWNS the first failed agent writes it (only its context holds that agent's corridor and plan inputs; every
        // rank sees all statuses, so they agree on who that is without talking).
        if (!param.log_dir.empty())
            for (int qi = 0; qi < N; qi++)
                if (h_status[qi] == LSC_STATUS_INFEASIBLE) {
                    int ws = 1, rk = 0, shard = N;
                    (void)lsc_comm_info(ctx, &ws, &rk, &shard, nullptr);
                    if (qi / (shard > 0 ? shard : 1) != rk) break;
                    if (lsc_dump_qp(ctx, qi, (param.log_dir + "/QPmodel.lp").c_str()) == LSC_OK)
                        std::fprintf(stderr, "[TrajOptimizer] QP of agent %d failed at tick %d: model written to %s/QPmodel.lp\n", qi,
                                     total_ticks, param.log_dir.c_str());
                    break;
                }
        for (int qi = 0; qi < N; qi++) {
            // what the reference's plan() does with these outcomes: the corridor's exception leaves the simulator
            // (include/corridor_constructor.hpp:35-38); nothing else stops a run (:323-328 only tests QPFAILED, which
            // planLSC never reports)
            if (h_status[qi] == LSC_STATUS_SFC_BLOCKED)
                throw std::invalid_argument("[CorridorConstructor] Invalid initial trajectory. Obstacle invades initial trajectory (agent " +
                                            std::to_string(qi) + ")");
            if (h_status[qi] == LSC_STATUS_GOAL_CAPACITY)
                throw std::runtime_error("[MultiSyncSimulator] goal planner: search grid / OPEN list outgrew the LDS capacity (agent " +
                                         std::to_string(qi) + "); no goal was guessed");
        }
        savePlanningResult();
        if (param.multisim_save_result && param.rank == 0) savePlanningResultAsCSV();
        return true;
    }

    // The reference's deadlock bookkeeping (src/traj_planner.cpp:396-409, "Not used in this work"): an agent whose horizon end point
    // traj_curr[M-1][n] has not moved by SP_EPSILON_FLOAT since the previous plan while it is farther than goal_threshold from its goal.
    // Swarm-level here: NO agent's end point moved for 20 consecutive ticks and somebody's goal is unmet.  With the exact optimum of the
    // active-set solve a perfectly symmetric mission ties in the priority rule (:540-577, strict comparisons) for good -- the reference's
    // remedy is multisim/max_noise (src/mission.cpp:386-395; 0.02 in launch/simulation.launch).  Deterministic on every rank (same plans).
    void checkDeadlock() {
        if (param.on_deadlock == 2) return;
        const int N = mission.qn;
        bool moved = false, unmet = false;
        if (endpoints.empty()) { endpoints.assign(N, point3d(SP_INFINITY, SP_INFINITY, SP_INFINITY)); }
        for (int qi = 0; qi < N; qi++) {
            const traj_t t = agents[qi]->getTraj();
            const point3d e = t[LSC_M - 1][LSC_NC - 1];
            if ((e - endpoints[qi]).norm() >= SP_EPSILON_FLOAT) moved = true;
            if ((e - mission.agents[qi].desired_goal_position).norm() > param.goal_threshold) unmet = true;
            endpoints[qi] = e;
        }
        still_ticks = (!moved && unmet) ? still_ticks + 1 : 0;
        // second sign of the same trap: the same agents' QPs infeasible tick after tick (an agent keeps its stale plan on a failure,
        // src/traj_planner.cpp:1548-1585, and two mirror-image agents that block each other head-on never get out of it)
        std::vector<int> failed;
        for (int qi = 0; qi < N; qi++) if (agents[qi]->last_status == LSC_STATUS_INFEASIBLE) failed.push_back(qi);
        failed_ticks = (!failed.empty() && failed == failed_prev) ? failed_ticks + 1 : 0;
        failed_prev = failed;
        if (failed_ticks == 20 && !failure_reported) {
            failure_reported = true;
            if (param.rank == 0) {
                std::string ids;
                for (int q : failed) ids += (ids.empty() ? "" : ",") + std::to_string(q);
                std::fprintf(stderr, "[MultiSyncSimulator] stuck: the QPs of agents %s have been infeasible for 20 ticks (tick %d); they keep their stale plans "
                                     "(the reference's failure semantics) and will not recover. On a symmetric mission start with the reference's remedy: "
                                     "multisim/max_noise (lsc_sim --max-noise 0.02, launch/simulation.launch:47)\n", ids.c_str(), total_ticks);
            }
        }
        if (still_ticks != 20 || deadlock_reported) return;
        deadlock_reported = true;
        deadlock_tick = total_ticks;
        const double noise = std::max(param.multisim_max_noise, 0.02);
        if (param.rank == 0)
            std::fprintf(stderr, "[MultiSyncSimulator] deadlock: no agent's horizon end point has moved for 20 ticks (tick %d) while goals are unmet. "
                                 "A symmetric mission ties in the priority rule under an exact QP optimum; the reference's remedy is "
                                 "multisim/max_noise (launch/simulation.launch:47: 0.02; lsc_sim --max-noise 0.02)%s\n", total_ticks,
                         param.on_deadlock == 0 ? " -- applying it now (--on-deadlock report|ignore to fly on as is)" : "");
        if (param.on_deadlock != 0) return;
        // Mission::addNoise on the desired goals, seeded (the same draw on every rank); the simulator's and the agents' copies
        mission.addNoise(noise, param.world_dimension, param.multisim_noise_seed ? param.multisim_noise_seed : 20260930u);
        for (int qi = 0; qi < N; qi++) agents[qi]->setDesiredGoal(mission.agents[qi].desired_goal_position);
    }

    // :358-380 (GOTO)
    bool isFinished() {
        for (int qi = 0; qi < mission.qn; qi++)
            if ((agents[qi]->getCurrentPosition() - mission.agents[qi].desired_goal_position).norm() > param.goal_threshold) return false;
        total_flight_time = sim_current_time - sim_start_time;
        return true;
    }

    // :408-510.  The agent-agent accounting -- every pair of agents at every record step, N^2 distance evaluations per tick on
    // the reference's host -- runs on the device (lsc_safety_ratio): every rank looks at its own agents against all others,
    // one all-reduce(min) brings the swarm's safety ratio to every rank.
    void savePlanningResult() {
        const int N = mission.qn;
        std::vector<double> times;
        for (double ft = 0; ft < param.multisim_time_step - SP_EPSILON_FLOAT; ft += param.multisim_record_time_step) {
            times.push_back(ft);
            for (int qi = 0; qi < N; qi++) points[qi].push_back(agents[qi]->getFutureStateMsg(ft).position);
        }
        int first = 0, count = N;
        if (sharded) {
            int world = 1, rank = 0, shard_rows = N, table_rows = N;
            check(lsc_comm_info(ctx, &world, &rank, &shard_rows, &table_rows));
            first = std::min(rank * shard_rows, N);
            count = std::min(shard_rows, N - first);
        }
        const int T = (int)times.size();
        std::vector<double> ratio((size_t)T * std::max(count, 1));
        std::vector<int> partner((size_t)T * std::max(count, 1));
        double tick_min = SP_INFINITY;
        check(lsc_safety_ratio(ctx, times.data(), T, ratio.data(), partner.data(), &tick_min));
        if (tick_min < safety_ratio_agent) safety_ratio_agent = tick_min;
        if (tick_min < 1) is_collided = true;
        for (int ti = 0; ti < T; ti++)
            for (int al = 0; al < count; al++)
                if (ratio[(size_t)ti * count + al] < 1)
                    std::fprintf(stderr, "[MultiSyncSimulator] collision with agents, agent_id: (%d,%d), safety_ratio:%g\n", first + al,
                                 partner[(size_t)ti * count + al], ratio[(size_t)ti * count + al]);
        for (int qi = 0; qi < N; qi++) { N_average++; planning_time_sum += agents[qi]->getPlanningTime(); }
    }

    // :513-587 : 15 columns per agent per record step
    void savePlanningResultAsCSV() {
        const std::string fn = param.log_dir + "/result_" + file_name_param + ".csv";
        std::ofstream csv(fn, sim_current_time == sim_start_time ? std::ios_base::trunc : std::ios_base::app);
        const int N = mission.qn;
        if (sim_current_time == sim_start_time)
            for (int qi = 0; qi < N; qi++) csv << "id,t,px,py,pz,vx,vy,vz,ax,ay,az,planning_time,qp_cost,planning_report,size" << (qi < N - 1 ? "," : "\n");
        double t = sim_current_time - sim_start_time;
        for (double ft = 0; ft < param.multisim_time_step; ft += param.multisim_record_time_step, t += param.multisim_record_time_step)
            for (int qi = 0; qi < N; qi++) {
                const State s = agents[qi]->getFutureStateMsg(ft);
                csv << qi << "," << t << "," << s.position.x() << "," << s.position.y() << "," << s.position.z() << "," << s.velocity.x() << ","
                    << s.velocity.y() << "," << s.velocity.z() << "," << s.acceleration.x() << "," << s.acceleration.y() << "," << s.acceleration.z()
                    << "," << agents[qi]->getPlanningTime() << "," << agents[qi]->getQPCost() << "," << (int)agents[qi]->getPlanningReport() << ","
                    << mission.agents[qi].radius << (qi < N - 1 ? "," : "\n");
            }
    }

    // :671-680
    double getTotalDistance() const {
        double d = 0;
        for (auto &p : points) for (size_t i = 0; i + 1 < p.size(); i++) d += (p[i + 1] - p[i]).norm();
        return d;
    }

    // :382-402 + :589-633 (25 columns; goal / corridor / optimisation times are the device times of the respective launches
    // per agent-plan, the LSC generation is part of the optimisation launch and prediction / initial trajectory are not
    // separate steps on the GPU: written as 0)
    void summarizeResult() {
        total_distance = getTotalDistance();
        if (param.rank != 0) return;
        const double avg = N_average ? planning_time_sum / N_average : 0.0;
        // PlanningTimeStatistics per agent-plan (src/multi_sync_simulator.cpp:589-633) from the launches that exist separately:
        // goal kernel, corridor kernel, plan kernel (LSC generation + QP in one launch); seconds per agent-plan
        auto per_plan = [&](int which) {
            double ms = 0; long n = 0;
            if (lsc_kernel_time_ms(ctx, which, &ms, &n) != LSC_OK || n == 0) return 0.0;
            return ms * 1e-3 / mission.qn;
        };
        const double t_goal = per_plan(3), t_sfc = per_plan(4);
        double t_plan = per_plan(0), t_init = 0, t_lsc = 0;
        if (phase_stats_on && total_ticks > 0) {
            // instrumented plan kernel: 100 MHz ticks of lane 0 per phase and agent.  initial_traj_planning_time <- the set-up phase (own
            // initial trajectory, goal stage, constants; the obstacles' predicted segments are loaded inside the LSC build, so
            // obstacle_prediction_time stays 0), lsc_generation_time <- the LSC build, traj_optimization_time <- the interior point
            std::vector<long long> ph((size_t)mission.qn * 16);
            if (lsc_phase_profile(ctx, -1, ph.data()) == LSC_OK) {
                double s_init = 0, s_lsc = 0, s_qp = 0;
                int planned = 0;
                for (int q = 0; q < mission.qn; q++) {
                    const long long *p = ph.data() + (size_t)q * 16;
                    long long all = 0;
                    for (int k = 0; k < 12; k++) all += p[k];
                    if (all == 0) continue;                                   // not in this rank's shard
                    planned++;
                    s_init += (double)p[0]; s_lsc += (double)p[1];
                    for (int k = 2; k < 12; k++) s_qp += (double)p[k];
                }
                if (planned) {
                    const double per = 1e-8 / ((double)planned * total_ticks);
                    t_init = s_init * per; t_lsc = s_lsc * per; t_plan = s_qp * per;
                }
            }
        }
        std::printf("[MultiSyncSimulator] total flight time: %g\n[MultiSyncSimulator] total distance: %g\n"
                    "[MultiSyncSimulator] planning time per agent: %g\n[MultiSyncSimulator] safety ratio between agent: %g\n"
                    "[MultiSyncSimulator] collided: %d, ticks: %d, mean tick %.3f ms -> %.1f agent-replans/s (host-buffer ABI)\n",
                    total_flight_time, total_distance, avg, safety_ratio_agent, (int)is_collided, total_ticks,
                    total_ticks ? total_tick_ms / total_ticks : 0.0, total_tick_ms > 0 ? mission.qn * total_ticks / (total_tick_ms * 1e-3) : 0.0);
        if (!param.multisim_save_result) return;
        const std::string fn = param.log_dir + "/summary_" + file_name_param + ".csv";
        std::ifstream in(fn);
        const bool header = !in || in.peek() == std::ifstream::traits_type::eof();
        std::ofstream out(fn, std::ios_base::app);
        if (header)
            out << "start_time,total_flight_time,total_flight_distance,is_collided,safety_ratio_agent,average_planning_time,min_planning_time,"
                   "max_planning_time,initial_traj_planning_time,obstacle_prediction_time,goal_planning_time,lsc_generation_time,"
                   "sfc_generation_time,traj_optimization_time,mission_file_name,world_file_name,planner_mode,prediction_mode,"
                   "initial_traj_mode,slack_mode,goal_mode,world_dimension,dt,horizon,N_constraint_segments\n";
        out << sim_start_time << "," << total_flight_time << "," << total_distance << "," << is_collided << "," << safety_ratio_agent << "," << avg
            << "," << avg << "," << avg << "," << t_init << ",0," << t_goal << "," << t_lsc << "," << t_sfc << "," << t_plan << "," << mission.mission_file_name << "," << mission.world_file_name
            // mode strings exactly as the reference's writer produces them, quirks included: "current_posiotion" is its
            // spelling (src/param.cpp:158), and getGoalModeStr() indexes its table with the PLANNER mode (src/param.cpp:168-171),
            // so the goal_mode column reads "static" for LSC and "orca" for BVC whatever mode/goal was
            << "," << param.getPlannerModeStr() << (param.planner_mode == 1 ? ",current_position,current_posiotion," : ",previous_solution,previous_solution,")
            << param.getSlackModeStr() << "," << (param.planner_mode == 1 ? "orca" : "static") << "," << param.world_dimension << "," << param.dt << ","
            << param.horizon << "," << param.N_constraint_segments << "\n";
    }

    bool is_collided = false, phase_stats_on = false;
    double safety_ratio_agent = SP_INFINITY, total_flight_time = 0, total_distance = 0;
    int total_ticks = 0, deadlock_tick = 0;
    double total_tick_ms = 0, last_tick_ms = 0;

  private:
    void check(int rc) { if (rc != LSC_OK) throw std::runtime_error(std::string("[MultiSyncSimulator] ") + lsc_last_error(ctx)); }
    Param param;
    Mission mission;
    std::vector<std::unique_ptr<TrajPlanner>> agents;
    lsc_ctx *ctx = nullptr;
    std::vector<float> h_state, h_goal, h_prev, h_next;
    std::vector<double> h_cost;
    std::vector<int> h_status, h_iters;
    std::vector<std::vector<point3d>> points;
    bool initial_update = true, sharded = false, deadlock_reported = false, failure_reported = false;
    int mission_index = 0, still_ticks = 0, failed_ticks = 0;
    std::vector<int> failed_prev;        // agents whose QP was infeasible in the previous tick
    std::vector<point3d> endpoints;      // horizon end point of every agent's previous plan
    double sim_start_time = 0, sim_current_time = 0, planning_time_sum = 0;
    long N_average = 0;
    std::string file_name_param;
};

}  // namespace DynamicPlanning

int main(int argc, char **argv)
{
    using namespace DynamicPlanning;
    Param param;
    std::string world_file, replay_file;
    std::vector<std::string> mission_files;      // Param::mission_file_names (src/param.cpp:106-122): flown back to back, src/multi_sync_simulator_node.cpp:43-70
    bool quiet = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--mission") mission_files.push_back(next());            // (may be repeated)
        else if (a == "--mission-dir") {
            // every *.json of a directory in name order, like the node's mission list
            const std::string dir = next();
            std::vector<std::string> found;
            if (DIR *d = opendir(dir.c_str())) {
                while (dirent *e = readdir(d)) { const std::string n = e->d_name; if (n.size() > 5 && n.substr(n.size() - 5) == ".json") found.push_back(dir + "/" + n); }
                closedir(d);
            }
            std::sort(found.begin(), found.end());
            if (found.empty()) { std::fprintf(stderr, "lsc_sim: no *.json in %s\n", dir.c_str()); return 2; }
            mission_files.insert(mission_files.end(), found.begin(), found.end());
        }
        else if (a == "--replay") replay_file = next();
        else if (a == "--world") { world_file = next(); param.world_use_octomap = true; }
        else if (a == "--max-iter") param.multisim_max_planner_iteration = std::stoi(next());
        else if (a == "--csv") { param.log_dir = next(); param.multisim_save_result = true; }
        else if (a == "--device") param.device = std::stoi(next());
        else if (a == "--quiet") quiet = true;
        else if (a == "--phase-stats") param.phase_stats = true;
        else if (a == "--solver") { const std::string v = next(); if (v == "active_set") param.solver = 1; else if (v == "interior_point") param.solver = 0; else { std::fprintf(stderr, "lsc_sim: --solver active_set|interior_point\n"); return 2; } }
        else if (a == "--on-deadlock") { const std::string v = next(); if (v == "noise") param.on_deadlock = 0; else if (v == "report") param.on_deadlock = 1; else if (v == "ignore") param.on_deadlock = 2; else { std::fprintf(stderr, "lsc_sim: --on-deadlock noise|report|ignore\n"); return 2; } }
        else if (a == "--static-goal") param.goal_mode_prior_based = false;
        else if (a == "--planner") { const std::string v = next(); param.planner_mode = v == "bvc" ? 1 : 0; }
        else if (a == "--slack") { const std::string v = next(); param.slack_mode = v == "dynamical_limit" ? 1 : (v == "collision_constraint" ? 2 : 0); }
        else if (a == "--constraint-segments") param.N_constraint_segments = std::stoi(next());
        else if (a == "--reset-threshold") param.multisim_reset_threshold = std::stod(next());
        else if (a == "--max-noise") param.multisim_max_noise = std::stod(next());
        else if (a == "--noise-seed") param.multisim_noise_seed = (unsigned)std::stoul(next());
        else if (a == "--dimension") param.world_dimension = std::stoi(next());
        else if (a == "--dt") { param.dt = std::stod(next()); param.multisim_time_step = param.dt; }      // LSC: multisim_time_step must equal the segment time (src/traj_planner.cpp:434-436)
        else if (a == "--horizon") param.horizon = std::stod(next());
        else if (a == "--z-2d") param.world_z_2d = std::stod(next());
        else if (a == "--ranks") param.world = std::stoi(next());
        else if (a == "--rank") param.rank = std::stoi(next());
        else if (a == "--comm-file") param.comm_file = next();
        else { std::fprintf(stderr, "usage: lsc_sim --mission m.json [--mission m2.json ...] [--mission-dir DIR] [--world map.bt] [--max-iter N] [--csv DIR] [--device D] [--static-goal] [--quiet] [--ranks W --rank R --comm-file PATH] [--planner lsc|bvc] [--slack none|dynamical_limit|collision_constraint] [--constraint-segments K] [--reset-threshold T] [--dimension 2|3] [--z-2d Z] [--max-noise X [--noise-seed S]] [--phase-stats] [--solver active_set|interior_point] [--on-deadlock noise|report|ignore] [--dt T --horizon H] | lsc_sim --replay result.csv\n"); return 2; }
    }
    if (!replay_file.empty()) {
        // MultiSyncReplayer (src/multi_sync_replayer.cpp): read a result CSV back -- needs no GPU -- and say what it holds
        try {
            const ReplayHistory h = readResultCSV(replay_file);
            const size_t recs = h.qn ? h.agent_state_history[0].size() : 0;
            double dist = 0;
            for (int qi = 0; qi < h.qn; qi++)
                for (size_t i = 0; i + 1 < recs; i++) dist += (h.agent_state_history[qi][i + 1].position - h.agent_state_history[qi][i].position).norm();
            std::printf("replay: agents %d obstacles %d records %zu makeSpan %.9g total_distance %.9g\n", h.qn, h.on, recs, h.makeSpan, dist);
            for (int qi = 0; qi < h.qn && recs; qi++) {
                const point3d &p = h.agent_state_history[qi][recs - 1].position;
                std::printf("agent %d radius %.9g final %.9g %.9g %.9g\n", qi, h.agent_radius[qi], (double)p.x(), (double)p.y(), (double)p.z());
            }
            return 0;
        } catch (const std::exception &e) {
            std::fprintf(stderr, "%s\n", e.what());
            return 3;
        }
    }
    if (mission_files.empty()) { std::fprintf(stderr, "lsc_sim: --mission (or --mission-dir) is required\n"); return 2; }
    if ((int)((param.horizon + 1e-9) / param.dt) != lsc_segments()) {
        std::fprintf(stderr, "lsc_sim: horizon %g / dt %g = %d segments; this binary is linked with the M = %d library (lsc_sim: M = 5, lsc_sim_m4: M = 4)\n",
                     param.horizon, param.dt, (int)((param.horizon + 1e-9) / param.dt), lsc_segments());
        return 2;
    }
    // torchrun / mpirun style environment: one process per GPU
    if (param.world == 1 && std::getenv("WORLD_SIZE")) {
        param.world = std::atoi(std::getenv("WORLD_SIZE"));
        if (const char *r = std::getenv("RANK")) param.rank = std::atoi(r);
        if (const char *lr = std::getenv("LOCAL_RANK")) param.device = std::atoi(lr);
    }
    // The mission list, one simulator after the other like the reference's node (result CSVs are per swarm size and are rewritten by a
    // later mission of the same size; the summary CSV gets one line per mission).  Independent missions IN FLIGHT TOGETHER are the
    // device-resident form: lsc_tick_device_fused_batch (bench.py --missions K).
    int rc = 0;
    for (size_t mi = 0; mi < mission_files.size(); mi++) {
        try {
            Mission mission;
            mission.initialize(mission_files[mi], world_file, param.world_dimension, param.world_z_2d);
            if (param.multisim_max_noise > 0.0) mission.addNoise(param.multisim_max_noise, param.world_dimension, param.multisim_noise_seed);   // src/mission.cpp:317
            if (mission_files.size() > 1) std::printf("[MultiSyncSimulator] mission %zu of %zu: %s\n", mi + 1, mission_files.size(), mission_files[mi].c_str());
            MultiSyncSimulator sim(param, mission, (int)mi);
            sim.run(quiet);
            if (sim.is_collided) rc = 1;
        } catch (const std::exception &e) {
            std::fprintf(stderr, "%s\n", e.what());
            return 3;
        }
    }
    return rc;
}
