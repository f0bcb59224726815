// lsc_host.hpp -- C++ host side above the C ABI: the reference's own class surface, headless (no ROS).
//
//   Mission / Param           src/mission.cpp:20-132, src/param.cpp:4-144 (only what reaches the hot path)
//   TrajPlanner               include/traj_planner.hpp:51-101  (per-agent view onto the batched GPU context)
//   MultiSyncSimulator        src/multi_sync_simulator.cpp: run :83-147, update :190-318, plan :320-337,
//                             isFinished :358-380, savePlanningResult :408-510, CSV writers :513-633
// The per-agent objects keep their names and argument meaning, but TrajPlanner::plan() does not compute: the
// simulator plans ALL agents with one lsc_replan_tick() (the inputs are frozen by update() before anybody plans).
#pragma once
#include <cmath>
#include <cstdio>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <random>
#include <string>
#include <vector>

#include "../../../include/lsc_planner_amd.h"

namespace DynamicPlanning {

constexpr double SP_EPSILON = 1e-9, SP_EPSILON_FLOAT = 1e-5, SP_INFINITY = 1e9;

// ---------------------------------------------------------------------------------------------- tiny JSON reader
// (the reference vendors rapidjson; the mission files only need objects, arrays, numbers and strings)
struct Json {
    enum Kind { Null, Num, Str, Arr, Obj } kind = Null;
    double num = 0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;
    const Json &operator[](const std::string &k) const {
        for (auto &kv : obj) if (kv.first == k) return kv.second;
        throw std::invalid_argument("[Mission] missing key " + k);
    }
    bool has(const std::string &k) const { for (auto &kv : obj) if (kv.first == k) return true; return false; }
    const Json &operator[](size_t i) const { return arr.at(i); }
};

class JsonParser {
  public:
    explicit JsonParser(const std::string &s) : s_(s) {}
    Json parse() { Json j = value(); ws(); return j; }
  private:
    const std::string &s_;
    size_t p_ = 0;
    void ws() { while (p_ < s_.size() && (isspace((unsigned char)s_[p_]))) p_++; }
    Json value() {
        ws();
        if (p_ >= s_.size()) throw std::invalid_argument("[Mission] unexpected end of JSON");
        char c = s_[p_];
        Json j;
        if (c == '{') {
            j.kind = Json::Obj; p_++; ws();
            if (s_[p_] == '}') { p_++; return j; }
            for (;;) {
                ws(); Json k = value(); ws();
                if (s_[p_] != ':') throw std::invalid_argument("[Mission] JSON: ':' expected");
                p_++;
                j.obj.emplace_back(k.str, value()); ws();
                if (s_[p_] == ',') { p_++; continue; }
                if (s_[p_] == '}') { p_++; break; }
                throw std::invalid_argument("[Mission] JSON: ',' or '}' expected");
            }
        } else if (c == '[') {
            j.kind = Json::Arr; p_++; ws();
            if (s_[p_] == ']') { p_++; return j; }
            for (;;) {
                j.arr.push_back(value()); ws();
                if (s_[p_] == ',') { p_++; continue; }
                if (s_[p_] == ']') { p_++; break; }
                throw std::invalid_argument("[Mission] JSON: ',' or ']' expected");
            }
        } else if (c == '"') {
            j.kind = Json::Str; p_++;
            while (p_ < s_.size() && s_[p_] != '"') { if (s_[p_] == '\\') p_++; j.str += s_[p_++]; }
            p_++;
        } else {
            size_t e = p_;
            while (e < s_.size() && (isdigit((unsigned char)s_[e]) || strchr("+-.eE", s_[e]))) e++;
            if (e == p_) throw std::invalid_argument("[Mission] JSON: value expected");
            j.kind = Json::Num; j.num = std::stod(s_.substr(p_, e - p_)); p_ = e;
        }
        return j;
    }
};

// ---------------------------------------------------------------------------------------------- data model
struct point3d {   // octomap::point3d: float storage, float arithmetic
    float v[3] = {0, 0, 0};
    point3d() = default;
    point3d(double x, double y, double z) { v[0] = (float)x; v[1] = (float)y; v[2] = (float)z; }
    float x() const { return v[0]; } float y() const { return v[1]; } float z() const { return v[2]; }
    float &operator()(int i) { return v[i]; } float operator()(int i) const { return v[i]; }
    point3d operator-(const point3d &o) const { point3d r; for (int i = 0; i < 3; i++) r.v[i] = v[i] - o.v[i]; return r; }
    point3d operator+(const point3d &o) const { point3d r; for (int i = 0; i < 3; i++) r.v[i] = v[i] + o.v[i]; return r; }
    point3d operator*(float s) const { point3d r; for (int i = 0; i < 3; i++) r.v[i] = v[i] * s; return r; }
    double norm() const { float n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2]; return std::sqrt((double)n2); }
};
typedef std::vector<std::vector<point3d>> traj_t;   // [m][control point]

struct State { point3d position, velocity, acceleration; int planner_seq = 0; };

struct Agent {
    int id = 0, cid = 0;
    State current_state;
    point3d start_position, desired_goal_position, current_goal_position;
    std::vector<double> max_vel, max_acc;
    double radius = 0.15, downwash = 2.0, nominal_velocity = 1.0;
};

enum class PlanningReport { Initialized = 0, INITTRAJGENERATIONFAILED, CONSTRAINTGENERATIONFAILED, QPFAILED, WAITFORROSMSG, SUCCESS };

struct Param {   // defaults = launch/testall_empty.launch
    double dt = 0.2, horizon = 1.0;
    int n = 5, phi = 3;
    double control_input_weight = 0.01, terminal_weight = 1.0;
    double multisim_time_step = 0.2, multisim_record_time_step = 0.1, multisim_reset_threshold = 0.15;
    int multisim_max_planner_iteration = 300;
    double multisim_max_noise = 0.0;   // multisim/max_noise (src/param.cpp:22; 0.0 in testall_*.launch:47, 0.02 in simulation.launch:47): uniform noise on the desired goals
    int on_deadlock = 0;               // lsc_sim --on-deadlock: 0 report + apply the goal noise once, 1 report, 2 ignore (multi_sync_simulator.cpp: checkDeadlock)
    bool phase_stats = false;          // lsc_sim --phase-stats: per-phase PlanningTimeStatistics from the instrumented plan kernel
    unsigned multisim_noise_seed = 0;  // 0 = std::random_device like the reference (src/mission.cpp:387); else reproducible
    bool multisim_save_result = false;
    double goal_threshold = 0.1;
    bool goal_mode_prior_based = true;   // mode/goal (launch/*.launch: prior_based)
    bool world_use_octomap = false;
    double world_resolution = 0.1;
    int world_dimension = 3;           // world/dimension (src/param.cpp:12)
    double world_z_2d = 1.0;           // world/z_2d (src/param.cpp:15): height of every agent when world/dimension is 2
    double grid_resolution = 0.3, grid_margin = 0.2;   // grid/resolution, grid/margin (launch/testall_forest.launch:88-89)
    std::string log_dir = ".";
    int device = 0;
    // agent-sharded multi-GPU: one lsc_sim process per GPU (rank), every rank runs the same (deterministic) bookkeeping
    int world = 1, rank = 0;
    std::string comm_file;   // rank 0 writes the RCCL rendezvous token here, the others read it
    // mode/planner, SlackMode, opt/slack_collision_weight, opt/N_constraint_segments (src/param.cpp:33-48, 72-76)
    int planner_mode = 0;              // 0 lsc, 1 bvc
    int solver = 1;                    // lsc_config.solver: 1 dual active set first (default), 0 interior point alone
    int slack_mode = 0;                // 0 none, 1 dynamical_limit, 2 collision_constraint
    double slack_collision_weight = 100000.0;
    int N_constraint_segments = -1;
    std::string getPlannerModeStr() const { return planner_mode == 1 ? "BVC" : "LSC"; }
    std::string getSlackModeStr() const { return slack_mode == 1 ? "dynamical_limit" : (slack_mode == 2 ? "collision_constraint" : "none"); }
};

class Mission {
  public:
    int qn = 0, on = 0;
    std::vector<Agent> agents;
    point3d world_min, world_max;
    std::string mission_file_name, world_file_name;

    // Mission::addNoise (src/mission.cpp:386-395): desired_goal(k) += U[0,1) * max_noise for k < dimension, float32 draws
    void addNoise(double max_noise, int dimension, unsigned seed = 0) {
        std::random_device rd;
        std::mt19937 gen(seed ? seed : rd());
        std::uniform_real_distribution<float> dis(0, 1);
        for (int qi = 0; qi < qn; qi++) {
            float g[3] = {agents[qi].desired_goal_position.x(), agents[qi].desired_goal_position.y(), agents[qi].desired_goal_position.z()};
            for (int k = 0; k < dimension && k < 3; k++) g[k] += dis(gen) * max_noise;   // float += float * double, as there
            agents[qi].desired_goal_position = point3d(g[0], g[1], g[2]);
        }
    }

    // world_dimension == 2: every start and goal is put at z = world_z_2d (src/mission.cpp:88-112)
    bool initialize(const std::string &mission_file, const std::string &world_file = "", int world_dimension = 3, double world_z_2d = 1.0) {
        mission_file_name = mission_file; world_file_name = world_file;
        std::ifstream ifs(mission_file);
        if (!ifs) throw std::invalid_argument("There is no such mission file " + mission_file + "\n");
        std::stringstream ss; ss << ifs.rdbuf();
        const std::string txt = ss.str();
        Json doc = JsonParser(txt).parse();
        const Json &world = doc["world"];
        if (world.arr.size() != 1) throw std::invalid_argument("[Mission] World must have one element");
        const Json &dim = world[0]["dimension"];
        world_min = point3d(dim[0].num, dim[1].num, dim[2].num);
        world_max = point3d(dim[3].num, dim[4].num, dim[5].num);
        std::map<std::string, Agent> quad;
        for (auto &kv : doc["quadrotors"].obj) {
            Agent q;
            for (int i = 0; i < 3; i++) { q.max_vel.push_back(kv.second["max_vel"][i].num); q.max_acc.push_back(kv.second["max_acc"][i].num); }
            q.radius = kv.second["radius"].num; q.downwash = kv.second["downwash"].num; q.nominal_velocity = kv.second["nominal_velocity"].num;
            quad[kv.first] = q;
        }
        const Json &al = doc["agents"];
        qn = (int)al.arr.size();
        agents.resize(qn);
        for (int qi = 0; qi < qn; qi++) {
            const Json &a = al[qi];
            if (!a.has("type") || !a.has("start") || !a.has("goal")) throw std::invalid_argument("[Mission] Agent must have type, start and goal");
            agents[qi] = quad.at(a["type"].str);
            agents[qi].id = qi;
            agents[qi].cid = a.has("cid") ? (int)a["cid"].num : qi;
            agents[qi].start_position = point3d(a["start"][0].num, a["start"][1].num, world_dimension == 2 ? world_z_2d : a["start"][2].num);
            agents[qi].desired_goal_position = point3d(a["goal"][0].num, a["goal"][1].num, world_dimension == 2 ? world_z_2d : a["goal"][2].num);
            if (a.has("downwash")) agents[qi].downwash = a["downwash"].num;
            if (a.has("nominal_velocity")) agents[qi].nominal_velocity = a["nominal_velocity"].num;
        }
        return true;
    }
};

// ---------------------------------------------------------------------------------------------- polynomial.hpp:9-121
inline int nChoosek(int n, int k) {
    if (k > n) return 0;
    if (k * 2 > n) k = n - k;
    if (k == 0) return 1;
    int r = n;
    for (int i = 2; i <= k; i++) { r *= (n - i + 1); r /= i; }
    return r;
}
inline point3d getPointFromControlPoints(const std::vector<point3d> &cp, double t) {
    const int n = (int)cp.size() - 1;
    double x = 0, y = 0, z = 0;
    for (int i = 0; i <= n; i++) {
        double b = nChoosek(n, i) * std::pow(t, i) * std::pow(1 - t, n - i);
        x += cp[i].x() * b; y += cp[i].y() * b; z += cp[i].z() * b;
    }
    return point3d((float)x, (float)y, (float)z);
}
inline State getStateFromControlPoints(const traj_t &cps, double current_time, int M, int n, double dt) {
    int m = (int)(current_time / dt);
    if (m == M && current_time < M * dt + SP_EPSILON) m = M - 1;
    else if (m >= M) throw std::invalid_argument("[Polynomial] Input of getOdom is out of bound");
    State s;
    const double tl = current_time / dt - m;
    s.position = getPointFromControlPoints(cps[m], tl);
    std::vector<point3d> vp(n), ap(n - 1);
    for (int i = 0; i < n; i++) vp[i] = ((cps[m][i + 1] - cps[m][i]) * (float)n) * (float)std::pow(dt, -1);
    s.velocity = getPointFromControlPoints(vp, tl);
    for (int i = 0; i < n - 1; i++) ap[i] = ((vp[i + 1] - vp[i]) * (float)(n - 1)) * (float)std::pow(dt, -1);
    s.acceleration = getPointFromControlPoints(ap, tl);
    return s;
}

// ---------------------------------------------------------------------------------------------- TrajPlanner facade
class TrajPlanner {
  public:
    TrajPlanner(int agent_id, const Param &p, const Mission &m) : param(p), mission(m) {
        agent = mission.agents[agent_id];
        agent.current_state.position = agent.start_position;
        agent.current_goal_position = agent.desired_goal_position;
        M = (int)((param.horizon + SP_EPSILON) / param.dt);
        n = param.n;
        traj_curr.assign(M, std::vector<point3d>(n + 1));
    }
    // --- surface used by MultiSyncSimulator (include/traj_planner.hpp:51-101) ---
    void setCurrentState(const State &s) { agent.current_state = s; current_state_seq = s.planner_seq; state_updated = true; }
    void setObsPrevTrajs(const std::vector<traj_t> &) { obstacles_updated = true; }   // the GPU context holds the table
    void setStart(const point3d &p) { agent.start_position = p; }
    void setDesiredGoal(const point3d &p) { agent.desired_goal_position = p; }
    State getCurrentStateMsg() const { State s = agent.current_state; s.planner_seq = planner_seq; return s; }
    State getFutureStateMsg(double t) const { State s = getStateFromControlPoints(traj_curr, t, M, n, param.dt); s.planner_seq = planner_seq; return s; }
    traj_t getTraj() const { return traj_curr; }
    double getQPCost() const { return current_qp_cost; }
    PlanningReport getPlanningReport() const { return planning_report; }
    double getPlanningTime() const { return planning_time; }
    point3d getCurrentPosition() const { return agent.current_state.position; }
    point3d getDesiredGoalPosition() const { return agent.desired_goal_position; }
    point3d getCurrentGoalPosition() const { return agent.current_goal_position; }
    int getPlannerSeq() const { return planner_seq; }
    // goalPlanning() itself runs behind the C ABI (lsc_config.goal_mode); the simulator copies the result back into
    // agent.current_goal_position after the tick
    void goalPlanning() {}
    // TrajPlanner::plan's bookkeeping (:99-145); the QP itself was solved by the batched tick
    bool inputsFresh() const { return state_updated && obstacles_updated && current_state_seq == planner_seq; }
    void acceptPlan(const float *traj90, double cost, int status, double seconds) {
        planner_seq++;
        for (int m = 0; m < M; m++)
            for (int i = 0; i <= n; i++)
                traj_curr[m][i] = point3d(traj90[m * (n + 1) + i], traj90[M * (n + 1) + m * (n + 1) + i], traj90[2 * M * (n + 1) + m * (n + 1) + i]);
        if (status == LSC_STATUS_OK) current_qp_cost = cost;
        // QP failures are swallowed: trajOptimization() returns true whatever the solver did, so planLSC reports SUCCESS
        // (src/traj_planner.cpp:1553-1584, :388-420).  A blocked corridor seed is different: expandBoxFromPoint throws
        // std::invalid_argument out of plan() (include/corridor_constructor.hpp:35-38) -- the simulator re-throws it.
        // Status 5 (goal search outgrew its LDS capacity) has no reference counterpart and must not look like a success.
        planning_report = status == LSC_STATUS_SFC_BLOCKED    ? PlanningReport::CONSTRAINTGENERATIONFAILED
                          : status == LSC_STATUS_GOAL_CAPACITY ? PlanningReport::INITTRAJGENERATIONFAILED
                                                               : PlanningReport::SUCCESS;
        last_status = status;
        planning_time = seconds;
        state_updated = obstacles_updated = false;
    }
    Agent agent;
    int last_status = 0;
  private:
    Param param;
    Mission mission;
    int M, n;
    traj_t traj_curr;
    int planner_seq = 0, current_state_seq = 0;
    bool state_updated = false, obstacles_updated = false;
    double current_qp_cost = 0, planning_time = 0;
    PlanningReport planning_report = PlanningReport::Initialized;
};

// MultiSyncReplayer::readCSVFile (src/multi_sync_replayer.cpp:53-114): the result CSV of a run -- 15 columns per agent
// (id, t, position, velocity, acceleration, planning_time, qp_cost, planning_report, size), then 6 per obstacle (obs_id ...) --
// back into per-agent state histories.  The counts come from the header row ("id" / "obs_id" cells), an agent's radius from
// its `size` column, the make span from the last record's time, exactly as the reference's replayer takes them, so that a file
// written by lsc_sim replays in the reference's tooling and the other way round.
struct ReplayHistory {
    int qn = 0, on = 0;
    std::vector<std::vector<State>> agent_state_history;      // [qn][records]
    std::vector<std::vector<point3d>> obstacle_position_history;
    std::vector<double> agent_radius, obstacle_radius;
    std::vector<double> record_time;
    double makeSpan = 0;
};
inline ReplayHistory readResultCSV(const std::string &file_name) {
    std::ifstream file(file_name);
    if (file.fail()) throw std::invalid_argument("[MultiSyncReplayer] invalid csv file, current file name: " + file_name);
    ReplayHistory h;
    const int offset_agent = 15, offset_obs = 6;
    std::string line;
    int row_idx = 0;
    while (std::getline(file, line)) {
        std::vector<std::string> row;
        std::string cell;
        std::stringstream ss(line);
        while (std::getline(ss, cell, ',')) row.push_back(cell);
        if (row.size() < 2) break;
        if (row_idx == 0) {
            for (size_t i = 0; i + 2 < row.size(); i++) {      // (the reference scans row.size() - 2 cells)
                if (row[i] == "id") h.qn++;
                else if (row[i] == "obs_id") h.on++;
            }
            h.agent_state_history.assign(h.qn, {});
            h.obstacle_position_history.assign(h.on, {});
            h.agent_radius.assign(h.qn, 0.0);
            h.obstacle_radius.assign(h.on, 0.0);
        } else {
            if ((int)row.size() < offset_agent * h.qn + offset_obs * h.on)
                throw std::invalid_argument("[MultiSyncReplayer] row " + std::to_string(row_idx) + " of " + file_name + " is short");
            for (int qi = 0; qi < h.qn; qi++) {
                auto v = [&](int c) { return std::stod(row[offset_agent * qi + c]); };
                State s;
                s.position = point3d(v(2), v(3), v(4));
                s.velocity = point3d(v(5), v(6), v(7));
                s.acceleration = point3d(v(8), v(9), v(10));
                h.agent_state_history[qi].push_back(s);
                h.agent_radius[qi] = v(14);
            }
            for (int oi = 0; oi < h.on; oi++) {
                auto v = [&](int c) { return std::stod(row[offset_agent * h.qn + offset_obs * oi + c]); };
                h.obstacle_position_history[oi].push_back(point3d(v(2), v(3), v(4)));
                h.obstacle_radius[oi] = v(5);
            }
            h.makeSpan = std::stod(row[1]);
            h.record_time.push_back(h.makeSpan);
        }
        row_idx++;
    }
    return h;
}

}  // namespace DynamicPlanning
