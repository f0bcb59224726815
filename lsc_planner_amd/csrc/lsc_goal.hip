// lsc_goal.hip -- goal planning with a distance field (mode/goal = prior_based on an octomap world):
//   TrajPlanner::goalPlanningWithPriority                                   src/traj_planner.cpp:540-608
//   GridBasedPlanner::plan / updateGridMap / updateGridMission / findLOSFreeGoal / castRay
//                                                                           src/grid_based_planner.cpp:53-433
//   Astar-3D: ISearch::startSearch / findSuccessors / findMin / deleteMin / addOpen   src/Astar-3D/isearch.cpp:46-283
//
// One wave per agent; the whole search state lives in LDS.
//
// What has to be reproduced to return the reference's path and not just *a* shortest path: the reference keeps its
// OPEN list as one std::unordered_map per grid row (index i), remembers one "row minimum" per row, and after every
// pop rescans that row in the container's iteration order keeping the LAST entry among equal (F, g).  Equal-cost
// ties are therefore broken by libstdc++'s hash-table order.  That order is emulated explicitly: a row is an array
// of entries in iteration order; libstdc++ (unique keys, identity hash) inserts a node at the head of its bucket
// when the bucket is not empty and at the head of the whole list otherwise, buckets are contiguous in the list, a
// rehash re-inserts every node in list order under the new bucket count, and the bucket count follows the prime
// policy (1 -> 13 -> 29 -> 59 -> ...; the sequence is read from the real container on the host).  The model of this
// file was validated against std::unordered_map itself (tests/test_goal_planning.py).
//
// Wave-parallel pieces: row scans (find / first-of-bucket / rescan) are 64 entries per step with ballots, the row
// minima are reduced with one lane per row, the line-of-sight tests of the path points run one point per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "lsc_kernels.h"

namespace lsc {

extern __shared__ __align__(16) unsigned char gsm[];   // the workgroup's dynamic LDS

namespace {

constexpr uint32_t KEY_BITS = 17;
constexpr uint32_t KEY_MASK = (1u << KEY_BITS) - 1u;     // C <= 131071 cells; g (steps) in the upper 15 bits
constexpr int G_MAX = (1 << (32 - KEY_BITS)) - 1;
// per-cell byte: bits 0-1 state (0 free, 1 occupied, 2 open, 3 closed), bits 2-4 direction of the move that reached
// the cell (7: none), bits 5-7 g (in steps) modulo 8
constexpr int ST_FREE = 0, ST_OCC = 1, ST_OPEN = 2, ST_CLOSED = 3;
__device__ __forceinline__ uint8_t st_open(int dir, int g) { return (uint8_t)(ST_OPEN | (dir << 2) | ((g & 7) << 5)); }
constexpr int RAY_STACK = 24;


// One wave per workgroup: LDS operations of a wave are performed in program order, so lanes only need the compiler
// not to move or merge LDS accesses across a hand-over point -- a wavefront-scope fence, no s_barrier.
__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// n / d for n < 2^32 with m = floor(2^32 / d): the estimate is at most one too small
__device__ __forceinline__ uint32_t div_magic(uint32_t n, uint32_t d, uint32_t m)
{
    uint32_t q = __umulhi(n, m);
    if (n - q * d >= d) q++;
    return q;
}

// Wave reductions without LDS traffic: four DPP row_shr steps inside each 16-lane row, then the four row results are
// combined through v_readlane (a __shfl_xor butterfly costs a ds_bpermute round trip per step).
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v, int identity) { return __builtin_amdgcn_update_dpp(identity, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v, double identity)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(identity), lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(identity), hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_min_d(double v)
{
    const double id = 1.7976931348623157e308;
    v = fmin(v, dpp_d<0x111>(v, id)); v = fmin(v, dpp_d<0x112>(v, id)); v = fmin(v, dpp_d<0x114>(v, id)); v = fmin(v, dpp_d<0x118>(v, id));
    auto rl = [&](int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); };
    return fmin(fmin(rl(15), rl(31)), fmin(rl(47), rl(63)));
}
__device__ __forceinline__ uint32_t wave_max_u(uint32_t u)
{
    // values stay below 2^31 here, so signed max is unsigned max
    int v = (int)u;
    v = max(v, dpp_i<0x111>(v, 0)); v = max(v, dpp_i<0x112>(v, 0)); v = max(v, dpp_i<0x114>(v, 0)); v = max(v, dpp_i<0x118>(v, 0));
    return (uint32_t)max(max(__builtin_amdgcn_readlane(v, 15), __builtin_amdgcn_readlane(v, 31)),
                         max(__builtin_amdgcn_readlane(v, 47), __builtin_amdgcn_readlane(v, 63)));
}
__device__ __forceinline__ int wave_min_i(int v)
{
    const int id = 0x7fffffff;
    v = min(v, dpp_i<0x111>(v, id)); v = min(v, dpp_i<0x112>(v, id)); v = min(v, dpp_i<0x114>(v, id)); v = min(v, dpp_i<0x118>(v, id));
    return min(min(__builtin_amdgcn_readlane(v, 15), __builtin_amdgcn_readlane(v, 31)),
               min(__builtin_amdgcn_readlane(v, 47), __builtin_amdgcn_readlane(v, 63)));
}

struct Ctx {
    int H, W, A, HW, C, cap, lane;
    uint32_t mW, mHW;                // floor(2^32 / W), floor(2^32 / HW)
    const uint32_t *nb_magic;
    int gi, gj, gz;                  // goal cell
    uint8_t *st;                     // [C]
    double *rowF;                    // [H]  F of the registered row minimum
    uint32_t *rowMin;                // [H]  its entry (key | g << 17)
    uint16_t *rowCnt;                // [H]
    int16_t *rowNb;                  // [H]  index into nb_seq, -1: the fresh container's single bucket
    uint32_t *rowNbv, *rowNbm;       // [H]  the bucket count itself and floor(2^32 / it): one read instead of a dependent pair
    uint32_t *tmp;                   // [cap]
    uint32_t *rows;                  // [H][cap]
    const int *nb_seq;
    int n_nb;
    int err;
};

__device__ __forceinline__ void decode(const Ctx &c, uint32_t key, int &i, int &j, int &z)
{
    z = (int)div_magic(key, (uint32_t)c.HW, c.mHW);
    const uint32_t rem = key - (uint32_t)z * (uint32_t)c.HW;
    i = (int)div_magic(rem, (uint32_t)c.W, c.mW);
    j = (int)(rem - (uint32_t)i * (uint32_t)c.W);
}

// F = g + 1.0f * H, H = linecost * sqrt(di^2 + dj^2 + dz^2), linecost = CN_MC_LINE = 10 (astar.cpp:26-29, isearch.cpp:81)
__device__ __forceinline__ double f_of(const Ctx &c, uint32_t e)
{
#pragma clang fp contract(off)
    int i, j, z;
    decode(c, e & KEY_MASK, i, j, z);
    const int di = c.gi - i, dj = c.gj - j, dz = c.gz - z;
    const double h = 10.0 * sqrt((double)(di * di + dj * dj + dz * dz));
    const double g = 10.0 * (double)(e >> KEY_BITS);
    return g + h;
}

__device__ int row_find(const Ctx &c, const uint32_t *row, int cnt, uint32_t key)
{
    for (int base = 0; base < cnt; base += 64) {
        const int p = base + c.lane;
        const bool m = p < cnt && (row[p] & KEY_MASK) == key;
        const unsigned long long mask = __ballot(m);
        if (mask) return base + __ffsll((long long)mask) - 1;
    }
    return -1;
}

// libstdc++ _M_insert_bucket_begin on the array form: before the first entry of the same bucket, else at the front
__device__ void row_place(const Ctx &c, uint32_t *row, int cnt, uint32_t e, uint32_t nb, uint32_t nbm)
{
    auto bucket = [&](uint32_t k) { return k - div_magic(k, nb, nbm) * nb; };
    const uint32_t b = bucket(e & KEY_MASK);
    int pos = 0;
    for (int base = 0; base < cnt; base += 64) {
        const int p = base + c.lane;
        const bool m = p < cnt && bucket(row[p] & KEY_MASK) == b;
        const unsigned long long mask = __ballot(m);
        if (mask) { pos = base + __ffsll((long long)mask) - 1; break; }
    }
    for (int hi = cnt; hi > pos; hi -= 64) {                 // shift [pos, cnt) one to the right, last chunk first
        const int lo = hi - 64 > pos ? hi - 64 : pos;
        const int p = lo + c.lane;
        const uint32_t v = p < hi ? row[p] : 0u;
        wsync();
        if (p < hi) row[p + 1] = v;
        wsync();
    }
    if (c.lane == 0) row[pos] = e;
    wsync();
}

// insertion of a NEW key (unordered_map::operator[] on a missing key): rehash first when the policy asks for it
__device__ void row_insert(Ctx &c, int i, uint32_t e)
{
    uint32_t *row = c.rows + (size_t)i * c.cap;
    int cnt = c.rowCnt[i];
    int nbi = c.rowNb[i];
    uint32_t nb = nbi < 0 ? 1u : (uint32_t)c.nb_seq[nbi];
    uint32_t nbm = nbi < 0 ? 0u : c.nb_magic[nbi];
    if (cnt + 1 > c.cap) { c.err = 1; return; }
    if ((uint32_t)(cnt + 1) > nb || nbi < 0) {
        nbi++;
        if (nbi >= c.n_nb) { c.err = 1; return; }
        nb = (uint32_t)c.nb_seq[nbi];
        nbm = c.nb_magic[nbi];
        for (int p = c.lane; p < cnt; p += 64) c.tmp[p] = row[p];
        wsync();
        for (int t = 0; t < cnt; t++) row_place(c, row, t, c.tmp[t], nb, nbm);
        if (c.lane == 0) { c.rowNb[i] = (int16_t)nbi; c.rowNbv[i] = nb; c.rowNbm[i] = nbm; }
    }
    row_place(c, row, cnt, e, nb, nbm);
    if (c.lane == 0) c.rowCnt[i] = (uint16_t)(cnt + 1);
    wsync();
}

// Pop bookkeeping of one row in a single sweep: erase `key` (the entries behind it move up by one) and redo
// deleteMin's rescan (isearch.cpp:216-240) on what remains: among the entries with the smallest F, the largest g;
// among those the LAST one in iteration order.
__device__ void row_pop(Ctx &c, int i, uint32_t key)
{
    uint32_t *row = c.rows + (size_t)i * c.cap;
    const int cnt = c.rowCnt[i];
    int pos = -1;
    double bf = 1e300;
    uint32_t bsel = 0, bent = 0;                              // g << 16 | new position, and the entry itself
    for (int base = 0; base < cnt; base += 64) {
        const int p = base + c.lane;
        uint32_t e = p < cnt ? row[p] : 0u;
        if (pos < 0) {
            const unsigned long long m = __ballot(p < cnt && (e & KEY_MASK) == key);
            if (m) pos = base + __ffsll((long long)m) - 1;
        }
        const bool moved = pos >= 0 && p >= pos;               // what sits at position p once `key` is gone
        if (moved) e = p + 1 < cnt ? row[p + 1] : 0u;
        wsync();                                               // every lane has read its successor before anybody overwrites it
        if (moved && p < cnt - 1) row[p] = e;
        if (p < cnt - 1) {
            const double f = f_of(c, e);
            const uint32_t sel = ((e >> KEY_BITS) << 16) | (uint32_t)p;
            if (f < bf || (f == bf && sel >= bsel)) { bf = f; bsel = sel; bent = e; }
        }
        wsync();
    }
    if (cnt > 1) {
        const double fmin = wave_min_d(bf);
        const uint32_t sel = wave_max_u(bf == fmin ? bsel : 0u);
        const unsigned long long own = __ballot(bf == fmin && bsel == sel);
        const uint32_t ent = (uint32_t)__builtin_amdgcn_readlane((int)bent, __ffsll((long long)own) - 1);
        if (c.lane == 0) { c.rowMin[i] = ent; c.rowF[i] = fmin; }
    }
    if (c.lane == 0) {
        c.rowCnt[i] = (uint16_t)(cnt - 1);
        if (cnt == 1) c.rowF[i] = 1e300;                       // empty row: never the minimum
    }
    wsync();
}

// ---- The common cases with one LDS round trip each.  The search is a chain of dependent LDS accesses on a single wave
// (a third of its cycles were LDS latency), so the paths every node takes read everything they need about a row at once:
// RowInfo = (count, bucket count and its magic, registered minimum and its F) is one batch of independent loads, and a
// row of at most 64 entries is read once into registers, searched with a ballot and shifted from registers.
struct RowInfo { int cnt, nbi; uint32_t nb, nbm, min; double F; };
__device__ __forceinline__ RowInfo row_info(const Ctx &c, int i)
{
    RowInfo r;
    r.cnt = c.rowCnt[i]; r.nbi = c.rowNb[i]; r.nb = c.rowNbv[i]; r.nbm = c.rowNbm[i]; r.min = c.rowMin[i]; r.F = c.rowF[i];
    return r;
}

// row_insert for a row whose state is already known; falls back to the general code for a rehash or a long row
__device__ __forceinline__ void row_insert_known(Ctx &c, int i, uint32_t e, const RowInfo &ri)
{
    const int cnt = ri.cnt;
    if (ri.nbi < 0 || (uint32_t)(cnt + 1) > ri.nb || cnt >= 64 || cnt + 1 > c.cap) { row_insert(c, i, e); return; }
    uint32_t *row = c.rows + (size_t)i * c.cap;
    const uint32_t nb = ri.nb, nbm = ri.nbm;
    auto bucket = [&](uint32_t k) { return k - div_magic(k, nb, nbm) * nb; };
    const uint32_t v = c.lane < cnt ? row[c.lane] : 0u;
    const unsigned long long mask = __ballot(c.lane < cnt && bucket(v & KEY_MASK) == bucket(e & KEY_MASK));
    const int pos = mask ? __ffsll((long long)mask) - 1 : 0;
    if (c.lane >= pos && c.lane < cnt) row[c.lane + 1] = v;        // every lane holds its entry: no second read
    if (c.lane == 0) { row[pos] = e; c.rowCnt[i] = (uint16_t)(cnt + 1); }
    wsync();
}

// row_pop for a row of at most 64 entries whose count is known
__device__ __forceinline__ void row_pop_known(Ctx &c, int i, uint32_t key, int cnt)
{
    if (cnt > 64) { row_pop(c, i, key); return; }
    uint32_t *row = c.rows + (size_t)i * c.cap;
    const int p = c.lane;
    uint32_t e = p < cnt ? row[p] : 0u;
    const uint32_t nxt = p + 1 < cnt ? row[p + 1] : 0u;             // issued with the load above: one round trip
    const unsigned long long m = __ballot(p < cnt && (e & KEY_MASK) == key);
    const int pos = __ffsll((long long)m) - 1;
    const bool moved = p >= pos;
    if (moved) e = nxt;
    if (moved && p < cnt - 1) row[p] = e;
    double bf = 1e300;
    uint32_t bsel = 0;
    if (p < cnt - 1) {
#pragma clang fp contract(off)
        // f_of with the row index known: key - W i = HW z + j, one division instead of two
        const uint32_t rem = (e & KEY_MASK) - (uint32_t)(c.W * i);
        const int z = (int)div_magic(rem, (uint32_t)c.HW, c.mHW), j = (int)(rem - (uint32_t)z * (uint32_t)c.HW);
        const int di = c.gi - i, dj = c.gj - j, dz = c.gz - z;
        bf = 10.0 * (double)(e >> KEY_BITS) + 10.0 * sqrt((double)(di * di + dj * dj + dz * dz));
        bsel = ((e >> KEY_BITS) << 16) | (uint32_t)p;
    }
    if (cnt > 1) {
        const double fmin = wave_min_d(bf);
        const uint32_t sel = wave_max_u(bf == fmin ? bsel : 0u);
        const unsigned long long own = __ballot(bf == fmin && bsel == sel);
        const uint32_t ent = (uint32_t)__builtin_amdgcn_readlane((int)e, __ffsll((long long)own) - 1);
        if (c.lane == 0) { c.rowMin[i] = ent; c.rowF[i] = fmin; }
    }
    if (c.lane == 0) {
        c.rowCnt[i] = (uint16_t)(cnt - 1);
        if (cnt == 1) c.rowF[i] = 1e300;                       // empty row: never the minimum
    }
    wsync();
}

// DynamicEDTOctomap::getDistance(point3d)
__device__ __forceinline__ float edt_at(const GoalArgs &a, const float p[3])
{
#pragma clang fp contract(off)
    const int dims[3] = {a.nx, a.ny, a.nz};
    int cc[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        cc[k] = (int)floor(a.rf * (double)p[k]) + 32768 - a.key_min[k];
        if (cc[k] < 0 || cc[k] >= dims[k]) return -1.0f;
    }
    return a.edt[((size_t)cc[0] * a.ny + cc[1]) * a.nz + cc[2]];
}

__device__ __forceinline__ double dist_f32(const float *p, const float *q)
{
#pragma clang fp contract(off)
    const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
    const float n2 = dx * dx + dy * dy + dz * dz;
    return sqrt((double)n2);
}

// castRay (grid_based_planner.cpp:409-433), recursion unrolled onto an explicit stack (pure boolean AND of the leaves)
__device__ bool cast_ray(const GoalArgs &a, const float from[3], const float to[3], double radius, float *stack /*[RAY_STACK][6]*/,
                         int &err)
{
#pragma clang fp contract(off)
    int sp = 0;
    float s[6] = {from[0], from[1], from[2], to[0], to[1], to[2]};
    for (;;) {
        const double d = dist_f32(s, s + 3);
        const double thr = sqrt(0.25 * d * d + radius * radius);
        const double sa = (double)edt_at(a, s), sb = (double)edt_at(a, s + 3);
        if (sa < radius + 0.5 * a.wres - 1e-5) return false;
        if (sb < radius + 0.5 * a.wres - 1e-5) return false;
        if (thr < 1.0 && sa > thr && sb > thr) {
            if (sp == 0) return true;
            sp--;
#pragma unroll
            for (int k = 0; k < 6; k++) s[k] = stack[sp * 6 + k];
            continue;
        }
        float mid[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { const float t = s[k] + s[3 + k]; mid[k] = t * 0.5f; }
        if (sp >= RAY_STACK) { err = 2; return false; }
        stack[sp * 6 + 0] = mid[0]; stack[sp * 6 + 1] = mid[1]; stack[sp * 6 + 2] = mid[2];   // right half, later
        stack[sp * 6 + 3] = s[3]; stack[sp * 6 + 4] = s[4]; stack[sp * 6 + 5] = s[5];
        sp++;
        s[3] = mid[0]; s[4] = mid[1]; s[5] = mid[2];                                             // left half, now
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Register-resident search (round 3).  Same container emulation, same order of operations on every row, far fewer
// instructions per expanded node (the search is a single dependent chain: its cost is its instruction count plus its LDS
// round trips).
//   * the per-row bookkeeping (count, bucket count and its magic, registered minimum and its F) lives in the registers
//     of lane (i mod 64), slot (i / 64), not in LDS: findMin is a lane-local select plus one wave reduction, an update
//     is a predicated move;
//   * an OPEN entry is  j | z << JB | g << 17  (the row index i is implied), so a lane gets (j, z) with two bit-field
//     extracts instead of two divisions; the reference's key  H W z + W i + j  (what the container hashes) is two
//     multiply-adds away;
//   * one batch of LDS loads per node: the popped row (entry p and p + 1 per lane), the cell bytes of the six neighbours
//     and of the popped cell itself (lane 6); the erase, the rescan and the insertions into the same row work on that
//     register copy, rows i - 1 / i + 1 are loaded when something is inserted there; all cell bytes of a node (CLOSED for
//     the popped cell, OPEN + direction + g mod 8 for the unseen neighbours) are one ds_write_b8;
//   * a row is shifted up by one entry with a DPP wave_shr (no LDS traffic).
// Rows longer than 64 entries, rehashes and the (practically never taken) "already OPEN with a larger g" case use the
// chunked LDS routines below -- same results, any length.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct FGeoL {
    int H, W, A, HW, cap, lane;
    int JB;                          // bits of j in an entry
    uint32_t JM;                     // (1 << JB) - 1
    int gi, gj, gz;
    uint8_t *st;
    uint32_t *rows, *tmp;
    const int *nb_seq;
    const uint32_t *nb_magic;
    int n_nb;
};

// the key the reference's container hashes: Node::get_id = H W z + W i + j
__device__ __forceinline__ uint32_t ref_key(const FGeoL &c, uint32_t e, uint32_t Wi)
{
    return (uint32_t)c.HW * ((e & KEY_MASK) >> c.JB) + Wi + (e & c.JM);
}
// k mod nb with m = floor(2^32 / nb): the quotient estimate is at most one too small
__device__ __forceinline__ uint32_t bucket_of(uint32_t k, uint32_t nb, uint32_t m)
{
    const uint32_t r = k - __umulhi(k, m) * nb;
    return min(r, r - nb);
}
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); }

__device__ int f_row_find(int lane, const uint32_t *row, int cnt, uint32_t jz)
{
    for (int base = 0; base < cnt; base += 64) {
        const int p = base + lane;
        const unsigned long long mask = __ballot(p < cnt && (row[p] & KEY_MASK) == jz);
        if (mask) return base + __ffsll((long long)mask) - 1;
    }
    return -1;
}

// _M_insert_bucket_begin on a row of any length held in LDS (chunks of 64)
__device__ void f_row_place(const FGeoL &c, uint32_t *row, int cnt, uint32_t e, uint32_t Wi, uint32_t nb, uint32_t nbm)
{
    auto bucket = [&](uint32_t v) { const uint32_t k = ref_key(c, v, Wi); return k - div_magic(k, nb, nbm) * nb; };
    const uint32_t b = bucket(e);
    int pos = 0;
    for (int base = 0; base < cnt; base += 64) {
        const int p = base + c.lane;
        const bool m = p < cnt && bucket(row[p]) == b;
        const unsigned long long mask = __ballot(m);
        if (mask) { pos = base + __ffsll((long long)mask) - 1; break; }
    }
    for (int hi = cnt; hi > pos; hi -= 64) {
        const int lo = hi - 64 > pos ? hi - 64 : pos;
        const int p = lo + c.lane;
        const uint32_t v = p < hi ? row[p] : 0u;
        wsync();
        if (p < hi) row[p + 1] = v;
        wsync();
    }
    if (c.lane == 0) row[pos] = e;
    wsync();
}

// unordered_map::operator[] on a missing key, row in LDS, bookkeeping passed in and out (uniform values)
__device__ void f_row_insert(const FGeoL &c, int i, uint32_t e, int &cnt, int &nbi, uint32_t &nb, uint32_t &nbm, int &err)
{
    uint32_t *row = c.rows + (size_t)i * c.cap;
    const uint32_t Wi = (uint32_t)(c.W * i);
    if (cnt + 1 > c.cap) { err = 1; return; }
    if ((uint32_t)(cnt + 1) > nb || nbi < 0) {
        nbi++;
        if (nbi >= c.n_nb) { err = 1; return; }
        nb = (uint32_t)uni(c.nb_seq[nbi]);                     // (a value loaded from LDS is divergent to the compiler)
        nbm = (uint32_t)uni((int)c.nb_magic[nbi]);
        for (int p = c.lane; p < cnt; p += 64) c.tmp[p] = row[p];
        wsync();
        for (int t = 0; t < cnt; t++) f_row_place(c, row, t, c.tmp[t], Wi, nb, nbm);
    }
    f_row_place(c, row, cnt, e, Wi, nb, nbm);
    cnt++;
}

// ---- wave primitives of the register-resident search.  One wave alone on its CU issues an instruction every 5-8 cycles
// whatever its kind (measured), so the search costs what its instruction count costs: the reductions are
// 32-bit DPP chains (one fused v_min_u32_dpp per stage) over the two halves of F's bit pattern -- non-negative doubles
// order like their bit patterns -- instead of v_min_f64 on moved copies.
// (the two wait states in front of every stage are the VALU-write -> DPP-read hazard of gfx9; lanes without a DPP source keep
// their own value, so no identity is needed)
__device__ __forceinline__ uint32_t wmin_u32(uint32_t v)
{
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wmax_u32(uint32_t v)
{
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// sqrt of a non-negative integer-valued double < 2^52: the instruction sequence the compiler emits for sqrt(double) without
// its range scaling and class checks -- bit-identical to sqrt() on every integer below 2^22 (checked on the device)
__device__ __forceinline__ double sqrt_int(double x)
{
#pragma clang fp contract(off)
    const double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = r * 0.5;
    const double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g); h = __builtin_fma(h, e, h);
    double d = __builtin_fma(-g, g, x); g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x); g = __builtin_fma(d, h, g);
    return x == 0.0 ? 0.0 : g;
}
// ---- the search key.  F = 10 (g + sqrt(d2)) with g = steps and d2 = squared cell distance to the goal, both integers; only its
// ORDER and its TIES matter to the search (findMin, the rescan, addOpen's "not larger").  Two exact representations:
//   Key64 : the bit pattern of the double the reference computes (non-negative doubles order like their bit patterns);
//   Key32 : ((g + floor sqrt d2) << RB) + rank of frac(sqrt d2) among all d2 of the grid -- a table of one word per d2 in LDS,
//           built on the host (lsc_abi.cpp: build_goal_grid, where the argument is spelled out and the minimal gap between
//           distinct keys is checked): integer part first, fractional part by rank; two keys are equal exactly when the
//           reals are, and then the doubles are too (equal d2, or two perfect squares).  Half the reduction work, no sqrt.
struct Key64 {
    using T = unsigned long long;
    static constexpr T NONE = ~0ull;
    __device__ __forceinline__ T entry(uint32_t steps, int d2) const
    {
#pragma clang fp contract(off)
        return (T)__double_as_longlong(__builtin_fma(10.0, (double)steps, 10.0 * sqrt_int((double)d2)));
    }
    // smallest key over the lanes, then the largest sel among those lanes (both uniform)
    static __device__ __forceinline__ void argmin(T f, uint32_t sel, T &fmin, uint32_t &smax)
    {
        const uint32_t hi = (uint32_t)(f >> 32), lo = (uint32_t)f;
        const uint32_t mh = wmin_u32(hi);
        const bool c1 = hi == mh;
        const uint32_t ml = wmin_u32(c1 ? lo : 0xffffffffu);
        const bool c2 = c1 && lo == ml;
        smax = wmax_u32(c2 ? sel : 0u);
        fmin = ((T)mh << 32) | ml;
    }
    static __device__ __forceinline__ T lane_value(T v, int l)
    {
        return ((T)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    }
};
struct Key32 {
    using T = uint32_t;
    static constexpr T NONE = ~0u;
    const uint32_t *tab;             // [d2] floor(sqrt d2) << rb | rank of the fractional part
    int rb;
    __device__ __forceinline__ T entry(uint32_t steps, int d2) const { return (steps << rb) + tab[d2]; }
    static __device__ __forceinline__ void argmin(T f, uint32_t sel, T &fmin, uint32_t &smax)
    {
        fmin = wmin_u32(f);
        smax = wmax_u32(f == fmin ? sel : 0u);
    }
    static __device__ __forceinline__ T lane_value(T v, int l) { return (T)__builtin_amdgcn_readlane((int)v, l); }
};
// (F = g + H of the reference with g = 10 steps, H = 10 sqrt(d2): 10 steps is exact, so Key64's fused form rounds like the sum)
template <typename T>
__device__ __forceinline__ T *uni_ptr(T *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}

// erase + deleteMin's rescan on a row of any length held in LDS, in the search's key (the general routine of the register-resident
// search: rows beyond 64 entries)
template <typename KP>
__device__ void f_row_pop_keyed(const FGeoL &c, const KP &key, int i, uint32_t jz, int cnt, typename KP::T &Fout, uint32_t &mout)
{
    using FK = typename KP::T;
    uint32_t *row = c.rows + (size_t)i * c.cap;
    const int di = c.gi - i, di2 = di * di;
    int pos = -1;
    FK bf = KP::NONE;
    uint32_t bsel = 0, bent = 0;
    for (int base = 0; base < cnt; base += 64) {
        const int p = base + c.lane;
        uint32_t e = p < cnt ? row[p] : 0u;
        if (pos < 0) {
            const unsigned long long m = __ballot(p < cnt && (e & KEY_MASK) == jz);
            if (m) pos = base + __ffsll((long long)m) - 1;
        }
        const bool moved = pos >= 0 && p >= pos;
        if (moved) e = p + 1 < cnt ? row[p + 1] : 0u;
        wsync();
        if (moved && p < cnt - 1) row[p] = e;
        if (p < cnt - 1) {
            const int dj = c.gj - (int)(e & c.JM), dz = c.gz - (int)((e & KEY_MASK) >> c.JB);
            const FK f = key.entry(e >> KEY_BITS, di2 + dj * dj + dz * dz);
            const uint32_t sel = ((e >> KEY_BITS) << 16) | (uint32_t)p;
            if (f < bf || (f == bf && sel >= bsel)) { bf = f; bsel = sel; bent = e; }
        }
        wsync();
    }
    Fout = KP::NONE; mout = 0;
    if (cnt > 1) {
        uint32_t sel;
        KP::argmin(bf, bsel, Fout, sel);
        const unsigned long long own = __ballot(bf == Fout && bsel == sel);
        mout = (uint32_t)__builtin_amdgcn_readlane((int)bent, __ffsll((long long)own) - 1);
    }
}

// ISearch::startSearch on the register-resident rows.  NS = slots of row bookkeeping per lane (H <= 64 NS).  Out of line on
// purpose: the search loop gets a register allocation of its own (inlined into the kernel, the scalar registers that hold the
// launch arguments across it were spilled into the loop).  Returns expansions | end_key << 32 | found << 52 | err << 56.
struct FGeo {
    int H, W, A, HW, cap;
    int JB;                          // bits of j in an entry
    int gi, gj, gz;
    int s0, s1, s2;                  // start cell
    int n_nb;
    int tab_off, rb;                 // Key32: the key table in LDS and its rank bits
    int st_off, rows_off, tmp_off, nbs_off, nbm_off;     // byte offsets into the workgroup's LDS (a pointer passed through a call loses its address space)
    long long *prof;                 // PROF: [8] counters of this agent
};

template <int NS, bool PROF, bool C32>
__device__ __attribute__((noinline)) unsigned long long search_fast(FGeo gin)
{
#pragma clang fp contract(off)
    using KP = typename std::conditional<C32, Key32, Key64>::type;
    using FK = typename KP::T;
    constexpr FK FK_NONE = KP::NONE;
    const int lane = (int)threadIdx.x;
    // everything uniform arrives in vector registers (calling convention): back to scalars
    const int H = uni(gin.H), W = uni(gin.W), A = uni(gin.A), HW = uni(gin.HW), cap = uni(gin.cap), JB = uni(gin.JB);
    const int gi = uni(gin.gi), gj = uni(gin.gj), gz = uni(gin.gz), n_nb = uni(gin.n_nb);
    const uint32_t JM = (1u << JB) - 1u;
    uint8_t *const st = gsm + uni(gin.st_off);
    uint32_t *const rows = reinterpret_cast<uint32_t *>(gsm + uni(gin.rows_off)), *const tmp = reinterpret_cast<uint32_t *>(gsm + uni(gin.tmp_off));
    const int *const nb_seq = reinterpret_cast<const int *>(gsm + uni(gin.nbs_off));
    const uint32_t *const nb_magic = reinterpret_cast<const uint32_t *>(gsm + uni(gin.nbm_off));
    FGeoL c;
    c.H = H; c.W = W; c.A = A; c.HW = HW; c.cap = cap; c.lane = lane; c.JB = JB; c.JM = JM; c.gi = gi; c.gj = gj; c.gz = gz;
    c.st = st; c.rows = rows; c.tmp = tmp; c.nb_seq = nb_seq; c.nb_magic = nb_magic; c.n_nb = n_nb;
    KP key;
    if constexpr (C32) { key.tab = reinterpret_cast<const uint32_t *>(gsm + uni(gin.tab_off)); key.rb = uni(gin.rb); }
    int err = 0, expansions = 0;
    long long pc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // 0-3 sections; 4/5 cycles / count of general pops; 6/7 of general insertions
    long long tk = 0;
    auto tick = [&](int slot) { if constexpr (PROF) { const long long t = (long long)__builtin_readcyclecounter(); pc[slot] += t - tk; tk = t; } };
    auto done = [&](bool found, uint32_t end_key) {
        if constexpr (PROF) { long long *pr = uni_ptr(gin.prof); if (lane == 0) for (int k = 0; k < 12; k++) pr[4 + k] += pc[k]; }
        return (unsigned long long)(unsigned)expansions | ((unsigned long long)end_key << 32) | ((unsigned long long)(found ? 1 : 0) << 52) |
               ((unsigned long long)err << 56);
    };
    // per-row state in registers: lane l, slot t <-> row l + 64 t.   rCn = count | lim << 10 | (nbi + 1) << 20 with
    // lim = min(bucket count, 64, cap) (0 for a fresh container): the in-register insertion applies while count < lim
    FK rF[NS];
    uint32_t rMn[NS], rCn[NS];
#pragma unroll
    for (int t = 0; t < NS; t++) { rF[t] = FK_NONE; rMn[t] = 0u; rCn[t] = 0u; }
    const uint32_t tabNb = lane < 16 ? (uint32_t)nb_seq[lane & 15] : 0u, tabNbm = lane < 16 ? nb_magic[lane & 15] : 0u;   // bucket counts by index
    auto get_u = [&](const uint32_t (&a)[NS], int i) { uint32_t v = a[0]; if constexpr (NS > 1) { if (i >= 64) v = a[1]; } return (uint32_t)__builtin_amdgcn_readlane((int)v, i & 63); };
    auto get_f = [&](int i) {
        FK v = rF[0];
        if constexpr (NS > 1) { if (i >= 64) v = rF[1]; }
        return KP::lane_value(v, i & 63);
    };
    auto put = [&](int i, FK F, uint32_t mn, uint32_t cn) {
#pragma unroll
        for (int t = 0; t < NS; t++) { const bool m = lane + 64 * t == i; rF[t] = m ? F : rF[t]; rMn[t] = m ? mn : rMn[t]; rCn[t] = m ? cn : rCn[t]; }
    };
    auto pack_cn = [&](int cnt, int nbi, uint32_t nb) {
        const uint32_t lim = nbi < 0 ? 0u : min(min(nb, 64u), (uint32_t)cap);
        return (uint32_t)cnt | (lim << 10) | ((uint32_t)(nbi + 1) << 20);
    };
    // lanes 0..5: one neighbour each, in the order of findSuccessors' nested loops; lane 6: the popped cell itself
    const int l_di = lane == 0 ? -1 : (lane == 5 ? 1 : 0), l_dj = lane == 1 ? -1 : (lane == 4 ? 1 : 0), l_dz = lane == 2 ? -1 : (lane == 3 ? 1 : 0);
    const int l_dkey = HW * l_dz + W * l_di + l_dj;
    const uint32_t l_open = (uint32_t)ST_OPEN | ((uint32_t)lane << 2);             // st_open(lane, .) without g

    {   // the start node, g = 0 (parent code 7: none)
        const int s0 = uni(gin.s0), s1 = uni(gin.s1), s2 = uni(gin.s2);
        int cnt = 0, nbi = -1;
        uint32_t nb = 1u, nbm = 0u;
        const uint32_t e0 = (uint32_t)s1 | ((uint32_t)s2 << JB);
        f_row_insert(c, s0, e0, cnt, nbi, nb, nbm, err);
        if (err) return done(false, 0u);
        if (lane == 0) st[HW * s2 + W * s0 + s1] = st_open(7, 0);
        const int di = gi - s0, dj = gj - s1, dz = gz - s2;
        put(s0, key.entry(0u, di * di + dj * dj + dz * dz), e0, pack_cn(1, nbi, nb));
        wsync();
    }
    auto slow_insert = [&](int i, uint32_t e, int &cnt, int &nbi, uint32_t &nb, uint32_t &nbm) {
        long long t0 = 0;
        if constexpr (PROF) t0 = (long long)__builtin_readcyclecounter();
        f_row_insert(c, i, e, cnt, nbi, nb, nbm, err);
        if constexpr (PROF) { pc[6] += (long long)__builtin_readcyclecounter() - t0; pc[7]++; }
    };
    int nopen = 1;
    if constexpr (PROF) tk = (long long)__builtin_readcyclecounter();
    while (nopen > 0) {
        expansions++;
        // ---- findMin (:181-209): smallest F over the registered row minima, then the largest g, then the LAST row
        FK bf = rF[0];
        uint32_t bsel = ((rMn[0] >> KEY_BITS) << 16) | (uint32_t)lane;
        if constexpr (NS > 1) {
            const uint32_t s1 = ((rMn[1] >> KEY_BITS) << 16) | (uint32_t)(lane + 64);
            const bool take = rF[1] < bf || (rF[1] == bf && s1 >= bsel);
            bf = take ? rF[1] : bf; bsel = take ? s1 : bsel;
        }
        FK fmin;
        uint32_t sel;
        KP::argmin(bf, bsel, fmin, sel);
        const int ci = (int)(sel & 0xffffu), cg = (int)(sel >> 16);
        const uint32_t cjz = get_u(rMn, ci) & KEY_MASK;
        const uint32_t ccn = get_u(rCn, ci);
        const int ccnt = (int)(ccn & 1023u);
        const int cj = (int)(cjz & JM), cz = (int)(cjz >> JB);
        const int ckey = HW * cz + W * ci + cj;
        uint32_t *rowc = rows + ci * cap;
        tick(0);
        // ---- the loads of this node in one batch (unpredicated: entries past the count are never looked at)
        const uint32_t e = rowc[lane], nxt = rowc[lane + 1];
        const int ni = ci + l_di, nj = cj + l_dj, nz = cz + l_dz;
        const bool inb = lane < 7 && (unsigned)ni < (unsigned)H && (unsigned)nj < (unsigned)W && (unsigned)nz < (unsigned)A;
        const int ncell = min(max(ckey + l_dkey, 0), HW * A - 1);
        const uint32_t sv = st[ncell];
        // ---- deleteMin (:211-241): erase the node, rescan what remains of its row
        FK Fc = FK_NONE;
        uint32_t mnc = 0u, R = 0u;
        int cntc = ccnt - 1;
        bool rvalid = ccnt <= 64;
        const int dic = gi - ci, dic2 = dic * dic;
        if (rvalid) {
            const unsigned long long m = __ballot(lane < ccnt && (e & KEY_MASK) == cjz);
            const int pos = __ffsll((long long)m) - 1;
            R = lane >= pos ? nxt : e;
            if (lane < cntc) rowc[lane] = R;                                       // (the lanes in front of pos rewrite their own entry)
            if (cntc > 0) {
                const int dj = gj - (int)(R & JM), dz = gz - (int)((R & KEY_MASK) >> JB);
                const FK f = lane < cntc ? key.entry(R >> KEY_BITS, dic2 + dj * dj + dz * dz) : FK_NONE;
                uint32_t w;
                KP::argmin(f, ((R >> KEY_BITS) << 16) | (uint32_t)lane, Fc, w);
                mnc = (uint32_t)__builtin_amdgcn_readlane((int)R, (int)(w & 63u));
            }
        } else {
            long long t0 = 0;
            if constexpr (PROF) t0 = (long long)__builtin_readcyclecounter();
            f_row_pop_keyed(c, key, ci, cjz, ccnt, Fc, mnc);
            if constexpr (PROF) { pc[4] += (long long)__builtin_readcyclecounter() - t0; pc[5]++; }
        }
        nopen--;
        tick(1);
        if (ci == gi && cj == gj) return done(true, (uint32_t)ckey);     // the altitude is not part of the goal test
        if (cg + 1 > G_MAX) { err = 1; return done(false, 0u); }
        // ---- findSuccessors (:100-141) + addOpen (:243-283)
        const int ng = cg + 1;
        const uint32_t state_l = sv & 3u;
        const uint32_t gdiff = ((sv >> 5) - (uint32_t)ng) & 7u;           // (g_old - g_new) mod 8: 1, 2 -> improvement (see the general search)
        const bool isn = lane < 6 && inb;
        const bool want_new = isn && state_l == ST_FREE;
        const bool want_imp = isn && state_l == ST_OPEN && (gdiff - 1u) < 2u;
        FK fs_l;
        {
            const int ei = gi - ni, ej = gj - nj, ez = gz - nz;
            fs_l = key.entry((uint32_t)ng, ei * ei + ej * ej + ez * ez);
        }
        // cell bytes of the node in one store: the popped cell is CLOSED, unseen neighbours are OPEN
        if ((lane == 6 && inb) || want_new) st[ncell] = (uint8_t)(lane == 6 ? (sv | ST_CLOSED) : (l_open | (((uint32_t)ng & 7u) << 5)));
        unsigned long long todo = __ballot(want_new || want_imp);
        const unsigned long long impm = __ballot(want_imp);
        int ccn_nbi = (int)(ccn >> 20) - 1;
        uint32_t ccn_lim = (ccn >> 10) & 1023u;
        tick(2);
        // in-register insertion (_M_insert_bucket_begin) of `ne` into a row of cnt < lim <= 64 entries held one per lane
        auto ins_reg = [&](uint32_t V, int cnt, int nbi, uint32_t Wi, uint32_t ne, uint32_t *row) {
            const uint32_t nb = (uint32_t)__builtin_amdgcn_readlane((int)tabNb, nbi), nbm = (uint32_t)__builtin_amdgcn_readlane((int)tabNbm, nbi);
            const uint32_t kb = bucket_of(ref_key(c, V, Wi), nb, nbm), eb = bucket_of(ref_key(c, ne, Wi), nb, nbm);
            const unsigned long long mm = __ballot(lane < cnt && kb == eb);
            const int pos = mm ? __ffsll((long long)mm) - 1 : 0;
            const uint32_t up = wave_shr1(V);
            V = lane < pos ? V : (lane == pos ? ne : up);
            if (lane <= cnt) row[lane] = V;
            return V;
        };
        if (__builtin_expect(impm == 0ull && rvalid, 1)) {
            // the usual node: nothing to improve, the popped row in registers.  Row by row, in the reference's order
            // (d = 0: row ci - 1; d = 1..4: row ci; d = 5: row ci + 1 -- the rows are independent containers)
            auto other_row = [&](int d, int ri) {
                const int rj = cj, rz = cz;
                const uint32_t ne = (uint32_t)rj | ((uint32_t)rz << JB) | ((uint32_t)ng << KEY_BITS);
                const FK fs = KP::lane_value(fs_l, d);
                uint32_t *row = rows + ri * cap;
                uint32_t cn = get_u(rCn, ri);
                int cnt = (int)(cn & 1023u);
                const uint32_t lim = (cn >> 10) & 1023u;
                if (__builtin_expect((uint32_t)cnt < lim, 1)) {
                    (void)ins_reg(row[lane], cnt, (int)(cn >> 20) - 1, (uint32_t)(W * ri), ne, row);
                    cn++;
                    cnt++;
                } else {
                    int nbi = (int)(cn >> 20) - 1;
                    uint32_t nb = nbi < 0 ? 1u : (uint32_t)__builtin_amdgcn_readlane((int)tabNb, nbi & 15), nbm = nbi < 0 ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)tabNbm, nbi & 15);
                    slow_insert(ri, ne, cnt, nbi, nb, nbm);
                    cn = pack_cn(cnt, nbi, nb);
                }
                nopen++;
                uint32_t mn = get_u(rMn, ri);
                FK Fr = get_f(ri);
                if (cnt == 1 || fs < Fr || (fs == Fr && ng >= (int)(mn >> KEY_BITS))) { Fr = fs; mn = ne; }
                put(ri, Fr, mn, cn);
            };
            if (todo & 1ull) other_row(0, ci - 1);
            if (err) return done(false, 0u);
            unsigned mid = (unsigned)(todo >> 1) & 15u;
            const uint32_t Wc = (uint32_t)(W * ci);
            while (mid) {
                const int d = __ffs((int)mid);                       // 1..4
                mid &= mid - 1;
                const int rj = cj + __builtin_amdgcn_readlane(l_dj, d), rz = cz + __builtin_amdgcn_readlane(l_dz, d);
                const uint32_t ne = (uint32_t)rj | ((uint32_t)rz << JB) | ((uint32_t)ng << KEY_BITS);
                const FK fs = KP::lane_value(fs_l, d);
                if (__builtin_expect(rvalid && (uint32_t)cntc < ccn_lim, 1)) {
                    R = ins_reg(R, cntc, ccn_nbi, Wc, ne, rowc);
                    cntc++;
                } else {
                    uint32_t nb = ccn_nbi < 0 ? 1u : (uint32_t)__builtin_amdgcn_readlane((int)tabNb, ccn_nbi & 15), nbm = ccn_nbi < 0 ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)tabNbm, ccn_nbi & 15);
                    slow_insert(ci, ne, cntc, ccn_nbi, nb, nbm);
                    if (err) return done(false, 0u);
                    ccn_lim = ccn_nbi < 0 ? 0u : min(min(nb, 64u), (uint32_t)cap);
                    rvalid = cntc <= 64;
                    if (rvalid) R = rowc[lane];
                }
                nopen++;
                if (cntc == 1 || fs < Fc || (fs == Fc && ng >= (int)(mnc >> KEY_BITS))) { Fc = fs; mnc = ne; }
            }
            if (todo & 32ull) other_row(5, ci + 1);
            if (err) return done(false, 0u);
        } else {
            while (todo) {
                const int d = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int ri = ci + __builtin_amdgcn_readlane(l_di, d);
                const int rj = cj + __builtin_amdgcn_readlane(l_dj, d), rz = cz + __builtin_amdgcn_readlane(l_dz, d);
                const uint32_t njz = (uint32_t)rj | ((uint32_t)rz << JB);
                const uint32_t ne = njz | ((uint32_t)ng << KEY_BITS);
                const FK fs = KP::lane_value(fs_l, d);
                const bool same = ri == ci;
                uint32_t *row = rows + ri * cap;
                const uint32_t Wi = (uint32_t)(W * ri);
                int cnt, nbi;
                uint32_t lim, mn;
                FK Fr;
                if (same) { cnt = cntc; nbi = ccn_nbi; lim = ccn_lim; mn = mnc; Fr = Fc; }
                else { const uint32_t cn = get_u(rCn, ri); cnt = (int)(cn & 1023u); lim = (cn >> 10) & 1023u; nbi = (int)(cn >> 20) - 1; mn = get_u(rMn, ri); Fr = get_f(ri); }
                if ((impm >> d) & 1ull) {
                    // already OPEN: keep the better of the two (same cell, same H: "F smaller" is "g smaller")
                    const int p = f_row_find(lane, row, cnt, njz);
                    const uint32_t old = (uint32_t)uni((int)row[p]);
                    if (ng < (int)(old >> KEY_BITS)) {
                        if (lane == 0) { row[p] = ne; st[HW * rz + W * ri + rj] = st_open(d, ng); }
                        wsync();
                        if (same && rvalid) R = lane == p ? ne : R;
                        const bool min_is_this = (mn & KEY_MASK) == njz;
                        const FK fm = min_is_this ? fs : Fr;
                        const int gm = min_is_this ? ng : (int)(mn >> KEY_BITS);
                        if (fs < fm || (fs == fm && ng >= gm)) { Fr = fs; mn = ne; }
                    }
                } else {
                    if ((uint32_t)cnt < lim && (!same || rvalid)) {
                        const uint32_t nb = (uint32_t)__builtin_amdgcn_readlane((int)tabNb, nbi), nbm = (uint32_t)__builtin_amdgcn_readlane((int)tabNbm, nbi);
                        uint32_t V = R;
                        if (!same) V = row[lane];
                        const uint32_t kb = bucket_of(ref_key(c, V, Wi), nb, nbm), eb = bucket_of(ref_key(c, ne, Wi), nb, nbm);
                        const unsigned long long mm = __ballot(lane < cnt && kb == eb);
                        const int pos = mm ? __ffsll((long long)mm) - 1 : 0;
                        const uint32_t up = wave_shr1(V);
                        V = lane < pos ? V : (lane == pos ? ne : up);
                        if (lane <= cnt) row[lane] = V;
                        if (same) R = V;
                        cnt++;
                    } else {
                        uint32_t nb = nbi < 0 ? 1u : (uint32_t)__builtin_amdgcn_readlane((int)tabNb, nbi & 15), nbm = nbi < 0 ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)tabNbm, nbi & 15);
                        slow_insert(ri, ne, cnt, nbi, nb, nbm);
                        if (err) return done(false, 0u);
                        lim = nbi < 0 ? 0u : min(min(nb, 64u), (uint32_t)cap);
                        if (same) {
                            rvalid = cnt <= 64;
                            if (rvalid) R = row[lane];
                            ccn_nbi = nbi; ccn_lim = lim;
                        }
                    }
                    nopen++;
                    if (cnt == 1 || fs < Fr || (fs == Fr && ng >= (int)(mn >> KEY_BITS))) { Fr = fs; mn = ne; }
                }
                const uint32_t cn = (uint32_t)cnt | (lim << 10) | ((uint32_t)(nbi + 1) << 20);
                if (same) { cntc = cnt; Fc = Fr; mnc = mn; }
                else put(ri, Fr, mn, cn);
            }
        }
        put(ci, cntc > 0 ? Fc : FK_NONE, mnc, (uint32_t)cntc | (ccn_lim << 10) | ((uint32_t)(ccn_nbi + 1) << 20));
        wsync();
        tick(3);
    }
    return done(false, 0u);
}


}  // namespace

// NS = 0: the general search; NS > 0: the register-resident search with NS slots of row bookkeeping per lane.  One wave per agent.
// (A cooperative variant -- four waves per agent, the rescan, the same-row and the other-row insertions on different SIMDs, two
// barriers per node -- was built and measured in round 3: it returned the same paths at the same 28.5 ms per tick on the tiled
// forest, because every wave has to repeat findMin and the hand-overs cost what the split saves; commit cda8913)
template <int NS, bool PROF, bool C32>
__global__ __launch_bounds__(64) void lsc_goal_kernel(GoalArgs a)
{
#pragma clang fp contract(off)
    constexpr int NT = 64;
    const int tid = threadIdx.x, lane = tid & 63;
    auto ksync = [&]() { wsync(); };
    const int al = blockIdx.x;
    const int qi = a.first + al;
    const int N = a.N;
    long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // PROF: cycles of prologue, grid set-up, search, path + line of sight; findMin, pop, screening, insertions
    long long tk0 = 0;
    auto ktick = [&](int slot) { if constexpr (PROF) { const long long t = (long long)__builtin_readcyclecounter(); pc[slot] += t - tk0; tk0 = t; } };
    if constexpr (PROF) tk0 = (long long)__builtin_readcyclecounter();
#ifdef LSC_POISON_LDS
    // debugging aid (not built into the product, see lsc_kernels.hip): the workgroup's LDS starts as 0xff bytes
    for (int i = tid; i < a.smem_bytes / 4; i += NT) reinterpret_cast<uint32_t *>(gsm)[i] = 0xffffffffu;
    ksync();
#endif
    int tab_off = 0;
    Ctx c;
    c.H = a.H; c.W = a.W; c.A = a.A; c.HW = a.H * a.W; c.C = a.H * a.W * a.A; c.cap = a.row_cap; c.lane = lane;
    c.nb_seq = a.nb_seq; c.nb_magic = a.nb_magic; c.n_nb = a.n_nb; c.err = 0;
    c.mW = (uint32_t)(0x100000000ull / (uint32_t)c.W); c.mHW = (uint32_t)(0x100000000ull / (uint32_t)c.HW);
    {
        size_t off = 0;
        c.st = gsm; off += ((size_t)c.C + 15) & ~(size_t)15;
        c.rowF = reinterpret_cast<double *>(gsm + off); off += sizeof(double) * (size_t)c.H;
        c.rowMin = reinterpret_cast<uint32_t *>(gsm + off); off += sizeof(uint32_t) * (size_t)c.H;
        c.tmp = reinterpret_cast<uint32_t *>(gsm + off); off += sizeof(uint32_t) * (size_t)c.cap;
        c.rows = reinterpret_cast<uint32_t *>(gsm + off); off += sizeof(uint32_t) * (size_t)c.H * c.cap;
        c.rowCnt = reinterpret_cast<uint16_t *>(gsm + off); off += sizeof(uint16_t) * (size_t)c.H;
        c.rowNb = reinterpret_cast<int16_t *>(gsm + off); off += sizeof(int16_t) * (size_t)c.H;
        off = (off + 3) & ~(size_t)3;
        c.rowNbv = reinterpret_cast<uint32_t *>(gsm + off); off += sizeof(uint32_t) * (size_t)c.H;
        c.rowNbm = reinterpret_cast<uint32_t *>(gsm + off); off += sizeof(uint32_t) * (size_t)c.H;
        int *nbs = reinterpret_cast<int *>(gsm + off); off += sizeof(int) * 16;
        uint32_t *nbm = reinterpret_cast<uint32_t *>(gsm + off); off += sizeof(uint32_t) * 16;
        if (tid < 16) { nbs[tid] = a.nb_seq[tid]; nbm[tid] = a.nb_magic[tid]; }
        c.nb_seq = nbs; c.nb_magic = nbm;
        if constexpr (C32) {                                  // Key32's table: one word per squared distance of the grid
            off = (off + 512 + 15) & ~(size_t)15;             // (behind the slack the unpredicated row loads may touch)
            tab_off = (int)off;
            uint32_t *tab = reinterpret_cast<uint32_t *>(gsm + off);
            for (int i = tid; i < a.fcode_n; i += NT) tab[i] = a.fcode[i];
        }
    }
    const float *pos = a.state + 9 * qi;
    const float *goal_i = a.goal + 3 * qi;
    const double dist_to_goal = dist_f32(pos, goal_i);
    const int cl = (M - 1) * NC + DEG, cf = DEG;

    // disturbance reset (traj_planner.cpp:866-878, 1047-1061): an agent off its plan by more than reset_threshold, now or at
    // any earlier tick (a.ever), is in everybody's slack set -- "higher priority" by decree (:548-551): stamped into the
    // grid, but no candidate for the retreat rule
    auto off_plan = [&](int q) {
        if (!(a.reset_thr > 0.0) || a.planner_seq < 2) return false;
        const float *t = a.traj_prev + (size_t)q * NV + NC;
        const float *s = a.state + 9 * q;
        const float dx = t[0] - s[0], dy = t[SEGV] - s[1], dz = t[2 * SEGV] - s[2];
        const float n2 = dx * dx + dy * dy + dz * dz;
        return sqrt((double)n2) > a.reset_thr;
    };
    const bool checks = a.reset_thr > 0.0 && a.ever != nullptr;
    const bool own_now = checks && off_plan(qi);
    const bool own_slack = checks && (own_now || a.ever[qi] != 0);
    auto in_slack = [&](int qj) { return checks && (own_slack || a.ever[qj] != 0 || off_plan(qj)); };
    // whether obstacle qj has priority over this agent (traj_planner.cpp:547-577); dist_to_obs returned for the retreat rule
    auto has_priority = [&](int qj, double &dist_to_obs) {
        const float *opos = a.state + 9 * qj, *ogoal = a.goal + 3 * qj;
        const double obs_dist_to_goal = dist_f32(opos, ogoal);
        dist_to_obs = dist_f32(opos, pos);
        if (obs_dist_to_goal < a.goal_threshold) return false;
        const float *pt = a.traj_prev + (size_t)qj * NV;
        const float ax = pt[cl] - pt[cf], ay = pt[SEGV + cl] - pt[SEGV + cf], az = pt[2 * SEGV + cl] - pt[2 * SEGV + cf];
        const float bx = pt[cf] - pos[0], by = pt[SEGV + cf] - pos[1], bz = pt[2 * SEGV + cf] - pos[2];
        const float dp = ax * bx + ay * by + az * bz;
        if (dist_to_goal > a.goal_threshold && (double)dp > 0.0) return false;
        return dist_to_goal < a.goal_threshold || obs_dist_to_goal < dist_to_goal;
    };

    // ---- retreat rule (:578-587): the closest higher-priority agent, first strict minimum in obstacle order
    {
        double best = 1e9;
        int bq = 0x7fffffff;
        for (int qj = lane; qj < N; qj += 64) {
            if (qj == qi) continue;
            double d;
            if (in_slack(qj)) continue;
            if (has_priority(qj, d) && d < best) { best = d; bq = qj; }
        }
        const double dmin = wave_min_d(best);
        const int q = wave_min_i(best == dmin ? bq : 0x7fffffff);
        if (dmin < a.priority_dist_threshold) {
            if (tid == 0) {
                const float *opos = a.state + 9 * q;
                float dx = opos[0] - pos[0], dy = opos[1] - pos[1], dz = opos[2] - pos[2];
                const float n2 = dx * dx + dy * dy + dz * dz;
                const double len = sqrt((double)n2);
                if (len > 0) { const float l = (float)len; dx /= l; dy /= l; dz /= l; }
                const float keep = (float)(a.priority_dist_threshold + 0.1);
                a.goal_out[3 * qi] = pos[0] - dx * keep; a.goal_out[3 * qi + 1] = pos[1] - dy * keep; a.goal_out[3 * qi + 2] = pos[2] - dz * keep;
                a.err[qi] = 0;
                if (a.flags) a.flags[qi] = 1;
                if (a.expansions) a.expansions[qi] = 0;
                if (a.path_len) a.path_len[al] = 0;
            }
            return;
        }
    }

    // ---- grid search: first with the higher-priority agents stamped into the grid, then without (:590-600)
    const unsigned char *occ_static = a.occ_static + (size_t)a.img_of_agent[qi] * (size_t)c.C;
    const double r_a = a.radius[qi], dw_a = a.downwash[qi];
    auto cell_of = [&](const float *p, int cc[3]) {          // point3DToGridVector :325-330
        for (int k = 0; k < 3; k++) cc[k] = (int)round(((double)p[k] - a.gmin[k]) / a.gres);
    };
    auto point_of = [&](int i, int j, int z, float p[3]) {   // gridVectorToPoint3D :300-305
        p[0] = (float)(a.gmin[0] + i * a.gres); p[1] = (float)(a.gmin[1] + j * a.gres); p[2] = (float)(a.gmin[2] + z * a.gres);
    };
    auto key_of = [&](int i, int j, int z) { return (uint32_t)(c.HW * z + c.W * i + j); };
    int gcell[3];
    cell_of(goal_i, gcell);
    c.gi = gcell[0]; c.gj = gcell[1]; c.gz = a.dim2 ? 0 : gcell[2];        // planar world: :199-202
    bool found = false;
    uint32_t end_key = 0;
    int flags = 0, expansions = 0;
    ktick(0);
    for (int attempt = 0; attempt < 2 && !found && !c.err; attempt++) {
        if (attempt > 0) ksync();                              // (everybody is done with the first attempt's grid)
        for (int p = tid; p < c.C; p += NT) c.st[p] = occ_static[p];
        if constexpr (NS == 0) for (int i = lane; i < c.H; i += 64) { c.rowCnt[i] = 0; c.rowNb[i] = -1; c.rowNbv[i] = 1u; c.rowNbm[i] = 0u; c.rowF[i] = 1e300; c.rowMin[i] = 0; }
        ksync();
        if (attempt == 0) {
            for (int qj = tid; qj < N; qj += NT) {           // updateGridMap, AGENT branch :163-189
                if (qj == qi) continue;
                double d;
                if (!in_slack(qj) && !has_priority(qj, d)) continue;
                const double r_o = a.radius_obs[qj], dw_o = a.downwash_obs[qj];
                const double px = (double)a.state[9 * qj], py = (double)a.state[9 * qj + 1], pz = (double)a.state[9 * qj + 2];
                const int oi = (int)round((px - a.gmin[0] + 1e-9) / a.gres), oj = (int)round((py - a.gmin[1] + 1e-9) / a.gres),
                          ok = a.dim2 ? 0 : (int)round((pz - a.gmin[2] + 1e-9) / a.gres);    // `obs_k = 0` stays in a planar world
                const int sxy = (int)ceil((r_a + r_o) / a.gres);
                const int sz = (int)ceil((r_a * dw_a + r_o * dw_o) / a.gres);
                const double dwt = (r_a * dw_a + r_o * dw_o) / (r_a + r_o);
                const int i0 = oi - sxy > 0 ? oi - sxy : 0, i1 = oi + sxy < c.H - 1 ? oi + sxy : c.H - 1;
                const int j0 = oj - sxy > 0 ? oj - sxy : 0, j1 = oj + sxy < c.W - 1 ? oj + sxy : c.W - 1;
                const int k0 = ok - sz > 0 ? ok - sz : 0, k1 = ok + sz < c.A - 1 ? ok + sz : c.A - 1;
                for (int i = i0; i <= i1; i++)
                    for (int j = j0; j <= j1; j++)
                        for (int k = k0; k <= k1; k++) {
                            float p[3];
                            point_of(i, j, k, p);
                            const double ex = (double)p[0] - px, ey = (double)p[1] - py, ez = ((double)p[2] - pz) / dwt;
                            const double dist = sqrt(ex * ex + ey * ey + ez * ez);
                            if (dist < r_a + r_o) c.st[key_of(i, j, k)] = ST_OCC;
                        }
            }
            ksync();
        } else {
            flags |= 2;
        }
        // updateGridMission (:193-239): a start cell inside an obstacle moves to the nearest free cell of its 5x5x3 block
        int s[3];
        cell_of(pos, s);
        for (int k = 0; k < 3; k++) {                          // (the reference indexes the grid unchecked here)
            const int hi = (k == 0 ? c.H : (k == 1 ? c.W : c.A)) - 1;
            s[k] = s[k] < 0 ? 0 : (s[k] > hi ? hi : s[k]);
        }
        if ((c.st[key_of(s[0], s[1], s[2])] & 3) == ST_OCC) {
            int best = 1000000000, bc[3] = {s[0], s[1], s[2]};
            for (int i = -2; i < 3; i++)
                for (int j = -2; j < 3; j++)
                    for (int k = -1; k < 2; k++) {
                        const int x = s[0] + i, y = s[1] + j, z = s[2] + k;
                        const bool occd = x < 0 || x > c.H - 1 || y < 0 || y > c.W - 1 || z < 0 || z > c.A - 1 ||
                                          ((c.st[key_of(x, y, z)] & 3) == ST_OCC);
                        if (!occd) {
                            const int dist = abs(i) + abs(j) + abs(k);
                            if (dist < best) { best = dist; bc[0] = x; bc[1] = y; bc[2] = z; }
                        }
                    }
            s[0] = bc[0]; s[1] = bc[1]; s[2] = bc[2];
            ksync();
            if (tid == 0 && (c.st[key_of(s[0], s[1], s[2])] & 3) == ST_OCC) c.st[key_of(s[0], s[1], s[2])] = ST_FREE;
            ksync();
        }
        ktick(1);
        if constexpr (NS > 0) {
            // register-resident search; rows hold (j | z << JB | g << 17)
            FGeo f;
            f.H = c.H; f.W = c.W; f.A = c.A; f.HW = c.HW; f.cap = c.cap; f.JB = a.jbits;
            f.gi = c.gi; f.gj = c.gj; f.gz = c.gz; f.s0 = s[0]; f.s1 = s[1]; f.s2 = s[2]; f.n_nb = c.n_nb;
            f.tab_off = tab_off; f.rb = a.fcode_rb;
            f.st_off = (int)(c.st - gsm); f.rows_off = (int)(reinterpret_cast<unsigned char *>(c.rows) - gsm);
            f.tmp_off = (int)(reinterpret_cast<unsigned char *>(c.tmp) - gsm);
            f.nbs_off = (int)(reinterpret_cast<const unsigned char *>(c.nb_seq) - gsm); f.nbm_off = (int)(reinterpret_cast<const unsigned char *>(c.nb_magic) - gsm);
            f.prof = PROF ? a.prof + (size_t)qi * 16 : nullptr;
            const unsigned long long r = search_fast<NS, PROF, C32>(f);
            expansions += (int)(unsigned)r;
            end_key = (uint32_t)(r >> 32) & KEY_MASK;
            found = ((r >> 52) & 1ull) != 0;
            if ((int)(r >> 56)) c.err = (int)(r >> 56);
            ktick(2);
        } else {
            // ---- ISearch::startSearch
            const uint32_t skey = key_of(s[0], s[1], s[2]);
            row_insert(c, s[0], skey);                            // g = 0
            if (lane == 0) {
                c.st[skey] = st_open(7, 0);                       // parent code 7: none
                c.rowMin[s[0]] = skey;
                c.rowF[s[0]] = f_of(c, skey);
            }
            wsync();
            int nopen = 1;
            while (nopen > 0 && !c.err) {
                expansions++;
                // findMin (:181-209): smallest F over the row minima, then the largest g, then the LAST row
                double bf = 1e300;
                uint32_t bsel = 0, bent = 0;
                int bcnt = 0;
                for (int i = lane; i < c.H; i += 64) {
                    const double f = c.rowF[i];                    // 1e300 while the row is empty
                    const uint32_t me = c.rowMin[i];
                    const int rc = c.rowCnt[i];                    // (same batch of loads: the pop below needs the winner's count)
                    const uint32_t sel = ((me >> KEY_BITS) << 16) | (uint32_t)i;
                    if (f < 1e300 && (f < bf || (f == bf && sel >= bsel))) { bf = f; bsel = sel; bent = me; bcnt = rc; }
                }
                const double fmin = wave_min_d(bf);
                const uint32_t sel = wave_max_u(bf == fmin ? bsel : 0u);
                const int ci = (int)(sel & 0xffffu);
                const unsigned long long owner = __ballot(bf == fmin && bsel == sel);
                const int own_lane = __ffsll((long long)owner) - 1;
                const uint32_t ce = (uint32_t)__builtin_amdgcn_readlane((int)bent, own_lane);
                const int ccnt = __builtin_amdgcn_readlane(bcnt, own_lane);
                const uint32_t ckey = ce & KEY_MASK;
                const int cg = (int)(ce >> KEY_BITS);
                int cj, cz, ci2;
                decode(c, ckey, ci2, cj, cz);
                if (lane == 0) atomicOr(reinterpret_cast<unsigned int *>(c.st + (ckey & ~3u)), (unsigned int)ST_CLOSED << (8u * (ckey & 3u)));
                row_pop_known(c, ci, ckey, ccnt);
                nopen--;
                if (ci == c.gi && cj == c.gj) { found = true; end_key = ckey; break; }   // the altitude is not part of the goal test
                if (cg + 1 > G_MAX) { c.err = 1; break; }
                // findSuccessors (:100-141): the six axis moves in the order of its nested loops.  Lanes 0..5 look at one
                // neighbour each (bounds, occupancy, closed); only the survivors are then handled one after the other.
                int nkey_l = -1;
                uint32_t sv_l = ST_OCC;
                double h_l = 0.0;                                  // H of the neighbour: one vector square root for all six
                if (lane < 6) {
                    const int d = lane;
                    const int di = d == 0 ? -1 : (d == 5 ? 1 : 0), dj = d == 1 ? -1 : (d == 4 ? 1 : 0), dz = d == 2 ? -1 : (d == 3 ? 1 : 0);
                    const int ni = ci + di, nj = cj + dj, nz = cz + dz;
                    if (ni >= 0 && ni < c.H && nj >= 0 && nj < c.W && nz >= 0 && nz < c.A) {
                        nkey_l = (int)key_of(ni, nj, nz);
                        sv_l = c.st[nkey_l];
                    }
                    const int ei = c.gi - ni, ej = c.gj - nj, ez = c.gz - nz;
                    h_l = 10.0 * sqrt((double)(ei * ei + ej * ej + ez * ez));
                }
                // Unseen cells are inserted.  A cell that is already OPEN only matters if the new g is smaller (same cell, same
                // H).  With a consistent heuristic the popped F never decreases, so an OPEN neighbour has
                // g_old >= g_cur - 1 step, and it was reached from a cell adjacent to it, so g_old <= g_cur + 3 steps: the
                // difference g_old - g_new lies in [-2, 2] and its sign can be read from g modulo 8 kept in the cell byte.
                const int ng = cg + 1;
                const int state_l = (int)(sv_l & 3u);
                const int gdiff = (int)(((sv_l >> 5) - (uint32_t)ng) & 7u);          // (g_old - g_new) mod 8: 1, 2 -> improvement
                const bool want = nkey_l >= 0 && (state_l == ST_FREE || (state_l == ST_OPEN && (gdiff == 1 || gdiff == 2)));
                unsigned long long todo = __ballot(want);
                while (todo) {
                    const int d = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const uint32_t nkey = (uint32_t)__builtin_amdgcn_readlane(nkey_l, d);
                    const uint32_t sv = (uint32_t)__builtin_amdgcn_readlane((int)sv_l, d);
                    const int di = d == 0 ? -1 : (d == 5 ? 1 : 0);
                    const int ni = ci + di;
                    const uint32_t ne = nkey | ((uint32_t)ng << KEY_BITS);
                    uint32_t *row = c.rows + (size_t)ni * c.cap;
                    const RowInfo ri = row_info(c, ni);            // everything the insertion and the bookkeeping need, one round trip
                    int cnt_after = ri.cnt;
                    bool inserted = false;
                    uint32_t stored = ne;                          // the row's entry for this key after addOpen
                    if ((sv & 3u) == ST_OPEN) {                    // addOpen (:243-283): keep the better of the two; same cell,
                        const int p = row_find(c, row, ri.cnt, nkey);          // same H, so "F smaller" is "g smaller"
                        const uint32_t old = row[p];
                        stored = old;
                        if (ng < (int)(old >> KEY_BITS)) {
                            if (lane == 0) {
                                row[p] = ne;
                                c.st[nkey] = st_open(d, ng);
                            }
                            stored = ne;
                            inserted = true;
                            wsync();
                        }
                    } else {
                        row_insert_known(c, ni, ne, ri);
                        if (c.err) break;
                        if (lane == 0) c.st[nkey] = st_open(d, ng);
                        inserted = true;
                        nopen++;
                        cnt_after = ri.cnt + 1;
                        wsync();
                    }
                    // row minimum bookkeeping of addOpen (:262-282)
                    const double hs = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(h_l), d), __builtin_amdgcn_readlane(__double2loint(h_l), d));
                    const double fs = 10.0 * (double)(stored >> KEY_BITS) + hs;
                    if (cnt_after == 1) {
                        if (lane == 0) { c.rowMin[ni] = stored; c.rowF[ni] = fs; }
                    } else {
                        const uint32_t me = ri.min;                // (nothing above touches the row's registered minimum)
                        const bool min_is_this = (me & KEY_MASK) == nkey;
                        // the registered minimum is read AFTER the assignment: if it is this very node it already has the new g
                        const double fm = min_is_this ? fs : ri.F;
                        const int gm = min_is_this ? (int)(stored >> KEY_BITS) : (int)(me >> KEY_BITS);
                        if (inserted && (fs < fm || (fs == fm && ng >= gm))) {
                            if (lane == 0) { c.rowMin[ni] = stored; c.rowF[ni] = fs; }
                        } else if (min_is_this && lane == 0) {
                            c.rowMin[ni] = stored; c.rowF[ni] = fs;
                        }
                    }
                    wsync();
                }
            }
        }
    }

    // ---- primary path (makePrimaryPath :143-151): parents back from the popped goal node, stored start -> goal
    uint32_t *path = c.rows;                                   // the OPEN rows are dead now: reuse them
    const int path_cap = c.H * c.cap;
    int n_path = 0;
    if (found && !c.err) {
        if (lane == 0) {
            uint32_t k = end_key;
            int n = 0;
            for (;;) {
                if (n >= path_cap) { n = -1; break; }
                path[path_cap - 1 - n] = k;
                n++;
                const int pd = (c.st[k] >> 2) & 7;
                if (pd == 7) break;
                // the move that reached k was direction pd; step back
                k = pd == 0 ? k + c.W : pd == 5 ? k - c.W : pd == 1 ? k + 1 : pd == 4 ? k - 1 : pd == 2 ? k + c.HW : k - c.HW;
            }
            c.tmp[0] = (uint32_t)n;
        }
        wsync();
        n_path = (int)c.tmp[0];
        if (n_path < 0) { c.err = 1; n_path = 0; }
    }
    const uint32_t *pk = path + (path_cap - n_path);          // pk[0] = start cell ... pk[n_path-1] = goal cell
    if (a.path_out && !c.err) {
        for (int t = lane; t < n_path && t < a.path_cap; t += 64) a.path_out[(size_t)al * a.path_cap + t] = (int)pk[t];
    }

    // ---- findLOSFreeGoal(initial_traj[M-1][n], desired goal) (:350-407): one path point per lane
    float cur[3];
    if (own_now) {
        cur[0] = pos[0]; cur[1] = pos[1]; cur[2] = pos[2];      // initial trajectory reset to the current position
    } else if (a.planner_seq < 2) {
        // initial trajectory = constant-velocity model, its end point is pos + vel * (M-1 + n/n) * dt in float32
        for (int k = 0; k < 3; k++) {
            const float tt = (float)((double)(M - 1) + (double)DEG / (double)DEG);
            cur[k] = pos[k] + pos[3 + k] * tt * a.dtf;
        }
    } else {
        const float *pt = a.traj_prev + (size_t)qi * NV;
        cur[0] = pt[cl]; cur[1] = pt[SEGV + cl]; cur[2] = pt[2 * SEGV + cl];
    }
    float los[3] = {cur[0], cur[1], cur[2]};
    float *stack = a.ray_stack + ((size_t)al * 64 + lane) * RAY_STACK * 6;   // bisection stack of this lane (HBM scratch)
    int rerr = 0;
    for (int it = 0; it < 6 && !c.err; it++) {
        const double margin_ratio = 1.5 - 0.1 * it;
        bool stop = false;
        for (int base = 0; base <= n_path && !stop; base += 64) {
            const int t = base + lane;
            bool safe = true;
            float p[3] = {0.f, 0.f, 0.f};
            if (t <= n_path) {
                if (t < n_path) { int i, j, z; decode(c, pk[t], i, j, z); point_of(i, j, z, p); }
                else { p[0] = goal_i[0]; p[1] = goal_i[1]; p[2] = goal_i[2]; }
                safe = cast_ray(a, cur, p, r_a * margin_ratio, stack, rerr);
            }
            const unsigned long long bad = __ballot(t <= n_path && !safe);
            const int nvalid = n_path + 1 - base < 64 ? n_path + 1 - base : 64;
            const int first_bad = bad ? __ffsll((long long)bad) - 1 : nvalid;
            if (first_bad > 0) {                               // the last safe point before the first unsafe one
                los[0] = __shfl(p[0], first_bad - 1); los[1] = __shfl(p[1], first_bad - 1); los[2] = __shfl(p[2], first_bad - 1);
            }
            if (bad) stop = true;
        }
        if (dist_f32(los, cur) > 0.3) break;
    }
    if (__ballot(rerr != 0)) c.err = 2;
    {
        float dx = los[0] - cur[0], dy = los[1] - cur[1], dz = los[2] - cur[2];
        const float n2 = dx * dx + dy * dy + dz * dz;
        const double len = sqrt((double)n2);
        if (len > a.goal_radius) {
            const float l = (float)len;
            dx /= l; dy /= l; dz /= l;
            const float r = (float)a.goal_radius;
            los[0] = cur[0] + dx * r; los[1] = cur[1] + dy * r; los[2] = cur[2] + dz * r;
        }
    }
    if (lane == 0) {
        a.goal_out[3 * qi] = los[0]; a.goal_out[3 * qi + 1] = los[1]; a.goal_out[3 * qi + 2] = los[2];
        a.err[qi] = c.err;
        if (a.flags) a.flags[qi] = flags;
        if (a.expansions) a.expansions[qi] = expansions;
        if (a.path_len) a.path_len[al] = n_path;
    }
    if constexpr (PROF) {
        ktick(3);
        if (lane == 0) for (int k = 0; k < 4; k++) a.prof[(size_t)qi * 16 + k] += pc[k];
    }
}

size_t goal_smem_bytes(int H, int W, int A, int cap, int key_words)
{
    size_t b = ((size_t)H * W * A + 15) & ~(size_t)15;
    b += sizeof(double) * (size_t)H + sizeof(uint32_t) * (size_t)H + sizeof(uint32_t) * (size_t)cap;
    b += sizeof(uint32_t) * (size_t)H * cap + 2 * sizeof(uint16_t) * (size_t)H;
    b += 4 + 2 * 16 * sizeof(int);                            // bucket-count / magic tables
    b += 512;                                                 // the register-resident search reads 65 entries of a row unpredicated
    b += 16 + sizeof(uint32_t) * (size_t)key_words;           // Key32's table (0 words: Key64)
    b += 2 * sizeof(uint32_t) * (size_t)H;                    // per-row bucket count and magic
    return (b + 15) & ~(size_t)15;
}

// register-resident search: row bookkeeping slots per lane (1: H <= 64, 2: H <= 128) and the bits of j in an OPEN entry;
// 0 when the grid does not fit that layout (more than 128 rows, or (j, z) does not pack into 17 bits)
int goal_fast_slots(int H, int W, int A, int *jbits)
{
    int jb = 0;
    while ((1 << jb) < W) jb++;
    *jbits = jb;
    if (H > 128) return 0;
    if ((((unsigned)(A - 1)) << jb | (unsigned)(W - 1)) > KEY_MASK) return 0;
    return H <= 64 ? 1 : 2;
}

hipError_t init_device_goal_kernel()
{
    const void *k[] = {reinterpret_cast<const void *>(&lsc_goal_kernel<0, false, false>),
                       reinterpret_cast<const void *>(&lsc_goal_kernel<1, false, false>), reinterpret_cast<const void *>(&lsc_goal_kernel<2, false, false>),
                       reinterpret_cast<const void *>(&lsc_goal_kernel<1, true, false>), reinterpret_cast<const void *>(&lsc_goal_kernel<2, true, false>),
                       reinterpret_cast<const void *>(&lsc_goal_kernel<1, false, true>), reinterpret_cast<const void *>(&lsc_goal_kernel<2, false, true>),
                       reinterpret_cast<const void *>(&lsc_goal_kernel<1, true, true>), reinterpret_cast<const void *>(&lsc_goal_kernel<2, true, true>)};
    for (const void *f : k) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_goal(const GoalArgs &a, hipStream_t st)
{
    if (a.count == 0) return hipSuccess;
    const size_t smem = goal_smem_bytes(a.H, a.W, a.A, a.row_cap, (a.variant & 8) ? a.fcode_n : 0);
    GoalArgs t = a;
    t.smem_bytes = (int)smem;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    // variant: 0 the general search (row bookkeeping in LDS, any grid), 1 / 2 the register-resident search (H <= 64 / 128
    // rows and (j, z) packed into 17 bits: goal_fast_slots() says which one a grid admits); a.prof != null selects the
    // instrumented build of the register-resident search (section cycle counters)
    const int slots = a.variant & 3;
    const bool prof = a.prof != nullptr, c32 = (a.variant & 8) != 0 && a.fcode != nullptr;
    const dim3 g(a.count), b(64);
#define LSC_GOAL_LAUNCH(NS_, PR_, C_) hipLaunchKernelGGL((lsc_goal_kernel<NS_, PR_, C_>), g, b, smem, st, t)
    if (slots == 0) LSC_GOAL_LAUNCH(0, false, false);
    else if (slots == 1) { if (c32) { if (prof) LSC_GOAL_LAUNCH(1, true, true); else LSC_GOAL_LAUNCH(1, false, true); }
                           else { if (prof) LSC_GOAL_LAUNCH(1, true, false); else LSC_GOAL_LAUNCH(1, false, false); } }
    else { if (c32) { if (prof) LSC_GOAL_LAUNCH(2, true, true); else LSC_GOAL_LAUNCH(2, false, true); }
           else { if (prof) LSC_GOAL_LAUNCH(2, true, false); else LSC_GOAL_LAUNCH(2, false, false); } }
#undef LSC_GOAL_LAUNCH
    return hipGetLastError();
}

}  // namespace lsc
