// shipsim_kernels.hip — hand-written gfx950 (CDNA4 / MI355X) kernels for the batched ShipEnv hot path.
//
// Design:
//  * One wavefront lane per env, body state as FP64 struct-of-arrays columns in HBM (coalesced 8-byte accesses).
//  * The map bank (river-bank hull planes + goal centres) is staged in LDS once per workgroup with LDS-DMA.
//  * ROLE-SPLIT, PIPELINED workgroups: a workgroup of 4*EPW threads serves EPW envs = EPW/64 tiles of 64 envs, each tile by
//    four waves with one role each — LIDAR-lo and LIDAR-hi (half the beams of LiDAR.query each, and the bank narrowphase
//    against one bank hull each), OBSERVER (sticky-lidar merge, nearest goal, reward / done / flags, the observation rows)
//    and BODY (action, integrator, ship transform, goal narrowphase, statistics, reset; its registers carry the state from
//    step to step).  A lone wave on a SIMD issues FP64 at a quarter of the pipe's rate and runs latency-bound; four co-resident
//    roles keep the SIMD's VALU and the CU's LDS pipe busy.  The roles of a tile exchange the pose, the collision bits and the
//    lidar results through LDS under two per-tile rendezvous words (no workgroup barrier after the one that publishes the
//    staged bank), and run one step apart: while BODY integrates step k+1, OBSERVER writes step k's rows and the LIDAR
//    roles query for step k+1 (see the step kernel's header comment for the timeline).
//  * K steps per launch (ssg_rollout / ssg_rollout_traj): the bank stays in LDS and the state in registers; step k's outputs
//    go to the same rows every step, or to their own slot of a [K][N] trajectory.
//  * Divergent work is made dense: the few lanes whose ship is near a bank / goal, and the (beam, hull) pairs
//    that survive bounding-box culling, are served by whole waves (ballot/readlane broadcast, LDS work queues).
//  * No dense contraction anywhere, so no MFMA.  Arithmetic that advances or judges the state (integrator, forces,
//    transforms, SAT dot products, point queries, the hit point of a beam) follows the reference's operation
//    order, compiled with -ffp-contract=off so every product and sum rounds where Chipmunk's does.
//
// Reference map (file:line under /root/reference):
//   ShipEnv.step                ship_gym/ship_env.py:136-156
//   handle_discrete_action      ship_gym/game.py:140-153      Ship.move_forward/rotate  ship_gym/models.py:129-146
//   LiDAR.query                 ship_gym/models.py:39-76
//   space.step -> cpSpaceStep   ship_gym/game.py:194          
//   collide_ship/collide_goal   ship_gym/game.py:232-257
//   determine_reward/is_done    ship_gym/ship_env.py:62-77,115-134
//   __add_states/closest_goal   ship_gym/ship_env.py:79-113, ship_gym/game.py:333-349
//   ShipEnv.reset/ShipGame.reset ship_gym/ship_env.py:171-184, ship_gym/game.py:260-277
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>
#include <type_traits>

#include "shipsim.h"
#include "shipsim_internal.h"

namespace ssg {

// ---------------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double *lds_base()
{
    extern __shared__ double s_lds[];
    return s_lds;
}

// cpfmin / cpfmax ((a < b) ? a : b): one v_min_f64 / v_max_f64; identical for the non-NaN values on this path
__device__ __forceinline__ double dmin(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ double dmax(double a, double b) { return __builtin_fmax(a, b); }

// The library is compiled with -ffp-contract=off: every product and sum of the state-advancing arithmetic rounds where
// Chipmunk's does.  obs_fma(a, b, c) = a * b + c as ONE fused multiply-add is used only in functions whose results reach
// OBSERVATIONS (lidar readings, the nearest goal's distance comparison) and never the state an env is stepped from: one
// instruction where the state-advancing code issues a multiply and an add (round 6: 5.34 -> 5.23 us per fused step, interleaved on
// one box; readings move by rounding only, ~1e-13, hit decisions unchanged over the parity soak).  Written out by hand, NOT left to
// `#pragma clang fp contract(fast)`: the compiler then fuses different pairs in different instantiations of the kernel, and the
// staged / gathered / 64- / 128- / 256-env layouts, fused and single-step launches, shards and the unsplit batch stop agreeing
// bit for bit (tests/test_parity_gpu.py::test_ragged_sizes_and_bank_in_global caught exactly that).
// -DSSG_NO_LIDAR_FMA (tools/build_variant.sh) builds with the multiply and the add kept apart.
__device__ __forceinline__ double obs_fma(double a, double b, double c)
{
#ifndef SSG_NO_LIDAR_FMA
    return __builtin_fma(a, b, c);
#else
    return a * b + c;
#endif
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane) // src_lane must be wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// cpvforangle(a) = (cos a, sin a).  Deliberately NOT inlined: inside the fused K-step loop the compiler otherwise hoists
// the polynomial coefficients of the inlined sincos out of the loop as live VGPR constants and, at the 128-VGPR
// budget of a 1024-thread workgroup, spills them to scratch and reloads them on the critical path every step.
//
// The arithmetic: the library's sincos is ~190 instructions on role 3's chain every step (full-range Payne-Hanek
// machinery); body angles stay within a few turns, so for |a| <= 2^18 this is a three-term Cody-Waite reduction by pi/2
// with FMAs (error < 2^-100 |a|) followed by the fdlibm / musl kernels on [-pi/4, pi/4] with the reduction's tail:
// within 1 ulp of a correctly rounded sin / cos (checked against glibc on 2e7 arguments: 97.6 % identical, the rest
// 1 ulp), the same class as the library's own result; sincos_call(0) = (0, 1) exactly (cpvforangle(0)).
__device__ __attribute__((noinline)) double2 sincos_call(double a) // returns (sin a, cos a) in registers
{
    double2 o;
    sincos_body(a, &o.x, &o.y);
    return o;
}

// State loads.  In the fused multi-step loop the columns were rewritten by another wave of this workgroup one
// barrier ago: `fresh` (wave-uniform) selects agent-scope relaxed atomic loads (global_load ... sc1), which are served
// by the L2 instead of this CU's possibly stale vector L1.
__device__ __forceinline__ double ld_f64(const double *p, bool fresh)
{
    if (fresh)
        return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
                                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return *p;
}
__device__ __forceinline__ int ld_i32(const int32_t *p, bool fresh)
{
    if (fresh) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
__device__ __forceinline__ unsigned ld_u8(const uint8_t *p, bool fresh)
{
    if (fresh) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// Bank record element i (absolute index in doubles): LDS-staged bank or L2/HBM gather.
template <bool LDS_BANK>
__device__ __forceinline__ double bank_at(const DevCfg &c, int i)
{
    if constexpr (LDS_BANK) return lds_base()[i];
    else return c.bank[i];
}
// Two neighbouring doubles of a record.  When the bank is gathered from L2 / HBM (per-env records of the `fresh` /
// `fresh_device` modes, banks too large for the LDS) a gather instruction costs its 64 addresses, not its bytes: one
// 16-byte load per lane instead of two 8-byte ones (records are 8-byte aligned: gfx950 serves the unaligned dwordx4).
template <bool LDS_BANK>
__device__ __forceinline__ double2 bank_at2(const DevCfg &c, int i)
{
    if constexpr (LDS_BANK) { double2 r; r.x = lds_base()[i]; r.y = lds_base()[i + 1]; return r; }
    else {
        typedef double double2_a8 __attribute__((ext_vector_type(2), aligned(8)));
        const double2_a8 v = *reinterpret_cast<const double2_a8 *>(c.bank + i);
        double2 r; r.x = v.x; r.y = v.y; return r;
    }
}

// Stage `bytes` (multiple of 8) from global memory into LDS at offset 0 with LDS-DMA (global_load_lds_dwordx4:
// 1 KiB per wave-instruction, no VGPR round trip, all requests in flight at once), tail < 1 KiB through registers.
// (The tail's 8 bytes per thread are only REQUESTED here: `tail_v` is written to LDS at `tail_o` (>= 0) by the caller right before
// its wait in front of the barrier — written here, the wave waited for its whole DMA batch before asking for anything else.)
template <int THREADS>
__device__ __forceinline__ void stage_bank_lds(const double *__restrict__ bank, int bytes, double &tail_v, int &tail_o)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int NW = THREADS / 64;
    const int nchunk = bytes >> 10;
    const char *g = reinterpret_cast<const char *>(bank);
    char *l = reinterpret_cast<char *>(lds_base());
    for (int ch = wave; ch < nchunk; ch += NW) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (size_t)ch * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(l + ch * 1024), 16, 0, 0);
    }
    const int tail0 = nchunk << 10;
    const int o = tail0 + (int)threadIdx.x * 8;
    tail_o = (o < bytes) ? o : -1;
    tail_v = 0.0;
    if (o < bytes) tail_v = *reinterpret_cast<const double *>(g + o);
    static_assert(THREADS * 8 >= 1024, "tail copy needs one thread per 8 bytes of a 1 KiB chunk");
}

// Goal g's centre (comp 0 = x, 1 = y).  Configs 1-3: the goal bodies never move, so the map record holds them
// (goff = the record's goal block).  Config 4 (DYN): goals are dynamic bodies; the dyn kernel has just written this
// step's positions into the env's goal columns (goff = env index, shipsim_dynamics.hip).
template <bool LDS_BANK, bool DYN>
__device__ __forceinline__ double goal_at(const DevCfg &c, int goff, int g, int comp)
{
    if constexpr (DYN) return c.dyn_f64[(size_t)(DC_GOALS + DC_GOAL_COLS * g + comp) * (size_t)c.n_pad + goff];
    else return bank_at<LDS_BANK>(c, goff + 2 * g + comp);
}

// ShipGame.closest_goal (game.py:333-349): strict '<', first listed goal wins ties; (-1,-1) when none left.
template <bool LDS_BANK, bool DYN>
__device__ __forceinline__ void nearest_goal(const DevCfg &c, int goff, unsigned gm, double x, double y, double &gx,
                                             double &gy, const double *hG = nullptr /* gathered bank: this env's goal columns in LDS */, int ld = 0)
{
    gx = -1.0;
    gy = -1.0;
    double best = INFINITY;
    // (the loops track the winner's INDEX and fetch its centre once at the end — two selects per goal instead of six; the
    // centre reported is the stored one either way)
    int gi = -1;
    if constexpr (!LDS_BANK && !DYN) { // gathered record: the goal centres are in this env's LDS columns
#pragma unroll
        for (int g = 0; g < SSG_MAX_GOALS; ++g) {
            if (g >= c.n_goals) break;
            const double px = hG[(2 * g) * ld], py = hG[(2 * g + 1) * ld];
            const double dx = px - x, dy = py - y;
            const double d = obs_fma(dx, dx, dy * dy);
            const bool take = ((gm >> g) & 1u) & (d < best);
            best = take ? d : best;
            gi = take ? g : gi;
        }
        if (gi >= 0) { gx = hG[(2 * gi) * ld]; gy = hG[(2 * gi + 1) * ld]; }
        return;
    }
    for (int g = 0; g < c.n_goals; ++g) {
        const double px = goal_at<LDS_BANK, DYN>(c, goff, g, 0);
        const double py = goal_at<LDS_BANK, DYN>(c, goff, g, 1);
        const double dx = px - x, dy = py - y;
        const double d = obs_fma(dx, dx, dy * dy); // squared distance orders exactly like Vec2d.get_distance's sqrt
        const bool take = ((gm >> g) & 1u) & (d < best); // first alive goal always beats +inf
        best = take ? d : best;
        gi = take ? g : gi;
    }
    if (gi >= 0) { gx = goal_at<LDS_BANK, DYN>(c, goff, gi, 0); gy = goal_at<LDS_BANK, DYN>(c, goff, gi, 1); }
}

// The same over goal centres held in registers (config 4: gp[2g], gp[2g+1], requested in one batch at the head of the launch).
__device__ __forceinline__ void nearest_goal_regs(int n_goals, unsigned gm, double x, double y, const double *gp, double &gx, double &gy)
{
    gx = -1.0;
    gy = -1.0;
    double best = INFINITY;
#pragma unroll
    for (int g = 0; g < SSG_MAX_GOALS; ++g) {
        const double px = gp[2 * g], py = gp[2 * g + 1];
        const double dx = px - x, dy = py - y;
        const double d = obs_fma(dx, dx, dy * dy);
        const bool take = (g < n_goals) & (bool)((gm >> g) & 1u) & (d < best);
        best = take ? d : best;
        gx = take ? px : gx;
        gy = take ? py : gy;
    }
}

// The bank record an env moves to when ShipGame.reset gives it its next world: the next record of the shared bank, or —
// map_ring mode — the next record of the env's own ring [base, base + R).
// (`env` = the env's index in this handle: its ring starts at record env * R, reset_kernel.  Not `map_id - map_id % R`: the
// reciprocal of a run-time divisor is loop-invariant, the compiler kept it live around the step loop and spilled it — a scratch
// reload, i.e. a wait for every store the wave had in flight, in every step of the observer.)
__device__ __forceinline__ int next_map(const DevCfg &c, int map_id, int env)
{
    if (c.map_ring > 0) {
        const int base = env * c.map_ring;
        const int nxt = map_id + 1;
        return (nxt - base >= c.map_ring) ? base : nxt;
    }
    const int nxt = map_id + 1;
    return (nxt >= c.n_maps) ? 0 : nxt;
}

// Beam i of an env whose body rotation is (ca, sa): direction heading + phi_i by the angle-addition identity from
// host-computed cos/sin(phi_i), endpoint = origin + range * direction.  Owner lanes (culling) and worker lanes
// (segment query) both call this, so they see the same endpoint bits.
__device__ __forceinline__ void beam_end(double cx, double cy, double ca, double sa, double cphi, double sphi, double dist,
                                         double &ex, double &ey)
{
    const double ux = obs_fma(ca, cphi, -(sa * sphi)), uy = obs_fma(sa, cphi, ca * sphi);
    ex = obs_fma(dist, ux, cx);
    ey = obs_fma(dist, uy, cy);
}

// ---------------------------------------------------------------------------------------------------------
// LDS layout of the step kernel (after the optional bank copy); EPW envs = EPW/64 tiles of 64 envs per workgroup:
//   beamtab  [2][16] f64    cos/sin(phi_i)
//   shiptab  [6][8]  f64    per ship vertex i: local vertex, local plane normal, previous vertex;
//                           [0][6], [1][6]: the lidar origin of a ship standing at the spawn pose (a reset env)
//   pose     [7][EPW] f64   this step's post-step pose: x, y, cos a, sin a, lidar origin x, y, angle (role 3 -> roles 0-2:
//                           published under the tile's `ready` word, consumed under its `ack` word, see the kernel)
//   posem    [EPW] i32      the env's map id;   poser [EPW] i32  its rudder angle after this step's action
//   gres     [2][EPW] u32   colliding with a bank, by step parity: low half-word = the left bank (role 0), high = the right
//                           one (role 1) (-> role 3, role 2, the lidar roles)
//   gdone    [2][EPW] u32   role 3's results of the step, by step parity: bit 0 = no goals left | out of bounds | max_steps,
//                           (bit 1 unused), bits 2-5 = goal reached, out of bounds,
//                           max_steps, no goals left, bits 8.. = goals still listed after this step
//                           (-> role 2: reset decision, reward / done / flags outputs, nearest goal; -> the lidar roles:
//                           reset decision)
//   gtraf    [2][EPW] u32   config 4: the player touches a traffic ship (collide_ship; one word per lidar role, before the rendezvous)
//   sync     [3][EPW/64] u32  per tile: `ready` = number of poses role 3 has published, `ack` = number of pose reads the
//                           three consumer waves have completed, `bar` = arrivals at the tile's per-step rendezvous
//   goal scratch per role-3 wave: (lane, goal) pair queue u16[64*6] + consumed-goal masks u32[64]
//   per tile: res [2 parities][NB][64] u64 lidar result keys (step k's results live in parity k & 1; the parity role 3
//             has just consumed is its transposition buffer for the observation rows), then one (beam, hull) pair
//             queue u16[2*NB0*64 + 64 trash] per lidar wave
// ---------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ constexpr int nb_lo(int nb) { return (nb + 1) / 2; }
constexpr int kBeamTabBytes = 2 * SSG_MAX_BEAMS * 8;
constexpr int kShipTabBytes = 6 * 8 * 8;
constexpr int kTrafficTabBytes = SSG_N_TRAFFIC * 4 * 8 * 8; // config 4: per traffic ship k, [k][0..3][i] = local vertex x, y, plane normal x, y
constexpr int kPoseDoubles = 7;
constexpr int kGoalScratchBytes = 64 * SSG_MAX_GOALS * 2 + 64 * 4; // per goals wave: pair queue (u16) + consumed-goal masks
constexpr int kTrafficScratchBytes = 64 * SSG_N_TRAFFIC * 2 + 64 * 4; // config 4, per lidar-hi wave: (lane, ship) pair queue (u16) + hit words (lidar-lo: inside its beam pair queue)
// (dyn: the config-4 instantiations — the traffic hull table, the per-env traffic words and the lidar-hi wave's (lane, ship) pair
// queue exist only there; the 1-ship kernels do not pay ~5.4 KB of LDS for them, which is 4-5 records of a staged bank)
__host__ __device__ __forceinline__ constexpr int lds_fixed_bytes(int epw, bool dyn)
{
    return kBeamTabBytes + kShipTabBytes + (dyn ? kTrafficTabBytes : 0) + kPoseDoubles * epw * 8 + (dyn ? 8 : 6) * epw * 4 + 4 * (epw / 64) * 4 +
           (epw / 64) * (kGoalScratchBytes + (dyn ? kTrafficScratchBytes : 0));
}
__host__ __device__ __forceinline__ constexpr int lds_res_bytes(int nb) { return nb * 64 * 8; } // one parity of one tile
__host__ __device__ __forceinline__ constexpr int lds_queue_bytes(int nb0) { return (2 * nb0 * 64 + 64) * 2; }
__host__ __device__ __forceinline__ constexpr int lds_tile_bytes(int nb)
{
    return 2 * lds_res_bytes(nb) + 2 * lds_queue_bytes(nb_lo(nb));
}

// Gathered banks (LDS_BANK = false: per-env records of the `fresh` / `fresh_device` modes, banks too large for the LDS): a
// divergent gather costs the CU's address path one request per LANE (~1 per cycle), and four roles re-gathering the head of
// their env's record every step (hull counts + boxes, goal centres: 30 of a tile-step's ~80 gathers, all 64 lanes each) was
// more than half of that.  The head of the record is therefore kept in per-lane LDS columns and re-fetched only by the lanes
// whose env has just moved to its next record (a reset): [2 lidar roles][10][EPW] = record doubles 0..9 (counts, boxes), one
// copy per lidar role because each follows its tile's resets on its own; [12][EPW] = doubles 10..21 (goal centres), written by
// the body role, read by it and by the observer.
constexpr int kHdrLidar = 10, kHdrGoals = 2 * SSG_MAX_GOALS;
static_assert(SSG_MAP_OFF_COUNTS == 0 && SSG_MAP_OFF_AABB == 2 && SSG_MAP_OFF_GOALS == kHdrLidar && SSG_MAP_OFF_SPAWN_GOAL == kHdrLidar + kHdrGoals,
              "the record's head: counts, boxes, goals, spawn goal");
// Waves per 64-env tile.  Four roles (see the step kernel); SIX on the 64- and 128-env workgroups of the 1-ship kernels — the
// smaller batches (<= 32 768 envs: at most one workgroup per CU), where a wave is alone on its SIMD or shares it with one other
// and a step is as long as its longest dependent chain, which was a lidar wave's: query k (5.5 k cycles), then collide_ship against
// its bank hull (3.2 k) on the pose the body had published 2.6 k cycles before the query ended, then the rendezvous.  collide_ship
// gets waves of its own (roles 4 and 5, one bank hull each): it runs beside the query instead of behind it.  4 096 envs x 10 beams:
// 4.39 -> 3.63 us per step; 32 768 envs (128-env workgroups): 4.69 -> 4.41.  (256-env workgroups: 24 waves would be 1 536 threads.)
#ifndef SSG_SIX_ROLE_MAX_EPW
#define SSG_SIX_ROLE_MAX_EPW 128
#endif
__host__ __device__ __forceinline__ constexpr int tile_roles(int epw, bool dyn) { return (epw <= SSG_SIX_ROLE_MAX_EPW && !dyn) ? 6 : 4; }
// gathered bank: record doubles 0..9 per role that queries or collides (2, or 4 with six roles), the goal centres once
__host__ __device__ __forceinline__ constexpr int lds_hdr_bytes(int epw, bool lds_bank, bool dyn) { return lds_bank ? 0 : ((tile_roles(epw, dyn) - 2) * kHdrLidar + kHdrGoals) * epw * 8; }

constexpr unsigned long long kLidarMiss = ~0ull; // result key of a beam no hull reported a hit for

// ---------------------------------------------------------------------------------------------------------
// LiDAR (models.py:39-76), wave-compacted.
//
// A beam can only touch a bank hull if the beam's bounding box meets the hull's (most beams of most ships do
// not: the river is wider than the 100-unit range).  Each owner lane therefore only CULLS its NB x 2 (beam, hull)
// pairs; the surviving pairs of the whole wave are compacted into a per-wave LDS queue (the v_cmp masks are the
// ballots; mbcnt gives the slot) and processed 64 at a time, one pair per lane, so the plane loops run on dense
// wavefronts.  A worker lane pulls the pose of the env it serves with ds_bpermute, rebuilds the beam and runs
// cpShapeSegmentQuery(shape, a=(cx,cy), b=(ex,ey), r=0) against one hull:
//   EXACT = true : cpPolyShapeSegmentQuery literally — every plane is intersected (one division per plane),
//                  accepted when the crossing lies inside the edge's extent, later planes overwrite.
//   EXACT = false: the same predicate with one division per beam: among the planes the beam crosses front-to-back
//                  within its length (d >= 0 and d <= den, i.e. 0 <= t <= 1) only the one with the largest t can
//                  be the entry edge of a convex polygon, so only that plane gets the exact t = d/den, lerp and
//                  edge-extent test.  Identical results except when a ray passes within rounding of a hull vertex.
// A hit is published as a 64-bit key (hull index << 63 | bits of the distance) with an LDS atomic min into
// res[beam][lane]: the smallest key is the hit of the FIRST shape in list order that reports one (models.py:61-72:
// the left bank before the right), whatever order the pairs were processed in; kLidarMiss = no hit.
// ---------------------------------------------------------------------------------------------------------
// hull planes fetched ahead of their arithmetic, per loop trip: 4 from LDS; 8 when the record is gathered from L2 / HBM (every
// trip is then a dependent ~1 k-cycle round trip on the lidar role's chain, and 99.4 % of the bank hulls have <= 8 planes)
template <bool LDS_BANK> struct PlaneChunk { static constexpr int n = LDS_BANK ? 4 : 8; };

template <bool LDS_BANK, bool EXACT>
__device__ __forceinline__ void lidar_pass(const DevCfg &c, const int n_items, const unsigned short *queue,
                                           unsigned long long *res /* [this role's first beam][64] */, const double *beamtab,
                                           const double cx, const double cy, const double ca, const double sa,
                                           const int rec_off, const int lane, const double *hLt = nullptr /* gathered bank: the
                                           tile's count columns of this role's header copy: [s * ld + lane of the tile] */, const int ld = 0)
{
    for (int base = 0; base < n_items; base += 64) {
        const int idx = base + lane;
        const bool act = idx < n_items;
        const unsigned code = queue[act ? idx : 0];
        const int src = code & 63, bi = (code >> 6) & (SSG_MAX_BEAMS - 1), s = (code >> 10) & 1;
        const double wcx = __shfl(cx, src), wcy = __shfl(cy, src), wca = __shfl(ca, src), wsa = __shfl(sa, src);
        const int woff = __shfl(rec_off, src);
        double ex, ey;
        beam_end(wcx, wcy, wca, wsa, beamtab[bi], beamtab[SSG_MAX_BEAMS + bi], c.lidar_dist, ex, ey);
        int cnt;
        if constexpr (LDS_BANK) cnt = (int)bank_at<LDS_BANK>(c, woff + SSG_MAP_OFF_COUNTS + s);
        else cnt = (int)hLt[s * ld + src];
        const int pb = woff + SSG_MAP_OFF_PLANES + s * (SSG_MAX_HULL * SSG_PLANE_DOUBLES);
        bool outside = false; // cpPolyShapePointQuery(a): some plane has a strictly in front
        bool out_sure = false, maybe = false;
        constexpr double kSignEps = 1e-9;
        bool ok = false;
        double ptx = ex, pty = ey;
        double bd = -1.0, bden = 1.0, t_hit = 0.0;
        int bj = 0;
        constexpr int kPlaneChunk = PlaneChunk<LDS_BANK>::n;
        for (int j0 = 0; __any(act & (j0 < cnt)); j0 += kPlaneChunk) {
            double pv0x[kPlaneChunk], pv0y[kPlaneChunk], pnx[kPlaneChunk], pny[kPlaneChunk], pv0n[kPlaneChunk];
            double pdtmin[kPlaneChunk], pdtmax[kPlaneChunk];
#pragma unroll
            for (int u = 0; u < kPlaneChunk; ++u) { // all LDS reads of the chunk first: one latency per chunk
                const int j = j0 + u;
                const int q = pb + SSG_PLANE_DOUBLES * ((j < SSG_MAX_HULL) ? j : 0); // independent of cnt: no LDS round trip in between
                // (gathered record: only the lanes that have a plane j request it — a gather costs per active lane, and the count
                // comes from LDS, so nothing is gained by fetching past the hull's last plane)
                const bool want = LDS_BANK || (act & (j < cnt));
                pv0x[u] = pv0y[u] = pnx[u] = pny[u] = pv0n[u] = 0.0;
                if (EXACT && want) { const double2 vv = bank_at2<LDS_BANK>(c, q + 0); pv0x[u] = vv.x; pv0y[u] = vv.y; }
                if (want) { const double2 nn = bank_at2<LDS_BANK>(c, q + 2); pnx[u] = nn.x; pny[u] = nn.y; }
                if (want) pv0n[u] = bank_at<LDS_BANK>(c, q + 4);
                if (EXACT) {
                    // the edge's extent along the plane: cpvcross(n, v[j-1]) .. cpvcross(n, v[j]); v[j-1] is the
                    // previous plane's v0 (the last plane's for j = 0)
                    const int jp = (j == 0) ? cnt - 1 : j - 1;
                    const int qp = pb + SSG_PLANE_DOUBLES * ((jp >= 0 && jp < SSG_MAX_HULL) ? jp : 0);
                    const double ux = bank_at<LDS_BANK>(c, qp + 0), uy = bank_at<LDS_BANK>(c, qp + 1);
                    pdtmin[u] = obs_fma(pnx[u], uy, -(pny[u] * ux));
                    pdtmax[u] = obs_fma(pnx[u], pv0y[u], -(pny[u] * pv0x[u]));
                }
            }
#pragma unroll
            for (int u = 0; u < kPlaneChunk; ++u) {
                const int j = j0 + u;
                const bool valid = act & (j < cnt);
                const double nx = pnx[u], ny = pny[u], v0n = pv0n[u];
                const double an = obs_fma(wcx, nx, wcy * ny);
                const double d = an - v0n;
                if (EXACT) {
                    outside = outside | (valid & ((nx * (wcx - pv0x[u]) + ny * (wcy - pv0y[u])) > 0.0));
                } else {
                    // cpPolyShapePointQuery's sign test cpvdot(n, a - v0) > 0 differs from d = a.n - v0.n by rounding only
                    // (< 1e-12 at these magnitudes): beyond kSignEps it is decided by d; closer, by the exact expression
                    // in the rare pass below (two fewer LDS gathers per plane)
                    out_sure = out_sure | (valid & (d > kSignEps));
                    maybe = maybe | (valid & !(d < -kSignEps));
                }
                const bool front = valid & !(d < 0.0);
                const double bn = obs_fma(ex, nx, ey * ny);
                const double den = dmax(an - bn, DBL_MIN);
                if (EXACT) {
                    const double t = d / den;
                    const double omt = 1.0 - t;
                    const double qx = obs_fma(ex, t, wcx * omt), qy = obs_fma(ey, t, wcy * omt); // cpvlerp(a,b,t)
                    const double dtv = obs_fma(nx, qy, -(ny * qx));                  // cpvcross(n, point)
                    const bool acc = front & !((t < 0.0) | (1.0 < t)) & (pdtmin[u] <= dtv) & (dtv <= pdtmax[u]);
                    ok = ok | acc;
                    ptx = acc ? qx : ptx;
                    pty = acc ? qy : pty;
                } else {
                    // candidate: 0 <= d/den <= 1; better: d/den >= best (cross-multiplied, dens > 0; ties -> later)
                    const bool better = front & (d <= den) & (d * bden >= bd * den);
                    bd = better ? d : bd;
                    bden = better ? den : bden;
                    bj = better ? j : bj;
                }
            }
        }
        if (!EXACT) {
            outside = out_sure;
            const bool need_exact = act & !out_sure & maybe; // the origin is within kSignEps of a plane and clearly in front of none
            if (__any(need_exact)) {
                bool o = false;
                for (int j = 0; __any(need_exact & (j < cnt)); ++j) {
                    const int qq = pb + SSG_PLANE_DOUBLES * ((j < SSG_MAX_HULL) ? j : 0);
                    const double v0x = bank_at<LDS_BANK>(c, qq + 0), v0y = bank_at<LDS_BANK>(c, qq + 1);
                    const double nx = bank_at<LDS_BANK>(c, qq + 2), ny = bank_at<LDS_BANK>(c, qq + 3);
                    o = o | ((j < cnt) & ((nx * (wcx - v0x) + ny * (wcy - v0y)) > 0.0));
                }
                outside = need_exact ? o : outside;
            }
            const int q = pb + SSG_PLANE_DOUBLES * bj;
            // the entry edge's extent: cpvcross(n, v[bj-1]) .. cpvcross(n, v[bj])
            const int jp = (bj == 0) ? cnt - 1 : bj - 1;
            const int qp = pb + SSG_PLANE_DOUBLES * ((jp >= 0 && jp < SSG_MAX_HULL) ? jp : 0);
            // the entry plane's normal and the two vertices of its edge: three 16-byte requests in ONE batch (six 8-byte loads in
            // program order between the products were compiled into three dependent round trips per pass — on a gathered bank each
            // is a trip to L2 / HBM on the lidar wave's chain)
            double2 nn = bank_at2<LDS_BANK>(c, q + 2), vj = bank_at2<LDS_BANK>(c, q + 0), vp = bank_at2<LDS_BANK>(c, qp + 0);
            if constexpr (!LDS_BANK) asm volatile("" : "+v"(nn.x), "+v"(nn.y), "+v"(vj.x), "+v"(vj.y), "+v"(vp.x), "+v"(vp.y));
            const double nx = nn.x, ny = nn.y;
            const double dtmin = obs_fma(nx, vp.y, -(ny * vp.x));
            const double dtmax = obs_fma(nx, vj.y, -(ny * vj.x));
            const double t = bd / bden;
            const double omt = 1.0 - t;
            ptx = obs_fma(ex, t, wcx * omt);
            pty = obs_fma(ey, t, wcy * omt);
            const double dtv = obs_fma(nx, pty, -(ny * ptx));
            ok = (bd >= 0.0) & (dtmin <= dtv) & (dtv <= dtmax);
            t_hit = t;
        }
        // start point inside (or on) the polygon: hit at alpha 0 whose reported point is the FAR end b (App. A.7)
        const bool hit = act & (ok | !outside);
        double dist;
        if constexpr (EXACT) {
            const double px = outside ? ptx : ex, py = outside ? pty : ey;
            const double dx = px - wcx, dy = py - wcy;
            dist = sqrt(obs_fma(dx, dx, dy * dy)); // Vec2d.get_distance, literally
        } else {
            // |cpvlerp(a, b, t) - a| = t |b - a| = t x the beam's length (the far end itself when the origin is inside the hull):
            // the reading without the point difference and the square root (~20 wave-instructions per pass); it differs from the
            // literal form by rounding only (~1e-13, held to 1e-9 by test_exact_and_one_division_lidar_agree)
            dist = outside ? t_hit * c.lidar_dist : c.lidar_dist;
        }
        if (hit) {
            const unsigned long long key = ((unsigned long long)s << 63) | (unsigned long long)__double_as_longlong(dist);
            atomicMin(&res[bi * 64 + src], key); // ds_min_u64: hull 0's hit beats hull 1's
        }
#ifdef SSG_LIDAR_COUNT /* diagnostic builds: how many of the pairs that survive the cull are hits? */
        if (c.dbg) {
            const unsigned long long ma = __ballot(act), mh = __ballot(hit), mo = __ballot(act & !outside);
            if (lane == 0) { atomicAdd(c.dbg + 65536 + 0, (unsigned long long)__popcll(ma)); atomicAdd(c.dbg + 65536 + 1, (unsigned long long)__popcll(mh));
                             atomicAdd(c.dbg + 65536 + 2, 1ull); atomicAdd(c.dbg + 65536 + 3, (unsigned long long)__popcll(mo)); }
        }
#endif
    }
}

// Output stores.  The per-XCD L2s are write-back and not coherent with each other, so the end of a kernel writes every
// line the kernel dirtied back to the fabric in one burst.  -DSSG_WT (experiment, not kept): write-through (sc1) stores
// of the same 8 bytes per lane are slower both ways on gfx950: single-step launch 19.7 vs 17.9 us, fused step 8.7 vs 5.6.
// Non-temporal stores (__builtin_nontemporal_store, round 3): 18.5 us per fused step instead of 5.4 — not an option either.
template <class T>
__device__ __forceinline__ void st_out(T *p, T v)
{
#ifdef SSG_WT
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *p = v;
#endif
}

// Diagnostic stamps (-DSSG_STAMPS builds only; the product kernel executes none): s_memtime at section boundaries,
// written by lane 0 of every wave to a buffer nothing else reads.
#ifdef SSG_STAMPS
#define SSG_STAMP(k)                                                                              \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        stamp_[k] = t_;                                                                           \
    } while (0)
#define SSG_STAMP_FLUSH(n)                                                                        \
    do {                                                                                          \
        if (c.dbg && lane == 0) {                                                                 \
            unsigned long long *d_ = c.dbg + 16 * (size_t)(blockIdx.x * (NR * EPW / 64) + (threadIdx.x >> 6)); \
            for (int k_ = 0; k_ < 16; ++k_) d_[k_] = stamp_[k_];  /* 8: kernel start, 9: after barrier 0, 10: role end */ \
        }                                                                                         \
    } while (0)
#ifdef SSG_STAMPS_ITER
#define SSG_STAMP_K(i) do { } while (0)
#else
#define SSG_STAMP_K(i) do { if (k == (K >= 2 ? K - 2 : 0)) SSG_STAMP(i); } while (0) /* the second-to-last step of a fused launch */
#endif
#else
#define SSG_STAMP(k) do { } while (0)
#define SSG_STAMP_K(i) do { } while (0)
#define SSG_STAMP_FLUSH(n) do { } while (0)
#endif

// The shader clock DURING a launch (product builds; ssg_debug_launch_clock): every wave reads the shader-clock counter (s_memtime)
// and the constant 100 MHz reference counter (s_memrealtime) when it starts; when the first wave of workgroup 0 ends it stores how
// far both advanced — two scalar reads per wave and, with no buffer installed, one untaken branch.  (-DSSG_STAMPS builds use
// c.dbg for their own stamps.)
#ifdef SSG_STAMPS
#define SSG_CLOCK_BEGIN() do { } while (0)
#define SSG_CLOCK_END() do { } while (0)
#else
#define SSG_CLOCK_BEGIN() const unsigned long long clk_c0_ = __builtin_amdgcn_s_memtime(), clk_r0_ = __builtin_amdgcn_s_memrealtime()
#define SSG_CLOCK_END()                                                                            \
    do {                                                                                           \
        if (c.dbg && blockIdx.x == 0 && threadIdx.x == 0) {                                        \
            c.dbg[0] = __builtin_amdgcn_s_memtime() - clk_c0_;                                     \
            c.dbg[1] = __builtin_amdgcn_s_memrealtime() - clk_r0_;                                 \
        }                                                                                          \
    } while (0)
#endif

// Timing-only ablation switches (development builds with -DSSG_ABLATION; never in the product library): bits
// 16.. of DevCfg.flags skip a section so its share of the kernel time can be measured.  Outputs are wrong.
#ifdef SSG_ABLATION
#define SSG_ABL(bit) (c.flags & (1u << (16 + (bit))))
#else
#define SSG_ABL(bit) false
#endif

// cpPolyShapeCacheData of the ship: world vertices and AABB for the body rotation (ca, sa) at (x, y).  Every role
// that needs them runs exactly these operations on the same inputs, so all of them hold the same bits.
__device__ __forceinline__ void ship_world(const double *shiptab, double ca, double sa, double x, double y,
                                           double (&swx)[SSG_SHIP_VERTS], double (&swy)[SSG_SHIP_VERTS], double &sbl,
                                           double &sbr, double &sbb, double &sbt)
{
    sbl = INFINITY; sbr = -INFINITY; sbb = INFINITY; sbt = -INFINITY;
#pragma unroll
    for (int i = 0; i < SSG_SHIP_VERTS; ++i) {
        const double hx = shiptab[0 * 8 + i], hy = shiptab[1 * 8 + i];
        swx[i] = ca * hx + (-sa) * hy + x;
        swy[i] = sa * hx + ca * hy + y;
        sbl = dmin(sbl, swx[i]); sbr = dmax(sbr, swx[i]);
        sbb = dmin(sbb, swy[i]); sbt = dmax(sbt, swy[i]);
    }
}

// One LiDAR.query of a lidar wave for its NB0 beams [b_first, b_first + b_count) of the 64 envs of its tile, from the
// lidar origin (cx, cy) and body rotation (ca, sa): cull, compact, segment queries; results into res[beam][lane].
template <int NB, bool LDS_BANK, bool EXACT>
__device__ __forceinline__ void lidar_query(const DevCfg &c, unsigned long long *res, unsigned short *queue,
                                            const double *beamtab, const int b_first, const int b_count, const double cx,
                                            const double cy, const double ca, const double sa, const int rec_off,
                                            const bool live, const int lane, const double *hL = nullptr /* gathered bank: this env's
                                            column of the role's header copy */, const int ld = 0)
{
    constexpr int NB0 = nb_lo(NB);
    constexpr int kTrash = 2 * NB0 * 64; // 64 u16 past the queue swallow the writes of culled pairs
    int n_items = 0;
    // hull AABBs widened by eps: culling must never drop a pair the reference would hit
    const double eps = 1e-6;
    double al[2], ab[2], ar[2], at[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        double2 lb, rt;
        if constexpr (LDS_BANK) {
            lb = bank_at2<LDS_BANK>(c, rec_off + SSG_MAP_OFF_AABB + 4 * s + 0);
            rt = bank_at2<LDS_BANK>(c, rec_off + SSG_MAP_OFF_AABB + 4 * s + 2);
        } else {
            lb.x = hL[(SSG_MAP_OFF_AABB + 4 * s + 0) * ld]; lb.y = hL[(SSG_MAP_OFF_AABB + 4 * s + 1) * ld];
            rt.x = hL[(SSG_MAP_OFF_AABB + 4 * s + 2) * ld]; rt.y = hL[(SSG_MAP_OFF_AABB + 4 * s + 3) * ld];
        }
        al[s] = lb.x - eps;
        ab[s] = lb.y - eps;
        ar[s] = rt.x + eps;
        at[s] = rt.y + eps;
    }
    (void)ab; (void)at; (void)al; (void)ar; // (the -DSSG_QUEUE_BEAM_MAJOR variant tests all four ranges)
#ifdef SSG_QUEUE_BEAM_MAJOR /* the round-2..5 order (tools/build_variant.sh beammajor -DSSG_QUEUE_BEAM_MAJOR), for A/B timing */
#pragma unroll
    for (int k = 0; k < NB0; ++k) {
        if (k < b_count) { // wave-uniform
            const int i = b_first + k;
            res[i * 64 + lane] = kLidarMiss;
            if (!SSG_ABL(1)) {
                double ex, ey;
                beam_end(cx, cy, ca, sa, beamtab[i], beamtab[SSG_MAX_BEAMS + i], c.lidar_dist, ex, ey);
                const double lox = dmin(cx, ex), hix = dmax(cx, ex), loy = dmin(cy, ey), hiy = dmax(cy, ey);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bool need = live & (lox <= ar[s]) & (al[s] <= hix) & (loy <= at[s]) & (ab[s] <= hiy);
                    const unsigned long long m = __ballot(need);
                    const int pos = n_items + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                                              __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    queue[need ? pos : (kTrash + lane)] = (unsigned short)(lane | (k << 6) | (s << 10));
                    n_items += __popcll(m);
                }
            }
        }
    }
#else
    // ENV-MAJOR queue (round 6): every lane first collects which of its 2 x NB0 (beam, hull) pairs survive the cull in a bit
    // mask, an inclusive scan of the counts over the wave (six DPP adds, no LDS) gives every lane the slot of its first pair,
    // and it writes its pairs back to back.  The worker lanes of a pass then serve the pairs of ONE env side by side: they read
    // the same record's planes at the same LDS addresses (a broadcast) where the beam-major order put 64 different envs —
    // 64 records, several to a bank — next to each other (rocprofv3 SQ_LDS_BANK_CONFLICT: 27 % of the LDS pipe's busy cycles).
    unsigned needmask = 0u;
    const double lca = c.lidar_dist * ca, lsa = c.lidar_dist * sa;
    const bool org0 = live & (cx <= ar[0]), org1 = live & (al[1] <= cx); // the origin itself is within the hull's x range
#pragma unroll
    for (int k = 0; k < NB0; ++k) {
        if (k < b_count) { // wave-uniform
            const int i = b_first + k;
            res[i * 64 + lane] = kLidarMiss;
            if (!SSG_ABL(1)) {
                // Beam i points along heading + phi_i, phi_i = rad(90 - spread/2) + i*rad(spread/n_beams)
                // (models.py:48-49,62-64); endpoint via the angle-addition identity (beam_end): agrees with the
                // reference's per-beam cos/sin to ~1e-13 and only feeds lidar readings, never the dynamics.
                // keep a pair unless the beam and the hull's (widened) box are disjoint IN X, tested from ONE side per hull.  (A cull
                // only has to be conservative: a pair it keeps in vain runs through the segment query and reports a miss.  The
                // banks span the whole height of the world — game_map.py:22-73: y from -100 to 1.2 H — so the y half of the box
                // test rejected nothing; hull 0 is the left bank, which a beam can only reach by getting as far LEFT as the bank's
                // right edge, hull 1 the right bank (game_map.py:67-71, models.py:172-182: the list order lidar readings depend
                // on).  For banks laid out otherwise the test stays correct — it never drops a pair that could hit — and merely
                // culls less.  Four compares, two min / max and the beam end's y per pair before; one compare now.)
                const double ex = obs_fma(lca, beamtab[i], obs_fma(-lsa, beamtab[SSG_MAX_BEAMS + i], cx)); // cx + dist * cos(heading + phi_i)
                needmask |= (org0 | (live & (ex <= ar[0]))) ? (1u << (2 * k + 0)) : 0u;
                needmask |= (org1 | (live & (al[1] <= ex))) ? (1u << (2 * k + 1)) : 0u;
            }
        }
    }
    {
        const int cnt = __popc(needmask);
        int incl = cnt; // inclusive scan over the wave: row_shr 1, 2, 4, 8 inside the rows of 16, then the rows' totals forwarded
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false); // row_bcast:15 into rows 1 and 3
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false); // row_bcast:31 into rows 2 and 3
        n_items = __builtin_amdgcn_readlane(incl, 63);
        int pos = incl - cnt;
#pragma unroll
        for (int k = 0; k < NB0; ++k) {
            if (k < b_count) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bool need = (needmask >> (2 * k + s)) & 1u;
                    queue[need ? pos : (kTrash + lane)] = (unsigned short)(lane | (k << 6) | (s << 10));
                    pos += need ? 1 : 0;
                }
            }
        }
    }
#endif
    if (!SSG_ABL(3))
        lidar_pass<LDS_BANK, EXACT>(c, n_items, queue, res + b_first * 64, beamtab + b_first, cx, cy, ca, sa, rec_off, lane,
                                    hL ? hL - lane : nullptr, ld);
}

// ---------------------------------------------------------------------------------------------------------
// Observation rows of one tile (64 envs, ship_env.py:79-113,156): row = [previous frame | new frame] (history 2) or
// [new frame] (history 1), DH doubles per row, rows of neighbouring envs adjacent in HBM.  A lane holds ITS env's row
// in registers; scattered 8-byte stores of it cost 64 write requests per instruction and used to dominate the step, so
// the tile is transposed through LDS: CP columns at a time (what fits the buffer role 3 is handed: the lidar result
// buffer it has just emptied) every lane writes its values column-major — conflict-free, full-wave ds_writes — and
// reads the chunk back row-major, so that consecutive lanes store consecutive doubles of a row's CP-column segment.
// LDS operations of one wave execute in issue order, so no wait separates the passes.  Column stride 73 doubles: the
// read-back of ~9 rows x CP columns by one wave-instruction then falls on distinct banks.
// ---------------------------------------------------------------------------------------------------------
template <int NB>
struct ObsTile {
    static constexpr int FF = 6 + NB;
    static constexpr int CS = (NB >= 2) ? 73 : 64;  // column stride (doubles)
    static constexpr int CP = (NB * 64) / CS;       // columns per pass
    static_assert(CP >= 1 && CP * CS * 8 <= lds_res_bytes(NB), "observation chunk does not fit the lidar result buffer");
    // Read-back position of this lane's jj-th element of a full pass (element t = lane + 64*jj of the 64 x CP chunk,
    // row-major): where it sits in the column buffer and where it goes in the tile's rows.  The same for every full pass
    // and every step, so it is computed once per launch and kept in registers (the pass offset is an immediate).
    // (a power-of-two chunk width, e.g. 8 columns at 10 beams: the positions are a shift and a mask of the lane id, computed
    // where they are used instead of held in 3*CP registers the 10-beam observer does not have)
    static constexpr bool kOnTheFly = (CP & (CP - 1)) == 0;
    int lds_at[kOnTheFly ? 1 : CP]; // (column * CS + row) * 8 bytes... in doubles
    int row[kOnTheFly ? 1 : CP], col[kOnTheFly ? 1 : CP];
    __device__ __forceinline__ void init(int lane)
    {
        if constexpr (!kOnTheFly) {
#pragma unroll
            for (int jj = 0; jj < CP; ++jj) {
                const int t = lane + 64 * jj;
                row[jj] = t / CP;
                col[jj] = t - row[jj] * CP;
                lds_at[jj] = col[jj] * CS + row[jj];
            }
        }
    }
    __device__ __forceinline__ int row_of(int jj, int lane) const { if constexpr (kOnTheFly) return (lane + 64 * jj) / CP; else return row[jj]; }
    __device__ __forceinline__ int col_of(int jj, int lane) const { if constexpr (kOnTheFly) return (lane + 64 * jj) % CP; else return col[jj]; }
    __device__ __forceinline__ int lds_of(int jj, int lane) const { if constexpr (kOnTheFly) return col_of(jj, lane) * CS + row_of(jj, lane); else return lds_at[jj]; }
};

// One range [P_BEGIN, P_END) of the passes of a tile (the observer splits a tile's passes around a workgroup barrier).
template <int NB, bool H2, int P_BEGIN, int P_END, class Val>
__device__ __forceinline__ void write_obs_tile(const ObsTile<NB> &ot, double *colbuf, const Val &val /* val(j): column j of this lane's row */,
                                               double *__restrict__ obase, const int rows_live, const int lane)
{
    constexpr int FF = 6 + NB;
    constexpr int DH = H2 ? 2 * FF : FF;
    constexpr int CS = ObsTile<NB>::CS, CP = ObsTile<NB>::CP;
    // Launder the per-launch positions once per call: what is DERIVED from them (byte offsets, 64-bit addresses, one per
    // pass) is loop-invariant across the fused steps too, and hoisted out of the step loop it is spilled to scratch.
    int lds_at[CP], gl_at[CP];
    unsigned okmask = 0;
    int ln_ = lane;
    asm volatile("" : "+v"(ln_)); // (positions derived from the lane id are not hoisted out of the step loop either)
#pragma unroll
    for (int jj = 0; jj < CP; ++jj) {
        lds_at[jj] = ot.lds_of(jj, ln_);
        gl_at[jj] = ot.row_of(jj, ln_) * DH + ot.col_of(jj, ln_);
        okmask |= (ot.row_of(jj, ln_) < rows_live) ? (1u << jj) : 0u;
        if constexpr (!ObsTile<NB>::kOnTheFly) asm volatile("" : "+v"(lds_at[jj]), "+v"(gl_at[jj]));
    }
    asm volatile("" : "+v"(okmask));
#pragma unroll
    for (int pi = P_BEGIN; pi < P_END; ++pi) {
        const int p0 = pi * CP;
        if (p0 >= DH) break;
        const int cpp = (DH - p0 < CP) ? (DH - p0) : CP; // columns of this pass (a constant once unrolled)
#pragma unroll
        for (int cc = 0; cc < CP; ++cc) {
            if (cc < cpp) colbuf[cc * CS + lane] = val(p0 + cc); // (computed here, not ahead: registers)
        }
        if (cpp == CP) {
#pragma unroll
            for (int jj = 0; jj < CP; ++jj) {
                const double v = colbuf[lds_at[jj]];
                if ((okmask >> jj) & 1u) st_out(&obase[(unsigned)(gl_at[jj] + p0)], v); // uniform base + 32-bit lane offset
            }
        } else { // a shorter last pass (row lengths that are no multiple of CP): positions computed on the spot
            int ln = lane;
            asm volatile("" : "+v"(ln)); // (not hoisted out of the step loop: registers)
#pragma unroll
            for (int jj = 0; jj < CP; ++jj) {
                if (jj < cpp) {
                    const int t = ln + 64 * jj;
                    const int r = t / cpp, cc = t - r * cpp;
                    const double v = colbuf[cc * CS + r];
                    if (r < rows_live) st_out(&obase[(unsigned)(r * DH + p0 + cc)], v);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0); // keep the passes apart: the next pass's values are not computed (and held) early
    }
}

// The same transposition with 16-byte operations (even row lengths, >= 8 beams: the 4 KiB buffer holds 64 rows x 8 columns): a lane
// writes its row's columns [C0, C0 + 8) as four double2 into row-major LDS — pair q of row r in slot q ^ ((r >> 2) & 3), so that
// the sixteen lanes of a ds_write_b128 / ds_read_b128 phase fall on sixteen different four-bank groups — and reads pair
// p = lane + 64 jj of the chunk back (row p / pairs, pair p % pairs): consecutive lanes then store consecutive 16-byte pieces of a
// row.  Twelve memory instructions per eight columns instead of twenty-four, and shifts instead of per-element positions.
template <int NB, bool H2, int C_BEGIN, int C_END, class Val>
__device__ __forceinline__ void write_obs_pairs(double *buf /* 16-byte aligned, >= 4 KiB */, const Val &val /* val(j): column j of this lane's row */,
                                                double *__restrict__ obase, const int rows_live, const int lane)
{
    constexpr int FF = 6 + NB;
    constexpr int DH = H2 ? 2 * FF : FF;
    static_assert(C_BEGIN % 2 == 0 && C_END % 2 == 0 && DH % 2 == 0 && NB >= 8, "pairs of columns; 64 x 8 doubles of buffer");
    int ln = lane;
    asm volatile("" : "+v"(ln)); // (positions derived from the lane id are not hoisted out of the step loop: registers)
    const int wsw = (ln >> 2) & 3;
#pragma unroll
    for (int c0 = C_BEGIN; c0 < C_END; c0 += 8) {
        constexpr int kMax = 4;
        const int np = ((C_END - c0) / 2 < kMax) ? (C_END - c0) / 2 : kMax; // pairs per row in this chunk (a constant once unrolled)
#pragma unroll
        for (int q = 0; q < kMax; ++q)
            if (q < np) *reinterpret_cast<double2 *>(buf + ln * 8 + 2 * (q ^ wsw)) = make_double2(val(c0 + 2 * q), val(c0 + 2 * q + 1));
#pragma unroll
        for (int jj = 0; jj < kMax; ++jj) {
            if (jj < np) {
                const int pidx = ln + 64 * jj;
                const int r = (np == 4) ? (pidx >> 2) : (np == 2) ? (pidx >> 1) : (np == 1) ? pidx : pidx / 3;
                const int q = pidx - r * np;
                const double2 v = *reinterpret_cast<const double2 *>(buf + r * 8 + 2 * (q ^ ((r >> 2) & 3)));
                if (r < rows_live) *reinterpret_cast<double2 *>(&obase[(unsigned)(r * DH + c0 + 2 * q)]) = v;
            }
        }
        __builtin_amdgcn_sched_barrier(0); // keep the chunks apart: the next chunk's values are not computed (and held) early
    }
}

// ---------------------------------------------------------------------------------------------------------
// player <-> bank hulls: collide_ship (game.py:232-241) for the 64 envs of a wave.  cpBBIntersects reject, then "closed
// convex sets intersect" (GJK distance <= 0) evaluated as SAT over both polygons' edge normals: separated iff some axis
// has every vertex of the other polygon strictly in front.
// Per lane only the cheap rejects run; the few lanes that pass are then served one at a time by the WHOLE wave: lane
// L = 5*q + i works on (bank plane q, ship vertex/edge i) of the served env, whose pose is broadcast with v_readlane.
// The arithmetic of every product and sum is exactly the per-env formulation's; only the min/any reductions over
// vertices and planes are done with ballots instead of sequential loops.
// ---------------------------------------------------------------------------------------------------------
template <bool LDS_BANK>
__device__ __forceinline__ bool bank_narrowphase(const DevCfg &c, const double *shiptab, const double x, const double y,
                                                 const double ca, const double sa, const int rec_off, const bool live,
                                                 const int lane, const int only /* wave-uniform: hull 0 or 1, or -1 = both */,
                                                 const double *hL = nullptr /* gathered bank: this env's column of the role's header copy */,
                                                 const int ld = 0)
{
    const int wq = lane / 5, wi = lane - 5 * wq; // worker coordinates of the cooperative stage: lane L = 5*q + i
    const double w_hx = shiptab[0 * 8 + wi], w_hy = shiptab[1 * 8 + wi]; // ship vertex i (local)
    const double w_nx = shiptab[2 * 8 + wi], w_ny = shiptab[3 * 8 + wi]; // ship plane normal i (local)
    double swx[SSG_SHIP_VERTS], swy[SSG_SHIP_VERTS], sbl, sbr, sbb, sbt;
    ship_world(shiptab, ca, sa, x, y, swx, swy, sbl, sbr, sbb, sbt);
    bool colliding = false;
    unsigned nearbits = 0;
    int cnts = 0; // plane counts of both hulls, packed, so the served lane's counts travel by readlane
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        double al, ab, ar, at, cn;
        if constexpr (LDS_BANK) {
            const double2 lb = bank_at2<LDS_BANK>(c, rec_off + SSG_MAP_OFF_AABB + 4 * s + 0);
            const double2 rt = bank_at2<LDS_BANK>(c, rec_off + SSG_MAP_OFF_AABB + 4 * s + 2);
            al = lb.x; ab = lb.y; ar = rt.x; at = rt.y;
            cn = bank_at<LDS_BANK>(c, rec_off + SSG_MAP_OFF_COUNTS + s);
        } else {
            al = hL[(SSG_MAP_OFF_AABB + 4 * s + 0) * ld]; ab = hL[(SSG_MAP_OFF_AABB + 4 * s + 1) * ld];
            ar = hL[(SSG_MAP_OFF_AABB + 4 * s + 2) * ld]; at = hL[(SSG_MAP_OFF_AABB + 4 * s + 3) * ld];
            cn = hL[(SSG_MAP_OFF_COUNTS + s) * ld];
        }
        // (the reject in front of the exact SAT, on the x ranges only: a reject only has to be conservative — boxes that are disjoint
        // mean polygons that are disjoint, which the SAT below finds by itself — and the banks span the whole height of the world
        // (game_map.py:22-73), so the y half rejected nothing; without it the ship's y extent is not computed at all)
        const bool near = live & !SSG_ABL(4) & ((only < 0) | (only == s)) & (sbl <= ar) & (al <= sbr);
        (void)ab; (void)at; (void)sbb; (void)sbt;
        nearbits |= near ? (1u << s) : 0u;
        cnts |= ((int)cn) << (8 * s);
    }
    // Stage 1, per lane, all near lanes at once: is some BANK plane a separating axis (all five ship vertices
    // strictly in front)?  That settles almost every ship that is merely close to a bank; the loop ends as soon
    // as every near lane of the wave has found its plane.
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const bool near_s = (nearbits >> s) & 1u;
        if (!__any(near_s)) continue;
        const int cnt = (cnts >> (8 * s)) & 0xFF;
        bool sep = false;
        if constexpr (LDS_BANK) {
            for (int j = 0; __any(near_s & !sep & (j < cnt)); ++j) {
                if (near_s & !sep & (j < cnt)) { // (only the lanes still looking gather a plane: the LDS pipe is the busiest unit)
                    const int q = rec_off + SSG_MAP_OFF_PLANES + s * (SSG_MAX_HULL * SSG_PLANE_DOUBLES) + SSG_PLANE_DOUBLES * j;
                    const double2 nn = bank_at2<LDS_BANK>(c, q + 2);
                    const double nx = nn.x, ny = nn.y;
                    const double v0n = bank_at<LDS_BANK>(c, q + 4);
                    bool allfront = true;
#pragma unroll
                    for (int i = 0; i < SSG_SHIP_VERTS; ++i) allfront = allfront & ((nx * swx[i] + ny * swy[i]) > v0n);
                    sep = allfront;
                }
            }
        } else {
            // the record is gathered from L2 / HBM: four planes per trip (one dependent round trip instead of four; eight per trip spill registers at 256 envs per workgroup); which plane
            // separates does not matter, only whether one does
            constexpr int kNpChunk = 4;
            for (int j0 = 0; __any(near_s & !sep & (j0 < cnt)); j0 += kNpChunk) {
                if (near_s & !sep & (j0 < cnt)) {
                    double nx[kNpChunk], ny[kNpChunk], v0n[kNpChunk];
#pragma unroll
                    for (int u = 0; u < kNpChunk; ++u) {
                        const int j = (j0 + u < SSG_MAX_HULL) ? j0 + u : 0;
                        const int q = rec_off + SSG_MAP_OFF_PLANES + s * (SSG_MAX_HULL * SSG_PLANE_DOUBLES) + SSG_PLANE_DOUBLES * j;
                        const double2 nn = bank_at2<LDS_BANK>(c, q + 2);
                        nx[u] = nn.x; ny[u] = nn.y;
                        v0n[u] = bank_at<LDS_BANK>(c, q + 4);
                    }
#pragma unroll
                    for (int u = 0; u < kNpChunk; ++u) {
                        bool allfront = j0 + u < cnt;
#pragma unroll
                        for (int i = 0; i < SSG_SHIP_VERTS; ++i) allfront = allfront & ((nx[u] * swx[i] + ny[u] * swy[i]) > v0n[u]);
                        sep = sep | allfront;
                    }
                }
            }
        }
        nearbits = sep ? (nearbits & ~(1u << s)) : nearbits;
    }
    // Stage 2, wave-cooperative, for the few lanes still unresolved (mostly real collisions): the full SAT over
    // both polygons' edge normals, lane L = 5*q + i on (bank plane q, ship vertex i) of the served env.
    unsigned long long todo = __ballot(nearbits != 0u);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const double bx = readlane_f64(x, src), by = readlane_f64(y, src);
        const double bca = readlane_f64(ca, src), bsa = readlane_f64(sa, src);
        const int boff = __builtin_amdgcn_readlane(rec_off, src);
        const unsigned bnear = (unsigned)__builtin_amdgcn_readlane((int)nearbits, src);
        const int bcnts = __builtin_amdgcn_readlane(cnts, src);
        const double svx = bca * w_hx + (-bsa) * w_hy + bx, svy = bsa * w_hx + bca * w_hy + by;
        const double snx_ = bca * w_nx + (-bsa) * w_ny, sny_ = bsa * w_nx + bca * w_ny;
        const double off_i = snx_ * svx + sny_ * svy;
        bool col = false;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (!(bnear & (1u << s))) continue; // wave-uniform
            const int cnt = (bcnts >> (8 * s)) & 0xFF;
            const bool valid = (lane < 60) & (wq < cnt);
            const int q = boff + SSG_MAP_OFF_PLANES + s * (SSG_MAX_HULL * SSG_PLANE_DOUBLES) +
                          SSG_PLANE_DOUBLES * ((wq < SSG_MAX_HULL) ? wq : 0);
            const double v0x = bank_at<LDS_BANK>(c, q + 0), v0y = bank_at<LDS_BANK>(c, q + 1);
            const double nx = bank_at<LDS_BANK>(c, q + 2), ny = bank_at<LDS_BANK>(c, q + 3);
            const double v0n = bank_at<LDS_BANK>(c, q + 4);
            const bool frontA = (nx * svx + ny * svy) > v0n;           // ship vertex i in front of bank plane q
            const bool frontB = (snx_ * v0x + sny_ * v0y) > off_i;     // bank vertex q in front of ship plane i
            const unsigned long long mV = __ballot(valid);
            const unsigned long long mA = __ballot(valid & frontA);
            const unsigned long long missB = mV & ~__ballot(valid & frontB);
            const unsigned long long P = 0x0084210842108421ull;       // bit 5q, q = 0..11
            // axis = bank plane q: all five (q,i) bits set
            const unsigned long long allA = mA & (mA >> 1) & (mA >> 2) & (mA >> 3) & (mA >> 4) & P;
            // axis = ship plane i: no valid (q,i) bit missing
            bool sepB = false;
#pragma unroll
            for (int i = 0; i < SSG_SHIP_VERTS; ++i) sepB = sepB | (((missB >> i) & P) == 0ull);
            const bool separated = (allA != 0ull) | sepB;
            col = col | !separated;
        }
        colliding = (lane == src) ? col : colliding;
    }
    return colliding;
}

// ---------------------------------------------------------------------------------------------------------
// The step kernel.  A workgroup of 4*EPW threads serves EPW envs with four wave ROLES (role = wave / (EPW/64)):
//   role 0  LIDAR-lo : LiDAR.query beams [0, NB0); narrowphase (collide_ship) against the left bank
//   role 1  LIDAR-hi : LiDAR.query beams [NB0, NB); narrowphase against the right bank
//   role 2  OBSERVER : sticky-lidar merge and the observation rows (__add_states); its registers carry the previous
//                      frame from step to step
//   role 3  BODY     : handle_discrete_action, integrator, ship transform, goal-circle narrowphase, nearest goals,
//                      reward / done, statistics; its registers carry the body state from step to step
// A lone wave on a SIMD issues FP64 at half rate and runs latency-bound, and 65 536 envs are only one wave per SIMD, so
// each env's step is cut into four instruction streams on four co-resident waves per SIMD.  They are PIPELINED through
// LDS.  The four waves of a 64-env TILE exchange data only among themselves, through two per-tile rendezvous words (no
// workgroup barrier after barrier 0, which publishes the staged bank):
//   pose hand-over: role 3 integrates and publishes the post-step pose of step k (`sync_ready` = k+1) once its three
//     consumers have acknowledged pose k-1 (`sync_ack`);
//   rendezvous B(k) (`sync_bar`): roles 0 / 1 have collided that pose with the banks, role 3
//     has done the goals and its share of is_done, the lidar roles have delivered step k's readings.
//   After B(k) three things run side by side: role 3 closes the step (statistics, reset) and starts the next one; role 2
//   writes reward / done / flags and the observation rows of step k; roles 0/1 run step k+1's lidar query.
//   LiDAR.query sees the pre-step pose, which is step k's post-step pose, or the spawn pose if the env is done — roles
//   0-2 read role 0's and role 3's done bits after B(k) and decide that themselves.  Lidar results are handed over in
//   LDS buffers indexed by the parity of the step.  (The first step's query runs right after barrier 0, from the state
//   columns.)  Inside a fused launch the state lives in registers: the state columns in HBM are read by the first step
//   and written back by the last one only; obs / reward / done / flags are written by every step.
// A single-step launch (K = 1) runs the same code with nothing to overlap: ~6 k cycles of prologue (state columns,
// 74 KB of bank by LDS-DMA), ~12 k cycles with the four roles sharing the SIMD, a ~6 k-cycle observer tail (role 3 writes
// the outputs and the lidar waves the sticky columns beside it), and the write-back of the ~25 MB it dirtied at the
// end of the kernel.
// ---------------------------------------------------------------------------------------------------------
template <int NB, int EPW, bool LDS_BANK, bool EXACT, bool DYN>
__global__ __launch_bounds__(tile_roles(EPW, DYN) * EPW) void step_kernel(const DevCfg c, const int32_t *__restrict__ actions_kn,
                                                       double *__restrict__ obs, double *__restrict__ reward_out,
                                                       uint8_t *__restrict__ done_out, uint8_t *__restrict__ flags_out,
                                                       const int K_launch, const long long traj)
{
    // (config 4 steps its dyn kernels between any two steps: its launches are single steps, and the compiler is told so — the
    // body role's queue entry, a returning atomic, otherwise counts as live around the step loop and is spilled, i.e. waited
    // for, the moment it is issued)
    const int K = DYN ? 1 : K_launch;
    SSG_CLOCK_BEGIN();
    // K consecutive steps in one launch (K = 1 for ssg_step): the bank is staged once and role 3 keeps the body state in
    // registers.  Step k reads actions_kn + k*n_envs and writes its obs / reward / done / flags `k * traj` env rows past the
    // buffers' starts: traj = 0 rewrites the same [n_envs] rows every step (ssg_rollout), traj >= n_envs lays the steps of
    // the launch out as a trajectory [K][traj] (ssg_rollout_traj: what train/random.py:14-27 consumes, every step).
    constexpr int NB0 = nb_lo(NB);
    constexpr int NR = tile_roles(EPW, DYN);                   // waves per tile: 4, or 6 (roles 4 / 5: collide_ship against bank hull 0 / 1)
    const int role = threadIdx.x / EPW;                        // wave-uniform (EPW is a multiple of 64)
    const int tl = threadIdx.x - role * EPW;                   // env slot inside the workgroup
    const int e = blockIdx.x * EPW + tl;
    const bool live = e < c.n_envs;
    const int el_ = live ? e : 0; // lanes past n_envs stay active as workers; they carry env 0 and store nothing
    const size_t np = (size_t)c.n_pad;
    const int lane = threadIdx.x & 63;

    double *__restrict__ colX = c.f64cols + COL_X * np;
    double *__restrict__ colY = c.f64cols + COL_Y * np;
    double *__restrict__ colVX = c.f64cols + COL_VX * np;
    double *__restrict__ colVY = c.f64cols + COL_VY * np;
    double *__restrict__ colA = c.f64cols + COL_A * np;
    double *__restrict__ colW = c.f64cols + COL_W * np;
    double *__restrict__ colCum = c.f64cols + COL_CUM * np;
    double *__restrict__ colLid = c.f64cols + COL_LIDAR * np;
    int32_t *__restrict__ colRud = c.i32cols + ICOL_RUDDER * np;
    int32_t *__restrict__ colStep = c.i32cols + ICOL_STEP * np;
    int32_t *__restrict__ colMap = c.i32cols + ICOL_MAP * np;

    const int bank_bytes = LDS_BANK ? ((c.n_maps * (SSG_MAP_STRIDE * 8) + 15) & ~15) : 0; // 16-byte aligned tables follow
    char *lds_fixed = reinterpret_cast<char *>(lds_base()) + bank_bytes;
    double *beamtab = reinterpret_cast<double *>(lds_fixed);
    double *shiptab = reinterpret_cast<double *>(lds_fixed + kBeamTabBytes);
    double *traffictab = reinterpret_cast<double *>(lds_fixed + kBeamTabBytes + kShipTabBytes); // [3][4][8], config 4
    double *pose = reinterpret_cast<double *>(lds_fixed + kBeamTabBytes + kShipTabBytes + (DYN ? kTrafficTabBytes : 0)); // [7][EPW]
    int *posem = reinterpret_cast<int *>(pose + kPoseDoubles * EPW);                         // [EPW]
    int *poser = posem + EPW;                                                                // [EPW]
    unsigned *gres = reinterpret_cast<unsigned *>(poser + EPW);                              // [2 parities][EPW]
    unsigned *gdone = gres + 2 * EPW;                                                        // [2 parities][EPW]
    unsigned *gtraf = gdone + 2 * EPW;                                                       // [2][EPW] config 4 only: the player touches a traffic ship (one word per lidar role -> all; DYN launches are single steps)
    unsigned *sync_ready = gtraf + (DYN ? 2 * EPW : 0);                                      // [EPW/64]
    unsigned *sync_ack = sync_ready + EPW / 64;                                              // [EPW/64]
    unsigned *sync_bar = sync_ack + EPW / 64;                                                // [EPW/64] (+ one pad word each)
    char *goal_scratch0 = reinterpret_cast<char *>(sync_bar + 2 * (EPW / 64));
    char *traffic_scratch0 = goal_scratch0 + (EPW / 64) * kGoalScratchBytes;
    char *scratch0 = lds_fixed + lds_fixed_bytes(EPW, DYN);
    char *tile_base = scratch0 + (tl >> 6) * lds_tile_bytes(NB); // this env tile's lidar buffers
    // gathered bank: the record heads in LDS (see lds_hdr_bytes); hL = this env's column of the lidar role's copy, hG = of the goals
    double *hdr0 = reinterpret_cast<double *>(scratch0 + (EPW / 64) * lds_tile_bytes(NB));
    double *hL = LDS_BANK ? nullptr : hdr0 + ((role & 1) + (role >= 4 ? 2 : 0)) * (kHdrLidar * EPW) + tl;
    double *hG = LDS_BANK ? nullptr : hdr0 + (NR - 2) * kHdrLidar * EPW + tl;
    auto load_hdr_lidar = [&](int rec_off) { // record doubles 0..9 -> this role's copy (five 16-byte gathers)
#pragma unroll
        for (int f = 0; f < kHdrLidar / 2; ++f) { const double2 v = bank_at2<LDS_BANK>(c, rec_off + 2 * f); hL[(2 * f) * EPW] = v.x; hL[(2 * f + 1) * EPW] = v.y; }
    };
    auto load_hdr_goals = [&](int rec_off) { // record doubles 10..21 -> the goal columns (six 16-byte gathers)
#pragma unroll
        for (int f = 0; f < kHdrGoals / 2; ++f) { const double2 v = bank_at2<LDS_BANK>(c, rec_off + SSG_MAP_OFF_GOALS + 2 * f); hG[(2 * f) * EPW] = v.x; hG[(2 * f + 1) * EPW] = v.y; }
    };

#ifdef SSG_STAMPS
    unsigned long long stamp_[16] = {};
#endif
    SSG_STAMP(8);

    // The bank's LDS-DMA goes out FIRST: the table entries below are vector loads from the kernel-argument segment (indexed by
    // the thread id) that the wave waits for before its LDS write — a memory round trip that used to sit in front of the staging
    // of the waves that own a table (three of a workgroup's sixteen), i.e. in front of barrier 0 of every launch.
    double bank_tail_v = 0.0;
    int bank_tail_o = -1;
    if (LDS_BANK && !SSG_ABL(8)) stage_bank_lds<NR * EPW>(c.bank, c.n_maps * (SSG_MAP_STRIDE * 8), bank_tail_v, bank_tail_o);
    // small constant tables (written once per workgroup, read after the first barrier).  Their entries are vector loads from the
    // kernel-argument segment, indexed by the thread id: they are only REQUESTED here and written to LDS by flush_tables(), which
    // every role calls right before its wait in front of barrier 0 — written here, each cost its wave a memory round trip of its own
    // in front of the state loads.
    double tv[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (threadIdx.x < 2 * SSG_MAX_BEAMS)
        tv[0] = (threadIdx.x < SSG_MAX_BEAMS) ? c.beam_cos[threadIdx.x & (SSG_MAX_BEAMS - 1)] : c.beam_sin[threadIdx.x & (SSG_MAX_BEAMS - 1)];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + SSG_SHIP_VERTS) {
        const int i = threadIdx.x - 64, ip = (i == 0) ? (SSG_SHIP_VERTS - 1) : (i - 1);
        tv[0] = c.hull[2 * i]; tv[1] = c.hull[2 * i + 1]; tv[2] = c.nrm[2 * i]; tv[3] = c.nrm[2 * i + 1]; tv[4] = c.hull[2 * ip]; tv[5] = c.hull[2 * ip + 1];
    }
    if constexpr (DYN) {
        if (threadIdx.x >= 160 && threadIdx.x < 160 + SSG_N_TRAFFIC * SSG_SHIP_VERTS) {
            const int kk = (threadIdx.x - 160) / SSG_SHIP_VERTS, i = (threadIdx.x - 160) % SSG_SHIP_VERTS;
            tv[0] = c.thull[kk][2 * i]; tv[1] = c.thull[kk][2 * i + 1]; tv[2] = c.tnrm[kk][2 * i]; tv[3] = c.tnrm[kk][2 * i + 1];
        }
    }
    auto flush_tables = [&]() {
        if (bank_tail_o >= 0) *reinterpret_cast<double *>(reinterpret_cast<char *>(lds_base()) + bank_tail_o) = bank_tail_v; // the staged bank's last < 1 KiB
        if (threadIdx.x < 2 * SSG_MAX_BEAMS) beamtab[threadIdx.x] = tv[0];
        if (threadIdx.x >= 64 && threadIdx.x < 64 + SSG_SHIP_VERTS) {
            const int i = threadIdx.x - 64;
            shiptab[0 * 8 + i] = tv[0]; shiptab[1 * 8 + i] = tv[1];   // vertex i
            shiptab[2 * 8 + i] = tv[2]; shiptab[3 * 8 + i] = tv[3];   // plane normal i
            shiptab[4 * 8 + i] = tv[4]; shiptab[5 * 8 + i] = tv[5];   // vertex i-1 (edge start)
        }
        if constexpr (DYN) {
            if (threadIdx.x >= 160 && threadIdx.x < 160 + SSG_N_TRAFFIC * SSG_SHIP_VERTS) {
                const int kk = (threadIdx.x - 160) / SSG_SHIP_VERTS, i = (threadIdx.x - 160) % SSG_SHIP_VERTS;
                traffictab[(kk * 4 + 0) * 8 + i] = tv[0]; traffictab[(kk * 4 + 1) * 8 + i] = tv[1];
                traffictab[(kk * 4 + 2) * 8 + i] = tv[2]; traffictab[(kk * 4 + 3) * 8 + i] = tv[3];
            }
        }
    };
    if (threadIdx.x == 128) {
        // lidar origin of a freshly reset ship (angle 0: cpvforangle(0) = (1, 0)): pos + half the world AABB extents,
        // by the very operations ship_world() runs
        double bl = INFINITY, br = -INFINITY, bb = INFINITY, bt = -INFINITY;
        for (int i = 0; i < SSG_SHIP_VERTS; ++i) {
            const double hx = c.hull[2 * i], hy = c.hull[2 * i + 1];
            const double wx = 1.0 * hx + (-0.0) * hy + c.spawn_x, wy = 0.0 * hx + 1.0 * hy + c.spawn_y;
            bl = dmin(bl, wx); br = dmax(br, wx); bb = dmin(bb, wy); bt = dmax(bt, wy);
        }
        shiptab[0 * 8 + 6] = c.spawn_x + (br - bl) / 2;
        shiptab[1 * 8 + 6] = c.spawn_y + (bt - bb) / 2;
    }
    if (threadIdx.x >= 192 && threadIdx.x < 192 + 4 * (EPW / 64)) sync_ready[threadIdx.x - 192] = 0u; // ready, ack, bar words
    const bool auto_reset = (c.flags & SSG_FLAG_AUTO_RESET) != 0u;
    const int tile = tl >> 6;
    // Pose hand-over, per tile (the four waves of a tile are the only ones that exchange anything): role 3 publishes pose k
    // once the three consumers have acknowledged pose k-1, and raises `ready` to k+1; a consumer waits for that, copies what it
    // needs into registers and acknowledges.  Replaces a second workgroup barrier per step, at which role 3 waited ~1.5 k
    // cycles for the slowest of twelve waves it has no business with.
    auto wait_pose = [&](int k) {
        while (__hip_atomic_load(&sync_ready[tile], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (unsigned)(k + 1))
            __builtin_amdgcn_s_sleep(2);
    };
    auto ack_pose = [&]() {
        if (lane == 0) __hip_atomic_fetch_add(&sync_ack[tile], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // The lidar waves say when the query of the launch's LAST step (a one-step launch's only one) has left its result keys in LDS,
    // so that the observer can build that step's rows BEFORE its rendezvous (see `spec` in the observer).  The fourth word of the
    // tile's sync block (zeroed with the others); raised once per launch by each of the two lidar waves.
    unsigned *sync_q = sync_bar + EPW / 64;
    auto query_done = [&]() {
        if (lane == 0) __hip_atomic_fetch_add(&sync_q[tile], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto wait_queries = [&]() {
        while (__hip_atomic_load(&sync_q[tile], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 2u)
            __builtin_amdgcn_s_sleep(1);
    };
    // The per-step rendezvous "B" of the tile's four waves (collide_ship, role 3's done bits and nearest goal, and this
    // step's lidar results are in; the result buffer of the other parity is free): an arrival counter in LDS instead of a
    // workgroup barrier — the four tiles of a workgroup share nothing but the staged bank, and at s_barrier each waited for
    // the slowest of the other three twice per step (~2.4 k of a step's 15 k cycles).
    // (s_wakeup by whoever completes a rendezvous, with longer sleeps between the waiters' polls, measured 0.6 % faster — and
    // raised GPU memory-access faults in the diagnostic builds that also execute s_memtime: not used.)
    auto tile_barrier = [&](int k) {
        if (lane == 0) __hip_atomic_fetch_add(&sync_bar[tile], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(&sync_bar[tile], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)NR * (unsigned)(k + 1))
            __builtin_amdgcn_s_sleep(1);
    };

    // config 4: collide_ship (game.py:232-241) against the traffic ships, where this step's cpSpaceStep (the dyn kernels, just
    // before this launch) left them: cpBBIntersects, then "touching counts" SAT over both hulls' edge normals.  Per lane only the
    // rejects run — no vertex of ship k's hull is further than its hull radius from its body position, so a player whose world
    // box is further than that from the position cannot touch it; then the exact box test — and the (lane, ship) pairs that pass
    // go into a pair queue of the tile and are served 12 at a time by the whole wave:
    // lane L = 5*p + i takes edge normal i of BOTH hulls of pair p.  Products and sums are the per-env formulation's.  (Rounds
    // 2-3 ran this test in the dyn kernels: on every wave of the full step's chain, and in extra workgroups for resting envs.
    // It runs on the two lidar waves, after their bank hulls — ship 0 on lidar-lo, ships 1 and 2 on lidar-hi.  (All three on
    // lidar-hi: that wave reached the rendezvous ~4 k cycles after everybody else.  On the OBSERVER wave, idle between the pose
    // hand-over and the rendezvous: the rendezvous came 3 k cycles earlier and the step 0.6 us later — holding the ships' 12
    // doubles next to its two frames, that wave spilled 32 registers into its tail, which is the end of the launch.)
    // Ships [K0, K1) of the env's traffic; `part` = which of the tile's two pair queues and result words (one per lidar role).
    auto traffic_collide = [&](auto k0_, auto k1_, const double *tpre, double x, double y, double ca, double sa, int part,
                               unsigned short *tq /* pair queue: 64 x (K1 - K0) */, unsigned *tw /* 64 hit words */) {
        constexpr int K0 = decltype(k0_)::value, K1 = decltype(k1_)::value;
        const int wq = lane / 5, wi = lane - 5 * wq;
        const double w_hx = shiptab[0 * 8 + wi], w_hy = shiptab[1 * 8 + wi], w_nx = shiptab[2 * 8 + wi], w_ny = shiptab[3 * 8 + wi];
        double sbl, sbr, sbb, sbt;
        {
            double swx[SSG_SHIP_VERTS], swy[SSG_SHIP_VERTS];
            ship_world(shiptab, ca, sa, x, y, swx, swy, sbl, sbr, sbb, sbt);
        }
        tw[lane] = 0u;
        int n_tp = 0;
#pragma unroll
        for (int kk = K0; kk < K1; ++kk) {
            const double tx = tpre[4 * kk + 0], ty = tpre[4 * kk + 1];
            const double dx = dmax(dmax(sbl - tx, tx - sbr), 0.0), dy = dmax(dmax(sbb - ty, ty - sbt), 0.0);
            bool cand = false;
            if (live & ((dx * dx + dy * dy) <= c.dyn_reach2[kk])) {
                const double tca = tpre[4 * kk + 2], tsa = tpre[4 * kk + 3];
                double bl = INFINITY, br = -INFINITY, bb = INFINITY, bt = -INFINITY;
#pragma unroll
                for (int i = 0; i < SSG_SHIP_VERTS; ++i) {
                    const double gx = traffictab[(kk * 4 + 0) * 8 + i], gy = traffictab[(kk * 4 + 1) * 8 + i];
                    const double wx = tca * gx + (-tsa) * gy + tx, wy = tsa * gx + tca * gy + ty;
                    bl = dmin(bl, wx); br = dmax(br, wx); bb = dmin(bb, wy); bt = dmax(bt, wy);
                }
                cand = (sbl <= br) & (bl <= sbr) & (sbb <= bt) & (bb <= sbt); // cpBBIntersects(player, ship k)
            }
            const unsigned long long m = __ballot(cand);
            const int pos = n_tp + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (cand) tq[pos] = (unsigned short)(lane | (kk << 6));
            n_tp += __popcll(m);
        }
        for (int base = 0; base < n_tp; base += 12) {
            const int p = base + wq;
            const bool valid = (lane < 60) & (p < n_tp);
            const unsigned code = tq[valid ? p : 0];
            const int src = code & 63, kk = code >> 6;
            const double bx = __shfl(x, src), by = __shfl(y, src), bca = __shfl(ca, src), bsa = __shfl(sa, src);
            // the source lane holds its env's ships
            double tx = __shfl(tpre[4 * K0 + 0], src), ty = __shfl(tpre[4 * K0 + 1], src), tca = __shfl(tpre[4 * K0 + 2], src), tsa = __shfl(tpre[4 * K0 + 3], src);
#pragma unroll
            for (int h = K0 + 1; h < K1; ++h) {
                const double a0 = __shfl(tpre[4 * h + 0], src), a1 = __shfl(tpre[4 * h + 1], src);
                const double a2 = __shfl(tpre[4 * h + 2], src), a3 = __shfl(tpre[4 * h + 3], src);
                tx = (kk == h) ? a0 : tx; ty = (kk == h) ? a1 : ty; tca = (kk == h) ? a2 : tca; tsa = (kk == h) ? a3 : tsa;
            }
            const double *tt = traffictab + kk * 32;
            bool sep;
            {   // axis = the player's edge normal i: every vertex of the ship strictly in front of the player's vertex i?
                const double nx = bca * w_nx + (-bsa) * w_ny, ny = bsa * w_nx + bca * w_ny;
                const double vx_ = bca * w_hx + (-bsa) * w_hy + bx, vy_ = bsa * w_hx + bca * w_hy + by;
                const double off = nx * vx_ + ny * vy_;
                double mn = INFINITY;
#pragma unroll
                for (int j = 0; j < SSG_SHIP_VERTS; ++j) {
                    const double gx = tt[0 * 8 + j], gy = tt[1 * 8 + j];
                    const double qx = tca * gx + (-tsa) * gy + tx, qy = tsa * gx + tca * gy + ty;
                    mn = dmin(mn, nx * qx + ny * qy);
                }
                sep = mn > off;
            }
            {   // axis = the ship's edge normal i
                const double lnx = tt[2 * 8 + wi], lny = tt[3 * 8 + wi], lvx = tt[0 * 8 + wi], lvy = tt[1 * 8 + wi];
                const double nx = tca * lnx + (-tsa) * lny, ny = tsa * lnx + tca * lny;
                const double vx_ = tca * lvx + (-tsa) * lvy + tx, vy_ = tsa * lvx + tca * lvy + ty;
                const double off = nx * vx_ + ny * vy_;
                double mn = INFINITY;
#pragma unroll
                for (int j = 0; j < SSG_SHIP_VERTS; ++j) {
                    const double hx = shiptab[0 * 8 + j], hy = shiptab[1 * 8 + j];
                    const double qx = bca * hx + (-bsa) * hy + bx, qy = bsa * hx + bca * hy + by;
                    mn = dmin(mn, nx * qx + ny * qy);
                }
                sep |= mn > off;
            }
            const unsigned long long ms = __ballot(valid & sep);
            const bool separated = ((ms >> (5 * wq)) & 31ull) != 0ull;
            if (valid & (wi == 0) & !separated) atomicOr(&tw[src], 1u);
        }
        gtraf[part * EPW + tl] = (tw[lane] != 0u) ? 1u : 0u;
    };
    // (DYN launches are single steps: gtraf's two parities serve as the two lidar roles' result words)
    auto traffic_hit = [&](int) -> unsigned { return gtraf[tl] | gtraf[EPW + tl]; };
    if (role < 2) {
        // =====================================================================================================
        // ROLES 0 / 1: LiDAR.query on the PRE-step pose (models.py:39-76; game.py:193 runs it before space.step)
        // =====================================================================================================
        unsigned short *queue = reinterpret_cast<unsigned short *>(tile_base + 2 * lds_res_bytes(NB) + role * lds_queue_bytes(NB0));
        const int b_first = role ? NB0 : 0, b_count = role ? (NB - NB0) : NB0;
        // the first step's pre-step pose comes from the state columns
        double ca, sa, cx, cy;
        int map_id;
        double tpre[DYN ? 4 * SSG_N_TRAFFIC : 1] = {};
        {
            const double x = colX[el_], y = colY[el_], ang = colA[el_];
            map_id = colMap[el_];
            if constexpr (!LDS_BANK) load_hdr_lidar(map_id * SSG_MAP_STRIDE);
            { const double2 sc = sincos_call(ang); sa = sc.x; ca = sc.y; } // body->transform rotation
            if (role == 0) { pose[2 * EPW + tl] = ca; pose[3 * EPW + tl] = sa; } // -> role 3: the first step's thrust direction
            flush_tables();
            if (LDS_BANK) __builtin_amdgcn_s_waitcnt(0); // vmcnt(0): the LDS-DMA writes of this wave have landed
            __syncthreads();                             // barrier 0: bank + tables visible
            SSG_STAMP(9);
            if constexpr (DYN) {
                // The traffic ships' positions and rotations of this step (the dyn kernels' output), asked for in one batch under
                // the first query: read where they are used, they were two dependent round trips per ship between the bank hull
                // and the rendezvous.  (Asked for BEFORE barrier 0, with the state, they delayed the barrier for every role: the
                // whole grid starts at once and the extra megabytes queue up behind the bank's staging.)
                {
                    const double *tcol = c.dyn_f64 + (size_t)DC_TRAFFIC * np, *trot = c.dyn_f64 + (size_t)DC_TROT * np;
#pragma unroll
                    for (int kk = 0; kk < SSG_N_TRAFFIC; ++kk) {
                        if ((kk == 0) != (role == 0)) continue; // ship 0: lidar-lo; ships 1, 2: lidar-hi
                        tpre[4 * kk + 0] = tcol[(size_t)(9 * kk) * np + el_]; tpre[4 * kk + 1] = tcol[(size_t)(9 * kk + 1) * np + el_];
                        tpre[4 * kk + 2] = trot[(size_t)(2 * kk) * np + el_]; tpre[4 * kk + 3] = trot[(size_t)(2 * kk + 1) * np + el_];
                    }
                }
            }
            double swx[SSG_SHIP_VERTS], swy[SSG_SHIP_VERTS], bl, br, bb, bt;
            ship_world(shiptab, ca, sa, x, y, swx, swy, bl, br, bb, bt);
            cx = x + (br - bl) / 2; // lidar origin: pos + half the world AABB extents (models.py:51-53)
            cy = y + (bt - bb) / 2;
        }
        // the first step's query needs nothing from role 3: it runs while role 3 integrates
        lidar_query<NB, LDS_BANK, EXACT>(c, reinterpret_cast<unsigned long long *>(tile_base), queue, beamtab, b_first, b_count,
                                         cx, cy, ca, sa, map_id * SSG_MAP_STRIDE, live, lane, hL, EPW);
        if (K == 1) query_done();
        for (int k = 0; k < K; ++k) {
            if constexpr (NR == 4) wait_pose(k); // role 3 has published this step's post-step pose (six roles: collide_ship is roles 4 / 5's,
                                                 // this wave goes straight to the rendezvous and picks the pose up behind it)
            if constexpr (DYN) {
#pragma unroll
                for (int i = 0; i < 4 * SSG_N_TRAFFIC; ++i) asm volatile("" : "+v"(tpre[i]));
            }
            SSG_STAMP_K(0);
            double nca = 0.0, nsa = 0.0, ncx = 0.0, ncy = 0.0, npx = 0.0, npy = 0.0;
            int nmap = 0;
            if constexpr (NR == 4) {
                nca = pose[2 * EPW + tl]; nsa = pose[3 * EPW + tl];
                ncx = pose[4 * EPW + tl]; ncy = pose[5 * EPW + tl];
                npx = pose[0 * EPW + tl]; npy = pose[1 * EPW + tl];
                nmap = posem[tl];
                ack_pose();
                // collide_ship of this step, one bank hull per lidar role (role 2 is writing the previous step's rows; in a launch's
                // first step the lidar waves have just finished the first query and would idle until B)
                reinterpret_cast<unsigned short *>(gres)[2 * ((k & 1) * EPW + tl) + role] =
                        bank_narrowphase<LDS_BANK>(c, shiptab, npx, npy, nca, nsa, nmap * SSG_MAP_STRIDE, live, lane, role, hL, EPW) ? 1 : 0;
            }
            SSG_STAMP_K(4);
            if constexpr (DYN) {
                // lidar-lo's queue and hit words: its (beam, hull) pair queue, idle between two queries (64 + 128 of its >= 192 half-words)
                static_assert(lds_queue_bytes(1) >= 64 * 2 + 64 * 4, "lidar-lo's pair queue holds ship 0's pair queue and hit words");
                unsigned short *tq1 = reinterpret_cast<unsigned short *>(traffic_scratch0 + (tl >> 6) * kTrafficScratchBytes);
                if (role == 0) traffic_collide(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, tpre, npx, npy, nca, nsa, 0,
                                               queue, reinterpret_cast<unsigned *>(queue + 64));
                else traffic_collide(std::integral_constant<int, 1>{}, std::integral_constant<int, SSG_N_TRAFFIC>{}, tpre, npx, npy, nca, nsa, 1,
                                     tq1, reinterpret_cast<unsigned *>(tq1 + 64 * SSG_N_TRAFFIC));
            }

            SSG_STAMP_K(3);
            tile_barrier(k); // rendezvous B(k): collide_ship and role 3's done bits are in
            SSG_STAMP_K(1);
            if constexpr (NR == 6) {
                if (k + 1 < K) { // pose k is still in its slots: role 3 publishes pose k+1 only once every consumer has acknowledged this one
                    nca = pose[2 * EPW + tl]; nsa = pose[3 * EPW + tl];
                    ncx = pose[4 * EPW + tl]; ncy = pose[5 * EPW + tl];
                    nmap = posem[tl];
                    ack_pose();
                }
            }
            if (k + 1 < K) {
                // step k+1's pre-step pose: this step's post-step pose, or ShipGame.reset's spawn pose on the next map
                const bool rs = auto_reset & ((gres[(k & 1) * EPW + tl] | (gdone[(k & 1) * EPW + tl] & 1u) | (DYN ? traffic_hit(k) : 0u)) != 0u);
                ca = rs ? 1.0 : nca; sa = rs ? 0.0 : nsa;
                cx = rs ? shiptab[0 * 8 + 6] : ncx; cy = rs ? shiptab[1 * 8 + 6] : ncy;
                map_id = rs ? next_map(c, nmap, el_) : nmap;
                if constexpr (!LDS_BANK) {
                    if (rs) load_hdr_lidar(map_id * SSG_MAP_STRIDE); // only the lanes whose env moved to its next record gather
                }
                lidar_query<NB, LDS_BANK, EXACT>(c, reinterpret_cast<unsigned long long *>(tile_base + ((k + 1) & 1) * lds_res_bytes(NB)),
                                                 queue, beamtab, b_first, b_count, cx, cy, ca, sa, map_id * SSG_MAP_STRIDE, live, lane, hL, EPW);
                if (k + 1 == K - 1) query_done(); // the launch's last query: the observer builds the last step's rows before its rendezvous
            }
            SSG_STAMP_K(2);
        }
        if (K == 1 && live) {
            // A single-step launch: the sticky readings' columns hold exactly the previous frame's values, so "a miss keeps
            // the previous reading" (models.py:68-72) is "a miss stores nothing".  The lidar waves, idle after the step's only
            // rendezvous, store their own beams' hits (or the -1 of a fresh episode) while the observer builds the rows.
            const bool rs = auto_reset & ((gres[tl] | (gdone[tl] & 1u) | (DYN ? traffic_hit(0) : 0u)) != 0u);
            const unsigned long long *rk = reinterpret_cast<const unsigned long long *>(tile_base); // parity 0
            for (int kb = 0; kb < b_count; ++kb) {
                const int i = b_first + kb;
                const unsigned long long key = rk[i * 64 + lane];
                const double hitd = __longlong_as_double((long long)(key & 0x7FFFFFFFFFFFFFFFull));
                if (rs | (key != kLidarMiss)) st_out(&colLid[(size_t)i * np + el_], rs ? -1.0 : hitd);
            }
        }
        SSG_STAMP(10);
        SSG_STAMP_FLUSH(3);
        SSG_CLOCK_END(); // (thread 0 is a lidar wave's: its role ends with the launch's last step)
        return;
    }

    if constexpr (NR == 6) {
        if (role >= 4) {
            // =================================================================================================
            // ROLES 4 / 5 (six-role tiles): collide_ship (game.py:232-241) against bank hull 0 / 1, every step, on the pose
            // role 3 has just published — beside the lidar waves' query instead of behind it
            // =================================================================================================
            const int h = role - 4;
            int map0 = colMap[el_];
            if constexpr (!LDS_BANK) load_hdr_lidar(map0 * SSG_MAP_STRIDE); // (this role's own copy of the record's head)
            asm volatile("" : "+v"(map0));
            flush_tables();
            if (LDS_BANK) __builtin_amdgcn_s_waitcnt(0);
            __syncthreads(); // barrier 0
            SSG_STAMP(9);
            for (int k = 0; k < K; ++k) {
                wait_pose(k);
                SSG_STAMP_K(0);
                const double nca = pose[2 * EPW + tl], nsa = pose[3 * EPW + tl];
                const double npx = pose[0 * EPW + tl], npy = pose[1 * EPW + tl];
                const int nmap = posem[tl];
                ack_pose();
                reinterpret_cast<unsigned short *>(gres)[2 * ((k & 1) * EPW + tl) + h] =
                        bank_narrowphase<LDS_BANK>(c, shiptab, npx, npy, nca, nsa, nmap * SSG_MAP_STRIDE, live, lane, h, hL, EPW) ? 1 : 0;
                SSG_STAMP_K(1);
                tile_barrier(k); // rendezvous B(k)
                SSG_STAMP_K(2);
                if constexpr (!LDS_BANK) {
                    if (k + 1 < K) { // a done env moves to its next record: its head goes into this role's columns
                        const bool rs = auto_reset & ((gres[(k & 1) * EPW + tl] | (gdone[(k & 1) * EPW + tl] & 1u)) != 0u);
                        if (rs) load_hdr_lidar(next_map(c, nmap, el_) * SSG_MAP_STRIDE);
                    }
                }
            }
            SSG_STAMP(10);
            SSG_STAMP_FLUSH(5);
            return;
        }
    }

    const int wq = lane / 5, wi = lane - 5 * wq; // worker coordinates of the cooperative sections: lane L = 5*q + i

    if (role == 2) {
        // =====================================================================================================
        // ROLE 2: the OBSERVER of every step
        // =====================================================================================================
        // The observer's previous frame (ship_env.py:79-113: [x, y, rudder, angle, goal x, goal y, L...]) of the first step =
        // the pre-step state: what the last step of the previous launch, or the reset, left in the state columns.
        constexpr int F = 6 + NB;
        double pv[F];
        double ogp[DYN ? 2 * SSG_MAX_GOALS : 1];
        unsigned gm_pre = 0u; // the goals listed BEFORE the step at hand (what the launch's last step speculates its nearest goal on)
        {
            const int el = el_;
            const double x0 = colX[el], y0 = colY[el], a0 = colA[el];
            const int rud0 = colRud[el], map0 = colMap[el];
            const unsigned gm0 = c.mask[el];
#pragma unroll
            for (int i = 0; i < NB; ++i) pv[6 + i] = colLid[(size_t)i * np + el];
            pv[0] = x0; pv[1] = y0; pv[2] = (double)rud0; pv[3] = a0;
            if constexpr (DYN) {
                // goals move in config 4: the previous frame's goal cannot be recomputed, it is kept in two columns
                pv[4] = c.dyn_f64[(size_t)(DC_PREV_GOAL + 0) * np + el];
                pv[5] = c.dyn_f64[(size_t)(DC_PREV_GOAL + 1) * np + el];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(pv[i])); // (in registers before barrier 0, as role 3's state)
#pragma unroll
            for (int i = 0; i < NB; ++i) asm volatile("" : "+v"(pv[6 + i]));
            if constexpr (DYN) asm volatile("" : "+v"(pv[4]), "+v"(pv[5]));
            int map0_ = map0; unsigned gm0_ = gm0;
            asm volatile("" : "+v"(map0_), "+v"(gm0_));
            gm_pre = gm0_;
            flush_tables();
            if (LDS_BANK) __builtin_amdgcn_s_waitcnt(0);
            __syncthreads(); // barrier 0
            SSG_STAMP(9);
            if constexpr (!DYN) // closest_goal (game.py:333-349) from the pre-step position
                nearest_goal<LDS_BANK, false>(c, map0_ * SSG_MAP_STRIDE + SSG_MAP_OFF_GOALS, gm0_, pv[0], pv[1], pv[4], pv[5], hG, EPW);
        }
        const bool hist2 = c.history >= 2;
        ObsTile<NB> ot;
        ot.init(lane);
        constexpr int kObsPasses = (2 * F + ObsTile<NB>::CP - 1) / ObsTile<NB>::CP; // passes of the widest row
        // A launch ends with the observer's rows of its LAST step (nobody is left to overlap them with).  Half of such a row
        // is the previous frame, in this wave's registers since the step before: on the last step those columns go out
        // EARLY, while the body role still integrates — speculatively, because an env that turns out to be done at this step
        // shows a history of -1 instead (ShipEnv.reset); its lanes rewrite their 6 + NB doubles after the rendezvous.
        // (Only when the previous frame is a whole number of column passes: 8 and 10 beams are.)
        // rows written with 16-byte LDS / global operations (write_obs_pairs; round 5: 5.36 -> 5.30 us per fused step, 14.1 -> 13.8 per
        // one-step launch): even row lengths, and at least 8 beams for the 4 KiB the transposition needs
        constexpr bool kPairs = (NB >= 8) && (NB % 2 == 0);
        constexpr bool kSplitOk = kPairs || (F % ObsTile<NB>::CP) == 0;
        const bool split_last = kSplitOk && hist2 && c.history == 2 && !SSG_ABL(7);
        for (int k = 0; k < K; ++k) {
            const bool early = split_last && (k == K - 1);
            if constexpr (kSplitOk) {
                if (early) {
                    int tw = __builtin_amdgcn_readfirstlane(tl >> 6);
                    int te0 = blockIdx.x * EPW + 64 * tw;
                    asm volatile("" : "+s"(tw), "+s"(te0));
                    // the other parity's result buffer: emptied by this wave a step ago, written again only after B(k)
                    double *colbuf = reinterpret_cast<double *>(scratch0 + tw * lds_tile_bytes(NB) + ((k + 1) & 1) * lds_res_bytes(NB));
                    double *__restrict__ ob = obs + ((size_t)te0 + (size_t)k * (size_t)traj) * (size_t)(2 * F);
                    if constexpr (kPairs)
                        write_obs_pairs<NB, true, 0, F>(colbuf, [&](int j) -> double { return pv[(j < F) ? j : 0]; }, ob, min(64, c.n_envs - te0), lane);
                    else
                        write_obs_tile<NB, true, 0, F / ObsTile<NB>::CP>(ot, colbuf, [&](int j) -> double { return pv[(j < F) ? j : 0]; }, ob,
                                                                        min(64, c.n_envs - te0), lane);
                }
            }
            if constexpr (DYN) {
                // This step's goal centres (the dyn kernels' output), asked for in one batch while the body role integrates: read
                // goal by goal in closest_goal, each was a dependent round trip in this wave's tail, the end of the launch.
#pragma unroll
                for (int g = 0; g < SSG_MAX_GOALS; ++g) {
                    const bool listed = g < c.n_goals;
                    ogp[2 * g] = listed ? goal_at<LDS_BANK, DYN>(c, el_, g, 0) : 0.0;
                    ogp[2 * g + 1] = listed ? goal_at<LDS_BANK, DYN>(c, el_, g, 1) : 0.0;
                }
            }
            wait_pose(k);
            if constexpr (DYN) {
#pragma unroll
                for (int g = 0; g < 2 * SSG_MAX_GOALS; ++g) asm volatile("" : "+v"(ogp[g]));
            }
            SSG_STAMP_K(0);
            const double x = pose[0 * EPW + tl], y = pose[1 * EPW + tl];
            const double ang = pose[6 * EPW + tl];
            const int rudder = poser[tl];
            const int map_id = posem[tl];
            ack_pose();
            const int rec_off = map_id * SSG_MAP_STRIDE;
            SSG_STAMP_K(1);
            // A launch ends with this wave's rows of its LAST step (in a one-step launch — ssg_step: every policy-in-the-loop
            // caller, every config-4 step — that is all there is), and used to build the new frame's half only after the
            // rendezvous — ~5 k cycles (12 k in config 4) during which every other wave of the grid had finished.  Everything the new frame is made of is there BEFORE the rendezvous except what
            // the rendezvous decides: whether the env is done (then the row is the next episode's first) and whether it reached a
            // goal this step (then the nearest listed goal may be another).  So the half goes out now, speculating "neither", and the
            // few lanes the rendezvous overrules rewrite their doubles afterwards (as the history half already does, below).
#ifdef SSG_NO_SPEC /* tools/build_variant.sh nospec -DSSG_NO_SPEC: the round-5 schedule, for A/B timing */
            const bool spec = false;
#else
            // (four-role tiles only: on the six-role tiles of the smaller batches the observer's row building is the longest chain
            // of the step already, and waiting for the queries in front of it cost 0.25 us per step at 4 096 envs)
            const bool spec = kSplitOk && NR == 4 && early && !SSG_ABL(7);
#endif
            double sp_gx = 0.0, sp_gy = 0.0;
            if (kSplitOk && spec) {
                wait_queries(); // both lidar waves' result keys are in LDS
                const unsigned listed_pre = gm_pre & ((1u << c.n_goals) - 1u);
                if constexpr (DYN) nearest_goal_regs(c.n_goals, listed_pre, x, y, ogp, sp_gx, sp_gy);
                else nearest_goal<LDS_BANK, false>(c, rec_off + SSG_MAP_OFF_GOALS, listed_pre, x, y, sp_gx, sp_gy, hG, EPW);
                double sv[F]; // (its own array, dead before the rendezvous: the frame kept across it would cost the fused path registers)
                sv[0] = x; sv[1] = y; sv[2] = (double)rudder; sv[3] = ang; sv[4] = sp_gx; sv[5] = sp_gy;
                int tw = __builtin_amdgcn_readfirstlane(tl >> 6);
                int te0 = blockIdx.x * EPW + 64 * tw;
                asm volatile("" : "+s"(tw), "+s"(te0));
                const unsigned long long *rk = reinterpret_cast<const unsigned long long *>(scratch0 + tw * lds_tile_bytes(NB) + (k & 1) * lds_res_bytes(NB));
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const unsigned long long key = rk[i * 64 + lane];
                    const double hitd = __longlong_as_double((long long)(key & 0x7FFFFFFFFFFFFFFFull));
                    sv[6 + i] = (key == kLidarMiss) ? pv[6 + i] : hitd;
                }
                // the other parity's buffer: nobody queries into it any more (this is the launch's last step)
                double *colbuf = reinterpret_cast<double *>(scratch0 + tw * lds_tile_bytes(NB) + ((k + 1) & 1) * lds_res_bytes(NB));
                double *__restrict__ ob = obs + ((size_t)te0 + (size_t)k * (size_t)traj) * (size_t)(2 * F);
                if constexpr (kPairs)
                    write_obs_pairs<NB, true, F, 2 * F>(colbuf, [&](int j) -> double { return sv[(j < F) ? 0 : j - F]; }, ob, min(64, c.n_envs - te0), lane);
                else
                    write_obs_tile<NB, true, kSplitOk ? F / ObsTile<NB>::CP : 0, kObsPasses>(ot, colbuf, [&](int j) -> double {
                        return sv[(j < F) ? 0 : j - F]; }, ob, min(64, c.n_envs - te0), lane);
            }
            tile_barrier(k); // rendezvous B(k)
            SSG_STAMP_K(2);

            // =================================================================================================
            // The observer: __add_states (ship_env.py:79-113,156) for step k, while role 3 already runs step k+1.
            // Row = [previous frame | new frame]; for a done env under VecEnv auto-reset, ShipGame.reset + ShipEnv.reset
            // onto the next bank record: a history of -1, then the spawn frame.  The new frame of this step is the
            // previous frame of the next one, so the frame (sticky lidar readings included) never leaves registers
            // inside a fused launch.
            // =================================================================================================
            const unsigned gd = gdone[(k & 1) * EPW + tl];
            const bool colliding = (gres[(k & 1) * EPW + tl] != 0u) | (DYN && traffic_hit(k) != 0u); // collide_ship: a bank, or traffic
            const bool do_reset = auto_reset & (colliding | ((gd & 1u) != 0u));
            // (a launch's last step: role 3, idle by then, writes these after its loop — the observer's tail is what the
            // launch waits for)
            if (live && k < K - 1) {
                // determine_reward (ship_env.py:62-77) and is_done (ship_env.py:115-134) from role 3's bits, as role 3 does
                const bool goal_reached = (gd & 4u) != 0u;
                double rew = goal_reached ? 1.0 : ((gd & 8u) ? -1.0 : -0.01);
                if ((c.flags & SSG_FLAG_FIX_COLLISION_REWARD) && (colliding & !goal_reached)) rew = -1.0;
                const size_t eo = (size_t)el_ + (size_t)k * (size_t)traj; // this step's slot of the trajectory (uniform offset)
                st_out(&reward_out[eo], rew);
                st_out(&done_out[eo], (uint8_t)((colliding | ((gd & 1u) != 0u)) ? 1 : 0));
                if (flags_out) {
                    unsigned ev = 0;
                    if (colliding) ev |= SSG_EV_COLLIDING;
                    if (goal_reached) ev |= SSG_EV_GOAL_REACHED;
                    if (gd & 8u) ev |= SSG_EV_OUT_OF_BOUNDS;
                    if (gd & 16u) ev |= SSG_EV_MAX_STEPS;
                    if (gd & 32u) ev |= SSG_EV_NO_GOALS_LEFT;
                    st_out(&flags_out[eo], (uint8_t)ev);
                }
            }
            SSG_STAMP_K(4);
            // closest_goal (game.py:333-349) among the goals still listed, from the post-step position
            double nf_gx = sp_gx, nf_gy = sp_gy;
            // (a speculating wave has computed it from the goals listed before the step: only a goal reached this step changes it)
            if (!spec || __any((gd >> 8) != (gm_pre & ((1u << c.n_goals) - 1u)))) {
                if constexpr (DYN) nearest_goal_regs(c.n_goals, gd >> 8, x, y, ogp, nf_gx, nf_gy);
                else if (!SSG_ABL(0)) nearest_goal<LDS_BANK, false>(c, rec_off + SSG_MAP_OFF_GOALS, gd >> 8, x, y, nf_gx, nf_gy, hG, EPW);
            }
            SSG_STAMP_K(5);
            int tile_w = __builtin_amdgcn_readfirstlane(tl >> 6);             // wave-uniform; laundered:
            int tile_e0 = blockIdx.x * EPW + 64 * tile_w;                    // no hoisted tile addresses
            asm volatile("" : "+s"(tile_w), "+s"(tile_e0));
            char *res_k = scratch0 + tile_w * lds_tile_bytes(NB) + (k & 1) * lds_res_bytes(NB);
            const int map_new = do_reset ? next_map(c, map_id, el_) : map_id;
            double rs_gx = 0.0, rs_gy = 0.0; // the reset frame's goal: only a reset env's lanes fetch it when the bank is gathered
            if (LDS_BANK || do_reset) {
                const double2 sg = bank_at2<LDS_BANK>(c, map_new * SSG_MAP_STRIDE + SSG_MAP_OFF_SPAWN_GOAL);
                rs_gx = sg.x; rs_gy = sg.y;
            }
            double nv[F];
            nv[0] = do_reset ? c.spawn_x : x;
            nv[1] = do_reset ? c.spawn_y : y;
            nv[2] = do_reset ? 0.0 : (double)rudder;
            nv[3] = do_reset ? 0.0 : ang;
            nv[4] = do_reset ? rs_gx : nf_gx;
            nv[5] = do_reset ? rs_gy : nf_gy;
            {
                const unsigned long long *rk = reinterpret_cast<const unsigned long long *>(res_k);
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    // smallest key = the first shape in list order that reported a hit (models.py:61-72); a miss keeps
                    // the previous reading (sticky, models.py:68-72)
                    const unsigned long long key = rk[i * 64 + lane];
                    const double hitd = __longlong_as_double((long long)(key & 0x7FFFFFFFFFFFFFFFull));
                    nv[6 + i] = (key == kLidarMiss) ? pv[6 + i] : hitd;
                }
            }
            // The TERMINAL observation of an env that is reset here (ssg_set_terminal_obs; what RLlib's vector_step reports for a
            // done env, train/rllib/ppo.py:21-44, while the reset observation — this step's row — is what its reset_at gets): the
            // previous frame and the frame the episode ended on, stored by the reset lanes only.  Overwrite mode only (traj == 0).
            if (c.term_obs != nullptr && traj == 0 && do_reset && live) {
                double *trow = c.term_obs + (size_t)el_ * (size_t)(F * c.history) + (hist2 ? F : 0);
                if (hist2) {
#pragma unroll
                    for (int j = 0; j < F; ++j) st_out(&trow[j - F], pv[j]);
                }
                st_out(&trow[0], x); st_out(&trow[1], y); st_out(&trow[2], (double)rudder); st_out(&trow[3], ang);
                st_out(&trow[4], nf_gx); st_out(&trow[5], nf_gy);
#pragma unroll
                for (int i = 0; i < NB; ++i) st_out(&trow[6 + i], nv[6 + i]);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) nv[6 + i] = do_reset ? -1.0 : nv[6 + i]; // a fresh episode starts from -1 (models.py:36)
            SSG_STAMP_K(6);
            {
                double *__restrict__ obase = obs + ((size_t)tile_e0 + (size_t)k * (size_t)traj) * (size_t)(F * c.history); // tile start in HBM, this step's slot
                const int rows_live = min(64, c.n_envs - tile_e0);                             // rows of this tile in range
                if (!SSG_ABL(7)) {
                    // (the result keys are in registers by now; in a single-step launch the lidar waves still read them for
                    // the sticky columns, and the other parity's buffer is free)
                    double *colbuf = reinterpret_cast<double *>((K == 1) ? res_k + lds_res_bytes(NB) : res_k);
                    if (kSplitOk && spec) {
                        // the rows went out before the rendezvous; what it overruled: a reset env shows the next episode's first
                        // observation (a history of -1, the spawn frame), an env that reached a goal may have another nearest goal
                        const bool goal_moved = !do_reset & ((nf_gx != sp_gx) | (nf_gy != sp_gy));
                        if (__any(do_reset | goal_moved)) {
                            __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the speculative stores of these addresses have completed
                            if (lane < rows_live) {
                                if (do_reset) {
#pragma unroll
                                    for (int j = 0; j < F; ++j) st_out(&obase[(unsigned)(lane * 2 * F + j)], -1.0);
#pragma unroll
                                    for (int j = 0; j < F; ++j) st_out(&obase[(unsigned)(lane * 2 * F + F + j)], nv[j]);
                                } else if (goal_moved) {
                                    st_out(&obase[(unsigned)(lane * 2 * F + F + 4)], nv[4]);
                                    st_out(&obase[(unsigned)(lane * 2 * F + F + 5)], nv[5]);
                                }
                            }
                        }
                    } else if (kSplitOk && early) {
                        // the history of an env that starts a new episode: -1 over what went out early.  BEFORE this step's own
                        // stores: the early ones were issued a rendezvous ago and the wait below is for nothing, where after the
                        // new frame's stores it drained them too — a tile's whole write burst, at the moment every tile of the
                        // grid writes
                        if (__any(do_reset)) {
                            __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the early stores of these addresses have completed
                            if (do_reset && lane < rows_live) {
#pragma unroll
                                for (int j = 0; j < F; ++j) st_out(&obase[(unsigned)(lane * 2 * F + j)], -1.0);
                            }
                        }
                        if constexpr (kPairs)
                            write_obs_pairs<NB, true, F, 2 * F>(colbuf, [&](int j) -> double { return nv[(j < F) ? 0 : j - F]; }, obase, rows_live, lane);
                        else
                            write_obs_tile<NB, true, kSplitOk ? F / ObsTile<NB>::CP : 0, kObsPasses>(ot, colbuf, [&](int j) -> double {
                                return nv[(j < F) ? 0 : j - F]; }, obase, rows_live, lane);
                    } else if (hist2) {
                        if constexpr (kPairs)
                            write_obs_pairs<NB, true, 0, 2 * F>(colbuf, [&](int j) -> double {
                                return (j < F) ? (do_reset ? -1.0 : pv[(j < F) ? j : 0]) : nv[(j < F) ? 0 : j - F]; }, obase, rows_live, lane);
                        else
                            write_obs_tile<NB, true, 0, kObsPasses>(ot, colbuf, [&](int j) -> double {
                                return (j < F) ? (do_reset ? -1.0 : pv[(j < F) ? j : 0]) : nv[(j < F) ? 0 : j - F]; },
                                obase, rows_live, lane);
                    } else {
                        if constexpr (kPairs)
                            write_obs_pairs<NB, false, 0, F>(colbuf, [&](int j) -> double { return nv[(j < F) ? j : 0]; }, obase, rows_live, lane);
                        else
                            write_obs_tile<NB, false, 0, kObsPasses>(ot, colbuf, [&](int j) -> double { return nv[(j < F) ? j : 0]; }, obase, rows_live, lane);
                    }
                }
            }
            SSG_STAMP_K(7);
            if constexpr (DYN) { // the newest frame's goal: the next observation's older frame (goals move: kept in two columns)
                if (live) {
                    c.dyn_f64[(size_t)(DC_PREV_GOAL + 0) * np + el_] = nv[4];
                    c.dyn_f64[(size_t)(DC_PREV_GOAL + 1) * np + el_] = nv[5];
                }
            }
#pragma unroll
            for (int i = 0; i < F; ++i) pv[i] = nv[i];
            gm_pre = do_reset ? ((1u << c.n_goals) - 1u) : (gd >> 8); // the goals listed before the next step (a fresh episode: all)
            if (k == K - 1 && K > 1 && live && !SSG_ABL(9)) { // the sticky readings go back to the state columns with the last step
                int el = el_; // (laundered: the addresses of the head's loads are not kept live — and spilled — around the step loop)
                asm volatile("" : "+v"(el));
#pragma unroll
                for (int i = 0; i < NB; ++i) st_out(&colLid[(size_t)i * np + el], pv[6 + i]);
            }
            SSG_STAMP_K(3);
        }
        SSG_STAMP(10);
        SSG_STAMP_FLUSH(4);
        return;
    }

    // =========================================================================================================
    // ROLE 3: the body.  Its registers carry the state from step to step.
    // =========================================================================================================
    double x, y, vx, vy, ang, w, cum;
    unsigned gm;
    int map_id, rudder, steps, episodes;
    {
        const int el = el_;
        x = colX[el]; y = colY[el]; vx = colVX[el]; vy = colVY[el]; ang = colA[el]; w = colW[el]; cum = colCum[el];
        gm = c.mask[el];
        map_id = colMap[el];
        rudder = colRud[el];
        steps = colStep[el];
        episodes = c.i32cols[(size_t)ICOL_EPISODE * np + el];
    }
    if constexpr (!LDS_BANK && !DYN) load_hdr_goals(map_id * SSG_MAP_STRIDE); // (gathered bank: this env's goal centres -> LDS, before barrier 0)
    int act_next = actions_kn[el_]; // step k+1's action is requested a rendezvous ahead of its use
    // Config 4: the goal circles are dynamic bodies whose centres the dyn kernel left in the env's columns before this launch
    // (a DYN launch is one step).  All of them are asked for in one batch right after barrier 0 and are there when the
    // integration is done: read goal by goal where they are used, each was a round trip of its own on the critical path to
    // rendezvous B.
    double gpre[DYN ? 2 * SSG_MAX_GOALS : 1];
    unsigned dflag = 0;
    unsigned long long dlive = 0ull; // the env's cached arbiters (needed by the classification of a resting env that reached a goal)
    auto ask_goals = [&]() {
#pragma unroll
        for (int g = 0; g < SSG_MAX_GOALS; ++g) {
            const bool listed = g < c.n_goals;
            gpre[2 * g] = listed ? goal_at<LDS_BANK, DYN>(c, el_, g, 0) : 0.0;
            gpre[2 * g + 1] = listed ? goal_at<LDS_BANK, DYN>(c, el_, g, 1) : 0.0;
        }
        dflag = c.dyn_flag[el_]; // bit 2 of the flag: the env's other bodies are at rest
        dlive = c.dyn_live[el_];
    };
    auto have_goals = [&]() {
#pragma unroll
        for (int g = 0; g < 2 * SSG_MAX_GOALS; ++g) asm volatile("" : "+v"(gpre[g]));
        asm volatile("" : "+v"(dflag), "+v"(dlive));
    };
    // (the state is wanted in registers BEFORE barrier 0, under the bank's staging: left to itself the compiler sinks the
    // loads below the barrier and the first step starts a memory round trip late)
    asm volatile("" : "+v"(x), "+v"(y), "+v"(vx), "+v"(vy), "+v"(ang), "+v"(w), "+v"(cum));
    asm volatile("" : "+v"(gm), "+v"(map_id), "+v"(rudder), "+v"(steps), "+v"(episodes), "+v"(act_next));
    flush_tables();
    if (LDS_BANK) __builtin_amdgcn_s_waitcnt(0); // vmcnt(0): the LDS-DMA writes of this wave have landed
    __syncthreads();                             // barrier 0: bank + tables + role 0's initial rotation visible
    SSG_STAMP(9);
    if constexpr (DYN) ask_goals();
    const double w_hx = shiptab[0 * 8 + wi], w_hy = shiptab[1 * 8 + wi]; // ship vertex i (local)
    const double w_nx = shiptab[2 * 8 + wi], w_ny = shiptab[3 * 8 + wi]; // ship plane normal i (local)
#ifdef SSG_PRIO
    __builtin_amdgcn_s_setprio(SSG_PRIO);
#endif

    bool q_need = false; // config 4: this env's entry in the next step's queue (see the classification after the rendezvous)
    // episode statistics of this lane's env over the launch: summed over the wave and added to the handle's counters ONCE, after
    // the launch's last step (per done lane and step they were three global atomics in the middle of the body role's chain:
    // 3.6 % of a fused step by the timing-only ablation)
    long long st_ret = 0;
    int st_len = 0, st_eps = 0, st_goals = 0;
    unsigned q_bucket = 0, q_arrival = 0;
    for (int k = 0; k < K; ++k) {
#ifdef SSG_STAMPS_ITER
    if (k < 8) SSG_STAMP(k); // (diagnostic: when does each of a launch's first 8 steps start, and the last one?)
    if (k == K - 1) SSG_STAMP(11);
    if (k == 12) SSG_STAMP(12);
    if (k == 16) SSG_STAMP(13);
#else
    SSG_STAMP_K(0);
#endif
    // Launder the env index once per step: the per-lane addresses are loop-invariant, and hoisted out of the loop they
    // cost live 64-bit pointers; recomputing an address is one v_lshl_add_u64.
    int el = el_;
    asm volatile("" : "+v"(el));
    const int act = act_next;
    const int rec_off = map_id * SSG_MAP_STRIDE;
    const int goff = DYN ? el_ : rec_off + SSG_MAP_OFF_GOALS; // where goal_at() finds this env's goal centres

    // ---- handle_discrete_action (game.py:140-153) on the pre-step pose --------------------------------------------
    double fx = 0.0, fy = 0.0, tq = 0.0;
    {
        // body->transform rotation of the pre-step angle: what the previous step (or role 0, for the first one) left in
        // the pose slot; a reset env was given (1, 0) there
        const double ca0 = pose[2 * EPW + tl], sa0 = pose[3 * EPW + tl];
        // Ship.move_forward -> cpBodyApplyForceAtLocalPoint(force_vector*1, point_of_thrust)
        const double px = (gm & 0x80u) ? (0.0 - (double)rudder) : c.px0; // models.py:109,146
        const double py = c.py0;
        const double fwx = (-sa0) * c.force_y, fwy = ca0 * c.force_y;    // cpTransformVect(transform, (0,F))
        const double pwx = ca0 * px + (-sa0) * py + x, pwy = sa0 * px + ca0 * py + y; // cpTransformPoint
        const double rx = pwx - x, ry = pwy - y;                         // minus transform * cog, cog = (0,0)
        const bool thrust = act == 0;
        fx = thrust ? fwx : 0.0;
        fy = thrust ? fwy : 0.0;
        tq = thrust ? (rx * fwy - ry * fwx) : 0.0;
    }
    if (act == 1 || act == 2) {
        // Ship.rotate(-5 / +5) + clamp_rudder (models.py:136-146)
        rudder += (act == 1) ? -c.rudder_step : c.rudder_step;
        rudder = max(-c.rudder_max, min(c.rudder_max, rudder));
        gm |= 0x80u;
    }

    // ---- cpSpaceStep (1) cpBodyUpdatePosition, (3) cpBodyUpdateVelocity (gravity 0) with the force / torque just
    //      accumulated (forces are cleared afterwards).  The narrowphase in between reads positions only, so updating the
    //      velocities here changes nothing. ----
    x = x + vx * c.dt;
    y = y + vy * c.dt;
    ang = ang + w * c.dt;
    vx = vx * c.damp + (fx * c.m_inv) * c.dt;
    vy = vy * c.damp + (fy * c.m_inv) * c.dt;
    w = w * c.damp + tq * c.i_inv * c.dt;
    // (4) impulse solver: its output cannot reach an observation before the env is reset.
    double sa, ca;
    if constexpr (DYN) {
        // (config 4 = single-step launches: no loop for the inlined polynomial's constants to be hoisted out of — and a CALL here
        // makes the wave wait for the goal / flag columns it asked for after barrier 0, a memory round trip before the pose)
        sincos_body(ang, &sa, &ca);
    } else {
        const double2 sc = sincos_call(ang); sa = sc.x; ca = sc.y;
    }

    // cpPolyShapeCacheData: world vertices and AABB of the ship
    double sbl, sbr, sbb, sbt;
    {
        double swx[SSG_SHIP_VERTS], swy[SSG_SHIP_VERTS];
        ship_world(shiptab, ca, sa, x, y, swx, swy, sbl, sbr, sbb, sbt);
    }
    // publish the post-step pose (once its consumers are done with the previous one): roles 0 / 2 collide it now, role 2
    // reports it in the observation, the lidar roles query from it for the next step
    while (__hip_atomic_load(&sync_ack[tile], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (unsigned)(NR - 1) * (unsigned)k)
        __builtin_amdgcn_s_sleep(1);
    pose[0 * EPW + tl] = x; pose[1 * EPW + tl] = y; pose[2 * EPW + tl] = ca; pose[3 * EPW + tl] = sa;
    pose[4 * EPW + tl] = x + (sbr - sbl) / 2; // lidar origin: pos + half the world AABB extents (models.py:51-53)
    pose[5 * EPW + tl] = y + (sbt - sbb) / 2;
    pose[6 * EPW + tl] = ang;
    posem[tl] = map_id;
    poser[tl] = rudder;
    SSG_STAMP_K(1);
    if (lane == 0) __hip_atomic_store(&sync_ready[tile], (unsigned)(k + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    SSG_STAMP_K(2);
    if constexpr (DYN) have_goals();

    const bool oob_x = (x < 0.0) | (x > c.width);
    const bool oob_y = (y < 0.0) | (y > c.height);

    // player <-> goal circles: collide_goal (game.py:243-257).  Contact iff cpPolyShapePointQuery distance of the
    // centre to the ship hull <= radius (negative inside), after the cpBBIntersects reject.
    bool goal_reached;
    {
        const double w_px = shiptab[4 * 8 + wi], w_py = shiptab[5 * 8 + wi]; // vertex i-1 (edge start)
        // Every (lane, goal) pair that passes the bounding-box reject goes into a per-wave LDS queue; the wave then
        // serves 12 pairs at a time, lane L = 5*p + i on (pair p, ship edge i).
        unsigned short *gq = reinterpret_cast<unsigned short *>(goal_scratch0 + (tl >> 6) * kGoalScratchBytes);
        unsigned *gw = reinterpret_cast<unsigned *>(gq + 64 * SSG_MAX_GOALS);
        gw[lane] = 0u;
        int n_pairs = 0;
        auto near_test = [&](int g, double gx, double gy) {
            const double r = c.goal_r;
            const bool near = live & !SSG_ABL(5) & (bool)((gm >> g) & 1u) & ((gx - r) <= sbr) & (sbl <= (gx + r)) &
                              ((gy - r) <= sbt) & (sbb <= (gy + r));
            const unsigned long long m = __ballot(near);
            const int pos = n_pairs + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (near) gq[pos] = (unsigned short)(lane | (g << 6));
            n_pairs += __popcll(m);
        };
        if constexpr (!LDS_BANK && !DYN) { // gathered record: the goal centres are in this env's LDS columns
#pragma unroll
            for (int g = 0; g < SSG_MAX_GOALS; ++g)
                if (g < c.n_goals) near_test(g, hG[(2 * g) * EPW], hG[(2 * g + 1) * EPW]);
        } else if constexpr (DYN) {
#pragma unroll
            for (int g = 0; g < SSG_MAX_GOALS; ++g)
                if (g < c.n_goals) near_test(g, gpre[2 * g], gpre[2 * g + 1]);
        } else {
            for (int g = 0; g < c.n_goals; ++g) near_test(g, goal_at<LDS_BANK, DYN>(c, goff, g, 0), goal_at<LDS_BANK, DYN>(c, goff, g, 1));
        }
        for (int base = 0; base < n_pairs; base += 12) {
            const int p = base + wq;
            const bool valid = (lane < 60) & (p < n_pairs);
            const unsigned code = gq[valid ? p : 0];
            const int src = code & 63, g = code >> 6;
            const double bx = __shfl(x, src), by = __shfl(y, src), bca = __shfl(ca, src), bsa = __shfl(sa, src);
            const int boff = __shfl(goff, src);
            double gx, gy;
            if constexpr (!LDS_BANK && !DYN) { gx = hG[(2 * g) * EPW - lane + src]; gy = hG[(2 * g + 1) * EPW - lane + src]; } // env `src` of this tile
            else if constexpr (DYN) { // the source lane holds its env's goal centres
                (void)boff;
                gx = __shfl(gpre[0], src); gy = __shfl(gpre[1], src);
#pragma unroll
                for (int h = 1; h < SSG_MAX_GOALS; ++h) {
                    const double hx = __shfl(gpre[2 * h], src), hy = __shfl(gpre[2 * h + 1], src);
                    gx = (g == h) ? hx : gx; gy = (g == h) ? hy : gy;
                }
            }
            else { gx = goal_at<LDS_BANK, DYN>(c, boff, g, 0); gy = goal_at<LDS_BANK, DYN>(c, boff, g, 1); }
            // lane = (pair p, ship edge i from vertex i-1 to vertex i)
            const double v1x = bca * w_hx + (-bsa) * w_hy + bx, v1y = bsa * w_hx + bca * w_hy + by;
            const double v0x = bca * w_px + (-bsa) * w_py + bx, v0y = bsa * w_px + bca * w_py + by;
            const double snx_ = bca * w_nx + (-bsa) * w_ny, sny_ = bsa * w_nx + bca * w_ny;
            const bool out_i = (snx_ * (gx - v1x) + sny_ * (gy - v1y)) > 0.0;
            // cpClosetPointOnSegment(p, v0, v1)
            const double dx = v0x - v1x, dy = v0y - v1y;
            const double tt = dmax(0.0, dmin((dx * (gx - v1x) + dy * (gy - v1y)) / (dx * dx + dy * dy), 1.0));
            const double qx = v1x + dx * tt, qy = v1y + dy * tt;
            const double ex_ = gx - qx, ey_ = gy - qy;
            const double dist = sqrt(ex_ * ex_ + ey_ * ey_);
            // min over the five edges of this pair (lanes 5p' .. 5p'+4), any(outside) over the same five lanes
            double md = dist;
#pragma unroll
            for (int kk = 1; kk < SSG_SHIP_VERTS; ++kk) {
                int o = wi + kk;
                o = (o >= SSG_SHIP_VERTS) ? o - SSG_SHIP_VERTS : o;
                md = dmin(md, __shfl(dist, 5 * wq + o));
            }
            const unsigned long long mo = __ballot(valid & out_i);
            const bool outside = ((mo >> (5 * wq)) & 31ull) != 0ull;
            const double sd = outside ? md : -md;
            if (valid & (wi == 0) & (sd <= c.goal_r)) atomicOr(&gw[src], 1u << g); // goal g of env src consumed
        }
        const unsigned gotmask = gw[lane];
        goal_reached = gotmask != 0u;
        gm &= ~gotmask;
    }
    // ---- step_count += 1; role 3's share of is_done (ship_env.py:115-134,152-154), handed to roles 0-2 ----
    steps += 1;
    const int steps_after = steps;
    const unsigned alive = gm & ((1u << c.n_goals) - 1u);
    const bool done3 = (alive == 0u) | oob_x | oob_y | (steps >= c.max_steps);
    // (the observer, role 2, finds the new frame's nearest goal among the goals this leaves listed)
    gdone[(k & 1) * EPW + tl] = (done3 ? 1u : 0u) | (goal_reached ? 4u : 0u) | ((oob_x | oob_y) ? 8u : 0u) |
                                ((steps_after >= c.max_steps) ? 16u : 0u) | ((alive == 0u) ? 32u : 0u) | (alive << 8);
    if (k + 1 < K) act_next = actions_kn[(size_t)(k + 1) * c.n_envs + el];

    SSG_STAMP_K(3);
    tile_barrier(k); // rendezvous B(k): collide_ship (role 0 / 2) is in
    SSG_STAMP_K(4);

    bool colliding = gres[(k & 1) * EPW + tl] != 0u; // collide_ship result (role 0; role 2 in a launch's first step)
    if constexpr (DYN) {
        colliding |= traffic_hit(k) != 0u; // ... and against the traffic ships (the lidar roles)
    }

    // ---- determine_reward (ship_env.py:62-77) ----
    double rew = goal_reached ? 1.0 : ((oob_x | oob_y) ? -1.0 : -0.01);
    if ((c.flags & SSG_FLAG_FIX_COLLISION_REWARD) && (colliding & !goal_reached)) rew = -1.0;
    cum += rew;

    const bool done = colliding | done3;
    const bool do_reset = done & auto_reset;
    if (do_reset) map_id = next_map(c, map_id, el_); // VecEnv auto-reset: ShipGame.reset + ShipEnv.reset onto the next bank record

    if constexpr (DYN) {
        if (live) {
            // bit 1 tells the dyn kernels to rebuild this env's traffic / goal bodies; bit 2 (bodies at rest) is theirs
            // (bit 3, set below: the env has an entry in the queue of the next full step)
        }
        // Which envs need the full cpSpaceStep of their other bodies NEXT step (shipsim_dynamics.hip)?  Everything that decides
        // it is in this role's registers now: a reset env (fresh bodies), an env whose bodies are not at rest, one that lost a
        // goal holding a cached arbiter this step.  The others keep their rest bit.  Queue = one array per sort bucket (steps
        // since the reset, bank record), appended to with a returning atomic on the bucket's counter.
        bool need_full = false;
        if (live) {
            bool resting = !do_reset & ((dflag & 4u) != 0u);
            if (resting & goal_reached) {
                // the goal(s) reached this step leave the space: the rest state survives unless one of them had a cached arbiter
                // (pair ids of shipsim_dynamics.hip: goal g x bank s = 9 + 2g + s, goal g x ship k = 21 + 3g + k, goals h < g = 39 + g(g-1)/2 + h)
                const unsigned long long lv = dlive;
                unsigned long long gone = 0ull;
                const unsigned removed = ~gm & ((1u << c.n_goals) - 1u);
                for (int g = 0; g < c.n_goals; ++g) {
                    if (!((removed >> g) & 1u)) continue;
                    gone |= 3ull << (9 + 2 * g);
                    gone |= 7ull << (21 + 3 * g);
                    for (int h = 0; h < c.n_goals; ++h)
                        if (h != g) gone |= 1ull << (h < g ? 39 + g * (g - 1) / 2 + h : 39 + h * (h - 1) / 2 + g);
                }
                resting = (lv & gone) == 0ull;
            }
            need_full = !resting;
        }
        if (need_full) {
            const unsigned bucket = dyn_bucket_of(do_reset ? 0 : steps, map_id); // (map_id is already the next episode's record)
            // The slot in the bucket's array is a returning atomic: a round trip of several microseconds behind the tile's stores.
            // It is asked for first thing after the rendezvous; the entry is stored as this role's LAST instruction.  (The NEXT
            // step's counter set: the dyn kernel of this step zeroed it.)
            q_bucket = bucket;
            q_arrival = atomicAdd(c.dyn_count + (size_t)(c.dyn_par ^ 1) * kDynCountWords + dyn_counter_word(bucket), 1u);
        }
        q_need = need_full;
        if (live) {
            c.dyn_flag[el_] = (uint8_t)((do_reset ? 2u : (dflag & 4u)) | (need_full ? 8u : 0u));
            if (need_full) c.dyn_qmap[el_] = map_id; // the record the entry is queued under (a later masked ssg_reset may move the env)
        }
    }
    SSG_STAMP_K(6);
    if (live && !SSG_ABL(6)) {
        // Episode statistics, per handle.  Integer counters in kStatsSlots slots (slot = workgroup mod slots): no
        // single hot address, and integer adds commute, so the totals are bitwise reproducible run to run.
        // cum is a sum of {1, -1, -0.01} terms, so round(100*cum) is the exact return in hundredths.
        if (done) {
            st_ret += (long long)llrint(cum * 100.0);
            st_len += steps;
            st_eps += 1;
        }
        st_goals += goal_reached ? 1 : 0;
    }
    // (reward / done / flags go to HBM from the observer, which holds the same bits)
    if (do_reset) {
        episodes += 1;
        if constexpr (!LDS_BANK && !DYN) load_hdr_goals(map_id * SSG_MAP_STRIDE); // the new world's goal centres (only these lanes gather)
    }
    SSG_STAMP_K(7);
    if (do_reset) {
        x = c.spawn_x; y = c.spawn_y; vx = 0.0; vy = 0.0; ang = 0.0; w = 0.0; cum = 0.0;
        rudder = 0; steps = 0;
        gm = (1u << c.n_goals) - 1u;
        // the pre-step rotation the next step's thrust will read: cpvforangle(0)
        pose[2 * EPW + tl] = 1.0; pose[3 * EPW + tl] = 0.0;
    }
    if (k == K - 1 && live && !SSG_ABL(9)) { // the state goes back to its columns with the last step of the launch
        st_out(&colX[el], x); st_out(&colY[el], y); st_out(&colVX[el], vx); st_out(&colVY[el], vy); st_out(&colA[el], ang);
        st_out(&colW[el], w); st_out(&colCum[el], cum);
        st_out(&colRud[el], rudder); st_out(&colStep[el], steps); st_out(&colMap[el], map_id);
        st_out(&c.i32cols[(size_t)ICOL_EPISODE * np + el], episodes);
        st_out(&c.mask[el], (uint8_t)gm);
    }
    SSG_STAMP_K(5);
    if constexpr (DYN) {
        if (k < K - 1 && q_need && q_arrival < (unsigned)c.n_pad) c.dyn_bucket[(size_t)q_bucket * np + q_arrival] = el_; // (DYN launches are single steps: never taken)
    }
    } // k
    if (!SSG_ABL(6)) {
        // the wave's episode statistics -> the handle's counters (integer adds commute: the totals are bitwise reproducible)
        if (__any((st_eps | st_goals) != 0)) {
            auto wave_sum = [&](int v) -> int {
                v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
                v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
                v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
                v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
                v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
                v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
                return __builtin_amdgcn_readlane(v, 63);
            };
            // (the return in hundredths of a wave over a launch: |sum| <= 64 envs x 127 steps x 100 — it fits 32 bits... but an
            // episode's return can reach max_steps x 1, so the per-lane term goes through two 32-bit halves of a biased value)
            const unsigned long long biased = (unsigned long long)(st_ret + (1ll << 40)); // >= 0: |st_ret| < 2^40 by far
            const int lo = wave_sum((int)(biased & 0xFFFFu)), mid = wave_sum((int)((biased >> 16) & 0xFFFFu)), hi = wave_sum((int)(biased >> 32));
            const int len = wave_sum(st_len), eps = wave_sum(st_eps), goals = wave_sum(st_goals);
            if (lane == 0) {
                unsigned long long *slot = reinterpret_cast<unsigned long long *>(c.stats) + 4 * (blockIdx.x % kStatsSlots);
                const long long ret = (long long)lo + ((long long)mid << 16) + ((long long)hi << 32) - 64ll * (1ll << 40);
                if (eps) {
                    atomicAdd(slot + 0, (unsigned long long)ret);
                    atomicAdd(slot + 1, (unsigned long long)(long long)len);
                    atomicAdd(slot + 2, (unsigned long long)(long long)eps);
                }
                if (goals) atomicAdd(slot + 3, (unsigned long long)(long long)goals);
            }
        }
    }
    if (live) {
        // The last step's outputs (the observer wrote those of the steps before), from the same LDS words the observer
        // reads: determine_reward (ship_env.py:62-77) and is_done (ship_env.py:115-134).
        const int par = (K - 1) & 1;
        const size_t el = (size_t)el_ + (size_t)(K - 1) * (size_t)traj;
        const unsigned gd = gdone[par * EPW + tl];
        const bool colliding = (gres[par * EPW + tl] != 0u) | (DYN && traffic_hit(par) != 0u);
        const bool goal_reached = (gd & 4u) != 0u;
        double rew = goal_reached ? 1.0 : ((gd & 8u) ? -1.0 : -0.01);
        if ((c.flags & SSG_FLAG_FIX_COLLISION_REWARD) && (colliding & !goal_reached)) rew = -1.0;
        st_out(&reward_out[el], rew);
        st_out(&done_out[el], (uint8_t)((colliding | ((gd & 1u) != 0u)) ? 1 : 0));
        if (flags_out) {
            unsigned ev = 0;
            if (colliding) ev |= SSG_EV_COLLIDING;
            if (goal_reached) ev |= SSG_EV_GOAL_REACHED;
            if (gd & 8u) ev |= SSG_EV_OUT_OF_BOUNDS;
            if (gd & 16u) ev |= SSG_EV_MAX_STEPS;
            if (gd & 32u) ev |= SSG_EV_NO_GOALS_LEFT;
            st_out(&flags_out[el], (uint8_t)ev);
        }
    }
    if constexpr (DYN) {
        asm volatile("" : "+v"(q_arrival)); // (first use of the atomic's result: not before this point)
        if (q_need && q_arrival < (unsigned)c.n_pad) c.dyn_bucket[(size_t)q_bucket * np + q_arrival] = el_; // (a bucket holds n_pad slots: every env once)
    }
    SSG_STAMP(10);
    SSG_STAMP_FLUSH(6);
}

#ifndef SSG_NB_GROUP
// ---------------------------------------------------------------------------------------------------------
// reset kernel: ShipEnv.reset / ShipGame.reset for the masked envs (ship_env.py:171-184, game.py:260-277)
// ---------------------------------------------------------------------------------------------------------
__global__ void reset_kernel(const DevCfg c, const uint8_t *__restrict__ mask, const int32_t *__restrict__ map_ids,
                             double *__restrict__ obs)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.n_envs) return;
    if (mask && !mask[e]) return;
    reset_env(c, e, map_ids, obs);
}

// After ssg_set_map_bank installed a SMALLER bank: stale record indices are folded into the new range.
__global__ void remap_map_ids_kernel(const DevCfg c)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.n_envs) return;
    int32_t *p = c.i32cols + (size_t)ICOL_MAP * (size_t)c.n_pad + e;
    const unsigned m = (unsigned)*p;
    if (m >= (unsigned)c.n_maps) *p = (int32_t)(m % (unsigned)c.n_maps);
}

// ---------------------------------------------------------------------------------------------------------
// HISTORY_SIZE > 2 (non-default; every reference script uses 2): the step kernel writes its two frames into the
// staging rows obs2 and this kernel maintains the handle's own [n_envs][H*F] rows (obsH, inside the state blob) like the
// reference's deque (ship_env.py:113,180-181): drop the oldest frame, append the new one; an auto-reset env gets (H-1)
// frames of -1 and the spawn frame.  The updated row is then copied to the caller's output row (which is never read: a
// masked ssg_reset into any buffer, or a trajectory slot per step, leave the history intact).  One lane per env,
// sequential over the row (a lane only reads ahead of what it writes).
// ---------------------------------------------------------------------------------------------------------
__global__ void history_shift_kernel(const DevCfg c, const uint8_t *__restrict__ done, double *__restrict__ obs)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.n_envs) return;
    const int F = 6 + c.n_beams, H = c.full_history;
    double *row = c.obsH + (size_t)e * (size_t)(F * H);
    double *out = obs + (size_t)e * (size_t)(F * H);
    const double *nf = c.obs2 + (size_t)e * (size_t)(2 * F) + F; // newest frame (or the spawn frame after a reset)
    const bool was_reset = done[e] && (c.flags & SSG_FLAG_AUTO_RESET);
    for (int j = 0; j < F * (H - 1); ++j) { const double v = was_reset ? -1.0 : row[j + F]; row[j] = v; out[j] = v; }
    for (int j = 0; j < F; ++j) { const double v = nf[j]; row[F * (H - 1) + j] = v; out[F * (H - 1) + j] = v; }
}

// ---------------------------------------------------------------------------------------------------------
// PMC calibration aid: a coalesced 8-byte-per-lane copy (the step kernel's access width), so FETCH_SIZE/WRITE_SIZE
// can be calibrated on a known byte count in this access pattern (MI355X_MICROARCH.md, HBM section).
// ---------------------------------------------------------------------------------------------------------
__global__ void calib_copy8_kernel(const double *__restrict__ src, double *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------------------------
// Measurement aid: the shader clock under an FP64 VALU load.  Every workgroup runs `iters` rounds of eight independent
// double mul+add chains per lane (what the step kernel's arithmetic is made of) and its first lane records how far the
// shader-clock counter (s_memtime) and the constant 100 MHz reference counter (s_memrealtime) advanced meanwhile:
// out[2*block] = shader cycles, out[2*block + 1] = reference ticks.  bench.py logs the clock in front of every timed repeat.
// ---------------------------------------------------------------------------------------------------------
__global__ void clock_probe_kernel(unsigned long long *__restrict__ out, int iters, double seed)
{
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = seed + (double)(threadIdx.x + j);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = a[j] * 0.999999 + 1.0e-6;
    }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    asm volatile("" : "+v"(s));
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (s == 1.2345e300) out[0] = 0; // (keeps the chains alive)
}

// ---------------------------------------------------------------------------------------------------------
// counter-based action stream: Philox4x32-10, counter = (env_lo, env_hi, step_lo, step_hi), key = seed
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&ctr)[4], const uint32_t (&key)[2])
{
    const uint64_t p0 = (uint64_t)0xD2511F53u * ctr[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * ctr[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ ctr[1] ^ key[0];
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ ctr[3] ^ key[1];
    const uint32_t n3 = (uint32_t)p0;
    ctr[0] = n0; ctr[1] = n1; ctr[2] = n2; ctr[3] = n3;
}

__global__ void fill_actions_kernel(uint64_t seed, uint64_t step0, int K, long long env_base, int n,
                                    int32_t *__restrict__ out)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)K * n;
    if (idx >= total) return;
    const int k = (int)(idx / n), e = (int)(idx % n);
    const uint64_t env = (uint64_t)(env_base + e), step = step0 + (uint64_t)k;
    uint32_t ctr[4] = {(uint32_t)env, (uint32_t)(env >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r) { key[0] += 0x9E3779B9u; key[1] += 0xBB67AE85u; }
        philox_round(ctr, key);
    }
    out[idx] = (int32_t)(((uint64_t)ctr[0] * 3u) >> 32);
}

#endif // !SSG_NB_GROUP

// ---------------------------------------------------------------------------------------------------------
// launchers (called from shipsim_api.cpp).  `epw` = envs per workgroup; the workgroup has 4*epw threads.
// The step kernel is instantiated per beam count; the instantiations are spread over four translation units
// (this file compiled with -DSSG_NB_GROUP=0..3, four beam counts each) so the library builds in parallel.
// ---------------------------------------------------------------------------------------------------------
using step_fn_t = void (*)(const DevCfg, const int32_t *, double *, double *, uint8_t *, uint8_t *, int, long long);
// variant: 0 = default, 1 = SSG_FLAG_EXACT_LIDAR (beam counts 8 and 10), 2 = config 4 / DYN (64 or 256 envs per workgroup)

#ifdef SSG_NB_GROUP
template <int NB, int EPW>
static step_fn_t step_fn_nb(bool lds, int variant)
{
    // the plane-by-plane lidar (SSG_FLAG_EXACT_LIDAR, a validation aid) is built for the two BASELINE beam counts
    if (variant == 1) {
        if constexpr (NB == 8 || NB == 10)
            return lds ? step_kernel<NB, EPW, true, true, false> : step_kernel<NB, EPW, false, true, false>;
        else return nullptr;
    }
    if (variant == 2) {
        if constexpr (EPW != 128) return lds ? step_kernel<NB, EPW, true, false, true> : step_kernel<NB, EPW, false, false, true>;
        else return nullptr;
    }
    return lds ? step_kernel<NB, EPW, true, false, false> : step_kernel<NB, EPW, false, false, false>;
}

template <int NB>
static step_fn_t step_fn_epw(int epw, bool lds, int variant)
{
    switch (epw) {
    case 64: return step_fn_nb<NB, 64>(lds, variant);
    case 128: return step_fn_nb<NB, 128>(lds, variant);
    case 256: return step_fn_nb<NB, 256>(lds, variant);
    default: return nullptr;
    }
}

#define SSG_GROUP_FN_(g) step_fn_group##g
#define SSG_GROUP_FN(g) SSG_GROUP_FN_(g)
step_fn_t SSG_GROUP_FN(SSG_NB_GROUP)(int nb, int epw, bool lds, int variant)
{
#ifdef SSG_GROUP_STUB /* development builds may leave a beam-count group out */
    (void)nb; (void)epw; (void)lds; (void)variant;
    return nullptr;
#else
    switch (nb - 4 * SSG_NB_GROUP) {
    case 1: return step_fn_epw<4 * SSG_NB_GROUP + 1>(epw, lds, variant);
    case 2: return step_fn_epw<4 * SSG_NB_GROUP + 2>(epw, lds, variant);
    case 3: return step_fn_epw<4 * SSG_NB_GROUP + 3>(epw, lds, variant);
    case 4: return step_fn_epw<4 * SSG_NB_GROUP + 4>(epw, lds, variant);
    default: return nullptr;
    }
#endif
}

#else // main translation unit: reset / action kernels and the launch entry points

step_fn_t step_fn_group0(int nb, int epw, bool lds, int variant);
step_fn_t step_fn_group1(int nb, int epw, bool lds, int variant);
step_fn_t step_fn_group2(int nb, int epw, bool lds, int variant);
step_fn_t step_fn_group3(int nb, int epw, bool lds, int variant);

static int variant_of(const DevCfg &c)
{
    if (c.n_ships > 1) return 2;
    return (c.flags & SSG_FLAG_EXACT_LIDAR) ? 1 : 0;
}

static step_fn_t step_fn(int nb, int epw, bool lds, int variant)
{
    if (nb < 1 || nb > SSG_MAX_BEAMS) return nullptr;
    switch ((nb - 1) / 4) {
    case 0: return step_fn_group0(nb, epw, lds, variant);
    case 1: return step_fn_group1(nb, epw, lds, variant);
    case 2: return step_fn_group2(nb, epw, lds, variant);
    default: return step_fn_group3(nb, epw, lds, variant);
    }
}

// dynamic LDS: [bank (if staged)] [tables + pose exchange] [per-tile lidar result buffers and queues]
size_t step_lds_bytes(int n_beams, int epw, bool lds_bank, int n_maps, bool dyn)
{
    size_t b = lds_bank ? (((size_t)n_maps * SSG_MAP_STRIDE * 8 + 15) & ~(size_t)15) : 0;
    b += (size_t)lds_fixed_bytes(epw, dyn);
    b += (size_t)(epw / 64) * (size_t)lds_tile_bytes(n_beams);
    b += (size_t)lds_hdr_bytes(epw, lds_bank, dyn);
    return b;
}

// Raise the dynamic-LDS cap of the selected instantiation to the CU's whole 160 KiB.  The attribute belongs to the
// kernel function, not to a handle: setting it to one handle's need would lower it under another handle's feet.
hipError_t prepare_step(const DevCfg &c, int epw, bool lds, size_t lds_bytes)
{
    step_fn_t k = step_fn(c.n_beams, epw, lds, variant_of(c));
    if (!k) return hipErrorInvalidValue;
    if (lds_bytes > 160u * 1024u) return hipErrorInvalidValue;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t launch_step(const DevCfg &c, int epw, bool lds, size_t lds_bytes, const int32_t *actions, int K, double *obs,
                       double *reward, uint8_t *done, uint8_t *flags, long long traj, hipStream_t stream)
{
    step_fn_t k = step_fn(c.n_beams, epw, lds, variant_of(c));
    if (!k) return hipErrorInvalidValue;
    if (variant_of(c) == 2 && K != 1) return hipErrorInvalidValue; // the DYN instantiations run exactly one step per launch
    const int grid = (c.n_envs + epw - 1) / epw;
    hipLaunchKernelGGL(k, dim3(grid), dim3(tile_roles(epw, variant_of(c) == 2) * epw), lds_bytes, stream, c, actions, obs, reward, done, flags, K, traj);
    return hipGetLastError();
}

hipError_t launch_reset(const DevCfg &c, const uint8_t *mask, const int32_t *map_ids, double *obs, hipStream_t stream)
{
    const int block = 256, grid = (c.n_envs + block - 1) / block;
    hipLaunchKernelGGL(reset_kernel, dim3(grid), dim3(block), 0, stream, c, mask, map_ids, obs);
    return hipGetLastError();
}

hipError_t launch_remap_map_ids(const DevCfg &c, hipStream_t stream)
{
    const int block = 256, grid = (c.n_envs + block - 1) / block;
    hipLaunchKernelGGL(remap_map_ids_kernel, dim3(grid), dim3(block), 0, stream, c);
    return hipGetLastError();
}

hipError_t launch_history_shift(const DevCfg &c, const uint8_t *done, double *obs, hipStream_t stream)
{
    const int block = 256, grid = (c.n_envs + block - 1) / block;
    hipLaunchKernelGGL(history_shift_kernel, dim3(grid), dim3(block), 0, stream, c, done, obs);
    return hipGetLastError();
}

hipError_t launch_calib_copy8(const double *src, double *dst, size_t n, hipStream_t stream)
{
    const int block = 256;
    hipLaunchKernelGGL(calib_copy8_kernel, dim3((unsigned)((n + block - 1) / block)), dim3(block), 0, stream, src, dst, n);
    return hipGetLastError();
}

hipError_t launch_clock_probe(unsigned long long *out, int n_blocks, int iters, hipStream_t stream)
{
    hipLaunchKernelGGL(clock_probe_kernel, dim3((unsigned)n_blocks), dim3(256), 0, stream, out, iters, 1.0);
    return hipGetLastError();
}

hipError_t launch_fill_actions(uint64_t seed, uint64_t step0, int K, long long env_base, int n, int32_t *out,
                               hipStream_t stream)
{
    const long long total = (long long)K * n;
    const int block = 256;
    const long long grid = (total + block - 1) / block;
    hipLaunchKernelGGL(fill_actions_kernel, dim3((unsigned)grid), dim3(block), 0, stream, seed, step0, K, env_base, n,
                       out);
    return hipGetLastError();
}
#endif // SSG_NB_GROUP

} // namespace ssg
