// shipsim_internal.h — shared between the kernels (.hip) and the C-ABI host layer (.cpp).  Not installed.
#ifndef SHIPSIM_INTERNAL_H
#define SHIPSIM_INTERNAL_H

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "shipsim.h"

namespace ssg {

// f64 column indices inside the state blob (each column = n_pad doubles, lane-contiguous)
enum { COL_X = 0, COL_Y, COL_VX, COL_VY, COL_A, COL_W, COL_CUM, COL_LIDAR /* + n_beams columns */ };
// i32 column indices
enum { ICOL_RUDDER = 0, ICOL_STEP, ICOL_MAP, ICOL_EPISODE /* episodes started so far (map_ring mode) */,
       ICOL_GEN /* worlds drawn so far for this env's ring */, ICOL_COUNT };

// config 4 (n_ships = 4) f64 columns of the dyn region
enum {
    DC_TRAFFIC = 0,                                   // 3 ships x (x, y, angle, vx, vy, w, v_bias.x, v_bias.y, w_bias)
    DC_GOAL_COLS = 8,
    DC_GOALS = DC_TRAFFIC + 9 * SSG_N_TRAFFIC,        // SSG_MAX_GOALS x (x, y, vx, vy, v_bias.x, v_bias.y, w, w_bias)
    DC_ARB = DC_GOALS + DC_GOAL_COLS * SSG_MAX_GOALS, // per pair: jnAcc[2], jtAcc[2]
    kDynPairs = 54,                                   // ship-bank 6, ship-ship 3, goal-bank 12, goal-ship 18, goal-goal 15
    kPolyPairs = 9,                                   // the first 9 pair ids are polygon pairs (two hashed contacts)
    DC_PREV_GOAL = DC_ARB + 4 * kDynPairs,            // (gx, gy) of the newest frame: the next observation's older frame
    DC_TROT = DC_PREV_GOAL + 2,                       // 3 ships x (cos a, sin a) of the angle column: the step kernel's collide_ship
                                                      // against traffic rebuilds the ship's world hull without a sincos
    DC_COUNT = DC_TROT + 2 * SSG_N_TRAFFIC
};
// u32 columns of the dyn region
enum { DU_META = 0 /* state | age << 3 | count << 5 */, DU_HASH = kDynPairs /* contact hashes, polygon pairs */,
       DU_COUNT = kDynPairs + kPolyPairs };

// ---------------------------------------------------------------------------------------------------------
// Config 4 pipeline (shipsim_dynamics.hip header has the whole picture).  Nothing the player does reaches the other bodies
// except through the goals it removes and through the end of its episode (PLAYER assumption: its arbiters are not solved),
// and both are decided by the player's pose after cpBodyUpdatePosition — which step k-1's velocities fix exactly.  So the
// full dyn step of API step k+1 needs nothing from the step kernel of step k: it runs BESIDE it, on a second stream.
//
// "vspace" = one Chipmunk space of non-player bodies: env e's current one (C, index e) and two slots for the ones of its coming
// episodes, built ahead and stepped once (N slot s, index (1 + s) * n_pad + e; episode q uses slot q & 1): an auto-reset continues
// from N instead of putting a fresh world's first — and hardest — cpSpaceStep on the critical path.  Every dyn column has
// 3 * n_pad elements.
//
// The queue of the full dyn step is BUCKETED by (bank record, steps since the reset): after a reset the traffic ships and goal
// bodies of an env replay a transient that depends on its world and age only (the player pushes nothing), so envs of one
// bucket walk the same code path and a wave of bucket-mates does not pay for the union of 64 different ones.  Every bucket owns
// a region of 2 * n_pad slots; a producer appends an entry with ONE returning atomic on the bucket's counter (arrival number =
// slot) — no sort pass.  The full step's waves read the 512 counters themselves and walk the buckets MAP-MAJOR, every map's
// stretch starting on a wave boundary: the lanes of a wave all sit on one bank record, which it then keeps once per wave
// instead of once per lane.  Two queues, by the parity of the step they are for: step k's is consumed (and its counters
// zeroed by the last workgroup to leave) while step k+1's is being produced.
constexpr int kDynAgeBuckets = 8, kDynMapBuckets = 64, kDynBuckets = kDynAgeBuckets * kDynMapBuckets;
constexpr int kDynGrp = 48;       // envs per wave of the full step (shipsim_dynamics.hip: kGrp)
constexpr int kDynSortedPad = kDynMapBuckets * kDynGrp; // wave slots beyond the entries: every map's stretch rounded up to a wave
constexpr int kDynBucket0 = 64;   // first bucket counter, in unsigned words after word 0 (the consumers' exit ticket)
constexpr int kDynBucketStride = 32; // one counter per 128-byte line: atomics on neighbouring words of ONE line serialise in the L2
constexpr int kDynCountWords = kDynBucket0 + kDynBuckets * kDynBucketStride; // per queue
// queue entry: vspace | type << 27 | (generation & 3) << 30
enum { DQ_STEP = 0 /* cpSpaceStep of a current space */, DQ_FRESH = 1 /* rebuild the current space from its record, then step */,
       DQ_NJOB = 2 /* build a next-episode space (the entry's vspace: N slot 0 or 1) from its record and step it once */,
       DQ_WAKE = 3 /* cpSpaceStep of a current space the step kernel woke: void if the space queued itself for the same step */,
       DQ_ADOPT0 = 4, DQ_ADOPT1 = 5 /* the env was auto-reset: cpSpaceStep FROM its N slot 0 / 1 INTO its current space */ };
constexpr unsigned kDynVMask = (1u << 27) - 1u;
__host__ __device__ __forceinline__ unsigned dyn_entry(unsigned v, unsigned type, unsigned gen) { return v | (type << 27) | ((gen & 3u) << 30); }
constexpr unsigned kDynNullEntry = 0xFFFFFFFFu; // (vspace out of range: ignored)
// what the step kernel reads of the other bodies, per env and step parity: goal g centre at 2g, 2g+1; traffic ship k
// (x, y, cos a, sin a) at 12 + 4k ..
constexpr int kDynObsGoals = 0, kDynObsTraffic = 2 * SSG_MAX_GOALS, kDynObs = 2 * SSG_MAX_GOALS + 4 * SSG_N_TRAFFIC;
constexpr int kDynObsPlanes = 4;  // planes 0, 1: the current space by step parity; 2, 3: N slot 0 / 1 after its one step (what an env sees in the
                                  // step after its auto-reset, before its current space has taken the N slot over)
constexpr int kDynPs = 6;         // the player state the dyn step predicts the goal removals from: x, y, vx, vy, angle, w
constexpr int kDynPsRow = 8;      // ... one 64-byte record per env and step parity: the six, [6] the goal mask (bits 0-5; kDynPsSkip), [7] spare
constexpr unsigned kDynPsSkip = 0x40u; // bit of the ps goal mask: this record predicts nothing (the mask is already current)
// bits of dyn_req (step kernel -> adopt pass)
enum { DR_RESET = 2 /* auto-reset in the last step: the coming step reads the N slot's table; an ADOPT entry waits in the queue of the step after */,
       DR_ADOPTING = 4 /* auto-reset two steps ago: the ADOPT entry is due in the coming step's dyn step */ };
// sort bucket of a space that is `age` steps into its episode on bank record `map_id`
__host__ __device__ __forceinline__ unsigned dyn_bucket_of(int age, int map_id)
{
    const unsigned agek = (unsigned)(age < kDynAgeBuckets - 1 ? (age < 0 ? 0 : age) : kDynAgeBuckets - 1);
    return ((unsigned)map_id & (unsigned)(kDynMapBuckets - 1)) * (unsigned)kDynAgeBuckets + agek;
}
// The sorted queue scatters the envs of a wave over the whole batch: gathered from the struct-of-arrays columns, an env's 75
// body fields cost the wave 75 x 64 cache lines.  The full dyn step therefore keeps a ROW-MAJOR shadow of them, five 128-byte
// lines per env: [0, 48) goal g field f at 8g + f, [48, 75) traffic ship k field f at 48 + 9k + f.  Written by everything that
// writes the columns (dyn_init, the full step's write-back, ssg_dyn_invalidate after a caller's own writes); read by the full
// step only.  The columns stay the interface of everything else (classify pass, step kernel, ssg_state_field).
constexpr int kDynRow = 80, kDynRowTraffic = 48;
// ... and what else the full step needs of the space, so that ONE batch of loads on five lines of the lane brings it all
// (every separately indexed table was another set of 48 scattered lines — and TLB entries — per wave): [75] the live-arbiter mask,
// [76] bank record | age << 32, [77] (N spaces) the N job's order: bank record | episode index << 32.
// [75], [76] mirror DevCfg::dyn_live / dyn_vmap / dyn_age, kept by whoever writes those (the generation is read from DevCfg::dyn_gen:
// the step kernel bumps it while a dyn step may be rewriting the row).
constexpr int kDynRowLive = 75, kDynRowMeta = 76, kDynRowOrder = 77;
// [79] the step the space has queued ITSELF for (its last dyn step changed something), 0 = none: a DQ_WAKE entry for that step is a
// duplicate.
constexpr int kDynRowSelf = 79;
__host__ __device__ __forceinline__ unsigned long long dyn_meta_pack(int vmap, unsigned age, unsigned gen)
{
    return (unsigned long long)(unsigned)vmap | ((unsigned long long)(age & 255u) << 32) | ((unsigned long long)(gen & 255u) << 40);
}
constexpr int kPadEnvs = 256;   // columns are padded to a multiple of this many envs
constexpr int kStatsSlots = 256;   // per-workgroup-slot i64 counters: [0] sum_return*100 [1] sum_length [2] episodes [3] goals hit
constexpr int kStatsDoubles = 4 * kStatsSlots;

// Kernel argument block (by value in kernarg memory; wave-uniform -> SGPRs).
struct DevCfg {
    int n_envs, n_pad;
    long long env_id_base;
    int n_beams, history /* frames the step kernel writes: min(full_history, 2) */, max_steps, n_goals;
    int full_history;     // EnvConfig.HISTORY_SIZE
    double *obs2;         // [n_envs][2F] staging rows of the step kernel when full_history > 2 (inside the state blob)
    double *obsH;         // [n_envs][H*F] the handle's own observation rows when full_history > 2 (frame-shift source)
    unsigned flags;
    int n_maps;
    int map_ring;         // 0, or R: env e owns bank records [e*R, e*R + R) as a ring of worlds (one per episode)
    int rudder_step, rudder_max;
    double spread_deg, lidar_dist, goal_r, width, height, dt, damp, spawn_x, spawn_y;
    double hull[2 * SSG_SHIP_VERTS], nrm[2 * SSG_SHIP_VERTS];
    double m_inv, i_inv, force_y, px0, py0;
    double beam_cos[SSG_MAX_BEAMS], beam_sin[SSG_MAX_BEAMS]; // cos/sin of the beam offsets phi_i from the heading
    double *f64cols;
    int32_t *i32cols;
    uint8_t *mask;
    double *stats;
    const double *bank;
    unsigned long long *dbg; // diagnostic builds only (-DSSG_STAMPS): per-wave s_memtime stamps
    // config 4 (n_ships == 4): columns of the non-player bodies (shipsim_dynamics.hip); null otherwise.  Every dyn column has
    // dyn_np = 3 * n_pad elements: vspace v = e (env e's current space) or (1 + s) * n_pad + e (N slot s).
    int n_ships;
    int dyn_np;
    unsigned dyn_tick;            // the API step this launch belongs to (k >= 1): its parity selects obs / ps / queue buffers
    double *dyn_f64;
    uint32_t *dyn_u32;
    unsigned long long *dyn_live; // bit p: pair p has a cached arbiter
    uint8_t *dyn_flag;            // [dyn_np] bit 2: the space's bodies are at rest (a fixed point of cpSpaceStep); written by the dyn kernels only
    unsigned long long *dyn_hash; // [dyn_np] bank generation (DynCfg::bank_epoch) the rest bit was established for
    uint8_t *dyn_gen;             // [dyn_np] generation of the space: queue entries of an older one are stale (the env was reset meanwhile)
    uint8_t *dyn_age;             // [dyn_np] cpSpaceSteps this space has had (saturating): its sort bucket
    int32_t *dyn_vmap;            // [dyn_np] bank record the space was built from
    uint8_t *dyn_req;             // [n_pad] DR_* bits, written by the step kernel (and cleared by a host-side reset)
    unsigned long long *dyn_nvalid; // [2][n_pad] per N slot: bank generation << 32 | episode index the slot was built for
    double *dyn_obs;              // [kDynObsPlanes][kDynObs][n_pad] what the step kernel reads of the other bodies
    double *dyn_ps;               // [2][n_pad][kDynPsRow] by step parity: the player state after the step (post-reset) and its goal mask, step kernel -> dyn step
    int32_t *dyn_region;          // [2][kDynBuckets][dyn_np] queue entries, bucket b of queue q at [(q * kDynBuckets + b) * dyn_np ..)
    unsigned *dyn_count;          // [2][kDynCountWords] per queue: [0] exit ticket of the consuming kernel; [kDynBucket0 + b * kDynBucketStride] bucket counters
    unsigned *dyn_err;            // [4] should-never-happen counters: [0] an ADOPT entry whose N slot was not usable, [1] queue overflow
    double thull[SSG_N_TRAFFIC][2 * SSG_SHIP_VERTS], tnrm[SSG_N_TRAFFIC][2 * SSG_SHIP_VERTS]; // traffic hulls (local), for collide_ship
    double dyn_reach2[SSG_N_TRAFFIC]; // (traffic ship k's hull radius + margin)^2: the step kernel's reject in front of collide_ship's exact test
    double dyn_hull_r;                // the player's hull radius about its body position
    double *dyn_row;              // [dyn_np][kDynRow] row-major shadow of the DC_TRAFFIC / DC_GOALS columns (see kDynRow)
};

// Constants of the traffic ships and of Chipmunk's solver, by value to the dyn kernels only.
struct DynCfg {
    double thull[SSG_N_TRAFFIC][2 * SSG_SHIP_VERTS], tnrm[SSG_N_TRAFFIC][2 * SSG_SHIP_VERTS];
    double tx[SSG_N_TRAFFIC], ty[SSG_N_TRAFFIC], t_i_inv[SSG_N_TRAFFIC], t_m_inv;
    double goal_m_inv, goal_i_inv;
    double ship_friction;  // 0.7 (models.py:98); banks and goals keep Chipmunk's default 0
    double bias_coef, slop; // 1 - pow(collisionBias, dt), collisionSlop
    unsigned bank_epoch;    // bumped whenever the map bank changes: part of the pose hash
    int stop_after;         // development aid (SSG_DYN_STOP): leave the dyn kernel after phase n; 0 = run it all
};

// traj: env rows between the output slots of consecutive steps of the launch (0 = every step rewrites the same rows)
hipError_t launch_step(const DevCfg &c, int epw, bool lds, size_t lds_bytes, const int32_t *actions_kn, int K, double *obs,
                       double *reward, uint8_t *done, uint8_t *flags, long long traj, hipStream_t stream);
size_t step_lds_bytes(int n_beams, int block, bool lds_bank, int n_maps);
hipError_t prepare_step(const DevCfg &c, int block, bool lds, size_t lds_bytes);
hipError_t launch_reset(const DevCfg &c, const uint8_t *mask, const int32_t *map_ids, double *obs, hipStream_t stream);
// Config 4, all with c.dyn_tick = k, the API step being prepared (shipsim_dynamics.hip):
// the cpSpaceStep of step k for the spaces queued for it (queue k & 1)
hipError_t launch_dyn_step(const DevCfg &c, const DynCfg &d, hipStream_t stream);
// rebuild the queues of steps k and k+1 and the player-state records from the columns (after the host touched the envs: ssg_reset,
// a new bank, ssg_dyn_invalidate, a fresh handle); the caller has zeroed both queues' counters on the stream
hipError_t launch_dyn_classify(const DevCfg &c, const DynCfg &d, hipStream_t stream);
hipError_t prepare_dyn(const DevCfg &c);
hipError_t launch_dyn_invalidate(const DevCfg &c, const uint8_t *mask, hipStream_t stream);
hipError_t launch_dyn_reset(const DevCfg &c, const DynCfg &d, const uint8_t *mask, hipStream_t stream);
hipError_t launch_render(const DevCfg &c, const DynCfg &d, int e, int width, int height, uint8_t *rgb, unsigned flags,
                         hipStream_t stream);
hipError_t launch_calib_copy8(const double *src, double *dst, size_t n, hipStream_t stream);
hipError_t launch_remap_map_ids(const DevCfg &c, hipStream_t stream); // ICOL_MAP %= n_maps after the bank shrank
hipError_t launch_history_shift(const DevCfg &c, const uint8_t *done, double *obs, hipStream_t stream);
hipError_t launch_generate_bank(uint64_t seed, int n_maps, int n_goals, double width, double height, double width_frac,
                                double spawn_x, double spawn_y, double *bank, double *raw, hipStream_t stream);
// map_ring mode: scan the envs for missing worlds (queue of env << 32 | episode at `queue`, its length at `count`, which
// the caller zeroed on the stream), then generate them densely; raw: optional per-slot debug rows
hipError_t launch_refill_worlds(const DevCfg &c, uint64_t seed, double width_frac, unsigned long long *queue, unsigned *count,
                                double *bank, double *raw, hipStream_t stream);
hipError_t launch_fill_actions(uint64_t seed, uint64_t step0, int K, long long env_base, int n, int32_t *out,
                               hipStream_t stream);

#ifdef __HIPCC__
// The bank record an env moves to when ShipGame.reset gives it its next world: the next record of the shared bank, or —
// map_ring mode — the next record of the env's own ring [base, base + R).
__device__ __forceinline__ int next_map_of(const DevCfg &c, int map_id)
{
    if (c.map_ring > 0) {
        const int base = map_id - map_id % c.map_ring;
        const int nxt = map_id + 1;
        return (nxt - base >= c.map_ring) ? base : nxt;
    }
    const int nxt = map_id + 1;
    return (nxt >= c.n_maps) ? 0 : nxt;
}

// Append an entry to the queue of step `tick`.  (A space may be queued twice for one step — by the dyn step that stepped it and, as
// DQ_WAKE, by the step kernel: the consumer drops a DQ_WAKE entry of a space that queued itself, kDynRowSelf.)
__device__ __forceinline__ void dyn_enqueue(const DevCfg &c, unsigned tick, int v, unsigned type, unsigned bucket)
{
    const unsigned q = tick & 1u;
    const unsigned slot = atomicAdd(c.dyn_count + (size_t)q * kDynCountWords + kDynBucket0 + bucket * kDynBucketStride, 1u);
    if (slot >= (unsigned)c.dyn_np) { atomicAdd(c.dyn_err + 1, 1u); return; }
    c.dyn_region[((size_t)q * kDynBuckets + bucket) * (size_t)c.dyn_np + slot] =
            (int32_t)dyn_entry((unsigned)v, type, (unsigned)c.dyn_gen[v]);
}

// sin / cos of a body angle (cpvforangle).  The library's sincos is ~190 instructions of full-range machinery; body
// angles stay within a few turns, so for |a| <= 2^18 this is a three-term Cody-Waite reduction by pi/2 with FMAs
// (error < 2^-100 |a|) followed by the fdlibm / musl kernels on [-pi/4, pi/4] with the reduction's tail: within 1 ulp
// of a correctly rounded sin / cos (checked against glibc on 2e7 arguments: 97.6 % identical, the rest 1 ulp), the
// same class as the library's own result; (0) -> (0, 1) exactly.  One definition for the step kernel and the dyn
// kernels, so the player's rotation has the same bits wherever it is recomputed.
__device__ __forceinline__ void sincos_body(double a, double *sn_out, double *cs_out)
{
    if (!(fabs(a) <= 262144.0)) { // (also NaN / inf)
        sincos(a, sn_out, cs_out);
        return;
    }
    const double k = rint(a * 6.36619772367581382433e-01);
    const double r1 = fma(-k, 1.57079632679489655800e+00, a);
    const double r = fma(-k, 6.12323399573676603587e-17, r1);
    double y = fma(-k, 6.12323399573676603587e-17, r1 - r); // the tail of the reduced argument
    y = fma(k, 1.4973849048591698e-33, y);                  // pi/2 = HI + MID - 1.497e-33
    const int n = (int)k;
    const double z = r * r, w = z * z;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double rs = fma(z, fma(z, S4, S3), S2) + z * w * fma(z, S6, S5);
    const double v = z * r;
    const double s0 = r - ((z * (0.5 * y - v * rs) - y) - v * S1);
    const double rc = z * fma(z, fma(z, C3, C2), C1) + w * w * fma(z, fma(z, C6, C5), C4);
    const double hz = 0.5 * z, ww = 1.0 - hz;
    const double c0 = ww + (((1.0 - ww) - hz) + (z * rc - r * y));
    double sn = (n & 1) ? c0 : s0, cs = (n & 1) ? s0 : c0;
    sn = (n & 2) ? -sn : sn;
    cs = ((n + 1) & 2) ? -cs : cs;
    *sn_out = sn;
    *cs_out = cs;
}
#endif

} // namespace ssg
#endif
