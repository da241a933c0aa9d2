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

// The queue of the full dyn step is BUCKETED by (bank record, steps since the reset): after a reset the traffic ships and goal
// bodies of an env replay a transient that depends on its world and age only (the player pushes nothing), so envs of one
// bucket walk the same code path and a wave of bucket-mates does not pay for the union of 64 different ones.  Every bucket has
// its own array of n_pad slots (DevCfg::dyn_bucket) and its own counter; a producer appends with one returning atomic.  The
// full step walks the buckets MAP-MAJOR, every map's stretch rounded up to a wave (kDynGrp slots): the lanes of a wave all sit
// on one bank record, which it then keeps once per wave instead of once per lane — and finds its 48 entries from the 512
// counters alone (rounds 2-4 ran a counting-sort kernel between the producers and the full step: 5.4 us per step).
constexpr int kDynAgeBuckets = 8, kDynMapBuckets = 64, kDynBuckets = kDynAgeBuckets * kDynMapBuckets;
constexpr int kDynGrp = 48;       // envs per wave of the full step (shipsim_dynamics.hip: kGrp)
constexpr int kDynPad = kDynMapBuckets * kDynGrp; // slots beyond n_pad the full step's grid covers: every map's stretch rounded up to a wave
// Counters: one per 128-byte line.  (Atomics on neighbouring words of ONE line serialise in the L2.  Measured with the bucketed
// queue: a map's eight age counters in one line — two 16-byte loads per lane at the head of the full step instead of eight
// 4-byte ones over 512 lines — made the full step 1.1 us faster and the step kernel, whose returning atomics then meet on 64
// lines, 1.7 us slower.)
constexpr int kDynBucketStride = 32;
constexpr int kDynCountWords = kDynBuckets * kDynBucketStride; // one set; DevCfg::dyn_count holds two (see dyn_par)
__host__ __device__ __forceinline__ constexpr int dyn_counter_word(unsigned bucket) { return (int)bucket * kDynBucketStride; }
// sort bucket of an env that is `age` steps into its episode on bank record `map_id`
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
// ---- the memo of the full dyn step (round 5; bank mode with at most kDynMapBuckets records) -------------------------------
// cpSpaceStep of an env's non-player bodies is a pure function of those bodies' cpBody fields, the cached arbiters, the goal
// mask and the bank record (the player pushes nothing).  In bank mode thousands of envs share a record and replay the SAME
// states after every reset: the first env that computes the step of a state stores (state -> next state) in a table kept in the
// state blob, and every later env in that state copies the result instead of walking the ~100 k-cycle GJK / EPA / solver chain
// that every launch of the full step used to end with.  A hit is verified against the COMPLETE input state (not a hash), so a
// memoised step writes bit for bit what the computed one writes.  Entry = header + key + value, in 8-byte words:
//   [0] tag  = (hash & ~0xFF) | generation (1..255; 0 = never used): claimed with one atomicCAS
//   [1] ready = tag once key and value are complete;  [2] born = the launch number that wrote it (entries are used from the
//   NEXT launch on: nothing written by a running launch is ever read by it);  [3] pad
//   key  [4 .. 4 + kMemoKeyWords): header (bank record | participating goals | live-arbiter count), live mask, the three
//        ships' 9 fields, the participating goals' 8 fields (0 for goals that are removed or inert: zero velocities and no
//        broadphase candidate this step — their step is the identity and nobody sees them), up to 4 cached arbiters
//        (pair id | state/age/count | contact hashes, 4 accumulated impulses)
//   value: header (changed | arbiters written | arbiters aged), live mask out, ships 3 x (9 fields, cos, sin), goals 6 x 8,
//        up to 8 arbiter records, up to 4 aged arbiters' meta words
constexpr int kMemoEntries = 1 << 14;
enum { ME_TAG = 0, ME_READY, ME_BORN, ME_PAD, ME_KEY };
constexpr int kMemoArbIn = 4, kMemoArbOut = 8, kMemoAged = 4, kMemoArbWords = 6 /* pair id | meta | hashes, 4 impulses, pad */;
// (every section starts on an even word: 16-byte loads; sections a lane's state does not use — goals that take no part, arbiter
// slots beyond its count — are neither written nor read: the key's header says which ones are in use)
constexpr int kMemoKeyShips = 2, kMemoKeyGoals = kMemoKeyShips + 28 /* 3 x 9 fields + pad */, kMemoKeyArbs = kMemoKeyGoals + 8 * SSG_MAX_GOALS;
constexpr int kMemoKeyWords = kMemoKeyArbs + kMemoArbIn * kMemoArbWords; // 102
constexpr int ME_VAL = ME_KEY + kMemoKeyWords;
constexpr int kMemoValShipWords = 12 /* 9 fields, cos, sin, pad */;
constexpr int kMemoValShips = 2, kMemoValGoals = kMemoValShips + kMemoValShipWords * SSG_N_TRAFFIC, kMemoValArbs = kMemoValGoals + 8 * SSG_MAX_GOALS;
constexpr int kMemoValAged = kMemoValArbs + kMemoArbOut * kMemoArbWords;
constexpr int kMemoValWords = kMemoValAged + kMemoAged; // 138
constexpr int kMemoStride = ME_VAL + kMemoValWords;     // words per entry (244)
static_assert(kMemoKeyGoals % 2 == 0 && kMemoKeyArbs % 2 == 0 && kMemoValGoals % 2 == 0 && kMemoValArbs % 2 == 0 && kMemoValAged % 2 == 0, "even sections");
constexpr int kMemoProbes = 4;
// The narrowphase memo (same launches, same generation / born protocol, a table of its own): cpCollide of a traffic ship against
// a bank hull — GJK + EPA from a cold start + support-edge clipping, ~28 k cycles of a lone wave — is a pure function of the bank
// record, the side, the ship and its pose (position, rotation).  A ship that rests against its bank keeps its pose bit for bit
// while other bodies of its env still move, so envs whose STATE is new (the state memo misses) mostly meet a ship x bank pair
// that is not.  Entry: header as above, key = (record | side | ship | fingerprint | bank epoch), p.x, p.y, cos, sin;
// value = count | contact hashes, normal, p1[2], p2[2].
constexpr int kNpmEntries = 1 << 14, kNpmProbes = 2;
enum { NE_KEY = 4, NE_VAL = NE_KEY + 6, kNpmStride = NE_VAL + 12 };
constexpr int kMemoStatSlots = 256, kMemoStatWords = 16; // per workgroup slot: [0] hits [1] computed [2] results stored [3] / [4] ship x bank narrowphase memo: hits / computed; [5..15] unused
static_assert(ME_KEY % 2 == 0 && ME_VAL % 2 == 0 && kMemoStride % 2 == 0, "16-byte loads of key and value");
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
    unsigned long long *dbg; // -DSSG_STAMPS builds: per-wave s_memtime stamps; product builds: the launch's clock stamps (ssg_debug_launch_clock)
    double *term_obs;        // nullable: [n_envs][history * (6 + n_beams)] rows that receive the TERMINAL observation of an env the
                             // step kernel auto-resets (ssg_set_terminal_obs)
    // config 4 (n_ships == 4): columns of the non-player bodies (shipsim_dynamics.hip); null otherwise
    int n_ships;
    double *dyn_f64;
    uint32_t *dyn_u32;
    unsigned long long *dyn_live; // bit p: pair p has a cached arbiter
    uint8_t *dyn_flag;            // (bit 0 unused: the step kernel runs collide_ship against the traffic ships itself)
                                  // bit 1: env was auto-reset by the step kernel (step -> dyn kernel)
                                  // bit 2: the env's non-player bodies are at rest (see dyn_classify_kernel)
                                  // bit 3: the env has an entry in the queue of the next full step (cleared by that step)
    unsigned long long *dyn_hash; // bank generation (DynCfg::bank_epoch) the rest bit was established for
    // The queue of the full dyn step.  Produced for step t+1 by the step kernel's body role at the end of step t (or, after a
    // host-side reset / bank change / ssg_dyn_invalidate, by dyn_classify_kernel): one array of n_pad env indices per sort bucket,
    // filled in arrival order (a returning atomic on the bucket's counter).  Two counter sets: the full step reads set dyn_par
    // and zeroes the other one, into which the step kernel that follows counts the next step's entries; the host flips dyn_par
    // after every step.
    int32_t *dyn_bucket;          // [kDynBuckets][n_pad]
    unsigned *dyn_count;          // [2][kDynCountWords]: bucket b's counter at dyn_counter_word(b) of its set
    int32_t *dyn_qmap;            // [n_pad] the bank record an env's queue entry was queued under (valid while flag bit 3 is set)
    int dyn_par;                  // the counter set that holds THIS step's queue
    double dyn_reach2[SSG_N_TRAFFIC]; // (traffic ship k's hull radius + margin)^2: the step kernel's reject in front of collide_ship's exact test
    double thull[SSG_N_TRAFFIC][2 * SSG_SHIP_VERTS], tnrm[SSG_N_TRAFFIC][2 * SSG_SHIP_VERTS]; // the traffic hulls (local), for that test
    double *dyn_row;              // [n_pad][kDynRow] row-major shadow of the DC_TRAFFIC / DC_GOALS columns (see kDynRow)
    // the memo of the full dyn step (see kMemoEntries); null = off (SSG_FLAG_DYN_MEMO_OFF, per-env worlds, banks of > 64 records)
    unsigned long long *dyn_memo;       // [kMemoEntries][kMemoStride]
    unsigned long long *dyn_memo_stats; // [kMemoStatSlots][kMemoStatWords]
    unsigned long long *dyn_npm;        // [kNpmEntries][kNpmStride] the narrowphase memo (null with dyn_memo)
    unsigned long long dyn_seq;         // number of this launch of the full step (entries born in it are not read by it)
    unsigned dyn_memo_gen;              // 1..255: entries of another generation count as empty (bank change = new generation)
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
    unsigned memo_fp;       // fingerprint of every constant the full step reads (this struct, dt, damping, goal radius, ...): part of
                            // the memo key, so a table never answers for another configuration
};

// traj: env rows between the output slots of consecutive steps of the launch (0 = every step rewrites the same rows)
hipError_t launch_step(const DevCfg &c, int epw, bool lds, size_t lds_bytes, const int32_t *actions_kn, int K, double *obs,
                       double *reward, uint8_t *done, uint8_t *flags, long long traj, hipStream_t stream);
size_t step_lds_bytes(int n_beams, int block, bool lds_bank, int n_maps, bool dyn /* the config-4 instantiations */);
hipError_t prepare_step(const DevCfg &c, int block, bool lds, size_t lds_bytes);
hipError_t launch_reset(const DevCfg &c, const uint8_t *mask, const int32_t *map_ids, double *obs, hipStream_t stream);
// sort + full step of the queue; classify = true: rebuild the queue first from the per-env flags (the step kernel did not
// produce it: first step after a host-side reset, bank change or ssg_dyn_invalidate)
hipError_t launch_dyn_step(const DevCfg &c, const DynCfg &d, bool classify, hipStream_t stream);
hipError_t prepare_dyn(const DevCfg &c);
hipError_t launch_dyn_invalidate(const DevCfg &c, const uint8_t *mask, hipStream_t stream);
hipError_t launch_dyn_reset(const DevCfg &c, const DynCfg &d, const uint8_t *mask, const int32_t *map_ids, double *obs, bool append, hipStream_t stream); // player + other bodies, one launch
hipError_t launch_render(const DevCfg &c, const DynCfg &d, int e, int width, int height, uint8_t *rgb, unsigned flags,
                         hipStream_t stream);
hipError_t launch_calib_copy8(const double *src, double *dst, size_t n, hipStream_t stream);
hipError_t launch_clock_probe(unsigned long long *out, int n_blocks, int iters, hipStream_t stream);
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
// ShipEnv.reset / ShipGame.reset of ONE env (ship_env.py:171-184, game.py:260-277): the player's columns, the goal mask, the
// observation rows (deque([-1]*n), then the spawn frame).  Shared by reset_kernel and, for config 4, the kernel that also rebuilds
// the env's traffic ships and goal bodies in the same launch (shipsim_dynamics.hip).  Returns the bank record the env is reset onto.
__device__ __forceinline__ int reset_env(const DevCfg &c, const int e, const int32_t *__restrict__ map_ids, double *__restrict__ obs)
{
    const size_t np = (size_t)c.n_pad;
    int m;
    const int started = c.i32cols[ICOL_EPISODE * np + e]; // episodes this env has started so far
    if (map_ids) m = (int)((unsigned)map_ids[e] % (unsigned)c.n_maps); // a record index never points outside the bank
    else if (c.map_ring > 0) m = e * c.map_ring + started % c.map_ring;   // the env's next brand-new world
    else m = (int)((c.env_id_base + (long long)e) % (long long)c.n_maps);
    c.i32cols[ICOL_EPISODE * np + e] = started + 1;
    const double *rec = c.bank + (size_t)m * SSG_MAP_STRIDE;
    c.f64cols[COL_X * np + e] = c.spawn_x;
    c.f64cols[COL_Y * np + e] = c.spawn_y;
    c.f64cols[COL_VX * np + e] = 0.0;
    c.f64cols[COL_VY * np + e] = 0.0;
    c.f64cols[COL_A * np + e] = 0.0;
    c.f64cols[COL_W * np + e] = 0.0;
    c.f64cols[COL_CUM * np + e] = 0.0;
    for (int i = 0; i < c.n_beams; ++i) c.f64cols[(COL_LIDAR + i) * np + e] = -1.0;
    c.i32cols[ICOL_RUDDER * np + e] = 0;
    c.i32cols[ICOL_STEP * np + e] = 0;
    c.i32cols[ICOL_MAP * np + e] = m;
    c.mask[e] = (uint8_t)((1u << c.n_goals) - 1u);
    // deque([-1]*n), ship_env.py:180-181, then the spawn frame — into the caller's rows and, with HISTORY_SIZE > 2, into the
    // handle's own copy of the rows (the frame-shift kernel's source: the caller's buffer is output only)
    const int F = 6 + c.n_beams;
    double *dst[2] = {obs ? obs + (size_t)e * (size_t)(F * c.full_history) : nullptr,
                      c.obsH ? c.obsH + (size_t)e * (size_t)(F * c.full_history) : nullptr};
    for (int t = 0; t < 2; ++t) {
        double *orow = dst[t];
        if (!orow) continue;
        for (int i = 0; i < F * (c.full_history - 1); ++i) orow[i] = -1.0;
        orow += F * (c.full_history - 1);
        orow[0] = c.spawn_x; orow[1] = c.spawn_y; orow[2] = 0.0; orow[3] = 0.0;
        orow[4] = rec[SSG_MAP_OFF_SPAWN_GOAL]; orow[5] = rec[SSG_MAP_OFF_SPAWN_GOAL + 1];
        for (int i = 0; i < c.n_beams; ++i) orow[6 + i] = -1.0;
    }
    return m;
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
