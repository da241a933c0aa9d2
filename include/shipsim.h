/*
 * shipsim.h — C ABI of libshipsim.so: the MI355X (gfx950) batched replacement for the pymunk-backed
 * ShipEnv.step()/reset() hot path of CapAI/ship-sim-gym.
 *
 * The reference reaches its physics through pymunk 5.4.0's cffi binding of libchipmunk.so
 * (notebooks/"Ship Sim Gym.ipynb":51, requirements.txt:78).  Each entry point below names the reference
 * call site(s) it replaces; all pointers are plain host or device addresses, sizes are explicit, no C++ or
 * torch types cross the boundary, nothing throws and nothing aborts.  Return value: 0 (SSG_OK) or a
 * negative ssg_status; ssg_last_error() gives the text.  A handle is used by one host thread at a time;
 * distinct handles (one per GPU / per env shard) are independent.  Device work is enqueued on the caller's
 * hipStream_t (passed as void*) and is asynchronous unless stated.
 *
 * Device memory is owned by the caller (PyTorch-ROCm tensors as containers): the state blob, the map bank
 * and every per-call buffer.  The library owns only the handle (host memory).
 */
#ifndef SHIPSIM_H
#define SHIPSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSG_ABI_VERSION 8 /* 2: ssg_config.n_ships, SSG_F_TRAFFIC / SSG_F_GOAL_BODIES (config 4); 3: ssg_init_state;
                             4: map record without dtMin/dtMax (SSG_MAP_STRIDE 145, SSG_PLANE_DOUBLES 5);
                             5: ssg_config.map_ring, ssg_refill_worlds (a brand-new world per episode, generated on the device);
                             6: ssg_rollout_traj (every step of a fused rollout lands in its own slot of a trajectory buffer);
                             7: SSG_FLAG_DYN_MEMO_OFF, SSG_F_DYN_MEMO_STATS (config 4: the memo table of the full cpSpaceStep lives
                                in the state blob, which grows by ~30 MB);
                             8: ssg_set_terminal_obs (the RLlib flow without a reset launch), ssg_step_host / ssg_wait_host (a numpy-protocol step in one
                                call), ssg_debug_launch_clock, ssg_debug_clock_probe */

typedef enum ssg_status {
    SSG_OK = 0,
    SSG_ERR_BAD_ARG = -1,
    SSG_ERR_HIP = -2,
    SSG_ERR_NOT_BOUND = -3,
    SSG_ERR_UNSUPPORTED = -4,
    SSG_ERR_NO_DEVICE = -5
} ssg_status;

/* ---- limits ---- */
#define SSG_MAX_BEAMS 16
#define SSG_MAX_GOALS 6       /* bits 0..5 of the goal mask; bit 7 = "rudder has been moved"; reference N_GOALS = 5 */
#define SSG_MAX_HULL 12       /* game_map.gen_river_poly: 10 jittered points + 2 corners, game_map.py:22-73 */
#define SSG_SHIP_VERTS 5      /* SHIP_TEMPLATE, models.py:6 */
#define SSG_MAX_HISTORY 8
#define SSG_N_TRAFFIC 3       /* ShipGame.add_default_traffic, game.py:279-286 */

/* ---- flags ---- */
#define SSG_FLAG_AUTO_RESET        0x1u /* VecEnv semantics: a done env is reset inside ssg_step and the returned
                                           observation is the reset observation (SubprocVecEnv worker behaviour,
                                           train/stable_baselines/ppo.py:123) */
#define SSG_FLAG_FIX_COLLISION_REWARD 0x2u /* off by default: the reference's determine_reward overwrites the
                                           collision reward (ship_env.py:66-77); set to make in-bounds collisions -1 */
#define SSG_FLAG_BANK_IN_GLOBAL    0x4u /* never stage the map bank in LDS (per-lane gathers from L2/HBM); forced
                                           when the bank does not fit LDS or when every env has its own slot */

#define SSG_FLAG_DYN_MEMO_OFF      0x10u /* config 4, development / measurement aid: never look a cpSpaceStep up in the memo table
                                           (see SSG_F_DYN_MEMO_STATS): every queued env walks the full narrowphase / solver chain.
                                           Results are bit for bit the same either way. */
#define SSG_FLAG_EXACT_LIDAR       0x8u /* lidar: intersect every hull plane with every beam exactly as
                                           cpPolyShapeSegmentQuery does (one division per plane and beam) instead of
                                           the default one-division-per-beam evaluation of the same predicate; the two
                                           differ only for rays within rounding of a hull vertex.  Validation aid. */

/*
 * Map bank record: SSG_MAP_STRIDE doubles per map, built on the host by ssg_host_build_map().
 *   [0] nL  [1] nR                      hull plane counts (as doubles)
 *   [2..5]  left  hull AABB l,b,r,t     [6..9] right hull AABB
 *   [10..22)  goal centres x0,y0,x1,y1,...  (SSG_MAX_GOALS pairs)
 *   [22] [23] the goal nearest to the spawn point (the reset observation's goal, ship_env.py:102-108)
 *   [24 + 5*j ..)  left  plane j: v0x v0y nx ny (v0.n)       j < 12
 *   [84 + 5*j ..)  right plane j
 * v0/n are Chipmunk's splitting planes of the hulled polygon (pm.Poly, models.py:180) and v0.n the plane offset
 * cpPolyShapeSegmentQuery uses; its per-plane edge extents dtMin = cross(n, v[j-1]) and dtMax = cross(n, v[j]) are
 * recomputed where needed from the neighbouring plane's v0 (the same two products and one difference).
 *   [144] spare
 * 145 doubles: an ODD stride in 8-byte units, so the same field of different maps falls on different LDS banks
 * (lanes of a wave sit on different maps; an even stride made such reads 8-way bank conflicts).  A 64-map bank is
 * 74 240 bytes and fits the CU's 160 KiB of LDS beside the pose exchange and the lidar waves' buffers, for up to 12
 * beams at 256 envs per workgroup.
 */
#define SSG_MAP_STRIDE 145
#define SSG_MAP_OFF_COUNTS 0
#define SSG_MAP_OFF_AABB 2
#define SSG_MAP_OFF_GOALS 10
#define SSG_MAP_OFF_SPAWN_GOAL 22
#define SSG_MAP_OFF_PLANES 24
#define SSG_PLANE_DOUBLES 5

typedef struct ssg_config {
    uint32_t struct_size;  /* sizeof(ssg_config), checked by ssg_create */
    uint32_t flags;        /* SSG_FLAG_* */
    int32_t device_id;     /* HIP device ordinal */
    int32_t n_envs;        /* envs owned by this handle (this rank's shard) */
    int64_t env_id_base;   /* global id of local env 0: keys the action stream and default map assignment */
    /* EnvConfig / LiDAR (config.py:14-17, models.py:29) */
    int32_t n_beams;       /* 1..SSG_MAX_BEAMS; reference LiDAR default 10 */
    int32_t history;       /* EnvConfig.HISTORY_SIZE, 1..SSG_MAX_HISTORY (reference default 2); above 2 every step is its own
                              launch followed by a frame-shift kernel (the fused rollout path needs history <= 2) */
    int32_t max_steps;     /* EnvConfig.MAX_STEPS */
    int32_t n_goals;       /* N_GOALS = 5, game.py:17; <= SSG_MAX_GOALS */
    double lidar_spread_deg; /* 90 */
    double lidar_dist;       /* 100 */
    double goal_radius;      /* 5, game.py:82 */
    /* GameConfig (config.py:20-24) */
    double width, height;  /* BOUNDS */
    double dt;             /* SPEED * base_dt (game.py:27,194), computed by the host in double */
    double damping_pow_dt; /* pow(space.damping = 0.4, dt) (game.py:270; cpSpaceStep) */
    /* player ship (models.py:87-111, game.py:274-275) */
    double spawn_x, spawn_y;
    double ship_hull[2 * SSG_SHIP_VERTS];    /* CCW hull of SHIP_TEMPLATE*(w,h) in cpConvexHull order */
    double ship_normals[2 * SSG_SHIP_VERTS]; /* local splitting-plane normals of that hull */
    double ship_m_inv, ship_i_inv;           /* 1/mass, 1/cpMomentForPoly */
    double force_y;                          /* force_vector = (0,100) */
    double thrust_px0, thrust_py0;           /* point_of_thrust before the first rotate(), models.py:109 */
    int32_t rudder_step, rudder_max;         /* 5, 10 */
    /* config 4 (BASELINE configs[3]) */
    int32_t n_ships;       /* 1 (default), or 4 = the player + ShipGame.add_default_traffic() after every reset
                              (game.py:279-286): goal bodies become dynamic and Chipmunk's contact solver runs for the
                              traffic ships and goals; every step is then two launches — cpSpaceStep of the queued envs, the step
                              kernel — (no fused rollout) */
    int32_t map_ring;      /* 0 (default): envs walk through a shared bank of worlds.  R in 2..128: EVERY EPISODE GETS A BRAND-NEW
                              WORLD (R up to 128; n_envs * R * SSG_MAP_STRIDE < 2^31), as ShipGame.reset does (game.py:260-277: gen_level + gen_goal_path at every reset): the bank
                              holds n_envs * R records, env e owns records [e*R, e*R + R) as a ring, episode p of env e lives
                              in record e*R + p mod R and is drawn on the device by ssg_refill_worlds from a Philox stream keyed
                              by (seed, global env id, p).  An (auto-)reset moves the env to its next record; the library
                              refills the rings between launches (at most R-1 steps are fused into one launch, so an env can
                              never outrun its ring).  The bank is read from L2/HBM in this mode (it does not fit LDS). */
} ssg_config;

typedef struct ssg_handle ssg_handle;

/* State blob fields (struct-of-arrays, one column of n_envs_padded elements per field, lane-contiguous). */
typedef enum ssg_field {
    SSG_F_X = 0, SSG_F_Y, SSG_F_VX, SSG_F_VY, SSG_F_ANGLE, SSG_F_W, /* f64: body p, v, a, w            */
    SSG_F_CUM_REWARD,                                              /* f64: ShipEnv.cumulative_reward    */
    SSG_F_LIDAR,                                                   /* f64 x n_beams: LiDAR.vals (sticky) */
    SSG_F_RUDDER,                                                  /* i32: Ship.rudder_angle            */
    SSG_F_STEP_COUNT,                                              /* i32: ShipEnv.step_count           */
    SSG_F_MAP_ID,                                                  /* i32: bank record of this env      */
    SSG_F_GOAL_MASK,                                               /* u8 : bit g = goal g still listed  */
    SSG_F_STATS,                                                   /* i64 [256 slots][4] per handle, to be summed
                                                                      over slots: 100*sum_return, sum_length,
                                                                      n_episodes, n_goals_hit */
    SSG_F_TRAFFIC,      /* f64 x 27, n_ships == 4 only: ship k = columns 9k..9k+8: x, y, angle, vx, vy, w, v_bias.x,
                           v_bias.y, w_bias (cpBody fields of add_default_traffic's ships) */
    SSG_F_GOAL_BODIES,  /* f64 x 8*SSG_MAX_GOALS, n_ships == 4 only: goal g = columns 8g..8g+7: x, y, vx, vy, v_bias.x,
                           v_bias.y, w, w_bias (add_goal's dynamic circle bodies, game.py:77-95) */
    SSG_F_DYN_FLAGS,    /* u8, n_ships == 4 only: bit 0 unused (rounds 2-3: the player touches a traffic ship; the step kernel
                           now runs that test itself), bit 1 = bodies to be rebuilt after an in-kernel auto-reset, bit 2 = the traffic ships and
                           goal bodies are at rest (their cpSpaceStep is skipped as the identity; inspection only), bit 3 = the env
                           has an entry in the queue of the next full cpSpaceStep */
    SSG_F_EPISODES,     /* i32: episodes this env has started so far (every reset counts; in map_ring mode episode p lives in
                           bank record e*R + p mod R) */
    SSG_F_DYN_MEMO_STATS, /* i64 [256 slots][16], n_ships == 4 only, to be summed over slots: [0] cpSpaceSteps answered by the memo
                           table, [1] computed, [2] results stored, [3] / [4] cpCollide(traffic ship, bank hull) answered by the narrowphase memo / computed
                           ([5..15] unused) — inspection only.  No reference counterpart: Chipmunk steps
                           every space every time (game.py:194).  In bank mode (shared worlds, <= 64 records) the traffic ships
                           and goal bodies of thousands of envs pass through the SAME states after every reset — cpSpaceStep of
                           those bodies is a pure function of their cpBody fields, the cached arbiters and the bank record (the
                           player pushes nothing) — so the first env to step a state stores (state -> next state) in a table
                           inside the state blob and later envs in that state copy the result after comparing the COMPLETE
                           state word for word: a memoised step writes exactly the bits a computed one writes. */
    SSG_F_COUNT
} ssg_field;

/* ---------------------------------------------------------------------------------------------------
 * Lifecycle
 * ------------------------------------------------------------------------------------------------- */
int ssg_abi_version(void);
const char *ssg_strerror(int status);
const char *ssg_last_error(const ssg_handle *h);

/* Replaces: ShipEnv.__init__ + ShipGame.__init__ (ship_env.py:23-48, game.py:32-58) and the pm.Space()/Body/
 * Poly construction inside them (game.py:269-270, models.py:87-111,153-196).  Host only: no GPU work. */
int ssg_create(const ssg_config *cfg, ssg_handle **out);
/* Replaces: the garbage collection of a ShipEnv and its pm.Space (the reference never closes one explicitly:
 * ship_env.py:23-48 creates, nothing destroys).  Frees the handle only — the state blob and the bank are the caller's. */
int ssg_destroy(ssg_handle *h);

/* Fill *cfg with the reference defaults (config.py:8-24, models.py:6,29,87-110, game.py:17,82,274-275). */
int ssg_default_config(ssg_config *cfg);
/* Recompute ship_hull / ship_normals / ship_m_inv / ship_i_inv for SHIP_TEMPLATE*(width_scale,height_scale):
 * Ship.__init__ (models.py:87-100): pm.moment_for_poly on the template order, pm.Poly hull order. */
int ssg_config_set_ship(ssg_config *cfg, double width_scale, double height_scale, double mass);

/* ---------------------------------------------------------------------------------------------------
 * Memory binding (caller-owned device memory)
 * What the reference keeps inside pymunk objects — cpBody position / velocity / angle per ship (models.py:87-111), the
 * goal list and its bodies (game.py:77-95), LiDAR.vals (models.py:36), ShipEnv.step_count / cumulative_reward / states
 * (ship_env.py:171-184) — lives here as struct-of-arrays columns in ONE blob the caller allocates; these three calls have
 * no reference counterpart beyond that.
 * ------------------------------------------------------------------------------------------------- */
int ssg_state_nbytes(const ssg_handle *h, size_t *nbytes);
/* offset (bytes) of a field's first column in the blob, element size, columns per env-field. */
int ssg_state_field(const ssg_handle *h, int field, size_t *offset, int *elem_size, int *n_columns,
                    size_t *column_stride_bytes);
/* dev_state: ssg_state_nbytes() bytes of device memory, 256-byte aligned.  The blob must start out ZEROED (episode
 * counters, the config-4 queue counter and rest/arbiter columns are only ever updated, never initialised, by the step
 * kernels): either hand over zeroed memory, or call ssg_init_state, or let the first full reset after binding do it —
 * ssg_reset with dev_mask == NULL on a freshly bound blob zeroes it first. */
int ssg_bind_state(ssg_handle *h, void *dev_state);
/* Zero the whole bound state blob asynchronously on `stream` (hipMemsetAsync): episode statistics, config-4 columns and
 * all body state; follow with ssg_reset.  Replaces nothing in the reference (a fresh ShipEnv object starts empty). */
int ssg_init_state(ssg_handle *h, void *stream);
/* Replaces: gen_level + PolyEnv (game.py:60-71, models.py:153-196) and the goal list (game.py:77-95).
 * dev_bank: n_maps records of SSG_MAP_STRIDE doubles in device memory.  Installing a bank with FEWER maps than the
 * previous one re-maps every env's record index modulo the new n_maps before the next reset / step (an env keeps
 * stepping on a record that exists; callers normally reset after a bank change anyway).  Map ids handed to ssg_reset
 * are likewise taken modulo n_maps: a record index can never point outside the bank. */
int ssg_set_map_bank(ssg_handle *h, const double *dev_bank, int n_maps);

/* ---------------------------------------------------------------------------------------------------
 * The hot path
 * ------------------------------------------------------------------------------------------------- */
/* Replaces: ShipEnv.reset -> ShipGame.reset (ship_env.py:171-184, game.py:260-277).
 * dev_mask: u8[n_envs], non-zero = reset this env; NULL = all.  dev_map_ids: i32[n_envs] record to install for
 * each reset env; NULL = (env_id_base + e) mod n_maps (map_ring mode: must be NULL — the env moves to the next record of
 * its own ring; SSG_ERR_BAD_ARG otherwise).  dev_obs: f64[n_envs][history*(6+n_beams)], rows of reset
 * envs are overwritten with the reset observation. */
int ssg_reset(ssg_handle *h, const uint8_t *dev_mask, const int32_t *dev_map_ids, double *dev_obs, void *stream);

/* Replaces: ShipEnv.step (ship_env.py:136-156) = handle_discrete_action (game.py:140-153) + update: LiDAR.query
 * (models.py:39-76) + space.step = cpSpaceStep (game.py:194) with the collide_ship / collide_goal callbacks
 * (game.py:232-257) + determine_reward / __add_states / is_done (ship_env.py:62-134).
 * dev_actions i32[n_envs] in {0,1,2,3}; dev_obs f64[n_envs][D]; dev_reward f64[n_envs]; dev_done u8[n_envs] (0/1);
 * dev_flags u8[n_envs] event bits SSG_EV_* or NULL. */
int ssg_step(ssg_handle *h, const int32_t *dev_actions, double *dev_obs, double *dev_reward, uint8_t *dev_done,
             uint8_t *dev_flags /* nullable */, void *stream);

/* Replaces: one `env.step(actions)` of the reference's trainers on HOST arrays (train/stable_baselines/ppo.py:122-123: SubprocVecEnv's
 * pipe round trip per step; train/rllib/ppo.py:21-44) — the whole step of the numpy protocols in ONE call, all asynchronous on `stream`:
 * host_actions (i32[n_envs], pinned host memory) -> dev_actions, ssg_step into dev_obs / dev_reward / dev_done / dev_flags, then ONE copy
 * of the caller's packed output block [dev_block, dev_block + block_bytes) — which those four buffers are sections of — into host_block
 * (pinned host memory), and a completion event of the handle's `slot` (0..7: callers rotate host blocks so that the arrays of one step
 * stay valid while the next steps run).  ssg_wait_host(h, slot) blocks the calling thread until that step's host block is complete.
 * What ShipVecEnv.step_async / step_wait are made of: one foreign call each instead of a dozen interpreter-level stream / copy / event
 * operations (which cost more than the 14-us step at small batches). */
int ssg_step_host(ssg_handle *h, const int32_t *host_actions, int32_t *dev_actions, double *dev_obs, double *dev_reward, uint8_t *dev_done,
                  uint8_t *dev_flags /* nullable */, const void *dev_block, void *host_block, size_t block_bytes, int slot, void *stream);
/* Replaces: the blocking half of that step — SubprocVecEnv.step_wait's `remote.recv()` (train/stable_baselines/ppo.py:122-123) — for
 * the step issued into `slot`: returns when its host block is complete. */
int ssg_wait_host(ssg_handle *h, int slot);

/* Replaces: what RLlib's VectorEnv flow sees of an episode's end (train/rllib/ppo.py:21-44 over ShipEnv.step / ShipEnv.reset,
 * ship_env.py:136-156,171-184): vector_step reports the TERMINAL observation of a done env, reset_at then resets it and returns the
 * reset observation.  With SSG_FLAG_AUTO_RESET a done env is reset inside ssg_step and its row of dev_obs is the reset observation;
 * with dev_term_obs != NULL (f64[n_envs][history*(6+n_beams)], caller-owned) the step kernel ALSO stores the terminal observation
 * of every env it resets into that env's row of dev_term_obs (rows of other envs are left untouched): one step, no reset launch,
 * both observations.  ssg_step / ssg_rollout only (a trajectory rollout keeps every step in its own slot and is not served);
 * history <= 2 (SSG_ERR_UNSUPPORTED otherwise: reset the done envs with a masked ssg_reset there).  NULL switches it off. */
int ssg_set_terminal_obs(ssg_handle *h, double *dev_term_obs);

/* Per-env event bits written to dev_flags (the reference's ShipGame.colliding / goal_reached attributes that
 * tests reach through env.game.*, game.py:190-191,240,254, plus which is_done branch fired). */
#define SSG_EV_COLLIDING 0x1u
#define SSG_EV_GOAL_REACHED 0x2u
#define SSG_EV_OUT_OF_BOUNDS 0x4u
#define SSG_EV_MAX_STEPS 0x8u
#define SSG_EV_NO_GOALS_LEFT 0x10u

/* K consecutive steps, step k reading dev_actions + k*n_envs (the random-action rollout loop of
 * train/random.py:14-27, batched), enqueued on `stream` as ceil(K / SSG_ROLLOUT_STEPS_PER_LAUNCH) launches of the step
 * kernel: inside a launch the map bank stays in LDS and the body state in registers from step to step.
 * obs/reward/done/flags are overwritten by every step and the state blob is updated every step; the final contents of
 * every buffer are bit for bit those of K separate ssg_step calls.  With history > 2 or n_ships = 4 every step is its own
 * launch sequence (frame shift / the two dyn kernels before the step kernel): same results, no fusion.
 * HIP graphs: a 1-ship handle on a shared bank with history <= 2 launches with constant arguments, so ssg_step / ssg_rollout on a
 * capturing stream can be captured and replayed (on this ROCm a replay costs more than the plain launch it replaces).  Handles
 * with n_ships > 1, map_ring or history > 2 take per-call host state in their kernel arguments: ssg_step / ssg_rollout / ssg_reset
 * return SSG_ERR_UNSUPPORTED on a capturing stream instead of recording a step that every replay would repeat. */
#define SSG_ROLLOUT_STEPS_PER_LAUNCH 100 /* steps fused into one launch of the step kernel by ssg_rollout */
int ssg_rollout(ssg_handle *h, const int32_t *dev_actions_KN, int K, double *dev_obs, double *dev_reward,
                uint8_t *dev_done, uint8_t *dev_flags /* nullable */, void *stream);

/* The same K steps, with EVERY step's outputs kept: step k writes its observation rows at dev_obs + k * step_stride_envs * D
 * doubles (D = history*(6+n_beams)) and its reward / done / flags at element k * step_stride_envs of their buffers — the
 * (obs, reward, done) of every step that the reference's rollout loop consumes (train/random.py:14-27: `obs, reward, done, _ =
 * env.step(action)` inside the loop), as trajectory tensors [K][step_stride_envs][...].  step_stride_envs = n_envs gives
 * contiguous [K][n_envs] tensors; a larger stride interleaves this handle's shard into a wider [K][total_envs] layout;
 * 0 = ssg_rollout (every step rewrites the same rows).  Must be 0 or >= n_envs.  Same launches, same fusion, same results
 * per step as K ssg_step calls — a fused step's outputs just no longer overwrite the previous step's, so all of them reach
 * HBM (233 B per env-step at 8 beams, history 2) and every fused step can be checked against the oracle. */
int ssg_rollout_traj(ssg_handle *h, const int32_t *dev_actions_KN, int K, double *dev_obs, double *dev_reward,
                     uint8_t *dev_done, uint8_t *dev_flags /* nullable */, int64_t step_stride_envs, void *stream);

/* Random-action rollout driver (train/random.py:14-27 batched): fills i32[K][n_envs] with a counter-based
 * Philox4x32-10 stream keyed by (seed, step0+k, env_id_base+e), uniform on Discrete(3) (ship_env.py:19). */
int ssg_fill_actions(ssg_handle *h, uint64_t seed, uint64_t step0, int K, int32_t *dev_actions, void *stream);

/* Reset-time world generation on the device: fills dev_bank with n_maps fresh records — river
 * banks as game_map.gen_river_poly draws them (game_map.py:22-73), hulls/planes as pm.Poly derives them, goals as
 * gen_goal_path places them (game.py:300-330) — from a Philox4x32-10 stream keyed by (seed, map index).  NOT
 * seed-compatible with the reference's Mersenne-Twister draws: a separate mode for refreshing the bank without the
 * host.  dev_raw (nullable): per map 48 + 3*n_goals doubles = the raw 2x12 polygon vertices, then per goal (y, the
 * uniform draw u, the fallback x), so a test can rebuild every record on the host and compare bit for bit.
 * Call ssg_set_map_bank afterwards (or pass the already-installed bank pointer to refresh it in place). */
int ssg_generate_bank(ssg_handle *h, uint64_t seed, double width_frac, double *dev_bank, int n_maps, double *dev_raw,
                      void *stream);

/* map_ring mode (ssg_config.map_ring = R >= 2): generate every world the rings are missing — for each env the episodes
 * from the first one not yet drawn up to (current episode + R - 1) — into the bank installed with ssg_set_map_bank
 * (n_maps = n_envs * R), on the device: river banks as game_map.gen_river_poly draws them (game_map.py:22-73), hulls /
 * planes as pm.Poly derives them, goals as gen_goal_path places them (game.py:300-330), Philox4x32-10 keyed by (seed,
 * env_id_base + e, episode).  Must be called once after ssg_set_map_bank and before the first ssg_reset (it fills the
 * rings and fixes seed / width_frac for the automatic refills ssg_reset / ssg_step / ssg_rollout issue afterwards).
 * dev_raw (nullable): [n_envs * R][48 + 3*n_goals] doubles, row e*R + slot receives the raw polygons and goal draws of the
 * world generated into that slot by THIS call (rows of slots not regenerated are left untouched), so a test can rebuild
 * the records on the host and compare bit for bit.  NOT seed-compatible with the reference's Mersenne-Twister draws. */
int ssg_refill_worlds(ssg_handle *h, uint64_t seed, double width_frac, double *dev_raw, void *stream);

/* Config 4 only; no reference counterpart (writing `ship.body.position` on a pymunk body, game.py:117-131, needs no
 * announcement there).  The traffic ships and goal bodies of an env whose space has reached a fixed point of cpSpaceStep are
 * not stepped again until something changes (SSG_F_DYN_FLAGS bit 2), and WHICH envs the next step's dyn kernels visit is
 * decided at the end of each step from the player state the step kernel holds in registers.  The library sees resets, goal
 * removals and bank changes itself; a caller that WRITES ANY state column of a config-4 handle between two steps — the
 * SSG_F_TRAFFIC / SSG_F_GOAL_BODIES columns, but also the player's own SSG_F_X .. SSG_F_W, SSG_F_GOAL_MASK, SSG_F_STEP_COUNT
 * or SSG_F_MAP_ID (scenario set-up, curriculum placement, tests) — tells it with this call: the queue of the next step is then
 * rebuilt from the columns, and the ships' rotation columns (what collide_ship's player x traffic test in the step kernel turns
 * their hulls with) are recomputed from the angles.
 * dev_mask: u8[n_envs], non-zero = also clear the env's rest bit and refresh its row-major shadow; NULL = all envs.  (The rotation
 * columns are recomputed for every env whatever the mask says.)
 * The caller-owned state blob also holds the queue of the next full cpSpaceStep, whose live counter set is named by the HANDLE:
 * a blob that is copied, restored in place or bound to another handle between two steps must be followed by ssg_bind_state or
 * ssg_dyn_invalidate(h, NULL, ...) — both make the next step rebuild the queue from the per-env flags AND start the memo tables
 * inside the blob empty (a memo key names the bank RECORD, not the bank's contents: entries another handle stored over another bank
 * must never answer for this one). */
int ssg_dyn_invalidate(ssg_handle *h, const uint8_t *dev_mask, void *stream);

/* Replaces: ShipGame.render + ShipGame.get_screen (game.py:133-138,197-229) for ONE env: an RGB frame of `width` x
 * `height` pixels covering the env's bounds, laid out like pygame.surfarray.array3d ([x][y][3], screen y down).
 * flags bit 0 = GameConfig.DEBUG drawing (shapes in their colours + one circle per lidar beam end); the yellow
 * player marker is always drawn.  Debugging / video aid (`metadata['render.modes']` lists 'rgb_array',
 * ship_env.py:18); not a hot path. */
int ssg_render(ssg_handle *h, int env_index, int width, int height, uint8_t *dev_rgb, uint32_t flags, void *stream);

/* Measurement aid (config 4; no reference counterpart): with enable != 0, every following ssg_step / ssg_rollout* call brackets the
 * two launches of each step — the full cpSpaceStep of the queued envs, the step kernel — with HIP events on the caller's stream and
 * waits for its own work at the end of the call.  Returns the averages (microseconds per step) and the number of steps accumulated
 * since the previous call, then starts over.  How bench.py splits a config-4 step into its two kernels. */
int ssg_debug_kernel_times(ssg_handle *h, int enable, double *dyn_step_us, double *step_kernel_us, uint64_t *steps);

/* Inspection aid (config 4; no reference counterpart): launches of the full cpSpaceStep so far, and how many of them had to
 * rebuild their queue from the per-env flags first (a pass over every env: after a full ssg_reset, a bank change,
 * ssg_dyn_invalidate, a second masked ssg_reset between two steps; NOT after one masked ssg_reset between two steps, whose envs
 * join the queue the step kernel left). */
int ssg_debug_dyn_counters(const ssg_handle *h, uint64_t *full_steps, uint64_t *queue_rebuilds);

/* Inspection aid (no reference counterpart): how ssg_set_map_bank laid the step kernel out for this handle — envs per workgroup (256, 128 or 64),
 * whether the bank is staged in LDS (0 = gathered from L2 / HBM) and the dynamic LDS bytes per workgroup.  The bank is staged when it
 * fits the CU's 160 KiB of LDS beside the exchange and lidar buffers at the workgroup size preferred for the env count; otherwise it is
 * staged beside a smaller workgroup only if gathering would not allow a larger one. */
int ssg_debug_launch_geometry(const ssg_handle *h, int *envs_per_workgroup, int *bank_in_lds, size_t *lds_bytes);

/* Measurement aid (no reference counterpart): coalesced 8-byte-per-lane device copy of n_doubles doubles, the
 * step kernel's access width, for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on a known byte count. */
int ssg_debug_copy8(const double *dev_src, double *dev_dst, size_t n_doubles, void *stream);

/* Measurement aid (no reference counterpart): the shader clock DURING the step kernel's launches.  With dev_buf != NULL (two u64 of
 * device memory), the first wave of workgroup 0 of every following step-kernel launch stores, when it ends, dev_buf[0] = shader-clock
 * cycles (s_memtime) and dev_buf[1] = ticks of the constant 100 MHz reference counter (s_memrealtime) that passed since it started:
 * clock = 100 MHz x cycles / ticks, measured inside the timed launch itself with nothing launched around it.  NULL switches it off
 * (the default; the kernel then executes two scalar counter reads per wave and one untaken branch).  How bench.py reports
 * `repeats_shader_clock_ghz`. */
int ssg_debug_launch_clock(ssg_handle *h, uint64_t *dev_buf);

/* Measurement aid (no reference counterpart): the shader clock under an FP64 VALU load.  n_blocks workgroups of 256 lanes run
 * `iters` rounds of eight independent double mul + add chains; dev_out[2*b] = shader-clock cycles (s_memtime) and dev_out[2*b+1] =
 * ticks of the constant 100 MHz reference counter (s_memrealtime) that workgroup b saw pass meanwhile: clock = 100 MHz x cycles /
 * ticks.  (A probe wide enough to load every CU pulls the chip into its power-limited clocks and slows whatever is timed right
 * after it: bench.py uses ssg_debug_launch_clock for the timed repeats instead.) */
int ssg_debug_clock_probe(uint64_t *dev_out, int n_blocks, int iters, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Host-side geometry (what pymunk's cffi exposed at reset time); no GPU needed.
 * ------------------------------------------------------------------------------------------------- */
/* Replaces cpConvexHull as reached by pm.Poly(...) (models.py:96,180).  out_xy holds >= count pairs. */
int ssg_host_convex_hull(int count, const double *verts_xy, double *out_xy, int *out_count);
/* Replaces pm.moment_for_poly (models.py:89). */
int ssg_host_moment_for_poly(double mass, int count, const double *verts_xy, double *out);
/* Replaces Space.segment_query((W/2,y),(edge,y),10,filter)[0] over the two bank shapes (game.py:322-323):
 * hulls are the records' planes.  hit=0 -> the reference's IndexError fallback applies. */
int ssg_host_goal_x_range(const double *map_record, double width, double y, double *lo, double *hi, int *hit);
/* Builds one bank record from the two raw 12-gons of gen_river_poly (game_map.py:22-73) and the goal centres of
 * gen_goal_path (game.py:300-330): what PolyEnv.__init__ (models.py:153-196: pm.Poly hulls the points, one static shape per
 * bank) and add_goal (game.py:77-95) leave in the pm.Space at reset. */
int ssg_host_build_map(const double *left_xy, int n_left, const double *right_xy, int n_right,
                       const double *goals_xy, int n_goals, double spawn_x, double spawn_y, double *record_out);
/* Generic fat/thin segment query against one hull of a record (side 0 = left, 1 = right): cpShapeSegmentQuery as reached by
 * Shape.segment_query (the lidar beams, models.py:67, radius 0) and Space.segment_query (the goal path's fat rays,
 * game.py:322-323, radius 10). */
int ssg_host_segment_query(const double *map_record, int side, double ax, double ay, double bx, double by,
                           double radius, int *hit, double *px, double *py, double *alpha);

#ifdef __cplusplus
}
#endif
#endif /* SHIPSIM_H */
