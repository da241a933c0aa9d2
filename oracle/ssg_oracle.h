/*
 * ssg_oracle.h — CPU ORACLE for the ShipEnv step/reset path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call this
 * code.  The product (ship_sim_gym_amd/ + libshipsim.so) never links or imports it.
 *
 * PARITY UNPINNED at the pymunk boundary: the arithmetic of the reference path lives in
 * pymunk==5.4.0 (requirements.txt:78), a cffi wrapper around Chipmunk2D 7.0.x whose source is not
 * under /root/reference and is not installed in the build container, and the reference's own test
 * file (tests/test_ship_env.py) is stale against its API and holds no numeric vectors.  This file
 * restates the published Chipmunk2D algorithms (cpSpaceStep, cpBodyUpdatePosition/Velocity,
 * cpConvexHull, cpPolyShape* queries, cpMomentForPoly) for exactly the subset the reference's call
 * sites reach, in the reference's operation order, double precision.  What IS pinned against the real
 * reference: map polygons (game_map.gen_river_poly), config defaults, Curriculum (tests/golden/).
 */
#ifndef SSG_ORACLE_H
#define SSG_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_MAX_VERTS 16
#define ORA_MAX_GOALS 8
#define ORA_MAX_BEAMS 32
#define ORA_MAX_HISTORY 8
#define ORA_MAX_SHIPS 4
#define ORA_MAP_POLY_VERTS 12 /* game_map.py:22-73: N=10 jittered points + 2 corners */

typedef struct { double x, y; } ora_v2;

/* Convex polygon shape: Chipmunk cpPolyShape restated (local + world splitting planes). */
typedef struct {
    int count;
    ora_v2 lv[ORA_MAX_VERTS], ln[ORA_MAX_VERTS]; /* local  planes: v0, n (cpPolyShape SetVerts) */
    ora_v2 wv[ORA_MAX_VERTS], wn[ORA_MAX_VERTS]; /* world  planes (cpPolyShapeCacheData)        */
    double bb_l, bb_b, bb_r, bb_t;               /* cached world AABB; zero until first update  */
} ora_poly;

typedef struct {
    ora_v2 p, v, f, rot;
    double a, w, t;
    double m_inv, i_inv;
    ora_v2 v_bias;  /* cpBody.v_bias / w_bias: penetration-correction pseudo-velocities (config 4 only) */
    double w_bias;
} ora_body;

typedef struct {
    int shape_hit;   /* 0 = info->shape == NULL */
    ora_v2 point, normal;
    double alpha;
} ora_seg_info;

typedef struct {
    /* GameConfig / EnvConfig / LiDAR constants */
    double width, height;     /* GameConfig.BOUNDS                   config.py:24  */
    double dt;                /* SPEED * base_dt                     game.py:27,194 */
    double space_damping;     /* space.damping = 0.4                 game.py:270   */
    int max_steps, history;   /* EnvConfig                           config.py:15-16 */
    int n_beams;              /* LiDAR defaults                      models.py:29  */
    double lidar_spread_deg, lidar_dist;
    int n_goals;              /* N_GOALS                             game.py:17    */
    double goal_radius;       /* add_goal radius                     game.py:82    */
    double ship_w, ship_h;    /* player scale (2,3)                  game.py:275   */
    double ship_mass;         /* Ship(mass=5)                        models.py:87  */
    double force_y;           /* force_vector = (0,100)              models.py:107 */
    int rudder_step, rudder_max; /* rotate(+-5), max_angle=10        game.py:149-151, models.py:110 */
    double thrust_px0, thrust_py0; /* shape.bb.center() before space.add = (0,0)  models.py:109, App. A.3 */
    double spawn_x, spawn_y;  /* (BOUNDS[0]/2, 25)                   game.py:274   */
    int n_traffic;            /* 0, or 3 = add_default_traffic() after every reset   game.py:279-286 (config 4) */
    int variant;              /* ORA_VAR_* bits: 0 = the named assumptions as documented; test-only switches that flip one
                                 assumption each, for the sensitivity table (tools/assumption_sensitivity.py) */
} ora_config;

/* Switches for the named, unverifiable assumptions (pymunk is absent).  Each flips ONE of them. */
#define ORA_VAR_TOUCH_STRICT      0x01 /* touching does NOT count: a zero-width gap separates (SAT `>=`, circle `<`) */
#define ORA_VAR_ORDER_REVERSED    0x02 /* ORDER: the solver walks the arbiter list in the reverse of the canonical order */
#define ORA_VAR_SWAP_AB           0x04 /* ORDER: poly-poly pairs are collided with a/b exchanged (which poly is "a") */
#define ORA_VAR_GJK_WARM          0x08 /* GJK-ID: GJK restarts from the pair's cached collision id while the AABBs overlap */
#define ORA_VAR_CHECK_SAT         0x10 /* count player pairs on which the SAT predicate and cpCollide's contact count differ */
#define ORA_VAR_PLAYER_CPCOLLIDE  0x20 /* the player's `colliding` comes from cpCollide's contact count, not from SAT */

/* ---- config 4 (BASELINE configs[3]): traffic ships, dynamic goal bodies, Chipmunk contact solver ----
 * Shape slots in space-insertion order: 0,1 banks (static) | 2..6 goal circles | 7 player | 8..10 traffic.   */
#define ORA_N_TRAFFIC 3
#define ORA_SLOT_GOAL0 2
#define ORA_SLOT_PLAYER 7
#define ORA_SLOT_TRAFFIC0 8
#define ORA_N_SLOTS 11
#define ORA_ARB_NONE 0    /* not in space->cachedArbiters                         */
#define ORA_ARB_FIRST 1   /* CP_ARBITER_STATE_FIRST_COLLISION                     */
#define ORA_ARB_NORMAL 2  /* CP_ARBITER_STATE_NORMAL                              */
#define ORA_ARB_IGNORE 3  /* CP_ARBITER_STATE_IGNORE                              */
#define ORA_ARB_CACHED 4  /* CP_ARBITER_STATE_CACHED (separated, kept <3 stamps)  */

typedef struct {           /* struct cpContact */
    ora_v2 r1, r2;
    double nMass, tMass, bounce, jnAcc, jtAcc, jBias, bias;
    uint32_t hash;
} ora_contact;

typedef struct {           /* struct cpArbiter (the fields the solver reads) */
    int state, stamp, count, a, b; /* a, b: shape slots in cpCollide's order */
    ora_contact con[2];
    ora_v2 n;
    double u;              /* friction a->u * b->u; elasticity is 0 for every shape on this path */
} ora_arbiter;

typedef struct {
    int stamp;                          /* space->stamp */
    double prev_dt;                     /* space->curr_dt of the previous step (0 before the first) */
    ora_body tbody[ORA_N_TRAFFIC];
    ora_poly tshape[ORA_N_TRAFFIC];
    ora_body gbody[ORA_MAX_GOALS];      /* goal circle bodies, by original goal index */
    int goal_in_space;                  /* bit g: goal g's body/shape still in the space */
    int last_arbiters;                  /* solver list length of the last step (inspection) */
    ora_body static_body;               /* the banks' body: all zero (cpBodyNewStatic at the origin) */
    ora_arbiter arb[ORA_N_SLOTS][ORA_N_SLOTS]; /* cachedArbiters keyed by (lower slot, higher slot) */
    uint32_t pair_id[ORA_N_SLOTS][ORA_N_SLOTS]; /* ORA_VAR_GJK_WARM only: cpCollisionID of the broadphase pair */
} ora_dyn;

typedef struct {
    ora_config cfg;
    ora_body ship;
    ora_poly ship_shape;
    double ship_moment;
    ora_poly bank[2];
    int n_goals_alive;                 /* len(self.goals) */
    ora_v2 goal_p[ORA_MAX_GOALS];      /* self.goals in list order (consumed ones are removed) */
    int goal_id[ORA_MAX_GOALS];        /* original index of each remaining goal */
    int rudder;
    ora_v2 thrust_pt;
    double lidar_vals[ORA_MAX_BEAMS];
    int colliding, goal_reached;
    int step_count;
    double reward, cumulative_reward;
    int n_states;                      /* 6 + n_beams */
    double states[ORA_MAX_HISTORY * (6 + ORA_MAX_BEAMS)]; /* deque, oldest first */
    /* bank mode (auto-reset) */
    int map_id;
    int64_t episodes;
    ora_dyn dyn;                       /* used only when cfg.n_traffic > 0 */
    /* ORA_VAR_CHECK_SAT: player pairs that passed the AABB test; disagreements SAT vs cpCollide(player, other) and
     * vs cpCollide(other, player); pairs where cpCollide's signed distance was within 1e-9 of zero */
    int64_t sat_checked, sat_disagree_ab, sat_disagree_ba, sat_near_zero;
} ora_world;

/* ---- geometry primitives (Chipmunk restated) ---- */
int ora_convex_hull(int count, const double *verts_xy, double *out_xy);
double ora_moment_for_poly(double m, int count, const double *verts_xy);
void ora_poly_init(ora_poly *poly, int count, const double *verts_xy); /* hulls its input */
void ora_poly_update(ora_poly *poly, ora_v2 p, ora_v2 rot);
double ora_poly_point_query(const ora_poly *poly, ora_v2 p, ora_v2 *closest);
int ora_poly_segment_query(const ora_poly *poly, ora_v2 a, ora_v2 b, double radius, ora_seg_info *info);
int ora_polys_collide(const ora_poly *a, const ora_poly *b);
int ora_circle_poly_collide(ora_v2 c, double r, const ora_poly *poly);
int ora_polys_collide_v(const ora_poly *a, const ora_poly *b, int strict);           /* strict: ORA_VAR_TOUCH_STRICT */
int ora_circle_poly_collide_v(ora_v2 c, double r, const ora_poly *poly, int strict);
void ora_batch_counters(const ora_world *ws, int n, int64_t *out4); /* sums of the four sat_* counters */

/* cpBody primitives (the same functions the oracle world steps with; also what tests/golden/shims/pymunk calls) */
void ora_body_update_position(ora_body *b, double dt);
void ora_body_update_velocity(ora_body *b, double damping, double dt);
void ora_body_apply_force_at_local_point(ora_body *b, ora_v2 force, ora_v2 point);
int ora_circle_segment_query(ora_v2 center, double r1, ora_v2 a, ora_v2 b, double r2, ora_seg_info *info);

/* hooks of the traffic-capable pymunk stand-in (tests/golden/shims): shadow world in, cpSpaceStep, everything back out */
void ora_world_set_ship(ora_world *w, const ora_body *b);
void ora_world_get_ship(const ora_world *w, ora_body *b);
void ora_world_space_step(ora_world *w);

/* ---- world ---- */
void ora_default_config(ora_config *cfg);
void ora_world_init(ora_world *w, const ora_config *cfg);
/* ShipGame.reset + ShipEnv.reset given the map polygons and goal positions the host RNG produced. */
void ora_world_reset(ora_world *w, const double *left_xy, const double *right_xy, const double *goals_xy,
                     double *obs_out);
/* goal placement helper, game.py:322-325: returns 1 and (lo,hi) for np.random.uniform(lo,hi); 0 = fallback */
int ora_goal_x_range(const ora_world *w, double y, double *lo, double *hi);
void ora_world_step(ora_world *w, int action, double *obs_out, double *reward, uint8_t *done);
int ora_world_sizeof(void);
/* debug/inspection dump: [x,y,vx,vy,angle,w,rudder,step_count,n_goals_alive,colliding,goal_reached,map_id,
 * cumulative_reward,alive_mask,episodes,bb_l,bb_b,bb_r,bb_t] */
#define ORA_PEEK_LEN 19
void ora_world_peek(const ora_world *w, double *out);
ora_world *ora_world_at(ora_world *ws, int i);
/* config 4 inspection: [T_k: x,y,angle,vx,vy,w]x3, [goal g (original index): x,y,vx,vy]x5, goal_in_space mask,
 * number of arbiters processed by the solver in the last step */
#define ORA_PEEK_DYN_LEN (6 * ORA_N_TRAFFIC + 4 * 5 + 2)
void ora_world_peek_dyn(const ora_world *w, double *out);
/* ---- config 4 internals (ssg_dynamics.c), called by ora_world_reset / space_step ---- */
void ora_world_poke_traffic(ora_world *w, int k, const double *v6 /* x,y,angle,vx,vy,w */);
void ora_dyn_reset(ora_world *w);
void ora_dyn_integrate(ora_world *w);
void ora_dyn_collide_solve(ora_world *w, int reached_mask);
/* narrowphase exposed for unit tests: cpCollide(a, b) restated.  Returns the contact count (0..2) and fills
 * n, and per contact the absolute points p1/p2 and the hash. */
int ora_collide_poly_poly(const ora_poly *a, const ora_poly *b, int slot_a, int slot_b, ora_v2 *n, ora_v2 *p1,
                          ora_v2 *p2, uint32_t *hash, double *dist);
int ora_collide_circle_poly(ora_v2 c, double r, const ora_poly *b, ora_v2 *n, ora_v2 *p1, ora_v2 *p2, double *dist);
void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* ---- batched driver with a map bank and auto-reset (VecEnv semantics), OpenMP over envs ---- */
typedef struct {
    int n_maps;
    const double *polys;  /* [n_maps][2][12][2] */
    const double *goals;  /* [n_maps][n_goals][2] */
    int ring;             /* 0: an auto-reset moves an env to map (m+1) mod n_maps; R > 0: env e owns maps [e*R, e*R + R) as a
                             ring, one brand-new world per episode (the HIP path's map_ring mode) */
} ora_bank;

void ora_batch_reset(ora_world *ws, int n, const ora_config *cfg, const ora_bank *bank, const int32_t *map_ids,
                     double *obs /* [n][H*F] */);
void ora_batch_step(ora_world *ws, int n, const ora_bank *bank, const int32_t *actions, double *obs, double *reward,
                    uint8_t *done, int auto_reset, int n_threads);
void ora_batch_auto_reset(ora_world *ws, int n, const ora_bank *bank, const uint8_t *done, double *obs);
/* counter-based action stream shared with the HIP side (Philox4x32-10) */
int32_t ora_action(uint64_t seed, uint64_t step, uint64_t env_id);
void ora_fill_actions(uint64_t seed, uint64_t step0, int K, int64_t env_base, int n, int32_t *out /* [K][n] */);
/* timed random-action rollout for bench.py's cpu_baseline: returns env-steps done */
int64_t ora_rollout(ora_world *ws, int n, const ora_bank *bank, uint64_t seed, int64_t env_base, int K, int n_threads,
                    double *obs, double *reward, uint8_t *done);
int ora_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
