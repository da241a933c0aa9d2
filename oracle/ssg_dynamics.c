/*
 * ssg_dynamics.c — CPU ORACLE, config 4 (BASELINE configs[3]): traffic ships, dynamic goal bodies and the
 * Chipmunk2D contact solver.  TEST INFRASTRUCTURE ONLY (see ssg_oracle.h).  PARITY UNPINNED (ibid.).
 *
 * Reference call sites:  ShipGame.add_default_traffic game.py:279-286, add_ship game.py:117-131,
 * add_goal game.py:77-95 (goals are mass-1 dynamic circle bodies), Ship.__init__ models.py:87-111
 * (friction 0.7, collision_type 1), setup_collision_handlers game.py:288-298, space.step game.py:194.
 *
 * What is restated (Chipmunk2D 7.0.x, the library pymunk 5.4.0 bundles; published algorithm, from the
 * library's documented behaviour — SURVEY.md App. A.4/A.7/A.8):
 *   cpCollision.c   GJK / EPA closest points, SupportEdgeForPoly, ContactPoints (edge clipping),
 *                   CircleToCircle, CircleToPoly, PolyToPoly
 *   cpArbiter.c     cpArbiterUpdate (contact-hash warm start), cpArbiterPreStep,
 *                   cpArbiterApplyCachedImpulse, cpArbiterApplyImpulse
 *   cpSpaceStep.c   cpSpaceStep ordering, cpSpaceCollideShapes, cpSpaceArbiterSetFilter (persistence 3)
 *   cpBody.c        cpBodyUpdatePosition (with v_bias/w_bias), cpBodyUpdateVelocity
 *
 * Named assumptions that a real pymunk run could overturn (each isolated below):
 *   ORDER   the order in which the broadphase reports pairs (cpBBTree) decides the solver's arbiter order
 *           and which poly is "a" in a poly-poly pair.  It depends on the tree's insertion history and is
 *           not restated; the canonical order used here is: dynamic shapes in space-insertion order
 *           (goals, player, traffic), each against the statics first and then against earlier dynamics;
 *           a = the later-inserted dynamic shape for dynamic-static pairs, the earlier one otherwise
 *           (cpCollide then forces circle-before-poly).
 *   GJK-ID  the broadphase pair caches a collision id that warm-starts GJK; here every query starts cold
 *           (id 0), i.e. from the bounding-box-centre axis.
 *   PLAYER  every player contact with a type-1 shape fires collide_ship => colliding => done
 *           (ship_env.py:115-134), so the player's arbiters never influence a later observation and are
 *           not solved (same argument as configs 1-3, ssg_oracle.c header).
 */
#include <stdlib.h>
#include <string.h>

#include "ssg_vec.h"

#define COLLISION_SLOP 0.1                 /* cpSpace default collisionSlop                     */
#define COLLISION_PERSISTENCE 3            /* cpSpace default collisionPersistence              */
#define SOLVER_ITERATIONS 10               /* cpSpace default iterations                        */
#define MAX_GJK_ITERATIONS 30
#define MAX_EPA_ITERATIONS 30
#define SHIP_FRICTION 0.7                  /* models.py:98; banks and goals keep the default 0  */

static double collision_bias(void) { return pow(1.0 - 0.1, 60.0); } /* cpSpace default collisionBias */

static const double SHIP_TEMPLATE[5][2] = {{0, 0}, {0, 10}, {5, 15}, {10, 10}, {10, 0}}; /* models.py:6 */
/* add_default_traffic game.py:284-286: (x, y, width, height) */
static const double TRAFFIC[ORA_N_TRAFFIC][4] = {{100, 200, 1, 1}, {300, 200, 1.5, 2}, {400, 350, 1, 3}};

/* ------------------------------------------------------------------------------------------------
 * cpCollision.c
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int is_circle;
    const ora_poly *poly;
    ora_v2 tc;                      /* circle centre */
    double r;                       /* circle radius (poly radius is 0) */
    double bb_l, bb_b, bb_r, bb_t;
    uint32_t hashid;
} shape_ref;

typedef struct { ora_v2 p; uint32_t index; } support_pt;
typedef struct { ora_v2 a, b, ab; uint32_t id; } mink_pt;
typedef struct { ora_v2 a, b, n; double d; uint32_t id; } closest_pts;
typedef struct { ora_v2 p; uint32_t hash; } edge_pt;
typedef struct { edge_pt a, b; double r; ora_v2 n; } edge_t;

static shape_ref ref_poly(const ora_poly *p, uint32_t hashid)
{
    shape_ref s;
    memset(&s, 0, sizeof(s));
    s.poly = p; s.bb_l = p->bb_l; s.bb_b = p->bb_b; s.bb_r = p->bb_r; s.bb_t = p->bb_t; s.hashid = hashid;
    return s;
}
static shape_ref ref_circle(ora_v2 c, double r, uint32_t hashid)
{
    shape_ref s;
    memset(&s, 0, sizeof(s));
    s.is_circle = 1; s.tc = c; s.r = r; s.hashid = hashid;
    s.bb_l = c.x - r; s.bb_b = c.y - r; s.bb_r = c.x + r; s.bb_t = c.y + r; /* cpCircleShapeCacheData */
    return s;
}

static inline int bb_intersects(const shape_ref *a, const shape_ref *b)
{
    return a->bb_l <= b->bb_r && b->bb_l <= a->bb_r && a->bb_b <= b->bb_t && b->bb_b <= a->bb_t;
}
static inline ora_v2 bb_center(const shape_ref *s)
{
    return vlerp(V(s->bb_l, s->bb_b), V(s->bb_r, s->bb_t), 0.5);
}

static int poly_support_index(const ora_poly *p, ora_v2 n)
{
    double max = -INFINITY;
    int index = 0;
    for (int i = 0; i < p->count; i++) {
        double d = vdot(p->wv[i], n);
        if (d > max) { max = d; index = i; }
    }
    return index;
}
static support_pt support_point(const shape_ref *s, ora_v2 n)
{
    support_pt r;
    if (s->is_circle) { r.p = s->tc; r.index = 0; }
    else { int i = poly_support_index(s->poly, n); r.p = s->poly->wv[i]; r.index = (uint32_t)i; }
    return r;
}
static inline mink_pt mink_new(support_pt a, support_pt b)
{
    mink_pt m;
    m.a = a.p; m.b = b.p; m.ab = vsub(b.p, a.p);
    m.id = (a.index & 0xFF) << 8 | (b.index & 0xFF);
    return m;
}
static inline mink_pt support(const shape_ref *s1, const shape_ref *s2, ora_v2 n)
{
    return mink_new(support_point(s1, vneg(n)), support_point(s2, n));
}

static inline double closest_t(ora_v2 a, ora_v2 b)
{
    ora_v2 delta = vsub(b, a);
    return -cfclamp(vdot(delta, vadd(a, b)) / vlengthsq(delta), -1.0, 1.0);
}
static inline ora_v2 lerp_t(ora_v2 a, ora_v2 b, double t)
{
    double ht = 0.5 * t;
    return vadd(vmult(a, 0.5 - ht), vmult(b, 0.5 + ht));
}
static inline double closest_dist(ora_v2 v0, ora_v2 v1) { return vlengthsq(lerp_t(v0, v1, closest_t(v0, v1))); }

static closest_pts closest_new(mink_pt v0, mink_pt v1)
{
    double t = closest_t(v0.ab, v1.ab);
    ora_v2 p = lerp_t(v0.ab, v1.ab, t);
    closest_pts r;
    r.a = lerp_t(v0.a, v1.a, t);
    r.b = lerp_t(v0.b, v1.b, t);
    r.id = (v0.id & 0xFFFF) << 16 | (v1.id & 0xFFFF);
    ora_v2 delta = vsub(v1.ab, v0.ab);
    ora_v2 n = vnormalize(vrperp(delta));
    double d = vdot(n, p);
    if (d <= 0.0 || (-1.0 < t && t < 1.0)) {
        r.n = n; r.d = d;            /* overlapping, or a regular vertex/edge case */
    } else {
        double d2 = vlength(p);      /* vertex/vertex */
        r.n = vmult(p, 1.0 / (d2 + DBL_MIN));
        r.d = d2;
    }
    return r;
}

static closest_pts epa(const shape_ref *s1, const shape_ref *s2, mink_pt v0, mink_pt v1, mink_pt v2)
{
    mink_pt hull[MAX_EPA_ITERATIONS + 4], hull2[MAX_EPA_ITERATIONS + 4];
    int count = 3;
    hull[0] = v0; hull[1] = v1; hull[2] = v2;
    for (int iteration = 1;; iteration++) {
        int mini = 0;
        double min_dist = INFINITY;
        for (int j = 0, i = count - 1; j < count; i = j, j++) {
            double d = closest_dist(hull[i].ab, hull[j].ab);
            if (d < min_dist) { min_dist = d; mini = i; }
        }
        mink_pt e0 = hull[mini], e1 = hull[(mini + 1) % count];
        mink_pt p = support(s1, s2, vperp(vsub(e1.ab, e0.ab)));
        double area2x = vcross(vsub(e1.ab, e0.ab), vadd(vsub(p.ab, e0.ab), vsub(p.ab, e1.ab)));
        if (area2x > 0.0 && iteration < MAX_EPA_ITERATIONS) {
            int count2 = 1;
            hull2[0] = p;
            for (int i = 0; i < count; i++) {
                int index = (mini + 1 + i) % count;
                ora_v2 h0 = hull2[count2 - 1].ab;
                ora_v2 h1 = hull[index].ab;
                ora_v2 h2 = (i + 1 < count ? hull[(index + 1) % count] : p).ab;
                if (vcross(vsub(h2, h0), vsub(h1, h0)) > 0.0) hull2[count2++] = hull[index];
            }
            memcpy(hull, hull2, sizeof(mink_pt) * (size_t)count2);
            count = count2;
        } else {
            return closest_new(e0, e1);
        }
    }
}

static support_pt shape_point(const shape_ref *s, uint32_t i) /* ShapePoint: vertex i of a poly, the centre of a circle */
{
    support_pt r;
    if (s->is_circle) { r.p = s->tc; r.index = 0; }
    else { uint32_t k = i < (uint32_t)s->poly->count ? i : 0u; r.p = s->poly->wv[k]; r.index = k; }
    return r;
}

static closest_pts gjk(const shape_ref *s1, const shape_ref *s2, uint32_t *id)
{
    /* GJK-ID: cold start from the axis perpendicular to the line between the bounding-box centres, unless the
     * ORA_VAR_GJK_WARM variant hands in the pair's cached collision id (cpCollide's `id` argument): then the
     * search restarts from the two Minkowski vertices the previous query ended on. */
    mink_pt v0, v1;
    if (id && *id) {
        v0 = mink_new(shape_point(s1, (*id >> 24) & 0xFF), shape_point(s2, (*id >> 16) & 0xFF));
        v1 = mink_new(shape_point(s1, (*id >> 8) & 0xFF), shape_point(s2, (*id) & 0xFF));
    } else {
        ora_v2 axis = vperp(vsub(bb_center(s1), bb_center(s2)));
        v0 = support(s1, s2, axis);
        v1 = support(s1, s2, vneg(axis));
    }
    closest_pts out;
    int iteration = 1;
    for (;;) {
        if (iteration > MAX_GJK_ITERATIONS) { out = closest_new(v0, v1); break; }
        ora_v2 delta = vsub(v1.ab, v0.ab);
        if (vcross(delta, vadd(v0.ab, v1.ab)) > 0.0) {
            mink_pt tmp = v0; v0 = v1; v1 = tmp; /* origin is behind the axis: flip, same iteration */
            continue;
        }
        double t = closest_t(v0.ab, v1.ab);
        ora_v2 n = (-1.0 < t && t < 1.0) ? vperp(delta) : vneg(lerp_t(v0.ab, v1.ab, t));
        mink_pt p = support(s1, s2, n);
        if (vcross(vsub(v1.ab, p.ab), vadd(v1.ab, p.ab)) > 0.0 && vcross(vsub(v0.ab, p.ab), vadd(v0.ab, p.ab)) < 0.0)
            { out = epa(s1, s2, v0, p, v1); break; } /* the triangle v0, p, v1 contains the origin */
        if (vdot(p.ab, n) <= cfmax(vdot(v0.ab, n), vdot(v1.ab, n))) { out = closest_new(v0, v1); break; }
        if (closest_dist(v0.ab, p.ab) < closest_dist(p.ab, v1.ab)) v1 = p; else v0 = p;
        iteration++;
    }
    if (id) *id = out.id;
    return out;
}

/* CP_HASH_PAIR mixes pointer-sized ids with a multiplicative constant; only equality of hashes is ever used
 * (cpArbiterUpdate's warm-start match), so a collision-free encoding is behaviourally identical:
 * edge-point hash = slot*16 + vertex + 1 (1..188), contact hash = (hash1 << 8) | hash2 (never 0). */
#define EDGE_HASH(hashid, i) ((uint32_t)(hashid) * 16u + (uint32_t)(i) + 1u)
#define CONTACT_HASH(h1, h2) ((uint32_t)(h1) << 8 | (uint32_t)(h2))

static edge_t support_edge(const shape_ref *s, ora_v2 n)
{
    const ora_poly *p = s->poly;
    int count = p->count;
    int i1 = poly_support_index(p, n);
    int i0 = (i1 - 1 + count) % count;
    int i2 = (i1 + 1) % count;
    edge_t e;
    e.r = 0.0;
    if (vdot(n, p->wn[i1]) > vdot(n, p->wn[i2])) {
        e.a.p = p->wv[i0]; e.a.hash = EDGE_HASH(s->hashid, i0);
        e.b.p = p->wv[i1]; e.b.hash = EDGE_HASH(s->hashid, i1);
        e.n = p->wn[i1];
    } else {
        e.a.p = p->wv[i1]; e.a.hash = EDGE_HASH(s->hashid, i1);
        e.b.p = p->wv[i2]; e.b.hash = EDGE_HASH(s->hashid, i2);
        e.n = p->wn[i2];
    }
    return e;
}

typedef struct { int count; ora_v2 n; ora_v2 p1[2], p2[2]; uint32_t hash[2]; double d; } collision_info;

static void contact_points(edge_t e1, edge_t e2, closest_pts points, collision_info *info)
{
    double mindist = e1.r + e2.r;
    if (points.d <= mindist) {
        ora_v2 n = info->n = points.n;
        double d_e1_a = vcross(e1.a.p, n), d_e1_b = vcross(e1.b.p, n);
        double d_e2_a = vcross(e2.a.p, n), d_e2_b = vcross(e2.b.p, n);
        double e1_denom = 1.0 / (d_e1_b - d_e1_a + DBL_MIN);
        double e2_denom = 1.0 / (d_e2_b - d_e2_a + DBL_MIN);
        {
            ora_v2 p1 = vadd(vmult(n, e1.r), vlerp(e1.a.p, e1.b.p, cfclamp01((d_e2_b - d_e1_a) * e1_denom)));
            ora_v2 p2 = vadd(vmult(n, -e2.r), vlerp(e2.a.p, e2.b.p, cfclamp01((d_e1_a - d_e2_a) * e2_denom)));
            double dist = vdot(vsub(p2, p1), n);
            if (dist <= 0.0) {
                info->p1[info->count] = p1; info->p2[info->count] = p2;
                info->hash[info->count] = CONTACT_HASH(e1.a.hash, e2.b.hash);
                info->count++;
            }
        }
        {
            ora_v2 p1 = vadd(vmult(n, e1.r), vlerp(e1.a.p, e1.b.p, cfclamp01((d_e2_a - d_e1_a) * e1_denom)));
            ora_v2 p2 = vadd(vmult(n, -e2.r), vlerp(e2.a.p, e2.b.p, cfclamp01((d_e1_b - d_e2_a) * e2_denom)));
            double dist = vdot(vsub(p2, p1), n);
            if (dist <= 0.0) {
                info->p1[info->count] = p1; info->p2[info->count] = p2;
                info->hash[info->count] = CONTACT_HASH(e1.b.hash, e2.a.hash);
                info->count++;
            }
        }
    }
}

static void poly_to_poly(const shape_ref *a, const shape_ref *b, collision_info *info, uint32_t *id)
{
    closest_pts points = gjk(a, b, id);
    info->d = points.d;
    if (points.d - 0.0 - 0.0 <= 0.0)
        contact_points(support_edge(a, points.n), support_edge(b, vneg(points.n)), points, info);
}

static void circle_to_poly(const shape_ref *c, const shape_ref *p, collision_info *info, uint32_t *id)
{
    closest_pts points = gjk(c, p, id);
    double mindist = c->r + 0.0;
    info->d = points.d;
    if (points.d <= mindist) {
        ora_v2 n = info->n = points.n;
        info->p1[0] = vadd(points.a, vmult(n, c->r));
        info->p2[0] = vadd(points.b, vmult(n, -0.0));
        info->hash[0] = 0;
        info->count = 1;
    }
}

static void circle_to_circle(const shape_ref *c1, const shape_ref *c2, collision_info *info)
{
    double mindist = c1->r + c2->r;
    ora_v2 delta = vsub(c2->tc, c1->tc);
    double distsq = vlengthsq(delta);
    if (distsq < mindist * mindist) {
        double dist = sqrt(distsq);
        ora_v2 n = info->n = (dist ? vmult(delta, 1.0 / dist) : V(1.0, 0.0));
        info->p1[0] = vadd(c1->tc, vmult(n, c1->r));
        info->p2[0] = vadd(c2->tc, vmult(n, -c2->r));
        info->hash[0] = 0;
        info->count = 1;
        info->d = dist - mindist;
    }
}

int ora_collide_poly_poly(const ora_poly *a, const ora_poly *b, int slot_a, int slot_b, ora_v2 *n, ora_v2 *p1,
                          ora_v2 *p2, uint32_t *hash, double *dist)
{
    shape_ref ra = ref_poly(a, (uint32_t)slot_a), rb = ref_poly(b, (uint32_t)slot_b);
    collision_info info;
    memset(&info, 0, sizeof(info));
    poly_to_poly(&ra, &rb, &info, NULL);
    *n = info.n; *dist = info.d;
    for (int i = 0; i < info.count; i++) { p1[i] = info.p1[i]; p2[i] = info.p2[i]; hash[i] = info.hash[i]; }
    return info.count;
}

int ora_collide_circle_poly(ora_v2 c, double r, const ora_poly *b, ora_v2 *n, ora_v2 *p1, ora_v2 *p2, double *dist)
{
    shape_ref rc = ref_circle(c, r, 0), rb = ref_poly(b, 1);
    collision_info info;
    memset(&info, 0, sizeof(info));
    circle_to_poly(&rc, &rb, &info, NULL);
    *n = info.n; *dist = info.d;
    if (info.count) { p1[0] = info.p1[0]; p2[0] = info.p2[0]; }
    return info.count;
}

/* ------------------------------------------------------------------------------------------------
 * World glue: slots, bodies, shapes
 * ---------------------------------------------------------------------------------------------- */
static ora_body *slot_body(ora_world *w, int slot)
{
    if (slot < ORA_SLOT_GOAL0) return &w->dyn.static_body; /* zero: p = v = 0, m_inv = i_inv = 0 */
    if (slot < ORA_SLOT_PLAYER) return &w->dyn.gbody[slot - ORA_SLOT_GOAL0];
    if (slot == ORA_SLOT_PLAYER) return &w->ship;
    return &w->dyn.tbody[slot - ORA_SLOT_TRAFFIC0];
}
static shape_ref slot_shape(ora_world *w, int slot)
{
    if (slot < ORA_SLOT_GOAL0) return ref_poly(&w->bank[slot], (uint32_t)slot);
    if (slot < ORA_SLOT_PLAYER) return ref_circle(w->dyn.gbody[slot - ORA_SLOT_GOAL0].p, w->cfg.goal_radius, (uint32_t)slot);
    if (slot == ORA_SLOT_PLAYER) return ref_poly(&w->ship_shape, (uint32_t)slot);
    return ref_poly(&w->dyn.tshape[slot - ORA_SLOT_TRAFFIC0], (uint32_t)slot);
}
static double slot_friction(int slot) { return slot >= ORA_SLOT_PLAYER ? SHIP_FRICTION : 0.0; }

void ora_dyn_reset(ora_world *w)
{
    /* a fresh pm.Space() (game.py:268): stamp 0, no cached arbiters, curr_dt 0 */
    ora_dyn *d = &w->dyn;
    const ora_config *c = &w->cfg;
    memset(d, 0, sizeof(*d));
    /* add_goal game.py:77-95: mass 1, moment_for_circle(1, 0, 5) = 12.5, radius 5, friction default 0 */
    for (int g = 0; g < c->n_goals; g++) {
        ora_body *b = &d->gbody[g];
        b->p = w->goal_p[g];
        b->m_inv = 1.0 / 1.0;
        b->i_inv = 1.0 / (1.0 * 0.5 * (0.0 * 0.0 + c->goal_radius * c->goal_radius)); /* cpMomentForCircle */
        b->rot = V(1.0, 0.0);
        d->goal_in_space |= 1 << g;
    }
    /* add_default_traffic game.py:279-286 -> add_ship -> Ship.__init__ (mass 5, moment about the local origin) */
    for (int k = 0; k < c->n_traffic && k < ORA_N_TRAFFIC; k++) {
        double pts[10];
        for (int i = 0; i < 5; i++) {
            pts[2 * i] = SHIP_TEMPLATE[i][0] * TRAFFIC[k][2];
            pts[2 * i + 1] = SHIP_TEMPLATE[i][1] * TRAFFIC[k][3];
        }
        ora_body *b = &d->tbody[k];
        b->m_inv = 1.0 / c->ship_mass;
        b->i_inv = 1.0 / ora_moment_for_poly(c->ship_mass, 5, pts);
        b->p = V(TRAFFIC[k][0], TRAFFIC[k][1]);
        b->a = 0.0;
        b->rot = V(cos(0.0), sin(0.0));
        ora_poly_init(&d->tshape[k], 5, pts);
        ora_poly_update(&d->tshape[k], b->p, b->rot); /* space.add */
    }
}

/* cpBodyUpdatePosition */
static void update_position(ora_body *b, double dt)
{
    b->p = vadd(b->p, vmult(vadd(b->v, b->v_bias), dt));
    b->a = b->a + (b->w + b->w_bias) * dt;
    b->rot = V(cos(b->a), sin(b->a));
    b->v_bias = V(0, 0);
    b->w_bias = 0.0;
}
/* cpBodyUpdateVelocity, gravity = 0 */
static void update_velocity(ora_body *b, double damping, double dt)
{
    b->v = vadd(vmult(b->v, damping), vmult(vadd(V(0, 0), vmult(b->f, b->m_inv)), dt));
    b->w = b->w * damping + b->t * b->i_inv * dt;
    b->f = V(0, 0);
    b->t = 0.0;
}

/* ------------------------------------------------------------------------------------------------
 * cpArbiter.c
 * ---------------------------------------------------------------------------------------------- */
static inline double k_scalar_body(const ora_body *b, ora_v2 r, ora_v2 n)
{
    double rcn = vcross(r, n);
    return b->m_inv + b->i_inv * rcn * rcn;
}
static inline double k_scalar(const ora_body *a, const ora_body *b, ora_v2 r1, ora_v2 r2, ora_v2 n)
{
    return k_scalar_body(a, r1, n) + k_scalar_body(b, r2, n);
}
static inline ora_v2 relative_velocity(const ora_body *a, const ora_body *b, ora_v2 r1, ora_v2 r2)
{
    ora_v2 v1_sum = vadd(a->v, vmult(vperp(r1), a->w));
    ora_v2 v2_sum = vadd(b->v, vmult(vperp(r2), b->w));
    return vsub(v2_sum, v1_sum);
}
static inline void apply_impulse(ora_body *b, ora_v2 j, ora_v2 r)
{
    b->v = vadd(b->v, vmult(j, b->m_inv));
    b->w += b->i_inv * vcross(r, j);
}
static inline void apply_impulses(ora_body *a, ora_body *b, ora_v2 r1, ora_v2 r2, ora_v2 j)
{
    apply_impulse(a, vneg(j), r1);
    apply_impulse(b, j, r2);
}
static inline void apply_bias_impulse(ora_body *b, ora_v2 j, ora_v2 r)
{
    b->v_bias = vadd(b->v_bias, vmult(j, b->m_inv));
    b->w_bias += b->i_inv * vcross(r, j);
}
static inline void apply_bias_impulses(ora_body *a, ora_body *b, ora_v2 r1, ora_v2 r2, ora_v2 j)
{
    apply_bias_impulse(a, vneg(j), r1);
    apply_bias_impulse(b, j, r2);
}

/* cpArbiterUpdate: new contact set, accumulated impulses inherited from old contacts with the same hash */
static void arbiter_update(ora_world *w, ora_arbiter *arb, int a, int b, const collision_info *info)
{
    ora_contact fresh[2];
    const ora_body *ba = slot_body(w, a), *bb = slot_body(w, b);
    for (int i = 0; i < info->count; i++) {
        ora_contact *con = &fresh[i];
        memset(con, 0, sizeof(*con));
        con->r1 = vsub(info->p1[i], ba->p);
        con->r2 = vsub(info->p2[i], bb->p);
        con->hash = info->hash[i];
        con->jnAcc = con->jtAcc = 0.0;
        for (int j = 0; j < arb->count; j++) {
            const ora_contact *old = &arb->con[j];
            if (con->hash == old->hash) { con->jnAcc = old->jnAcc; con->jtAcc = old->jtAcc; }
        }
    }
    for (int i = 0; i < info->count; i++) arb->con[i] = fresh[i];
    arb->count = info->count;
    arb->n = info->n;
    arb->a = a; arb->b = b;
    arb->u = slot_friction(a) * slot_friction(b);
    if (arb->state == ORA_ARB_CACHED) arb->state = ORA_ARB_FIRST;
}

static void arbiter_prestep(ora_world *w, ora_arbiter *arb, double dt, double slop, double bias)
{
    const ora_body *a = slot_body(w, arb->a), *b = slot_body(w, arb->b);
    ora_v2 n = arb->n;
    ora_v2 body_delta = vsub(b->p, a->p);
    for (int i = 0; i < arb->count; i++) {
        ora_contact *con = &arb->con[i];
        con->nMass = 1.0 / k_scalar(a, b, con->r1, con->r2, n);
        con->tMass = 1.0 / k_scalar(a, b, con->r1, con->r2, vperp(n));
        double dist = vdot(vadd(vsub(con->r2, con->r1), body_delta), n);
        con->bias = -bias * cfmin(0.0, dist + slop) / dt;
        con->jBias = 0.0;
        con->bounce = vdot(relative_velocity(a, b, con->r1, con->r2), n) * 0.0; /* arb->e = 0 */
    }
}

static void arbiter_apply_cached(ora_world *w, ora_arbiter *arb, double dt_coef)
{
    if (arb->state == ORA_ARB_FIRST) return; /* cpArbiterIsFirstContact */
    ora_body *a = slot_body(w, arb->a), *b = slot_body(w, arb->b);
    for (int i = 0; i < arb->count; i++) {
        ora_contact *con = &arb->con[i];
        ora_v2 j = vrotate(arb->n, V(con->jnAcc, con->jtAcc));
        apply_impulses(a, b, con->r1, con->r2, vmult(j, dt_coef));
    }
}

static void arbiter_apply_impulse(ora_world *w, ora_arbiter *arb)
{
    ora_body *a = slot_body(w, arb->a), *b = slot_body(w, arb->b);
    ora_v2 n = arb->n;
    ora_v2 surface_vr = V(0, 0);
    double friction = arb->u;
    for (int i = 0; i < arb->count; i++) {
        ora_contact *con = &arb->con[i];
        double nMass = con->nMass;
        ora_v2 r1 = con->r1, r2 = con->r2;
        ora_v2 vb1 = vadd(a->v_bias, vmult(vperp(r1), a->w_bias));
        ora_v2 vb2 = vadd(b->v_bias, vmult(vperp(r2), b->w_bias));
        ora_v2 vr = vadd(relative_velocity(a, b, r1, r2), surface_vr);
        double vbn = vdot(vsub(vb2, vb1), n);
        double vrn = vdot(vr, n);
        double vrt = vdot(vr, vperp(n));
        double jbn = (con->bias - vbn) * nMass;
        double jbnOld = con->jBias;
        con->jBias = cfmax(jbnOld + jbn, 0.0);
        double jn = -(con->bounce + vrn) * nMass;
        double jnOld = con->jnAcc;
        con->jnAcc = cfmax(jnOld + jn, 0.0);
        double jtMax = friction * con->jnAcc;
        double jt = -vrt * con->tMass;
        double jtOld = con->jtAcc;
        con->jtAcc = cfclamp(jtOld + jt, -jtMax, jtMax);
        apply_bias_impulses(a, b, r1, r2, vmult(n, con->jBias - jbnOld));
        apply_impulses(a, b, r1, r2, vrotate(n, V(con->jnAcc - jnOld, con->jtAcc - jtOld)));
    }
}

/* ------------------------------------------------------------------------------------------------
 * cpSpaceStep for the traffic ships and goal bodies (the player's own update stays in ssg_oracle.c)
 * ---------------------------------------------------------------------------------------------- */
static ora_arbiter *arb_at(ora_world *w, int s1, int s2)
{
    return s1 < s2 ? &w->dyn.arb[s1][s2] : &w->dyn.arb[s2][s1];
}

/* cpSpaceCollideShapes for one candidate pair in cpCollide's a/b order; returns 1 if the arbiter goes on the
 * solver list.  (Default handler: begin/preSolve return true.) */
static int collide_pair(ora_world *w, int a, int b)
{
    const int var = w->cfg.variant;
    shape_ref sa = slot_shape(w, a), sb = slot_shape(w, b);
    uint32_t *id = NULL;
    if (var & ORA_VAR_GJK_WARM) id = a < b ? &w->dyn.pair_id[a][b] : &w->dyn.pair_id[b][a];
    if (!bb_intersects(&sa, &sb)) { /* queryReject; the broadphase pair (and its cached id) goes when the boxes part */
        if (id) *id = 0;
        return 0;
    }
    if ((var & ORA_VAR_SWAP_AB) && !sa.is_circle && !sb.is_circle) { /* ORDER variant: the other poly is "a" */
        shape_ref ts = sa; sa = sb; sb = ts;
        int ti = a; a = b; b = ti;
        if (id) *id = 0; /* (an id is only meaningful for one a/b order; the variant is run cold) */
    }
    collision_info info;
    memset(&info, 0, sizeof(info));
    if (sa.is_circle && sb.is_circle) circle_to_circle(&sa, &sb, &info);
    else if (sa.is_circle) circle_to_poly(&sa, &sb, &info, id);
    else poly_to_poly(&sa, &sb, &info, id);
    if (info.count == 0) return 0;
    ora_arbiter *arb = arb_at(w, a, b);
    if (arb->state == ORA_ARB_NONE) { /* cpArbiterInit */
        memset(arb, 0, sizeof(*arb));
        arb->state = ORA_ARB_FIRST;
    }
    arbiter_update(w, arb, a, b, &info);
    arb->stamp = w->dyn.stamp;
    return 1;
}

/* cpSpaceStep part 1 for the non-player bodies: positions and shape caches.  space_step() in ssg_oracle.c
 * calls this right after the player's own position update, so that the player's goal tests see this step's
 * goal positions, as they do inside the one cpSpaceStep of the reference. */
void ora_dyn_integrate(ora_world *w)
{
    ora_dyn *d = &w->dyn;
    const ora_config *c = &w->cfg;
    const double dt = c->dt;
    const int nt = c->n_traffic < ORA_N_TRAFFIC ? c->n_traffic : ORA_N_TRAFFIC;
    d->stamp++;
    /* arbiters that were on the solver list last step go back to "normal" */
    for (int i = 0; i < ORA_N_SLOTS; i++)
        for (int j = i + 1; j < ORA_N_SLOTS; j++)
            if (d->arb[i][j].state == ORA_ARB_FIRST) d->arb[i][j].state = ORA_ARB_NORMAL;
    /* (1) positions */
    for (int g = 0; g < c->n_goals; g++)
        if (d->goal_in_space >> g & 1) update_position(&d->gbody[g], dt);
    for (int k = 0; k < nt; k++) update_position(&d->tbody[k], dt);
    /* (2) shape cache */
    for (int k = 0; k < nt; k++) ora_poly_update(&d->tshape[k], d->tbody[k].p, d->tbody[k].rot);
    /* the player's view of the goals (self.goals, list order) follows the bodies */
    for (int i = 0; i < w->n_goals_alive; i++) w->goal_p[i] = d->gbody[w->goal_id[i]].p;
}

/* cpSpaceStep parts 2-5 for the non-player bodies: narrowphase, arbiter bookkeeping, velocity update, solver.
 * `reached_mask` = goals (original indices) the player touched this step: pymunk defers their space.remove
 * to the end of the step (game.py:252), so they still collide and are solved here, then leave the space. */
void ora_dyn_collide_solve(ora_world *w, int reached_mask)
{
    ora_dyn *d = &w->dyn;
    const ora_config *c = &w->cfg;
    const double dt = c->dt;
    const int nt = c->n_traffic < ORA_N_TRAFFIC ? c->n_traffic : ORA_N_TRAFFIC;
    /* (3) collide — ORDER assumption (file header) */
    ora_arbiter *list[64];
    int n_list = 0;
    for (int g = 0; g < c->n_goals; g++) {
        if (!(d->goal_in_space >> g & 1)) continue;
        int sg = ORA_SLOT_GOAL0 + g;
        for (int s = 0; s < 2; s++)
            if (collide_pair(w, sg, s)) list[n_list++] = arb_at(w, sg, s);              /* circle, poly */
        for (int h = 0; h < g; h++)
            if ((d->goal_in_space >> h & 1) && collide_pair(w, ORA_SLOT_GOAL0 + h, sg))
                list[n_list++] = arb_at(w, ORA_SLOT_GOAL0 + h, sg);                       /* circle, circle */
    }
    /* player pairs: handled by the caller (booleans only, PLAYER assumption) */
    for (int k = 0; k < nt; k++) {
        int st = ORA_SLOT_TRAFFIC0 + k;
        for (int s = 0; s < 2; s++)
            if (collide_pair(w, st, s)) list[n_list++] = arb_at(w, st, s);                /* ship, bank */
        for (int g = 0; g < c->n_goals; g++)
            if ((d->goal_in_space >> g & 1) && collide_pair(w, ORA_SLOT_GOAL0 + g, st))
                list[n_list++] = arb_at(w, ORA_SLOT_GOAL0 + g, st);                       /* circle, poly */
        for (int j = 0; j < k; j++)
            if (collide_pair(w, ORA_SLOT_TRAFFIC0 + j, st)) list[n_list++] = arb_at(w, ORA_SLOT_TRAFFIC0 + j, st);
    }
    /* cpSpaceArbiterSetFilter: separated arbiters become "cached", and are dropped after 3 stamps */
    for (int i = 0; i < ORA_N_SLOTS; i++)
        for (int j = i + 1; j < ORA_N_SLOTS; j++) {
            ora_arbiter *arb = &d->arb[i][j];
            if (arb->state == ORA_ARB_NONE) continue;
            int ticks = d->stamp - arb->stamp;
            if (ticks >= 1 && arb->state != ORA_ARB_CACHED) arb->state = ORA_ARB_CACHED;
            if (ticks >= COLLISION_PERSISTENCE) { arb->state = ORA_ARB_NONE; arb->count = 0; }
        }
    /* prestep */
    const double slop = COLLISION_SLOP;
    const double bias_coef = 1.0 - pow(collision_bias(), dt);
    for (int i = 0; i < n_list; i++) arbiter_prestep(w, list[i], dt, slop, bias_coef);
    /* (4) velocities */
    const double damping = pow(c->space_damping, dt);
    for (int g = 0; g < c->n_goals; g++)
        if (d->goal_in_space >> g & 1) update_velocity(&d->gbody[g], damping, dt);
    for (int k = 0; k < nt; k++) update_velocity(&d->tbody[k], damping, dt);
    /* (5) cached impulses + solver */
    const double dt_coef = (d->prev_dt == 0.0 ? 0.0 : dt / d->prev_dt);
    if (c->variant & ORA_VAR_ORDER_REVERSED) /* ORDER variant: the broadphase reported the pairs the other way round */
        for (int i = 0; i < n_list / 2; i++) { ora_arbiter *t = list[i]; list[i] = list[n_list - 1 - i]; list[n_list - 1 - i] = t; }
    for (int i = 0; i < n_list; i++) arbiter_apply_cached(w, list[i], dt_coef);
    for (int it = 0; it < SOLVER_ITERATIONS; it++)
        for (int i = 0; i < n_list; i++) arbiter_apply_impulse(w, list[i]);
    d->prev_dt = dt;
    d->last_arbiters = n_list;
    /* deferred space.remove(goal shape, body): cpSpaceRemoveShape also drops the shape's cached arbiters */
    for (int g = 0; g < c->n_goals; g++) {
        if (!(reached_mask >> g & 1)) continue;
        d->goal_in_space &= ~(1 << g);
        for (int s = 0; s < ORA_N_SLOTS; s++)
            if (s != ORA_SLOT_GOAL0 + g) arb_at(w, ORA_SLOT_GOAL0 + g, s)->state = ORA_ARB_NONE;
    }
}

void ora_world_peek_dyn(const ora_world *w, double *o)
{
    const ora_dyn *d = &w->dyn;
    int k = 0;
    for (int t = 0; t < ORA_N_TRAFFIC; t++) {
        const ora_body *b = &d->tbody[t];
        o[k++] = b->p.x; o[k++] = b->p.y; o[k++] = b->a; o[k++] = b->v.x; o[k++] = b->v.y; o[k++] = b->w;
    }
    for (int g = 0; g < 5; g++) {
        const ora_body *b = &d->gbody[g];
        o[k++] = b->p.x; o[k++] = b->p.y; o[k++] = b->v.x; o[k++] = b->v.y;
    }
    o[k++] = d->goal_in_space;
    o[k++] = d->last_arbiters;
}

/* test hook: overwrite a traffic ship's pose / velocity (mirrors writing the HIP path's state columns) */
void ora_world_poke_traffic(ora_world *w, int k, const double *v6)
{
    ora_body *b = &w->dyn.tbody[k];
    b->p = V(v6[0], v6[1]); b->a = v6[2]; b->rot = V(cos(b->a), sin(b->a));
    b->v = V(v6[3], v6[4]); b->w = v6[5];
    b->v_bias = V(0, 0); b->w_bias = 0.0;
    ora_poly_update(&w->dyn.tshape[k], b->p, b->rot);
}
