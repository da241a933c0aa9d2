/*
 * ssg_oracle.c — CPU ORACLE for the ShipEnv step/reset path.  TEST INFRASTRUCTURE ONLY.
 * See ssg_oracle.h for the usage rule and the "PARITY UNPINNED at the pymunk boundary" statement.
 *
 * Scalar double precision, one ora_world per env, written to mirror the reference's operation order:
 *   ShipEnv.step            ship_env.py:136-156      -> ora_world_step
 *   ShipGame.update         game.py:185-195          -> lidar_query + space_step
 *   LiDAR.query             models.py:39-76          -> lidar_query
 *   Ship.move_forward/rotate models.py:129-146       -> apply_action
 *   ShipGame.reset          game.py:260-277          -> ora_world_reset
 *   ShipEnv.reset           ship_env.py:171-184      -> ora_world_reset (obs part)
 * and the Chipmunk2D 7.0.x routines those call (published algorithm, restated from the library's
 * documented behaviour; pymunk 5.4.0 pins it, requirements.txt:78):
 *   cpConvexHull/QHullReduce, cpPolyShape SetVerts, cpPolyShapeCacheData, cpPolyShapePointQuery,
 *   cpShapeSegmentQuery, cpPolyShapeSegmentQuery, CircleSegmentQuery, cpMomentForPoly,
 *   cpBodyApplyForceAtLocalPoint, cpBodyUpdatePosition, cpBodyUpdateVelocity, cpSpaceStep ordering.
 *
 * Known, documented restriction: the impulse solver is not restated.  In the 1-ship
 * configurations every player contact sets `colliding` => done, and positions are integrated before the
 * narrowphase inside cpSpaceStep, so no solver output reaches an observation as long as the caller
 * resets on done (VecEnv semantics).
 */
#include "ssg_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "ssg_vec.h"

/* ------------------------------------------------------------------------------------------------
 * cpConvexHull (QuickHull, tol = 0): CCW hull, first vertex = lexicographic (x, then y) minimum,
 * collinear points dropped.  pymunk.Poly() hulls every vertex list it is given (models.py:96,180).
 * ---------------------------------------------------------------------------------------------- */
#define SWAPV(a, b) do { ora_v2 _t = (a); (a) = (b); (b) = _t; } while (0)

static int qhull_partition(ora_v2 *verts, int count, ora_v2 a, ora_v2 b, double tol)
{
    if (count == 0) return 0;
    double max = 0.0;
    int pivot = 0;
    ora_v2 delta = vsub(b, a);
    double value_tol = tol * vlength(delta);
    int head = 0;
    for (int tail = count - 1; head <= tail;) {
        double value = vcross(vsub(verts[head], a), delta);
        if (value > value_tol) {
            if (value > max) { max = value; pivot = head; }
            head++;
        } else {
            SWAPV(verts[head], verts[tail]);
            tail--;
        }
    }
    if (pivot != 0) SWAPV(verts[0], verts[pivot]);
    return head;
}

static int qhull_reduce(double tol, ora_v2 *verts, int count, ora_v2 a, ora_v2 pivot, ora_v2 b, ora_v2 *result)
{
    if (count < 0) {
        return 0;
    } else if (count == 0) {
        result[0] = pivot;
        return 1;
    } else {
        int left_count = qhull_partition(verts, count, a, pivot, tol);
        int index = qhull_reduce(tol, verts + 1, left_count - 1, a, verts[0], pivot, result);
        result[index++] = pivot;
        int right_count = qhull_partition(verts + left_count, count - left_count, pivot, b, tol);
        return index + qhull_reduce(tol, verts + left_count + 1, right_count - 1, pivot, verts[left_count], b,
                                    result + index);
    }
}

int ora_convex_hull(int count, const double *verts_xy, double *out_xy)
{
    ora_v2 buf[64];
    if (count <= 0) return 0;
    if (count > 64) count = 64;
    for (int i = 0; i < count; i++) buf[i] = V(verts_xy[2 * i], verts_xy[2 * i + 1]);
    /* cpLoopIndexes */
    int start = 0, end = 0;
    ora_v2 mn = buf[0], mx = buf[0];
    for (int i = 1; i < count; i++) {
        ora_v2 v = buf[i];
        if (v.x < mn.x || (v.x == mn.x && v.y < mn.y)) { mn = v; start = i; }
        else if (v.x > mx.x || (v.x == mx.x && v.y > mx.y)) { mx = v; end = i; }
    }
    int n;
    if (start == end) {
        n = 1;
    } else {
        SWAPV(buf[0], buf[start]);
        SWAPV(buf[1], buf[end == 0 ? start : end]);
        ora_v2 a = buf[0], b = buf[1];
        n = qhull_reduce(0.0, buf + 2, count - 2, a, b, a, buf + 1) + 1;
    }
    for (int i = 0; i < n; i++) { out_xy[2 * i] = buf[i].x; out_xy[2 * i + 1] = buf[i].y; }
    return n;
}

/* cpMomentForPoly(m, count, verts, offset=(0,0), r=0): moment about the LOCAL ORIGIN (models.py:89). */
double ora_moment_for_poly(double m, int count, const double *verts_xy)
{
    double sum1 = 0.0, sum2 = 0.0;
    for (int i = 0; i < count; i++) {
        ora_v2 v1 = V(verts_xy[2 * i] + 0.0, verts_xy[2 * i + 1] + 0.0);
        int j = (i + 1) % count;
        ora_v2 v2 = V(verts_xy[2 * j] + 0.0, verts_xy[2 * j + 1] + 0.0);
        double a = vcross(v2, v1);
        double b = vdot(v1, v1) + vdot(v1, v2) + vdot(v2, v2);
        sum1 += a * b;
        sum2 += a;
    }
    return (m * sum1) / (6.0 * sum2);
}

/* cpPolyShapeInit: hull the input, then SetVerts: plane i = { v0 = v[i], n = normalize(rperp(v[i]-v[i-1])) }. */
void ora_poly_init(ora_poly *poly, int count, const double *verts_xy)
{
    double hull[2 * 64];
    memset(poly, 0, sizeof(*poly)); /* cpcalloc: bb stays (0,0,0,0) until the first cache update */
    int n = ora_convex_hull(count, verts_xy, hull);
    if (n > ORA_MAX_VERTS) n = ORA_MAX_VERTS;
    poly->count = n;
    for (int i = 0; i < n; i++) {
        int im = (i - 1 + n) % n;
        ora_v2 a = V(hull[2 * im], hull[2 * im + 1]);
        ora_v2 b = V(hull[2 * i], hull[2 * i + 1]);
        poly->lv[i] = b;
        poly->ln[i] = vnormalize(vrperp(vsub(b, a)));
    }
}

/* cpPolyShapeCacheData */
void ora_poly_update(ora_poly *poly, ora_v2 p, ora_v2 rot)
{
    double l = INFINITY, r = -INFINITY, b = INFINITY, t = -INFINITY;
    for (int i = 0; i < poly->count; i++) {
        ora_v2 v = xf_point(p, rot, poly->lv[i]);
        ora_v2 n = xf_vect(rot, poly->ln[i]);
        poly->wv[i] = v;
        poly->wn[i] = n;
        l = fmin(l, v.x); r = fmax(r, v.x);
        b = fmin(b, v.y); t = fmax(t, v.y);
    }
    poly->bb_l = l; poly->bb_b = b; poly->bb_r = r; poly->bb_t = t;
}

static inline ora_v2 closest_point_on_segment(ora_v2 p, ora_v2 a, ora_v2 b)
{
    ora_v2 delta = vsub(a, b);
    double t = fclamp01(vdot(delta, vsub(p, b)) / vdot(delta, delta));
    return vadd(b, vmult(delta, t));
}

/* cpPolyShapePointQuery (r = 0): signed distance, negative inside. */
double ora_poly_point_query(const ora_poly *poly, ora_v2 p, ora_v2 *closest_out)
{
    int count = poly->count;
    ora_v2 v0 = poly->wv[count - 1];
    double min_dist = INFINITY;
    ora_v2 closest_point = V(0, 0);
    int outside = 0;
    for (int i = 0; i < count; i++) {
        ora_v2 v1 = poly->wv[i];
        outside = outside || (vdot(poly->wn[i], vsub(p, v1)) > 0.0);
        ora_v2 closest = closest_point_on_segment(p, v0, v1);
        double dist = vdist(p, closest);
        if (dist < min_dist) { min_dist = dist; closest_point = closest; }
        v0 = v1;
    }
    if (closest_out) *closest_out = closest_point;
    return outside ? min_dist : -min_dist;
}

/* CircleSegmentQuery (cpShape.h inline) */
static void circle_segment_query(ora_v2 center, double r1, ora_v2 a, ora_v2 b, double r2, ora_seg_info *info)
{
    ora_v2 da = vsub(a, center);
    ora_v2 db = vsub(b, center);
    double rsum = r1 + r2;
    double qa = vdot(da, da) - 2.0 * vdot(da, db) + vdot(db, db);
    double qb = vdot(da, db) - vdot(da, da);
    double det = qb * qb - qa * (vdot(da, da) - rsum * rsum);
    if (det >= 0.0) {
        double t = (-qb - sqrt(det)) / (qa);
        if (0.0 <= t && t <= 1.0) {
            ora_v2 n = vnormalize(vlerp(da, db, t));
            info->shape_hit = 1;
            info->point = vsub(vlerp(a, b, t), vmult(n, r2));
            info->normal = n;
            info->alpha = t;
        }
    }
}

/* cpShapeSegmentQuery -> cpPolyShapeSegmentQuery (poly radius 0), as pymunk Shape.segment_query /
 * Space.segment_query reach it (models.py:67, game.py:322-323). */
int ora_poly_segment_query(const ora_poly *poly, ora_v2 a, ora_v2 b, double r2, ora_seg_info *info)
{
    ora_seg_info blank = {0, b, {0, 0}, 1.0};
    *info = blank;

    ora_v2 nearest_pt;
    double nearest_d = ora_poly_point_query(poly, a, &nearest_pt);
    if (nearest_d <= r2) {
        info->shape_hit = 1;
        info->alpha = 0.0;
        info->normal = vnormalize(vsub(a, nearest_pt));
        return 1; /* info->point stays b: the FAR end (App. A.7) */
    }

    int count = poly->count;
    double r = 0.0;
    double rsum = r + r2;
    for (int i = 0; i < count; i++) {
        ora_v2 n = poly->wn[i];
        double an = vdot(a, n);
        double d = an - vdot(poly->wv[i], n) - rsum;
        if (d < 0.0) continue;
        double bn = vdot(b, n);
        double t = d / fmax(an - bn, DBL_MIN);
        if (t < 0.0 || 1.0 < t) continue;
        ora_v2 point = vlerp(a, b, t);
        double dt = vcross(n, point);
        double dt_min = vcross(n, poly->wv[(i - 1 + count) % count]);
        double dt_max = vcross(n, poly->wv[i]);
        if (dt_min <= dt && dt <= dt_max) {
            info->shape_hit = 1;
            info->point = vsub(vlerp(a, b, t), vmult(n, r2));
            info->normal = n;
            info->alpha = t;
        }
    }
    if (rsum > 0.0) {
        for (int i = 0; i < count; i++) {
            ora_seg_info ci = {0, b, {0, 0}, 1.0};
            circle_segment_query(poly->wv[i], r, a, b, r2, &ci);
            if (ci.alpha < info->alpha) *info = ci;
        }
    }
    return info->shape_hit;
}

/* Narrowphase predicates (App. A.7): Chipmunk reports a contact iff the GJK/EPA distance is <= 0
 * (poly-poly) or <= r (circle-poly), after the cpBBIntersects reject (closed intervals).  For convex
 * sets that is "closed sets intersect"; restated here as SAT over both polygons' edge normals and as the
 * point-query distance. */
static inline int bb_intersects(double al, double ab, double ar, double at, double bl, double bb, double br,
                                double bt)
{
    return (al <= br && bl <= ar && ab <= bt && bb <= at);
}

int ora_polys_collide(const ora_poly *a, const ora_poly *b) { return ora_polys_collide_v(a, b, 0); }
int ora_circle_poly_collide(ora_v2 c, double r, const ora_poly *poly) { return ora_circle_poly_collide_v(c, r, poly, 0); }

int ora_polys_collide_v(const ora_poly *a, const ora_poly *b, int strict)
{
    if (!bb_intersects(a->bb_l, a->bb_b, a->bb_r, a->bb_t, b->bb_l, b->bb_b, b->bb_r, b->bb_t)) return 0;
    for (int pass = 0; pass < 2; pass++) {
        const ora_poly *p = pass ? b : a, *q = pass ? a : b;
        for (int i = 0; i < p->count; i++) {
            ora_v2 n = p->wn[i];
            double off = vdot(n, p->wv[i]);
            double mn = INFINITY;
            for (int j = 0; j < q->count; j++) mn = fmin(mn, vdot(n, q->wv[j]));
            if (strict ? (mn >= off) : (mn > off)) return 0; /* separating axis (strict: touching separates too) */
        }
    }
    return 1;
}

int ora_circle_poly_collide_v(ora_v2 c, double r, const ora_poly *poly, int strict)
{
    if (!bb_intersects(c.x - r, c.y - r, c.x + r, c.y + r, poly->bb_l, poly->bb_b, poly->bb_r, poly->bb_t)) return 0;
    const double d = ora_poly_point_query(poly, c, NULL);
    return strict ? (d < r) : (d <= r);
}

/* cpCollide restated (ssg_dynamics.c) */
int ora_collide_poly_poly(const ora_poly *a, const ora_poly *b, int slot_a, int slot_b, ora_v2 *n, ora_v2 *p1, ora_v2 *p2,
                          uint32_t *hash, double *dist);

/* collide_ship's predicate for one player pair under the variant switches (+ the SAT / cpCollide agreement census) */
static int player_pair_collides(ora_world *w, const ora_poly *other, int other_slot)
{
    const int var = w->cfg.variant;
    const ora_poly *pl = &w->ship_shape;
    const int sat = ora_polys_collide_v(pl, other, var & ORA_VAR_TOUCH_STRICT);
    if (!(var & (ORA_VAR_CHECK_SAT | ORA_VAR_PLAYER_CPCOLLIDE))) return sat;
    if (!bb_intersects(pl->bb_l, pl->bb_b, pl->bb_r, pl->bb_t, other->bb_l, other->bb_b, other->bb_r, other->bb_t))
        return 0; /* queryReject: Chipmunk never runs cpCollide */
    ora_v2 n, p1[2], p2[2];
    uint32_t hash[2];
    double d_ab = 0.0, d_ba = 0.0;
    /* begin() fires when cpCollide pushes >= 1 contact; both a/b orders are evaluated (ORDER is unknown) */
    const int c_ab = ora_collide_poly_poly(pl, other, 7, other_slot, &n, p1, p2, hash, &d_ab) > 0;
    const int c_ba = ora_collide_poly_poly(other, pl, other_slot, 7, &n, p1, p2, hash, &d_ba) > 0;
    w->sat_checked++;
    if (c_ab != sat) w->sat_disagree_ab++;
    if (c_ba != sat) w->sat_disagree_ba++;
    if (fabs(d_ab) < 1e-9 || fabs(d_ba) < 1e-9) w->sat_near_zero++;
    return (var & ORA_VAR_PLAYER_CPCOLLIDE) ? c_ab : sat;
}

void ora_batch_counters(const ora_world *ws, int n, int64_t *out4)
{
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    for (int i = 0; i < n; i++) {
        out4[0] += ws[i].sat_checked; out4[1] += ws[i].sat_disagree_ab;
        out4[2] += ws[i].sat_disagree_ba; out4[3] += ws[i].sat_near_zero;
    }
}

/* ------------------------------------------------------------------------------------------------
 * World
 * ---------------------------------------------------------------------------------------------- */
static const double SHIP_TEMPLATE[5][2] = {{0, 0}, {0, 10}, {5, 15}, {10, 10}, {10, 0}}; /* models.py:6 */
#define DEFAULT_STATE_VAL (-1.0) /* ship_env.py:12 */
#define STEP_PENALTY (-0.01)     /* ship_env.py:13 */

void ora_default_config(ora_config *c)
{
    memset(c, 0, sizeof(*c));
    c->width = 600; c->height = 600;
    c->dt = 10 * 0.1;
    c->space_damping = 0.4;
    c->max_steps = 1000; c->history = 2;
    c->n_beams = 10; c->lidar_spread_deg = 90; c->lidar_dist = 100;
    c->n_goals = 5; c->goal_radius = 5;
    c->ship_w = 2; c->ship_h = 3; c->ship_mass = 5; c->force_y = 100;
    c->rudder_step = 5; c->rudder_max = 10;
    c->thrust_px0 = 0.0; c->thrust_py0 = 0.0;
    c->spawn_x = 600 / 2.0; c->spawn_y = 25;
}

int ora_world_sizeof(void) { return (int)sizeof(ora_world); }

void ora_world_init(ora_world *w, const ora_config *cfg)
{
    memset(w, 0, sizeof(*w));
    w->cfg = *cfg;
    w->n_states = 2 + 1 + 1 + 2 + cfg->n_beams; /* ship_env.py:43 */
}

static double nearest_goal(const ora_world *w, ora_v2 *gp)
{
    /* ShipGame.closest_goal game.py:333-349: strict '<', first listed wins ties */
    if (w->n_goals_alive == 0) return -1.0;
    int mi = 0;
    double md = vdist(w->goal_p[0], w->ship.p);
    for (int i = 1; i < w->n_goals_alive; i++) {
        double d = vdist(w->goal_p[i], w->ship.p);
        if (d < md) { md = d; mi = i; }
    }
    *gp = w->goal_p[mi];
    return md;
}

static void add_states(ora_world *w)
{
    /* ShipEnv.__add_states ship_env.py:79-113: frame = [x, y, rudder, angle, gx, gy, L...] */
    int F = w->n_states, total = F * w->cfg.history;
    double frame[6 + ORA_MAX_BEAMS];
    ora_v2 gp = V(-1.0, -1.0);
    nearest_goal(w, &gp);
    frame[0] = w->ship.p.x; frame[1] = w->ship.p.y;
    frame[2] = (double)w->rudder; frame[3] = w->ship.a;
    frame[4] = gp.x; frame[5] = gp.y;
    for (int i = 0; i < w->cfg.n_beams; i++) frame[6 + i] = w->lidar_vals[i];
    memmove(w->states, w->states + F, sizeof(double) * (size_t)(total - F)); /* deque(maxlen).extend */
    memcpy(w->states + (total - F), frame, sizeof(double) * (size_t)F);
}

void ora_world_reset(ora_world *w, const double *left_xy, const double *right_xy, const double *goals_xy,
                     double *obs_out)
{
    const ora_config *c = &w->cfg;
    /* gen_level game.py:60-71 -> PolyEnv models.py:153-196: static bodies at the origin, pm.Poly hulls */
    ora_poly_init(&w->bank[0], ORA_MAP_POLY_VERTS, left_xy);
    ora_poly_init(&w->bank[1], ORA_MAP_POLY_VERTS, right_xy);
    ora_poly_update(&w->bank[0], V(0, 0), V(1, 0)); /* cpSpaceAddShape caches the static shape once */
    ora_poly_update(&w->bank[1], V(0, 0), V(1, 0));
    /* gen_goal_path game.py:300-330 (positions produced by the caller's RNG) */
    w->n_goals_alive = c->n_goals;
    for (int i = 0; i < c->n_goals; i++) {
        w->goal_p[i] = V(goals_xy[2 * i], goals_xy[2 * i + 1]);
        w->goal_id[i] = i;
    }
    /* add_player_ship game.py:97-115, Ship.__init__ models.py:87-111 */
    double pts[10];
    for (int i = 0; i < 5; i++) { pts[2 * i] = SHIP_TEMPLATE[i][0] * c->ship_w; pts[2 * i + 1] = SHIP_TEMPLATE[i][1] * c->ship_h; }
    w->ship_moment = ora_moment_for_poly(c->ship_mass, 5, pts);
    memset(&w->ship, 0, sizeof(w->ship));
    w->ship.m_inv = 1.0 / c->ship_mass;
    w->ship.i_inv = 1.0 / w->ship_moment;
    w->ship.p = V(c->spawn_x, c->spawn_y);
    w->ship.a = 0.0;
    w->ship.rot = V(cos(0.0), sin(0.0));
    ora_poly_init(&w->ship_shape, 5, pts);
    w->rudder = 0;
    w->thrust_pt = V(c->thrust_px0, c->thrust_py0); /* shape.bb.center() before space.add, models.py:109 */
    for (int i = 0; i < c->n_beams; i++) w->lidar_vals[i] = -1.0; /* models.py:36 */
    ora_poly_update(&w->ship_shape, w->ship.p, w->ship.rot);      /* space.add(body, shape) game.py:113 */
    w->colliding = 0; w->goal_reached = 0;
    /* ShipEnv.reset ship_env.py:171-184 */
    w->reward = 0; w->cumulative_reward = 0; w->step_count = 0;
    int total = w->n_states * c->history;
    for (int i = 0; i < total; i++) w->states[i] = DEFAULT_STATE_VAL;
    if (c->n_traffic > 0) ora_dyn_reset(w); /* env.game.add_default_traffic() after reset (config 4) */
    add_states(w);
    if (obs_out) memcpy(obs_out, w->states, sizeof(double) * (size_t)total);
}

int ora_goal_x_range(const ora_world *w, double y, double *lo, double *hi)
{
    /* game.py:322-325: fat (r=10) rays from the mid-line to each edge; [0] of the hit list */
    const ora_config *c = &w->cfg;
    const double tolerance = 60.0;
    ora_v2 a = V(c->width / 2, y);
    ora_seg_info li, ri;
    int lh = 0, rh = 0;
    /* Space.segment_query visits every shape; with one bank per side at most one reports a hit. */
    for (int k = 0; k < 2 && !lh; k++) lh = ora_poly_segment_query(&w->bank[k], a, V(0, y), 10.0, &li);
    for (int k = 0; k < 2 && !rh; k++) rh = ora_poly_segment_query(&w->bank[k], a, V(c->width, y), 10.0, &ri);
    if (!lh || !rh) return 0; /* IndexError -> fallback branch game.py:328-330 */
    *lo = li.point.x + tolerance;
    *hi = ri.point.x - tolerance;
    return 1;
}

static void lidar_query(ora_world *w)
{
    /* LiDAR.query models.py:39-76 */
    const ora_config *c = &w->cfg;
    const double deg2rad = M_PI / 180.0; /* CPython math.radians */
    double angle_delta = (c->lidar_spread_deg / c->n_beams) * deg2rad;
    double angle_start = w->ship.a + (90 - c->lidar_spread_deg / 2) * deg2rad;
    const ora_poly *bb = &w->ship_shape;
    double cx = w->ship.p.x + (bb->bb_r - bb->bb_l) / 2;
    double cy = w->ship.p.y + (bb->bb_t - bb->bb_b) / 2;
    ora_v2 origin = V(cx, cy);
    for (int i = 0; i < c->n_beams; i++) {
        for (int s = 0; s < 2; s++) {
            double rotation = angle_start + (angle_delta * i);
            double x_end = cx + c->lidar_dist * cos(rotation);
            double y_end = cy + c->lidar_dist * sin(rotation);
            ora_seg_info info;
            if (ora_poly_segment_query(&w->bank[s], origin, V(x_end, y_end), 0.0, &info)) {
                double dx = info.point.x - origin.x, dy = info.point.y - origin.y;
                w->lidar_vals[i] = sqrt(dx * dx + dy * dy);
                break; /* first shape in list order wins, not the nearest (App. B-5) */
            }
        }
    }
}

/* ---- cpBody primitives, exported so that tests/golden/shims/pymunk (the stand-in the reference's own Python is executed
 * against when the control-flow goldens are made) runs on exactly the arithmetic the oracle world runs on ---- */
/* cpBodyUpdatePosition (v_bias / w_bias are zero for the player: no solver) */
void ora_body_update_position(ora_body *b, double dt)
{
    b->p = vadd(b->p, vmult(vadd(b->v, V(0, 0)), dt));
    b->a = b->a + (b->w + 0.0) * dt;
    b->rot = V(cos(b->a), sin(b->a));
}
/* cpBodyUpdateVelocity with gravity 0; `damping` = pow(space.damping, dt); forces are cleared afterwards (cpSpaceStep) */
void ora_body_update_velocity(ora_body *b, double damping, double dt)
{
    b->v = vadd(vmult(b->v, damping), vmult(vadd(V(0, 0), vmult(b->f, b->m_inv)), dt));
    b->w = b->w * damping + b->t * b->i_inv * dt;
    b->f = V(0, 0);
    b->t = 0.0;
}
/* cpBodyApplyForceAtLocalPoint: force and point in body coordinates, centre of gravity (0,0) */
void ora_body_apply_force_at_local_point(ora_body *b, ora_v2 force, ora_v2 point)
{
    ora_v2 fw = xf_vect(b->rot, force);
    ora_v2 pw = xf_point(b->p, b->rot, point);
    b->f = vadd(b->f, fw);
    ora_v2 r = vsub(pw, xf_point(b->p, b->rot, V(0, 0)));
    b->t += vcross(r, fw);
}
/* cpCircleShapeSegmentQuery (Space.segment_query also visits the goal circles already in the space) */
int ora_circle_segment_query(ora_v2 center, double r1, ora_v2 a, ora_v2 b, double r2, ora_seg_info *info)
{
    ora_seg_info blank = {0, b, {0, 0}, 1.0};
    *info = blank;
    circle_segment_query(center, r1, a, b, r2, info);
    return info->shape_hit;
}

static void space_step(ora_world *w)
{
    const ora_config *c = &w->cfg;
    double dt = c->dt;
    ora_body *b = &w->ship;
    /* (1) cpBodyUpdatePosition for every dynamic body.  Goal bodies have v = w = 0 and never move. */
    ora_body_update_position(b, dt);
    const int dyn = c->n_traffic > 0;
    int reached_mask = 0;
    if (dyn) ora_dyn_integrate(w); /* config 4: traffic + goal bodies move in the same pass (ssg_dynamics.c) */
    /* (2) cpShapeUpdateFunc + collide */
    ora_poly_update(&w->ship_shape, b->p, b->rot);
    for (int k = 0; k < 2; k++)
        if (player_pair_collides(w, &w->bank[k], k)) w->colliding = 1; /* collide_ship game.py:232-241 */
    for (int k = 0; dyn && k < c->n_traffic; k++) /* traffic ships are collision_type 1 too (models.py:100) */
        if (player_pair_collides(w, &w->dyn.tshape[k], ORA_SLOT_TRAFFIC0 + k)) w->colliding = 1;
    for (int g = 0; g < w->n_goals_alive;) {
        if (ora_circle_poly_collide_v(w->goal_p[g], c->goal_radius, &w->ship_shape, c->variant & ORA_VAR_TOUCH_STRICT)) {
            /* collide_goal game.py:243-257: goal dropped from the list, no physical response */
            w->goal_reached = 1;
            reached_mask |= 1 << w->goal_id[g];
            for (int j = g; j + 1 < w->n_goals_alive; j++) { w->goal_p[j] = w->goal_p[j + 1]; w->goal_id[j] = w->goal_id[j + 1]; }
            w->n_goals_alive--;
        } else {
            g++;
        }
    }
    /* (3) cpBodyUpdateVelocity, damping = pow(space.damping, dt), gravity = 0 */
    double damping = pow(c->space_damping, dt);
    ora_body_update_velocity(b, damping, dt);
    /* (4) impulse solver: not restated for the player (see file header); config 4 solves the other bodies */
    if (dyn) ora_dyn_collide_solve(w, reached_mask);
}

static void apply_action(ora_world *w, int action)
{
    /* ShipGame.handle_discrete_action game.py:140-153 */
    const ora_config *c = &w->cfg;
    ora_body *b = &w->ship;
    if (action == 0) {
        /* Ship.move_forward -> cpBodyApplyForceAtLocalPoint(force_vector*1, point_of_thrust) */
        ora_v2 force = V(0.0 * 1, c->force_y * 1);
        ora_body_apply_force_at_local_point(b, force, w->thrust_pt);
    } else if (action == 1 || action == 2) {
        /* Ship.rotate models.py:142-146 */
        w->rudder += (action == 1) ? -c->rudder_step : c->rudder_step;
        if (w->rudder < -c->rudder_max) w->rudder = -c->rudder_max;
        else if (w->rudder > c->rudder_max) w->rudder = c->rudder_max;
        w->thrust_pt.x = 0.0 - (double)w->rudder; /* center_of_gravity.x - rudder_angle */
    }
}

void ora_world_step(ora_world *w, int action, double *obs_out, double *reward, uint8_t *done)
{
    const ora_config *c = &w->cfg;
    apply_action(w, action);
    /* ShipGame.update game.py:185-195 */
    w->colliding = 0;
    w->goal_reached = 0;
    lidar_query(w);
    space_step(w);
    /* determine_reward ship_env.py:62-77 (the collision branch is overwritten by the chain below) */
    if (w->colliding) w->reward = -1.0;
    if (w->goal_reached) w->reward = 1.0;
    else if (w->ship.p.x < 0 || w->ship.p.x > c->width) w->reward = -1;
    else if (w->ship.p.y < 0 || w->ship.p.y > c->height) w->reward = -1;
    else w->reward = STEP_PENALTY;
    w->cumulative_reward += w->reward;
    add_states(w);
    w->step_count += 1;
    /* is_done ship_env.py:115-134 */
    int d = 0;
    if (w->colliding) d = 1;
    else if (w->n_goals_alive == 0) d = 1;
    else if (w->ship.p.x < 0 || w->ship.p.x > c->width) d = 1;
    else if (w->ship.p.y < 0 || w->ship.p.y > c->height) d = 1;
    else if (w->step_count >= c->max_steps) d = 1;
    if (obs_out) memcpy(obs_out, w->states, sizeof(double) * (size_t)(w->n_states * c->history));
    if (reward) *reward = w->reward;
    if (done) *done = (uint8_t)d;
}

/* ---- hooks for tests/golden/shims/pymunk when the reference's game holds traffic ships (config 4): the stand-in keeps a
 * shadow world, hands it the player's cpBody before each space.step and takes everything back afterwards, so that the
 * reference's own Python runs on the oracle's full cpSpaceStep (contact solver included) ---- */
void ora_world_set_ship(ora_world *w, const ora_body *b) { w->ship = *b; }
void ora_world_get_ship(const ora_world *w, ora_body *b) { *b = w->ship; }
/* ShipGame.update's `self.space.step(dt)` alone: cpSpaceStep with the begin-callbacks' effects (colliding, goal_reached, the
 * goal list), without the action, the lidar and the ShipEnv bookkeeping around it */
void ora_world_space_step(ora_world *w)
{
    w->colliding = 0;
    w->goal_reached = 0;
    space_step(w);
}

void ora_world_peek(const ora_world *w, double *o)
{
    int mask = 0;
    for (int i = 0; i < w->n_goals_alive; i++) mask |= 1 << w->goal_id[i];
    o[0] = w->ship.p.x; o[1] = w->ship.p.y; o[2] = w->ship.v.x; o[3] = w->ship.v.y;
    o[4] = w->ship.a; o[5] = w->ship.w; o[6] = w->rudder; o[7] = w->step_count;
    o[8] = w->n_goals_alive; o[9] = w->colliding; o[10] = w->goal_reached; o[11] = w->map_id;
    o[12] = w->cumulative_reward; o[13] = mask; o[14] = (double)w->episodes;
    o[15] = w->ship_shape.bb_l; o[16] = w->ship_shape.bb_b; o[17] = w->ship_shape.bb_r; o[18] = w->ship_shape.bb_t;
}

ora_world *ora_world_at(ora_world *ws, int i) { return ws + i; }

/* ------------------------------------------------------------------------------------------------
 * Batched driver: map bank + auto-reset (VecEnv semantics: a done env is reset in the same call and the
 * returned observation is the reset observation).  Env with map m resets onto map (m+1) mod n_maps.
 * ---------------------------------------------------------------------------------------------- */
static int next_map(const ora_bank *bank, int map_id)
{
    if (bank->ring > 0) {
        int base = map_id - map_id % bank->ring;
        return base + (map_id - base + 1) % bank->ring;
    }
    return (map_id + 1) % bank->n_maps;
}

static void reset_from_bank(ora_world *w, const ora_bank *bank, int map_id, double *obs)
{
    const double *poly = bank->polys + (size_t)map_id * 2 * ORA_MAP_POLY_VERTS * 2;
    const double *goals = bank->goals + (size_t)map_id * w->cfg.n_goals * 2;
    w->map_id = map_id;
    ora_world_reset(w, poly, poly + ORA_MAP_POLY_VERTS * 2, goals, obs);
}

void ora_batch_reset(ora_world *ws, int n, const ora_config *cfg, const ora_bank *bank, const int32_t *map_ids,
                     double *obs)
{
    int D = (6 + cfg->n_beams) * cfg->history;
    for (int e = 0; e < n; e++) {
        ora_world_init(&ws[e], cfg);
        reset_from_bank(&ws[e], bank, map_ids[e], obs ? obs + (size_t)e * D : NULL);
    }
}

void ora_batch_step(ora_world *ws, int n, const ora_bank *bank, const int32_t *actions, double *obs, double *reward,
                    uint8_t *done, int auto_reset, int n_threads)
{
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int e = 0; e < n; e++) {
        ora_world *w = &ws[e];
        int D = w->n_states * w->cfg.history;
        uint8_t d;
        ora_world_step(w, actions[e], obs + (size_t)e * D, &reward[e], &d);
        done[e] = d;
        if (d && auto_reset) {
            w->episodes++;
            reset_from_bank(w, bank, next_map(bank, w->map_id), obs + (size_t)e * D);
        }
    }
}

/* The auto-reset half of ora_batch_step on its own: after a step taken with auto_reset = 0 (so that the step's
 * colliding / goal_reached attributes can still be read), move every done env to its next bank record exactly as the
 * fused call does; the reset observation overwrites the env's row. */
void ora_batch_auto_reset(ora_world *ws, int n, const ora_bank *bank, const uint8_t *done, double *obs)
{
    for (int e = 0; e < n; e++) {
        ora_world *w = &ws[e];
        int D = w->n_states * w->cfg.history;
        if (!done[e]) continue;
        w->episodes++;
        reset_from_bank(w, bank, next_map(bank, w->map_id), obs + (size_t)e * D);
    }
}

/* Philox4x32-10 (Salmon et al. 2011), counter = (env_lo, env_hi, step_lo, step_hi), key = seed. */
static inline void philox_round(uint32_t c[4], const uint32_t k[2])
{
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int r = 0; r < 10; r++) {
        if (r) { k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u; }
        philox_round(c, k);
    }
    memcpy(out, c, sizeof(c));
}

int32_t ora_action(uint64_t seed, uint64_t step, uint64_t env_id)
{
    uint32_t ctr[4] = {(uint32_t)env_id, (uint32_t)(env_id >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t out[4];
    ora_philox4x32_10(ctr, key, out);
    return (int32_t)(((uint64_t)out[0] * 3u) >> 32); /* uniform on {0,1,2} = Discrete(3), ship_env.py:19 */
}

void ora_fill_actions(uint64_t seed, uint64_t step0, int K, int64_t env_base, int n, int32_t *out)
{
    for (int k = 0; k < K; k++)
        for (int e = 0; e < n; e++) out[(size_t)k * n + e] = ora_action(seed, step0 + (uint64_t)k, (uint64_t)(env_base + e));
}

int ora_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int64_t ora_rollout(ora_world *ws, int n, const ora_bank *bank, uint64_t seed, int64_t env_base, int K, int n_threads,
                    double *obs, double *reward, uint8_t *done)
{
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int e = 0; e < n; e++) {
        ora_world *w = &ws[e];
        int D = w->n_states * w->cfg.history;
        for (int k = 0; k < K; k++) {
            uint8_t d;
            int32_t a = ora_action(seed, (uint64_t)k, (uint64_t)(env_base + e));
            ora_world_step(w, a, obs + (size_t)e * D, &reward[e], &d);
            done[e] = d;
            if (d) {
                w->episodes++;
                reset_from_bank(w, bank, next_map(bank, w->map_id), obs + (size_t)e * D);
            }
        }
    }
    return (int64_t)n * K;
}
