// shipsim_dynamics.hip — config 4 (BASELINE configs[3]: 4 ships per env): the bodies other than the player.
//
// Replaces, for N envs at once, what `space.step(dt)` (game.py:194 -> Chipmunk2D cpSpaceStep) does to the three
// traffic ships of ShipGame.add_default_traffic (game.py:279-286, add_ship game.py:117-131, Ship.__init__
// models.py:87-111) and to the five goal bodies (add_goal game.py:77-95: mass-1 dynamic circles), including the
// contact solver between them and the river banks:
//   cpBodyUpdatePosition (v_bias / w_bias), cpPolyShapeCacheData, cpCollide (GJK + EPA closest points, support-edge
//   clipping, circle cases), cpArbiterUpdate / PreStep / ApplyCachedImpulse / ApplyImpulse (10 iterations),
//   cpSpaceArbiterSetFilter (collision persistence 3), cpBodyUpdateVelocity.
// and the player's `collide_ship` begin-callback against traffic (collision_type 1, models.py:100; game.py:232-241).
//
// One lane per env (a 64-lane workgroup = one wave): the work per env is a short, branchy chain (at most a
// handful of touching pairs), and envs diverge, so this kernel is latency/occupancy bound rather than HBM bound;
// it keeps its own launch so that the hot step kernel's register budget is untouched.  Per step it runs BEFORE the
// step kernel, which then reads this step's goal positions and the traffic-contact bit from the dyn columns
// (DevCfg::dyn_*).  State lives in struct-of-arrays columns like the player's; arbiter records (accumulated
// impulses, contact hashes, state) are only touched for pairs whose bit is set in the env's 64-bit live mask.
//
// The canonical pair order, the cold GJK start and the unsolved player arbiters are the named assumptions of the
// oracle (oracle/ssg_dynamics.c header); this file follows the same ones.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

#include "shipsim_internal.h"

namespace ssg {
namespace {

constexpr int kIter = 10;          // cpSpace iterations
constexpr int kPersist = 3;        // collisionPersistence
constexpr int kMaxGjk = 30, kMaxEpa = 30;
constexpr int kMaxActive = 8;      // arbiters on one env's solver list (5 circles + 3 ships never reach this)
enum { ST_NONE = 0, ST_FIRST = 1, ST_NORMAL = 2, ST_IGNORE = 3, ST_CACHED = 4 };
// shape slots: 0,1 banks | 2..7 goals | 8..10 traffic ships
constexpr int kSlotGoal0 = 2, kSlotTraffic0 = 2 + SSG_MAX_GOALS, kSlots = kSlotTraffic0 + SSG_N_TRAFFIC;

struct V2 { double x, y; };
__device__ __forceinline__ V2 mk(double x, double y) { V2 r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ V2 operator+(V2 a, V2 b) { return mk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ V2 operator-(V2 a, V2 b) { return mk(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ V2 operator*(V2 a, double s) { return mk(a.x * s, a.y * s); }
__device__ __forceinline__ V2 neg(V2 a) { return mk(-a.x, -a.y); }
__device__ __forceinline__ double dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ double cross(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ V2 perp(V2 a) { return mk(-a.y, a.x); }
__device__ __forceinline__ V2 rperp(V2 a) { return mk(a.y, -a.x); }
__device__ __forceinline__ double lensq(V2 a) { return dot(a, a); }
__device__ __forceinline__ double len(V2 a) { return sqrt(dot(a, a)); }
__device__ __forceinline__ V2 lerp(V2 a, V2 b, double t) { return a * (1.0 - t) + b * t; }
__device__ __forceinline__ V2 normalize(V2 a) { return a * (1.0 / (len(a) + DBL_MIN)); }
__device__ __forceinline__ V2 rotate(V2 a, V2 b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// cpfmin / cpfmax / cpfclamp are ternaries in chipmunk_types.h
__device__ __forceinline__ double cmin(double a, double b) { return (a < b) ? a : b; }
__device__ __forceinline__ double cmax(double a, double b) { return (a > b) ? a : b; }
__device__ __forceinline__ double cclamp(double f, double lo, double hi) { return cmin(cmax(f, lo), hi); }
__device__ __forceinline__ double cclamp01(double f) { return cmax(0.0, cmin(f, 1.0)); }

struct Poly {
    int n;
    double l, b, r, t;
    V2 v[SSG_MAX_HULL], nr[SSG_MAX_HULL];
};
struct Ref {
    const Poly *poly; // nullptr = circle
    V2 c;
    double rad;
    double l, b, r, t;
    unsigned hashid;
};
struct Body {
    V2 p, v, vb;
    double a, w, wb, m_inv, i_inv;
};
struct Mink { V2 a, b, ab; };
struct Closest { V2 a, b, n; double d; };
struct Info { int count; V2 n; V2 p1[2], p2[2]; unsigned hash[2]; };
struct Active {
    int pid, a, b, count, state;
    V2 n;
    double u;
    V2 r1[2], r2[2];
    double nMass[2], tMass[2], bias[2], bounce[2], jBias[2], jn[2], jt[2];
    unsigned hash[2];
};

__device__ __forceinline__ Ref ref_poly(const Poly *p, unsigned hashid)
{
    Ref s;
    s.poly = p; s.c = mk(0, 0); s.rad = 0.0; s.l = p->l; s.b = p->b; s.r = p->r; s.t = p->t; s.hashid = hashid;
    return s;
}
__device__ __forceinline__ Ref ref_circle(V2 c, double rad, unsigned hashid)
{
    Ref s;
    s.poly = nullptr; s.c = c; s.rad = rad; s.hashid = hashid;
    s.l = c.x - rad; s.b = c.y - rad; s.r = c.x + rad; s.t = c.y + rad; // cpCircleShapeCacheData
    return s;
}
__device__ __forceinline__ bool bb_hit(const Ref &a, const Ref &b)
{
    return (a.l <= b.r) & (b.l <= a.r) & (a.b <= b.t) & (b.b <= a.t);
}
__device__ __forceinline__ V2 bb_center(const Ref &s) { return lerp(mk(s.l, s.b), mk(s.r, s.t), 0.5); }

__device__ int support_index(const Poly *p, V2 n)
{
    double mx = -INFINITY;
    int index = 0;
    for (int i = 0; i < p->n; ++i) {
        const double d = dot(p->v[i], n);
        if (d > mx) { mx = d; index = i; }
    }
    return index;
}
__device__ __forceinline__ V2 support_point(const Ref &s, V2 n)
{
    if (!s.poly) return s.c;
    return s.poly->v[support_index(s.poly, n)];
}
__device__ __forceinline__ Mink support(const Ref &s1, const Ref &s2, V2 n)
{
    Mink m;
    m.a = support_point(s1, neg(n));
    m.b = support_point(s2, n);
    m.ab = m.b - m.a;
    return m;
}
__device__ __forceinline__ double closest_t(V2 a, V2 b)
{
    const V2 delta = b - a;
    return -cclamp(dot(delta, a + b) / lensq(delta), -1.0, 1.0);
}
__device__ __forceinline__ V2 lerp_t(V2 a, V2 b, double t)
{
    const double ht = 0.5 * t;
    return a * (0.5 - ht) + b * (0.5 + ht);
}
__device__ __forceinline__ double closest_dist(V2 v0, V2 v1) { return lensq(lerp_t(v0, v1, closest_t(v0, v1))); }

__device__ Closest closest_new(const Mink &v0, const Mink &v1)
{
    const double t = closest_t(v0.ab, v1.ab);
    const V2 p = lerp_t(v0.ab, v1.ab, t);
    Closest r;
    r.a = lerp_t(v0.a, v1.a, t);
    r.b = lerp_t(v0.b, v1.b, t);
    const V2 delta = v1.ab - v0.ab;
    const V2 n = normalize(rperp(delta));
    const double d = dot(n, p);
    if (d <= 0.0 || (-1.0 < t && t < 1.0)) {
        r.n = n; r.d = d;
    } else {
        const double d2 = len(p);
        r.n = p * (1.0 / (d2 + DBL_MIN));
        r.d = d2;
    }
    return r;
}

__device__ __attribute__((noinline)) Closest epa(const Ref &s1, const Ref &s2, const Mink &v0, const Mink &v1, const Mink &v2)
{
    Mink hull[kMaxEpa + 4], hull2[kMaxEpa + 4];
    int count = 3;
    hull[0] = v0; hull[1] = v1; hull[2] = v2;
    for (int iteration = 1;; ++iteration) {
        int mini = 0;
        double min_dist = INFINITY;
        for (int j = 0, i = count - 1; j < count; i = j, ++j) {
            const double d = closest_dist(hull[i].ab, hull[j].ab);
            if (d < min_dist) { min_dist = d; mini = i; }
        }
        const Mink e0 = hull[mini], e1 = hull[(mini + 1) % count];
        const Mink p = support(s1, s2, perp(e1.ab - e0.ab));
        const double area2x = cross(e1.ab - e0.ab, (p.ab - e0.ab) + (p.ab - e1.ab));
        if (area2x > 0.0 && iteration < kMaxEpa) {
            int count2 = 1;
            hull2[0] = p;
            for (int i = 0; i < count; ++i) {
                const int index = (mini + 1 + i) % count;
                const V2 h0 = hull2[count2 - 1].ab;
                const V2 h1 = hull[index].ab;
                const V2 h2 = (i + 1 < count) ? hull[(index + 1) % count].ab : p.ab;
                if (cross(h2 - h0, h1 - h0) > 0.0) hull2[count2++] = hull[index];
            }
            for (int i = 0; i < count2; ++i) hull[i] = hull2[i];
            count = count2;
        } else {
            return closest_new(e0, e1);
        }
    }
}

__device__ __attribute__((noinline)) Closest gjk(const Ref &s1, const Ref &s2)
{
    const V2 axis = perp(bb_center(s1) - bb_center(s2)); // cold start (no cached collision id)
    Mink v0 = support(s1, s2, axis);
    Mink v1 = support(s1, s2, neg(axis));
    int iteration = 1;
    for (;;) {
        if (iteration > kMaxGjk) return closest_new(v0, v1);
        const V2 delta = v1.ab - v0.ab;
        if (cross(delta, v0.ab + v1.ab) > 0.0) {
            const Mink tmp = v0; v0 = v1; v1 = tmp; // origin behind the axis: flip, same iteration
            continue;
        }
        const double t = closest_t(v0.ab, v1.ab);
        const V2 n = (-1.0 < t && t < 1.0) ? perp(delta) : neg(lerp_t(v0.ab, v1.ab, t));
        const Mink p = support(s1, s2, n);
        if (cross(v1.ab - p.ab, v1.ab + p.ab) > 0.0 && cross(v0.ab - p.ab, v0.ab + p.ab) < 0.0)
            return epa(s1, s2, v0, p, v1);
        if (dot(p.ab, n) <= cmax(dot(v0.ab, n), dot(v1.ab, n))) return closest_new(v0, v1);
        if (closest_dist(v0.ab, p.ab) < closest_dist(p.ab, v1.ab)) v1 = p; else v0 = p;
        ++iteration;
    }
}

// Contact hashes only ever get compared for equality (cpArbiterUpdate): edge point = slot*16 + vertex + 1,
// contact = (hash1 << 8) | hash2, collision free and never 0.
__device__ __forceinline__ unsigned edge_hash(unsigned hashid, int i) { return hashid * 16u + (unsigned)i + 1u; }

struct Edge { V2 ap, bp; unsigned ah, bh; V2 n; };
__device__ Edge support_edge(const Ref &s, V2 n)
{
    const Poly *p = s.poly;
    const int count = p->n;
    const int i1 = support_index(p, n);
    const int i0 = (i1 - 1 + count) % count;
    const int i2 = (i1 + 1) % count;
    Edge e;
    if (dot(n, p->nr[i1]) > dot(n, p->nr[i2])) {
        e.ap = p->v[i0]; e.ah = edge_hash(s.hashid, i0);
        e.bp = p->v[i1]; e.bh = edge_hash(s.hashid, i1);
        e.n = p->nr[i1];
    } else {
        e.ap = p->v[i1]; e.ah = edge_hash(s.hashid, i1);
        e.bp = p->v[i2]; e.bh = edge_hash(s.hashid, i2);
        e.n = p->nr[i2];
    }
    return e;
}

__device__ void contact_points(const Edge &e1, const Edge &e2, const Closest &points, Info &info)
{
    const double mindist = 0.0 + 0.0;
    if (points.d <= mindist) {
        const V2 n = info.n = points.n;
        const double d_e1_a = cross(e1.ap, n), d_e1_b = cross(e1.bp, n);
        const double d_e2_a = cross(e2.ap, n), d_e2_b = cross(e2.bp, n);
        const double e1_denom = 1.0 / (d_e1_b - d_e1_a + DBL_MIN);
        const double e2_denom = 1.0 / (d_e2_b - d_e2_a + DBL_MIN);
        {
            const V2 p1 = n * 0.0 + lerp(e1.ap, e1.bp, cclamp01((d_e2_b - d_e1_a) * e1_denom));
            const V2 p2 = n * -0.0 + lerp(e2.ap, e2.bp, cclamp01((d_e1_a - d_e2_a) * e2_denom));
            const double dist = dot(p2 - p1, n);
            if (dist <= 0.0) {
                info.p1[info.count] = p1; info.p2[info.count] = p2;
                info.hash[info.count] = e1.ah << 8 | e2.bh;
                info.count++;
            }
        }
        {
            const V2 p1 = n * 0.0 + lerp(e1.ap, e1.bp, cclamp01((d_e2_a - d_e1_a) * e1_denom));
            const V2 p2 = n * -0.0 + lerp(e2.ap, e2.bp, cclamp01((d_e1_b - d_e2_a) * e2_denom));
            const double dist = dot(p2 - p1, n);
            if (dist <= 0.0) {
                info.p1[info.count] = p1; info.p2[info.count] = p2;
                info.hash[info.count] = e1.bh << 8 | e2.ah;
                info.count++;
            }
        }
    }
}

__device__ void collide(const Ref &a, const Ref &b, Info &info)
{
    info.count = 0;
    info.n = mk(0, 0);
    if (!a.poly && !b.poly) { // CircleToCircle
        const double mindist = a.rad + b.rad;
        const V2 delta = b.c - a.c;
        const double distsq = lensq(delta);
        if (distsq < mindist * mindist) {
            const double dist = sqrt(distsq);
            const V2 n = info.n = (dist != 0.0) ? delta * (1.0 / dist) : mk(1.0, 0.0);
            info.p1[0] = a.c + n * a.rad;
            info.p2[0] = b.c + n * -b.rad;
            info.hash[0] = 0u;
            info.count = 1;
        }
    } else if (!a.poly) { // CircleToPoly
        const Closest points = gjk(a, b);
        const double mindist = a.rad + 0.0;
        if (points.d <= mindist) {
            const V2 n = info.n = points.n;
            info.p1[0] = points.a + n * a.rad;
            info.p2[0] = points.b + n * -0.0;
            info.hash[0] = 0u;
            info.count = 1;
        }
    } else { // PolyToPoly
        const Closest points = gjk(a, b);
        if (points.d - 0.0 - 0.0 <= 0.0) contact_points(support_edge(a, points.n), support_edge(b, neg(points.n)), points, info);
    }
}

// SAT over both polygons' edge normals after the cpBBIntersects reject: "touching counts" (collide_ship's begin)
__device__ bool polys_touch(const Poly &a, const Poly &b)
{
    if (!((a.l <= b.r) & (b.l <= a.r) & (a.b <= b.t) & (b.b <= a.t))) return false;
    for (int pass = 0; pass < 2; ++pass) {
        const Poly &p = pass ? b : a, &q = pass ? a : b;
        for (int i = 0; i < p.n; ++i) {
            const V2 n = p.nr[i];
            const double off = dot(n, p.v[i]);
            double mn = INFINITY;
            for (int j = 0; j < q.n; ++j) mn = fmin(mn, dot(n, q.v[j]));
            if (mn > off) return false;
        }
    }
    return true;
}

// cpPolyShapeCacheData for a 5-vertex ship hull
__device__ void ship_world(Poly &out, const double *hull, const double *nrm, V2 p, double ca, double sa)
{
    out.n = SSG_SHIP_VERTS;
    double l = INFINITY, r = -INFINITY, b = INFINITY, t = -INFINITY;
    for (int i = 0; i < SSG_SHIP_VERTS; ++i) {
        const double hx = hull[2 * i], hy = hull[2 * i + 1], nx = nrm[2 * i], ny = nrm[2 * i + 1];
        const V2 v = mk(ca * hx + (-sa) * hy + p.x, sa * hx + ca * hy + p.y);
        out.v[i] = v;
        out.nr[i] = mk(ca * nx + (-sa) * ny, sa * nx + ca * ny);
        l = fmin(l, v.x); r = fmax(r, v.x); b = fmin(b, v.y); t = fmax(t, v.y);
    }
    out.l = l; out.b = b; out.r = r; out.t = t;
}

__device__ void load_bank(Poly &out, const double *rec, int s)
{
    const int n = (int)rec[SSG_MAP_OFF_COUNTS + s];
    out.n = n;
    out.l = rec[SSG_MAP_OFF_AABB + 4 * s + 0]; out.b = rec[SSG_MAP_OFF_AABB + 4 * s + 1];
    out.r = rec[SSG_MAP_OFF_AABB + 4 * s + 2]; out.t = rec[SSG_MAP_OFF_AABB + 4 * s + 3];
    const double *pl = rec + SSG_MAP_OFF_PLANES + s * SSG_MAX_HULL * SSG_PLANE_DOUBLES;
    for (int j = 0; j < n; ++j) {
        out.v[j] = mk(pl[SSG_PLANE_DOUBLES * j + 0], pl[SSG_PLANE_DOUBLES * j + 1]);
        out.nr[j] = mk(pl[SSG_PLANE_DOUBLES * j + 2], pl[SSG_PLANE_DOUBLES * j + 3]);
    }
}

// ---- arbiter pair ids (bits of the live mask, rows of the arbiter columns) ----
__device__ __forceinline__ int pid_tb(int k, int s) { return 2 * k + s; }                       // [0, 6)
__device__ __forceinline__ int pid_tt(int j, int k) { return 6 + j + k - 1; }                   // j < k: [6, 9)
__device__ __forceinline__ int pid_gb(int g, int s) { return 9 + 2 * g + s; }                   // [9, 21)
__device__ __forceinline__ int pid_gt(int g, int k) { return 21 + 3 * g + k; }                  // [21, 39)
__device__ __forceinline__ int pid_gg(int h, int g) { return 39 + g * (g - 1) / 2 + h; }        // h < g: [39, 54)

__device__ __forceinline__ double k_scalar_body(const Body &b, V2 r, V2 n)
{
    const double rcn = cross(r, n);
    return b.m_inv + b.i_inv * rcn * rcn;
}
__device__ __forceinline__ V2 relative_velocity(const Body &a, const Body &b, V2 r1, V2 r2)
{
    const V2 v1 = a.v + perp(r1) * a.w;
    const V2 v2 = b.v + perp(r2) * b.w;
    return v2 - v1;
}
__device__ __forceinline__ void apply_impulse(Body &b, V2 j, V2 r)
{
    b.v = b.v + j * b.m_inv;
    b.w += b.i_inv * cross(r, j);
}
__device__ __forceinline__ void apply_bias_impulse(Body &b, V2 j, V2 r)
{
    b.vb = b.vb + j * b.m_inv;
    b.wb += b.i_inv * cross(r, j);
}

struct DynCols {
    double *f64;
    uint32_t *u32;
    unsigned long long *live;
    uint8_t *flag;
    size_t np;
};

// (re)create the non-player bodies of one env: a fresh pm.Space() after ShipGame.reset + add_default_traffic
__device__ void dyn_init(const DevCfg &c, const DynCfg &d, const DynCols &col, int e, const double *rec)
{
    for (int k = 0; k < SSG_N_TRAFFIC; ++k) {
        double *t = col.f64 + (size_t)(DC_TRAFFIC + 9 * k) * col.np + e;
        t[0 * col.np] = d.tx[k]; t[1 * col.np] = d.ty[k];
        for (int f = 2; f < 9; ++f) t[(size_t)f * col.np] = 0.0;
    }
    for (int g = 0; g < c.n_goals; ++g) {
        double *q = col.f64 + (size_t)(DC_GOALS + DC_GOAL_COLS * g) * col.np + e;
        q[0 * col.np] = rec[SSG_MAP_OFF_GOALS + 2 * g];
        q[1 * col.np] = rec[SSG_MAP_OFF_GOALS + 2 * g + 1];
        for (int f = 2; f < DC_GOAL_COLS; ++f) q[(size_t)f * col.np] = 0.0;
    }
    col.live[e] = 0ull;
}

} // namespace

__global__ void dyn_reset_kernel(const DevCfg c, const DynCfg d, const uint8_t *__restrict__ mask)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.n_envs) return;
    if (mask && !mask[e]) return;
    DynCols col{c.dyn_f64, c.dyn_u32, c.dyn_live, c.dyn_flag, (size_t)c.n_pad};
    const int m = c.i32cols[(size_t)ICOL_MAP * col.np + e]; // written by reset_kernel just before (same stream)
    const double *rec = c.bank + (size_t)m * SSG_MAP_STRIDE;
    dyn_init(c, d, col, e, rec);
    col.f64[(size_t)(DC_PREV_GOAL + 0) * col.np + e] = rec[SSG_MAP_OFF_SPAWN_GOAL]; // the reset frame's goal
    col.f64[(size_t)(DC_PREV_GOAL + 1) * col.np + e] = rec[SSG_MAP_OFF_SPAWN_GOAL + 1];
    col.flag[e] = 0;
}

__global__ __launch_bounds__(64) void dyn_step_kernel(const DevCfg c, const DynCfg d)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.n_envs) return;
    DynCols col{c.dyn_f64, c.dyn_u32, c.dyn_live, c.dyn_flag, (size_t)c.n_pad};
    const size_t np = col.np;
    const double dt = c.dt;
    const int map_id = c.i32cols[(size_t)ICOL_MAP * np + e];
    const double *rec = c.bank + (size_t)map_id * SSG_MAP_STRIDE;
    if (col.flag[e] & 2) dyn_init(c, d, col, e, rec); // the step kernel auto-reset this env at the end of the last step
    const unsigned gmask = (unsigned)c.mask[e] & ((1u << c.n_goals) - 1u); // goals still in the space
    unsigned long long live = col.live[e];

    // deferred space.remove of goals the player reached last step (game.py:252): their cached arbiters go too
    for (int g = 0; g < c.n_goals; ++g) {
        if ((gmask >> g) & 1u) continue;
        for (int s = 0; s < 2; ++s) live &= ~(1ull << pid_gb(g, s));
        for (int k = 0; k < SSG_N_TRAFFIC; ++k) live &= ~(1ull << pid_gt(g, k));
        for (int h = 0; h < c.n_goals; ++h)
            if (h != g) live &= ~(1ull << (h < g ? pid_gg(h, g) : pid_gg(g, h)));
    }

    // ---- bodies -------------------------------------------------------------------------------------------
    Body bod[kSlots];
    for (int s = 0; s < kSlots; ++s) {
        Body &b = bod[s];
        b.p = mk(0, 0); b.v = mk(0, 0); b.vb = mk(0, 0); b.a = 0.0; b.w = 0.0; b.wb = 0.0; b.m_inv = 0.0; b.i_inv = 0.0;
    }
    for (int g = 0; g < c.n_goals; ++g) {
        if (!((gmask >> g) & 1u)) continue;
        const double *q = col.f64 + (size_t)(DC_GOALS + DC_GOAL_COLS * g) * np + e;
        Body &b = bod[kSlotGoal0 + g];
        b.p = mk(q[0 * np], q[1 * np]); b.v = mk(q[2 * np], q[3 * np]); b.vb = mk(q[4 * np], q[5 * np]);
        b.w = q[6 * np]; b.wb = q[7 * np];
        b.m_inv = d.goal_m_inv; b.i_inv = d.goal_i_inv;
    }
    for (int k = 0; k < SSG_N_TRAFFIC; ++k) {
        const double *t = col.f64 + (size_t)(DC_TRAFFIC + 9 * k) * np + e;
        Body &b = bod[kSlotTraffic0 + k];
        b.p = mk(t[0 * np], t[1 * np]); b.a = t[2 * np]; b.v = mk(t[3 * np], t[4 * np]); b.w = t[5 * np];
        b.vb = mk(t[6 * np], t[7 * np]); b.wb = t[8 * np];
        b.m_inv = d.t_m_inv; b.i_inv = d.t_i_inv[k];
    }

    // ---- (1) cpBodyUpdatePosition ---------------------------------------------------------------------------
    for (int s = kSlotGoal0; s < kSlots; ++s) {
        Body &b = bod[s];
        if (b.m_inv == 0.0) continue; // goal no longer in the space
        b.p = b.p + (b.v + b.vb) * dt;
        b.a = b.a + (b.w + b.wb) * dt;
        b.vb = mk(0, 0);
        b.wb = 0.0;
    }
    // the player's pose after its own cpBodyUpdatePosition (same expressions as the step kernel)
    Poly player;
    {
        const double x = c.f64cols[(size_t)COL_X * np + e], y = c.f64cols[(size_t)COL_Y * np + e];
        const double vx = c.f64cols[(size_t)COL_VX * np + e], vy = c.f64cols[(size_t)COL_VY * np + e];
        const double ang = c.f64cols[(size_t)COL_A * np + e], w = c.f64cols[(size_t)COL_W * np + e];
        const double nx = x + vx * dt, ny = y + vy * dt, na = ang + w * dt;
        double sa, ca;
        sincos(na, &sa, &ca);
        ship_world(player, c.hull, c.nrm, mk(nx, ny), ca, sa);
    }
    // ---- (2) shape caches -----------------------------------------------------------------------------------
    Poly ship[SSG_N_TRAFFIC], bank[2];
    bool hit = false;
    for (int k = 0; k < SSG_N_TRAFFIC; ++k) {
        const Body &b = bod[kSlotTraffic0 + k];
        double sa, ca;
        sincos(b.a, &sa, &ca);
        ship_world(ship[k], d.thull[k], d.tnrm[k], b.p, ca, sa);
        hit |= polys_touch(player, ship[k]); // collide_ship: player (type 0) x traffic (type 1)
    }
    bool bank_loaded[2] = {false, false};

    auto shape_of = [&](int slot) -> Ref {
        if (slot < kSlotGoal0) return ref_poly(&bank[slot], (unsigned)slot);
        if (slot < kSlotTraffic0) return ref_circle(bod[slot].p, c.goal_r, (unsigned)slot);
        return ref_poly(&ship[slot - kSlotTraffic0], (unsigned)slot);
    };
    auto friction_of = [&](int slot) -> double { return slot >= kSlotTraffic0 ? d.ship_friction : 0.0; };

    // ---- (3) collide, canonical order -----------------------------------------------------------------------
    Active act[kMaxActive];
    int n_act = 0;
    unsigned long long touched = 0ull;

    auto collide_pair = [&](int a, int b, int pid) {
        if (a < kSlotGoal0 || b < kSlotGoal0) { // a bank is involved: cheap reject on the record's AABB first
            const int s = (a < kSlotGoal0) ? a : b, o = (a < kSlotGoal0) ? b : a;
            const double bl = rec[SSG_MAP_OFF_AABB + 4 * s + 0], bb = rec[SSG_MAP_OFF_AABB + 4 * s + 1];
            const double br = rec[SSG_MAP_OFF_AABB + 4 * s + 2], bt = rec[SSG_MAP_OFF_AABB + 4 * s + 3];
            double ol, ob, orr, ot;
            if (o < kSlotTraffic0) { ol = bod[o].p.x - c.goal_r; ob = bod[o].p.y - c.goal_r; orr = bod[o].p.x + c.goal_r; ot = bod[o].p.y + c.goal_r; }
            else { const Poly &q = ship[o - kSlotTraffic0]; ol = q.l; ob = q.b; orr = q.r; ot = q.t; }
            if (!((ol <= br) & (bl <= orr) & (ob <= bt) & (bb <= ot))) return;
            if (!bank_loaded[s]) { load_bank(bank[s], rec, s); bank_loaded[s] = true; }
        }
        const Ref sa = shape_of(a), sb = shape_of(b);
        if (!bb_hit(sa, sb)) return; // queryReject
        Info info;
        collide(sa, sb, info);
        if (info.count == 0 || n_act >= kMaxActive) return;
        // cached arbiter of this pair, if any
        int state = ST_NONE, old_count = 0;
        unsigned old_hash[2] = {0u, 0u};
        double old_jn[2] = {0.0, 0.0}, old_jt[2] = {0.0, 0.0};
        if ((live >> pid) & 1ull) {
            const unsigned meta = col.u32[(size_t)(DU_META + pid) * np + e];
            state = meta & 7u;
            old_count = (meta >> 5) & 3u;
            if (pid < kPolyPairs) {
                const unsigned hh = col.u32[(size_t)(DU_HASH + pid) * np + e];
                old_hash[0] = hh & 0xFFFFu; old_hash[1] = hh >> 16;
            }
            const double *acc = col.f64 + (size_t)(DC_ARB + 4 * pid) * np + e;
            old_jn[0] = acc[0 * np]; old_jn[1] = acc[1 * np]; old_jt[0] = acc[2 * np]; old_jt[1] = acc[3 * np];
            if (state == ST_FIRST) state = ST_NORMAL; // it was on last step's solver list
        }
        if (state == ST_NONE) { state = ST_FIRST; old_count = 0; } // cpArbiterInit
        Active &A = act[n_act++];
        A.pid = pid; A.a = a; A.b = b; A.count = info.count; A.n = info.n;
        A.u = friction_of(a) * friction_of(b);
        for (int i = 0; i < info.count; ++i) { // cpArbiterUpdate
            A.r1[i] = info.p1[i] - bod[a].p;
            A.r2[i] = info.p2[i] - bod[b].p;
            A.hash[i] = info.hash[i];
            A.jn[i] = 0.0; A.jt[i] = 0.0;
            for (int j = 0; j < old_count; ++j)
                if (info.hash[i] == old_hash[j]) { A.jn[i] = old_jn[j]; A.jt[i] = old_jt[j]; }
        }
        if (state == ST_CACHED) state = ST_FIRST;
        A.state = state;
        touched |= 1ull << pid;
        live |= 1ull << pid;
    };

    for (int g = 0; g < c.n_goals; ++g) {
        if (!((gmask >> g) & 1u)) continue;
        for (int s = 0; s < 2; ++s) collide_pair(kSlotGoal0 + g, s, pid_gb(g, s));
        for (int h = 0; h < g; ++h)
            if ((gmask >> h) & 1u) collide_pair(kSlotGoal0 + h, kSlotGoal0 + g, pid_gg(h, g));
    }
    for (int k = 0; k < SSG_N_TRAFFIC; ++k) {
        for (int s = 0; s < 2; ++s) collide_pair(kSlotTraffic0 + k, s, pid_tb(k, s));
        for (int g = 0; g < c.n_goals; ++g)
            if ((gmask >> g) & 1u) collide_pair(kSlotGoal0 + g, kSlotTraffic0 + k, pid_gt(g, k));
        for (int j = 0; j < k; ++j) collide_pair(kSlotTraffic0 + j, kSlotTraffic0 + k, pid_tt(j, k));
    }

    // ---- cpSpaceArbiterSetFilter for the cached arbiters that were not touched this step -------------------------
    {
        unsigned long long rest = live & ~touched;
        while (rest) {
            const int pid = __ffsll((long long)rest) - 1;
            rest &= rest - 1ull;
            unsigned meta = col.u32[(size_t)(DU_META + pid) * np + e];
            unsigned state = meta & 7u, age = (meta >> 3) & 3u;
            if (state == ST_FIRST) state = ST_NORMAL;
            age += 1u;
            if (state != ST_CACHED) state = ST_CACHED; // ticks >= 1
            if (age >= (unsigned)kPersist) {
                live &= ~(1ull << pid);
            } else {
                meta = (meta & ~0x1Fu) | state | (age << 3);
                col.u32[(size_t)(DU_META + pid) * np + e] = meta;
            }
        }
    }

    // ---- cpArbiterPreStep ---------------------------------------------------------------------------------------
    for (int i = 0; i < n_act; ++i) {
        Active &A = act[i];
        const Body &a = bod[A.a], &b = bod[A.b];
        const V2 n = A.n;
        const V2 body_delta = b.p - a.p;
        for (int k = 0; k < A.count; ++k) {
            A.nMass[k] = 1.0 / (k_scalar_body(a, A.r1[k], n) + k_scalar_body(b, A.r2[k], n));
            A.tMass[k] = 1.0 / (k_scalar_body(a, A.r1[k], perp(n)) + k_scalar_body(b, A.r2[k], perp(n)));
            const double dist = dot((A.r2[k] - A.r1[k]) + body_delta, n);
            A.bias[k] = -d.bias_coef * cmin(0.0, dist + d.slop) / dt;
            A.jBias[k] = 0.0;
            A.bounce[k] = dot(relative_velocity(a, b, A.r1[k], A.r2[k]), n) * 0.0; // arb->e = 0 for every shape here
        }
    }
    // ---- (4) cpBodyUpdateVelocity (no forces on these bodies) -------------------------------------------------------
    for (int s = kSlotGoal0; s < kSlots; ++s) {
        Body &b = bod[s];
        if (b.m_inv == 0.0) continue;
        b.v = b.v * c.damp + (mk(0, 0) + mk(0, 0) * b.m_inv) * dt;
        b.w = b.w * c.damp + 0.0 * b.i_inv * dt;
    }
    // ---- (5) cached impulses (dt_coef = dt/prev_dt = 1; first contacts skip), then the solver ----------------------
    for (int i = 0; i < n_act; ++i) {
        Active &A = act[i];
        if (A.state == ST_FIRST) continue;
        Body &a = bod[A.a], &b = bod[A.b];
        for (int k = 0; k < A.count; ++k) {
            const V2 j = rotate(A.n, mk(A.jn[k], A.jt[k])) * 1.0;
            apply_impulse(a, neg(j), A.r1[k]);
            apply_impulse(b, j, A.r2[k]);
        }
    }
    for (int it = 0; it < kIter; ++it) {
        for (int i = 0; i < n_act; ++i) {
            Active &A = act[i];
            Body &a = bod[A.a], &b = bod[A.b];
            const V2 n = A.n;
            for (int k = 0; k < A.count; ++k) {
                const V2 r1 = A.r1[k], r2 = A.r2[k];
                const V2 vb1 = a.vb + perp(r1) * a.wb;
                const V2 vb2 = b.vb + perp(r2) * b.wb;
                const V2 vr = relative_velocity(a, b, r1, r2) + mk(0, 0);
                const double vbn = dot(vb2 - vb1, n);
                const double vrn = dot(vr, n);
                const double vrt = dot(vr, perp(n));
                const double jbn = (A.bias[k] - vbn) * A.nMass[k];
                const double jbnOld = A.jBias[k];
                A.jBias[k] = cmax(jbnOld + jbn, 0.0);
                const double jn = -(A.bounce[k] + vrn) * A.nMass[k];
                const double jnOld = A.jn[k];
                A.jn[k] = cmax(jnOld + jn, 0.0);
                const double jtMax = A.u * A.jn[k];
                const double jt = -vrt * A.tMass[k];
                const double jtOld = A.jt[k];
                A.jt[k] = cclamp(jtOld + jt, -jtMax, jtMax);
                const V2 jb = n * (A.jBias[k] - jbnOld);
                apply_bias_impulse(a, neg(jb), r1);
                apply_bias_impulse(b, jb, r2);
                const V2 j = rotate(n, mk(A.jn[k] - jnOld, A.jt[k] - jtOld));
                apply_impulse(a, neg(j), r1);
                apply_impulse(b, j, r2);
            }
        }
    }

    // ---- write back -----------------------------------------------------------------------------------------------
    for (int i = 0; i < n_act; ++i) {
        const Active &A = act[i];
        col.u32[(size_t)(DU_META + A.pid) * np + e] = (unsigned)A.state | (0u << 3) | ((unsigned)A.count << 5);
        if (A.pid < kPolyPairs)
            col.u32[(size_t)(DU_HASH + A.pid) * np + e] = (A.hash[0] & 0xFFFFu) | ((A.count > 1 ? A.hash[1] : 0u) << 16);
        double *acc = col.f64 + (size_t)(DC_ARB + 4 * A.pid) * np + e;
        acc[0 * np] = A.jn[0]; acc[1 * np] = A.count > 1 ? A.jn[1] : 0.0;
        acc[2 * np] = A.jt[0]; acc[3 * np] = A.count > 1 ? A.jt[1] : 0.0;
    }
    for (int g = 0; g < c.n_goals; ++g) {
        if (!((gmask >> g) & 1u)) continue;
        double *q = col.f64 + (size_t)(DC_GOALS + DC_GOAL_COLS * g) * np + e;
        const Body &b = bod[kSlotGoal0 + g];
        q[0 * np] = b.p.x; q[1 * np] = b.p.y; q[2 * np] = b.v.x; q[3 * np] = b.v.y; q[4 * np] = b.vb.x; q[5 * np] = b.vb.y;
        q[6 * np] = b.w; q[7 * np] = b.wb;
    }
    for (int k = 0; k < SSG_N_TRAFFIC; ++k) {
        double *t = col.f64 + (size_t)(DC_TRAFFIC + 9 * k) * np + e;
        const Body &b = bod[kSlotTraffic0 + k];
        t[0 * np] = b.p.x; t[1 * np] = b.p.y; t[2 * np] = b.a; t[3 * np] = b.v.x; t[4 * np] = b.v.y; t[5 * np] = b.w;
        t[6 * np] = b.vb.x; t[7 * np] = b.vb.y; t[8 * np] = b.wb;
    }
    col.live[e] = live;
    col.flag[e] = hit ? 1 : 0;
}

hipError_t launch_dyn_step(const DevCfg &c, const DynCfg &d, hipStream_t stream)
{
    const int block = 64;
    hipLaunchKernelGGL(dyn_step_kernel, dim3((unsigned)((c.n_envs + block - 1) / block)), dim3(block), 0, stream, c, d);
    return hipGetLastError();
}

hipError_t launch_dyn_reset(const DevCfg &c, const DynCfg &d, const uint8_t *mask, hipStream_t stream)
{
    const int block = 256;
    hipLaunchKernelGGL(dyn_reset_kernel, dim3((unsigned)((c.n_envs + block - 1) / block)), dim3(block), 0, stream, c, d, mask);
    return hipGetLastError();
}

} // namespace ssg
