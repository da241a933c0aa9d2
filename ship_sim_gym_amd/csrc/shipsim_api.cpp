// shipsim_api.cpp — the extern "C" boundary of libshipsim.so (include/shipsim.h) and the host-side geometry
// that pymunk's cffi layer provided to the reference at reset time (hulling, splitting planes, moments, the
// fat segment queries of gen_goal_path).  No torch types, no exceptions across the boundary.
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "shipsim.h"
#include "shipsim_internal.h"

struct ssg_handle {
    ssg_config cfg;
    ssg::DevCfg dev{};
    int n_pad = 0;
    size_t off_stats = 0, off_f64 = 0, off_i32 = 0, off_mask = 0, off_obs2 = 0, off_obsH = 0, nbytes = 0;
    size_t off_dyn_f64 = 0, off_dyn_live = 0, off_dyn_u32 = 0, off_dyn_flag = 0; // config 4 only
    size_t off_dyn_hash = 0, off_dyn_count = 0, off_dyn_row = 0, off_dyn_bucket = 0;
    size_t off_dyn_memo = 0, off_dyn_memo_stats = 0, off_dyn_npm = 0, off_dyn_qmap = 0;
    bool time_kernels = false;        // ssg_debug_kernel_times: HIP events around the two launches of every config-4 step
    double t_dyn_ms = 0.0, t_step_ms = 0.0;
    unsigned long long t_steps = 0;
    unsigned long long n_classify = 0; // launches of the classify pass (the queue of the full step rebuilt from the per-env flags)
    int masked_resets_since_step = 0; // masked ssg_reset calls that joined the live dyn queue since the last step (at most one may) // the memo of the full dyn step (shipsim_internal.h, kMemoEntries)
    unsigned long long dyn_seq = 0;   // launches of the full step so far
    unsigned memo_gen = 1;            // generation of the memo's entries (1..255)
    bool memo_clear_pending = false;  // the generation counter wrapped: zero the table before the next launch
    int memo_steps = 0;               // launches since the generation began (a generation ends after kMemoGenSteps of them)
    bool dyn_queue_valid = false; // the step kernel's last launch left the next step's dyn queue (nothing host-side touched the envs since)
    ssg::DynCfg dyn{};
    void *state = nullptr;
    const double *bank = nullptr;
    int n_maps = 0;
    int block = 256;
    bool lds = false;
    size_t lds_bytes = 0;
    bool prepared = false;
    bool zeroed = false;       // the bound blob is known to have been zeroed by us (ssg_init_state / first full reset)
    // map_ring mode: the ring's source (fixed by ssg_refill_worlds) and how many more episodes an env may start before the
    // rings must be refilled (an env consumes at most one world per step or reset)
    bool ring_ready = false;
    uint64_t ring_seed = 0;
    double ring_width_frac = 0.5;
    int ring_credit = 0;
    size_t off_ring_queue = 0, off_ring_count = 0;
    bool remap_pending = false; // the bank shrank: ICOL_MAP must be taken modulo n_maps before the next kernel reads it
    // ssg_step_host / ssg_wait_host: one completion event per host block slot, created on first use
    static constexpr int kHostSlots = 8;
    hipEvent_t host_ev[kHostSlots] = {};
    bool host_ev_made[kHostSlots] = {};
    std::string err;
};

namespace {

thread_local std::string g_err; // errors raised before a handle exists

int fail(ssg_handle *h, int code, const std::string &msg)
{
    if (h) h->err = msg;
    else g_err = msg;
    return code;
}

// ---------------------------------------------------------------------------------------------------------
// host geometry
// ---------------------------------------------------------------------------------------------------------
struct P2 { double x, y; };
inline double cross3(const P2 &o, const P2 &a, const P2 &b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); }

// Strict convex hull, counter-clockwise, first vertex = lexicographic (x, then y) minimum: the vertex order
// cpConvexHull (QuickHull, tol 0) hands to cpPolyShape.  Implemented as Andrew's monotone chain.
std::vector<P2> convex_hull(std::vector<P2> pts)
{
    std::sort(pts.begin(), pts.end(), [](const P2 &a, const P2 &b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
    pts.erase(std::unique(pts.begin(), pts.end(), [](const P2 &a, const P2 &b) { return a.x == b.x && a.y == b.y; }),
              pts.end());
    const int n = (int)pts.size();
    if (n <= 2) return pts;
    std::vector<P2> h(2 * n);
    int k = 0;
    for (int i = 0; i < n; ++i) { // lower chain
        while (k >= 2 && cross3(h[k - 2], h[k - 1], pts[i]) <= 0.0) --k;
        h[k++] = pts[i];
    }
    for (int i = n - 2, t = k + 1; i >= 0; --i) { // upper chain
        while (k >= t && cross3(h[k - 2], h[k - 1], pts[i]) <= 0.0) --k;
        h[k++] = pts[i];
    }
    h.resize(k - 1);
    return h;
}

struct Plane { double v0x, v0y, nx, ny, v0n, dtmin, dtmax; };

// cpPolyShape SetVerts: plane i = { v0 = v[i], n = normalize(rperp(v[i] - v[i-1])) }, rperp(x,y) = (y,-x),
// normalize(v) = v * (1/(|v| + DBL_MIN)); plus the constants cpPolyShapeSegmentQuery derives per plane.
std::vector<Plane> planes_of(const std::vector<P2> &v)
{
    const int n = (int)v.size();
    std::vector<Plane> pl(n);
    for (int i = 0; i < n; ++i) {
        const P2 &a = v[(i - 1 + n) % n], &b = v[i];
        const double ex = b.x - a.x, ey = b.y - a.y;
        const double rx = ey, ry = -ex;
        const double inv = 1.0 / (std::sqrt(rx * rx + ry * ry) + DBL_MIN);
        Plane p;
        p.v0x = b.x; p.v0y = b.y;
        p.nx = rx * inv; p.ny = ry * inv;
        pl[i] = p;
    }
    for (int i = 0; i < n; ++i) {
        Plane &p = pl[i];
        const P2 &prev = v[(i - 1 + n) % n];
        p.v0n = p.v0x * p.nx + p.v0y * p.ny;          // cpvdot(v0, n)
        p.dtmin = p.nx * prev.y - p.ny * prev.x;      // cpvcross(n, v[i-1])
        p.dtmax = p.nx * p.v0y - p.ny * p.v0x;        // cpvcross(n, v[i])
    }
    return pl;
}

struct HullView {
    int n;
    const double *pl; // n planes of SSG_PLANE_DOUBLES doubles
    double v0x(int i) const { return pl[SSG_PLANE_DOUBLES * i + 0]; }
    double v0y(int i) const { return pl[SSG_PLANE_DOUBLES * i + 1]; }
    double nx(int i) const { return pl[SSG_PLANE_DOUBLES * i + 2]; }
    double ny(int i) const { return pl[SSG_PLANE_DOUBLES * i + 3]; }
    double v0n(int i) const { return pl[SSG_PLANE_DOUBLES * i + 4]; }
    // cpPolyShapeSegmentQuery's edge extents: cpvcross(n, v[i-1]) and cpvcross(n, v[i])
    double dtmin(int i) const { const int p = (i - 1 + n) % n; return nx(i) * v0y(p) - ny(i) * v0x(p); }
    double dtmax(int i) const { return nx(i) * v0y(i) - ny(i) * v0x(i); }
};

inline double clamp01(double f) { return std::max(0.0, std::min(f, 1.0)); }

// cpPolyShapePointQuery (radius 0): signed distance of p to the hull and the closest boundary point.
double point_query(const HullView &h, double px, double py, double *cx, double *cy)
{
    double v0x = h.v0x(h.n - 1), v0y = h.v0y(h.n - 1);
    double best = INFINITY, bx = 0, by = 0;
    bool outside = false;
    for (int i = 0; i < h.n; ++i) {
        const double v1x = h.v0x(i), v1y = h.v0y(i);
        outside = outside || ((h.nx(i) * (px - v1x) + h.ny(i) * (py - v1y)) > 0.0);
        const double dx = v0x - v1x, dy = v0y - v1y;
        const double t = clamp01((dx * (px - v1x) + dy * (py - v1y)) / (dx * dx + dy * dy));
        const double qx = v1x + dx * t, qy = v1y + dy * t;
        const double ex = px - qx, ey = py - qy;
        const double d = std::sqrt(ex * ex + ey * ey);
        if (d < best) { best = d; bx = qx; by = qy; }
        v0x = v1x; v0y = v1y;
    }
    if (cx) *cx = bx;
    if (cy) *cy = by;
    return outside ? best : -best;
}

struct SegHit { bool hit; double px, py, alpha; };

// cpShapeSegmentQuery -> cpPolyShapeSegmentQuery + CircleSegmentQuery for the bevelled corners (poly radius 0,
// query radius r2), as Space.segment_query / Shape.segment_query reach it.
SegHit segment_query(const HullView &h, double ax, double ay, double bx, double by, double r2)
{
    SegHit out{false, bx, by, 1.0};
    if (point_query(h, ax, ay, nullptr, nullptr) <= r2) {
        out.hit = true;
        out.alpha = 0.0;
        return out; // reported point stays the far end b
    }
    for (int i = 0; i < h.n; ++i) {
        const double nx = h.nx(i), ny = h.ny(i);
        const double an = ax * nx + ay * ny;
        const double d = an - h.v0n(i) - r2;
        if (d < 0.0) continue;
        const double bn = bx * nx + by * ny;
        const double t = d / std::max(an - bn, DBL_MIN);
        if (t < 0.0 || 1.0 < t) continue;
        const double omt = 1.0 - t;
        const double ptx = ax * omt + bx * t, pty = ay * omt + by * t;
        const double dtv = nx * pty - ny * ptx;
        if (h.dtmin(i) <= dtv && dtv <= h.dtmax(i)) {
            out.hit = true;
            out.px = ptx - nx * r2;
            out.py = pty - ny * r2;
            out.alpha = t;
        }
    }
    if (r2 > 0.0) {
        for (int i = 0; i < h.n; ++i) {
            const double cx = h.v0x(i), cy = h.v0y(i);
            const double dax = ax - cx, day = ay - cy, dbx = bx - cx, dby = by - cy;
            const double daa = dax * dax + day * day, dab = dax * dbx + day * dby, dbb = dbx * dbx + dby * dby;
            const double qa = daa - 2.0 * dab + dbb;
            const double qb = dab - daa;
            const double det = qb * qb - qa * (daa - r2 * r2);
            if (det >= 0.0) {
                const double t = (-qb - std::sqrt(det)) / qa;
                if (0.0 <= t && t <= 1.0 && t < out.alpha) {
                    const double omt = 1.0 - t;
                    double nx = dax * omt + dbx * t, ny = day * omt + dby * t;
                    const double inv = 1.0 / (std::sqrt(nx * nx + ny * ny) + DBL_MIN);
                    nx *= inv; ny *= inv;
                    out.hit = true;
                    out.px = (ax * omt + bx * t) - nx * r2;
                    out.py = (ay * omt + by * t) - ny * r2;
                    out.alpha = t;
                }
            }
        }
    }
    return out;
}

HullView hull_of_record(const double *rec, int side)
{
    return HullView{(int)rec[SSG_MAP_OFF_COUNTS + side],
                    rec + SSG_MAP_OFF_PLANES + side * (SSG_MAX_HULL * SSG_PLANE_DOUBLES)};
}

const double kShipTemplate[SSG_SHIP_VERTS][2] = {{0, 0}, {0, 10}, {5, 15}, {10, 10}, {10, 0}}; // models.py:6

double moment_for_poly(double m, int n, const double *v)
{
    // cpMomentForPoly, offset (0,0): about the local origin
    double sum1 = 0.0, sum2 = 0.0;
    for (int i = 0; i < n; ++i) {
        const double x1 = v[2 * i], y1 = v[2 * i + 1];
        const int j = (i + 1) % n;
        const double x2 = v[2 * j], y2 = v[2 * j + 1];
        const double a = x2 * y1 - y2 * x1;
        const double b = (x1 * x1 + y1 * y1) + (x1 * x2 + y1 * y2) + (x2 * x2 + y2 * y2);
        sum1 += a * b;
        sum2 += a;
    }
    return (m * sum1) / (6.0 * sum2);
}

int set_ship(ssg_config *cfg, double ws, double hs, double mass)
{
    double pts[2 * SSG_SHIP_VERTS];
    std::vector<P2> v(SSG_SHIP_VERTS);
    for (int i = 0; i < SSG_SHIP_VERTS; ++i) {
        pts[2 * i] = kShipTemplate[i][0] * ws;
        pts[2 * i + 1] = kShipTemplate[i][1] * hs;
        v[i] = P2{pts[2 * i], pts[2 * i + 1]};
    }
    const double moment = moment_for_poly(mass, SSG_SHIP_VERTS, pts); // on the template order, as models.py:88-89
    std::vector<P2> hull = convex_hull(v);
    if ((int)hull.size() != SSG_SHIP_VERTS) return SSG_ERR_BAD_ARG;
    std::vector<Plane> pl = planes_of(hull);
    for (int i = 0; i < SSG_SHIP_VERTS; ++i) {
        cfg->ship_hull[2 * i] = hull[i].x;
        cfg->ship_hull[2 * i + 1] = hull[i].y;
        cfg->ship_normals[2 * i] = pl[i].nx;
        cfg->ship_normals[2 * i + 1] = pl[i].ny;
    }
    cfg->ship_m_inv = 1.0 / mass;
    cfg->ship_i_inv = 1.0 / moment;
    return SSG_OK;
}

// ShipGame.add_default_traffic (game.py:279-286): add_ship(x, y, width, height) -> Ship.__init__ (models.py:87-111),
// mass = the Ship default, moment about the local origin, hull in cpConvexHull order; plus Chipmunk's space defaults.
int set_traffic(ssg_handle *h)
{
    static const double kTraffic[SSG_N_TRAFFIC][4] = {{100, 200, 1, 1}, {300, 200, 1.5, 2}, {400, 350, 1, 3}};
    ssg::DynCfg &d = h->dyn;
    const double mass = 1.0 / h->cfg.ship_m_inv;
    for (int k = 0; k < SSG_N_TRAFFIC; ++k) {
        ssg_config tmp = h->cfg;
        const int rc = set_ship(&tmp, kTraffic[k][2], kTraffic[k][3], mass);
        if (rc != SSG_OK) return rc;
        std::memcpy(d.thull[k], tmp.ship_hull, sizeof(d.thull[k]));
        std::memcpy(d.tnrm[k], tmp.ship_normals, sizeof(d.tnrm[k]));
        d.t_i_inv[k] = tmp.ship_i_inv;
        d.tx[k] = kTraffic[k][0];
        d.ty[k] = kTraffic[k][1];
    }
    d.t_m_inv = h->cfg.ship_m_inv;
    // add_goal (game.py:77-95): mass 1, pm.moment_for_circle(1, 0, radius) = m * 0.5 * (r1^2 + r2^2)
    d.goal_m_inv = 1.0 / 1.0;
    d.goal_i_inv = 1.0 / (1.0 * 0.5 * (0.0 * 0.0 + h->cfg.goal_radius * h->cfg.goal_radius));
    d.ship_friction = 0.7;                                      // models.py:98
    d.slop = 0.1;                                               // cpSpace collisionSlop
    d.bias_coef = 1.0 - std::pow(std::pow(1.0 - 0.1, 60.0), h->cfg.dt); // 1 - collisionBias^dt (cpSpaceStep)
    return SSG_OK;
}

void refresh_dev(ssg_handle *h)
{
    const ssg_config &c = h->cfg;
    ssg::DevCfg &d = h->dev;
    d.n_envs = c.n_envs;
    d.n_pad = h->n_pad;
    d.env_id_base = c.env_id_base;
    d.n_beams = c.n_beams;
    d.history = c.history < 2 ? c.history : 2;
    d.full_history = c.history;
    d.max_steps = c.max_steps;
    d.n_goals = c.n_goals;
    d.flags = c.flags;
    d.n_maps = h->n_maps;
    d.map_ring = c.map_ring;
    d.rudder_step = c.rudder_step;
    d.rudder_max = c.rudder_max;
    d.spread_deg = c.lidar_spread_deg;
    d.lidar_dist = c.lidar_dist;
    d.goal_r = c.goal_radius;
    d.width = c.width;
    d.height = c.height;
    d.dt = c.dt;
    d.damp = c.damping_pow_dt;
    d.spawn_x = c.spawn_x;
    d.spawn_y = c.spawn_y;
    std::memcpy(d.hull, c.ship_hull, sizeof(d.hull));
    std::memcpy(d.nrm, c.ship_normals, sizeof(d.nrm));
    d.m_inv = c.ship_m_inv;
    d.i_inv = c.ship_i_inv;
    d.force_y = c.force_y;
    d.px0 = c.thrust_px0;
    d.py0 = c.thrust_py0;
    {
        // LiDAR.query (models.py:48-49,62): rotation_i = angle + rad(90 - spread/2) + rad(spread/n_beams) * i
        const double deg2rad = 0.017453292519943295; // CPython math.radians: pi/180
        const double delta = (c.lidar_spread_deg / (double)c.n_beams) * deg2rad;
        const double phi0 = (90.0 - c.lidar_spread_deg / 2) * deg2rad;
        for (int i = 0; i < SSG_MAX_BEAMS; ++i) {
            const double phi = phi0 + delta * (double)i;
            d.beam_cos[i] = std::cos(phi);
            d.beam_sin[i] = std::sin(phi);
        }
    }
    char *base = static_cast<char *>(h->state);
    d.stats = base ? reinterpret_cast<double *>(base + h->off_stats) : nullptr;
    d.f64cols = base ? reinterpret_cast<double *>(base + h->off_f64) : nullptr;
    d.i32cols = base ? reinterpret_cast<int32_t *>(base + h->off_i32) : nullptr;
    d.mask = base ? reinterpret_cast<uint8_t *>(base + h->off_mask) : nullptr;
    d.obs2 = (base && c.history > 2) ? reinterpret_cast<double *>(base + h->off_obs2) : nullptr;
    d.obsH = (base && c.history > 2) ? reinterpret_cast<double *>(base + h->off_obsH) : nullptr;
    d.bank = h->bank;
    d.n_ships = c.n_ships;
    const bool dyn = base && c.n_ships > 1;
    d.dyn_f64 = dyn ? reinterpret_cast<double *>(base + h->off_dyn_f64) : nullptr;
    d.dyn_live = dyn ? reinterpret_cast<unsigned long long *>(base + h->off_dyn_live) : nullptr;
    d.dyn_u32 = dyn ? reinterpret_cast<uint32_t *>(base + h->off_dyn_u32) : nullptr;
    d.dyn_flag = dyn ? reinterpret_cast<uint8_t *>(base + h->off_dyn_flag) : nullptr;
    d.dyn_hash = dyn ? reinterpret_cast<unsigned long long *>(base + h->off_dyn_hash) : nullptr;
    d.dyn_count = dyn ? reinterpret_cast<unsigned *>(base + h->off_dyn_count) : nullptr;
    d.dyn_bucket = dyn ? reinterpret_cast<int32_t *>(base + h->off_dyn_bucket) : nullptr;
    d.dyn_qmap = dyn ? reinterpret_cast<int32_t *>(base + h->off_dyn_qmap) : nullptr;
    d.dyn_par = 0; // (the queue is rebuilt from the flags after every refresh: dyn_queue_valid = false below)
    d.dyn_row = dyn ? reinterpret_cast<double *>(base + h->off_dyn_row) : nullptr;
    // the memo: shared bank records are what makes states repeat (not per-env worlds; the key holds 16 bits of record index)
    const bool memo = dyn && !(c.flags & SSG_FLAG_DYN_MEMO_OFF) && c.map_ring == 0 && h->n_maps > 0 && h->n_maps <= 65536;
    d.dyn_memo = memo ? reinterpret_cast<unsigned long long *>(base + h->off_dyn_memo) : nullptr;
    d.dyn_memo_stats = dyn ? reinterpret_cast<unsigned long long *>(base + h->off_dyn_memo_stats) : nullptr;
    d.dyn_npm = memo ? reinterpret_cast<unsigned long long *>(base + h->off_dyn_npm) : nullptr;
    d.dyn_memo_gen = h->memo_gen;
    d.dyn_seq = h->dyn_seq;
    {   // every constant the full step reads, folded into 16 bits of the memo key
        unsigned long long fp = 0xcbf29ce484222325ull;
        auto eat = [&](const void *p, size_t n) { const unsigned char *b = static_cast<const unsigned char *>(p); for (size_t i = 0; i < n; ++i) fp = (fp ^ b[i]) * 0x100000001b3ull; };
        eat(h->dyn.thull, sizeof h->dyn.thull); eat(h->dyn.tnrm, sizeof h->dyn.tnrm); eat(h->dyn.tx, sizeof h->dyn.tx); eat(h->dyn.ty, sizeof h->dyn.ty);
        eat(h->dyn.t_i_inv, sizeof h->dyn.t_i_inv); eat(&h->dyn.t_m_inv, 8); eat(&h->dyn.goal_m_inv, 8); eat(&h->dyn.goal_i_inv, 8);
        eat(&h->dyn.ship_friction, 8); eat(&h->dyn.bias_coef, 8); eat(&h->dyn.slop, 8);
        eat(&c.dt, 8); eat(&c.damping_pow_dt, 8); eat(&c.goal_radius, 8); eat(&c.n_goals, 4);
        h->dyn.memo_fp = (unsigned)((fp ^ (fp >> 16) ^ (fp >> 32) ^ (fp >> 48)) & 0xFFFFu);
    }
    {   // the step kernel's reject in front of collide_ship's exact player x traffic test: no vertex of ship k's hull is further than
        // its hull radius from its body position
        auto radius = [](const double *hull) {
            double r2 = 0.0;
            for (int i = 0; i < SSG_SHIP_VERTS; ++i) r2 = std::max(r2, hull[2 * i] * hull[2 * i] + hull[2 * i + 1] * hull[2 * i + 1]);
            return std::sqrt(r2);
        };
        for (int k = 0; k < SSG_N_TRAFFIC; ++k) {
            const double r = radius(h->dyn.thull[k]) * 1.001 + 1.0;
            d.dyn_reach2[k] = r * r;
        }
        std::memcpy(d.thull, h->dyn.thull, sizeof(d.thull));
        std::memcpy(d.tnrm, h->dyn.tnrm, sizeof(d.tnrm));
    }
    h->dyn_queue_valid = false; // (anything that refreshes the kernel arguments may have changed what the queue was built from)
}


// The memo of the full dyn step: a new GENERATION makes every stored entry count as empty (no memset: the generation is part of
// the tag).  Started whenever the bank changes (entries of the old bank could never match again and would only fill the table)
// and every kMemoGenSteps launches (states a caller poked in once, or a long tail of rare ones, do not silt the table up).
constexpr int kMemoGenSteps = 1 << 15;
static void memo_new_generation(ssg_handle *h)
{
    h->memo_gen += 1;
    if (h->memo_gen > 255) { h->memo_gen = 1; h->memo_clear_pending = true; } // a wrapped generation could meet its own old tags
    h->memo_steps = 0;
    h->dev.dyn_memo_gen = h->memo_gen;
}

int pick_block(int n_envs)
{
    if (const char *s = std::getenv("SSG_BLOCK")) {
        const int b = std::atoi(s);
        if (b == 64 || b == 128 || b == 256) return b;
    }
    // Envs per workgroup (the workgroup has twice as many threads: role-split waves).  MI355X has 256 CUs and a
    // workgroup that stages the bank owns its CU's LDS: aim for >= 256 workgroups before growing them.
    if (n_envs <= 64 * 256) return 64;
    if (n_envs <= 128 * 256) return 128;
    return 256;
}

// Config 4 and map_ring launches carry per-call HOST state in their kernel arguments (which counter set holds the live queue, the
// launch number the memo's entries are stamped with, the ring's credit): captured into a HIP graph they would be replayed with
// the arguments of the captured call — silently wrong steps.  Refuse a capturing stream for those handles.  (1-ship handles on a
// shared bank launch with constant arguments and can be captured.)
int refuse_capture(ssg_handle *h, void *stream, const char *what)
{
    if (h->cfg.n_ships <= 1 && h->cfg.map_ring <= 0 && h->cfg.history <= 2) return SSG_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs) != hipSuccess) { (void)hipGetLastError(); return SSG_OK; }
    if (cs == hipStreamCaptureStatusNone) return SSG_OK;
    return fail(h, SSG_ERR_UNSUPPORTED, std::string(what) + ": the stream is capturing a HIP graph, and this handle's launches (n_ships > 1, map_ring or "
                                        "history > 2) take per-call host state in their arguments — a replay would repeat the captured step's");
}

int check_ready(ssg_handle *h, bool need_bank)
{
    if (!h) return SSG_ERR_BAD_ARG;
    if (!h->state) return fail(h, SSG_ERR_NOT_BOUND, "no state blob bound: call ssg_bind_state first");
    if (need_bank && (!h->bank || h->n_maps <= 0))
        return fail(h, SSG_ERR_NOT_BOUND, "no map bank set: call ssg_set_map_bank first");
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(h, SSG_ERR_NO_DEVICE, std::string("hipGetDevice: ") + hipGetErrorString(e));
    if (dev != h->cfg.device_id) {
        char buf[160];
        std::snprintf(buf, sizeof buf, "current HIP device %d differs from the handle's device %d", dev,
                      h->cfg.device_id);
        return fail(h, SSG_ERR_BAD_ARG, buf);
    }
    return SSG_OK;
}

} // namespace

extern "C" {

int ssg_abi_version(void) { return SSG_ABI_VERSION; }

const char *ssg_strerror(int status)
{
    switch (status) {
    case SSG_OK: return "ok";
    case SSG_ERR_BAD_ARG: return "bad argument";
    case SSG_ERR_HIP: return "HIP runtime error";
    case SSG_ERR_NOT_BOUND: return "state or map bank not bound";
    case SSG_ERR_UNSUPPORTED: return "unsupported configuration";
    case SSG_ERR_NO_DEVICE: return "no usable HIP device";
    default: return "unknown status";
    }
}

const char *ssg_last_error(const ssg_handle *h) { return h ? h->err.c_str() : g_err.c_str(); }

int ssg_default_config(ssg_config *cfg)
{
    if (!cfg) return fail(nullptr, SSG_ERR_BAD_ARG, "cfg is NULL");
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (uint32_t)sizeof(ssg_config);
    cfg->flags = SSG_FLAG_AUTO_RESET;
    cfg->device_id = 0;
    cfg->n_envs = 1;
    cfg->env_id_base = 0;
    cfg->n_beams = 10;          // models.py:29
    cfg->history = 2;           // config.py:15
    cfg->max_steps = 1000;      // config.py:16
    cfg->n_goals = 5;           // game.py:17
    cfg->lidar_spread_deg = 90; // models.py:29
    cfg->lidar_dist = 100;
    cfg->goal_radius = 5;       // game.py:82
    cfg->width = 600;           // config.py:24
    cfg->height = 600;
    cfg->dt = 10 * 0.1;         // SPEED * base_dt, config.py:23, game.py:27
    cfg->damping_pow_dt = std::pow(0.4, cfg->dt); // game.py:270
    cfg->spawn_x = 600 / 2.0;   // game.py:274
    cfg->spawn_y = 25;
    cfg->force_y = 100;         // models.py:107
    cfg->thrust_px0 = 0.0;      // models.py:109: shape.bb.center() of a shape not yet in a space
    cfg->thrust_py0 = 0.0;
    cfg->rudder_step = 5;       // game.py:149-151
    cfg->rudder_max = 10;       // models.py:110
    cfg->n_ships = 1;           // ShipGame.reset adds the player only (game.py:274-275)
    return set_ship(cfg, 2.0, 3.0, 5.0); // game.py:275, models.py:87
}

int ssg_config_set_ship(ssg_config *cfg, double width_scale, double height_scale, double mass)
{
    if (!cfg || !(mass > 0.0) || !(width_scale > 0.0) || !(height_scale > 0.0))
        return fail(nullptr, SSG_ERR_BAD_ARG, "ssg_config_set_ship: bad argument");
    return set_ship(cfg, width_scale, height_scale, mass);
}

int ssg_create(const ssg_config *cfg, ssg_handle **out)
{
    if (!cfg || !out) return fail(nullptr, SSG_ERR_BAD_ARG, "ssg_create: NULL argument");
    if (cfg->struct_size != sizeof(ssg_config)) return fail(nullptr, SSG_ERR_BAD_ARG, "ssg_create: struct_size mismatch");
    if (cfg->n_envs < 1) return fail(nullptr, SSG_ERR_BAD_ARG, "ssg_create: n_envs must be >= 1");
    if (cfg->n_beams < 1 || cfg->n_beams > SSG_MAX_BEAMS)
        return fail(nullptr, SSG_ERR_UNSUPPORTED, "ssg_create: n_beams must be in 1..16");
    // ship_env.py:46-47 raises ValueError("history_size must be greater than zero")
    if (cfg->history < 1) return fail(nullptr, SSG_ERR_BAD_ARG, "history_size must be greater than zero");
    if (cfg->history > SSG_MAX_HISTORY) return fail(nullptr, SSG_ERR_UNSUPPORTED, "ssg_create: history must be <= 8");
    if (cfg->n_goals < 1 || cfg->n_goals > SSG_MAX_GOALS) return fail(nullptr, SSG_ERR_UNSUPPORTED, "ssg_create: n_goals must be in 1..6");
    if (!(cfg->dt > 0.0)) return fail(nullptr, SSG_ERR_BAD_ARG, "ssg_create: dt must be > 0");
    if (cfg->max_steps < 1) return fail(nullptr, SSG_ERR_BAD_ARG, "ssg_create: max_steps must be >= 1");
    if (cfg->n_ships != 0 && cfg->n_ships != 1 && cfg->n_ships != 1 + SSG_N_TRAFFIC)
        return fail(nullptr, SSG_ERR_UNSUPPORTED, "ssg_create: n_ships must be 1 or 4 (player + add_default_traffic)");
    if (cfg->map_ring != 0 && (cfg->map_ring < 2 || cfg->map_ring > 128))
        return fail(nullptr, SSG_ERR_UNSUPPORTED, "ssg_create: map_ring must be 0 or in 2..128");
    // (record offsets are 32-bit: record index x SSG_MAP_STRIDE doubles must stay below 2^31)
    if (cfg->map_ring != 0 && (long long)cfg->n_envs * cfg->map_ring * SSG_MAP_STRIDE > 2147483647LL)
        return fail(nullptr, SSG_ERR_UNSUPPORTED, "ssg_create: n_envs * map_ring * SSG_MAP_STRIDE must be < 2^31 (use a smaller ring, or shard the envs over more handles)");
    if (cfg->map_ring != 0 && cfg->history > 2)
        return fail(nullptr, SSG_ERR_UNSUPPORTED, "ssg_create: map_ring needs history <= 2");
    if (cfg->n_ships > 1 && (cfg->flags & SSG_FLAG_EXACT_LIDAR))
        return fail(nullptr, SSG_ERR_UNSUPPORTED, "ssg_create: SSG_FLAG_EXACT_LIDAR is not built for n_ships = 4");
    ssg_handle *h = new (std::nothrow) ssg_handle();
    if (!h) return fail(nullptr, SSG_ERR_BAD_ARG, "ssg_create: out of host memory");
    h->cfg = *cfg;
    if (h->cfg.n_ships == 0) h->cfg.n_ships = 1;
    h->n_pad = (cfg->n_envs + ssg::kPadEnvs - 1) / ssg::kPadEnvs * ssg::kPadEnvs;
    const size_t np = (size_t)h->n_pad;
    h->off_stats = 0;
    h->off_f64 = ssg::kStatsDoubles * sizeof(double);
    h->off_i32 = h->off_f64 + (size_t)(ssg::COL_LIDAR + cfg->n_beams) * np * sizeof(double);
    h->off_mask = h->off_i32 + (size_t)ssg::ICOL_COUNT * np * sizeof(int32_t);
    h->off_obs2 = (h->off_mask + np + 255) & ~(size_t)255;
    // history > 2: the step kernel's two-frame staging rows, then the handle's own [n][H*F] observation rows
    h->off_obsH = h->off_obs2 + np * (size_t)(2 * (6 + cfg->n_beams)) * sizeof(double);
    h->nbytes = (cfg->history > 2) ? h->off_obsH + np * (size_t)(cfg->history * (6 + cfg->n_beams)) * sizeof(double) : h->off_mask + np;
    if (cfg->map_ring > 0) { // work queue of the ring refill: (env, episode) pairs + its length
        h->off_ring_queue = (h->nbytes + 255) & ~(size_t)255;
        h->off_ring_count = h->off_ring_queue + np * (size_t)cfg->map_ring * sizeof(unsigned long long);
        h->nbytes = h->off_ring_count + 256;
        h->cfg.flags |= SSG_FLAG_BANK_IN_GLOBAL; // n_envs * R records never fit LDS
    }
    if (h->cfg.n_ships > 1) {
        // config 4: columns of the traffic ships, goal bodies and cached arbiters (shipsim_internal.h DC_* / DU_*)
        h->off_dyn_f64 = (h->nbytes + 255) & ~(size_t)255;
        h->off_dyn_live = h->off_dyn_f64 + (size_t)ssg::DC_COUNT * np * sizeof(double);
        h->off_dyn_u32 = h->off_dyn_live + np * sizeof(unsigned long long);
        h->off_dyn_flag = h->off_dyn_u32 + (size_t)ssg::DU_COUNT * np * sizeof(uint32_t);
        h->off_dyn_hash = h->off_dyn_flag + np;
        h->off_dyn_count = h->off_dyn_hash + np * sizeof(unsigned long long);
        h->off_dyn_row = h->off_dyn_count + ((2 * (size_t)ssg::kDynCountWords * sizeof(unsigned) + 255) & ~(size_t)255);
        h->off_dyn_bucket = h->off_dyn_row + np * (size_t)ssg::kDynRow * sizeof(double);
        h->off_dyn_memo_stats = (h->off_dyn_bucket + (size_t)ssg::kDynBuckets * np * sizeof(int32_t) + 255) & ~(size_t)255; // one array of n_pad slots per sort bucket
        h->off_dyn_memo = h->off_dyn_memo_stats + (size_t)ssg::kMemoStatSlots * ssg::kMemoStatWords * sizeof(unsigned long long);
        h->off_dyn_npm = h->off_dyn_memo + (size_t)ssg::kMemoEntries * ssg::kMemoStride * sizeof(unsigned long long);
        h->off_dyn_qmap = h->off_dyn_npm + (size_t)ssg::kNpmEntries * ssg::kNpmStride * sizeof(unsigned long long);
        h->nbytes = h->off_dyn_qmap + np * sizeof(int32_t);
        const int rc = set_traffic(h);
        if (rc != SSG_OK) { delete h; return fail(nullptr, rc, "ssg_create: traffic ship geometry"); }
    }
    h->block = pick_block(cfg->n_envs);
    if (h->cfg.n_ships > 1 && h->block == 128) h->block = 64; // the config-4 step kernel is built for 64 and 256
    refresh_dev(h);
    *out = h;
    return SSG_OK;
}

int ssg_destroy(ssg_handle *h)
{
    if (h)
        for (int i = 0; i < ssg_handle::kHostSlots; ++i)
            if (h->host_ev_made[i]) (void)hipEventDestroy(h->host_ev[i]);
    delete h;
    return SSG_OK;
}

int ssg_state_nbytes(const ssg_handle *h, size_t *nbytes)
{
    if (!h || !nbytes) return SSG_ERR_BAD_ARG;
    *nbytes = h->nbytes;
    return SSG_OK;
}

int ssg_state_field(const ssg_handle *h, int field, size_t *offset, int *elem_size, int *n_columns,
                    size_t *column_stride_bytes)
{
    if (!h || !offset || !elem_size || !n_columns || !column_stride_bytes) return SSG_ERR_BAD_ARG;
    const size_t np = (size_t)h->n_pad;
    size_t off;
    int es, nc;
    if (field >= SSG_F_X && field <= SSG_F_CUM_REWARD) {
        off = h->off_f64 + (size_t)field * np * 8; es = 8; nc = 1;
    } else if (field == SSG_F_LIDAR) {
        off = h->off_f64 + (size_t)ssg::COL_LIDAR * np * 8; es = 8; nc = h->cfg.n_beams;
    } else if (field == SSG_F_RUDDER || field == SSG_F_STEP_COUNT || field == SSG_F_MAP_ID) {
        off = h->off_i32 + (size_t)(field - SSG_F_RUDDER) * np * 4; es = 4; nc = 1;
    } else if (field == SSG_F_GOAL_MASK) {
        off = h->off_mask; es = 1; nc = 1;
    } else if (field == SSG_F_TRAFFIC || field == SSG_F_GOAL_BODIES) {
        if (h->cfg.n_ships <= 1) return SSG_ERR_BAD_ARG;
        const int c0 = field == SSG_F_TRAFFIC ? ssg::DC_TRAFFIC : ssg::DC_GOALS;
        off = h->off_dyn_f64 + (size_t)c0 * np * 8; es = 8;
        nc = field == SSG_F_TRAFFIC ? 9 * SSG_N_TRAFFIC : ssg::DC_GOAL_COLS * SSG_MAX_GOALS;
    } else if (field == SSG_F_EPISODES) {
        off = h->off_i32 + (size_t)ssg::ICOL_EPISODE * np * 4; es = 4; nc = 1;
    } else if (field == SSG_F_DYN_FLAGS) {
        if (h->cfg.n_ships <= 1) return SSG_ERR_BAD_ARG;
        off = h->off_dyn_flag; es = 1; nc = 1;
    } else if (field == SSG_F_DYN_MEMO_STATS) {
        if (h->cfg.n_ships <= 1) return SSG_ERR_BAD_ARG;
        off = h->off_dyn_memo_stats; es = 8; nc = ssg::kMemoStatSlots * ssg::kMemoStatWords;
        *offset = off; *elem_size = es; *n_columns = nc; *column_stride_bytes = 8;
        return SSG_OK;
    } else if (field == SSG_F_STATS) {
        // kStatsSlots rows of 4 int64 counters; sum over rows: [0] sum_return*100 [1] sum_length [2] episodes [3] goals
        off = h->off_stats; es = 8; nc = 4 * ssg::kStatsSlots;
        *offset = off; *elem_size = es; *n_columns = nc; *column_stride_bytes = 8;
        return SSG_OK;
    } else {
        return SSG_ERR_BAD_ARG;
    }
    *offset = off; *elem_size = es; *n_columns = nc; *column_stride_bytes = np * (size_t)es;
    return SSG_OK;
}

int ssg_bind_state(ssg_handle *h, void *dev_state)
{
    if (!h || !dev_state) return fail(h, SSG_ERR_BAD_ARG, "ssg_bind_state: NULL argument");
    if (reinterpret_cast<uintptr_t>(dev_state) % 256 != 0)
        return fail(h, SSG_ERR_BAD_ARG, "ssg_bind_state: state blob must be 256-byte aligned");
    h->state = dev_state;
    h->zeroed = false;
    // whatever memo tables the blob carries were not filled by this handle on this bank: they are zeroed before the first launch
    // that could read them (a blob this handle filled itself — a snapshot copied back IN PLACE — keeps its entries: they are results
    // of the same pure function)
    memo_new_generation(h);
    h->memo_clear_pending = true;
    refresh_dev(h);
    return SSG_OK;
}

static int ring_refill(ssg_handle *h, double *dev_raw, void *stream);

int ssg_init_state(ssg_handle *h, void *stream)
{
    int rc = check_ready(h, false);
    if (rc != SSG_OK) return rc;
    hipError_t e = hipMemsetAsync(h->state, 0, h->nbytes, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(h, SSG_ERR_HIP, std::string("ssg_init_state: ") + hipGetErrorString(e));
    h->zeroed = true;
    h->dyn_queue_valid = false;
    // map_ring mode: the episode counters and the rings' "worlds drawn" counters were just zeroed together, so the rings
    // must hold episodes 0 .. R-1 again: redraw them (the bookkeeping and the bank never disagree)
    if (h->cfg.map_ring > 0 && h->ring_ready) return ring_refill(h, nullptr, stream);
    return SSG_OK;
}

int ssg_set_map_bank(ssg_handle *h, const double *dev_bank, int n_maps)
{
    if (!h || !dev_bank || n_maps < 1) return fail(h, SSG_ERR_BAD_ARG, "ssg_set_map_bank: bad argument");
    if ((long long)n_maps * SSG_MAP_STRIDE > 2147483647LL) return fail(h, SSG_ERR_UNSUPPORTED, "ssg_set_map_bank: n_maps * SSG_MAP_STRIDE must be < 2^31");
    if (reinterpret_cast<uintptr_t>(dev_bank) % 16 != 0)
        return fail(h, SSG_ERR_BAD_ARG, "ssg_set_map_bank: bank must be 16-byte aligned");
    if (h->cfg.map_ring > 0 && (long long)n_maps != (long long)h->cfg.n_envs * h->cfg.map_ring)
        return fail(h, SSG_ERR_BAD_ARG, "ssg_set_map_bank: map_ring mode needs n_maps == n_envs * map_ring");
    if (h->cfg.map_ring > 0) h->ring_ready = false; // a new bank: its rings are empty until ssg_refill_worlds
    // Envs per workgroup, and whether the bank is staged in the CU's LDS or gathered from L2 / HBM.  Staged: the largest size, from
    // the one preferred for this env count down, at which the bank fits the LDS beside the lidar buffers.  Gathered: the largest
    // size at which the record heads fit.  A staged bank wins at EQUAL workgroup size (65 536 envs, 10 beams, 64 records: 6.45
    // against 7.2 us per step), but a workgroup smaller than the preferred one means more workgroups than the chip holds at once
    // — they own their CU's LDS — i.e. launches of several rounds: 65 536 envs x 8 beams on 120 records took 16.1 us per step
    // staged on 64-env workgroups (four rounds) against 6.4 gathered on 256-env ones; 100 records, 128-env workgroups: 8.6
    // against 6.5.  (Round 5; before, the gathered path was only taken when nothing could be staged, and then on 64-env
    // workgroups, whose instantiation holds 134 VGPRs = three per CU: two rounds again.)
    const bool dyn_cfg = h->cfg.n_ships > 1;
    auto fix_dyn = [&](int b) { return (dyn_cfg && b == 128) ? 64 : b; }; // the config-4 step kernel is built for 64 and 256
    const int preferred = fix_dyn(pick_block(h->cfg.n_envs));
    int staged_block = preferred, gathered_block = preferred;
    while (staged_block > 64 && ssg::step_lds_bytes(h->cfg.n_beams, staged_block, true, n_maps, dyn_cfg) > 160u * 1024u) staged_block = fix_dyn(staged_block / 2);
    while (gathered_block > 64 && ssg::step_lds_bytes(h->cfg.n_beams, gathered_block, false, n_maps, dyn_cfg) > 160u * 1024u) gathered_block = fix_dyn(gathered_block / 2);
    const bool can_stage = !(h->cfg.flags & SSG_FLAG_BANK_IN_GLOBAL) &&
                           ssg::step_lds_bytes(h->cfg.n_beams, staged_block, true, n_maps, dyn_cfg) <= 160u * 1024u;
    if (h->bank && n_maps < h->n_maps) h->remap_pending = true; // stale record indices >= n_maps must not survive
    h->bank = dev_bank;
    h->n_maps = n_maps;
    h->dyn.bank_epoch++; // resting traffic must be re-collided against the new banks
    memo_new_generation(h);
    h->lds = can_stage && (staged_block == preferred || gathered_block <= staged_block);
    h->block = h->lds ? staged_block : gathered_block;
    h->lds_bytes = ssg::step_lds_bytes(h->cfg.n_beams, h->block, h->lds, n_maps, h->cfg.n_ships > 1);
    h->prepared = false;
    refresh_dev(h);
    return SSG_OK;
}

// map_ring mode: draw the worlds the rings are missing and restore the credit of R-1 episode starts per env
static int ring_refill(ssg_handle *h, double *dev_raw, void *stream)
{
    char *base = static_cast<char *>(h->state);
    hipError_t e = ssg::launch_refill_worlds(h->dev, h->ring_seed, h->ring_width_frac,
                                             reinterpret_cast<unsigned long long *>(base + h->off_ring_queue),
                                             reinterpret_cast<unsigned *>(base + h->off_ring_count),
                                             const_cast<double *>(h->bank), dev_raw, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(h, SSG_ERR_HIP, std::string("ring refill launch: ") + hipGetErrorString(e));
    h->ring_credit = h->cfg.map_ring - 1;
    // (A refill draws the worlds of FUTURE episodes — episodes current+1 .. current+R-1 of every env, into the R-1 slots its current
    // episode does not occupy — so no env's current world changes: the rest bits (tagged with the bank epoch) and the queue of the
    // next full cpSpaceStep stay valid.  Until round 6 a refill bumped the epoch and dropped the queue: every env of a config-4
    // handle was classified and stepped in full once per refill cycle, ~13 us per step on rings of 32.)
    return SSG_OK;
}

static int flush_remap(ssg_handle *h, void *stream)
{
    if (!h->remap_pending) return SSG_OK;
    hipError_t e = ssg::launch_remap_map_ids(h->dev, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(h, SSG_ERR_HIP, std::string("map id remap launch: ") + hipGetErrorString(e));
    h->remap_pending = false;
    return SSG_OK;
}

int ssg_reset(ssg_handle *h, const uint8_t *dev_mask, const int32_t *dev_map_ids, double *dev_obs, void *stream)
{
    int rc = check_ready(h, true);
    if (rc != SSG_OK) return rc;
    rc = refuse_capture(h, stream, "ssg_reset");
    if (rc != SSG_OK) return rc;
    if (!dev_mask && !h->zeroed) { // first full reset on a freshly bound blob: start from zeroed counters / columns
        rc = ssg_init_state(h, stream);
        if (rc != SSG_OK) return rc;
    }
    rc = flush_remap(h, stream);
    if (rc != SSG_OK) return rc;
    if (h->cfg.map_ring > 0) {
        if (dev_map_ids) // episode p of env e lives in record e*R + p mod R: there is no other record to put it on
            return fail(h, SSG_ERR_BAD_ARG, "ssg_reset: dev_map_ids must be NULL in map_ring mode (an env's records are its own ring)");
        if (!h->ring_ready) return fail(h, SSG_ERR_NOT_BOUND, "map_ring mode: call ssg_refill_worlds before the first ssg_reset");
        if (h->ring_credit < 1) { // a reset starts an episode: make sure every ring still holds an unused world
            rc = ring_refill(h, nullptr, stream);
            if (rc != SSG_OK) return rc;
        }
        h->ring_credit -= 1;
    }
    // Config 4: a MASKED reset between two steps joins the queue the step kernel left for the next full cpSpaceStep (the RLlib flow
    // resets its done envs this way after every step; rebuilding the queue from the per-env flags instead — memset + the classify
    // pass over every env — cost the next step 12-19 us).  One such reset per step: every env then has at most one entry per sort
    // bucket, which is what a bucket's n_pad slots hold; a second one, a full reset, or a queue that is not there rebuild it.
    // Only where the full step runs its one-record-per-wave kernel (shared banks of at most kDynMapBuckets records): that kernel
    // drops an entry queued under another record than the env now sits on, and a record IS its map bucket there.  The per-lane-planes
    // kernel (larger banks, map_ring) cannot tell a stale entry from the new one — both would be stepped, by different waves,
    // i.e. an env's bodies could advance twice in one step — so those handles rebuild the queue after any reset.
    const bool dyn_uni = h->cfg.map_ring == 0 && h->n_maps <= ssg::kDynMapBuckets;
    const bool append = h->cfg.n_ships > 1 && dev_mask && h->dyn_queue_valid && h->masked_resets_since_step == 0 && dyn_uni;
    if (append) h->masked_resets_since_step = 1;
    else h->dyn_queue_valid = false;
    hipError_t e;
    if (h->cfg.n_ships > 1) // the player, then add_default_traffic + fresh goal bodies of the reset envs: one launch
        e = ssg::launch_dyn_reset(h->dev, h->dyn, dev_mask, dev_map_ids, dev_obs, append, static_cast<hipStream_t>(stream));
    else
        e = ssg::launch_reset(h->dev, dev_mask, dev_map_ids, dev_obs, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(h, SSG_ERR_HIP, std::string("reset launch: ") + hipGetErrorString(e));
    return SSG_OK;
}

static int prepare(ssg_handle *h)
{
    if (h->prepared) return SSG_OK;
    hipError_t e = ssg::prepare_step(h->dev, h->block, h->lds, h->lds_bytes);
    if (e != hipSuccess) return fail(h, SSG_ERR_HIP, std::string("prepare_step: ") + hipGetErrorString(e));
    if (h->cfg.n_ships > 1) {
        e = ssg::prepare_dyn(h->dev);
        if (e != hipSuccess) return fail(h, SSG_ERR_HIP, std::string("prepare_dyn: ") + hipGetErrorString(e));
    }
    h->prepared = true;
    return SSG_OK;
}

int ssg_step(ssg_handle *h, const int32_t *dev_actions, double *dev_obs, double *dev_reward, uint8_t *dev_done,
             uint8_t *dev_flags, void *stream)
{
    return ssg_rollout_traj(h, dev_actions, 1, dev_obs, dev_reward, dev_done, dev_flags, 0, stream);
}

int ssg_step_host(ssg_handle *h, const int32_t *host_actions, int32_t *dev_actions, double *dev_obs, double *dev_reward, uint8_t *dev_done,
                  uint8_t *dev_flags, const void *dev_block, void *host_block, size_t block_bytes, int slot, void *stream)
{
    if (!h || !host_actions || !dev_actions || !dev_block || !host_block || block_bytes == 0 || slot < 0 || slot >= ssg_handle::kHostSlots)
        return fail(h, SSG_ERR_BAD_ARG, "ssg_step_host: NULL buffer, empty block or slot out of range");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!h->host_ev_made[slot]) {
        if (hipEventCreateWithFlags(&h->host_ev[slot], hipEventDisableTiming) != hipSuccess) return fail(h, SSG_ERR_HIP, "ssg_step_host: hipEventCreate failed");
        h->host_ev_made[slot] = true;
    }
    hipError_t e = hipMemcpyAsync(dev_actions, host_actions, (size_t)h->cfg.n_envs * sizeof(int32_t), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return fail(h, SSG_ERR_HIP, std::string("ssg_step_host: actions host -> device: ") + hipGetErrorString(e));
    int rc = ssg_step(h, dev_actions, dev_obs, dev_reward, dev_done, dev_flags, stream);
    if (rc != SSG_OK) return rc;
    e = hipMemcpyAsync(host_block, dev_block, block_bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipEventRecord(h->host_ev[slot], st);
    if (e != hipSuccess) return fail(h, SSG_ERR_HIP, std::string("ssg_step_host: outputs device -> host: ") + hipGetErrorString(e));
    return SSG_OK;
}

int ssg_wait_host(ssg_handle *h, int slot)
{
    if (!h || slot < 0 || slot >= ssg_handle::kHostSlots || !h->host_ev_made[slot]) return fail(h, SSG_ERR_BAD_ARG, "ssg_wait_host: no ssg_step_host was issued into this slot");
    hipError_t e = hipEventSynchronize(h->host_ev[slot]);
    if (e != hipSuccess) return fail(h, SSG_ERR_HIP, std::string("ssg_wait_host: ") + hipGetErrorString(e));
    return SSG_OK;
}

int ssg_rollout(ssg_handle *h, const int32_t *dev_actions, int K, double *dev_obs, double *dev_reward,
                uint8_t *dev_done, uint8_t *dev_flags, void *stream)
{
    return ssg_rollout_traj(h, dev_actions, K, dev_obs, dev_reward, dev_done, dev_flags, 0, stream);
}

int ssg_rollout_traj(ssg_handle *h, const int32_t *dev_actions, int K, double *dev_obs, double *dev_reward,
                     uint8_t *dev_done, uint8_t *dev_flags, int64_t step_stride_envs, void *stream)
{
    int rc = check_ready(h, true);
    if (rc != SSG_OK) return rc;
    if (!dev_actions || !dev_obs || !dev_reward || !dev_done || K < 1)
        return fail(h, SSG_ERR_BAD_ARG, "ssg_step/ssg_rollout: NULL buffer or K < 1");
    if (step_stride_envs != 0 && step_stride_envs < (int64_t)h->cfg.n_envs)
        return fail(h, SSG_ERR_BAD_ARG, "ssg_rollout_traj: step_stride_envs must be 0 or >= n_envs (steps would overlap)");
    rc = refuse_capture(h, stream, "ssg_step/ssg_rollout");
    if (rc != SSG_OK) return rc;
    rc = prepare(h);
    if (rc != SSG_OK) return rc;
    rc = flush_remap(h, stream);
    if (rc != SSG_OK) return rc;
    // One launch runs up to kFuse consecutive steps (state in registers, bank staged once): 100 by default, which
    // keeps a launch near a millisecond and amortises the fixed launch cost to < 1 %.  SSG_FUSE=1 in the environment
    // forces one launch per step (the path a policy-in-the-loop caller gets through ssg_step).
    static const int kFuse = [] {
        const char *s = std::getenv("SSG_FUSE");
        const int v = s ? std::atoi(s) : SSG_ROLLOUT_STEPS_PER_LAUNCH;
        return v < 1 ? 1 : v;
    }();
    const long long traj = (long long)step_stride_envs;
    const size_t D = (size_t)h->cfg.history * (size_t)(6 + h->cfg.n_beams);
    // output slots of step k (the same rows for every step when traj == 0)
    auto obs_at = [&](int k) { return dev_obs + (size_t)k * (size_t)traj * D; };
    auto rew_at = [&](int k) { return dev_reward + (size_t)k * (size_t)traj; };
    auto done_at = [&](int k) { return dev_done + (size_t)k * (size_t)traj; };
    auto flags_at = [&](int k) { return dev_flags ? dev_flags + (size_t)k * (size_t)traj : nullptr; };
    const bool dyn = h->cfg.n_ships > 1;
    if (h->cfg.history > 2 || dyn) {
        // non-default history: one launch per step into the staging rows, then the frame shift (see the kernel).
        // config 4: every step is the dyn kernel (traffic ships, goal bodies, contact solver) followed by the step
        // kernel, which reads this step's goal positions and the traffic-contact bit it left in the dyn columns.
        const bool shift = h->cfg.history > 2;
        if (h->cfg.map_ring > 0 && !h->ring_ready) return fail(h, SSG_ERR_NOT_BOUND, "map_ring mode: call ssg_refill_worlds first");
        // measurement aid: three events per step (before the full cpSpaceStep, between, after the step kernel); destroyed on every
        // way out of this call, and no timing at all if one of them cannot be created
        struct EventSet {
            std::vector<hipEvent_t> v;
            bool empty() const { return v.empty(); }
            hipEvent_t operator[](size_t i) const { return v[i]; }
            void create(size_t n) {
                v.reserve(n);
                for (size_t i = 0; i < n; ++i) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) { clear(); return; } v.push_back(ev); }
            }
            void clear() { for (auto ev : v) (void)hipEventDestroy(ev); v.clear(); }
            ~EventSet() { clear(); }
        } evs;
        if (dyn && h->time_kernels) evs.create(3 * (size_t)K);
        for (int k = 0; k < K; ++k) {
            if (!evs.empty()) (void)hipEventRecord(evs[3 * k], static_cast<hipStream_t>(stream));
            if (h->cfg.map_ring > 0) { // every step may start one episode per env: keep an unused world in every ring
                if (h->ring_credit < 1) {
                    rc = ring_refill(h, nullptr, stream);
                    if (rc != SSG_OK) return rc;
                }
                h->ring_credit -= 1;
            }
            if (dyn) {
                hipError_t e = hipSuccess;
                if (!h->dyn_queue_valid) { // the classify pass rebuilds the queue: this step's bucket counters start from zero
                    h->dev.dyn_par = 0;
                    e = hipMemsetAsync(h->dev.dyn_count, 0, ssg::kDynCountWords * sizeof(unsigned), static_cast<hipStream_t>(stream));
                }
                if (h->dev.dyn_memo) {
                    if (++h->memo_steps > kMemoGenSteps) memo_new_generation(h);
                    if (h->memo_clear_pending && e == hipSuccess) {
                        e = hipMemsetAsync(h->dev.dyn_memo, 0, ((size_t)ssg::kMemoEntries * ssg::kMemoStride + (size_t)ssg::kNpmEntries * ssg::kNpmStride) * sizeof(unsigned long long),
                                           static_cast<hipStream_t>(stream)); // (the narrowphase memo follows the state memo in the blob)
                        h->memo_clear_pending = false;
                    }
                }
                h->dev.dyn_seq = ++h->dyn_seq;
                if (!h->dyn_queue_valid) h->n_classify++;
                if (e == hipSuccess) e = ssg::launch_dyn_step(h->dev, h->dyn, !h->dyn_queue_valid, static_cast<hipStream_t>(stream));
                if (e != hipSuccess) {
                    h->dyn_queue_valid = false; // (a queue the full step never consumed must not survive: the next call starts over)
                    return fail(h, SSG_ERR_HIP, std::string("dyn step launch: ") + hipGetErrorString(e));
                }
            }
            if (!evs.empty()) (void)hipEventRecord(evs[3 * k + 1], static_cast<hipStream_t>(stream));
            hipError_t e = ssg::launch_step(h->dev, h->block, h->lds, h->lds_bytes, dev_actions + (size_t)k * h->cfg.n_envs, 1,
                                            shift ? h->dev.obs2 : obs_at(k), rew_at(k), done_at(k), flags_at(k), 0,
                                            static_cast<hipStream_t>(stream));
            if (!evs.empty()) (void)hipEventRecord(evs[3 * k + 2], static_cast<hipStream_t>(stream));
            if (e == hipSuccess && shift) e = ssg::launch_history_shift(h->dev, done_at(k), obs_at(k), static_cast<hipStream_t>(stream));
            if (e != hipSuccess) {
                h->dyn_queue_valid = false; // (the next call rebuilds the dyn queue from the flags, counters zeroed)
                return fail(h, SSG_ERR_HIP, std::string("step launch: ") + hipGetErrorString(e));
            }
            if (dyn) { // the step kernel's body role has queued the envs whose bodies must be stepped next, in the other counter set
                h->dyn_queue_valid = true;
                h->masked_resets_since_step = 0;
                h->dev.dyn_par ^= 1;
            }
        }
        if (!evs.empty()) { // (this call then waits for its own work: a measurement run, not the product path)
            (void)hipStreamSynchronize(static_cast<hipStream_t>(stream));
            for (int k = 0; k < K; ++k) {
                float a = 0.f, b = 0.f;
                (void)hipEventElapsedTime(&a, evs[3 * k], evs[3 * k + 1]);
                (void)hipEventElapsedTime(&b, evs[3 * k + 1], evs[3 * k + 2]);
                h->t_dyn_ms += a; h->t_step_ms += b; h->t_steps++;
            }
        }
        return SSG_OK;
    }
    if (h->cfg.map_ring > 0) {
        // a brand-new world per episode: an env starts at most one episode per step, and its ring holds `ring_credit`
        // unused worlds — fuse at most that many steps, then refill the rings (two small launches on the same stream)
        if (!h->ring_ready) return fail(h, SSG_ERR_NOT_BOUND, "map_ring mode: call ssg_refill_worlds first");
        for (int k = 0; k < K;) {
            if (h->ring_credit < 1) {
                rc = ring_refill(h, nullptr, stream);
                if (rc != SSG_OK) return rc;
            }
            // (up to the ring's whole credit, at most 127 steps, in ONE launch: 100 + 27 was a launch more per refill cycle)
            int kk = (K - k < h->ring_credit) ? (K - k) : h->ring_credit;
            if (kFuse < SSG_ROLLOUT_STEPS_PER_LAUNCH && kk > kFuse) kk = kFuse; // (SSG_FUSE = 1: the one-launch-per-step experiment)
            hipError_t e = ssg::launch_step(h->dev, h->block, h->lds, h->lds_bytes, dev_actions + (size_t)k * h->cfg.n_envs, kk,
                                            obs_at(k), rew_at(k), done_at(k), flags_at(k), traj, static_cast<hipStream_t>(stream));
            if (e != hipSuccess) return fail(h, SSG_ERR_HIP, std::string("step launch: ") + hipGetErrorString(e));
            h->ring_credit -= kk;
            k += kk;
        }
        return SSG_OK;
    }
    for (int k = 0; k < K; k += kFuse) {
        const int kk = (K - k < kFuse) ? (K - k) : kFuse;
        hipError_t e = ssg::launch_step(h->dev, h->block, h->lds, h->lds_bytes, dev_actions + (size_t)k * h->cfg.n_envs, kk,
                                        obs_at(k), rew_at(k), done_at(k), flags_at(k), traj, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return fail(h, SSG_ERR_HIP, std::string("step launch: ") + hipGetErrorString(e));
    }
    return SSG_OK;
}

int ssg_fill_actions(ssg_handle *h, uint64_t seed, uint64_t step0, int K, int32_t *dev_actions, void *stream)
{
    if (!h || !dev_actions || K < 1) return fail(h, SSG_ERR_BAD_ARG, "ssg_fill_actions: bad argument");
    hipError_t e = ssg::launch_fill_actions(seed, step0, K, h->cfg.env_id_base, h->cfg.n_envs, dev_actions,
                                            static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(h, SSG_ERR_HIP, std::string("fill_actions launch: ") + hipGetErrorString(e));
    return SSG_OK;
}

int ssg_set_terminal_obs(ssg_handle *h, double *dev_term_obs)
{
    if (!h) return SSG_ERR_BAD_ARG;
    if (dev_term_obs && h->cfg.history > 2)
        return fail(h, SSG_ERR_UNSUPPORTED, "ssg_set_terminal_obs: history > 2 builds its rows in the frame-shift kernel (reset the done envs with a masked ssg_reset instead)");
    if (dev_term_obs && !(h->cfg.flags & SSG_FLAG_AUTO_RESET))
        return fail(h, SSG_ERR_BAD_ARG, "ssg_set_terminal_obs: only a handle that resets its done envs in-kernel (SSG_FLAG_AUTO_RESET) replaces terminal observations");
    h->dev.term_obs = dev_term_obs;
    return SSG_OK;
}

#ifndef SSG_STAMPS
int ssg_debug_launch_clock(ssg_handle *h, uint64_t *dev_buf)
{
    if (!h) return SSG_ERR_BAD_ARG;
    h->dev.dbg = reinterpret_cast<unsigned long long *>(dev_buf);
    return SSG_OK;
}
#else
int ssg_debug_launch_clock(ssg_handle *h, uint64_t *) { return fail(h, SSG_ERR_UNSUPPORTED, "ssg_debug_launch_clock: a -DSSG_STAMPS build uses the buffer for its own stamps"); }
#endif

#ifdef SSG_STAMPS
// diagnostic builds only: where per-wave s_memtime stamps go (16 u64 per wave)
int ssg_debug_set_stamp_buffer(ssg_handle *h, void *dev_buf)
{
    if (!h) return SSG_ERR_BAD_ARG;
    h->dev.dbg = static_cast<unsigned long long *>(dev_buf);
    return SSG_OK;
}
#endif

int ssg_generate_bank(ssg_handle *h, uint64_t seed, double width_frac, double *dev_bank, int n_maps, double *dev_raw,
                      void *stream)
{
    if (!h || !dev_bank || n_maps < 1 || !(width_frac > 0.0) || !(width_frac <= 1.0))
        return fail(h, SSG_ERR_BAD_ARG, "ssg_generate_bank: bad argument");
    hipError_t e = ssg::launch_generate_bank(seed, n_maps, h->cfg.n_goals, h->cfg.width, h->cfg.height, width_frac,
                                             h->cfg.spawn_x, h->cfg.spawn_y, dev_bank, dev_raw, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(h, SSG_ERR_HIP, std::string("generate_bank launch: ") + hipGetErrorString(e));
    if (dev_bank == h->bank) { h->dyn.bank_epoch++; h->dyn_queue_valid = false; memo_new_generation(h); } // regenerated in place
    return SSG_OK;
}

int ssg_refill_worlds(ssg_handle *h, uint64_t seed, double width_frac, double *dev_raw, void *stream)
{
    int rc = check_ready(h, true);
    if (rc != SSG_OK) return rc;
    if (h->cfg.map_ring <= 0) return fail(h, SSG_ERR_BAD_ARG, "ssg_refill_worlds: the handle was not created with map_ring");
    if (!(width_frac > 0.0) || !(width_frac <= 1.0)) return fail(h, SSG_ERR_BAD_ARG, "ssg_refill_worlds: bad width_frac");
    h->ring_seed = seed;
    h->ring_width_frac = width_frac;
    if (!h->zeroed) { // zero a freshly bound blob HERE, before the rings' counters are written (a later full ssg_reset must not wipe them)
        hipError_t e = hipMemsetAsync(h->state, 0, h->nbytes, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return fail(h, SSG_ERR_HIP, std::string("ssg_refill_worlds: ") + hipGetErrorString(e));
        h->zeroed = true;
    }
    rc = ring_refill(h, dev_raw, stream);
    if (rc == SSG_OK) h->ring_ready = true;
    return rc;
}

int ssg_dyn_invalidate(ssg_handle *h, const uint8_t *dev_mask, void *stream)
{
    int rc = check_ready(h, false);
    if (rc != SSG_OK) return rc;
    if (h->cfg.n_ships <= 1) return SSG_OK; // nothing to wake
    h->dyn_queue_valid = false;
    if (!dev_mask) { // "the blob may have been copied or restored": its memo tables may hold another bank's / handle's results under
        memo_new_generation(h);        // this handle's generation (the key names the record, not the bank's contents), so they are
        h->memo_clear_pending = true;  // emptied before the next launch that could read them
    }
    hipError_t e = ssg::launch_dyn_invalidate(h->dev, dev_mask, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(h, SSG_ERR_HIP, std::string("dyn_invalidate launch: ") + hipGetErrorString(e));
    return SSG_OK;
}

int ssg_render(ssg_handle *h, int env_index, int width, int height, uint8_t *dev_rgb, uint32_t flags, void *stream)
{
    int rc = check_ready(h, true);
    if (rc != SSG_OK) return rc;
    if (!dev_rgb || env_index < 0 || env_index >= h->cfg.n_envs || width < 1 || height < 1 || width > 8192 || height > 8192)
        return fail(h, SSG_ERR_BAD_ARG, "ssg_render: bad argument");
    hipError_t e = ssg::launch_render(h->dev, h->dyn, env_index, width, height, dev_rgb, flags, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(h, SSG_ERR_HIP, std::string("render launch: ") + hipGetErrorString(e));
    return SSG_OK;
}

int ssg_debug_kernel_times(ssg_handle *h, int enable, double *dyn_step_us, double *step_kernel_us, uint64_t *steps)
{
    if (!h) return SSG_ERR_BAD_ARG;
    if (dyn_step_us) *dyn_step_us = h->t_steps ? h->t_dyn_ms * 1e3 / (double)h->t_steps : 0.0;
    if (step_kernel_us) *step_kernel_us = h->t_steps ? h->t_step_ms * 1e3 / (double)h->t_steps : 0.0;
    if (steps) *steps = h->t_steps;
    h->time_kernels = enable != 0;
    h->t_dyn_ms = h->t_step_ms = 0.0;
    h->t_steps = 0;
    return SSG_OK;
}

int ssg_debug_dyn_counters(const ssg_handle *h, uint64_t *full_steps, uint64_t *queue_rebuilds)
{
    if (!h) return SSG_ERR_BAD_ARG;
    if (full_steps) *full_steps = h->dyn_seq;
    if (queue_rebuilds) *queue_rebuilds = h->n_classify;
    return SSG_OK;
}

int ssg_debug_launch_geometry(const ssg_handle *h, int *envs_per_workgroup, int *bank_in_lds, size_t *lds_bytes)
{
    if (!h) return SSG_ERR_BAD_ARG;
    if (envs_per_workgroup) *envs_per_workgroup = h->block;
    if (bank_in_lds) *bank_in_lds = h->lds ? 1 : 0;
    if (lds_bytes) *lds_bytes = h->lds_bytes;
    return SSG_OK;
}

int ssg_debug_copy8(const double *dev_src, double *dev_dst, size_t n_doubles, void *stream)
{
    if (!dev_src || !dev_dst) return SSG_ERR_BAD_ARG;
    return ssg::launch_calib_copy8(dev_src, dev_dst, n_doubles, static_cast<hipStream_t>(stream)) == hipSuccess ? SSG_OK : SSG_ERR_HIP;
}

int ssg_debug_clock_probe(uint64_t *dev_out, int n_blocks, int iters, void *stream)
{
    if (!dev_out || n_blocks < 1 || iters < 1) return SSG_ERR_BAD_ARG;
    return ssg::launch_clock_probe(reinterpret_cast<unsigned long long *>(dev_out), n_blocks, iters, static_cast<hipStream_t>(stream)) == hipSuccess
               ? SSG_OK : SSG_ERR_HIP;
}

// ---- host geometry ----
int ssg_host_convex_hull(int count, const double *verts_xy, double *out_xy, int *out_count)
{
    if (count < 1 || !verts_xy || !out_xy || !out_count) return SSG_ERR_BAD_ARG;
    std::vector<P2> v(count);
    for (int i = 0; i < count; ++i) v[i] = P2{verts_xy[2 * i], verts_xy[2 * i + 1]};
    std::vector<P2> h = convex_hull(v);
    for (size_t i = 0; i < h.size(); ++i) { out_xy[2 * i] = h[i].x; out_xy[2 * i + 1] = h[i].y; }
    *out_count = (int)h.size();
    return SSG_OK;
}

int ssg_host_moment_for_poly(double mass, int count, const double *verts_xy, double *out)
{
    if (count < 3 || !verts_xy || !out) return SSG_ERR_BAD_ARG;
    *out = moment_for_poly(mass, count, verts_xy);
    return SSG_OK;
}

int ssg_host_build_map(const double *left_xy, int n_left, const double *right_xy, int n_right, const double *goals_xy,
                       int n_goals, double spawn_x, double spawn_y, double *rec)
{
    if (!left_xy || !right_xy || !rec || n_left < 3 || n_right < 3 || n_goals < 0 || n_goals > SSG_MAX_GOALS ||
        (n_goals > 0 && !goals_xy))
        return SSG_ERR_BAD_ARG;
    std::fill(rec, rec + SSG_MAP_STRIDE, 0.0);
    const double *src[2] = {left_xy, right_xy};
    const int cnt[2] = {n_left, n_right};
    for (int s = 0; s < 2; ++s) {
        std::vector<P2> v(cnt[s]);
        for (int i = 0; i < cnt[s]; ++i) v[i] = P2{src[s][2 * i], src[s][2 * i + 1]};
        std::vector<P2> hull = convex_hull(v); // pm.Poly hulls its vertex list, models.py:180
        if (hull.size() < 3 || hull.size() > SSG_MAX_HULL) return SSG_ERR_UNSUPPORTED;
        std::vector<Plane> pl = planes_of(hull);
        rec[SSG_MAP_OFF_COUNTS + s] = (double)hull.size();
        double l = INFINITY, r = -INFINITY, b = INFINITY, t = -INFINITY; // cpPolyShapeCacheData on a static body
        for (const P2 &p : hull) {
            l = std::min(l, p.x); r = std::max(r, p.x);
            b = std::min(b, p.y); t = std::max(t, p.y);
        }
        double *bbp = rec + SSG_MAP_OFF_AABB + 4 * s;
        bbp[0] = l; bbp[1] = b; bbp[2] = r; bbp[3] = t;
        double *pp = rec + SSG_MAP_OFF_PLANES + s * (SSG_MAX_HULL * SSG_PLANE_DOUBLES);
        for (size_t i = 0; i < pl.size(); ++i) {
            double *q = pp + SSG_PLANE_DOUBLES * i;
            q[0] = pl[i].v0x; q[1] = pl[i].v0y; q[2] = pl[i].nx; q[3] = pl[i].ny;
            q[4] = pl[i].v0n;
        }
    }
    for (int g = 0; g < n_goals; ++g) {
        rec[SSG_MAP_OFF_GOALS + 2 * g] = goals_xy[2 * g];
        rec[SSG_MAP_OFF_GOALS + 2 * g + 1] = goals_xy[2 * g + 1];
    }
    // the reset observation's nearest goal (closest_goal, game.py:333-349: strict '<', first listed wins)
    double gx = -1.0, gy = -1.0, best = 0.0;
    for (int g = 0; g < n_goals; ++g) {
        const double dx = goals_xy[2 * g] - spawn_x, dy = goals_xy[2 * g + 1] - spawn_y;
        const double d = std::sqrt(dx * dx + dy * dy);
        if (g == 0 || d < best) { best = d; gx = goals_xy[2 * g]; gy = goals_xy[2 * g + 1]; }
    }
    rec[SSG_MAP_OFF_SPAWN_GOAL] = gx;
    rec[SSG_MAP_OFF_SPAWN_GOAL + 1] = gy;
    return SSG_OK;
}

int ssg_host_segment_query(const double *rec, int side, double ax, double ay, double bx, double by, double radius,
                           int *hit, double *px, double *py, double *alpha)
{
    if (!rec || side < 0 || side > 1 || !hit) return SSG_ERR_BAD_ARG;
    SegHit r = segment_query(hull_of_record(rec, side), ax, ay, bx, by, radius);
    *hit = r.hit ? 1 : 0;
    if (px) *px = r.px;
    if (py) *py = r.py;
    if (alpha) *alpha = r.alpha;
    return SSG_OK;
}

int ssg_host_goal_x_range(const double *rec, double width, double y, double *lo, double *hi, int *hit)
{
    if (!rec || !lo || !hi || !hit) return SSG_ERR_BAD_ARG;
    // game.py:322-325: fat (radius 10) rays from the mid-line to x=0 and x=W; [0] of each hit list; tolerance 60.
    const double tolerance = 60.0;
    const double ax = width / 2, ay = y;
    SegHit l{false, 0, 0, 1}, r{false, 0, 0, 1};
    for (int s = 0; s < 2 && !l.hit; ++s) l = segment_query(hull_of_record(rec, s), ax, ay, 0.0, y, 10.0);
    for (int s = 0; s < 2 && !r.hit; ++s) r = segment_query(hull_of_record(rec, s), ax, ay, width, y, 10.0);
    *hit = (l.hit && r.hit) ? 1 : 0;
    if (*hit) {
        *lo = l.px + tolerance;
        *hi = r.px - tolerance;
    }
    return SSG_OK;
}

} // extern "C"
