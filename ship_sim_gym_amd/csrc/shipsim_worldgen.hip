// shipsim_worldgen.hip — reset-time world generation ON THE DEVICE.
//
// One lane builds one complete map-bank record: river banks (game_map.gen_river_poly, game_map.py:22-73), the
// convex hulls and splitting planes pm.Poly derives (models.py:180), and the goal path (gen_goal_path,
// game.py:300-330) including its two radius-10 segment queries per goal.  The random draws come from a
// counter-based Philox4x32-10 stream keyed by (seed, map index), so this mode is NOT seed-compatible with the
// reference's Mersenne-Twister draws (random.gauss / random.randint / np.random.uniform) — it is a separate mode for
// refreshing the bank without the host (ShipVecEnv.regenerate_bank).  The GEOMETRY is the same arithmetic as the
// host path (shipsim_api.cpp): tests rebuild every record on the host from the raw polygons and draws this kernel
// emits and compare bit for bit.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "shipsim.h"
#include "shipsim_internal.h"

namespace ssg {

namespace {

struct Rng {
    uint32_t key[2];
    uint32_t ctr[4];
    uint32_t out[4];
    int have;

    __device__ void refill()
    {
        uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]}, k[2] = {key[0], key[1]};
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            if (r) { k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u; }
            const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t)p1;
            const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t)p0;
            c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        }
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
        have = 4;
        if (++ctr[2] == 0) ++ctr[3];
    }
    __device__ uint32_t u32()
    {
        if (!have) refill();
        --have; // (selected, not indexed: a dynamically indexed private array lives in scratch memory)
        return have == 3 ? out[3] : (have == 2 ? out[2] : (have == 1 ? out[1] : out[0]));
    }
    __device__ double uniform() // [0, 1) with 53 random bits
    {
        const uint64_t a = u32() >> 5, b = u32() >> 6;
        return (double)(a * 67108864ull + b) * (1.0 / 9007199254740992.0);
    }
    __device__ double gauss(double mu, double sigma) // Box-Muller
    {
        const double u1 = 1.0 - uniform(), u2 = uniform();
        return mu + sigma * (sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
    }
    __device__ int randint(int lo, int hi) { return lo + (int)(((uint64_t)u32() * (uint64_t)(hi - lo + 1)) >> 32); }
};

struct P2 { double x, y; };

__device__ __forceinline__ double cross3(P2 o, P2 a, P2 b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); }
__device__ __forceinline__ double clamp01(double f) { return fmax(0.0, fmin(f, 1.0)); }

// A world is built by ONE lane: with ~1 600 episodes starting per step the generator is THROUGHPUT-bound (a refill draws more
// worlds than the chip has lanes), so what counts is instructions per world.  The twelve polygon points live in REGISTERS (every
// loop over them is unrolled to the fixed bound SSG_MAX_HULL and predicated on the count, so every index is static); the hull —
// the monotone chain's stack, whose depth is data-dependent, and from round 5 on its planes — sits in per-lane LDS columns, so
// that the goal path's fat segment queries can visit only the edges and vertices a ray can touch, by index, instead of running
// twelve predicated bodies each (15.8 k of the kernel's 23 k static instructions were four inlined copies of those).
// (A first version kept points / hull in private arrays = scratch memory and read the planes back from the record in global
// memory: 130 us of its 277 us per refill went into hulling, 116 us into the goal path's twenty segment queries.)
constexpr int kHullStack = SSG_MAX_HULL + 2;
constexpr int kHullLdsDoubles = (2 * kHullStack + 3 * SSG_MAX_HULL) * 64; // vertex stack columns, then (nx, ny, v0.n) per plane

struct HullL { // one bank hull: vertex k in stk[(2k + comp) * 64 + lane], plane k in stk[(2 * kHullStack + 3k + f) * 64 + lane]
    int n;
    double l, b, r, t; // the cached AABB
};
__device__ __forceinline__ double hull_v(const double *stk, int lane, int k, int comp) { return stk[(2 * k + comp) * 64 + lane]; }
__device__ __forceinline__ double hull_p(const double *stk, int lane, int k, int f) { return stk[(2 * kHullStack + 3 * k + f) * 64 + lane]; }

// strict convex hull, CCW, first vertex = lexicographic (x, then y) minimum (same as the host's monotone chain), of the
// SSG_MAX_HULL points p[]; the hull's vertices end up in stk[(2*k + comp) * 64 + lane], k < returned count
__device__ __forceinline__ int convex_hull(P2 (&p)[SSG_MAX_HULL], double *stk, int lane)
{
    // sort by (x, y): odd-even transposition network (static indices; any correct sort gives the host's order)
#pragma unroll
    for (int pass = 0; pass < SSG_MAX_HULL; ++pass) {
#pragma unroll
        for (int i = pass & 1; i + 1 < SSG_MAX_HULL; i += 2) {
            const bool sw = (p[i].x > p[i + 1].x) | ((p[i].x == p[i + 1].x) & (p[i].y > p[i + 1].y));
            const P2 a = p[i], b = p[i + 1];
            p[i].x = sw ? b.x : a.x; p[i].y = sw ? b.y : a.y;
            p[i + 1].x = sw ? a.x : b.x; p[i + 1].y = sw ? a.y : b.y;
        }
    }
    bool keep[SSG_MAX_HULL]; // exact duplicates are dropped (the host does; practically never with random vertices)
    int n = 0, last = 0;     // number of distinct points, index of the last one
#pragma unroll
    for (int i = 0; i < SSG_MAX_HULL; ++i) {
        keep[i] = (i == 0) || (p[i].x != p[i - 1].x) || (p[i].y != p[i - 1].y);
        n += keep[i] ? 1 : 0;
        last = keep[i] ? i : last;
    }
    auto put = [&](int k, P2 v) { stk[(2 * k) * 64 + lane] = v.x; stk[(2 * k + 1) * 64 + lane] = v.y; };
    auto get = [&](int k) -> P2 { return P2{stk[(2 * k) * 64 + lane], stk[(2 * k + 1) * 64 + lane]}; };
    if (n <= 2) {
        int k = 0;
#pragma unroll
        for (int i = 0; i < SSG_MAX_HULL; ++i)
            if (keep[i]) put(k++, p[i]);
        return n;
    }
    // monotone chain; the two topmost entries are mirrored in registers (h1 = stack[k-1], h2 = stack[k-2])
    int k = 0;
    P2 h1 = P2{0, 0}, h2 = P2{0, 0};
    auto push = [&](P2 v, int floor_) {
        while (k >= floor_ && cross3(h2, h1, v) <= 0.0) { // pop
            --k;
            h1 = h2;
            if (k >= 2) h2 = get(k - 2);
        }
        put(k++, v);
        h2 = h1; h1 = v;
    };
#pragma unroll
    for (int i = 0; i < SSG_MAX_HULL; ++i) // lower chain
        if (keep[i]) push(p[i], 2);
    const int t = k + 1;
#pragma unroll
    for (int i = SSG_MAX_HULL - 1; i >= 0; --i) // upper chain: from the second-to-last distinct point down to the first
        if (keep[i] && i != last) push(p[i], t);
    return k - 1;
}

// cpPolyShapePointQuery distance (negative inside) — only reached when a ray starts within r2 of the hull's box (rare)
__device__ __forceinline__ double point_query(const HullL &h, const double *stk, const int lane, double px, double py)
{
    double v0x = hull_v(stk, lane, h.n - 1, 0), v0y = hull_v(stk, lane, h.n - 1, 1); // the last vertex
    double best = INFINITY;
    bool outside = false;
    for (int i = 0; i < h.n; ++i) {
        const double v1x = hull_v(stk, lane, i, 0), v1y = hull_v(stk, lane, i, 1);
        outside = outside || ((hull_p(stk, lane, i, 0) * (px - v1x) + hull_p(stk, lane, i, 1) * (py - v1y)) > 0.0);
        const double dx = v0x - v1x, dy = v0y - v1y;
        const double t = clamp01((dx * (px - v1x) + dy * (py - v1y)) / (dx * dx + dy * dy));
        const double qx = v1x + dx * t, qy = v1y + dy * t;
        const double ex = px - qx, ey = py - qy;
        const double d = sqrt(ex * ex + ey * ey);
        if (d < best) best = d;
        v0x = v1x; v0y = v1y;
    }
    return outside ? best : -best;
}

// cpShapeSegmentQuery with query radius r2 against one hull: returns hit, reported point x (all gen_goal_path uses).
// Every hit of the fat segment is a crossing of an edge offset by r2 between the edge's end points, or a pass within r2 of a
// vertex: an edge (a vertex) whose y-extent widened by r2 does not meet the segment's cannot be hit, so each lane walks only ITS
// candidates, in index order — the order matters: a later edge's hit replaces an earlier one's — with the same arithmetic as the
// loop over all twelve.
__device__ __forceinline__ bool segment_query_x(const HullL &h, const double *stk, const int lane, double ax, double ay, double bx, double by,
                                                double r2, double &outx)
{
    double alpha = 1.0;
    bool hit = false;
    outx = bx;
    // The fat segment cannot touch a hull whose box it does not reach: the left-going ray against the right bank, the right-going
    // one against the left bank — half of the goal path's queries — end here, with the same "no hit" the full query would report.
    if (fmax(ax, bx) + r2 < h.l || fmin(ax, bx) - r2 > h.r || fmax(ay, by) + r2 < h.b || fmin(ay, by) - r2 > h.t) return false;
    // (the start point's distance to the hull is at least its distance to the hull's box: beyond r2 of the box the point
    // query — a division and a square root per edge — cannot report a hit at alpha 0)
    const double bx_ = fmax(fmax(h.l - ax, ax - h.r), 0.0), by_ = fmax(fmax(h.b - ay, ay - h.t), 0.0);
    if (bx_ * bx_ + by_ * by_ <= r2 * r2 * 1.0000001 && point_query(h, stk, lane, ax, ay) <= r2) return true; // reported point stays the far end
    const double ylo = fmin(ay, by) - r2 - 1e-6, yhi = fmax(ay, by) + r2 + 1e-6; // (1e-6: far above the rounding of a hit point's y)
    unsigned ce = 0u, cv = 0u; // candidate edges (edge i runs from vertex i-1 to vertex i) / vertices
    {
        double pvy = hull_v(stk, lane, h.n - 1, 1);
#pragma unroll
        for (int i = 0; i < SSG_MAX_HULL; ++i) {
            const double vy = hull_v(stk, lane, (i < h.n) ? i : 0, 1);
            const bool in = i < h.n;
            ce |= (in && fmin(pvy, vy) <= yhi && ylo <= fmax(pvy, vy)) ? (1u << i) : 0u;
            cv |= (in && vy <= yhi && ylo <= vy) ? (1u << i) : 0u;
            pvy = in ? vy : pvy;
        }
    }
    while (__any(ce != 0u)) {
        const bool on = ce != 0u;
        const int i = on ? (__ffs((int)ce) - 1) : 0;
        ce &= ce - 1u;
        const int ip = (i == 0) ? h.n - 1 : i - 1;
        const double vx = hull_v(stk, lane, i, 0), vy = hull_v(stk, lane, i, 1);
        const double pvx = hull_v(stk, lane, ip, 0), pvy = hull_v(stk, lane, ip, 1);
        const double nx = hull_p(stk, lane, i, 0), ny = hull_p(stk, lane, i, 1), hd = hull_p(stk, lane, i, 2);
        const double an = ax * nx + ay * ny;
        const double d = an - hd - r2;
        const double bn = bx * nx + by * ny;
        const double t = d / fmax(an - bn, DBL_MIN);
        const double omt = 1.0 - t;
        const double ptx = ax * omt + bx * t, pty = ay * omt + by * t;
        const double dtv = nx * pty - ny * ptx;
        const double dtmin = nx * pvy - ny * pvx, dtmax = nx * vy - ny * vx; // cpvcross(n, v[i-1]), cpvcross(n, v[i])
        if (on && !(d < 0.0) && !(t < 0.0 || 1.0 < t) && dtmin <= dtv && dtv <= dtmax) {
            hit = true;
            outx = ptx - nx * r2;
            alpha = t;
        }
    }
    if (r2 > 0.0) {
        while (__any(cv != 0u)) {
            const bool on = cv != 0u;
            const int i = on ? (__ffs((int)cv) - 1) : 0;
            cv &= cv - 1u;
            const double cx = hull_v(stk, lane, i, 0), cy = hull_v(stk, lane, i, 1);
            const double dax = ax - cx, day = ay - cy, dbx = bx - cx, dby = by - cy;
            const double daa = dax * dax + day * day, dab = dax * dbx + day * dby, dbb = dbx * dbx + dby * dby;
            const double qa = daa - 2.0 * dab + dbb;
            const double qb = dab - daa;
            const double det = qb * qb - qa * (daa - r2 * r2);
            if (on && det >= 0.0) {
                const double t = (-qb - sqrt(det)) / qa;
                if (0.0 <= t && t <= 1.0 && t < alpha) {
                    const double omt = 1.0 - t;
                    double nx = dax * omt + dbx * t, ny = day * omt + dby * t;
                    const double inv = 1.0 / (sqrt(nx * nx + ny * ny) + DBL_MIN);
                    nx *= inv;
                    hit = true;
                    outx = (ax * omt + bx * t) - nx * r2;
                    alpha = t;
                }
            }
        }
    }
    return hit;
}

// One world: ShipGame.reset's gen_level + gen_goal_path (game.py:60-71,300-330) into the record `rec`, from the stream
// `rng`.  rw (nullable): [2][12][2] polygon vertices, then per goal (y, u, fallback_x) = 3 doubles -> 48 + 3*n_goals.
// stk: kHullLdsDoubles doubles of LDS (per-lane columns); lane = this thread's lane (one wave per workgroup).
__device__ __forceinline__ void generate_world(Rng &rng, int n_goals, double width, double height, double width_frac, double spawn_x,
                                               double spawn_y, double *__restrict__ rec, double *__restrict__ rw, double *stk, int lane)
{
    // (every double of the record is written exactly once below — the unused plane and goal slots as zeros where their used
    // neighbours are written; a zeroing sweep over all 145 first was 145 scattered stores per lane in front of everything)
    static_assert(SSG_MAP_OFF_PLANES + 2 * SSG_MAX_HULL * SSG_PLANE_DOUBLES + 1 == SSG_MAP_STRIDE && SSG_MAP_OFF_SPAWN_GOAL + 2 == SSG_MAP_OFF_PLANES &&
                  SSG_MAP_OFF_GOALS + 2 * SSG_MAX_GOALS == SSG_MAP_OFF_SPAWN_GOAL && SSG_MAP_OFF_AABB + 8 == SSG_MAP_OFF_GOALS && SSG_MAP_OFF_COUNTS + 2 == SSG_MAP_OFF_AABB,
                  "record layout: counts, boxes, goals, spawn goal, planes, one pad double");
    rec[SSG_MAP_STRIDE - 1] = 0.0;

    // ---- gen_river_poly (game_map.py:22-73) ----
    constexpr int N = 10;
    static_assert(N + 2 == SSG_MAX_HULL, "ten jittered points + two corners");
    const double y_start = -100.0;
    const double y_delta = (height * 1.2 - y_start) / N;
    const double bank_width = width_frac * width / 2;
    // Only ONE hull's planes are live at a time (two were 128 doubles = every VGPR of the lane, and the kernel ran on 241
    // spilled registers): all random draws are taken first, in the stream's order — side 0's vertices, side 1's, the goals' —
    // then side 0 is hulled and every goal's two rays are cast against it, then side 1.  gen_goal_path asks the left bank
    // first and the right one only on a miss; asking them in two passes returns the same answers (the queries are pure).
    auto draw_side = [&](const int s, P2 (&pts)[SSG_MAX_HULL]) {
        const double x_min = s ? width - bank_width : 0.0, x_max = s ? width : bank_width;
        const double centre = x_min + (x_max - x_min); // the reference's x_middle is x_max (game_map.py:48)
#pragma unroll
        for (int k = 0; k < SSG_MAX_HULL; ++k) pts[k] = P2{0.0, 0.0};
        // gen_river_poly draws, per vertex, x ~ gauss(centre, 50) until it falls inside the bank's strip [x_min, x_max] (at most
        // 1000 tries, game_map.py:52-60) and y ~ y_start + gauss(y_delta * i, 20).  The reference's centre IS x_max
        // (game_map.py:48), so the accepted x is a normal truncated to [centre - strip, centre]: drawn here as the REFLECTED
        // half-normal centre - |50 z| (same conditional law; rejected only beyond the strip's 3 sigma: 0.27 % instead of the
        // 50.1 % of the two-sided draw), and one Box-Muller transform yields both z (its cosine branch) and the y deviate (its
        // sine branch) — independent standard normals.  ~1 transform per vertex instead of ~4 transcendental gauss() calls;
        // same distribution, a different (Philox, not Mersenne-Twister anyway) stream.
        // (ONE rolled loop over candidates in which every lane advances its own vertex index; the accepted vertex goes to
        // its register by a static select sweep, so that pts[] is never indexed dynamically.)
        for (int i = 1, tries = 0; i <= N;) {
            const double u1 = 1.0 - rng.uniform(), u2 = rng.uniform();
            const double rad = sqrt(-2.0 * log(u1));
            double sn, cs;
            sincospi(2.0 * u2, &sn, &cs); // (exact argument reduction: no Payne-Hanek machinery on the chain.  The transform in single
            // precision on the hardware's v_log_f32 / v_sin_f32 / v_cos_f32 was measured: 223 against 228 us per refill — not kept)
            const double x = centre - fabs(50.0 * (rad * cs));
            const double y = y_start + (y_delta * i + 20.0 * (rad * sn));
            ++tries;
            if (!((x < x_min || x > x_max) && tries < 1000)) {
#pragma unroll
                for (int k = 0; k < N; ++k) { pts[k].x = (k == i - 1) ? x : pts[k].x; pts[k].y = (k == i - 1) ? y : pts[k].y; }
                ++i;
                tries = 0;
            }
        }
        pts[N] = P2{s ? width : 0.0, height};
        pts[N + 1] = P2{s ? width : 0.0, 0.0};
        if (rw) {
#pragma unroll
            for (int i = 0; i < SSG_MAX_HULL; ++i) { rw[s * 24 + 2 * i] = pts[i].x; rw[s * 24 + 2 * i + 1] = pts[i].y; }
        }
    };
    // ---- pm.Poly: hull + splitting planes + cached AABB (models.py:180) ----
    auto build_hull = [&](const int s, P2 (&pts)[SSG_MAX_HULL], HullL &h) {
        const int n = convex_hull(pts, stk, lane);
        h.n = n;
        rec[SSG_MAP_OFF_COUNTS + s] = (double)n;
        double l = INFINITY, r = -INFINITY, b = INFINITY, t = -INFINITY;
        double *pp = rec + SSG_MAP_OFF_PLANES + s * (SSG_MAX_HULL * SSG_PLANE_DOUBLES);
        P2 a = P2{stk[(2 * (n - 1)) * 64 + lane], stk[(2 * (n - 1) + 1) * 64 + lane]}; // hull[n-1]
#pragma unroll
        for (int i = 0; i < SSG_MAX_HULL; ++i) {
            double *q = pp + SSG_PLANE_DOUBLES * i;
            if (i < n) {
                const P2 bb = P2{stk[(2 * i) * 64 + lane], stk[(2 * i + 1) * 64 + lane]};
                const double ex = bb.x - a.x, ey = bb.y - a.y;
                const double rx = ey, ry = -ex;
                const double inv = 1.0 / (sqrt(rx * rx + ry * ry) + DBL_MIN);
                const double nx = rx * inv, ny = ry * inv;
                const double hd = bb.x * nx + bb.y * ny;
                q[0] = bb.x; q[1] = bb.y; q[2] = nx; q[3] = ny; q[4] = hd;
                stk[(2 * kHullStack + 3 * i + 0) * 64 + lane] = nx; // the plane's LDS columns (the queries below read them by index)
                stk[(2 * kHullStack + 3 * i + 1) * 64 + lane] = ny;
                stk[(2 * kHullStack + 3 * i + 2) * 64 + lane] = hd;
                l = fmin(l, bb.x); r = fmax(r, bb.x); b = fmin(b, bb.y); t = fmax(t, bb.y);
                a = bb;
            } else {
                q[0] = 0.0; q[1] = 0.0; q[2] = 0.0; q[3] = 0.0; q[4] = 0.0;
            }
        }
        h.l = l; h.b = b; h.r = r; h.t = t;
        double *bbp = rec + SSG_MAP_OFF_AABB + 4 * s;
        bbp[0] = l; bbp[1] = b; bbp[2] = r; bbp[3] = t;
    };
    P2 pts1[SSG_MAX_HULL];
    HullL h;
    {
        P2 pts0[SSG_MAX_HULL];
        draw_side(0, pts0);
        draw_side(1, pts1);
#if defined(SSG_GEN_STOP) && SSG_GEN_STOP == 1 /* timing-only development builds */
        rec[SSG_MAP_OFF_PLANES] = pts0[3].x + pts0[7].y + pts1[3].x + pts1[7].y;
        return;
#endif
        // ---- gen_goal_path's draws (game.py:300-330), before any geometry: see above ----
        // (per-goal values sit in registers; the loops over goals stay ROLLED — unrolled, their twenty-four inlined segment
        // queries would be four times the instruction cache — and reach them through static select sweeps)
        build_hull(0, pts0, h);
    }
    const double gy_delta = height / (n_goals + 1), x_middle = width / 2;
    double gy[SSG_MAX_GOALS], gu[SSG_MAX_GOALS], gfb[SSG_MAX_GOALS], glx[SSG_MAX_GOALS], grx[SSG_MAX_GOALS];
    unsigned ghit = 0u; // bit 2i: the left ray of goal i has its hit, bit 2i + 1: the right one
#pragma unroll
    for (int k = 0; k < SSG_MAX_GOALS; ++k) gy[k] = gu[k] = gfb[k] = glx[k] = grx[k] = 0.0;
    auto put = [](double (&arr)[SSG_MAX_GOALS], int i, double v) {
#pragma unroll
        for (int k = 0; k < SSG_MAX_GOALS; ++k) arr[k] = (k == i) ? v : arr[k];
    };
    auto at = [](const double (&arr)[SSG_MAX_GOALS], int i) -> double {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < SSG_MAX_GOALS; ++k) v = (k == i) ? arr[k] : v;
        return v;
    };
    for (int i = 1; i <= n_goals; ++i) { // (the stream's order: y, u, fallback per goal — after both sides' vertices)
        put(gy, i - 1, gy_delta * i + rng.randint(-20, 20));
        put(gu, i - 1, rng.uniform());
        put(gfb, i - 1, x_middle * i + rng.randint(-50, 50));
    }
#if defined(SSG_GEN_STOP) && SSG_GEN_STOP <= 2
    build_hull(1, pts1, h);
    return;
#endif
    // ---- pass 1: both rays of every goal against the LEFT bank ----
    for (int i = 0; i < n_goals; ++i) {
        const double y = at(gy, i);
        double lx, rx2;
        const bool lh = segment_query_x(h, stk, lane, x_middle, y, 0.0, y, 10.0, lx);
        const bool rh = segment_query_x(h, stk, lane, x_middle, y, width, y, 10.0, rx2);
        put(glx, i, lx); put(grx, i, rx2);
        ghit |= (lh ? 1u : 0u) << (2 * i) | (rh ? 1u : 0u) << (2 * i + 1);
    }
    build_hull(1, pts1, h);
    // ---- pass 2: the rays that missed, against the RIGHT bank; the goal itself ----
    double best = 0.0, sgx = -1.0, sgy = -1.0;
    for (int i = 0; i < n_goals; ++i) {
        const double y = at(gy, i), u = at(gu, i), fallback = at(gfb, i);
        double lx = at(glx, i), rx2 = at(grx, i);
        bool lh = (ghit >> (2 * i)) & 1u, rh = (ghit >> (2 * i + 1)) & 1u;
        if (!lh) lh = segment_query_x(h, stk, lane, x_middle, y, 0.0, y, 10.0, lx);
        if (!rh) rh = segment_query_x(h, stk, lane, x_middle, y, width, y, 10.0, rx2);
        double x;
        if (lh && rh) {
            const double lo = lx + 60.0, hi = rx2 - 60.0;
            x = lo + (hi - lo) * u; // np.random.uniform(lo, hi) = lo + (hi - lo) * random_sample()
        } else {
            x = fallback;
        }
        rec[SSG_MAP_OFF_GOALS + 2 * i] = x;
        rec[SSG_MAP_OFF_GOALS + 2 * i + 1] = y;
        if (rw) { rw[48 + 3 * i] = y; rw[48 + 3 * i + 1] = u; rw[48 + 3 * i + 2] = fallback; }
        const double dx = x - spawn_x, dy = y - spawn_y;
        const double d = sqrt(dx * dx + dy * dy);
        if (i == 0 || d < best) { best = d; sgx = x; sgy = y; }
    }
    for (int i = n_goals; i < SSG_MAX_GOALS; ++i) { rec[SSG_MAP_OFF_GOALS + 2 * i] = 0.0; rec[SSG_MAP_OFF_GOALS + 2 * i + 1] = 0.0; }
    rec[SSG_MAP_OFF_SPAWN_GOAL] = sgx;
    rec[SSG_MAP_OFF_SPAWN_GOAL + 1] = sgy;
}

} // namespace

// raw: per map 48 + 3*n_goals doubles (generate_world)
__global__ __launch_bounds__(64) void generate_bank_kernel(uint64_t seed, int n_maps, int n_goals, double width, double height,
                                     double width_frac, double spawn_x, double spawn_y, double *__restrict__ bank,
                                     double *__restrict__ raw)
{
    __shared__ double stk[kHullLdsDoubles];
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_maps) return;
    Rng rng;
    rng.key[0] = (uint32_t)seed; rng.key[1] = (uint32_t)(seed >> 32);
    rng.ctr[0] = (uint32_t)m; rng.ctr[1] = 0x57474e00u /* "WGN" domain tag */; rng.ctr[2] = 0; rng.ctr[3] = 0;
    rng.have = 0;
    generate_world(rng, n_goals, width, height, width_frac, spawn_x, spawn_y, bank + (size_t)m * SSG_MAP_STRIDE,
                   raw ? raw + (size_t)m * (48 + 3 * n_goals) : nullptr, stk, (int)threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// map_ring mode (ssg_config.map_ring = R): every episode of every env gets a brand-new world, as ShipGame.reset draws one
// at every reset (game.py:260-277).  Episode p of env e lives in bank record e*R + p mod R.  Pass 1 finds, per env, the
// episodes not yet drawn up to (current episode + R - 1) and queues them (wave-aggregated: one atomic per wave); pass 2
// draws one world per lane, DENSELY (a world is a ~50 k-instruction sequential chain: scattered over the envs' lanes it
// would cost every wave the whole chain for the one or two lanes that need it).
// ---------------------------------------------------------------------------------------------------------
__global__ void refill_scan_kernel(const DevCfg c, unsigned long long *__restrict__ queue, unsigned *__restrict__ count)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t np = (size_t)c.n_pad;
    int need = 0, gen = 0;
    if (e < c.n_envs) {
        const int started = c.i32cols[(size_t)ICOL_EPISODE * np + e];
        gen = c.i32cols[(size_t)ICOL_GEN * np + e];
        const int cur = started > 0 ? started - 1 : 0;
        need = cur + c.map_ring - gen;
        need = need < 0 ? 0 : (need > c.map_ring ? c.map_ring : need);
        if (need) c.i32cols[(size_t)ICOL_GEN * np + e] = gen + need;
    }
    // exclusive prefix of `need` over the wave, one atomic per wave
    const int lane = threadIdx.x & 63;
    int incl = need;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        incl += (lane >= o) ? v : 0;
    }
    const int total = __shfl(incl, 63);
    unsigned base = 0;
    if (total && lane == 0) base = atomicAdd(count, (unsigned)total);
    base = __shfl(base, 0) + (unsigned)(incl - need);
    for (int i = 0; i < need; ++i) queue[base + i] = ((unsigned long long)(unsigned)e << 32) | (unsigned)(gen + i);
}

__global__ __launch_bounds__(64) void refill_gen_kernel(const DevCfg c, uint64_t seed, double width_frac, const unsigned long long *__restrict__ queue,
                                  const unsigned *__restrict__ count, double *__restrict__ bank, double *__restrict__ raw)
{
    __shared__ double stk[kHullLdsDoubles];
    const unsigned n = min(*count, (unsigned)c.n_envs * (unsigned)c.map_ring);
    // (a shared hand-out cursor instead of the grid-stride loop — a deep ring asks for several worlds per lane of the chip — was
    // measured: 239 against 228 us per refill of ~208 k worlds; the dispatcher already hands a finished wave's SIMD to the next
    // workgroup)
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long it = queue[i];
        const int e = (int)(it >> 32);
        const unsigned episode = (unsigned)it;
        const unsigned long long gid = (unsigned long long)(c.env_id_base + (long long)e);
        Rng rng;
        rng.key[0] = (uint32_t)seed; rng.key[1] = (uint32_t)(seed >> 32);
        rng.ctr[0] = (uint32_t)gid; rng.ctr[1] = 0x52494e47u /* "RING" domain tag */ ^ (uint32_t)(gid >> 32);
        rng.ctr[2] = 0; rng.ctr[3] = episode;
        rng.have = 0;
        const size_t slot = (size_t)e * (size_t)c.map_ring + (size_t)(episode % (unsigned)c.map_ring);
        generate_world(rng, c.n_goals, c.width, c.height, width_frac, c.spawn_x, c.spawn_y, bank + slot * SSG_MAP_STRIDE,
                       raw ? raw + slot * (size_t)(48 + 3 * c.n_goals) : nullptr, stk, (int)threadIdx.x);
    }
}

hipError_t launch_refill_worlds(const DevCfg &c, uint64_t seed, double width_frac, unsigned long long *queue, unsigned *count,
                                double *bank, double *raw, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(count, 0, sizeof(unsigned), stream);
    if (e != hipSuccess) return e;
    const int block = 256;
    hipLaunchKernelGGL(refill_scan_kernel, dim3((unsigned)((c.n_envs + block - 1) / block)), dim3(block), 0, stream, c, queue, count);
    // one wave per workgroup: a world's chain is long and sequential, spread the (few hundred) waves over all SIMDs
    const long long max_items = (long long)c.n_envs * c.map_ring;
    const long long wg = (max_items + 63) / 64;
    hipLaunchKernelGGL(refill_gen_kernel, dim3((unsigned)(wg < 2048 ? wg : 2048)), dim3(64), 0, stream, c, seed, width_frac, queue,
                       count, bank, raw);
    return hipGetLastError();
}

hipError_t launch_generate_bank(uint64_t seed, int n_maps, int n_goals, double width, double height, double width_frac,
                                double spawn_x, double spawn_y, double *bank, double *raw, hipStream_t stream)
{
    const int block = 64, grid = (n_maps + block - 1) / block;
    hipLaunchKernelGGL(generate_bank_kernel, dim3(grid), dim3(block), 0, stream, seed, n_maps, n_goals, width, height,
                       width_frac, spawn_x, spawn_y, bank, raw);
    return hipGetLastError();
}

} // namespace ssg
