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
        return out[--have];
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

__device__ double cross3(P2 o, P2 a, P2 b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); }

// strict convex hull, CCW, first vertex = lexicographic minimum (same as the host's monotone chain)
__device__ int convex_hull(P2 *pts, int n, P2 *hull)
{
    for (int i = 1; i < n; ++i) { // insertion sort by (x, y)
        P2 v = pts[i];
        int j = i - 1;
        while (j >= 0 && (pts[j].x > v.x || (pts[j].x == v.x && pts[j].y > v.y))) { pts[j + 1] = pts[j]; --j; }
        pts[j + 1] = v;
    }
    int m = 0; // drop exact duplicates
    for (int i = 0; i < n; ++i)
        if (i == 0 || pts[i].x != pts[m - 1].x || pts[i].y != pts[m - 1].y) pts[m++] = pts[i];
    n = m;
    if (n <= 2) {
        for (int i = 0; i < n; ++i) hull[i] = pts[i];
        return n;
    }
    int k = 0;
    for (int i = 0; i < n; ++i) {
        while (k >= 2 && cross3(hull[k - 2], hull[k - 1], pts[i]) <= 0.0) --k;
        hull[k++] = pts[i];
    }
    for (int i = n - 2, t = k + 1; i >= 0; --i) {
        while (k >= t && cross3(hull[k - 2], hull[k - 1], pts[i]) <= 0.0) --k;
        hull[k++] = pts[i];
    }
    return k - 1;
}

__device__ double clamp01(double f) { return fmax(0.0, fmin(f, 1.0)); }

struct Hull {
    int n;
    const double *pl; // n planes of SSG_PLANE_DOUBLES doubles inside the record being built
    double l, b, r, t; // the hull's AABB
};

__device__ double point_query(const Hull &h, double px, double py)
{
    double v0x = h.pl[SSG_PLANE_DOUBLES * (h.n - 1) + 0], v0y = h.pl[SSG_PLANE_DOUBLES * (h.n - 1) + 1];
    double best = INFINITY;
    bool outside = false;
    for (int i = 0; i < h.n; ++i) {
        const double *p = h.pl + SSG_PLANE_DOUBLES * i;
        const double v1x = p[0], v1y = p[1];
        outside = outside || ((p[2] * (px - v1x) + p[3] * (py - v1y)) > 0.0);
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

// cpShapeSegmentQuery with query radius r2 against one hull: returns hit, reported point x (all gen_goal_path uses)
__device__ bool segment_query_x(const Hull &h, double ax, double ay, double bx, double by, double r2, double &outx)
{
    double alpha = 1.0;
    bool hit = false;
    outx = bx;
    // (the start point's distance to the hull is at least its distance to the hull's box: beyond r2 of the box the point
    // query — a division and a square root per edge — cannot report a hit at alpha 0)
    const double bx_ = fmax(fmax(h.l - ax, ax - h.r), 0.0), by_ = fmax(fmax(h.b - ay, ay - h.t), 0.0);
    if (bx_ * bx_ + by_ * by_ <= r2 * r2 * 1.0000001 && point_query(h, ax, ay) <= r2) return true; // reported point stays the far end
    for (int i = 0; i < h.n; ++i) {
        const double *p = h.pl + SSG_PLANE_DOUBLES * i;
        const double nx = p[2], ny = p[3];
        const double an = ax * nx + ay * ny;
        const double d = an - p[4] - r2;
        if (d < 0.0) continue;
        const double bn = bx * nx + by * ny;
        const double t = d / fmax(an - bn, DBL_MIN);
        if (t < 0.0 || 1.0 < t) continue;
        const double omt = 1.0 - t;
        const double ptx = ax * omt + bx * t, pty = ay * omt + by * t;
        const double dtv = nx * pty - ny * ptx;
        const double *pp = h.pl + SSG_PLANE_DOUBLES * ((i - 1 + h.n) % h.n);       // the edge's start vertex v[i-1]
        const double dtmin = nx * pp[1] - ny * pp[0], dtmax = nx * p[1] - ny * p[0]; // cpvcross(n, v[i-1]), cpvcross(n, v[i])
        if (dtmin <= dtv && dtv <= dtmax) {
            hit = true;
            outx = ptx - nx * r2;
            alpha = t;
        }
    }
    if (r2 > 0.0) {
        for (int i = 0; i < h.n; ++i) {
            const double *p = h.pl + SSG_PLANE_DOUBLES * i;
            const double cx = p[0], cy = p[1];
            const double dax = ax - cx, day = ay - cy, dbx = bx - cx, dby = by - cy;
            const double daa = dax * dax + day * day, dab = dax * dbx + day * dby, dbb = dbx * dbx + dby * dby;
            const double qa = daa - 2.0 * dab + dbb;
            const double qb = dab - daa;
            const double det = qb * qb - qa * (daa - r2 * r2);
            if (det >= 0.0) {
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
__device__ void generate_world(Rng &rng, int n_goals, double width, double height, double width_frac, double spawn_x,
                               double spawn_y, double *__restrict__ rec, double *__restrict__ rw)
{
    for (int i = 0; i < SSG_MAP_STRIDE; ++i) rec[i] = 0.0;

    // ---- gen_river_poly (game_map.py:22-73) ----
    const int N = 10;
    const double y_start = -100.0;
    const double y_delta = (height * 1.2 - y_start) / N;
    const double bank_width = width_frac * width / 2;
    for (int s = 0; s < 2; ++s) {
        const double x_min = s ? width - bank_width : 0.0, x_max = s ? width : bank_width;
        const double centre = x_min + (x_max - x_min); // the reference's x_middle is x_max (game_map.py:48)
        P2 pts[SSG_MAX_HULL], hull[2 * SSG_MAX_HULL];
        // gen_river_poly draws, per vertex, x ~ gauss(centre, 50) until it falls inside the bank's strip [x_min, x_max] (at most
        // 1000 tries, game_map.py:52-60) and y ~ y_start + gauss(y_delta * i, 20).  The reference's centre IS x_max
        // (game_map.py:48), so the accepted x is a normal truncated to [centre - strip, centre]: drawn here as the REFLECTED
        // half-normal centre - |50 z| (same conditional law; rejected only beyond the strip's 3 sigma: 0.27 % instead of the
        // 50.1 % of the two-sided draw), and one Box-Muller transform yields both z (its cosine branch) and the y deviate (its
        // sine branch) — independent standard normals.  ~1 transform per vertex instead of ~4 transcendental gauss() calls;
        // same distribution, a different (Philox, not Mersenne-Twister anyway) stream.
        for (int i = 1, tries = 0; i <= N;) {
            const double u1 = 1.0 - rng.uniform(), u2 = rng.uniform();
            const double rad = sqrt(-2.0 * log(u1));
            double sn, cs;
            sincos(6.283185307179586 * u2, &sn, &cs);
            const double x = centre - fabs(50.0 * (rad * cs));
            const double y = y_start + (y_delta * i + 20.0 * (rad * sn));
            ++tries;
            if (!((x < x_min || x > x_max) && tries < 1000)) {
                pts[i - 1] = P2{x, y};
                ++i;
                tries = 0;
            }
        }
        pts[N] = P2{s ? width : 0.0, height};
        pts[N + 1] = P2{s ? width : 0.0, 0.0};
        if (rw)
            for (int i = 0; i < SSG_MAX_HULL; ++i) { rw[s * 24 + 2 * i] = pts[i].x; rw[s * 24 + 2 * i + 1] = pts[i].y; }
        // ---- pm.Poly: hull + splitting planes + cached AABB (models.py:180) ----
        const int n = convex_hull(pts, N + 2, hull);
        rec[SSG_MAP_OFF_COUNTS + s] = (double)n;
        double l = INFINITY, r = -INFINITY, b = INFINITY, t = -INFINITY;
        double *pp = rec + SSG_MAP_OFF_PLANES + s * (SSG_MAX_HULL * SSG_PLANE_DOUBLES);
        for (int i = 0; i < n; ++i) {
            const P2 a = hull[(i - 1 + n) % n], bb = hull[i];
            const double ex = bb.x - a.x, ey = bb.y - a.y;
            const double rx = ey, ry = -ex;
            const double inv = 1.0 / (sqrt(rx * rx + ry * ry) + DBL_MIN);
            double *q = pp + SSG_PLANE_DOUBLES * i;
            q[0] = bb.x; q[1] = bb.y; q[2] = rx * inv; q[3] = ry * inv;
            q[4] = q[0] * q[2] + q[1] * q[3];
            l = fmin(l, bb.x); r = fmax(r, bb.x); b = fmin(b, bb.y); t = fmax(t, bb.y);
        }
        double *bbp = rec + SSG_MAP_OFF_AABB + 4 * s;
        bbp[0] = l; bbp[1] = b; bbp[2] = r; bbp[3] = t;
    }
    // ---- gen_goal_path (game.py:300-330) ----
    const Hull hl{(int)rec[SSG_MAP_OFF_COUNTS + 0], rec + SSG_MAP_OFF_PLANES, rec[SSG_MAP_OFF_AABB + 0], rec[SSG_MAP_OFF_AABB + 1],
                  rec[SSG_MAP_OFF_AABB + 2], rec[SSG_MAP_OFF_AABB + 3]};
    const Hull hr{(int)rec[SSG_MAP_OFF_COUNTS + 1], rec + SSG_MAP_OFF_PLANES + SSG_MAX_HULL * SSG_PLANE_DOUBLES, rec[SSG_MAP_OFF_AABB + 4],
                  rec[SSG_MAP_OFF_AABB + 5], rec[SSG_MAP_OFF_AABB + 6], rec[SSG_MAP_OFF_AABB + 7]};
    const double gy_delta = height / (n_goals + 1), x_middle = width / 2;
    double best = 0.0, sgx = -1.0, sgy = -1.0;
    for (int i = 1; i <= n_goals; ++i) {
        const double y = gy_delta * i + rng.randint(-20, 20);
        const double u = rng.uniform();
        const double fallback = x_middle * i + rng.randint(-50, 50);
        double lx, rx2;
        bool lh = segment_query_x(hl, x_middle, y, 0.0, y, 10.0, lx);
        if (!lh) lh = segment_query_x(hr, x_middle, y, 0.0, y, 10.0, lx);
        bool rh = segment_query_x(hl, x_middle, y, width, y, 10.0, rx2);
        if (!rh) rh = segment_query_x(hr, x_middle, y, width, y, 10.0, rx2);
        double x;
        if (lh && rh) {
            const double lo = lx + 60.0, hi = rx2 - 60.0;
            x = lo + (hi - lo) * u; // np.random.uniform(lo, hi) = lo + (hi - lo) * random_sample()
        } else {
            x = fallback;
        }
        rec[SSG_MAP_OFF_GOALS + 2 * (i - 1)] = x;
        rec[SSG_MAP_OFF_GOALS + 2 * (i - 1) + 1] = y;
        if (rw) { rw[48 + 3 * (i - 1)] = y; rw[48 + 3 * (i - 1) + 1] = u; rw[48 + 3 * (i - 1) + 2] = fallback; }
        const double dx = x - spawn_x, dy = y - spawn_y;
        const double d = sqrt(dx * dx + dy * dy);
        if (i == 1 || d < best) { best = d; sgx = x; sgy = y; }
    }
    rec[SSG_MAP_OFF_SPAWN_GOAL] = sgx;
    rec[SSG_MAP_OFF_SPAWN_GOAL + 1] = sgy;
}

} // namespace

// raw: per map 48 + 3*n_goals doubles (generate_world)
__global__ void generate_bank_kernel(uint64_t seed, int n_maps, int n_goals, double width, double height,
                                     double width_frac, double spawn_x, double spawn_y, double *__restrict__ bank,
                                     double *__restrict__ raw)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_maps) return;
    Rng rng;
    rng.key[0] = (uint32_t)seed; rng.key[1] = (uint32_t)(seed >> 32);
    rng.ctr[0] = (uint32_t)m; rng.ctr[1] = 0x57474e00u /* "WGN" domain tag */; rng.ctr[2] = 0; rng.ctr[3] = 0;
    rng.have = 0;
    generate_world(rng, n_goals, width, height, width_frac, spawn_x, spawn_y, bank + (size_t)m * SSG_MAP_STRIDE,
                   raw ? raw + (size_t)m * (48 + 3 * n_goals) : nullptr);
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

__global__ void refill_gen_kernel(const DevCfg c, uint64_t seed, double width_frac, const unsigned long long *__restrict__ queue,
                                  const unsigned *__restrict__ count, double *__restrict__ bank, double *__restrict__ raw)
{
    const unsigned n = min(*count, (unsigned)c.n_envs * (unsigned)c.map_ring);
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
                       raw ? raw + slot * (size_t)(48 + 3 * c.n_goals) : nullptr);
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
